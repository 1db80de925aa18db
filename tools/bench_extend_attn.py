#!/usr/bin/env python3
"""Micro-benchmark of sp_extend_attention (ragged prefill) on config-3 shapes:
bs prompts with lengths U[128,4096] seed 0, optional shared cached prefix.
  python tools/bench_extend_attn.py [--bs 64] [--prefix 0] [--Hq 32 --Hkv 8 --D 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratchpad_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--prefix", type=int, default=0)
    ap.add_argument("--len", default="uniform")
    ap.add_argument("--Hq", type=int, default=32)
    ap.add_argument("--Hkv", type=int, default=8)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--waves", default="1,0", help="comma list: 1 = with the sp_extend_plan work list, 0 = without")
    ap.add_argument("--dma", default="1", help="comma list of sp_debug_set('extend_dma') values to compare")
    ap.add_argument("--lib", default="", help="alternative library file under scratchpad_amd/lib (diagnostic builds)")
    ap.add_argument("--rounds", type=int, default=3, help="interleaved timing rounds per variant")
    ap.add_argument("--defer-x10", type=int, default=-1, help="sp_debug_set('extend_defer_x10'): -1 = shipped")
    a = ap.parse_args()
    if a.lib:
        _native._LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratchpad_amd", "lib", a.lib)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    ext = (torch.randint(128, 4097, (a.bs,), generator=g) if a.len == "uniform"
           else torch.full((a.bs,), int(a.len)))
    pre = torch.full((a.bs,), a.prefix)
    seq = ext + pre
    total = int(seq.sum())
    P = total + 64
    kb = torch.empty(P + 1, a.Hkv, a.D, dtype=dt, device=dev).normal_(0, 0.5)
    vb = torch.empty(P + 1, a.Hkv, a.D, dtype=dt, device=dev).normal_(0, 0.5)
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(a.bs, int(seq.max()) + 8, dtype=torch.int32)
    off = 0
    for b in range(a.bs):
        n = int(seq[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(dev)
    T = int(ext.sum())
    q = torch.randn(T, a.Hq, a.D, device=dev).to(dt)
    o = torch.empty_like(q)
    req = torch.arange(a.bs, device=dev)
    ext_d = ext.to(torch.int32).to(dev)
    start = torch.zeros(a.bs, dtype=torch.int32, device=dev)
    start[1:] = torch.cumsum(ext_d[:-1], 0)
    ws = torch.empty(_native.extend_workspace_bytes(T, a.bs, a.Hq, a.D, dt), dtype=torch.uint8, device=dev)
    seq_d = seq.to(dev)
    plan = _native.extend_plan(ext_d, seq_d, T, a.Hq, a.Hkv, True)
    use_plan = [None]
    run = lambda: _native.extend_attention(o, q, kb, vb, r2t, req, seq_d, ext_d, start, a.D ** -0.5, 0.0, True,
                                           int(ext.max()), int(seq.max()), ws, plan=use_plan[0])
    flops = 4 * a.Hq * a.D * float(((ext.double() ** 2) / 2 + ext.double() * pre.double()).sum())
    _native.debug_set("extend_defer_x10", a.defer_x10)
    variants = [(int(w), int(x)) for w in a.waves.split(",") for x in a.dma.split(",")]
    times = {w: [] for w in variants}
    outs = []
    for w in variants:
        use_plan[0] = plan if w[0] else None
        _native.debug_set("extend_dma", w[1])
        o.zero_()
        run()
        outs.append(o.clone())
    torch.cuda.synchronize()
    for w, x in zip(variants[1:], outs[1:]):   # the variants differ in schedule only: same bits expected
        print(f"variant {w} vs {variants[0]}: max |diff| = {(x.float() - outs[0].float()).abs().max().item():.3e}", flush=True)
    for _ in range(a.rounds):          # interleaved rounds in one process (same clocks, same device)
        for w in variants:
            use_plan[0] = plan if w[0] else None
            _native.debug_set("extend_dma", w[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[w].append(e0.elapsed_time(e1) / a.iters)
    for w in variants:
        ms = sorted(times[w])[len(times[w]) // 2]
        print(f"extend bs={a.bs} tokens={T} prefix={a.prefix} {a.dtype} plan={w[0]} dma={w[1]} defer={a.defer_x10}: {ms:.3f} ms (best {min(times[w]):.3f})  "
              f"{flops / ms / 1e9:.1f} TFLOP/s (causal flops {flops / 1e12:.2f} T)", flush=True)


if __name__ == "__main__":
    main()
