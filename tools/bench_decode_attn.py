#!/usr/bin/env python3
"""Micro-benchmark of sp_decode_attention on the headline workload's shapes (no model around it).

  python tools/bench_decode_attn.py [--bs 256] [--ctx uniform|N] [--chunks 128,256,512] [--ranges -1|0|N] [--iters 20]
Prints one line per chunk: avg launch time (HIP events on the launch stream), algorithmic GB/s.  --ranges: the plan's
range geometry as HipAttnBackend uses it (-1, the default: sp_decode_ranges() pieces where the range kernel takes the
shape; the split size then only sizes the plan's item section), 0: the (request, split) items at the split size."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratchpad_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--ctx", default="uniform")
    ap.add_argument("--Hq", type=int, default=32)
    ap.add_argument("--Hkv", type=int, default=8)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--chunks", default="128,256,512,1024")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--kv", default="same", choices=["same", "fp8"], help="fp8 = e5m2 byte pool")
    ap.add_argument("--lib", default="", help="diagnostic library under scratchpad_amd/lib")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--no-plan", action="store_true")
    ap.add_argument("--no-fuse", action="store_true", help="(accepted for old command lines: the merge is always a separate launch)")
    ap.add_argument("--interleave", action="store_true",
                    help="one [P+1, 2, Hkv, D] arena: a token's K and V rows are adjacent (pool.py's layout)")
    ap.add_argument("--slots", type=int, default=0,
                    help="work items the launch covers (graph replay covers max(1024, 8 bs) + bs; default: what the step needs)")
    ap.add_argument("--gemm", action="store_true", help="interleave a bf16 GEMM between launches (as in a model)")
    ap.add_argument("--ranges", type=int, default=-1, help="pieces of the range geometry (-1: sp_decode_ranges(), 0: none)")
    ap.add_argument("--items", action="store_true",
                    help="build the (request, split) item section beside the range section (until round 6 every plan carried it; "
                         "HipAttnBackend now builds it only for models with a launch that reads it)")
    a = ap.parse_args()
    if a.lib:
        _native._LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratchpad_amd", "lib", a.lib)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    if a.ctx == "uniform":
        ctx = torch.randint(128, 4097, (a.bs,), generator=g)
    elif ":" in a.ctx:                                   # "lo:hi" = uniform integers in [lo, hi]
        lo, hi = (int(x) for x in a.ctx.split(":"))
        ctx = torch.randint(lo, hi + 1, (a.bs,), generator=g)
    else:
        ctx = torch.full((a.bs,), int(a.ctx))
    total = int(ctx.sum())
    P = total + 1024
    if a.interleave:
        arena = torch.empty(P + 1, 2, a.Hkv, a.D, dtype=dt, device=dev).normal_(0, 0.5)
        if a.kv == "fp8":
            arena = arena.to(torch.float8_e5m2).view(torch.uint8)
        kb, vb = arena[:, 0], arena[:, 1]
    else:
        kb = torch.empty(P + 1, a.Hkv, a.D, dtype=dt, device=dev).normal_(0, 0.5)
        vb = torch.empty(P + 1, a.Hkv, a.D, dtype=dt, device=dev).normal_(0, 0.5)
        if a.kv == "fp8":
            kb = kb.to(torch.float8_e5m2).view(torch.uint8)
            vb = vb.to(torch.float8_e5m2).view(torch.uint8)
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(a.bs, int(ctx.max()) + 8, dtype=torch.int32)
    off = 0
    for b in range(a.bs):
        n = int(ctx[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(dev)
    req = torch.arange(a.bs, dtype=torch.int32, device=dev)
    seq = ctx.to(torch.int32).to(dev)
    q = torch.randn(a.bs, a.Hq, a.D, device=dev).to(dt)
    o = torch.empty_like(q)
    eb = q.element_size()
    kv_eb = 1 if a.kv == "fp8" else eb
    alg = total * 2 * a.Hkv * a.D * kv_eb + 2 * a.bs * a.Hq * a.D * eb + 4 * total
    max_len = int(ctx.max())
    ref = None
    ranges = a.ranges if a.ranges >= 0 else _native.decode_ranges(a.Hq, a.Hkv, a.D, dt, kb.dtype)
    if a.no_plan:
        ranges = 0
    print(f"range geometry: {ranges} pieces" if ranges else "(request, split) items", flush=True)
    for chunk in [int(c) for c in a.chunks.split(",")]:
        slots = a.slots or None
        if ranges > 0 and not a.items and not a.slots:
            slots = 0                                    # the range section alone, as the backend builds it for a Llama shape
        ws = torch.empty(_native.decode_workspace_bytes(a.bs, a.Hq, a.D, max_len, chunk, slots, ranges), dtype=torch.uint8, device=dev)
        plan = None
        if not a.no_plan:
            plan = torch.empty(_native.decode_plan_bytes(a.bs, max_len, chunk, slots, ranges) // 4, dtype=torch.int32, device=dev)
            _native.decode_plan(plan, seq, max_len, chunk, slots, ranges)
        run = lambda: _native.decode_attention(o, q, kb, vb, r2t, req, seq, a.D ** -0.5, 0.0, max_len, chunk, ws, None, plan,
                                               max_slots=slots, ranges=ranges)
        for _ in range(a.warmup):
            run()
        torch.cuda.synchronize()
        evs = []
        if a.gemm:
            ga = torch.randn(256, 4096, device=dev, dtype=torch.bfloat16)
            gw = torch.randn(28672, 4096, device=dev, dtype=torch.bfloat16)
        for _ in range(a.iters):
            if a.gemm:
                torch.nn.functional.linear(ga, gw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in evs)
        med, best = ms[len(ms) // 2], ms[0]
        msg = f"chunk {chunk:5d}: median {med * 1e3:8.1f} us  best {best * 1e3:8.1f} us  {alg / med / 1e6:8.1f} GB/s (median)  alg bytes {alg / 1e9:.3f} GB"
        if a.check:
            if ref is None:
                ref = o.float().clone()
            msg += f"  max|d| vs first {float((o.float() - ref).abs().max()):.2e}"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
