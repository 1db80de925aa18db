#!/usr/bin/env python3
"""A/B of decode-attention builds / switches in ONE process on one box (boxes differ by ~2.5 % in this kernel).

Every LIBSPEC named on the command line - FILE[@key=value,...] under scratchpad_amd/lib, the switches applied through
sp_debug_set after loading - runs the SAME decode launch in interleaved rounds: the shipped form (plan + separate
merge launch, K/V as the two strided views of one interleaved [P+1, 2, Hkv, D] arena as MHATokenToKVPool makes them,
non-temporal gathers by default).  The pseudo-switch ranges=N (N = -1: what sp_decode_ranges() asks for) gives the plan
and the launch the range geometry (ABI 8); without it the launch uses the (request, split) items at --chunk.  Outputs
are compared with the first library's (bit for bit within one geometry; the geometries differ in the last bits).
NOTE: dlopen hands out ONE copy of a library file however often it is named - to compare switches of one build, copy
the file under a second name.

  python tools/ab_decode.py libscratchpad_hip.so libscratchpad_hip_x.so [--shape headline|hkv1|bs64|...] [--chunk 768]
  shapes: headline = bs 256, Hq 32 / Hkv 8, contexts U[128,4096] seed 0;  hkv1 = bs 128, Hq 8 / Hkv 1 (config 4's rank)

(Replaces the one-off tools/r4_*.sh drivers of round 4; their measurements are in profiles/r04_decode_variants.txt.)"""
import argparse
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"headline": (256, 32, 8, "128:4096"), "hkv1": (128, 8, 1, "128:4096"), "bs128": (128, 32, 8, "128:4096"),
          "bs64": (64, 32, 8, "128:4096"), "bs8": (8, 32, 8, "1024"), "bs1": (1, 32, 8, "1024"),
          "ctx1024": (256, 32, 8, "1024"), "ctx4096": (256, 32, 8, "4096"),
          # uniform contexts that deal out evenly over 768 resident workgroups at split size 768 (2 and 3 units each)
          "ctx2304": (256, 32, 8, "2304"), "ctx3456": (256, 32, 8, "3456"),
          # a tensor-parallel rank's one or two kv heads at larger batches
          "hkv1_bs256": (256, 8, 1, "128:4096"), "hkv1_ctx4096": (256, 8, 1, "4096"), "hkv2": (256, 16, 2, "128:4096")}


def load_native(libspec, tag):
    libfile, _, switches = libspec.partition("@")
    spec = importlib.util.spec_from_file_location(f"sp_native_{tag}", os.path.join(ROOT, "scratchpad_amd", "_native.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m._LIB_PATH = os.path.join(ROOT, "scratchpad_amd", "lib", libfile)
    m.load()
    ranges = 0
    for kv in filter(None, switches.split(",")):
        k, v = kv.split("=")
        if k == "ranges":
            ranges = int(v)
        else:
            m.debug_set(k, int(v))
    return m, ranges


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--shape", default="headline", choices=sorted(SHAPES))
    ap.add_argument("--chunk", type=int, default=768, help="split size (attention.py ships 768 on these shapes)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--kv", default="same", choices=["same", "fp8"])
    ap.add_argument("--graph-slots", action="store_true", help="the launch covers max(1024, 8 bs) + bs items, as under graph replay")
    ap.add_argument("--sequential-slots", action="store_true",
                    help="requests take consecutive KV slots (default: a random permutation of the pool, page_size 1)")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    bs, Hq, Hkv, ctxspec = SHAPES[a.shape]
    D, dev = 128, "cuda"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    g = torch.Generator().manual_seed(0)
    if ":" in ctxspec:
        lo, hi = (int(x) for x in ctxspec.split(":"))
        ctx = torch.randint(lo, hi + 1, (bs,), generator=g)
    else:
        ctx = torch.full((bs,), int(ctxspec))
    total, max_len = int(ctx.sum()), int(ctx.max())
    P = total + 1024
    arena = torch.empty(P + 1, 2, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    if a.kv == "fp8":
        arena = arena.to(torch.float8_e5m2).view(torch.uint8)
    kb, vb = arena[:, 0], arena[:, 1]
    perm = ((torch.arange(P) if a.sequential_slots else torch.randperm(P, generator=g)) + 1).to(torch.int32)
    r2t = torch.zeros(bs, max_len + 8, dtype=torch.int32)
    off = 0
    for b in range(bs):
        n = int(ctx[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(dev)
    req = torch.arange(bs, dtype=torch.int32, device=dev)
    seq = ctx.to(torch.int32).to(dev)
    q = torch.randn(bs, Hq, D, device=dev).to(dt)
    eb = q.element_size()
    alg = total * 2 * Hkv * D * (1 if a.kv == "fp8" else eb) + 2 * bs * Hq * D * eb + 4 * total
    slots = (max(1024, 8 * bs) + bs) if a.graph_slots else None
    loaded = [load_native(lib, i) for i, lib in enumerate(a.libs)]
    nats = [n for n, _ in loaded]
    rngs = [n.decode_ranges(Hq, Hkv, D, dt, arena.dtype) if r < 0 else r for n, r in loaded]
    ws = torch.empty(max(n.decode_workspace_bytes(bs, Hq, D, max_len, a.chunk, slots, ranges=r) for n, r in zip(nats, rngs)),
                     dtype=torch.uint8, device=dev)
    plans, outs = [], []
    for n, r in zip(nats, rngs):
        pl = torch.empty(n.decode_plan_bytes(bs, max_len, a.chunk, slots, ranges=r) // 4, dtype=torch.int32, device=dev)
        n.decode_plan(pl, seq, max_len, a.chunk, slots, ranges=r)
        plans.append(pl)
        outs.append(torch.full_like(q, float("nan")))

    def run(i):
        nats[i].decode_attention(outs[i], q, kb, vb, r2t, req, seq, D ** -0.5, 0.0, max_len, a.chunk, ws, None, plans[i],
                                 max_slots=slots, ranges=rngs[i])
    for i in range(len(nats)):
        run(i)
    torch.cuda.synchronize()
    print(f"shape {a.shape}: bs {bs} Hq {Hq} Hkv {Hkv} D {D} {a.dtype} kv {a.kv} chunk {a.chunk} sum {total} "
          f"plan + merge launch, {'graph slots' if a.graph_slots else 'exact slots'}; "
          f"alg bytes {alg / 1e9:.3f} GB", flush=True)
    for i in range(1, len(nats)):
        same = torch.equal(outs[i], outs[0])
        err = float((outs[i].float() - outs[0].float()).abs().max())
        print(f"{a.libs[i]} (ranges {rngs[i]}) vs {a.libs[0]} (ranges {rngs[0]}): bit-identical={same} max|diff|={err:.3g} "
              f"finite={bool(torch.isfinite(outs[i].float()).all())}", flush=True)
    times = [[] for _ in nats]
    for r in range(a.rounds + 1):
        for i in range(len(nats)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            if r:                                        # round 0 warms up
                times[i].append(e0.elapsed_time(e1) / a.iters * 1e3)
    for i, lib in enumerate(a.libs):
        t = sorted(times[i])
        med = t[len(t) // 2]
        print(f"{lib:50s} median {med:8.1f} us  min {t[0]:8.1f}  max {t[-1]:8.1f}  {alg / med / 1e6:7.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
