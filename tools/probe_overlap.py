#!/usr/bin/env python3
"""Round 6, the gated experiment: can the library's projections run UNDER the attention stream?

No product change: this probe builds, on one MI355X, the two kinds of per-layer work of the headline decode step
(Llama-3-8B shapes, bf16, contexts U[128, 4096] seed 0, slots randomly permuted, interleaved K|V arena) -

  attention   sp_decode_plan once + per layer sp_decode_attention (range kernel + merge) on that layer's own KV arena
  projections the four GEMMs of a layer (qkv 6144x4096, o 4096x4096, gate|up 28672x4096, down 4096x14336) through
              torch.mm = hipBLASLt, every layer its own weights (L layers are cycled: nothing is found in a cache)

- captures them into HIP graphs and reports microseconds PER LAYER (HIP events around graph replays, median):

  stage A (the verdict's three numbers, for HALF the batch = 128 requests, at 1 and at 2 workgroups per CU):
      t_attn   the attention chain alone           t_gemm   the projection chain alone (M = 128)
      t_both   both chains in one graph on two streams (fork at the start, join at the end)
      gate:    t_both <= 0.80 x (t_attn + t_gemm)

  the figure stage B would have to beat, measured directly in the same process:
      full     one stream, per layer: attention(256 requests) + projections(M = 256)           [today's step]
      serial   one stream, per layer: attn(A) proj(A) attn(B) proj(B), A / B = the two halves   [micro-batching alone]
      overlap  two streams: chain A = attn, proj, attn, ...; chain B = proj, attn, proj, ... (half a layer out of phase),
               independent until the end - the dependency structure a two-micro-batch step has
      stage B pays only if overlap < full by more than the 5 % the verdict asks of bench.py.

  python tools/probe_overlap.py [--layers 8] [--reps 20] [--out gpurun_out/r6/probe_overlap.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratchpad_amd import _native  # noqa: E402

HQ, HKV, D, HID, INTER = 32, 8, 128, 4096, 14336
DT = torch.bfloat16
DEV = "cuda"


class Attention:
    """decode attention of `rows` (a slice of the 256-request batch) over `layers` KV arenas, `ranges` pieces"""

    def __init__(self, arenas, r2t, ctx, rows, ranges):
        self.arenas = arenas
        self.bs = len(rows)
        self.r2t = r2t
        self.req = torch.tensor(rows, dtype=torch.int32, device=DEV)
        self.seq = ctx[rows].to(torch.int32).to(DEV)
        self.tokens = int(ctx[rows].sum())
        self.max_len = int(ctx.max())
        self.ranges = ranges
        self.q = torch.randn(self.bs, HQ, D, device=DEV).to(DT)
        self.o = torch.empty_like(self.q)
        self.ws = torch.empty(_native.decode_workspace_bytes(self.bs, HQ, D, self.max_len, 64, 0, ranges), dtype=torch.uint8, device=DEV)
        self.plan = torch.empty(_native.decode_plan_bytes(self.bs, self.max_len, 64, 0, ranges) // 4, dtype=torch.int32, device=DEV)
        _native.decode_plan(self.plan, self.seq, self.max_len, 64, 0, ranges)      # (range section alone, as the backend builds it)
        self.alg_bytes = self.tokens * 2 * HKV * D * 2 + 2 * self.bs * HQ * D * 2 + 4 * self.tokens

    def __call__(self, layer):
        a = self.arenas[layer]
        _native.decode_attention(self.o, self.q, a[:, 0], a[:, 1], self.r2t, self.req, self.seq, D ** -0.5, 0.0, self.max_len,
                                 64, self.ws, None, self.plan, max_slots=0, ranges=self.ranges)


class Projections:
    """the four projections of a layer at M rows through the library (rows handed over as _native.linear does)"""
    SHAPES = ((HID + 2 * HKV * D, HID), (HID, HID), (2 * INTER, HID), (HID, INTER))      # (N, K): qkv, o, gate|up, down

    def __init__(self, weights, M):
        self.w = weights
        self.M = M
        self.x, self.y = [], []
        for N, K in self.SHAPES:
            m = _native.library_rows(M, N, K)
            self.x.append((torch.randn(m, K, device=DEV) * 0.1).to(DT))
            self.y.append(torch.empty(m, N, dtype=DT, device=DEV))
        self.weight_bytes = sum(N * K * 2 for N, K in self.SHAPES)

    def __call__(self, layer):
        for i, w in enumerate(self.w[layer]):
            torch.mm(self.x[i], w.t(), out=self.y[i])


def capture(build, side=None):
    """build(main_stream, side_stream) issues the work; returns a replayable graph.  `side`: a second stream the build
    may fork to (it must wait on main before its first launch and main must wait on it after the last)."""
    main = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        build(main, side)                      # warm-up (library heuristics, lazy module loads) outside the capture
        main.synchronize()
        if side is not None:
            side.synchronize()
        with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
            build(main, side)
    torch.cuda.synchronize()
    return g


def time_graph(g, reps, layers):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / layers)
    ts.sort()
    return ts[len(ts) // 2]


def chain(stream_work):
    """one stream: the callables in order"""
    def build(main, side):
        for f in stream_work:
            f()
    return build


def two_chains(work_a, work_b):
    def build(main, side):
        side.wait_stream(main)
        for f in work_a:
            f()
        with torch.cuda.stream(side):
            for f in work_b:
                f()
        main.wait_stream(side)
    return build


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--out", default="")
    ap.add_argument("--pieces", default="", help="comma-separated piece counts for the attention launches (default: one and two "
                                                 "workgroups per CU = sp_decode_ranges() / 2 and sp_decode_ranges())")
    a = ap.parse_args()
    L = a.layers
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    ctx = torch.randint(128, 4097, (a.bs,), generator=g)              # bench.py's headline contexts
    total = int(ctx.sum())
    P = total + 1024
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(a.bs, int(ctx.max()) + 8, dtype=torch.int32)
    off = 0
    for b in range(a.bs):
        n = int(ctx[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(DEV)
    arenas = [torch.empty(P + 1, 2, HKV, D, dtype=DT, device=DEV).normal_(0, 0.5) for _ in range(L)]
    weights = [[(torch.randn(N, K, device=DEV) * 0.02).to(DT) for N, K in Projections.SHAPES] for _ in range(L)]
    half = a.bs // 2
    rows_all, rows_a, rows_b = list(range(a.bs)), list(range(half)), list(range(half, a.bs))
    auto = _native.decode_ranges(HQ, HKV, D, DT)                      # two workgroups per CU
    side = torch.cuda.Stream()
    wg = lambda r: f"{r * HKV / 4 / 256:g}wg"                          # workgroups per CU of a launch of r pieces (MI355X: 256 CUs)
    variants = [(wg(int(r)), int(r)) for r in a.pieces.split(",")] if a.pieces else [("1wg", auto // 2), ("2wg", auto)]
    res = {"layers": L, "reps": a.reps, "bs": a.bs, "tokens": total, "ranges_2wg": auto, "ranges_1wg": auto // 2,
           "device": torch.cuda.get_device_name(0), "library_rows": {}}
    print(f"{res['device']}: {a.bs} requests, {total} context tokens, {L} layers cycled, "
          f"sp_decode_ranges = {auto} (two workgroups per CU)", flush=True)

    # ---- stage A: half the batch, attention at one and at two workgroups per CU
    proj_half = Projections(weights, half)
    t_gemm = time_graph(capture(chain([lambda l=l: proj_half(l) for l in range(L)])), a.reps, L)
    res["stage_a"] = {"t_gemm_us": t_gemm, "gemm_weight_GB": proj_half.weight_bytes / 1e9,
                      "gemm_TBps": proj_half.weight_bytes / t_gemm / 1e6}
    print(f"stage A  projections alone, M = {half}: {t_gemm:7.1f} us / layer  ({proj_half.weight_bytes / t_gemm / 1e6:.2f} TB/s of weights)", flush=True)
    for name, ranges in variants:
        att = Attention(arenas, r2t, ctx, rows_a, ranges)
        t_attn = time_graph(capture(chain([lambda l=l: att(l) for l in range(L)])), a.reps, L)
        both = capture(two_chains([lambda l=l: att(l) for l in range(L)], [lambda l=l: proj_half(l) for l in range(L)]), side)
        t_both = time_graph(both, a.reps, L)
        ratio = t_both / (t_attn + t_gemm)
        res["stage_a"][name] = {"ranges": ranges, "t_attn_us": t_attn, "attn_TBps": att.alg_bytes / t_attn / 1e6,
                                "t_both_us": t_both, "ratio_to_sum": ratio, "gate_0.80": ratio <= 0.80}
        print(f"stage A  attention alone, {half} requests, {ranges} pieces ({name[:-2]} workgroup(s) per CU): {t_attn:7.1f} us / layer "
              f"({att.alg_bytes / t_attn / 1e6:.2f} TB/s);  both on two streams: {t_both:7.1f} us = {ratio:.3f} x the sum "
              f"-> gate (<= 0.80) {'PASSES' if ratio <= 0.80 else 'FAILS'}", flush=True)

    # ---- what stage B would have to beat
    att_full = Attention(arenas, r2t, ctx, rows_all, auto)
    proj_full = Projections(weights, a.bs)
    work = []
    for l in range(L):
        work += [lambda l=l: att_full(l), lambda l=l: proj_full(l)]
    t_full = time_graph(capture(chain(work)), a.reps, L)
    res["full_us"] = t_full
    print(f"full     one stream, attention({a.bs}) + projections(M = {a.bs}): {t_full:7.1f} us / layer", flush=True)
    for name, ranges in variants:
        att_a, att_b = Attention(arenas, r2t, ctx, rows_a, ranges), Attention(arenas, r2t, ctx, rows_b, ranges)
        proj_b = Projections(weights, half)
        serial, wa, wb = [], [], []
        for l in range(L):
            serial += [lambda l=l: att_a(l), lambda l=l: proj_half(l), lambda l=l: att_b(l), lambda l=l: proj_b(l)]
            wa += [lambda l=l: att_a(l), lambda l=l: proj_half(l)]
            wb += [lambda l=l: proj_b(l), lambda l=l: att_b(l)]               # half a layer out of phase
        t_serial = time_graph(capture(chain(serial)), a.reps, L)
        t_overlap = time_graph(capture(two_chains(wa, wb), side), a.reps, L)
        res[f"micro_{name}"] = {"ranges": ranges, "serial_us": t_serial, "overlap_us": t_overlap,
                                "overlap_vs_full": t_overlap / t_full, "serial_vs_full": t_serial / t_full}
        print(f"micro    two halves, {ranges} pieces each: serial {t_serial:7.1f} us / layer ({t_serial / t_full:.3f} x full), "
              f"two streams {t_overlap:7.1f} us / layer ({t_overlap / t_full:.3f} x full)", flush=True)
    res["library_rows"] = _native.library_rows_report()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
