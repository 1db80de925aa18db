#!/usr/bin/env python3
"""Microseconds of sp_decode_plan (one workgroup, once per decode step) on the headline batch: the range section alone -
what HipAttnBackend builds for a Llama shape since round 6 - against the range section + the (request, split) items it
built every step until then.  HIP events over back-to-back launches; also a HIP graph of 100 launches (launch gaps removed).

  python tools/bench_decode_plan.py [--bs 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratchpad_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    ctx = torch.randint(128, 4097, (a.bs,), generator=g)
    seq = ctx.to(torch.int32).cuda()
    max_len, ranges = 8192, _native.decode_ranges(32, 8, 128, torch.bfloat16)
    graph_slots = max(1024, 8 * a.bs) + a.bs
    for name, slots, chunk in (("range section alone (max_slots = 0)", 0, 64),
                               (f"range section + items ({graph_slots} slots, split 768)", graph_slots, 768),
                               ("items alone (ranges = 0)", graph_slots, 768)):
        r = 0 if name.startswith("items alone") else ranges
        plan = torch.empty(_native.decode_plan_bytes(a.bs, max_len, chunk, slots, r) // 4, dtype=torch.int32, device="cuda")
        run = lambda: _native.decode_plan(plan, seq, max_len, chunk, slots, r)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / a.iters * 1e3
        st = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                for _ in range(100):
                    run()
        torch.cuda.synchronize()
        gr.replay()
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"sp_decode_plan bs {a.bs}, {ranges if r else 0} pieces, {name}: {eager:6.1f} us per call back to back, "
              f"{e0.elapsed_time(e1) / 500 * 1e3:6.1f} us inside a graph", flush=True)


if __name__ == "__main__":
    main()
