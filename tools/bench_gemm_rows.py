#!/usr/bin/env python3
"""F.linear time of the four decode projections at every HIP-graph batch bucket (rows 16 .. 256 step 8), device
time under graph replay with cold weights: hipBLASLt's kernel choice is erratic in the row count (down_proj,
N 4096 x K 14336: 48 us at 112 rows, 74 at 128, 55 at 144, 102 at 192, 61 at 256), so for some buckets the
cheapest way to run M rows is to hand the library M' > M rows (scratchpad_amd/_native.py:library_rows)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemm256 import timeit  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b-tp8"])
    a = ap.parse_args()
    dev, dt, L = "cuda", torch.bfloat16, 8
    shapes = {"llama3-8b": [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)],
              # per-rank shards at TP = 8 (hidden 8192, 64/8 heads, inter 28672)
              "llama3-70b-tp8": [("qkv", 1280, 8192), ("o", 8192, 1024), ("gate_up", 7168, 8192), ("down", 8192, 3584)]}
    for name, N, K in shapes[a.model]:
        Ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(L)]
        rows = list(range(16, 257, 8)) + list(range(264, 385, 8))
        t = {}
        for M in rows:
            x = torch.randn(M, K, device=dev, dtype=dt) * 0.1
            t[M] = timeit(lambda i: F.linear(x, Ws[i % L]), n=16)
        best = {}
        for M in rows:
            best[M] = min((t[m], m) for m in rows if m >= M)
        print(f"{name} N={N} K={K}: " + " ".join(f"{M}:{t[M]:.0f}" for M in rows), flush=True)
        print(f"   pays (rows -> rows handed to the library, us saved): " +
              " ".join(f"{M}->{best[M][1]}(-{t[M] - best[M][0]:.0f})" for M in rows if t[M] - best[M][0] > 3.0), flush=True)


if __name__ == "__main__":
    main()
