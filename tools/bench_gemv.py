#!/usr/bin/env python3
"""Weight-streaming rate of the small-batch projections: sp_gemm_skinny vs the library GEMM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scratchpad_amd import _native  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336),
              "lm_head": (128256, 4096)}
    for M in (1, 8, 16):
        tot_a = tot_b = 0.0
        for name, (N, K) in shapes.items():
            x = torch.randn(M, K, device="cuda").bfloat16()
            ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(1 if name == "lm_head" else 6)]
            it = [0]

            def nxt():
                it[0] += 1
                return ws[it[0] % len(ws)]
            _native.debug_set("skinny_nt", 0)
            a = timeit(lambda: _native.linear(x, nxt()))
            _native.debug_set("skinny_nt", 1)
            a0 = timeit(lambda: _native.linear(x, nxt()))          # non-temporal weight loads, for comparison (not shipped)
            _native.debug_set("skinny_nt", 0)
            _native.debug_set("skinny_unroll16", 1)
            a16 = timeit(lambda: _native.linear(x, nxt()))         # 16 k-steps in flight per wave where the share allows
            _native.debug_set("skinny_unroll16", 0)
            b = timeit(lambda: torch.nn.functional.linear(x, nxt()))
            gb = N * K * 2 / 1e9
            print(f"M={M:2d} {name:8s} N={N:6d} K={K:5d}: skinny {a:7.1f} us ({gb / a * 1e3:5.2f} TB/s; nt loads {a0:6.1f}; unroll 16 {a16:6.1f})   "
                  f"library {b:7.1f} us ({gb / b * 1e3:5.2f} TB/s)", flush=True)
            if name != "lm_head":
                tot_a += a
                tot_b += b
        print(f"M={M:2d} per-layer total: skinny {tot_a:.1f} us, library {tot_b:.1f} us", flush=True)
        # gate_up_proj + act_fn (LlamaMLP): SiluAndMul in the skinny projection's epilogue vs projection + activation launch
        N, K = shapes["gate_up"]
        x = torch.randn(M, K, device="cuda").bfloat16()
        ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(6)]
        it = [0]

        def w_():
            it[0] += 1
            return ws[it[0] % len(ws)]
        fused = timeit(lambda: _native.linear_silu_mul(x, w_(), any_rows=True))
        two = timeit(lambda: _native.silu_and_mul(_native.linear(x, w_())))
        print(f"M={M:2d} gate_up + silu_mul: fused skinny {fused:7.1f} us   two launches {two:7.1f} us", flush=True)

if __name__ == "__main__":
    main()
