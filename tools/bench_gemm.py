#!/usr/bin/env python3
"""Time the decode-step GEMM shapes (M=bs) through torch (hipBLASLt / rocBLAS / TunableOp)."""
import os, sys, time
import torch
import torch.nn.functional as F

def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336), "lm_head": (128256, 4096)}
    dev = "cuda"
    total = 0.0
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(4 if name != "lm_head" else 1)]
        for i in range(10):
            F.linear(x, ws[i % len(ws)])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        e0.record()
        for i in range(n):
            F.linear(x, ws[i % len(ws)])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        gb = N * K * 2 / 1e9
        print(f"{name:8s} M={M} N={N} K={K}: {us:8.1f} us  weights {gb / (us * 1e-6) / 1e3:6.2f} TB/s  {2 * M * N * K / (us * 1e-6) / 1e12:7.1f} TFLOP/s", flush=True)
        if name != "lm_head":
            total += us
    print(f"per-layer GEMM total {total:.1f} us", flush=True)

if __name__ == "__main__":
    main()
