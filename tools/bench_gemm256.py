#!/usr/bin/env python3
"""M = 256 projections of the Llama-3-8B decode step: what the BLAS libraries offer per shape.
  python tools/bench_gemm256.py [--M 256]
Times F.linear (x [M,K] bf16, W [N,K] bf16) through hipBLASLt and rocBLAS, with W pre-transposed, and as a
hand-made split-K (partials in fp32 where torch allows it).  Cold-cache timing: the weights of 8 different
layers are cycled so that no launch finds its weights in L2 / Infinity Cache."""
import argparse
import torch
import torch.nn.functional as F


def timeit(fn, n=32):
    """device time per call: n calls captured into one HIP graph (eager launches are host-bound at ~20 us)"""
    for _ in range(2):
        fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(8):
            fn(i)
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for i in range(n):
                fn(i)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=256)
    a = ap.parse_args()
    dev, dt = "cuda", torch.bfloat16
    shapes = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)]
    L = 8
    for name, N, K in shapes:
        x = torch.randn(a.M, K, device=dev, dtype=dt) * 0.1
        Ws = [torch.randn(N, K, device=dev, dtype=dt) * 0.02 for _ in range(L)]
        WTs = [w.t().contiguous() for w in Ws]      # [K, N]
        res = {}
        for lib in ("cublaslt", "cublas"):
            try:
                torch.backends.cuda.preferred_blas_library(lib)
            except Exception as e:  # noqa: BLE001
                res[lib] = f"n/a ({e})"
                continue
            res[f"{lib} linear"] = timeit(lambda i: F.linear(x, Ws[i % L]))
            res[f"{lib} x@WT"] = timeit(lambda i: torch.mm(x, WTs[i % L]))
        torch.backends.cuda.preferred_blas_library("cublaslt")
        for S in (2, 4, 8):
            if K % S:
                continue
            kc = K // S
            out = torch.empty(a.M, N, device=dev, dtype=dt)

            def splitk(i, S=S, kc=kc, out=out):
                w = Ws[i % L]
                torch.mm(x[:, :kc], w[:, :kc].t(), out=out)
                for s in range(1, S):
                    out.addmm_(x[:, s * kc:(s + 1) * kc], w[:, s * kc:(s + 1) * kc].t())
            res[f"splitK{S} (bf16 partial sums)"] = timeit(splitk)
        for S in (2, 4):
            if N % S:
                continue
            nc = N // S
            res[f"splitN{S}"] = timeit(lambda i, S=S, nc=nc: [F.linear(x, Ws[i % L][s * nc:(s + 1) * nc]) for s in range(S)])
        wb = N * K * 2
        print(f"M={a.M} {name:8s} N={N:6d} K={K:6d}: " + "  ".join(
            f"{k} {v:.1f}us ({wb / v / 1e6:.2f} TB/s)" if isinstance(v, float) else f"{k} {v}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
