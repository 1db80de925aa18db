#!/usr/bin/env python3
"""Time the sampler kernels at the headline shape (bs=256, vocab=128256, fp32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scratchpad_amd import _native


def timeit(fn, n=20):
    for _ in range(3):
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
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    vocab = 128256
    g = torch.Generator(device="cuda").manual_seed(0)
    logits = torch.randn(bs, vocab, device="cuda", generator=g) * 3
    probs = torch.softmax(logits, -1)
    ks = torch.full((bs,), 50, dtype=torch.int32, device="cuda")
    kall = torch.full((bs,), 1 << 30, dtype=torch.int32, device="cuda")
    ps = torch.full((bs,), 0.9, device="cuda")
    p1 = torch.full((bs,), 1.0, device="cuda")
    ms = torch.full((bs,), 0.05, device="cuda")
    u = torch.rand(bs, device="cuda", generator=g)
    temps = torch.full((bs, 1), 0.8, device="cuda")
    row_mb = bs * vocab * 4 / 1e6
    print(f"bs={bs} vocab={vocab}: one pass over the rows = {row_mb:.0f} MB")
    print(f"argmax fp32            {timeit(lambda: _native.argmax(logits)):8.1f} us   (torch.argmax {timeit(lambda: torch.argmax(logits, -1)):8.1f} us)")
    scratch = logits.clone()
    print(f"softmax_temperature_   {timeit(lambda: _native.softmax_temperature_(scratch, temps)):8.1f} us   (torch div+softmax {timeit(lambda: torch.softmax(logits / temps, -1)):8.1f} us)")
    print(f"sample k=50 p=.9       {timeit(lambda: _native.top_k_top_p_min_p_sample(probs, ks, ps, None, u)):8.1f} us")
    print(f"sample k=all p=.9 minp {timeit(lambda: _native.top_k_top_p_min_p_sample(probs, kall, ps, ms, u)):8.1f} us")
    print(f"sample keep-all        {timeit(lambda: _native.top_k_top_p_min_p_sample(probs, kall, None, None, u)):8.1f} us")
    print(f"renorm top-p           {timeit(lambda: _native.top_k_top_p_min_p_renorm(probs, None, ps, None)):8.1f} us")

    def torch_ref():
        srt, idx = probs.sort(dim=-1, descending=True)
        cs = torch.cumsum(srt, -1)
        srt[(cs - srt) > ps.view(-1, 1)] = 0
        srt[torch.arange(vocab, device="cuda").view(1, -1) >= ks.view(-1, 1)] = 0
        return torch.gather(idx, 1, torch.multinomial(srt, 1))
    print(f"reference formulation (sort+cumsum+multinomial, torch on GPU) {timeit(torch_ref, 5):8.1f} us")


if __name__ == "__main__":
    main()
