#!/usr/bin/env python3
"""Debug aid: the 4 x 64-row extend kernel against the shipped one on small cases, error pattern by row / head / d."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_extend import load_native  # noqa: E402


def case(nat_a, nat_b, lens, prefix, seed=0):
    Hq, Hkv, D, dt, dev = 32, 8, 128, torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(seed)
    ext = torch.tensor(lens)
    pre = torch.full((len(lens),), prefix)
    seq = ext + pre
    P = int(seq.sum()) + 64
    kb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    vb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(len(lens), int(seq.max()) + 8, dtype=torch.int32)
    off = 0
    for b in range(len(lens)):
        n = int(seq[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(dev)
    T = int(ext.sum())
    q = torch.randn(T, Hq, D, device=dev).to(dt)
    req = torch.arange(len(lens), device=dev)
    ext_d = ext.to(torch.int32).to(dev)
    start = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    start[1:] = torch.cumsum(ext_d[:-1], 0)
    seq_d = seq.to(dev)
    outs = []
    for nat in (nat_a, nat_b):
        ws = torch.empty(nat.extend_workspace_bytes(T, len(lens), Hq, D, dt), dtype=torch.uint8, device=dev)
        plan = nat.extend_plan(ext_d, seq_d, T, Hq, Hkv, True)
        o = torch.full_like(q, float("nan"))
        nat.extend_attention(o, q, kb, vb, r2t, req, seq_d, ext_d, start, D ** -0.5, 0.0, True, int(ext.max()),
                             int(seq.max()), ws, plan=plan)
        torch.cuda.synchronize()
        outs.append(o.float())
    nan_rows = torch.isnan(outs[1]).any(dim=2).any(dim=1).nonzero().flatten().tolist()
    if nan_rows:
        nan_heads = torch.isnan(outs[1]).any(dim=2).any(dim=0).nonzero().flatten().tolist()
        print(f"  NaN in {len(nan_rows)} of {T} rows: {nan_rows[:24]}...; heads {nan_heads[:12]}; total NaN {int(torch.isnan(outs[1]).sum())} of {outs[1].numel()}")
    d = (outs[1] - outs[0]).abs()
    print(f"lens={lens} prefix={prefix}: max diff {d.max().item():.3e} finite={bool(torch.isfinite(outs[1]).all())}", flush=True)
    if d.max() > 2e-2:
        rows = d.amax(dim=(1, 2))
        bad = (rows > 2e-2).nonzero().flatten().tolist()
        print("  bad rows:", bad[:40], "..." if len(bad) > 40 else "", f"({len(bad)} of {T})")
        print("  by head:", [f"{x:.2f}" for x in d.amax(dim=(0, 2)).tolist()])
        print("  by d block of 8:", [f"{x:.2f}" for x in d.amax(dim=(0, 1)).view(16, 8).amax(1).tolist()])
        r = bad[0]
        print("  row", r, "head0 ref", [f"{x:.3f}" for x in outs[0][r, 0, :8].tolist()], "got", [f"{x:.3f}" for x in outs[1][r, 0, :8].tolist()])


def main():
    a = load_native(sys.argv[1], 0)
    b = load_native(sys.argv[2], 1)
    for lens, prefix in (([64], 0), ([32], 0), ([128], 0), ([200], 0), ([64], 64), ([64], 100), ([300, 70], 0), ([640], 0), ([1000, 3, 129], 37)):
        case(a, b, lens, prefix)


if __name__ == "__main__":
    main()
