#!/usr/bin/env python3
"""Where a pipelined iteration of the 4 x 64-row extend kernel spends its cycles: runs a -DSP_W64_STAMPS build
(tools/build_w64_variant.sh w64stamps -DSP_W64_STAMPS) on a long-prefix launch and prints cycles per segment and wave."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_extend import load_native  # noqa: E402

SEG = ["gaps 0-7 (S^T kb0)", "gaps 8-15", "gaps 16-23 (S^T kb1)", "gaps 24-31", "gaps 32-39 (PV s0)", "gaps 40-47 (PV s1)",
       "gaps 48-55 (PV s2 + DMA)", "gaps 56-59 (PV s3)", "vmcnt wait + barrier", "gaps 60-63"]


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else "libscratchpad_hip_w64stamps.so@extend_w64=2,extend_w64_persist=0"
    names = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else SEG   # custom --stamp-gaps builds: the gap list
    nat = load_native(lib, 0)
    bs, ln, prefix = (int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (128, 128, 8192)
    Hq, Hkv, D, dt, dev = 32, 8, 128, torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(0)
    ext = torch.full((bs,), ln)
    seq = ext + prefix
    P = int(seq.sum()) + 64
    kb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    vb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = perm[: bs * (prefix + ln)].view(bs, prefix + ln).contiguous().to(dev)
    T = bs * ln
    q = torch.randn(T, Hq, D, device=dev).to(dt)
    req = torch.arange(bs, device=dev)
    ext_d = ext.to(torch.int32).to(dev)
    start = (torch.arange(bs, dtype=torch.int32) * ln).to(dev)
    seq_d = seq.to(dev)
    ws = torch.empty(nat.extend_workspace_bytes(T, bs, Hq, D, dt), dtype=torch.uint8, device=dev)
    plan = nat.extend_plan(ext_d, seq_d, T, Hq, Hkv, True)
    out = torch.empty_like(q)
    buf = torch.zeros(24, dtype=torch.int64, device=dev)
    fn = nat.load().sp_debug_w64_stamp_buffer
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int

    def run():
        nat.extend_attention(out, q, kb, vb, r2t, req, seq_d, ext_d, start, D ** -0.5, 0.0, True, ln, prefix + ln, ws, plan=plan)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert fn(buf.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    v = buf.cpu().tolist()
    n = v[23]
    tot = sum(v[:len(names)])
    print(f"{n} wave-iterations, {tot / n:.0f} cycles per iteration and wave (64 MFMAs = 2048 matrix cycles)")
    for i, name in enumerate(names):
        print(f"  {name:28s} {v[i] / n:8.1f} cycles  {100 * v[i] / tot:5.1f} %")


if __name__ == "__main__":
    main()
