#!/usr/bin/env python3
"""Timeline of a workgroup of the 4 x 64-row extend kernel (-DSP_W64_WGSTAMPS build): entry -> item decoded -> first
tiles landed and published -> tiles done -> output stored, in cycles, for short and long prompts."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ab_extend import load_native  # noqa: E402


def main():
    nat = load_native(sys.argv[1] if len(sys.argv) > 1 else "libscratchpad_hip_wgstamps.so@extend_w64=2", 0)
    fn = nat.load().sp_debug_w64_stamp_buffer
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    Hq, Hkv, D, dt, dev = 32, 8, 128, torch.bfloat16, "cuda"
    for bs, ln in ((2048, 64), (256, 512), (64, 2048), (16, 4096)):
        g = torch.Generator().manual_seed(0)
        P = bs * ln + 64
        kb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
        vb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
        r2t = (torch.randperm(P, generator=g) + 1).to(torch.int32)[: bs * ln].view(bs, ln).contiguous().to(dev)
        q = torch.randn(bs * ln, Hq, D, device=dev).to(dt)
        req = torch.arange(bs, device=dev)
        ext = torch.full((bs,), ln, dtype=torch.int32, device=dev)
        start = (torch.arange(bs, dtype=torch.int32) * ln).to(dev)
        seq = torch.full((bs,), ln, device=dev)
        ws = torch.empty(nat.extend_workspace_bytes(bs * ln, bs, Hq, D, dt), dtype=torch.uint8, device=dev)
        plan = nat.extend_plan(ext, seq, bs * ln, Hq, Hkv, True)
        out = torch.empty_like(q)
        buf = torch.zeros(24, dtype=torch.int64, device=dev)

        def run():
            nat.extend_attention(out, q, kb, vb, r2t, req, seq, ext, start, D ** -0.5, 0.0, True, ln, ln, ws, plan=plan)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        assert fn(buf.data_ptr()) == 0
        run()
        torch.cuda.synchronize()
        v = buf.cpu().tolist()
        if "persist" in sys.argv:      # the persistent form's per-item stamps
            n = v[17]
            print(f"bs={bs} len={ln} PERSISTENT: {n} items, {v[18] / n:.1f} tiles each; cycles per item: start -> tiles + Q landed "
                  f"{v[12] / n:.0f}, way in {v[13] / n:.0f}, hot iterations {v[14] / n:.0f}, last iterations {v[15] / n:.0f} (their start {v[22] / n:.0f}, the way out {v[21] / n:.0f}), "
                  f"seam + output {v[16] / n:.0f}; per workgroup: mean {sum(v[12:17]) / max(v[20], 1):.0f} cycles, slowest {v[19]}", flush=True)
            assert fn(0) == 0
            continue
        n = v[16]
        print(f"bs={bs} len={ln}: {n} workgroups, {v[17] / n:.1f} tiles each; cycles: decode item {v[12] / n:.0f}, "
              f"Q + indices + first tiles {v[13] / n:.0f}, tiles {v[14] / n:.0f} ({v[14] / max(v[17], 1):.0f} per tile), "
              f"drain + output {v[15] / n:.0f}", flush=True)
        assert fn(0) == 0


if __name__ == "__main__":
    main()
