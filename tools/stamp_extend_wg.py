#!/usr/bin/env python3
"""Timeline of a whole extend workgroup (diagnostic twin built with SP_STAMP_DEFS=-DSP_EXTEND_WGSTAMPS
bash tools/build_stamps.sh): cycles from entry to metadata decoded, to first tile landed, through the tile
loop, to the output written - averaged over workgroups - next to the wall time per workgroup.
  python tools/stamp_extend_wg.py [--bs 2048 --len 64]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scratchpad_amd import _native  # noqa: E402

_native._LIB_PATH = os.path.join(ROOT, "scratchpad_amd", "lib", "libscratchpad_hip_stamps.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--len", default="uniform")
    a = ap.parse_args()
    lib = _native.load()
    lib.sp_debug_extend_stamp_buffer.argtypes = [ctypes.c_void_p]
    dev, dt, Hq, Hkv, D = "cuda", torch.bfloat16, 32, 8, 128
    g = torch.Generator().manual_seed(0)
    ext = torch.randint(128, 4097, (a.bs,), generator=g) if a.len == "uniform" else torch.full((a.bs,), int(a.len))
    total = int(ext.sum())
    P = total + 64
    kb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    vb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(a.bs, int(ext.max()) + 8, dtype=torch.int32)
    off = 0
    for b in range(a.bs):
        n = int(ext[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(dev)
    q = torch.randn(total, Hq, D, device=dev).to(dt)
    o = torch.empty_like(q)
    req = torch.arange(a.bs, device=dev)
    ext_d = ext.to(torch.int32).to(dev)
    start = torch.zeros(a.bs, dtype=torch.int32, device=dev)
    start[1:] = torch.cumsum(ext_d[:-1], 0)
    ws = torch.empty(_native.extend_workspace_bytes(total, a.bs, Hq, D, dt), dtype=torch.uint8, device=dev)
    seq_d = ext.to(dev)
    run = lambda: _native.extend_attention(o, q, kb, vb, r2t, req, seq_d, ext_d, start, D ** -0.5, 0.0, True,
                                           int(ext.max()), int(ext.max()), ws)
    run()
    torch.cuda.synchronize()
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.sp_debug_extend_stamp_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.sp_debug_extend_stamp_buffer(None)
    v = buf.tolist()
    n = max(v[7], 1)
    ms = e0.elapsed_time(e1)
    print(f"{v[7]} workgroups, kernel {ms:.3f} ms = {ms * 1e3 * 256 / n:.2f} us of one CU per workgroup")
    for name, x in zip(["entry -> metadata decoded", "-> first tile landed", "tile loop", "barrier + output"], v[:4]):
        print(f"   {name:28s} {x / n:9.0f} cycles")


if __name__ == "__main__":
    main()
