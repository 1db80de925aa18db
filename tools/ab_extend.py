#!/usr/bin/env python3
"""A/B of extend-kernel builds in ONE process on one box (boxes differ by ~5 %): every library named on the command
line (files under scratchpad_amd/lib, built by tools/build_variant.sh) runs the same config-3 launch in interleaved
rounds; outputs are compared with the first library's.
  python tools/ab_extend.py libscratchpad_hip.so libscratchpad_hip_x.so ... [--bs 64] [--len uniform|N] [--prefix 0]"""
import argparse
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_native(libspec, tag):
    """libspec: FILE[@key=value[,key=value...]] - the switches go through sp_debug_set after loading.  Two specs of
    the same FILE share the library's globals (one dlopen handle): give the second one a copy of the file."""
    libfile, _, switches = libspec.partition("@")
    spec = importlib.util.spec_from_file_location(f"sp_native_{tag}", os.path.join(ROOT, "scratchpad_amd", "_native.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m._LIB_PATH = os.path.join(ROOT, "scratchpad_amd", "lib", libfile)
    m.load()
    for kv in filter(None, switches.split(",")):
        k, v = kv.split("=")
        m.debug_set(k, int(v))
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--prefix", type=int, default=0)
    ap.add_argument("--len", default="uniform")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    Hq, Hkv, D, dt, dev = 32, 8, 128, torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(0)
    ext = torch.randint(128, 4097, (a.bs,), generator=g) if a.len == "uniform" else torch.full((a.bs,), int(a.len))
    pre = torch.full((a.bs,), a.prefix)
    seq = ext + pre
    P = int(seq.sum()) + 64
    kb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    vb = torch.empty(P + 1, Hkv, D, dtype=dt, device=dev).normal_(0, 0.5)
    perm = (torch.randperm(P, generator=g) + 1).to(torch.int32)
    r2t = torch.zeros(a.bs, int(seq.max()) + 8, dtype=torch.int32)
    off = 0
    for b in range(a.bs):
        n = int(seq[b])
        r2t[b, :n] = perm[off:off + n]
        off += n
    r2t = r2t.to(dev)
    T = int(ext.sum())
    q = torch.randn(T, Hq, D, device=dev).to(dt)
    req = torch.arange(a.bs, device=dev)
    ext_d = ext.to(torch.int32).to(dev)
    start = torch.zeros(a.bs, dtype=torch.int32, device=dev)
    start[1:] = torch.cumsum(ext_d[:-1], 0)
    seq_d = seq.to(dev)
    flops = 4 * Hq * D * float(((ext.double() ** 2) / 2 + ext.double() * pre.double()).sum())
    nats = [load_native(lib, i) for i, lib in enumerate(a.libs)]
    ws = torch.empty(nats[0].extend_workspace_bytes(T, a.bs, Hq, D, dt), dtype=torch.uint8, device=dev)
    plans = [n.extend_plan(ext_d, seq_d, T, Hq, Hkv, True) for n in nats]
    outs = [torch.empty_like(q) for _ in nats]

    def run(i):
        nats[i].extend_attention(outs[i], q, kb, vb, r2t, req, seq_d, ext_d, start, D ** -0.5, 0.0, True,
                                 int(ext.max()), int(seq.max()), ws, plan=plans[i])
    for i in range(len(nats)):
        outs[i].fill_(float("nan"))
        run(i)
    torch.cuda.synchronize()
    for i in range(1, len(nats)):
        d = (outs[i].float() - outs[0].float()).abs().max().item()
        print(f"{a.libs[i]} vs {a.libs[0]}: max |diff| = {d:.3e}  finite={bool(torch.isfinite(outs[i].float()).all())}", flush=True)
    times = [[] for _ in nats]
    for _ in range(a.rounds):
        for i in range(len(nats)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / a.iters)
    for i, lib in enumerate(a.libs):
        ms = sorted(times[i])[len(times[i]) // 2]
        print(f"{lib:44s} bs={a.bs} len={a.len} prefix={a.prefix}: {ms:.3f} ms (best {min(times[i]):.3f})  "
              f"{flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
