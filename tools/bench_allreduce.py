#!/usr/bin/env python3
"""Direct all-reduce micro-benchmark on the ranks of ONE GPU (a rehearsal of the code path, not of xGMI):
the [T, hidden] SUM all-reduce of a TP decode step alone, followed by sp_fused_add_rmsnorm (two launches), and
as the one fused kernel, for several workgroup counts of the fused kernel.
  python tools/bench_allreduce.py [--world 2] [--T 128] [--hidden 8192]"""
import argparse
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rank_main(rank, a, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from scratchpad_amd import _native, distributed as d
    from scratchpad_amd.custom_all_reduce import CustomAllReduce
    d.init_distributed_environment(a.world, rank, f"tcp://127.0.0.1:{port}", 0, backend="gloo")
    d.initialize_model_parallel(a.world, backend="gloo", local_rank=0)
    torch.cuda.set_device(0)
    tp = d.get_tp_group()
    ca = CustomAllReduce(tp)
    tp.ca_comm = ca
    dt = torch.bfloat16
    x = torch.randn(a.T, a.hidden, device="cuda").to(dt)
    res = torch.randn(a.T, a.hidden, device="cuda").to(dt)
    w = torch.ones(a.hidden, device="cuda", dtype=dt)

    def timed(fn, n=100):
        xs = [x.clone() for _ in range(n)]
        for xi in xs[:10]:
            fn(xi)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for xi in xs:
            fn(xi)
        e1.record()
        torch.cuda.synchronize()
        dist.barrier()
        return e0.elapsed_time(e1) / n * 1e3

    def two_step(xi):
        y = ca.custom_all_reduce(xi)
        _native.fused_add_rmsnorm(y, res, w, 1e-5)

    out = {"all_reduce": timed(lambda xi: ca.custom_all_reduce(xi)), "all_reduce+norm (2 launches)": timed(two_step)}
    for blocks in a.blocks:
        _native.debug_set("ar_fused_blocks", blocks)
        out[f"fused, {blocks or 'default'} workgroups"] = timed(lambda xi: ca.fused_all_reduce_add_rmsnorm(xi, res, w, 1e-5))
    _native.debug_set("ar_fused_blocks", 0)
    ca.check()
    dist.barrier()
    ca.close()
    if rank == 0:
        q.put(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=8192)
    ap.add_argument("--blocks", type=int, nargs="*", default=[0, 16, 32, 64, 128])
    a = ap.parse_args()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=rank_main, args=(r, a, port, q)) for r in range(a.world)]
    for p in procs:
        p.start()
    out = q.get(timeout=280)
    for p in procs:
        p.join(timeout=60)
    print(f"direct all-reduce, {a.world} ranks on ONE GPU (rehearsal), [{a.T}, {a.hidden}] bf16 = "
          f"{a.T * a.hidden * 2 / 2**20:.2f} MiB, us per call:")
    for k, v in out.items():
        print(f"  {k:34s} {v:8.1f}")


if __name__ == "__main__":
    main()
