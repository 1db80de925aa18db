#!/usr/bin/env python3
"""Headline benchmark: Llama-3-8B bf16 TP=1 continuous-batching decode, bs=256, on MI355X.

One "step" = one decode step of the whole hot path for one batch: the scheduler-side
prepare_for_decode (slot allocation + req_to_token write), ForwardBatch.init_new, 32 decoder
layers (RMSNorm -> QKV GEMM -> rotary -> KV store -> paged decode attention -> o_proj -> RMSNorm
-> gate/up GEMM -> SiLU-mul -> down GEMM), final norm, lm_head, greedy argmax.  Weights are
random-init (N(0, 0.02), norm weights 1), the KV pool is filled with random bf16 data, contexts
are uniform-int [128, 4096] (seed 0) or fixed (--ctx N), KV slots are a seeded random
permutation of the pool (a long-running server's fragmented free list).  All inputs are resident
in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs N independent TP=1 replicas, one process per GPU: started under torch.distributed.run
(RANK/WORLD_SIZE in the environment) the process is one rank; started plainly, `python bench.py
--gpus N` spawns its own N ranks through torch.distributed.run before touching a GPU and exits with
their code (the reference's server spawns one process per rank the same way, server/server.py:252-265).
There is no data-path collective: ranks only meet at the timing barriers.
Rank 0 prints ONE JSON line.  After the decode measurement the same engine runs config 3 (64 prompts,
lengths U[128,4096]) once warm and once timed: ttft_p50_ms / ttft_p99_ms / prefill_tokens_per_sec and a
`roofline_prefill` block (extend attention kernel vs the dense bf16 MFMA peak) are part of the line.

  python bench.py --mode tp --gpus 8 --tp 8 --model llama3-70b          (config 4)
runs the REAL sharded model (heads, KV pool, linears per rank; RCCL all-reduce after o_proj / down_proj
and the vocab-parallel embedding, vocab-parallel greedy) eager and, over RCCL, under HIP-graph replay;
ranks that have to share GPUs (rehearsal on a 1-GPU box) talk over gloo instead and say so in the line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--ctx", default="uniform", help='"uniform" = U[128,4096] seed 0, or a fixed length')
    ap.add_argument("--model", default=None,
                    choices=["llama3-8b", "llama32-1b", "llama3-70b", "llama3-70b-tp8-rank"],
                    help="default llama3-8b (llama3-70b in --mode tp); llama3-70b-tp8-rank = one rank's shard "
                         "of config 4 as a TP=1 model (no all-reduce): supplementary")
    ap.add_argument("--tp", type=int, default=None, help="tp mode: tensor-parallel degree (default: --gpus)")
    ap.add_argument("--layers", type=int, default=None, help="override layer count (debug only)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="run the forward on the overlap worker's thread/stream (tp_worker_client.py)")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e5m2"],
                    help="decode mode: the reference's --kv-cache-dtype (fp8_e5m2 is NOT the headline config)")
    ap.add_argument("--requests", type=int, default=512, help="serve mode: number of requests in the trace")
    ap.add_argument("--max-input", type=int, default=2048, help="serve mode: prompts are U[128, max-input]")
    ap.add_argument("--max-output", type=int, default=128, help="serve mode: outputs are U[16, max-output]")
    ap.add_argument("--sample", action="store_true",
                    help="serve mode: temperature 0.8 / top-p 0.9 / top-k 50 sampling instead of greedy")
    ap.add_argument("--rate", type=float, default=0.0, help="serve mode: Poisson arrival rate (req/s); 0 = all at t=0")
    ap.add_argument("--mode", default="decode", choices=["decode", "prefill", "serve", "tp"],
                    help="decode = the headline metric (+ the TTFT half); prefill = config 3 alone; "
                         "tp = config 4 (sharded model, RCCL)")
    ap.add_argument("--no-ttft", action="store_true", help="decode mode: skip the config-3 TTFT passes")
    ap.add_argument("--prefix", type=int, default=0, help="prefill mode: shared cached prefix length")
    ap.add_argument("--max-prefill-tokens", type=int, default=16384,
                    help="prefill mode: tokens per extend batch (server/args.py max_prefill_tokens)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="allow ranks to SHARE a GPU (a box with fewer GPUs than --gpus): the line is then labelled "
                         "REHEARSAL and is not a scaling measurement; without it such a run exits non-zero")
    ap.add_argument("--profile-steps", type=int, default=4,
                    help="extra eager steps with HIP events around every attention launch")
    args = ap.parse_args()
    if args.model is None:
        args.model = "llama3-70b" if args.mode == "tp" else "llama3-8b"
    if args.mode == "tp" and args.bs == 256:
        args.bs = 128                                   # config 4: bs 128
    return args


def self_launch_if_needed(args) -> None:
    """`python bench.py --gpus N` without a launcher: become the launcher.  Runs before anything
    touches a GPU (importing torch does not); the children are ordinary torch.distributed.run ranks."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    pick_device(args, 0, args.gpus)         # refuses here, before any rank starts, when GPUs would be shared
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def pick_device(args, local_rank, world) -> int:
    """One process per GPU (the reference spawns one process per rank, server/server.py:252-265).  A box with
    fewer GPUs than ranks is refused unless --rehearsal says the ranks may share devices."""
    ndev = torch.cuda.device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a GPU")
    if world > ndev and not args.rehearsal:
        raise SystemExit(f"--gpus {world} but this box has {ndev} GPU(s): ranks would share devices and the line "
                         f"would not be an N-GPU measurement; pass --rehearsal to run it anyway (labelled)")
    return local_rank % ndev if args.rehearsal else local_rank


def device_identity(rank, local_rank, device_index):
    props = torch.cuda.get_device_properties(device_index)
    uuid = str(getattr(props, "uuid", "")) or f"{socket.gethostname()}:{device_index}"
    return {"rank": rank, "host": socket.gethostname(), "local_rank": local_rank, "device": device_index,
            "name": props.name, "uuid": uuid}


def gather_identities(me, world, group=None):
    """Every rank's (host, device, UUID): the proof that an N-GPU line ran on N devices."""
    import torch.distributed as dist
    seen = [None] * world
    if world > 1:
        dist.all_gather_object(seen, me, group=group)
    else:
        seen = [me]
    return seen


def devices_seen(seen) -> int:
    return len({(r["host"], r["uuid"]) for r in seen})


def rccl_sanity(world, device_index, shared):
    """One RCCL SUM all-reduce over all ranks (a fresh nccl group beside the gloo one): the N replicas
    can talk over xGMI, and their count is what the launcher said.  Ranks that share a GPU cannot form an
    RCCL communicator (RCCL refuses two ranks on one device): reported, not attempted."""
    import torch.distributed as dist
    if world <= 1:
        return None, None
    if shared:
        return None, "not attempted: ranks share a GPU (rehearsal) and RCCL refuses two ranks on one device"
    try:
        g = dist.new_group(list(range(world)), backend="nccl")
        t = torch.ones(1, dtype=torch.float32, device=f"cuda:{device_index}")
        dist.all_reduce(t, group=g)
        torch.cuda.synchronize()
        return int(round(float(t.item()))), None
    except Exception as e:        # noqa: BLE001 - reported in the line; the measurement does not depend on it
        return None, f"{type(e).__name__}: {e}"


def build_engine(args, device_index, seed, tp_rank=0, tp_size=1):
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs
    total_steps = 2 * (args.warmup + args.steps) + args.profile_steps + 16
    gen = torch.Generator().manual_seed(seed)
    if args.ctx == "uniform":
        ctx = torch.randint(128, 4097, (args.bs,), generator=gen)
    else:
        ctx = torch.full((args.bs,), int(args.ctx), dtype=torch.int64)
    context_len = int(ctx.max()) + total_steps + 4
    cfg = {"llama3-8b": ModelConfig.llama3_8b, "llama32-1b": ModelConfig.llama32_1b,
           "llama3-70b": ModelConfig.llama3_70b,
           "llama3-70b-tp8-rank": ModelConfig.llama3_70b_tp8_rank}[args.model](context_len)
    if args.layers:
        cfg.num_hidden_layers = args.layers
    pool_tokens = int(ctx.sum()) + args.bs * total_steps + 64
    sargs = ServerArgs(max_total_tokens=pool_tokens, max_running_requests=args.bs,
                       disable_cuda_graph=args.no_graph, cuda_graph_max_bs=args.bs,
                       cuda_graph_bs=[args.bs], kv_cache_dtype=args.kv_cache_dtype)
    mr = ModelRunner(cfg, sargs, tp_rank=tp_rank, tp_size=tp_size, dtype=torch.bfloat16, gpu_id=device_index,
                     seed=seed)
    # synthetic cache contents (random, not zeros: zero operands run at a higher clock)
    for arena in (mr.token_to_kv_pool._k_arena, mr.token_to_kv_pool._v_arena):
        for layer in range(arena.shape[0]):
            if arena.dtype == torch.uint8:      # e5m2 bytes of the same distribution
                rows = arena[layer]
                for lo in range(0, rows.shape[0], 1 << 17):
                    blk = rows[lo:lo + (1 << 17)]
                    blk.copy_(torch.empty(blk.shape, dtype=torch.bfloat16, device=blk.device).normal_(0.0, 0.5)
                              .to(torch.float8_e5m2).view(torch.uint8))
            else:
                arena[layer].normal_(0.0, 0.5)
    return mr, ctx, gen


def populate_batch(mr, ctx, gen):
    """Lay the batch out as a scheduler would have left it after prefill: request rows, slots in
    random (fragmented) order, req_to_token rows written, last sampled token as the next input."""
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    dev = mr.device
    bs = len(ctx)
    alloc = mr.token_to_kv_pool_allocator
    # fragment the free list with a seeded permutation (pool.py:225-232: free() appends in
    # arbitrary order over a server's lifetime)
    perm = torch.randperm(alloc.size, generator=gen) + 1
    alloc.free_slots = perm.to(torch.int64).to(dev)
    reqs = [Req(rid=str(i), origin_input_ids=[]) for i in range(bs)]
    batch = ScheduleBatch(reqs, mr.req_to_token_pool, alloc, device=dev)
    rows = batch.alloc_req_slots(bs)
    batch.req_pool_indices = torch.tensor(rows, dtype=torch.int64, device=dev)
    batch.seq_lens = ctx.to(torch.int64).to(dev)
    batch.seq_lens_sum = int(ctx.sum())
    table = mr.req_to_token_pool.req_to_token
    slots = alloc.alloc(int(ctx.sum()))
    off = 0
    for i in range(bs):
        n = int(ctx[i])
        table[rows[i], :n] = slots[off:off + n].to(torch.int32)
        off += n
    batch.forward_mode = ForwardMode.DECODE
    batch.output_ids = torch.randint(0, mr.model_config.vocab_size, (bs,), generator=gen).to(dev)
    return batch


def engine_step(worker, batch):
    batch.prepare_for_decode()
    out, next_ids = worker.forward_batch_generation(batch.get_model_worker_batch())
    batch.output_ids = next_ids
    return out


def attention_algorithmic_bytes(cfg, seq_sum, bs, kv_elem=2):
    """K+V rows of every context token + q,o rows + int32 slot indices, per layer."""
    kv = seq_sum * 2 * cfg.num_key_value_heads * cfg.head_dim * kv_elem
    qo = 2 * bs * cfg.num_attention_heads * cfg.head_dim * 2
    return kv + qo + 4 * seq_sum


def usable_cores() -> int:
    """host cores this process may actually use (affinity mask and cgroup CPU quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def physical_cores() -> int:
    """physical cores of the host (distinct (package, core) pairs in /proc/cpuinfo); 0 if unknown"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen)
    except OSError:
        return 0


def cores_note(cores: int) -> str:
    return (f"torch.set_num_threads({cores}) = the cores this process may use (affinity mask / cgroup quota); "
            f"the host has {physical_cores() or 'an unknown number of'} physical cores")


def cpu_baseline_cfg1(prompt=16, steps=16):
    """Config 1 in full (the reference's only e2e check has this shape,
    tests/e2e/test_engine.py:8-57): Llama-3.2-1B shapes, bs = 1, a 16-token prompt and 16 greedy decode
    steps on the oracle (oracle/llama.py, plain fp32 torch on the host cores), random-init weights."""
    from oracle import llama as ollama
    from oracle import ops
    cores = usable_cores()
    torch.set_num_threads(cores)
    shape = ollama.LlamaShape(2048, 8192, 16, 32, 8, 128256, True, 500000.0, (32.0, 1.0, 4.0, 8192), 8192, 1e-5)
    g = torch.Generator().manual_seed(0)
    w = {"model.embed_tokens.weight": torch.randn(shape.vocab, shape.hidden, generator=g) * 0.02,
         "model.norm.weight": torch.ones(shape.hidden)}
    qkv_rows = (shape.Hq + 2 * shape.Hkv) * shape.D
    for i in range(shape.layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = torch.ones(shape.hidden)
        w[p + "post_attention_layernorm.weight"] = torch.ones(shape.hidden)
        w[p + "self_attn.qkv_proj.weight"] = torch.randn(qkv_rows, shape.hidden, generator=g) * 0.02
        w[p + "self_attn.o_proj.weight"] = torch.randn(shape.hidden, shape.hidden, generator=g) * 0.02
        w[p + "mlp.gate_up_proj.weight"] = torch.randn(2 * shape.inter, shape.hidden, generator=g) * 0.02
        w[p + "mlp.down_proj.weight"] = torch.randn(shape.hidden, shape.inter, generator=g) * 0.02
    ctx = prompt + steps + 4
    kv = ollama.OracleKV(shape, ctx, 1, ctx)
    kv.req_to_token[0, :ctx] = torch.arange(1, ctx + 1, dtype=torch.int32)
    cos_sin = ops.rope_cos_sin_cache(ctx, shape.rope_theta, shape.D, shape.rope_scaling)
    ids = torch.randint(0, shape.vocab, (prompt,), generator=g)
    req = torch.zeros(1, dtype=torch.int64)
    ext = torch.tensor([prompt], dtype=torch.int32)
    pos, start = ops.compute_position(torch.zeros(1, dtype=torch.int32), ext)
    t0 = time.perf_counter()
    logits = ollama.forward(shape, w, kv, mode="extend", input_ids=ids, positions=pos, req_pool_indices=req,
                            seq_lens=torch.tensor([prompt]), out_cache_loc=torch.arange(1, prompt + 1),
                            extend_seq_lens=ext, extend_start_loc=start, cos_sin_cache=cos_sin)
    t_prefill = time.perf_counter() - t0
    out, tok = [], logits.argmax(-1)
    t0 = time.perf_counter()
    for s in range(steps):
        out.append(int(tok))
        seq = torch.tensor([prompt + s + 1])
        logits = ollama.forward(shape, w, kv, mode="decode", input_ids=tok, positions=ops.clamp_position(seq),
                                req_pool_indices=req, seq_lens=seq, out_cache_loc=torch.tensor([prompt + s + 1]),
                                cos_sin_cache=cos_sin)
        tok = logits.argmax(-1)
    t_decode = time.perf_counter() - t0
    return {"value": round(steps / t_decode, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "ttft_ms": round(t_prefill * 1e3, 1),
            "sample": f"config 1 in full: oracle/llama.py, fp32 torch, Llama-3.2-1B shapes (random-init), bs=1, "
                      f"{prompt}-token prompt ({t_prefill * 1e3:.0f} ms) + {steps} greedy decode steps "
                      f"({t_decode / steps * 1e3:.0f} ms/step); " + cores_note(cores)}


def cpu_baseline(bs=8, ctx=512, layers=2, seconds=12.0):
    """The oracle's decode step (oracle/llama.py, plain torch fp32) on the host cores: Llama-3-8B
    layer shapes, `layers` layers, bs x ctx; extrapolated to 32 layers + lm_head."""
    from oracle import llama as ollama
    from oracle import ops
    cores = usable_cores()
    torch.set_num_threads(cores)
    shape = ollama.LlamaShape(4096, 14336, layers, 32, 8, 128256, False, 500000.0, None, 8192, 1e-5)
    g = torch.Generator().manual_seed(0)
    w = {"model.embed_tokens.weight": torch.randn(4096, 4096, generator=g) * 0.02,  # stand-in rows
         "model.norm.weight": torch.ones(4096)}
    for i in range(layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = torch.ones(4096)
        w[p + "post_attention_layernorm.weight"] = torch.ones(4096)
        w[p + "self_attn.qkv_proj.weight"] = torch.randn(6144, 4096, generator=g) * 0.02
        w[p + "self_attn.o_proj.weight"] = torch.randn(4096, 4096, generator=g) * 0.02
        w[p + "mlp.gate_up_proj.weight"] = torch.randn(28672, 4096, generator=g) * 0.02
        w[p + "mlp.down_proj.weight"] = torch.randn(4096, 14336, generator=g) * 0.02
    head = torch.randn(128256, 4096, generator=g) * 0.02
    kv = ollama.OracleKV(shape, bs * (ctx + 64), bs, ctx + 64)
    for i in range(layers):
        kv.k[i].normal_(0, 0.5, generator=g)
        kv.v[i].normal_(0, 0.5, generator=g)
    slots = torch.randperm(bs * (ctx + 64), generator=g) + 1
    for b in range(bs):
        kv.req_to_token[b, :ctx + 64] = slots[b * (ctx + 64):(b + 1) * (ctx + 64)].to(torch.int32)
    cos_sin = ops.rope_cos_sin_cache(ctx + 64, 500000.0, 128, None)
    req = torch.arange(bs)
    ids = torch.randint(0, 4096, (bs,), generator=g)
    shape_nohead = shape
    w["lm_head.weight"] = head[:8]           # the layer timing excludes the real head (timed below)
    t_layers, n, seq = 0.0, 0, torch.full((bs,), ctx, dtype=torch.int64)
    t_end = time.perf_counter() + seconds
    while n < 2 or (time.perf_counter() < t_end and n < 32):
        seq = seq + 1
        loc = kv.req_to_token[req, seq - 1].long()
        t0 = time.perf_counter()
        ollama.forward(shape_nohead, w, kv, mode="decode", input_ids=ids, positions=ops.clamp_position(seq),
                       req_pool_indices=req, seq_lens=seq, out_cache_loc=loc, cos_sin_cache=cos_sin)
        dt = time.perf_counter() - t0
        if n > 0:
            t_layers += dt
        n += 1
    t_layer = t_layers / (n - 1) / layers
    h = torch.randn(bs, 4096, generator=g)
    t0 = time.perf_counter()
    torch.matmul(h, head.T)
    t_head = time.perf_counter() - t0
    step = 32 * t_layer + t_head
    return {"value": round(bs / step, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/llama.py decode step, fp32 torch, Llama-3-8B layer shapes, bs={bs} ctx={ctx}, "
                      f"{layers} layers x {n - 1} timed steps ({t_layer * 1e3:.1f} ms/layer) + lm_head "
                      f"({t_head * 1e3:.1f} ms), extrapolated to 32 layers; " + cores_note(cores)}


def prefill_passes(args, mr, worker, rank, bs, warm, timed, profile_attention=False):
    """Config 3 on an existing engine: bs prompts with lengths U[128,4096] (seed = rank), optionally on top
    of a shared cached prefix, admitted in arrival order into extend batches of <= max_prefill_tokens new
    tokens (the reference's PrefillAdder budget, server/args.py:33-34); TTFT of a request = time from the
    start of the pass to the end of the batch that contains it (all requests arrive at t = 0).
    Returns (median pass seconds, sorted TTFTs of that pass, lens, extend batches, attention profile)."""
    from scratchpad_amd import _native
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    gen = torch.Generator().manual_seed(rank)
    lens = torch.randint(128, 4097, (bs,), generator=gen).tolist()
    dev, vocab, cfg = mr.device, mr.model_config.vocab_size, mr.model_config
    prompts = [torch.randint(0, vocab, (n,), generator=gen).tolist() for n in lens]
    prefix_ids = torch.randint(0, vocab, (args.prefix,), generator=gen).tolist()
    batches, cur, cur_tok = [], [], 0
    for i, n in enumerate(lens):
        if cur and cur_tok + n > args.max_prefill_tokens:
            batches.append(cur)
            cur, cur_tok = [], 0
        cur.append(i)
        cur_tok += n
    batches.append(cur)

    def one_pass():
        mr.req_to_token_pool.clear()
        mr.token_to_kv_pool_allocator.clear()
        prefix_slots = None
        if args.prefix:     # a radix-cache hit: the prefix KV is already in the pool (computed once)
            pre = ScheduleBatch([Req("prefix", "", prefix_ids, None)], mr.req_to_token_pool,
                                mr.token_to_kv_pool_allocator, device=dev)
            pre.prepare_for_extend()
            worker.forward_batch_generation(pre.get_model_worker_batch())
            prefix_slots = pre.out_cache_loc.clone()
            mr.req_to_token_pool.free(pre.reqs[0].req_pool_idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ttft = [0.0] * bs
        for ids in batches:
            reqs = [Req(str(i), "", prefix_ids + prompts[i], None, prefix_indices=prefix_slots) for i in ids]
            sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=dev)
            sb.prepare_for_extend()
            _, nxt = worker.forward_batch_generation(sb.get_model_worker_batch())
            nxt.cpu()                       # first token delivered to the host
            t = time.perf_counter() - t0
            for i in ids:
                ttft[i] = t
        return time.perf_counter() - t0, ttft

    for _ in range(warm):
        one_pass()
    times, ttfts = [], []
    for _ in range(timed):
        el, tt = one_pass()
        times.append(el)
        ttfts.append(tt)
    el = sorted(times)[len(times) // 2]
    tt = sorted(ttfts[times.index(el)])

    prof = None
    if profile_attention:
        # one more pass with HIP events around every extend-attention launch (on the stream it is
        # launched on); flops of a launch = 4 Hq D sum(L^2 / 2 + L prefix) over its requests
        import scratchpad_amd.attention as att
        orig, events, forms = _native.extend_attention, [], []

        def timed_call(*a_, **k_):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(*a_, **k_)
            e1.record()
            events.append((e0, e1))
            forms.append(_native.debug_get("extend_last_kernel"))      # which kernel the library chose for this launch

        att._native.extend_attention = timed_call
        try:
            one_pass()
            torch.cuda.synchronize()
        finally:
            att._native.extend_attention = orig
        per_layer = cfg.num_hidden_layers
        calls = events[-per_layer * len(batches):]         # the pass's own launches (not the prefix step)
        ms = sum(a_.elapsed_time(b_) for a_, b_ in calls)
        heads = cfg.num_attention_heads // mr.tp_size
        flops = per_layer * sum(4.0 * heads * cfg.head_dim * (n * n / 2.0 + n * args.prefix) for n in lens)
        ran = forms[-per_layer * len(batches):]
        prof = {"launches": len(calls), "avg_launch_ms": ms / len(calls), "tflops": flops / (ms * 1e-3) / 1e12,
                "flops_per_launch": flops / len(calls),
                "kernels": {_native.EXTEND_KERNELS[f]: ran.count(f) for f in sorted(set(ran))},
                "w64_descriptor_patched": _native.debug_get("w64_descriptor_patched")}
    return el, tt, lens, len(batches), prof


def prefill_main(args, rank, local_rank, world):
    """Config 3 alone (`--mode prefill`)."""
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    bs = 64 if args.bs == 256 else args.bs
    cfg = ModelConfig.llama3_8b(4096 + args.prefix + 8)
    if args.layers:
        cfg.num_hidden_layers = args.layers
    sargs = ServerArgs(max_total_tokens=bs * 4096 + args.prefix + 64, max_running_requests=bs, disable_cuda_graph=True)
    mr = ModelRunner(cfg, sargs, dtype=torch.bfloat16, gpu_id=local_rank, seed=rank)
    worker = TpModelWorker(mr)
    el, tt, lens, nb, prof = prefill_passes(args, mr, worker, rank, bs, max(args.warmup, 1), args.steps,
                                            profile_attention=True)
    if rank != 0:
        return
    out = {"metric": "ttft_p50_ms", "value": round(tt[len(tt) // 2] * 1e3, 2), "unit": "ms", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el * 1e3, 2),
           "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic (random-init weights, random token ids)",
           "config": {"workload": f"llama3-8b TP=1 bf16 ragged prefill bs={bs}, prompt lengths U[128,4096] seed 0, "
                                  f"cached prefix {args.prefix}, extend batches <= {args.max_prefill_tokens} tokens",
                      "prompt_tokens": sum(lens), "extend_batches": nb, "layers": cfg.num_hidden_layers},
           "ttft_p99_ms": round(tt[int(len(tt) * 0.99)] * 1e3, 2),
           "prefill_tokens_per_sec": round(sum(lens) / el, 1),
           "roofline_prefill": prefill_roofline(prof)}
    print(json.dumps(out), flush=True)


def prefill_roofline(prof):
    if prof is None:
        return None
    # "kernel": what the library actually launched in the profiled pass (sp_debug_get("extend_last_kernel") after every
    # call), most frequent first - the 4-wave x 64-row kernels only launch from a library whose descriptor patch is in place
    kernels = sorted(prof["kernels"].items(), key=lambda kv: -kv[1])
    return {"bound": "mfma", "kernel": " | ".join(f"{k} x{n}" for k, n in kernels),
            "w64_descriptor_patched": prof["w64_descriptor_patched"],
            "achieved": round(prof["tflops"], 1),
            "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(prof["tflops"] / MFMA_PEAK_TFLOPS, 4),
            "traffic": None, "avg_launch_ms": round(prof["avg_launch_ms"], 4), "launches": prof["launches"],
            "algorithmic_flops_per_launch": int(prof["flops_per_launch"]),
            "note": "useful (causal) flops 4 Hq D sum(L^2/2) of every extend-attention launch / its HIP-event time; "
                    "attention is ~5 % of the prefill flops, the projections (hipBLASLt) are the rest of TTFT"}


def serve_main(args, rank, local_rank, world):
    """A synthetic continuous-batching trace through the scheduler-side producers: requests arrive
    (all at t = 0, or Poisson at --rate), are admitted FCFS into extend batches (<= --max-prefill-tokens
    new tokens, prefix looked up in the RadixCache), merged into the running batch, decoded by HIP-graph
    replay until their output length is reached, and handed back to the cache.  The loop is the
    reference's get_next_batch_to_run order (prefill first, else decode; scheduler.py) without its
    policy knobs; metrics follow tools/benchmark/common.py:306-420 (TTFT, TPOT = (latency - ttft) /
    (output_len - 1), ITL, output throughput)."""
    import random
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    from scratchpad_amd.radix_cache import RadixCache
    from scratchpad_amd.sampler import SamplingBatchInfo, SamplingParams
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    rnd = random.Random(rank)
    n_req = args.requests
    max_running = min(args.bs, n_req)
    in_lo, in_hi = 128, args.max_input
    out_lo, out_hi = 16, args.max_output
    vocab = 128256
    shared = [rnd.randrange(vocab) for _ in range(args.prefix)]
    prompts = [shared + [rnd.randrange(vocab) for _ in range(rnd.randint(in_lo, in_hi))] for _ in range(n_req)]
    out_lens = [rnd.randint(out_lo, out_hi) for _ in range(n_req)]
    arrivals = [0.0] * n_req
    if args.rate > 0:
        t = 0.0
        for i in range(n_req):
            t += rnd.expovariate(args.rate)
            arrivals[i] = t
    ctx_len = args.prefix + in_hi + out_hi + 8
    cfg = ModelConfig.llama3_8b(ctx_len)
    if args.layers:
        cfg.num_hidden_layers = args.layers
    pool = max_running * (args.prefix + in_hi + out_hi) + 4096
    buckets = sorted(set([1, 2, 4] + list(range(8, max_running + 1, 8)) + [max_running]))
    sargs = ServerArgs(max_total_tokens=pool, max_running_requests=max_running, disable_cuda_graph=args.no_graph,
                       cuda_graph_max_bs=max_running, cuda_graph_bs=buckets, kv_cache_dtype=args.kv_cache_dtype)
    mr = ModelRunner(cfg, sargs, dtype=torch.bfloat16, gpu_id=local_rank, seed=rank)
    mr.init_cuda_graphs()
    worker = TpModelWorker(mr)
    alloc, r2t, dev = mr.token_to_kv_pool_allocator, mr.req_to_token_pool, mr.device
    tree = RadixCache(r2t, alloc)
    sampling = SamplingParams(temperature=0.8, top_p=0.9, top_k=50) if args.sample else None
    mr.sampler.generator = torch.Generator(device=dev).manual_seed(rank)

    def run_trace():
        tree.reset()
        r2t.clear()
        alloc.clear()
        reqs = [Req(str(i), "", list(prompts[i]), sampling) for i in range(n_req)]
        waiting = list(range(n_req))
        running = None
        first_t, last_t, itl = [None] * n_req, [None] * n_req, []
        reserved = 0            # output tokens promised to running requests but not yet allocated
        # device_s: wall time between handing a batch to the worker and its token ids reaching the host - the only spans in
        # which the GPU has work; the rest of the trace's duration is the scheduler side alone (admission, prefix cache,
        # prepare_for_*, result bookkeeping) with the GPU idle: what an overlapped event loop could hide
        steps = {"extend": 0, "decode": 0, "hit_tokens": 0, "device_s": 0.0}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        now = lambda: time.perf_counter() - t0
        while waiting or running is not None:
            n_run = 0 if running is None else len(running.reqs)
            admit, budget = [], args.max_prefill_tokens
            while waiting and n_run + len(admit) < max_running and arrivals[waiting[0]] <= now():
                r = reqs[waiting[0]]
                r.init_next_round_input(tree)
                need = r.extend_input_len + out_lens[waiting[0]] - 1
                if admit and r.extend_input_len > budget:
                    break
                if alloc.available_size() + tree.evictable_size() - reserved < need:
                    break
                tree.inc_lock_ref(r.last_node)
                steps["hit_tokens"] += r.prefix_len
                admit.append(waiting.pop(0))
                budget -= r.extend_input_len
                reserved += out_lens[admit[-1]] - 1      # one slot per decode step of this request
            if admit:
                nb = ScheduleBatch([reqs[i] for i in admit], r2t, alloc, device=dev, tree_cache=tree)
                nb.prepare_for_extend()
                if args.sample:
                    nb.sampling_info = SamplingBatchInfo.from_schedule_batch(nb, cfg.vocab_size)
                mwb = nb.get_model_worker_batch()
                t_dev = now()
                _, ids = worker.forward_batch_generation(mwb)
                toks = ids.tolist()                     # first tokens reach the host
                t = now()
                steps["device_s"] += t - t_dev
                for i, tok in zip(admit, toks):
                    reqs[i].output_ids.append(tok)
                    first_t[i] = last_t[i] = t
                nb.output_ids = ids
                steps["extend"] += 1
                done = [i for i in admit if len(reqs[i].output_ids) >= out_lens[i]]
                if done:
                    for i in done:
                        reqs[i].finished_reason = "length"
                        reserved -= out_lens[i] - 1
                        tree.cache_finished_req(reqs[i])
                    nb.filter_batch()
                # the prompts' KV becomes shareable at once (the reference's prefill result handler calls
                # cache_unfinished_req for every request that keeps running)
                for r in nb.reqs:
                    tree.cache_unfinished_req(r)
                if nb.reqs:
                    if running is None:
                        running = nb
                    else:
                        running.merge_batch(nb)
                continue
            if running is None:                         # idle until the next arrival
                time.sleep(max(0.0, arrivals[waiting[0]] - now()))
                continue
            if not running.check_decode_mem():
                raise RuntimeError("serve trace: KV pool exhausted despite the admission reserve")
            running.prepare_for_decode()
            mwb = running.get_model_worker_batch()
            t_dev = now()
            _, ids = worker.forward_batch_generation(mwb)
            toks = ids.tolist()
            t = now()
            steps["device_s"] += t - t_dev
            running.output_ids = ids
            steps["decode"] += 1
            any_done = False
            for r, tok in zip(running.reqs, toks):
                i = int(r.rid)
                r.output_ids.append(tok)
                itl.append(t - last_t[i])
                last_t[i] = t
                reserved -= 1
                if len(r.output_ids) >= out_lens[i]:
                    r.finished_reason = "length"
                    tree.cache_finished_req(r)
                    any_done = True
            if any_done:
                running.filter_batch()
                if not running.reqs:
                    running = None
        dur = now()
        # conservation: every KV slot is free or owned by the prefix cache, every request row is back
        assert alloc.available_size() + tree.total_size() == alloc.size, "KV slots leaked"
        assert r2t.available_size() == r2t.size and tree.protected_size() == 0 and reserved == 0
        assert all(len(reqs[i].output_ids) == out_lens[i] for i in range(n_req))
        ttft = sorted(first_t[i] - arrivals[i] for i in range(n_req))
        tpot = sorted((last_t[i] - first_t[i]) / (out_lens[i] - 1) for i in range(n_req) if out_lens[i] > 1)
        return dur, ttft, tpot, sorted(itl), steps

    run_trace()                                         # warm-up pass (allocator, caches, clocks)
    dur, ttft, tpot, itl, steps = run_trace()
    if rank != 0:
        return
    pct = lambda xs, q: xs[min(len(xs) - 1, int(len(xs) * q))] * 1e3
    total_in, total_out = sum(len(p) for p in prompts), sum(out_lens)
    out = {"metric": "serve_output_tokens_per_sec", "value": round(total_out / dur, 1), "unit": "tokens/s",
           "n_gpus": world, "steps": steps["extend"] + steps["decode"], "warmup": 1,
           "ms_per_step": round(dur * 1e3 / (steps["extend"] + steps["decode"]), 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic (random-init weights, random token ids, "
                   + ("temperature 0.8 / top-p 0.9 / top-k 50 sampling)" if args.sample else "greedy sampling)"),
           "config": {"workload": f"llama3-8b TP=1 bf16 continuous batching: {n_req} requests, prompts U[{in_lo},{in_hi}]"
                                  f" (+{args.prefix} shared prefix), outputs U[{out_lo},{out_hi}], max running {max_running}, "
                                  f"arrival {'all at t=0' if args.rate <= 0 else f'Poisson {args.rate}/s'}, "
                                  f"extend batches <= {args.max_prefill_tokens} tokens, RadixCache on"
                                  f"{', KV cache fp8_e5m2' if args.kv_cache_dtype != 'auto' else ''}",
                      "input_tokens": total_in, "output_tokens": total_out, "extend_steps": steps["extend"],
                      "decode_steps": steps["decode"], "prefix_cache_hit_tokens": steps["hit_tokens"]},
           "duration_s": round(dur, 3), "total_tokens_per_sec": round((total_in + total_out) / dur, 1),
           # the scheduler side alone, GPU idle (this loop is synchronous: no overlap worker in it)
           "scheduler_only_s": round(dur - steps["device_s"], 3),
           "scheduler_only_frac": round((dur - steps["device_s"]) / dur, 4),
           "ttft_ms": {"p50": round(pct(ttft, 0.5), 1), "p99": round(pct(ttft, 0.99), 1),
                       "mean": round(sum(ttft) / len(ttft) * 1e3, 1)},
           "tpot_ms": {"p50": round(pct(tpot, 0.5), 2), "p99": round(pct(tpot, 0.99), 2),
                       "mean": round(sum(tpot) / len(tpot) * 1e3, 2)},
           "itl_ms": {"p50": round(pct(itl, 0.5), 2), "p99": round(pct(itl, 0.99), 2)}}
    print(json.dumps(out), flush=True)

def pmc_workload_key(args) -> str:
    """the workload a PMC record under profiles/ was taken for (tools/pmc_decode.sh writes the same string)"""
    return f"{args.model}|bs{args.bs}|ctx{args.ctx}|kv{args.kv_cache_dtype}"


def pmc_traffic(alg_bytes, args):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; tools/pmc_decode.sh): the measured
    traffic / algorithmic ratio of the same kernel SOURCES and the same workload (attention shapes, batch,
    contexts), else None - a ratio recorded for other kernel sources or another shape says nothing about this run."""
    import glob
    sha = decode_kernel_sources_sha1()
    key = pmc_workload_key(args)
    # newest round first; within a round the in-model record (tools/pmc_bench.sh: rocprofv3 --pmc around bench.py itself)
    # before the kernel-alone ones (tools/pmc_decode.sh)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_bench_pmc*.json")), reverse=True) + \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_decode_attn_pmc*.json")), reverse=True)
    stale = False
    for path in paths:                   # newest round first: the pass recorded for THESE kernel sources
        rec = json.load(open(path))
        if rec.get("bench_workload", "llama3-8b|bs256|ctxuniform|kvauto") != key:
            continue
        if rec.get("kernel_source_sha1") == sha:
            name = os.path.relpath(path, ROOT)
            return int(alg_bytes * rec["traffic_over_algorithmic"]), \
                f"{name} (PMC ratio x algorithmic, same kernel sources)"
        stale = True
    if stale:
        return None, "the PMC passes under profiles/ were recorded for other kernel sources (stale): not used"
    return None, "no PMC pass for this workload"


def decode_kernel_sources_sha1() -> str:
    h = hashlib.sha1()
    for name in ("decode_mfma.hip", "decode_attention.hip", "attention_internal.h", "sp_common.h"):
        h.update(open(os.path.join(ROOT, "scratchpad_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def tp_main(args, rank, local_rank, world):
    """Config 4: the sharded model over a TP group - every rank holds its heads, its KV pool and its
    slices of the linears; SUM all-reduce of [bs, hidden] after o_proj and down_proj and after the
    vocab-parallel embedding (RCCL, or the direct IPC kernel with SP_CUSTOM_ALLREDUCE=1), vocab-parallel
    greedy.  Measured eager, then under HIP-graph replay with the collectives captured inside the graph
    (distributed/parallel_state.py:256-302, model_executor/cuda_graph_runner.py:295-329)."""
    import datetime
    import torch.distributed as dist
    from scratchpad_amd import distributed as d
    from scratchpad_amd.model_runner import TpModelWorker
    tp = args.tp or world
    if world % tp:
        raise SystemExit(f"--gpus {world} is not a multiple of --tp {tp}")
    ndev = torch.cuda.device_count()
    shared = ndev < world
    backend = "gloo" if shared else "nccl"      # RCCL refuses two ranks on one device: rehearsal over gloo
    if world > 1:
        d.init_distributed_environment(world, rank, "env://", local_rank, backend=backend)
    d.initialize_model_parallel(tp, backend=backend if world > 1 else None, local_rank=local_rank)
    group = rank // tp                            # dp replica this rank belongs to
    mr, ctx, gen = build_engine(args, local_rank, seed=group, tp_rank=rank % tp, tp_size=tp)
    worker = TpModelWorker(mr)
    batch = populate_batch(mr, ctx, gen)
    cfg = mr.model_config
    tpg = d.get_tp_group()
    ca = getattr(tpg, "ca_comm", None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            engine_step(worker, batch)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if backend == "gloo" else mr.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        if ca is not None:
            ca.check()
        return el

    for _ in range(args.warmup):
        engine_step(worker, batch)
    eager_s = timed(args.steps)
    graph_s, graph_note = None, None
    if args.no_graph:
        graph_note = "--no-graph"
    elif tp > 1 and backend == "gloo" and ca is None:
        graph_note = "not captured: gloo collectives cannot be recorded into a HIP graph (rehearsal; RCCL or " \
                     "SP_CUSTOM_ALLREDUCE=1 can)"
    else:
        try:
            mr.init_cuda_graphs()
            for _ in range(max(args.warmup, 2)):
                engine_step(worker, batch)
            graph_s = timed(args.steps)
        except Exception as e:        # noqa: BLE001 - reported in the line; the eager figure stands
            graph_note = f"capture failed: {type(e).__name__}: {e}"
            mr.graph_runner = None
    # the collective alone, at the step's message size
    ar_us, ar_bytes = None, args.bs * cfg.hidden_size * 2
    if tp > 1:
        x = torch.randn(args.bs, cfg.hidden_size, device=mr.device).to(torch.bfloat16)
        for _ in range(10):
            d.tensor_model_parallel_all_reduce(x.clone())
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xs = [x.clone() for _ in range(50)]
        e0.record()
        for xi in xs:
            d.tensor_model_parallel_all_reduce(xi)
        e1.record()
        barrier()
        ar_us = e0.elapsed_time(e1) / len(xs) * 1e3
    # the collective TOGETHER with what follows it in every layer half (residual add + RMSNorm): as the two
    # launches of the unfused path, and as the one fused kernel where the direct all-reduce offers it
    ar_norm_two_step_us = ar_norm_fused_us = None
    if tp > 1:
        from scratchpad_amd import _native as nat
        wn = torch.ones(cfg.hidden_size, dtype=torch.bfloat16, device=mr.device)
        res = torch.randn(args.bs, cfg.hidden_size, device=mr.device).to(torch.bfloat16)

        def time_pairs(fn, n=50):
            xs_ = [x.clone() for _ in range(n)]
            for xi in xs_[:5]:
                fn(xi)
            barrier()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for xi in xs_:
                fn(xi)
            a1.record()
            barrier()
            return a0.elapsed_time(a1) / n * 1e3

        def two_step(xi):
            y = d.tensor_model_parallel_all_reduce(xi)
            nat.fused_add_rmsnorm(y, res, wn, 1e-5)

        ar_norm_two_step_us = time_pairs(two_step)
        if ca is not None and ca.should_fuse_norm(x, res, wn):
            ar_norm_fused_us = time_pairs(lambda xi: tpg.fused_all_reduce_add_rmsnorm(xi, res, wn, 1e-5))
    seen = gather_identities(device_identity(rank, int(os.environ.get("LOCAL_RANK", "0")), local_rank), world)
    if ca is not None:
        ca.close()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    # the faster of the two is the figure (ranks that share one GPU are time-sliced by the driver, and a
    # replayed graph that spins on a peer then costs a scheduling quantum per all-reduce: rehearsal only)
    best = eager_s if graph_s is None else min(graph_s, eager_s)
    dp = world // tp
    value = dp * args.bs * args.steps / best
    n_params = sum(p_.numel() for p_ in mr.model.parameters())
    step_bytes = n_params * 2 + batch.seq_lens_sum * 2 * cfg.get_num_kv_heads(tp) * cfg.head_dim * 2 * cfg.num_hidden_layers
    roof = args.bs / (step_bytes / (HBM_PEAK_GBPS * 1e9))
    out = {"metric": "decode_tokens_per_sec", "value": round(value, 1), "unit": "tokens/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(best / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong" if dp == 1 else "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic (random-init weights, random KV, seeded contexts and slot permutation)",
           "config": {"workload": ("REHEARSAL (ranks share a GPU: not an xGMI measurement) " if shared else "")
                                  + f"{args.model} TP={tp} bf16 decode bs={args.bs} seq_len=1, "
                                  f"ctx={'U[128,4096] seed 0' if args.ctx == 'uniform' else args.ctx}, page_size=1",
                      "tp": tp, "dp_replicas": dp, "layers": cfg.num_hidden_layers,
                      "backend": "RCCL (nccl)" if backend == "nccl" else
                                 "gloo - REHEARSAL: ranks share GPUs, not an xGMI measurement",
                      "all_reduce": "direct IPC kernel (SP_CUSTOM_ALLREDUCE=1)" if ca is not None else backend,
                      "graph": graph_s is not None, "graph_note": graph_note,
                      "value_from": "graph replay" if graph_s is not None and graph_s <= eager_s else "eager"},
           "eager_tokens_per_sec": round(dp * args.bs * args.steps / eager_s, 1),
           "eager_ms_per_step": round(eager_s / args.steps * 1e3, 3),
           "graph_ms_per_step": None if graph_s is None else round(graph_s / args.steps * 1e3, 3),
           "allreduce_us_per_call": None if ar_us is None else round(ar_us, 1), "allreduce_bytes": ar_bytes,
           "allreduces_per_step": 2 * cfg.num_hidden_layers + 1,
           "allreduce_plus_norm_us": {"two_launches": None if ar_norm_two_step_us is None else round(ar_norm_two_step_us, 1),
                                      "fused_kernel": None if ar_norm_fused_us is None else round(ar_norm_fused_us, 1),
                                      "per_step_calls": 2 * cfg.num_hidden_layers,
                                      "fused_in_model": bool(ca is not None and ca.fused_calls > 0)},
           "step_frac_of_hbm_roofline_per_rank": round(args.bs / (best / args.steps) / roof, 4),
           "ranks_seen": len(seen), "devices_seen": devices_seen(seen), "rehearsal": bool(shared),
           "ranks": [{k: r[k] for k in ("rank", "host", "device", "name", "uuid")} for r in seen]}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    self_launch_if_needed(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    import torch.distributed as dist
    launcher_local_rank = local_rank
    local_rank = pick_device(args, local_rank, world)
    torch.cuda.set_device(local_rank)
    from scratchpad_amd import _native
    from scratchpad_amd.model_runner import TpModelWorker
    _native.load()
    if args.mode == "tp":
        if args.steps == 64:
            args.steps, args.warmup = 32, 4
        tp_main(args, rank, local_rank, world)
        return
    if world > 1:
        # replicas share nothing on the data path: the only cross-rank traffic is the timing barrier
        # and the max over ranks of the elapsed time, so a host-side (gloo) group is all that is needed
        dist.init_process_group("gloo")
    # who runs where, before anything is built: an N-GPU line must come from N distinct devices
    seen = gather_identities(device_identity(rank, launcher_local_rank, local_rank), world)
    shared = devices_seen(seen) < world
    if shared and not args.rehearsal:      # (pick_device already refuses on one host; this covers odd launchers)
        raise SystemExit(f"{world} ranks on {devices_seen(seen)} device(s): refusing without --rehearsal")
    if args.mode == "prefill":
        if args.steps == 64:
            args.steps, args.warmup = 3, 1
        prefill_main(args, rank, local_rank, world)
        return
    if args.mode == "serve":
        serve_main(args, rank, local_rank, world)
        return
    mr, ctx, gen = build_engine(args, local_rank, seed=rank)
    if not args.no_graph:
        mr.init_cuda_graphs()
    worker = TpModelWorker(mr)
    overlap_worker = None
    if args.overlap:
        from scratchpad_amd.tp_worker_client import TpModelWorkerClient
        overlap_worker = TpModelWorkerClient(mr)
    batch = populate_batch(mr, ctx, gen)

    def run_steps(n):
        """n decode steps; with --overlap the scheduler side runs one step ahead of the results"""
        if overlap_worker is None:
            for _ in range(n):
                engine_step(worker, batch)
            return
        pending = 0
        for _ in range(n):
            batch.prepare_for_decode()
            _, placeholders = overlap_worker.forward_batch_generation(batch.get_model_worker_batch())
            batch.output_ids = placeholders
            if pending:
                overlap_worker.resolve_last_batch_result()
            pending = 1
        if pending:
            overlap_worker.resolve_last_batch_result()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    barrier()
    seq_sum_start = batch.seq_lens_sum
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    seq_sum_end = batch.seq_lens_sum
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant kernel: paged decode attention, timed per launch with HIP events on the
    # stream it is launched on (eager steps continuing the same trace)
    cfg = mr.model_config
    roofline = None
    if overlap_worker is not None:
        overlap_worker.close()
    if rank == 0 and args.profile_steps > 0:
        saved = mr.graph_runner
        mr.graph_runner = None
        events, sums, calls = [], [], []
        orig = _native.decode_attention

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(*a, **k)
            e1.record()
            events.append((e0, e1))
            calls.append((a, k))

        engine_step(worker, batch)  # eager warm-up (workspace growth, autotune of nothing)
        import scratchpad_amd.attention as att
        att._native.decode_attention = timed
        try:
            for _ in range(args.profile_steps):
                engine_step(worker, batch)
                sums.extend([batch.seq_lens_sum] * cfg.num_hidden_layers)
            torch.cuda.synchronize()
        finally:
            att._native.decode_attention = orig
            mr.graph_runner = saved
        ms = [a.elapsed_time(b) for a, b in events]
        per_call_ms = sum(ms) / len(ms)
        # The same launches once more, back to back: the last step's per-layer attention calls (own q,
        # own layer of the pool, the step's plan) between ONE event pair per pass.  An event pair per
        # call adds its own two barrier packets to every measurement (about 4 % here); this form times
        # what rocprofv3 times (kernel + merge + dispatch), and is the figure the roofline uses.
        layer_calls = calls[-cfg.num_hidden_layers:]
        passes = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for a_, k_ in layer_calls:
                orig(*a_, **k_)
            e1.record()
            passes.append((e0, e1))
        torch.cuda.synchronize()
        avg_ms = sum(a.elapsed_time(b) for a, b in passes[1:]) / (len(passes) - 1) / len(layer_calls)
        kv_elem = 1 if args.kv_cache_dtype == "fp8_e5m2" else 2
        alg = attention_algorithmic_bytes(cfg, sums[-1], args.bs, kv_elem)     # the replayed step's lengths
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(alg, args)
        roofline = {"bound": "hbm",
                    # (the launches carry the plan's range geometry where the backend has one: include/scratchpad_hip.h)
                    "kernel": ("decode_mfma_range_kernel" if layer_calls[-1][1].get("ranges") else "decode_mfma_kernel")
                    + "+decode_merge_kernel",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "avg_launch_ms": round(avg_ms, 4), "launches": 3 * len(layer_calls),
                    "avg_launch_ms_with_per_call_events": round(per_call_ms, 4),
                    "algorithmic_bytes_per_launch": int(alg)}

    # ---- the other half of the metric: p50 TTFT of config 3 on the same engine (64 prompts, lengths
    # U[128,4096], extend batches <= max_prefill_tokens): one warm pass, one timed pass, one pass with
    # HIP events around the extend-attention launches (every rank runs them; rank 0 reports its own)
    ttft = None
    if not args.no_ttft and args.model == "llama3-8b" and cfg.context_len >= 4100:
        el_p, tt_p, lens_p, nb_p, prof_p = prefill_passes(args, mr, worker, rank, 64, 1, 1, profile_attention=rank == 0)
        ttft = {"ttft_p50_ms": round(tt_p[len(tt_p) // 2] * 1e3, 2), "ttft_p99_ms": round(tt_p[int(len(tt_p) * 0.99)] * 1e3, 2),
                "prefill_tokens_per_sec": round(sum(lens_p) / el_p, 1),
                "prefill_config": {"workload": f"llama3-8b TP=1 bf16 ragged prefill bs=64, prompt lengths U[128,4096] "
                                               f"seed {rank}, extend batches <= {args.max_prefill_tokens} tokens",
                                   "prompt_tokens": sum(lens_p), "extend_batches": nb_p, "pass_ms": round(el_p * 1e3, 1)},
                "roofline_prefill": prefill_roofline(prof_p)}
        barrier()
    rccl_ranks, rccl_note = rccl_sanity(world, local_rank, shared)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.bs * args.steps / elapsed
    # whole-step HBM roofline: weights once + KV of every context token + new KV
    n_params = sum(p.numel() for p in mr.model.parameters())
    avg_seq = (seq_sum_start + seq_sum_end) / 2 + args.bs / 2
    kv_elem = 1 if args.kv_cache_dtype == "fp8_e5m2" else 2
    step_bytes = n_params * 2 + avg_seq * 2 * cfg.num_key_value_heads * cfg.head_dim * kv_elem * cfg.num_hidden_layers
    step_roofline_tok_s = args.bs / (step_bytes / (HBM_PEAK_GBPS * 1e9))
    out = {
        "metric": "decode_tokens_per_sec", "value": round(value, 1), "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "kv_cache_dtype": "bf16" if args.kv_cache_dtype == "auto" else args.kv_cache_dtype,
        "data": "synthetic (random-init weights, random KV, seeded contexts and slot permutation)",
        "config": {"workload": f"{args.model} TP=1 bf16 continuous-batching decode bs={args.bs} seq_len=1, "
                               f"ctx={'U[128,4096] seed 0' if args.ctx == 'uniform' else args.ctx}, page_size=1, "
                               f"{'HIP-graph replay' if not args.no_graph else 'eager'}{', overlap worker' if args.overlap else ''}"
                               f"{', KV cache fp8_e5m2' if args.kv_cache_dtype != 'auto' else ''}",
                   "batch_size": args.bs, "mean_context": round(avg_seq / args.bs, 1),
                   "layers": cfg.num_hidden_layers, "replicas": world},
        "step_hbm_roofline_tokens_per_sec": round(step_roofline_tok_s, 1),
        "step_frac_of_hbm_roofline": round(value / world / step_roofline_tok_s, 4),
    }
    if world > 1:
        # the proof of N devices (a replica line must not be able to lie about n_gpus)
        if shared:
            out["config"]["workload"] = "REHEARSAL (ranks share a GPU: not an N-GPU measurement) " + \
                out["config"]["workload"]
        out.update({"ranks_seen": len(seen), "devices_seen": devices_seen(seen), "rehearsal": bool(shared),
                    "rccl_ranks": rccl_ranks, "rccl_note": rccl_note,
                    "ranks": [{k: r[k] for k in ("rank", "host", "device", "name", "uuid")} for r in seen]})
    out["library_rows"] = _native.library_rows_report()     # which projections ran with a substituted row count
    if ttft is not None:
        out.update(ttft)
    if roofline is not None:
        out["roofline"] = roofline
    if not args.no_cpu_baseline and world == 1:     # the CPU legs are timed on rank 0 of the 1-GPU run only
        out["cpu_baseline"] = cpu_baseline()
        out["cpu_baseline_cfg1"] = cpu_baseline_cfg1()
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    # A/B switches of the library for measurement runs only (never read on the product's call path):
    # SP_BENCH_DEBUG_SET="extend_w64_persist=2,extend_dma=0" -> sp_debug_set(key, value) before anything runs
    if os.environ.get("SP_BENCH_DEBUG_SET"):
        from scratchpad_amd import _native as _nat
        for kv in os.environ["SP_BENCH_DEBUG_SET"].split(","):
            k, v = kv.split("=")
            _nat.debug_set(k.strip(), int(v))
    main()
