"""world_size-2 gloo tests of the tensor-parallel path (the N>1 code that RCCL runs on GPUs):
GroupCoordinator collectives and the sharded callers of the hot path reproduce the unsharded
result (linear.py:1148-1149, vocab_parallel_embedding.py:471, logits_processor.py:368-369)."""
import os
import socket
import traceback

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn_name, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        from scratchpad_amd import distributed as d
        d.init_distributed_environment(world, rank, f"tcp://127.0.0.1:{port}", rank, backend="gloo")
        d.initialize_model_parallel(world, backend="gloo", local_rank=rank)
        globals()[fn_name](rank, world)
        torch.distributed.barrier()
        q.put((rank, None))
    except Exception:
        q.put((rank, traceback.format_exc()))


def _spawn(fn_name, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in results:
        assert err is None, f"rank {rank}:\n{err}"


def _collectives(rank, world):
    from scratchpad_amd import distributed as d
    tp = d.get_tp_group()
    assert tp.world_size == world and tp.rank_in_group == rank
    x = torch.full((3, 4), float(rank + 1))
    y = d.tensor_model_parallel_all_reduce(x)
    assert y.data_ptr() == x.data_ptr(), "in place (parallel_state.py:352)"
    assert torch.equal(y, torch.full((3, 4), float(sum(range(1, world + 1)))))
    g = d.tensor_model_parallel_all_gather(torch.full((2, 3), float(rank)), dim=-1)
    assert g.shape == (2, 3 * world)
    for r in range(world):
        assert torch.equal(g[:, 3 * r:3 * r + 3], torch.full((2, 3), float(r)))
    g0 = tp.all_gather(torch.full((2, 3), float(rank)), dim=0)
    assert g0.shape == (2 * world, 3) and torch.equal(g0[2:], torch.full((2, 3), 1.0))
    assert tp.broadcast_object({"a": rank} if rank == 0 else None) == {"a": 0}

    class FakeCA:                      # the custom all-reduce slot (parallel_state.py:326-347)
        calls = 0

        def should_custom_ar(self, t):
            return t.numel() <= 8

        def custom_all_reduce(self, t):
            FakeCA.calls += 1
            out = t.clone()
            torch.distributed.all_reduce(out, group=tp.device_group)
            return out
    tp.ca_comm = FakeCA()
    small = torch.ones(4)
    out = tp.all_reduce(small)
    assert FakeCA.calls == 1 and out.data_ptr() != small.data_ptr() and torch.equal(out, torch.full((4,), float(world)))
    tp.all_reduce(torch.ones(64))
    assert FakeCA.calls == 1, "large messages fall through to the process-group all-reduce"
    tp.ca_comm = None


def _sharded_layers(rank, world):
    from scratchpad_amd import llama
    torch.manual_seed(0)
    hidden, D, Hq, Hkv, inter, vocab = 64, 16, 4, 2, 96, 100
    x = torch.randn(5, hidden)
    # QKV: per-rank [q | k | v] slices of the tp=1 merged weight
    full_qkv = torch.randn((Hq + 2 * Hkv) * D, hidden)
    qkv = llama.QKVParallelLinear(hidden, D, Hq, Hkv)
    qkv.weight.data = qkv.shard_from_full(full_qkv)
    out, _ = qkv(x)
    ref = torch.nn.functional.linear(x, full_qkv)
    hq, hk = Hq // world, Hkv // world
    q_ref = ref[:, rank * hq * D:(rank + 1) * hq * D]
    k_ref = ref[:, Hq * D + rank * hk * D: Hq * D + (rank + 1) * hk * D]
    v_ref = ref[:, (Hq + Hkv) * D + rank * hk * D:(Hq + Hkv) * D + (rank + 1) * hk * D]
    assert torch.allclose(out, torch.cat((q_ref, k_ref, v_ref), -1), atol=1e-5)
    # MLP: column-parallel gate/up, SiLU-mul stays local, row-parallel down + all-reduce
    full_gu, full_down = torch.randn(2 * inter, hidden), torch.randn(hidden, inter)
    gu = llama.MergedColumnParallelLinear(hidden, [inter, inter])
    gu.weight.data = gu.shard_from_full(full_gu)
    down = llama.RowParallelLinear(inter, hidden)
    down.weight.data = down.shard_from_full(full_down)
    h, _ = gu(x)
    d = h.shape[-1] // 2
    act = torch.nn.functional.silu(h[:, :d]) * h[:, d:]
    y, _ = down(act)
    full_h = torch.nn.functional.linear(x, full_gu)
    full_y = torch.nn.functional.linear(torch.nn.functional.silu(full_h[:, :inter]) * full_h[:, inter:], full_down)
    assert torch.allclose(y, full_y, atol=1e-3), float((y - full_y).abs().max())
    # vocab-parallel embedding + logits all-gather
    full_emb = torch.randn(vocab, hidden)
    emb = llama.VocabParallelEmbedding(vocab, hidden)
    assert emb.num_embeddings_padded == 128 and emb.num_embeddings_per_partition == 64
    emb.weight.data = emb.shard_from_full(full_emb)
    ids = torch.tensor([0, 63, 64, 99, 5])
    assert torch.allclose(emb(ids), full_emb[ids], atol=1e-6)

    class Cfg:
        vocab_size = vocab
    from scratchpad_amd.forward_info import ForwardBatch, ForwardMode
    fb = ForwardBatch(forward_mode=ForwardMode.EXTEND, batch_size=2, input_ids=ids, req_pool_indices=None,
                      seq_lens=None, out_cache_loc=None, seq_lens_sum=5,
                      extend_seq_lens=torch.tensor([2, 3], dtype=torch.int32))
    out = llama.LogitsProcessor(Cfg())(ids, x, emb, fb)
    want = torch.matmul(x[[1, 4]], full_emb.T)
    import pytest
    with pytest.raises(RuntimeError, match="gather_full_logits"):
        out.next_token_logits              # under TP the gather is an explicit collective, never an attribute read
    full = out.gather_full_logits()        # every rank calls it
    assert full.shape == (2, vocab) and full.dtype == torch.float32 and out.next_token_logits is full
    assert torch.allclose(full, want, atol=1e-4)
    with pytest.raises(RuntimeError, match="persistent graph output buffer"):
        out.rows(1)                        # a buffer object that cached full logits must not be re-sliced
    # prompt logprobs under TP (logits_processor.py:206-340): the projection of every kept position is gathered
    # across the vocabulary shards inside the processor (the decision depends on host-side batch fields only)
    from oracle import logprobs as olp
    fb.extend_seq_lens_cpu, fb.return_logprob = [2, 3], True
    fb.extend_logprob_start_lens_cpu, fb.top_logprobs_nums, fb.token_ids_logprobs = [0, 1], [2, 0], [None, [3, 70]]
    fb.extend_input_logprob_token_ids_gpu = torch.tensor([63, 0, 99, 0])
    out = llama.LogitsProcessor(Cfg())(ids, x, emb, fb)
    ref = olp.input_logprobs(x, full_emb, vocab, [2, 3], [0, 1], fb.extend_input_logprob_token_ids_gpu, [2, 0],
                             [None, [3, 70]])
    assert torch.allclose(out.next_token_logits, want, atol=1e-4)
    assert torch.allclose(out.input_token_logprobs, ref["input_token_logprobs"], atol=1e-4)
    assert out.input_top_logprobs_idx == ref["input_top_logprobs_idx"]
    assert out.input_token_ids_logprobs_idx == [[], [[3, 70], [3, 70]]]
    assert torch.allclose(torch.tensor(out.input_token_ids_logprobs_val[1]),
                          torch.tensor(ref["input_token_ids_logprobs_val"][1]), atol=1e-4)


def _kv_head_replication(rank, world):
    """tp (4) > total kv heads (2): each KV head lives on tp/kv = 2 ranks (linear.py:716-722)"""
    from scratchpad_amd import llama
    from scratchpad_amd.model_runner import ModelConfig
    qkv = llama.QKVParallelLinear(64, 16, 8, 2)
    assert (qkv.num_heads, qkv.num_kv_heads, qkv.num_kv_head_replicas) == (2, 1, 2)
    full = torch.arange((8 + 4) * 16, dtype=torch.float32).view(-1, 1).expand(-1, 64).contiguous()
    shard = qkv.shard_from_full(full)
    assert shard.shape == (4 * 16, 64)
    assert shard[0, 0].item() == rank * 2 * 16            # q rows of this rank
    assert shard[2 * 16, 0].item() == 8 * 16 + (rank // 2) * 16     # its (replicated) k head
    assert shard[3 * 16, 0].item() == 10 * 16 + (rank // 2) * 16    # and v head
    assert ModelConfig(64, 96, 1, 8, 2, 100).get_num_kv_heads(world) == 1


def test_group_coordinator_collectives_gloo():
    _spawn("_collectives", 2)


def test_sharded_hot_path_callers_reproduce_unsharded_gloo():
    _spawn("_sharded_layers", 2)


def test_kv_head_replication_tp4_gloo():
    _spawn("_kv_head_replication", 4)
