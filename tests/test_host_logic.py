"""CPU tests of the host-side mirror: pools/allocator against the reference's recorded trace,
ForwardMode semantics, rope cache construction, TP shard math, graph buckets."""
import numpy as np
import pytest
import torch

from scratchpad_amd import distributed as dist_
from scratchpad_amd.forward_info import CaptureHiddenMode, ForwardMode
from scratchpad_amd.layers import get_rope
from scratchpad_amd.model_runner import ModelConfig, ServerArgs, get_batch_sizes_to_capture
from scratchpad_amd.pool import KVCache, MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator
from tests import golden


def test_allocator_trace_matches_reference():
    g = golden.load("kv_pool")
    alloc = TokenToKVPoolAllocator(16, torch.float32, "cpu", None)
    assert alloc.free_slots.dtype == torch.int64 and alloc.page_size == 1
    a0 = alloc.alloc(5)
    a1 = alloc.alloc(4)
    assert alloc.available_size() == int(g["alloc_avail0"])
    alloc.free(a0[1:3])
    a2 = alloc.alloc(8)
    assert (alloc.alloc(100) is None) == bool(g["alloc_too_many_is_none"])
    state = alloc.backup_state()
    a3 = alloc.alloc(1)
    alloc.restore_state(state)
    a4 = alloc.alloc(1)
    alloc.free_group_begin()
    alloc.free(a1[:2])
    alloc.free(a2[:3])
    assert alloc.available_size() == int(g["alloc_avail_in_group"])
    alloc.free_group_end()
    for name, val in (("a0", a0), ("a1", a1), ("a2", a2), ("a3", a3), ("a4", a4)):
        assert np.array_equal(val.numpy(), g["alloc_" + name]), name
    assert np.array_equal(alloc.free_slots.numpy(), g["alloc_free_final"])
    alloc.free(torch.empty(0, dtype=torch.int64))          # no-op
    alloc.clear()
    assert np.array_equal(alloc.free_slots.numpy(), g["alloc_free_after_clear"])
    assert 0 not in alloc.free_slots.tolist(), "slot 0 is the reserved dummy slot"


def test_req_to_token_pool_trace_matches_reference():
    g = golden.load("kv_pool")
    r2t = ReqToTokenPool(5, 12, "cpu", False)
    assert r2t.req_to_token.dtype == torch.int32
    r0 = r2t.alloc(2)
    r1 = r2t.alloc(2)
    r2t.free(r0[0])
    r2 = r2t.alloc(2)
    assert (r2t.alloc(3) is None) == bool(g["r2t_none"])
    r2t.write((torch.tensor([1, 3]), torch.tensor([4, 7])), torch.tensor([11, 13], dtype=torch.int32))
    r2t.write((2, slice(0, 3)), torch.tensor([5, 6, 7], dtype=torch.int32))
    assert r0 == g["r2t_r0"].tolist() and r1 == g["r2t_r1"].tolist() and r2 == g["r2t_r2"].tolist()
    assert np.array_equal(r2t.req_to_token.numpy(), g["r2t_table"])
    assert r2t.available_size() == int(g["r2t_avail"])
    rec = ReqToTokenPool(3, 4, "cpu", True)
    rec.write((0, slice(0, 2)), torch.tensor([9, 8], dtype=torch.int32))
    records = rec.get_write_records()
    other = ReqToTokenPool(3, 4, "cpu", False)
    other.apply_write_records(records)
    assert torch.equal(other.req_to_token, rec.req_to_token) and rec.get_write_records() == []


@pytest.mark.parametrize("interleave", [True, False])
def test_kv_pool_layout_and_guards(interleave, monkeypatch):
    monkeypatch.setattr(MHATokenToKVPool, "interleave_kv", interleave)
    pool = MHATokenToKVPool(10, 1, torch.bfloat16, 2, 64, 3, "cpu")
    assert isinstance(pool, KVCache)
    k = pool.get_key_buffer(1)
    kb, vb = pool.get_kv_buffer(0)
    assert kb.data_ptr() != vb.data_ptr()
    assert k.shape == (11, 2, 64) and k.stride()[1:] == (64, 1), "[size+1, Hkv, D] token-major"
    if interleave:
        # ONE arena [layers, size+1, 2, Hkv, D]: a token's K row and V row of a layer are adjacent
        assert k.stride(0) == 256
        assert vb.data_ptr() - kb.data_ptr() == 128 * 2
        assert pool.get_key_buffer(2).data_ptr() - pool.get_key_buffer(1).data_ptr() == 11 * 256 * 2
        kb[3].fill_(1.0)
        vb[3].fill_(2.0)
        assert pool._kv_arena[0, 3].flatten().tolist() == [1.0] * 128 + [2.0] * 128
    else:
        # one arena per K/V: layer l is a view at l * (size+1) rows
        assert k.stride(0) == 128
        assert pool.get_key_buffer(2).data_ptr() - pool.get_key_buffer(1).data_ptr() == 11 * 128 * 2
    assert pool.get_kv_size_bytes() == (3 * 11 * 128 * 2,) * 2
    # the flat-data / transfer pair keeps the reference's semantics (memory/pool.py:348-372) on either layout
    idx = torch.tensor([2, 5, 9])
    for l in range(3):
        pool.k_buffer[l][idx] = torch.randn(3, 2, 64).to(torch.bfloat16)
        pool.v_buffer[l][idx] = torch.randn(3, 2, 64).to(torch.bfloat16)
    flat = pool.get_flat_data(idx)
    assert flat.shape == (2, 3, 3, 2, 64) and torch.equal(flat[0, 1], pool.get_key_buffer(1)[idx])
    other = MHATokenToKVPool(10, 1, torch.bfloat16, 2, 64, 3, "cpu")
    other.transfer(idx, flat)
    assert torch.equal(other.get_value_buffer(2)[idx], pool.get_value_buffer(2)[idx])
    other.transfer_per_layer(torch.tensor([1]), torch.stack([kb[3:4], vb[3:4]]), 1)
    assert torch.equal(other.get_key_buffer(1)[1], kb[3]) and torch.equal(other.get_value_buffer(1)[1], vb[3])
    # memory/pool.py:329-346: the transfer engines upstream address token i of buffer b at ptrs[b] + i * items[b]; walk
    # that contract with ctypes over every listed buffer and compare with the views the kernels use
    import ctypes
    ptrs, lens, items = pool.get_contiguous_buf_infos()
    rows = {}
    for b, (ptr, n, item) in enumerate(zip(ptrs, lens, items)):
        assert n == 11 * item
        for tok in (0, 3, 10):
            raw = (ctypes.c_uint16 * (item // 2)).from_address(ptr + tok * item)
            rows[b, tok] = torch.frombuffer(bytearray(raw), dtype=torch.bfloat16).clone()
    regions = sorted((p_, p_ + n) for p_, n in zip(ptrs, lens))
    assert all(a[1] <= b_[0] for a, b_ in zip(regions, regions[1:])), "listed buffers must not overlap"
    if interleave:
        assert len(ptrs) == 3 and items == [512] * 3                  # one K|V pair per token per layer
        for l in range(3):
            for tok in (0, 3, 10):
                assert torch.equal(rows[l, tok][:128], pool.get_key_buffer(l)[tok].flatten())
                assert torch.equal(rows[l, tok][128:], pool.get_value_buffer(l)[tok].flatten())
    else:
        assert len(ptrs) == 6 and items == [256] * 6                  # K buffers, then V buffers, as upstream
        for l in range(3):
            for tok in (0, 3, 10):
                assert torch.equal(rows[l, tok], pool.get_key_buffer(l)[tok].flatten())
                assert torch.equal(rows[3 + l, tok], pool.get_value_buffer(l)[tok].flatten())
    with pytest.raises(NotImplementedError):
        MHATokenToKVPool(10, 16, torch.bfloat16, 2, 64, 1, "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # the store itself is HIP-only
        from types import SimpleNamespace
        pool.set_kv_buffer(SimpleNamespace(layer_id=0), torch.tensor([1]), torch.zeros(1, 2, 64, dtype=torch.bfloat16),
                           torch.zeros(1, 2, 64, dtype=torch.bfloat16))


def test_forward_mode_semantics():
    # forward_info.py:38-66 (the reference compares against the raw ints 1..7)
    assert [int(m) for m in ForwardMode] == [1, 2, 3, 4, 5, 6, 7]
    assert ForwardMode.EXTEND.is_extend() and ForwardMode.MIXED.is_extend()
    assert not ForwardMode.DECODE.is_extend() and ForwardMode.DECODE.is_decode()
    assert ForwardMode.MIXED.is_mixed() and ForwardMode.IDLE.is_idle()
    assert ForwardMode.DECODE.is_cuda_graph() and ForwardMode.IDLE.is_cuda_graph()
    assert ForwardMode.TARGET_VERIFY.is_cuda_graph() and not ForwardMode.EXTEND.is_cuda_graph()
    assert ForwardMode.IDLE.is_decode_or_idle() and ForwardMode.DUMMY_FIRST.is_dummy_first()
    assert not CaptureHiddenMode.NULL.need_capture() and CaptureHiddenMode.LAST.is_last()


def test_rope_cache_is_the_references_bit_for_bit():
    g = golden.load("rotary")
    for i in range(int(g["num_cases"])):
        sc = golden.rope_scaling(g, i)          # None, llama3, or (round 4) linear / dynamic NTK / YaRN
        rope = get_rope(int(g[f"c{i}_head_size"]), int(g[f"c{i}_rotary_dim"]), int(g[f"c{i}_max_pos"]),
                        float(g[f"c{i}_base"]), bool(g[f"c{i}_neox"]), sc, dtype=torch.float32)
        assert np.array_equal(rope.cos_sin_cache.numpy(), g[f"c{i}_cos_sin_cache"]), f"case {i} {sc}"
    a = get_rope(64, 64, 128, 10000, True, None, torch.float32)
    assert a is get_rope(64, 64, 128, 10000, True, None, torch.float32), "cached per key"
    assert get_rope(64, 64, 128, 10000, True, None, torch.float32, partial_rotary_factor=0.5).rotary_dim == 32
    kinds = [golden.rope_scaling(g, i)["rope_type"] for i in range(int(g["num_cases"])) if golden.rope_scaling(g, i)]
    assert sorted(set(kinds)) == ["dynamic", "linear", "llama3", "yarn"]
    for unbuilt in ("deepseek_yarn", "longrope"):           # MLA / Phi-3 model families: out of scope
        with pytest.raises(ValueError):
            get_rope(64, 64, 128, 10000, True, {"rope_type": unbuilt, "factor": 2.0}, torch.float32)
    with pytest.raises(NotImplementedError):                 # per-LoRA tables (a list of linear factors)
        get_rope(64, 64, 128, 10000, True, {"rope_type": "linear", "factor": [2.0, 4.0]}, torch.float32)


def test_model_config_head_math_and_graph_buckets():
    c8, c70 = ModelConfig.llama3_8b(), ModelConfig.llama3_70b()
    assert (c8.head_dim, c8.get_num_kv_heads(1), c8.get_num_kv_heads(8), c8.get_num_kv_heads(16)) == (128, 8, 1, 1)
    assert c70.num_attention_heads // 8 == 8 and c70.get_num_kv_heads(8) == 1
    assert ModelConfig.llama32_1b().head_dim == 64
    assert get_batch_sizes_to_capture(ServerArgs(cuda_graph_max_bs=160), 4096) == \
        [1, 2, 4, 8] + list(range(16, 161, 8))          # the reference's list (cuda_graph_runner.py:100)
    assert get_batch_sizes_to_capture(ServerArgs(), 4096)[-1] == 256
    assert get_batch_sizes_to_capture(ServerArgs(cuda_graph_bs=[4, 300, 64]), 100) == [4, 64]


def test_runner_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scratchpad_amd.model_runner import ModelRunner
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ModelRunner(ModelConfig(64, 128, 1, 2, 1, 100))
    with pytest.raises(RuntimeError, match="unsupported device"):
        ModelRunner(ModelConfig(64, 128, 1, 2, 1, 100), device="cpu")


def test_qkv_shard_math_single_rank():
    """tp=1: the merged layout is [q | k | v] (linear.py:746-760)"""
    dist_.destroy_model_parallel()
    dist_.initialize_model_parallel(1)
    from scratchpad_amd.llama import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    qkv = QKVParallelLinear(64, 16, 4, 2)
    assert (qkv.num_heads, qkv.num_kv_heads, qkv.num_kv_head_replicas) == (4, 2, 1)
    assert qkv.weight.shape == ((4 + 2 * 2) * 16, 64)
    full = torch.arange(128 * 64, dtype=torch.float32).view(128, 64)
    assert torch.equal(qkv.shard_from_full(full), full)
    assert MergedColumnParallelLinear(64, [96, 96]).weight.shape == (192, 64)
    assert RowParallelLinear(96, 64).weight.shape == (64, 96)
    dist_.destroy_model_parallel()


# ---------------------------------------------------------------- host logic added with the §8f rows
def test_sampling_params_and_batch_info_host_side():
    import torch
    from scratchpad_amd.sampler import TOP_K_ALL, SamplingBatchInfo, SamplingParams
    greedy = SamplingParams(temperature=0.0)
    assert greedy.top_k == 1 and greedy.temperature == 1.0          # sampling_params.py:66-69
    assert SamplingParams(top_k=-1).top_k == TOP_K_ALL               # whole vocabulary
    for bad in (dict(top_p=0.0), dict(top_p=1.5), dict(min_p=-0.1), dict(temperature=-1.0), dict(top_k=0)):
        with pytest.raises(ValueError):
            SamplingParams(**bad).verify()
    a = SamplingBatchInfo.from_params([greedy, SamplingParams(temperature=0.7, top_k=40, min_p=0.05)], 100, "cpu")
    b = SamplingBatchInfo.from_params([greedy], 100, "cpu")
    assert not a.is_all_greedy and a.need_min_p_sampling and b.is_all_greedy and not b.need_min_p_sampling
    assert a.temperatures.shape == (2, 1) and a.top_ks.dtype == torch.int32 and len(a) == 2
    a.merge_batch(b)
    assert len(a) == 3 and a.top_ks.tolist() == [1, 40, 1] and not a.is_all_greedy
    a.filter_batch([0, 2], torch.tensor([0, 2]))
    assert len(a) == 2 and a.top_ks.tolist() == [1, 1] and a.min_ps.tolist() == [0.0, 0.0]


def test_skinny_gemm_dispatch_rule_and_cpu_passthrough():
    import torch
    from scratchpad_amd import _native
    pays = _native.skinny_gemm_pays
    assert pays(1, 6144, 4096) and pays(1, 28672, 4096) and pays(1, 128256, 4096)       # bs 1: all but down_proj
    assert pays(16, 4096, 4096) and pays(8, 6144, 4096) and pays(8, 28672, 4096)
    assert not pays(1, 4096, 14336) and not pays(8, 4096, 14336)                          # long rows: library
    assert not pays(16, 28672, 4096) and not pays(8, 128256, 4096)
    x, w = torch.randn(3, 64), torch.randn(5, 64)
    assert torch.equal(_native.linear(x, w), torch.nn.functional.linear(x, w))           # host tensors: plain F.linear


def test_vision_attention_plan_partition(monkeypatch):
    import torch
    from scratchpad_amd import vision
    from scratchpad_amd.mllama_vision import padding_positions
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: object())
    P, Pp = 1025, 1032
    pad = padding_positions(torch.tensor([[1, 1, 1, 1], [1, 0, 0, 0]]), P, Pp)
    assert pad.shape == (2, 4 * Pp) and int(pad[0].sum()) == 4 * 7 and int(pad[1].sum()) == 3 * Pp + 7
    plan = vision.VisionAttnPlan(2, 4 * Pp, "cpu", pad_rows=pad)
    assert plan.main.seq_lens == [4 * Pp] * 2 and plan.main_rows is None                  # natural order
    assert plan.side.seq_lens == [28, 3 * Pp + 7] and plan.side.key_lens == [4 * P, P]
    assert plan.side.key_index.shape == (2, 4 * P) and plan.side.key_index[1, P - 1] == 4 * Pp + P - 1
    assert torch.equal(plan.side_rows[:7], torch.arange(P, Pp))
    rag = vision.VisionAttnPlan(1, 10, "cpu", cu_seqlens=[0, 3, 3, 10])
    assert rag.main.seq_lens == [3, 0, 7] and rag.side is None and rag.main.start.tolist() == [0, 3, 3]
    monkeypatch.setattr(vision.VisionAttnPlan, "MOVE_SPILL_ROWS", True)
    moved = vision.VisionAttnPlan(1, 4 * Pp, "cpu", pad_rows=pad[:1])
    assert moved.main.seq_lens == [4096] and moved.side.seq_lens == [28, 4] and moved.side.key_lens == [4 * P, 4 * Pp]
    both = torch.cat([moved.main_rows, moved.side_rows]).sort().values
    assert torch.equal(both, torch.arange(4 * Pp)), "every position is computed exactly once"


def test_ring_allocators_match_a_list_model_across_wraps():
    """Both free lists are rings (pool.py): a long random alloc / free / free-group / backup-restore
    sequence, which wraps the rings many times, must hand out exactly what the reference's
    slice-the-head / append-at-the-tail lists would."""
    import random
    rnd = random.Random(7)
    size = 37
    alloc = TokenToKVPoolAllocator(size, torch.float32, "cpu", None)
    model = list(range(1, size + 1))
    held = []
    for step in range(3000):
        op = rnd.random()
        if op < 0.45:
            n = rnd.randint(0, 9)
            got = alloc.alloc(n)
            if n > len(model):
                assert got is None
            else:
                assert got.tolist() == model[:n], step
                model = model[n:]
                held.extend(got.tolist())
        elif op < 0.85 and held:
            rnd.shuffle(held)
            k = rnd.randint(1, min(len(held), 11))
            back, held = held[:k], held[k:]
            if rnd.random() < 0.3:
                alloc.free_group_begin()
                assert not alloc.is_not_in_free_group
                mid = rnd.randint(0, k)
                alloc.free(torch.tensor(back[:mid], dtype=torch.int64))
                alloc.free(torch.tensor(back[mid:], dtype=torch.int64))
                assert alloc.available_size() == len(model), "a group's frees land at free_group_end"
                alloc.free_group_end()
            else:
                alloc.free(torch.tensor(back, dtype=torch.int64))
            model = model + back
        elif op < 0.9:
            state = alloc.backup_state()
            n = min(len(model), rnd.randint(0, 5))
            assert alloc.alloc(n).tolist() == model[:n]
            alloc.restore_state(state)              # the allocation is undone
        assert alloc.available_size() == len(model)
    assert alloc.free_slots.tolist() == model and 0 not in model
    with pytest.raises(RuntimeError, match="overflow"):
        alloc.free(torch.arange(1, size + 2))

    rows = ReqToTokenPool(9, 4, "cpu")
    rmodel = list(range(9))
    rheld = []
    for step in range(2000):
        if rnd.random() < 0.5:
            n = rnd.randint(0, 4)
            got = rows.alloc(n)
            if n > len(rmodel):
                assert got is None
            else:
                assert got == rmodel[:n] and all(isinstance(x, int) for x in got)
                rmodel = rmodel[n:]
                rheld.extend(got)
        elif rheld:
            rnd.shuffle(rheld)
            if rnd.random() < 0.5:
                x = rheld.pop()
                rows.free(x)
                rmodel.append(x)
            else:
                k = rnd.randint(1, len(rheld))
                back, rheld = rheld[:k], rheld[k:]
                rows.free(back)
                rmodel.extend(back)
        assert rows.available_size() == len(rmodel) and rows.free_slots == rmodel


def test_mixed_batch_merges_sampling_info_like_the_reference():
    """mix_with_running goes through merge_batch (schedule_batch.py:1073-1101, 1361-1397): the running
    decodes' sampling parameters, pending output ids and request list are merged in request order - a
    running request that samples with temperature must not silently become greedy in a MIXED batch."""
    from scratchpad_amd.sampler import SamplingBatchInfo, SamplingParams
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch

    def batches(prefill_params, running_params):
        r2t = ReqToTokenPool(8, 16, "cpu")
        alloc = TokenToKVPoolAllocator(32, torch.float32, "cpu", None)
        new = ScheduleBatch([Req("n0", "", [1, 2, 3, 4], prefill_params)], r2t, alloc, device="cpu")
        new.forward_mode = ForwardMode.EXTEND
        new.input_ids = torch.tensor([1, 2, 3, 4])
        new.out_cache_loc = torch.tensor([10, 11, 12, 13])
        new.req_pool_indices = torch.tensor([0])
        new.seq_lens = torch.tensor([4])
        new.seq_lens_sum, new.extend_num_tokens = 4, 4
        new.prefix_lens, new.extend_lens = [0], [4]
        run = ScheduleBatch([Req("r0", "", [5, 6], running_params[0], output_ids=[7]),
                             Req("r1", "", [8], running_params[1], output_ids=[9, 9])], r2t, alloc, device="cpu")
        run.forward_mode = ForwardMode.DECODE
        run.input_ids = torch.tensor([7, 9])
        run.out_cache_loc = torch.tensor([20, 21])
        run.req_pool_indices = torch.tensor([1, 2])
        run.seq_lens = torch.tensor([3, 3])
        run.seq_lens_sum = 6
        for b in (new, run):
            if any(getattr(r, "sampling_params", None) is not None for r in b.reqs):
                b.sampling_info = SamplingBatchInfo.from_schedule_batch(b, 100)
        return new, run

    hot = SamplingParams(temperature=0.7, top_k=40, top_p=0.9)
    # prefill greedy (no info at all), running requests sample: the merged info has 3 rows in request order
    new, run = batches(None, (hot, None))
    assert new.sampling_info is None and run.sampling_info is not None
    new.mix_with_running(run)
    assert new.forward_mode == ForwardMode.MIXED and [r.rid for r in new.reqs] == ["n0", "r0", "r1"]
    assert new.input_ids.tolist() == [1, 2, 3, 4, 7, 9] and new.out_cache_loc.tolist() == [10, 11, 12, 13, 20, 21]
    assert new.extend_lens == [4, 1, 1] and new.prefix_lens == [0, 2, 2] and new.extend_num_tokens == 6
    assert new.seq_lens.tolist() == [4, 3, 3] and new.seq_lens_sum == 10
    info = new.sampling_info
    assert len(info) == 3 and not info.is_all_greedy
    assert info.top_ks.tolist() == [1, 40, 1] and info.temperatures.view(-1).tolist() == pytest.approx([1.0, 0.7, 1.0])
    assert new.get_model_worker_batch().sampling_info is info
    # the other way round, and the overlap scheduler's prefix_lens (output_ids lags one step)
    new, run = batches(hot, (None, None))
    new.mix_with_running(run, enable_overlap=True)
    assert new.sampling_info.top_ks.tolist() == [40, 1, 1] and new.prefix_lens == [0, 3, 3]
    # nobody samples: the batch stays "all greedy" without materialising anything
    new, run = batches(None, (None, None))
    new.mix_with_running(run)
    assert new.sampling_info is None and len(new.reqs) == 3
    # the mixed-in running requests are extend rows of length 1 (schedule_batch.py:1077-1079), their
    # logprob bookkeeping and flags are carried (1361-1397), and pending output ids must line up or fail loudly
    new, run = batches(None, (None, None))
    run.return_logprob, run.top_logprobs_nums, run.token_ids_logprobs = True, [2, 0], [None, [3, 4]]
    run.has_grammar = True
    new.extend_logprob_start_lens = [0]
    new.mix_with_running(run)
    assert [r.extend_input_len for r in new.reqs] == [4, 1, 1]
    assert new.extend_logprob_start_lens == [0, 0, 0]
    # one id per KEPT position - the prefill row's 4 (zeros: it asked for none) + one zero per running row
    # (its next token is not known yet) - so LogitsProcessor's logprobs[arange, ids] lines up (the reference leaves
    # the list short, schedule_batch.py:1073-1101, and its scheduler never builds this batch, scheduler.py:944-949)
    assert new.extend_input_logprob_token_ids.tolist() == [0] * 6
    assert new.return_logprob and new.top_logprobs_nums == [0, 2, 0] and new.token_ids_logprobs == [None, None, [3, 4]]
    assert new.has_grammar and not new.has_stream
    new.reqs[1].init_next_round_input()                 # the pin does not outlive the round
    assert new.reqs[1].extend_input_len == 3
    new, run = batches(None, (None, None))
    new.output_ids = torch.tensor([42])
    run.output_ids = torch.tensor([7, 9])
    new.mix_with_running(run)
    assert new.output_ids.tolist() == [42, 7, 9]
    new, run = batches(None, (None, None))
    new.output_ids = torch.tensor([42])                 # one side without output ids: rows would not line up
    with pytest.raises(RuntimeError, match="output_ids"):
        new.mix_with_running(run)


def test_spare_row_views_and_library_row_table():
    """_native.extend_rows re-views an activation over more rows of its own storage only when the storage
    really extends that far; library_rows never shrinks a product and is the identity off its table."""
    import torch
    from scratchpad_amd import _native
    x = _native.empty_rows(5, 8, torch.float32, "cpu")
    assert x.shape == (5, 8) and x.is_contiguous()
    v = _native.extend_rows(x, 5 + _native.ROW_SLACK)
    assert v is not None and v.shape == (5 + _native.ROW_SLACK, 8) and v.data_ptr() == x.data_ptr()
    assert _native.extend_rows(x, 6 + _native.ROW_SLACK) is None
    assert _native.extend_rows(torch.zeros(5, 8), 6) is None
    part = torch.zeros(10, 16)[2:6, :8]                   # row-strided view in the middle of a buffer
    assert _native.extend_rows(part, 8).shape == (8, 8) and _native.extend_rows(part, 9) is None
    assert _native.extend_rows(x, 3) is x
    for (N, K), table in _native._LIBRARY_ROWS_TABLE.items():
        for M, Mp in table.items():
            assert M < Mp <= M + _native.ROW_SLACK
            if _native._LIBROWS_MODE == "table" and _native.library_versions_match():
                assert _native.library_rows(M, N, K) == Mp
    assert _native.library_rows(255, 6144, 4096) == 255 and _native.library_rows(256, 1234, 4096) == 256


def test_library_row_table_is_gated_on_the_build_it_was_measured_on(monkeypatch):
    """The (shape, M) -> M' table is a measurement of one hipBLASLt build: on another torch / HIP version, with
    SP_LIBRARY_ROWS=0, and under SP_LIBRARY_ROWS=auto before the calibration has run, nothing is substituted."""
    import importlib
    import torch
    from scratchpad_amd import _native
    try:
        monkeypatch.setenv("SP_LIBRARY_ROWS", "table")
        monkeypatch.setattr(torch, "__version__", "2.11.0+rocm7.1")
        importlib.reload(_native)
        assert not _native.library_versions_match() and _native._LIBRARY_ROWS == {}
        assert _native.library_rows(256, 6144, 4096) == 256
        rep = _native.library_rows_report()
        assert rep["versions_match"] is False and rep["substituted"] == {}
        monkeypatch.undo()
        for mode in ("0", "auto"):
            monkeypatch.setenv("SP_LIBRARY_ROWS", mode)
            importlib.reload(_native)
            assert _native._LIBROWS_MODE == mode and _native.library_rows(256, 6144, 4096) == 256
        monkeypatch.undo()
    finally:
        monkeypatch.undo()
        importlib.reload(_native)
    if _native.library_versions_match():
        assert _native.library_rows(256, 6144, 4096) == 264


def test_decode_split_size_policy_and_where_items_are_planned():
    """HipAttnBackend host logic only.  (a) _plan_chunk: about one (request, split) item per CU between MIN_CHUNK and
    MAX_CHUNK, from the host's bound on sum(seq_lens) alone (the advisory longest-request hint left in round 6).
    (b) which plans carry items at all: only where a launch of the model can read them - a shape the range kernel refuses
    (decode_ranges 0), a layer with a logit soft-cap, or a step whose line does not fit the range section."""
    from scratchpad_amd.attention import HipAttnBackend
    b = HipAttnBackend.__new__(HipAttnBackend)
    b.num_kv_head, b.head_dim = 8, 128
    dt = torch.bfloat16
    assert (b.MIN_CHUNK, b.MAX_CHUNK, b.TARGET_ITEMS) == (64, 768, 256)
    assert b._plan_chunk(553000, dt) == 768                      # the headline batch
    assert b._plan_chunk(1064 * 64, dt) == 512 and b._plan_chunk(40 * 8, dt) == 64 and b._plan_chunk(10 ** 9, dt) == 768
    b.num_kv_head = 1                                            # 70B / TP 8 rank: one workgroup per item
    assert b._plan_chunk(288000, dt) == 768
    # (b) the graph launch covers no items at all where nothing reads them
    b.max_context_len, b.decode_ranges = 8192, 256
    b.plan_items = False
    assert b._graph_slots(256) == 0 and b._ranges_for(256, 8192) == 256
    b.plan_items = True
    assert b._graph_slots(256) == max(b.GRAPH_SLOTS_FLOOR, b.GRAPH_SLOTS_PER_REQ * 256) + 256
    b.plan_items, b.decode_ranges = False, 0                     # the range kernel refuses the shape: items after all
    assert b._graph_slots(16) == b.GRAPH_SLOTS_FLOOR + 16
    b.decode_ranges, b.max_context_len = 256, 2 ** 24            # bs x (context + 16) >= 2^31: the line does not fit
    assert b._ranges_for(256, 2 ** 24) == 0 and b._graph_slots(256) > 0


def test_plan_registry_trusts_only_live_plan_buffers():
    """_native remembers what each decode plan buffer was built with (data_ptr -> (bs, max_slots, ranges)) so that a launch
    given other values raises on the host (ABI 9).  A view of the buffer is the same plan; an entry whose tensor has died is
    dropped - the allocator may have handed the address to a buffer that was never built through decode_plan()."""
    import weakref
    from scratchpad_amd import _native
    t = torch.zeros(16, dtype=torch.int32)
    key = t.data_ptr()
    _native._BUILT_PLANS[key] = (weakref.ref(t), (4, 0, 256))
    assert _native._built_with(t) == (4, 0, 256) and _native._built_with(t[:8]) == (4, 0, 256)

    class SameAddress:                      # whatever the allocator puts at that address next
        def data_ptr(self):
            return key
    del t
    assert _native._built_with(SameAddress()) is None and key not in _native._BUILT_PLANS


def test_forward_batch_refuses_what_nothing_behind_it_reads():
    """ScheduleBatch.get_model_worker_batch forwards return_hidden_states (as CaptureHiddenMode.FULL) and
    input_embeds like the reference; nothing behind ForwardBatch.init_new reads either, so it raises instead of serving such
    a request as an ordinary one (the DP-attention refusal beside them was already there)."""
    from types import SimpleNamespace
    from scratchpad_amd.forward_info import CaptureHiddenMode, ForwardBatch, ForwardMode, ModelWorkerBatch
    runner = SimpleNamespace(device="cpu", req_to_token_pool=None, token_to_kv_pool=None, attn_backend=None)

    def batch(**kw):
        return ModelWorkerBatch(bid=0, forward_mode=ForwardMode.DECODE, input_ids=torch.zeros(2, dtype=torch.int64),
                                req_pool_indices=torch.zeros(2, dtype=torch.int64), seq_lens=torch.ones(2, dtype=torch.int64),
                                out_cache_loc=torch.zeros(2, dtype=torch.int64), seq_lens_sum=2, **kw)
    with pytest.raises(NotImplementedError, match="hidden states"):
        ForwardBatch.init_new(batch(capture_hidden_mode=CaptureHiddenMode.FULL), runner)
    with pytest.raises(NotImplementedError, match="input_embeds"):
        ForwardBatch.init_new(batch(input_embeds=torch.zeros(2, 8)), runner)
    with pytest.raises(NotImplementedError, match="DP attention"):
        ForwardBatch.init_new(batch(global_num_tokens=[2]), runner)
