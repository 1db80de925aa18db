"""fp8 (e5m2) KV pool - the reference's `--kv-cache-dtype fp8_e5m2` (memory/pool.py:274-280, 401-412;
model_runner.py:368-369).  The store is bit-exact against torch's own .to(float8_e5m2) (which IS
the reference's conversion); attention over the byte pool is compared with the oracle on the same,
exactly widened values, at the 16-bit tolerances."""
import math

import pytest
import torch

from oracle import llama as ollama
from oracle import ops
from tests import smoke_impl
from tests.helpers import assert_close, cpu, paged_problem

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_store_is_bitwise_torch_conversion(dtype):
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(5)
    T, Hkv, D, P = 37, 4, 64, 90
    k = (torch.randn(T, Hkv, D, generator=g) * 3).to(dtype)
    v = (torch.randn(T, Hkv, D, generator=g) * 0.01).to(dtype)
    # edge values: zeros, ties between e5m2 neighbours, subnormals, overflow, inf, nan
    special = torch.tensor([0.0, -0.0, 1.0, 1.125, 1.25, 1.375, 1.5, 1.625, 3.0e-5, 1.5e-5, 7.6e-6, 2.0e-6,
                            57344.0, 61439.0, 61440.0, 65504.0, 1.0e6, float("inf"), -float("inf"), float("nan"),
                            -1.125, -61440.0, 0.3, 1.0e-8], dtype=torch.float32).to(dtype)
    k.view(-1)[:special.numel()] = special
    loc = (torch.randperm(P, generator=g)[:T] + 1)
    kb = torch.zeros(P + 1, Hkv, D, dtype=torch.uint8, device=DEV)
    vb = torch.zeros(P + 1, Hkv, D, dtype=torch.uint8, device=DEV)
    _native.kv_store_fp8(kb, vb, loc.to(DEV), k.to(DEV), v.to(DEV))
    want_k = k.to(torch.float8_e5m2).view(torch.uint8)
    want_v = v.to(torch.float8_e5m2).view(torch.uint8)
    got_k, got_v = kb.cpu()[loc], vb.cpu()[loc]
    nan = torch.isnan(k.float())
    assert torch.equal(got_k[~nan], want_k[~nan]) and torch.equal(got_v, want_v)
    assert bool(torch.isnan(got_k[nan].view(torch.float8_e5m2).float()).all())
    assert int(kb.cpu()[0].sum()) == 0, "slot 0 untouched"
    # scales divide first, IN the activation dtype, then the cast (pool.py:403-408: cache_k.div_(k_scale);
    # cache_k.to(self.dtype)): two roundings - reproduced bit for bit also for scales that are not
    # powers of two
    for ks, vs in ((2.0, 0.5), (1.7, 0.37), (3.0, 1.0 / 3.0)):
        _native.kv_store_fp8(kb, vb, loc.to(DEV), k.to(DEV), v.to(DEV), ks, vs)
        fin = torch.isfinite(k.float())
        want_k = k.clone().div_(ks).to(torch.float8_e5m2).view(torch.uint8)
        want_v = v.clone().div_(vs).to(torch.float8_e5m2).view(torch.uint8)
        assert torch.equal(kb.cpu()[loc][fin], want_k[fin]), (ks, dtype)
        assert torch.equal(vb.cpu()[loc], want_v), (vs, dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_applies_the_layer_scales_of_a_scaled_fp8_pool(dtype):
    """flashinfer_backend.py:470-482 hands layer.k_scale / v_scale to the store AND to the kernels: the pool
    holds k / k_scale and v / v_scale, so decode and extend over it with the scales must equal attention
    over the re-scaled widened values (and differ from the unscaled call)."""
    from scratchpad_amd import _native
    Hq, Hkv, D = 8, 2, 128
    lens = [5, 64, 130, 300]
    bs = len(lens)
    ks, vs = 1.7, 0.37
    p = _fp8_problem(75, bs, Hq, Hkv, D, lens, dtype)
    c = cpu(p)
    kw = c["k_buffer"].view(torch.float8_e5m2).float() * ks
    vw = c["v_buffer"].view(torch.float8_e5m2).float() * vs
    scale = D ** -0.5
    ref = ops.decode_attention(c["q"].float(), kw, vw, c["req_to_token"], c["req_pool_indices"], c["seq_lens"], scale)
    vmax = float(vw.abs().max())
    for chunk in (64, 512):
        ws = torch.empty(_native.decode_workspace_bytes(bs, Hq, D, max(lens), chunk), dtype=torch.uint8, device=DEV)
        o = torch.full_like(p["q"], float("nan"))
        _native.decode_attention(o, p["q"], p["k_buffer"], p["v_buffer"], p["req_to_token"], p["req_pool_indices"],
                                 p["seq_lens"], scale, 0.0, max(lens), chunk, ws, None, None, k_scale=ks, v_scale=vs)
        assert_close(o, ref, dtype, what=f"scaled fp8 decode chunk {chunk}", vmax=vmax)
    o1 = torch.full_like(p["q"], float("nan"))
    ws = torch.empty(_native.decode_workspace_bytes(bs, Hq, D, max(lens), 64), dtype=torch.uint8, device=DEV)
    _native.decode_attention(o1, p["q"], p["k_buffer"], p["v_buffer"], p["req_to_token"], p["req_pool_indices"],
                             p["seq_lens"], scale, 0.0, max(lens), 64, ws)
    assert float((o1.float() - o.float()).abs().max()) > 0.05, "the scales change the result"
    # extend: every row of every request is new (prefix 0)
    g = torch.Generator().manual_seed(76)
    q = torch.randn(sum(lens), Hq, D, generator=g).to(dtype).to(DEV)
    ext_t = torch.tensor(lens, dtype=torch.int32, device=DEV)
    start = torch.zeros(bs, dtype=torch.int32, device=DEV)
    start[1:] = torch.cumsum(ext_t[:-1], 0)
    ws = torch.empty(_native.extend_workspace_bytes(sum(lens), bs, Hq, D, dtype), dtype=torch.uint8, device=DEV)
    oe = torch.full_like(q, float("nan"))
    _native.extend_attention(oe, q, p["k_buffer"], p["v_buffer"], p["req_to_token"], p["req_pool_indices"],
                             p["seq_lens"], ext_t, start, scale, 0.0, True, max(lens), max(lens), ws,
                             k_scale=ks, v_scale=vs)
    refe = ops.extend_attention(q.cpu().float(), kw, vw, c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                                ext_t.cpu(), start.cpu(), scale)
    assert_close(oe, refe, dtype, what="scaled fp8 extend", vmax=vmax)
    with pytest.raises(RuntimeError, match="invalid argument"):
        _native.decode_attention(o, p["q"], p["k_buffer"], p["v_buffer"], p["req_to_token"], p["req_pool_indices"],
                                 p["seq_lens"], scale, 0.0, max(lens), 64, ws, k_scale=0.0)


def _fp8_problem(seed, bs, Hq, Hkv, D, lens, dtype):
    p = paged_problem(seed, bs, Hq, Hkv, D, lens, dtype, DEV)
    for name in ("k_buffer", "v_buffer"):
        p[name] = p[name].float().to(torch.float8_e5m2).view(torch.uint8)
    return p


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (16, 2, 64), (8, 8, 64)])
def test_decode_over_fp8_pool(dtype, Hq, Hkv, D):
    from scratchpad_amd import _native
    lens = [1, 2, 63, 64, 65, 127, 129, 255, 300, 513, 1000, 17]
    bs = len(lens)
    p = _fp8_problem(71, bs, Hq, Hkv, D, lens, dtype)
    scale = 1.0 / math.sqrt(D)
    c = cpu(p)
    ref = ops.decode_attention(c["q"].float(), c["k_buffer"], c["v_buffer"], c["req_to_token"],
                               c["req_pool_indices"], c["seq_lens"], scale)
    vmax = float(c["v_buffer"].view(torch.float8_e5m2).float().abs().max())
    for chunk, use_plan in ((64, True), (512, False)):
        max_len = max(lens)
        ws = torch.empty(_native.decode_workspace_bytes(bs, Hq, D, max_len, chunk), dtype=torch.uint8, device=DEV)
        plan = None
        if use_plan:
            plan = torch.empty(_native.decode_plan_bytes(bs, max_len, chunk) // 4, dtype=torch.int32, device=DEV)
            _native.decode_plan(plan, p["seq_lens"], max_len, chunk)
        o = torch.full_like(p["q"], float("nan"))
        kb, vb = p["k_buffer"], p["v_buffer"]
        if use_plan:        # what MHATokenToKVPool.get_kv_buffer hands out: float8_e5m2 views of the byte pool
            kb, vb = kb.view(torch.float8_e5m2), vb.view(torch.float8_e5m2)
        _native.decode_attention(o, p["q"], kb, vb, p["req_to_token"],
                                 p["req_pool_indices"], p["seq_lens"], scale, 0.0, max_len, chunk, ws, None, plan)
        assert_close(o, ref, dtype, what=f"fp8 decode chunk {chunk}", vmax=vmax)
    with pytest.raises(RuntimeError, match="does not go with"):
        _native.decode_attention(torch.empty_like(p["q"].float()), p["q"].float(), p["k_buffer"], p["v_buffer"],
                                 p["req_to_token"], p["req_pool_indices"], p["seq_lens"], scale, 0.0, max_len, 64, ws)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (64, 8, 128), (8, 8, 64)])
def test_decode_range_geometry_over_fp8_pool(dtype, Hq, Hkv, D):
    """the range kernel on a byte pool (ABI 8: one workgroup per (piece of the step's keys, four kv heads); e5m2 widened
    on the way into the LDS tile, fp16 tile math): the oracle on the widened pool, for the piece count the library asks
    for and for 1, 5 and 300 pieces; same bits launch after launch and on int64 index tensors; 2 units from the
    (request, split) items of the same plan"""
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(Hq + D)
    lens = [1400, 1025, 64, 65, 1, 2, 513] + torch.randint(1, 500, (29,), generator=g).tolist()
    bs, chunk, max_len = len(lens), 64, 1400
    p = _fp8_problem(73, bs, Hq, Hkv, D, lens, dtype)
    scale = 1.0 / math.sqrt(D)
    c = cpu(p)
    ref = ops.decode_attention(c["q"].float(), c["k_buffer"], c["v_buffer"], c["req_to_token"],
                               c["req_pool_indices"], c["seq_lens"], scale)
    vmax = float(c["v_buffer"].view(torch.float8_e5m2).float().abs().max())
    kb, vb = p["k_buffer"].view(torch.float8_e5m2), p["v_buffer"].view(torch.float8_e5m2)
    auto = _native.decode_ranges(Hq, Hkv, D, dtype, kb.dtype)
    assert auto > 0
    outs = {}
    for ranges in (0, 1, 5, auto, 300):
        ws = torch.empty(_native.decode_workspace_bytes(bs, Hq, D, max_len, chunk, None, ranges), dtype=torch.uint8, device=DEV)
        ws.fill_(0x7f)
        plan = torch.empty(_native.decode_plan_bytes(bs, max_len, chunk, None, ranges) // 4, dtype=torch.int32, device=DEV)
        for idt in (torch.int32, torch.int64, torch.int32):
            _native.decode_plan(plan, p["seq_lens"].to(idt), max_len, chunk, None, ranges)
            o = torch.full_like(p["q"], float("nan"))
            _native.decode_attention(o, p["q"], kb, vb, p["req_to_token"], p["req_pool_indices"].to(idt),
                                     p["seq_lens"].to(idt), scale, 0.0, max_len, chunk, ws, None, plan, ranges=ranges)
            assert torch.equal(o, outs.setdefault(ranges, o)), (ranges, idt)
        assert_close(outs[ranges], ref, dtype, what=f"fp8 range decode, {ranges} pieces", vmax=vmax)
    u = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]
    for ranges in (1, 5, auto, 300):
        assert float((outs[ranges].float() - outs[0].float()).abs().max()) <= 2.5 * u * vmax, ranges


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (8, 2, 64)])
def test_extend_over_fp8_pool(dtype, Hq, Hkv, D):
    from scratchpad_amd import _native
    pre = [0, 64, 300, 0, 5]
    ext = [130, 1, 70, 33, 257]
    bs = len(pre)
    seq = [a + b for a, b in zip(pre, ext)]
    p = _fp8_problem(72, bs, Hq, Hkv, D, seq, dtype)
    g = torch.Generator().manual_seed(73)
    q = torch.randn(sum(ext), Hq, D, generator=g).to(dtype).to(DEV)
    ext_t = torch.tensor(ext, dtype=torch.int32, device=DEV)
    start = torch.zeros(bs, dtype=torch.int32, device=DEV)
    start[1:] = torch.cumsum(ext_t[:-1], 0)
    scale = D ** -0.5
    c = cpu(p)
    vmax = float(c["v_buffer"].view(torch.float8_e5m2).float().abs().max())
    for window in (-1, 40):
        ws = torch.empty(_native.extend_workspace_bytes(sum(ext), bs, Hq, D, dtype), dtype=torch.uint8, device=DEV)
        o = torch.full_like(q, float("nan"))
        _native.extend_attention(o, q, p["k_buffer"], p["v_buffer"], p["req_to_token"], p["req_pool_indices"],
                                 p["seq_lens"], ext_t, start, scale, 0.0, True, max(ext), max(seq), ws, None,
                                 window_left=window)
        ref = ops.extend_attention(q.cpu().float(), c["k_buffer"], c["v_buffer"], c["req_to_token"],
                                   c["req_pool_indices"], c["seq_lens"], ext_t.cpu(), start.cpu(), scale,
                                   window_left=window)
        assert_close(o, ref, dtype, what=f"fp8 extend window {window}", vmax=vmax)


@pytest.mark.parametrize("dtype,bar", [(torch.float16, 2e-3), (torch.bfloat16, 3e-2)])
def test_tiny_llama_with_fp8_kv_cache(dtype, bar):
    """End to end through ModelRunner(kv_cache_dtype="fp8_e5m2"): prefill + 3 decode steps (eager),
    against the oracle model whose pool quantises the same way."""
    from scratchpad_amd.forward_info import ForwardMode
    from scratchpad_amd.model_runner import ModelConfig, ModelRunner, ServerArgs, TpModelWorker
    from scratchpad_amd.schedule_batch import Req, ScheduleBatch
    g, pfx, shape, w = smoke_impl.load_case("a")
    scaling = None
    if shape.rope_scaling is not None:
        f = shape.rope_scaling
        scaling = {"rope_type": "llama3", "factor": f[0], "low_freq_factor": f[1], "high_freq_factor": f[2],
                   "original_max_position_embeddings": int(f[3])}
    cfg = ModelConfig(shape.hidden, shape.inter, shape.layers, shape.Hq, shape.Hkv, shape.vocab, context_len=60,
                      rms_norm_eps=shape.rms_eps, rope_theta=shape.rope_theta, rope_scaling=scaling,
                      max_position_embeddings=shape.max_pos, tie_word_embeddings=shape.tie)
    mr = ModelRunner(cfg, ServerArgs(max_total_tokens=96, max_running_requests=3, disable_cuda_graph=True,
                                     kv_cache_dtype="fp8_e5m2"), dtype=dtype, init_weights=False)
    assert mr.token_to_kv_pool.store_dtype == torch.uint8 and mr.token_to_kv_pool.dtype == torch.float8_e5m2
    mr.model.load_full_state_dict({k: v.to(mr.device) for k, v in w.items()})
    worker = TpModelWorker(mr)
    wd = {k: v.to(dtype) for k, v in w.items()}
    okv = ollama.OracleKV(shape, 96, 4, 64, dtype=torch.uint8)
    gen = torch.Generator().manual_seed(9)
    reqs = [Req(str(i), "", torch.randint(0, shape.vocab, (n,), generator=gen).tolist(), None) for i, n in enumerate((9, 5))]
    sb = ScheduleBatch(reqs, mr.req_to_token_pool, mr.token_to_kv_pool_allocator, device=mr.device)
    sb.prepare_for_extend()
    out, nxt = worker.forward_batch_generation(sb.get_model_worker_batch())
    okv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu())
    pos, start = ops.compute_position(torch.tensor([0, 0], dtype=torch.int32), torch.tensor([9, 5], dtype=torch.int32))
    ref = ollama.forward(shape, wd, okv, mode="extend", input_ids=sb.input_ids.cpu(), positions=pos,
                         req_pool_indices=sb.req_pool_indices.cpu(), seq_lens=sb.seq_lens.cpu(),
                         out_cache_loc=sb.out_cache_loc.cpu(), extend_seq_lens=torch.tensor([9, 5], dtype=torch.int32),
                         extend_start_loc=start)
    rel = lambda a, b: float((a.float().cpu() - b.float()).abs().max() / b.float().abs().max())
    assert rel(out.next_token_logits, ref) <= bar
    # the byte pools agree exactly where the rounded inputs agree; compare widened values loosely
    kk = mr.token_to_kv_pool.get_key_buffer(0).cpu().float()
    assert rel(kk[sb.out_cache_loc.cpu()], okv.k[0].view(torch.float8_e5m2).float()[sb.out_cache_loc.cpu()]) <= 0.26
    for step in range(3):
        sb.output_ids = ref.argmax(-1).to(mr.device)     # feed the oracle's tokens: both sides see the same inputs
        sb.prepare_for_decode()
        out, _ = worker.forward_batch_generation(sb.get_model_worker_batch())
        okv.req_to_token.copy_(mr.req_to_token_pool.req_to_token.cpu())
        ref = ollama.forward(shape, wd, okv, mode="decode", input_ids=sb.input_ids.cpu(),
                             positions=ops.clamp_position(sb.seq_lens.cpu()), req_pool_indices=sb.req_pool_indices.cpu(),
                             seq_lens=sb.seq_lens.cpu(), out_cache_loc=sb.out_cache_loc.cpu())
        assert rel(out.next_token_logits, ref) <= bar, step
