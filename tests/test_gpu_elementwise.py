"""GPU parity (through the C ABI) of the row kernels: RMSNorm (+fused add), SiLU-mul, rotary
(+fused KV store), KV store, req_to_token scatter, positions.

* fp32: compared directly with the golden vectors recorded from the reference (<= 2e-6);
* fp16 / bf16: the golden INPUTS are exactly representable in both (tests/golden/gen_golden.py
  `_grid`), so the same inputs are fed and compared with the oracle evaluated in that dtype -
  the kernels reproduce torch's rounding points, so agreement is to one ulp of the dtype;
* integer / index work: bit-exact."""
import numpy as np
import pytest
import torch

from oracle import ops
from tests import golden
from tests.helpers import DTYPES, T, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def nat():
    from scratchpad_amd import _native
    _native.load()
    return _native


def ulp_close(hip, ref, dtype, what=""):
    """equal up to 1 ulp of `dtype` (fp32: 2e-6 relative)."""
    hip, ref = hip.float().cpu(), ref.float().cpu()
    assert hip.shape == ref.shape
    eps = {torch.float32: 2e-6, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
    err = (hip - ref).abs()
    bound = eps * ref.abs() + (1e-6 if dtype == torch.float32 else 1e-4)
    assert bool((err <= bound).all()), f"{what}: max err {float(err.max()):.3e}"


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_rmsnorm_golden(nat, dt):
    dtype = DTYPES[dt]
    g = golden.load("rmsnorm")
    for i in range(int(g["num_cases"])):
        x, w, r = (T(g[f"c{i}_{n}"], DEV, dtype) for n in ("x", "w", "res"))
        eps = float(g[f"c{i}_eps"])
        y = nat.rmsnorm(x, w, eps)
        ref = ops.rmsnorm(x.cpu(), w.cpu(), eps)
        ulp_close(y, ref, dtype, f"rmsnorm c{i}")
        if dtype == torch.float32:
            ulp_close(y, T(g[f"c{i}_y"]), dtype, f"rmsnorm golden c{i}")
        x2, r2 = x.clone(), r.clone()
        nat.fused_add_rmsnorm(x2, r2, w, eps)          # in place on both
        yref, rref = ops.rmsnorm(x.cpu(), w.cpu(), eps, r.cpu())
        ulp_close(x2, yref, dtype, f"fused y c{i}")
        assert torch.equal(r2.cpu(), rref), "residual' = round(x + residual) must be bit-exact"
        if dtype == torch.float32:
            ulp_close(x2, T(g[f"c{i}_y_fused"]), dtype)
            assert np.array_equal(r2.cpu().numpy(), g[f"c{i}_res_out"])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_rmsnorm_module_contract_and_odd_shapes(nat, dt):
    from scratchpad_amd.layers import RMSNorm
    dtype = DTYPES[dt]
    torch.manual_seed(0)
    for T_, H in [(3, 100), (2, 4104), (5, 8192), (1, 12288), (4, 7)]:
        m = RMSNorm(H, 1e-5).to(DEV)
        m.weight.data = (torch.randn(H, device=DEV) * 0.3 + 1).to(dtype)
        x = torch.randn(T_, H, device=DEV).to(dtype)
        r = torch.randn(T_, H, device=DEV).to(dtype)
        y = m(x)
        assert y.data_ptr() != x.data_ptr()
        ulp_close(y, ops.rmsnorm(x.cpu(), m.weight.data.cpu(), 1e-5), dtype, f"H={H}")
        x0, r0 = x.clone(), r.clone()
        y2, r2 = m(x, r)
        assert y2.data_ptr() == x.data_ptr() and r2.data_ptr() == r.data_ptr(), "in-place contract"
        yref, rref = ops.rmsnorm(x0.cpu(), m.weight.data.cpu(), 1e-5, r0.cpu())
        ulp_close(y2, yref, dtype, f"fused H={H}")
        assert torch.equal(r2.cpu(), rref)
    # strided rows (a column slice of a wider buffer)
    big = torch.randn(6, 512, device=DEV).to(dtype)
    x = big[:, 128:384]
    w = torch.ones(256, device=DEV, dtype=dtype)
    ulp_close(nat.rmsnorm(x, w, 1e-6), ops.rmsnorm(x.cpu(), w.cpu(), 1e-6), dtype, "strided")
    assert nat.rmsnorm(torch.empty(0, 64, device=DEV, dtype=dtype), w[:64], 1e-6).shape == (0, 64)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_silu_mul(nat, dt):
    dtype = DTYPES[dt]
    g = golden.load("silu_mul")
    for i in range(int(g["num_cases"])):
        x = T(g[f"c{i}_x"], DEV, dtype)
        y = nat.silu_and_mul(x)
        ulp_close(y, ops.silu_and_mul(x.cpu()), dtype, f"silu c{i}")
        if dtype == torch.float32:
            ulp_close(y, T(g[f"c{i}_y"]), dtype)
    torch.manual_seed(1)
    for T_, d in [(3, 14336), (5, 100), (2, 7), (257, 64)]:
        x = (torch.randn(T_, 2 * d, device=DEV) * 3).to(dtype)
        ulp_close(nat.silu_and_mul(x), ops.silu_and_mul(x.cpu()), dtype, f"d={d}")
    x3 = (torch.randn(2, 3, 64, device=DEV)).to(dtype)     # leading dims are kept
    assert nat.silu_and_mul(x3).shape == (2, 3, 32)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_rotary_golden_and_cache(nat, dt):
    from scratchpad_amd.layers import get_rope
    dtype = DTYPES[dt]
    g = golden.load("rotary")
    for i in range(int(g["num_cases"])):
        hs, rd, mp = (int(g[f"c{i}_{n}"]) for n in ("head_size", "rotary_dim", "max_pos"))
        sc = golden.rope_scaling(g, i)           # None / llama3 / linear / dynamic NTK / YaRN
        rope = get_rope(hs, rd, mp, float(g[f"c{i}_base"]), bool(g[f"c{i}_neox"]), sc, dtype=dtype)
        if dtype == torch.float32:   # cache = the reference's, bit for bit (host fp32 math)
            assert np.array_equal(rope.cos_sin_cache.cpu().numpy(), g[f"c{i}_cos_sin_cache"])
        pos = T(g[f"c{i}_positions"], DEV)
        q, k = T(g[f"c{i}_q"], DEV, dtype), T(g[f"c{i}_k"], DEV, dtype)
        q0, k0 = q.clone(), k.clone()
        q2, k2 = rope(pos, q, k)
        assert q2.data_ptr() == q.data_ptr() and k2.data_ptr() == k.data_ptr(), "in place"
        qr, kr = ops.rotary_embedding(pos.cpu(), q0.cpu(), k0.cpu(), hs, rope.cos_sin_cache.cpu(),
                                      bool(g[f"c{i}_neox"]))
        ulp_close(q, qr, dtype, f"rotary q c{i}")
        ulp_close(k, kr, dtype, f"rotary k c{i}")
        if dtype == torch.float32:
            ulp_close(q, T(g[f"c{i}_q_out"]), dtype)
            ulp_close(k, T(g[f"c{i}_k_out"]), dtype)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_rotary_on_qkv_views_and_fused_kv_store(nat, dt):
    """q,k are column slices of the fused qkv GEMM output (llama.py:149-150); the fused variant
    also scatters rotated k and v into the pool (= rotary then set_kv_buffer)."""
    dtype = DTYPES[dt]
    torch.manual_seed(2)
    T_, Hq, Hkv, D, P = 9, 8, 2, 128, 40
    cache = ops.rope_cos_sin_cache(64, 500000.0, D, None, dtype).to(DEV)
    qkv = torch.randn(T_, (Hq + 2 * Hkv) * D, device=DEV).to(dtype)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    pos = torch.randint(0, 64, (T_,), device=DEV)
    loc = (torch.randperm(P, device=DEV)[:T_] + 1).to(torch.int64)
    qkv0 = qkv.clone()
    q0, k0, v0 = qkv0.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    kb = torch.zeros(P + 1, Hkv, D, device=DEV, dtype=dtype)
    vb = torch.zeros_like(kb)
    nat.rotary_embedding(pos, q, k, D, cache, True, value=v, k_buffer=kb, v_buffer=vb, out_cache_loc=loc)
    qr, kr = ops.rotary_embedding(pos.cpu(), q0.cpu(), k0.cpu(), D, cache.cpu(), True)
    ulp_close(q, qr, dtype, "q view")
    ulp_close(k, kr, dtype, "k view")
    assert torch.equal(v, v0), "v must be untouched"
    assert torch.equal(kb[loc].reshape(T_, -1), k), "pool K rows == rotated k"
    assert torch.equal(vb[loc].reshape(T_, -1), v), "pool V rows == v"
    untouched = torch.ones(P + 1, dtype=torch.bool)
    untouched[loc.cpu()] = False
    assert not kb.cpu()[untouched].any() and not vb.cpu()[untouched].any()


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
def test_kv_store_bit_exact(nat, dt):
    from types import SimpleNamespace
    from scratchpad_amd.pool import MHATokenToKVPool
    dtype = DTYPES[dt]
    g = golden.load("kv_pool")
    L, size, H, D = (int(g[k]) for k in ("layer_num", "size", "head_num", "head_dim"))
    pool = MHATokenToKVPool(size, 1, dtype, H, D, L, DEV)
    for layer in range(L):
        loc = T(g[f"l{layer}_loc"], DEV)
        k, v = T(g[f"l{layer}_k"], DEV, dtype), T(g[f"l{layer}_v"], DEV, dtype)
        if layer == 2:   # the padded-row write to dummy slot 0 races with nothing else here
            pass
        pool.set_kv_buffer(SimpleNamespace(layer_id=layer), loc, k, v)
    for layer in range(L):
        assert torch.equal(pool.get_key_buffer(layer).cpu(), T(g[f"l{layer}_k_buffer"], dtype=dtype))
        assert torch.equal(pool.get_value_buffer(layer).cpu(), T(g[f"l{layer}_v_buffer"], dtype=dtype))
    # a production-size row (8 KV heads x 128) through strided [T, Hkv, D] views
    pool = MHATokenToKVPool(300, 1, dtype, 8, 128, 1, DEV)
    torch.manual_seed(3)
    qkv = torch.randn(77, 6144, device=DEV).to(dtype)
    k, v = qkv[:, 4096:5120].view(77, 8, 128), qkv[:, 5120:].view(77, 8, 128)
    loc = torch.randperm(300, device=DEV)[:77] + 1
    pool.set_kv_buffer(SimpleNamespace(layer_id=0), loc, k, v)
    assert torch.equal(pool.get_key_buffer(0)[loc], k) and torch.equal(pool.get_value_buffer(0)[loc], v)


def test_positions_and_req_to_token_bit_exact(nat):
    g = golden.load("positions")
    pos, start = nat.compute_position(T(g["prefix_lens"], DEV), T(g["extend_lens"], DEV),
                                      int(g["extend_lens"].sum()))
    assert pos.dtype == torch.int64 and np.array_equal(pos.cpu().numpy(), g["positions"])
    assert start.dtype == torch.int32 and np.array_equal(start.cpu().numpy(), g["extend_start_loc"])
    for dt in (torch.int64, torch.int32):
        out = nat.clamp_position(T(g["decode_seq_lens"], DEV).to(dt))
        assert np.array_equal(out.cpu().numpy(), g["decode_positions"])
    table = T(g["w_table_in"], DEV).clone()
    nat.write_req_to_token(table, T(g["w_req_pool_indices"], DEV), T(g["w_pre_lens"], DEV),
                           T(g["w_seq_lens"], DEV), T(g["w_extend_lens"], DEV),
                           T(g["w_out_cache_loc"], DEV))
    assert np.array_equal(table.cpu().numpy(), g["w_table_out"])
    # larger ragged batch vs the oracle (prefix sums across > 256 requests, empty rows)
    gen = torch.Generator().manual_seed(4)
    bs = 700
    pre = torch.randint(0, 50, (bs,), generator=gen, dtype=torch.int32)
    ext = torch.randint(0, 300, (bs,), generator=gen, dtype=torch.int32)
    ext[5] = 0
    pos, start = nat.compute_position(pre.to(DEV), ext.to(DEV), int(ext.sum()))
    rpos, rstart = ops.compute_position(pre, ext)
    assert torch.equal(pos.cpu(), rpos) and torch.equal(start.cpu(), rstart)
    table = torch.zeros(bs + 5, 360, dtype=torch.int32)
    req = torch.randperm(bs + 5, generator=gen)[:bs]
    loc = torch.randperm(int(ext.sum()) + 10, generator=gen)[: int(ext.sum())] + 1
    seq = (pre + ext).long()
    ref = table.clone()
    ops.write_req_to_token(ref, req, pre.long(), seq, ext.long(), loc)
    tg = table.to(DEV)
    nat.write_req_to_token(tg, req.to(DEV), pre.long().to(DEV), seq.to(DEV), ext.long().to(DEV), loc.to(DEV))
    assert torch.equal(tg.cpu(), ref)
