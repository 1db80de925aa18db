"""BASELINE.json's configurations at their FULL sizes, through size-independent properties plus spot
rows against the oracle (the oracle itself cannot run these sizes in test time):

  config 2  Llama-3-8B head shape, bs = 256 decode, contexts U[128, 4096], random slot permutation
  config 3  Llama-3-8B head shape, bs = 64 ragged prefill, prompt lengths U[128, 4096]
  config 4  Llama-3-70B at TP = 8, one rank's head shape (Hq 8 / Hkv 1), bs = 128 decode, contexts U[128, 4096]

The decode cases run the form that SHIPS: split plan + separate merge launch, K/V as the two strided views of one
interleaved [P+1, 2, Hkv, D] arena, non-temporal gathers at the library's default.

16-bit outputs are held to the error model of helpers.attn_error_units (units of u * (|ref| + A))."""
import pytest
import torch

from oracle import ops
from tests.helpers import DTYPES, assert_attn_close, attn_error_units

pytestmark = pytest.mark.gpu
DEV = "cuda"
HQ, HKV, D = 32, 8, 128
SCALE = D ** -0.5


@pytest.fixture(scope="module")
def nat():
    from scratchpad_amd import _native
    _native.load()
    return _native


def big_pool(seed, lens, dtype, extra=64, hkv=HKV):
    """K/V pools and a fragmented req_to_token for the given context lengths, generated on the GPU.  The pool is
    what MHATokenToKVPool._create_buffers makes for one layer by default: ONE arena [P+1, 2, Hkv, D] (a token's K row
    and V row adjacent) whose two strided views are the K and the V buffer the kernels are handed."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    gc = torch.Generator().manual_seed(seed)
    total = sum(lens)
    P = total + extra
    arena = torch.empty(P + 1, 2, hkv, D, dtype=dtype, device=DEV)
    arena[:, 0].copy_(torch.empty(P + 1, hkv, D, dtype=dtype, device=DEV).normal_(0, 1, generator=g))
    arena[:, 1].copy_(torch.empty(P + 1, hkv, D, dtype=dtype, device=DEV).normal_(0, 1, generator=g))
    kb, vb = arena[:, 0], arena[:, 1]
    assert kb.stride(0) == vb.stride(0) == 2 * hkv * D and not kb.is_contiguous()
    perm = (torch.randperm(P, generator=gc) + 1).to(torch.int32)
    bs = len(lens)
    r2t = torch.zeros(bs, max(lens) + 8, dtype=torch.int32)
    off = 0
    for b, n in enumerate(lens):
        r2t[b, :n] = perm[off:off + n]
        off += n
    return kb, vb, r2t.to(DEV)


def relocate(kb, vb, r2t, gen):
    """every row of the pool (and the table) moved to another slot, in a second interleaved arena"""
    P1, hkv = kb.shape[0], kb.shape[1]
    perm = torch.randperm(P1 - 1, generator=gen).to(DEV) + 1
    perm = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), perm])
    arena2 = torch.empty(P1, 2, hkv, D, dtype=kb.dtype, device=DEV)
    arena2[perm, 0], arena2[perm, 1] = kb, vb
    return arena2[:, 0], arena2[:, 1], perm[r2t.long()].to(torch.int32)


def oracle_rows(q_rows, kb, vb, r2t, req_rows, seq_rows, abs_v=False):
    """fp32 oracle for single query rows: row i attends to the first seq_rows[i] keys of request req_rows[i]."""
    v = vb.float().cpu()
    return ops.decode_attention(q_rows.float().cpu(), kb.float().cpu(), v.abs() if abs_v else v, r2t.cpu(),
                                torch.tensor(req_rows), torch.tensor(seq_rows), SCALE)


def run_decode(nat, q, kb, vb, r2t, req, seq, chunk, plan=True, ranges=0):
    """A plan built once per step, the matrix-core kernel writing partials, and the separate merge launch; non-temporal
    gathers as the library defaults them.  ranges > 0: THE SHIPPED FORM where the range kernel takes the shape
    (HipAttnBackend passes sp_decode_ranges()): the plan's range geometry, one workgroup per (piece of the step's keys,
    four kv heads).  ranges = 0: the plan's (request, split) items at split size `chunk` (what the backend runs on the
    other shapes).  plan=False: the plan-less static (request, split) grid."""
    bs, hq = q.shape[0], q.shape[1]
    max_len = int(seq.max())
    ws = torch.empty(nat.decode_workspace_bytes(bs, hq, D, max_len, chunk, None, ranges), dtype=torch.uint8, device=DEV)
    o = torch.full_like(q, float("nan"))
    pl = None
    if plan:
        pl = torch.empty(nat.decode_plan_bytes(bs, max_len, chunk, None, ranges) // 4, dtype=torch.int32, device=DEV)
        nat.decode_plan(pl, seq, max_len, chunk, None, ranges)
    nat.decode_attention(o, q, kb, vb, r2t, req, seq, SCALE, 0.0, max_len, chunk, ws, None, pl, ranges=ranges)
    return o


def full_size_decode_checks(nat, dt, bs, hq, hkv, rows, what):
    """contexts U[128, 4096] seed 0 (bench.py's), random slot permutation, interleaved arena; (a) spot rows against
    the fp32 oracle, (b) split invariance, (c) slot relocation bit-exact, (d) request-order invariance bit-exact"""
    dtype = DTYPES[dt]
    gen = torch.Generator().manual_seed(0)
    lens = torch.randint(128, 4097, (bs,), generator=gen).tolist()
    lens[3], lens[bs - 56] = 4096, 128
    kb, vb, r2t = big_pool(2, lens, dtype, hkv=hkv)
    q = torch.randn(bs, hq, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).to(dtype)
    req = torch.arange(bs, device=DEV)
    seq = torch.tensor(lens, device=DEV)
    o512 = run_decode(nat, q, kb, vb, r2t, req, seq, 512)
    assert torch.isfinite(o512.float()).all()
    # (a) spot rows against the fp32 oracle, error model without a max-scaled allowance
    ref = oracle_rows(q[rows], kb, vb, r2t, rows, [lens[i] for i in rows])
    aref = oracle_rows(q[rows], kb, vb, r2t, rows, [lens[i] for i in rows], abs_v=True)
    assert_attn_close(o512[rows], ref, aref, dtype, what=f"{what} {dt}: {len(rows)} rows vs oracle")
    # (b) split invariance at full size: the split size changes the partial sums, not the softmax; each
    # result is within the model of the oracle rows, and the two roundings differ by at most 2 units
    # (768: the split cap HipAttnBackend ships since round 4; 4096: every request in ONE split - what the backend
    # picks for near-uniform batches - so no partials and no merge at all; 256 also runs plan-less)
    u = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]
    for chunk in (64, 256, 768, 4096):
        oc = run_decode(nat, q, kb, vb, r2t, req, seq, chunk)
        assert_attn_close(oc[rows], ref, aref, dtype, what=f"{what} {dt}: chunk {chunk} rows vs oracle")
        diff = (oc.float() - o512.float()).abs()
        assert float(diff.max()) <= 2.5 * u * float(vb.float().abs().max()), f"chunk {chunk} vs 512: {float(diff.max()):.3e}"
        if chunk == 256:
            assert torch.equal(run_decode(nat, q, kb, vb, r2t, req, seq, chunk, plan=False), oc), "plan-less grid"
    # (c) KV page indexing is bit-exact: relocate every row of the pool (and the table) - same bits
    kb2, vb2, r2t2 = relocate(kb, vb, r2t, gen)
    assert torch.equal(run_decode(nat, q, kb2, vb2, r2t2, req, seq, 512), o512)
    # (d) request-order invariance: a permuted batch gives the permuted rows, bit for bit
    order = torch.randperm(bs, generator=gen).to(DEV)
    o_perm = run_decode(nat, q[order].contiguous(), kb, vb, r2t, req[order].contiguous(), seq[order].contiguous(), 512)
    assert torch.equal(o_perm, o512[order])
    # (e) the layout is not part of the result: the same rows in two separate contiguous K / V buffers - same bits
    assert torch.equal(run_decode(nat, q, kb.contiguous(), vb.contiguous(), r2t, req, seq, 512), o512)
    # (f) the range geometry (what HipAttnBackend ships where sp_decode_ranges() > 0: both shapes), at the piece
    # count the library asks for and at two others: the oracle rows, at most 2 units from the split results, slot
    # relocation bit-exact; a permuted batch is cut at other places - another split of the same sums: 2 units
    auto = nat.decode_ranges(hq, hkv, D, dtype)
    assert auto > 0
    if auto:
        for ranges in (auto, 61, 1500):
            orr = run_decode(nat, q, kb, vb, r2t, req, seq, 512, ranges=ranges)
            assert torch.isfinite(orr.float()).all()
            assert_attn_close(orr[rows], ref, aref, dtype, what=f"{what} {dt}: {ranges} ranges, rows vs oracle")
            diff = (orr.float() - o512.float()).abs()
            assert float(diff.max()) <= 2.5 * u * float(vb.float().abs().max()), f"{ranges} ranges vs chunk 512: {float(diff.max()):.3e}"
            if ranges == auto:
                assert torch.equal(run_decode(nat, q, kb2, vb2, r2t2, req, seq, 512, ranges=ranges), orr)
                op = run_decode(nat, q[order].contiguous(), kb, vb, r2t, req[order].contiguous(), seq[order].contiguous(),
                                512, ranges=ranges)
                diff = (op.float() - orr[order].float()).abs()
                assert float(diff.max()) <= 2.5 * u * float(vb.float().abs().max()), f"permuted batch: {float(diff.max()):.3e}"


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_config2_decode_bs256_full_size(nat, dt):
    """config 2: Llama-3-8B heads, bs 256 - the HPW branch of decode_mfma_kernel (a wave per kv head)"""
    full_size_decode_checks(nat, dt, 256, HQ, HKV, [0, 3, 17, 64, 129, 200, 254, 255], "config 2 bs=256")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_config4_rank_shape_decode_bs128_full_size(nat, dt):
    """config 4's per-rank shape (Llama-3-70B at TP = 8: Hq 8 / Hkv 1, bs 128): the non-HPW branch of
    decode_mfma_kernel - the 4 waves share one head's keys and merge (m, l, O) through LDS behind two barriers"""
    full_size_decode_checks(nat, dt, 128, 8, 1, [0, 3, 17, 64, 72, 100, 126, 127], "config 4 rank shape bs=128")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_config3_extend_bs64_full_size(nat, dt):
    dtype = DTYPES[dt]
    gen = torch.Generator().manual_seed(0)
    bs = 64
    lens = torch.randint(128, 4097, (bs,), generator=gen).tolist()       # bench.py's prompt lengths (seed 0)
    lens[5], lens[40] = 4096, 128
    kb, vb, r2t = big_pool(4, lens, dtype)
    T = sum(lens)
    q = torch.randn(T, HQ, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)).to(dtype)
    req = torch.arange(bs, device=DEV)
    seq = torch.tensor(lens, device=DEV)
    ext = seq.to(torch.int32)
    start = torch.zeros(bs, dtype=torch.int32, device=DEV)
    start[1:] = torch.cumsum(ext[:-1], 0)
    ws = torch.empty(nat.extend_workspace_bytes(T, bs, HQ, D, dtype), dtype=torch.uint8, device=DEV)

    def run(plan):
        o = torch.full_like(q, float("nan"))
        nat.extend_attention(o, q, kb, vb, r2t, req, seq, ext, start, SCALE, 0.0, True, max(lens), max(lens), ws,
                             plan=plan)
        return o
    plan = nat.extend_plan(ext, seq, T, HQ, HKV, True)
    o = run(plan)
    assert torch.isfinite(o.float()).all()
    assert torch.equal(run(None), o), "the work plan changes nothing"
    # the LDS-DMA ring's waits and barriers are counted by hand: a miscount is a race, and a race shows as
    # run-to-run differences at this size (17,976 workgroups x 8 kv heads, every CU busy) long before it shows
    # in a small case; the register-staged path must give the same bits
    for _ in range(12):
        assert torch.equal(run(plan), o), "LDS-DMA ring: run-to-run difference"
    try:
        nat.debug_set("extend_dma", 0)
        assert torch.equal(run(plan), o), "register-staged tiles vs LDS-DMA ring"
    finally:
        nat.debug_set("extend_dma", 1)
    starts = start.cpu().tolist()
    # (a) spot rows (first, interior, block edges, last) of several requests against the fp32 oracle
    spots = [(5, 0), (5, 63), (5, 64), (5, 2047), (5, 4095), (40, 127), (0, lens[0] - 1), (17, lens[17] // 2),
             (63, lens[63] - 1), (33, 1)]
    tok = [starts[b] + t for b, t in spots]
    ref = oracle_rows(q[tok], kb, vb, r2t, [b for b, _ in spots], [t + 1 for _, t in spots])
    aref = oracle_rows(q[tok], kb, vb, r2t, [b for b, _ in spots], [t + 1 for _, t in spots], abs_v=True)
    assert_attn_close(o[tok], ref, aref, dtype, what=f"config 3 bs=64 {dt}: {len(spots)} rows vs oracle")
    # (b) the last row of EVERY prompt is a decode step over the same keys: decode kernel vs extend kernel
    last = [starts[b] + lens[b] - 1 for b in range(bs)]
    od = torch.full_like(q[last], float("nan"))
    wsd = torch.empty(nat.decode_workspace_bytes(bs, HQ, D, max(lens), 512), dtype=torch.uint8, device=DEV)
    nat.decode_attention(od, q[last].contiguous(), kb, vb, r2t, req, seq, SCALE, 0.0, max(lens), 512, wsd)
    aall = oracle_rows(q[last], kb, vb, r2t, list(range(bs)), lens, abs_v=True)
    rall = oracle_rows(q[last], kb, vb, r2t, list(range(bs)), lens)
    assert_attn_close(o[last], rall, aall, dtype, what=f"config 3 {dt}: last rows (extend kernel) vs oracle")
    assert_attn_close(od, rall, aall, dtype, what=f"config 3 {dt}: last rows (decode kernel) vs oracle")
    # (c) chunked prefill (the server splits at 8192 new tokens, server/args.py:33-34): a prompt that is
    # cut in two - first part with no prefix, second part behind it as a cached prefix - gives the
    # unchunked rows again (another summation order: equal within the model, not bit for bit)
    for b, cut in ((5, 2048), (5, 1000), (0, 100), (17, lens[17] - 1)):
        L = lens[b]
        one = torch.tensor([b], device=DEV)
        o2 = torch.full_like(q[starts[b]:starts[b] + L], float("nan"))
        z = torch.zeros(1, dtype=torch.int32, device=DEV)
        nat.extend_attention(o2[:cut], q[starts[b]:starts[b] + cut], kb, vb, r2t, one, torch.tensor([cut], device=DEV),
                             torch.tensor([cut], dtype=torch.int32, device=DEV), z, SCALE, 0.0, True, cut, cut, ws)
        nat.extend_attention(o2[cut:], q[starts[b] + cut:starts[b] + L], kb, vb, r2t, one, torch.tensor([L], device=DEV),
                             torch.tensor([L - cut], dtype=torch.int32, device=DEV), z, SCALE, 0.0, True, L - cut, L, ws)
        whole = o[starts[b]:starts[b] + L]
        u = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}[dtype]
        diff = float((o2.float() - whole.float()).abs().max())
        assert diff <= 2.5 * u * float(vb.float().abs().max()), f"chunked at {cut}: {diff:.3e}"
        same = float((o2 == whole).float().mean())
        print(f"[parity] config 3 {dt}: request {b} cut at {cut}: {100 * same:.1f} % of the outputs bit-identical, "
              f"max |diff| {diff:.2e}")
        # rows of the first part see exactly the same keys in the same tiles: bit for bit
        assert torch.equal(o2[:cut // 64 * 64], whole[:cut // 64 * 64])
