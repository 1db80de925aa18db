"""Sampler kernels (csrc/sampling.hip) against oracle/sampling.py and the reference-generated
fixture tests/golden/sampling.npz.  Token ids and keep counts are integer results: bit-exact.
Probabilities: softmax within 1e-5 relative, renormalised values exact to fp32 rounding."""
import numpy as np
import pytest
import torch

from oracle import sampling as osamp
from tests import golden

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bs,vocab", [(1, 33), (7, 1000), (16, 128256)])
def test_argmax_first_maximum(dtype, bs, vocab):
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(vocab)
    x = (torch.randn(bs, vocab, generator=g) * 4).to(dtype)
    x[0, vocab // 2] = x[0, 3] = x.float().max() + 1      # exact tie: the lower id wins
    if bs > 1:
        x[1] = 0                                          # a whole row of ties
    got = _native.argmax(x.cuda()).cpu()
    want = torch.argmax(x.float(), dim=-1)
    assert got.dtype == torch.int64 and torch.equal(got, want)
    assert got[0] == 3 and (bs == 1 or got[1] == 0)
    # strided rows (a view into a wider logits buffer)
    wide = torch.zeros(bs, vocab + 8, dtype=dtype)
    wide[:, :vocab] = x
    assert torch.equal(_native.argmax(wide.cuda()[:, :vocab]).cpu(), want)


def test_softmax_temperature_matches_oracle():
    from scratchpad_amd import _native
    g = golden.load("sampling")
    logits, temps = torch.from_numpy(g["logits"]), torch.from_numpy(g["temperatures"])
    got = _native.softmax_temperature_(logits.clone().cuda(), temps.cuda()).cpu()
    want = osamp.softmax_temperature(logits, temps)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-12)
    assert torch.allclose(got.sum(-1), torch.ones(got.shape[0]), atol=1e-5)
    big = torch.randn(5, 128256, generator=torch.Generator().manual_seed(1)) * 6
    got = _native.softmax_temperature_(big.clone().cuda(), None).cpu()
    assert torch.allclose(got, torch.softmax(big, -1), rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("use_minp", [True, False])
def test_fixture_rows_sample_and_keep_count_exact(use_minp):
    from scratchpad_amd import _native
    g = golden.load("sampling")
    probs = g["probs"].astype(np.float32)
    bs = probs.shape[0]
    ks, ps, ms = g["top_ks"], g["top_ps"].astype(np.float32), g["min_ps"].astype(np.float32)
    dp = _dev(probs)
    for u_row in (g["uniform"].astype(np.float32), np.full(bs, 0.0, np.float32),
                  np.full(bs, np.nextafter(np.float32(1), np.float32(0)), np.float32),
                  np.linspace(0.01, 0.99, bs).astype(np.float32)):
        ids, cnt = _native.top_k_top_p_min_p_sample(dp, _dev(ks), _dev(ps), _dev(ms) if use_minp else None,
                                                    _dev(u_row), return_keep_count=True)
        ids, cnt = ids.cpu().numpy(), cnt.cpu().numpy()
        for b in range(bs):
            mp = float(ms[b]) if use_minp else 0.0
            keep, _ = osamp.select(probs[b], int(ks[b]), float(ps[b]), mp)
            assert cnt[b] == keep.sum(), (b, cnt[b], keep.sum())
            assert ids[b] == osamp.sample(probs[b], int(ks[b]), float(ps[b]), mp, float(u_row[b])), b
            assert keep[ids[b]]
    # the reference's own keep sets (sort + fp32 cumsum), away from its rounding boundary
    ref_cnt = g["minp_keep_count" if use_minp else "nominp_keep_count"]
    assert np.abs(cnt - ref_cnt).max() <= 1


def test_fixture_rows_renorm_exact_and_close_to_reference():
    from scratchpad_amd import _native
    from scratchpad_amd.sampler import top_k_renorm_prob, top_p_normalize_probs
    g = golden.load("sampling")
    probs = g["probs"].astype(np.float32)
    ks, ps, ms = g["top_ks"], g["top_ps"].astype(np.float32), g["min_ps"].astype(np.float32)
    got = _native.top_k_top_p_min_p_renorm(_dev(probs), _dev(ks), _dev(ps), _dev(ms)).cpu().numpy()
    for b in range(probs.shape[0]):
        want = osamp.renorm(probs[b], int(ks[b]), float(ps[b]), float(ms[b]))
        assert np.array_equal(got[b] > 0, want > 0), b
        assert np.allclose(got[b], want, rtol=2e-7, atol=0), b
    # top_p_normalize_probs_torch of the reference, rows without boundary flips
    got = top_p_normalize_probs(_dev(probs), _dev(ps)).cpu().numpy()
    ref = g["top_p_normalized"]
    same = [(got[b] > 0).sum() == (ref[b] > 0).sum() for b in range(20)]
    assert sum(same) >= 18
    for b in range(20):
        if same[b]:
            assert np.allclose(got[b], ref[b], rtol=1e-5, atol=1e-9), b
    # top-k only: exactly k survivors, unit mass
    got = top_k_renorm_prob(_dev(probs), _dev(np.minimum(ks, 1000))).cpu().numpy()
    assert np.array_equal((got > 0).sum(-1)[:20], np.minimum(ks, 1000)[:20])
    assert np.allclose(got.sum(-1), 1.0, atol=1e-5)


def test_full_vocab_rows_against_oracle_and_properties():
    """vocab 128256, bs 64: a sample of rows is checked token-exact against the oracle; all rows
    against size-independent properties (kept count bound, unit mass, draws land on kept tokens,
    bit-identical reruns)."""
    from scratchpad_amd import _native
    gen = torch.Generator().manual_seed(99)
    bs, vocab = 64, 128256
    scale = torch.tensor([0.5, 1, 2, 3, 5, 8, 12, 1.5] * 8).view(-1, 1)
    probs = torch.softmax(torch.randn(bs, vocab, generator=gen) * scale, dim=-1)
    ks = torch.tensor([1 << 30, 50, 1, 1000, 1 << 30, 7, 40000, 1 << 30] * 8, dtype=torch.int32)
    ps = torch.tensor([1.0, 0.9, 1.0, 0.95, 0.5, 0.99, 1.0, 0.2] * 8)
    ms = torch.tensor([0.0, 0.0, 0.0, 0.01, 0.05, 0.0, 0.001, 0.0] * 8)
    u = torch.rand(bs, generator=gen)
    dp = probs.cuda()
    ids, cnt = _native.top_k_top_p_min_p_sample(dp, ks.cuda(), ps.cuda(), ms.cuda(), u.cuda(), return_keep_count=True)
    ids2, cnt2 = _native.top_k_top_p_min_p_sample(dp, ks.cuda(), ps.cuda(), ms.cuda(), u.cuda(), return_keep_count=True)
    assert torch.equal(ids, ids2) and torch.equal(cnt, cnt2), "order-independent integer accumulation"
    ren = _native.top_k_top_p_min_p_renorm(dp, ks.cuda(), ps.cuda(), ms.cuda())
    ids, cnt, renc = ids.cpu(), cnt.cpu(), ren.cpu()
    assert bool((cnt >= 1).all()) and bool((cnt <= torch.clamp(ks, max=vocab)).all())
    assert torch.equal((renc > 0).sum(-1).to(torch.int32), torch.minimum(cnt, (probs > 0).sum(-1).to(torch.int32)))
    assert torch.allclose(renc.sum(-1), torch.ones(bs), atol=2e-5)
    assert bool((renc[torch.arange(bs), ids] > 0).all())
    assert torch.equal(ids[2::8], probs[2::8].argmax(-1)), "top_k = 1 is greedy"
    pn = probs.numpy()
    for b in (0, 1, 3, 4, 5, 6, 7, 12, 31, 63):
        keep, _ = osamp.select(pn[b], int(ks[b]), float(ps[b]), float(ms[b]))
        assert int(cnt[b]) == int(keep.sum()), b
        assert int(ids[b]) == osamp.sample(pn[b], int(ks[b]), float(ps[b]), float(ms[b]), float(u[b])), b


def test_degenerate_rows():
    from scratchpad_amd import _native
    probs = torch.zeros(3, 100)
    probs[1, 17] = 1.0                       # one-hot
    probs[2] = 0.01                          # uniform: every token ties
    ks = torch.tensor([5, 5, 5], dtype=torch.int32)
    ps = torch.tensor([0.9, 0.9, 0.035])
    u = torch.tensor([0.5, 0.99, 0.70])
    ids, cnt = _native.top_k_top_p_min_p_sample(probs.cuda(), ks.cuda(), ps.cuda(), None, u.cuda(), return_keep_count=True)
    assert ids.tolist()[0] == 0 and ids.tolist()[1] == 17
    # uniform row: exclusive mass 0, .01, .02, .03 <= .035 -> the 4 lowest ids survive; u = .7 -> the third
    assert cnt.tolist()[1:] == [1, 4] and ids.tolist()[2] == 2
    for b in (1, 2):
        assert ids.tolist()[b] == osamp.sample(probs[b].numpy(), 5, float(ps[b]), 0.0, float(u[b]))
    with pytest.raises(RuntimeError):
        _native.top_k_top_p_min_p_sample(probs, ks, ps, None, u)          # host tensors: no CPU fallback


def test_draw_frequencies_follow_the_filtered_distribution():
    """One row replicated 8192 times with a regular grid of uniforms: exact inverse CDF."""
    from scratchpad_amd import _native
    g = golden.load("sampling")
    N = 8192
    for b in (3, 8, 21):
        row = g["probs"][b].astype(np.float32)
        k, p, m = int(g["top_ks"][b]), float(g["top_ps"][b]), float(g["min_ps"][b])
        probs = _dev(np.tile(row, (N, 1)))
        u = _dev(((np.arange(N) + 0.5) / N).astype(np.float32))
        full = lambda v, dt: torch.full((N,), v, dtype=dt, device="cuda")
        ids = _native.top_k_top_p_min_p_sample(probs, full(k, torch.int32), full(p, torch.float32),
                                               full(m, torch.float32), u).cpu().numpy()
        want = osamp.renorm(row, k, p, m).astype(np.float64)
        hits = np.bincount(ids, minlength=row.shape[0])
        assert hits[want == 0].sum() == 0
        assert np.abs(hits - N * want).max() <= 1.0 + 1e-6


def test_sampler_module_greedy_and_stochastic_paths():
    from scratchpad_amd.llama import LogitsProcessorOutput
    from scratchpad_amd.sampler import Sampler, SamplingBatchInfo, SamplingParams
    gen = torch.Generator().manual_seed(5)
    bs, vocab = 6, 4096
    logits = torch.randn(bs, vocab, generator=gen) * 3
    sampler = Sampler()
    greedy = SamplingBatchInfo.from_params([SamplingParams(temperature=0.0)] * bs, vocab, "cuda")
    assert greedy.is_all_greedy and greedy.top_ks.tolist() == [1] * bs
    out = LogitsProcessorOutput(next_token_logits=logits.clone().cuda())
    ids = sampler(out, greedy, return_logprob=True, top_logprobs_nums=[2] * bs)
    assert torch.equal(ids.cpu(), logits.argmax(-1))
    lp = torch.log_softmax(logits, -1)
    assert torch.allclose(out.next_token_logprobs.cpu(), lp[torch.arange(bs), ids.cpu()], atol=1e-5)
    assert [i[0] for i in out.next_token_top_logprobs_idx] == ids.tolist()

    params = [SamplingParams(temperature=0.8, top_p=0.9, top_k=50), SamplingParams(temperature=1.2),
              SamplingParams(temperature=0.0), SamplingParams(temperature=1.0, min_p=0.1),
              SamplingParams(temperature=0.5, top_k=5, top_p=0.5), SamplingParams(temperature=2.0, top_p=0.3)]
    info = SamplingBatchInfo.from_params(params, vocab, "cuda")
    assert not info.is_all_greedy and info.need_min_p_sampling
    assert info.top_ks.tolist() == [50, 1 << 30, 1, 1 << 30, 5, 1 << 30]
    u = torch.rand(bs, generator=gen)
    out = LogitsProcessorOutput(next_token_logits=logits.clone().cuda())
    ids = sampler(out, info, uniform=u.cuda()).cpu()
    probs = out.next_token_logits.cpu().numpy()        # softmax was applied in place (sampler.py:72)
    assert np.allclose(probs, osamp.softmax_temperature(logits, info.temperatures.cpu()).numpy(), rtol=1e-5, atol=1e-12)
    for b in range(bs):
        assert int(ids[b]) == osamp.sample(probs[b], int(info.top_ks[b]), float(info.top_ps[b]),
                                           float(info.min_ps[b]), float(u[b])), b
    assert int(ids[2]) == int(logits[2].argmax())      # the greedy request inside a sampling batch
    with pytest.raises(ValueError):
        SamplingParams(top_p=0.0).verify()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_argmax_vector_path_ties_tails_and_vocab_shards(dtype):
    """sp_argmax on 16-bit logits (16-byte loads, unaligned rows, ragged tails) returns torch.argmax's first
    maximal index, and the vocab-parallel form (sp_argmax_shard per shard -> [bs, 2] words -> sp_argmax_merge)
    returns the same ids as the argmax of the concatenated row - including ties across shards and shards
    whose real-vocabulary columns are fewer than their width (padding columns must never win)."""
    from scratchpad_amd import _native
    g = torch.Generator().manual_seed(17)
    bs, vocab = 9, 128256
    logits = (torch.randn(bs, vocab, generator=g) * 3).to(dtype)
    logits[1, 5] = logits[1, 77777] = 50.0            # tie: the first index wins
    logits[2, vocab - 1] = 60.0                       # the maximum sits in the ragged tail
    logits[3, 0] = 60.0
    logits[4] = 1.0                                   # a constant row: index 0
    dev = logits.cuda()
    want = logits.float().argmax(-1)
    assert torch.equal(_native.argmax(dev).cpu(), want)
    odd = dev[:, 3:vocab - 5]                         # rows that start off a 16-byte boundary, odd length
    assert torch.equal(_native.argmax(odd).cpu(), logits[:, 3:vocab - 5].float().argmax(-1))
    # 8 shards of a vocabulary padded to 8 x 16064 (the last shard holds 15808 real columns + padding)
    tp, width = 8, 16064
    padded = torch.full((bs, tp * width), 99.0, dtype=dtype)      # padding columns hold a LARGER value
    padded[:, :vocab] = logits
    pairs = []
    for r in range(tp):
        shard = padded[:, r * width:(r + 1) * width].contiguous().cuda()
        cols = max(0, min(width, vocab - r * width))
        pairs.append(_native.argmax_shard(shard, cols, r * width))
    ids = _native.argmax_merge(torch.stack(pairs).contiguous())
    assert torch.equal(ids.cpu(), want)
    empty = _native.argmax_shard(padded[:, :8].contiguous().cuda(), 0, 12345)     # a shard of padding only
    assert torch.equal(_native.argmax_merge(torch.stack([pairs[0], empty]).contiguous()).cpu(),
                       logits[:, :width].float().argmax(-1))
