"""Pin the CPU oracle (oracle/) to vectors recorded from the reference's own code.

Tolerances: fp32 everywhere, so <= 2e-6 absolute on O(1) values for float ops (different
summation order than the Triton-interpreter kernels) and bit-exact for integer/index work."""
import numpy as np
import pytest
import torch

from oracle import llama as ollama
from oracle import ops
from tests import golden


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol=2e-6, rtol=2e-6):
    a, b = T(np.asarray(a)).double(), T(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    assert bool((err <= atol + rtol * b.abs()).all()), f"max err {err.max().item():.3e}"


def test_rmsnorm():
    g = golden.load("rmsnorm")
    for i in range(int(g["num_cases"])):
        x, w, r = T(g[f"c{i}_x"]), T(g[f"c{i}_w"]), T(g[f"c{i}_res"])
        eps = float(g[f"c{i}_eps"])
        close(ops.rmsnorm(x, w, eps), g[f"c{i}_y"])
        y, r2 = ops.rmsnorm(x, w, eps, r)
        close(y, g[f"c{i}_y_fused"])
        assert np.array_equal(r2.numpy(), g[f"c{i}_res_out"])  # fp32 add: exact


def test_silu_mul():
    g = golden.load("silu_mul")
    for i in range(int(g["num_cases"])):
        close(ops.silu_and_mul(T(g[f"c{i}_x"])), g[f"c{i}_y"])


def test_rotary_cache_and_apply():
    g = golden.load("rotary")
    for i in range(int(g["num_cases"])):
        hs, rd, mp = int(g[f"c{i}_head_size"]), int(g[f"c{i}_rotary_dim"]), int(g[f"c{i}_max_pos"])
        sc = tuple(g[f"c{i}_scaling"]) if f"c{i}_scaling" in g else None
        other = golden.rope_scaling(g, i) if f"c{i}_scaling_json" in g else None      # linear / dynamic / yarn
        cache = ops.rope_cos_sin_cache(mp, float(g[f"c{i}_base"]), rd, sc, scaling=other)
        # cache built with the same torch ops on the same CPU: bit-exact
        assert np.array_equal(cache.numpy(), g[f"c{i}_cos_sin_cache"]), f"case {i}"
        q, k = ops.rotary_embedding(T(g[f"c{i}_positions"]), T(g[f"c{i}_q"]), T(g[f"c{i}_k"]), hs,
                                    cache, bool(g[f"c{i}_neox"]))
        close(q, g[f"c{i}_q_out"], atol=1e-6)
        close(k, g[f"c{i}_k_out"], atol=1e-6)


def test_kv_store_bit_exact():
    g = golden.load("kv_pool")
    L, size, H, D = (int(g[k]) for k in ("layer_num", "size", "head_num", "head_dim"))
    for layer in range(L):
        kb = torch.zeros(size + 1, H, D)
        vb = torch.zeros(size + 1, H, D)
        ops.kv_store(kb, vb, T(g[f"l{layer}_loc"]), T(g[f"l{layer}_k"]), T(g[f"l{layer}_v"]))
        assert np.array_equal(kb.numpy(), g[f"l{layer}_k_buffer"])
        assert np.array_equal(vb.numpy(), g[f"l{layer}_v_buffer"])


def test_positions_and_req_to_token_bit_exact():
    g = golden.load("positions")
    pos, start = ops.compute_position(T(g["prefix_lens"]), T(g["extend_lens"]))
    assert pos.dtype == torch.int64 and np.array_equal(pos.numpy(), g["positions"])
    assert start.dtype == torch.int32 and np.array_equal(start.numpy(), g["extend_start_loc"])
    assert np.array_equal(ops.clamp_position(T(g["decode_seq_lens"])).numpy(), g["decode_positions"])
    table = T(g["w_table_in"]).clone()
    ops.write_req_to_token(table, T(g["w_req_pool_indices"]), T(g["w_pre_lens"]), T(g["w_seq_lens"]),
                           T(g["w_extend_lens"]), T(g["w_out_cache_loc"]))
    assert np.array_equal(table.numpy(), g["w_table_out"])


def test_decode_attention():
    g = golden.load("decode_attention")
    for i in range(int(g["num_cases"])):
        o = ops.decode_attention(T(g[f"c{i}_q"]), T(g[f"c{i}_k_buffer"]), T(g[f"c{i}_v_buffer"]),
                                 T(g[f"c{i}_req_to_token"]), T(g[f"c{i}_req_pool_indices"]),
                                 T(g[f"c{i}_seq_lens"]), float(g[f"c{i}_sm_scale"]),
                                 float(g[f"c{i}_logit_cap"]))
        close(o, g[f"c{i}_o"], atol=3e-6)


@pytest.mark.parametrize("use_contiguous_kv", [False, True])
def test_extend_attention(use_contiguous_kv):
    g = golden.load("extend_attention")
    for i in range(int(g["num_cases"])):
        kb, vb = T(g[f"c{i}_k_buffer"]), T(g[f"c{i}_v_buffer"])
        loc = T(g[f"c{i}_out_cache_loc"])
        kw = dict(k_extend=kb[loc], v_extend=vb[loc]) if use_contiguous_kv else {}
        o = ops.extend_attention(T(g[f"c{i}_q"]), kb, vb, T(g[f"c{i}_req_to_token"]),
                                 T(g[f"c{i}_req_pool_indices"]), T(g[f"c{i}_seq_lens"]),
                                 T(g[f"c{i}_extend_seq_lens"]), T(g[f"c{i}_extend_start_loc"]),
                                 float(g[f"c{i}_sm_scale"]), float(g[f"c{i}_logit_cap"]), **kw)
        close(o, g[f"c{i}_o"], atol=3e-6)


def test_context_attention():
    """the cache-less causal varlen path against the reference's context_attention_fwd (Triton interpreter run)"""
    g = golden.load("prefill_attention")
    for i in range(int(g["num_cases"])):
        o = ops.context_attention(T(g[f"c{i}_q"]), T(g[f"c{i}_k"]), T(g[f"c{i}_v"]), T(g[f"c{i}_b_start_loc"]),
                                  T(g[f"c{i}_b_seq_len"]))
        close(o, g[f"c{i}_o"], atol=3e-6)
        # the same numbers from the extend oracle with no cached prefix and an identity req_to_token table: the
        # form scratchpad_amd.vision.varlen_attention(causal=True) hands to sp_extend_attention
        lens, start = T(g[f"c{i}_b_seq_len"]), T(g[f"c{i}_b_start_loc"])
        table = torch.arange(int(lens.max())).view(1, -1) + start.view(-1, 1).long()
        table = table.clamp_(max=int(lens.sum()) - 1).to(torch.int32)
        D = g[f"c{i}_q"].shape[-1]
        o2 = ops.extend_attention(T(g[f"c{i}_q"]), T(g[f"c{i}_k"]), T(g[f"c{i}_v"]), table,
                                  torch.arange(lens.shape[0]), lens.long(), lens, start, D ** -0.5)
        close(o2, g[f"c{i}_o"], atol=3e-6)


def test_merge_state_matches_joint_softmax():
    # flashinfer merge_state is third-party (absent): pinned by the identity
    # attention(K1 u K2) == merge(attention(K1), attention(K2)), which the reference relies on
    # (flashinfer_backend.py:419-439).
    g = torch.Generator().manual_seed(5)
    q = torch.randn(3, 4, 16, generator=g)
    k = torch.randn(3, 40, 4, 16, generator=g)
    v = torch.randn(3, 40, 4, 16, generator=g)

    def part(ks, vs):
        s = torch.einsum("thd,tlhd->thl", q, ks) * 0.25
        return torch.einsum("thl,tlhd->thd", torch.softmax(s, -1), vs), torch.logsumexp(s, -1)

    o1, l1 = part(k[:, :13], v[:, :13])
    o2, l2 = part(k[:, 13:], v[:, 13:])
    o, l = ops.merge_state(o1, l1, o2, l2)
    of, lf = part(k, v)
    close(o, of, atol=1e-6)
    close(l, lf, atol=1e-6)


def tiny_llama_case(name):
    g = golden.load("tiny_llama")
    pfx = name + "_"
    hidden, inter, nl, Hq, Hkv, vocab, tie = (int(x) for x in g[pfx + "cfg"])
    sc = tuple(g[pfx + "rope_scaling"]) if pfx + "rope_scaling" in g else None
    shape = ollama.LlamaShape(hidden, inter, nl, Hq, Hkv, vocab, bool(tie), float(g[pfx + "rope_theta"]),
                              sc, int(g[pfx + "max_pos"]), float(g[pfx + "rms_eps"]))
    w = {k[len(pfx) + 3:]: T(v) for k, v in g.items() if k.startswith(pfx + "w::")}
    return g, pfx, shape, w


@pytest.mark.parametrize("name", ["a", "b"])
def test_tiny_llama_logits(name):
    g, pfx, shape, w = tiny_llama_case(name)
    kv = ollama.OracleKV(shape, 96, 4, 64)
    ext = T(g[pfx + "extend_lens"])
    req = T(g[pfx + "req_pool_indices"])
    loc = T(g[pfx + "out_cache_loc"])
    pre = torch.zeros_like(ext)
    ops.write_req_to_token(kv.req_to_token, req, pre, ext, ext, loc)
    positions, start = ops.compute_position(pre, ext)
    assert np.array_equal(positions.numpy(), g[pfx + "positions"])
    logits = ollama.forward(shape, w, kv, mode="extend", input_ids=T(g[pfx + "input_ids"]),
                            positions=positions, req_pool_indices=req, seq_lens=ext.long(),
                            out_cache_loc=loc, extend_seq_lens=ext, extend_start_loc=start)
    close(logits, g[pfx + "prefill_logits"], atol=2e-5, rtol=1e-5)
    nxt = logits.argmax(-1)
    assert np.array_equal(nxt.numpy(), g[pfx + "next_ids"])
    seq2 = ext.long() + 1
    dloc = T(g[pfx + "decode_out_cache_loc"])
    ops.write_req_to_token(kv.req_to_token, req, ext.long(), seq2, torch.ones_like(seq2), dloc)
    logits2 = ollama.forward(shape, w, kv, mode="decode", input_ids=nxt, positions=ops.clamp_position(seq2),
                             req_pool_indices=req, seq_lens=seq2, out_cache_loc=dloc)
    close(logits2, g[pfx + "decode_logits"], atol=2e-5, rtol=1e-5)
    close(kv.k[0], g[pfx + "k_buffer0_after"], atol=1e-5)
    close(kv.v[1], g[pfx + "v_buffer1_after"], atol=1e-5)


def test_tiny_mllama_text_model():
    """cross-attention layers, per-head q/k RMSNorm, tanh gates, row mask, encoder slots first"""
    from oracle import mllama as omllama
    g = golden.load("tiny_mllama")
    hidden, inter, nl, Hq, Hkv, vocab = (int(x) for x in g["cfg"])
    shape = ollama.LlamaShape(hidden, inter, nl, Hq, Hkv, vocab, False, 500000.0, None, 128, 1e-5)
    w = {k[3:]: T(v) for k, v in g.items() if k.startswith("w::")}
    close(ops.rmsnorm(T(g["qnorm_x"]), T(g["qnorm_w"]), 1e-5), g["qnorm_y"], atol=1e-6)
    kv = ollama.OracleKV(shape, 96, 4, 64)
    text, enc, req = T(g["text_lens"]), T(g["encoder_lens"]), T(g["req_pool_indices"])
    kv.req_to_token.copy_(T(g["req_to_token_extend_rows"]))     # decode slot entries are harmless extras
    start = torch.zeros_like(text)
    start[1:] = torch.cumsum(text[:-1], 0)
    common = dict(req_pool_indices=req, encoder_lens=enc)
    l1 = omllama.forward(shape, [1], w, kv, mode="extend", input_ids=T(g["input_ids"]), positions=T(g["positions"]),
                         seq_lens=text.long(), out_cache_loc=T(g["out_cache_loc"]), row_mask=T(g["row_mask_extend"]),
                         extend_seq_lens=text, extend_start_loc=start,
                         cross_attention_states=T(g["cross_attention_states"]),
                         encoder_out_cache_loc=T(g["encoder_out_cache_loc"]), **common)
    close(l1, g["prefill_logits"], atol=2e-5, rtol=1e-5)
    assert np.array_equal(l1.argmax(-1).numpy(), g["next_ids"])
    seq2 = text.long() + 1
    l2 = omllama.forward(shape, [1], w, kv, mode="decode", input_ids=T(g["next_ids"]),
                         positions=ops.clamp_position(seq2), seq_lens=seq2, out_cache_loc=T(g["decode_out_cache_loc"]),
                         row_mask=T(g["row_mask_decode"]), **common)
    close(l2, g["decode_logits"], atol=2e-5, rtol=1e-5)
    close(kv.k[1], g["k_buffer1_after"], atol=1e-5)


# ---------------------------------------------------------------- sampler (oracle/sampling.py)
def test_sampling_reference_layer_matches_reference_run():
    from oracle import sampling as osamp
    g = golden.load("sampling")
    probs = torch.from_numpy(g["probs"])
    top_ks, top_ps, min_ps = (torch.from_numpy(g[k]) for k in ("top_ks", "top_ps", "min_ps"))
    sm = osamp.softmax_temperature(torch.from_numpy(g["logits"]), torch.from_numpy(g["temperatures"]))
    assert torch.equal(sm[:20], probs[:20])          # rows 20.. were overwritten with tied grids
    for tag, mp in (("minp", min_ps), ("nominp", None)):
        w, idx = osamp.filter_sorted_reference(probs.clone(), top_ks, top_ps, mp)
        assert torch.equal(w, torch.from_numpy(g[f"{tag}_sorted_weights"]))
        assert torch.equal((w > 0).sum(-1), torch.from_numpy(g[f"{tag}_keep_count"]))
    assert torch.equal(osamp.top_p_normalize_reference(probs.clone(), top_ps),
                       torch.from_numpy(g["top_p_normalized"]))


def test_sampling_integer_definition_agrees_with_reference_filter():
    """Layer 2 (sort-free, integer mass) keeps the same tokens as the reference's sort + fp32 cumsum;
    a difference is only tolerated for tokens whose exclusive cumulative mass is within fp32
    cumsum noise of top_p, and (tied rows) in WHICH of several equal tokens is kept."""
    from oracle import sampling as osamp
    g = golden.load("sampling")
    probs = g["probs"].astype(np.float32)
    for tag, use_minp in (("minp", True), ("nominp", False)):
        ref_keep = g[f"{tag}_keep"].astype(bool)
        for b in range(probs.shape[0]):
            keep, total = osamp.select(probs[b], int(g["top_ks"][b]), float(g["top_ps"][b]),
                                       float(g["min_ps"][b]) if use_minp else 0.0)
            ref = ref_keep[b]
            if b < 20:
                if not np.array_equal(keep, ref):
                    diff = np.flatnonzero(keep != ref)
                    order = np.argsort(-probs[b].astype(np.float64), kind="stable")
                    excl = np.concatenate([[0.0], np.cumsum(probs[b][order].astype(np.float64))[:-1]])
                    pos = {int(t): i for i, t in enumerate(order)}
                    assert all(abs(excl[pos[int(t)]] - float(g["top_ps"][b])) < 4e-6 for t in diff), (tag, b)
            else:       # tied rows: same number kept, same multiset of kept probabilities
                assert keep.sum() == ref.sum(), (tag, b)
                assert np.array_equal(np.sort(probs[b][keep]), np.sort(probs[b][ref])), (tag, b)
            assert total == sum(int(x) for x in osamp.fx(probs[b][keep]))


def test_sampling_inverse_cdf_covers_the_kept_distribution():
    """u on a regular grid of N points must hit token i round(N * p_i) times (+-1): the draw is an
    exact inverse CDF of the renormalised kept distribution, and never returns a dropped token."""
    from oracle import sampling as osamp
    g = golden.load("sampling")
    probs = g["probs"].astype(np.float32)
    N = 2000
    for b in (2, 3, 8, 13, 21):
        k, p, m = int(g["top_ks"][b]), float(g["top_ps"][b]), float(g["min_ps"][b])
        want = osamp.renorm(probs[b], k, p, m)
        hits = np.zeros(probs.shape[1], dtype=np.int64)
        for i in range(N):
            hits[osamp.sample(probs[b], k, p, m, (i + 0.5) / N)] += 1
        assert hits[want == 0].sum() == 0
        assert np.abs(hits - N * want.astype(np.float64)).max() <= 1.0 + 1e-6, b
        assert abs(float(want.sum()) - 1.0) < 1e-5


# ---------------------------------------------------------------- Mllama vision tower
@pytest.mark.parametrize("tag", ["full", "ragged"])
def test_mllama_vision_oracle_matches_reference_run(tag):
    from oracle import mllama_vision as ov
    g = golden.load("mllama_vision")
    sh = ov.VisionShape.from_fixture(g)
    w = {k[2:]: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in g.items() if k.startswith("w.")}
    out = ov.forward(sh, w, torch.from_numpy(g[f"{tag}_pixel_values"]),
                     torch.from_numpy(g[f"{tag}_aspect_ratio_ids"]), torch.from_numpy(g[f"{tag}_aspect_ratio_mask"]))
    want = torch.from_numpy(g[f"{tag}_out"])
    assert out.shape == want.shape
    assert float((out - want).abs().max()) <= 2e-5 * float(want.abs().max())
