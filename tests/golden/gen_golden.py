#!/usr/bin/env python3
"""Regenerate ``tests/golden/*.npz`` by RUNNING the upstream reference on CPU.

BUILD-CONTAINER ONLY (needs ``/root/reference``).  Each fixture stores seeded inputs
and the outputs produced by the reference's own code:

  * ``forward_native`` of RMSNorm / SiluAndMul / RotaryEmbedding / Llama3RotaryEmbedding
    (scratchpad/nn/layers/{layernorm,activation,rotary_embedding}.py),
  * ``MHATokenToKVPool.set_kv_buffer``, ``ReqToTokenPool``, ``TokenToKVPoolAllocator``
    (scratchpad/memory/pool.py),
  * ``compute_position_triton/torch`` (scratchpad/model_executor/forward_info.py),
  * ``write_req_to_token_pool_triton`` (scratchpad/scheduler/schedule_batch.py),
  * the in-tree Triton kernels ``decode_attention_fwd`` / ``extend_attention_fwd`` / ``context_attention_fwd``
    (scratchpad/nn/attention/triton_attn/) executed by the Triton interpreter,
  * the full reference ``LlamaForCausalLM`` (2 layers, tiny) for prefill + decode logits.

The fixtures are data only (inputs and expected outputs); nothing of the reference's
source text is stored.  Usage:  python tests/golden/gen_golden.py [name ...]
"""
import math
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_loader  # noqa: E402

_ref_loader.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def _grid(t, step=64.0, lim=255.0):
    """Round to multiples of 1/64 with |k| <= 255: exactly representable in bf16, fp16 and fp32,
    so the SAME fixture inputs can be fed to the HIP kernels in any dtype without input rounding."""
    return torch.clamp(torch.round(t * step), -lim, lim) / step


def _randn(*shape, generator, scale=1.0):
    return _grid(torch.randn(*shape, generator=generator) * scale)


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    packed = {}
    for k, v in arrays.items():
        a = _np(v)
        # grid inputs are exact in fp16: store them as fp16 (tests/golden/__init__.py widens them back)
        if a.dtype == np.float32 and a.size > 64 and np.array_equal(a.astype(np.float16).astype(np.float32), a):
            a = a.astype(np.float16)
        packed[k] = a
    np.savez_compressed(path, **packed)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def gen_rmsnorm():
    from scratchpad.nn.layers.layernorm import RMSNorm

    g = torch.Generator().manual_seed(101)
    out = {}
    for i, (T, H, eps) in enumerate([(5, 64, 1e-5), (3, 256, 1e-6), (1, 4096, 1e-5), (7, 96, 1e-5)]):
        m = RMSNorm(H, eps)
        m.weight.data = _grid(torch.randn(H, generator=g) * 0.5 + 1.0)
        x = _randn(T, H, generator=g, scale=1.5)
        r = _randn(T, H, generator=g)
        y = m.forward_native(x.clone())
        y2, r2 = m.forward_native(x.clone(), r.clone())
        out.update({f"c{i}_x": x, f"c{i}_w": m.weight.data, f"c{i}_eps": np.float64(eps),
                    f"c{i}_res": r, f"c{i}_y": y, f"c{i}_y_fused": y2, f"c{i}_res_out": r2})
    out["num_cases"] = np.int64(4)
    _save("rmsnorm", **out)


def gen_silu_mul():
    from scratchpad.nn.layers.activation import SiluAndMul

    g = torch.Generator().manual_seed(102)
    m = SiluAndMul()
    out = {}
    for i, (T, d) in enumerate([(4, 32), (3, 176), (1, 1024)]):
        x = _randn(T, 2 * d, generator=g, scale=2.0)
        out[f"c{i}_x"] = x
        out[f"c{i}_y"] = m.forward_native(x)
    out["num_cases"] = np.int64(3)
    _save("silu_mul", **out)


def gen_rotary():
    from scratchpad.nn.layers.rotary_embedding import get_rope

    g = torch.Generator().manual_seed(103)
    out = {}
    llama3 = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
              "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    cases = [
        # head, rot_dim, max_pos, base, neox, scaling, Hq, Hkv, T
        (64, 64, 256, 10000, True, None, 4, 2, 9),
        (128, 128, 192, 500000, True, None, 8, 2, 6),
        (64, 64, 384, 500000, True, llama3, 8, 2, 7),
        (128, 128, 320, 500000, True, dict(llama3, factor=8.0), 4, 1, 5),
        (64, 64, 128, 10000, False, None, 4, 4, 5),      # GPT-J interleaved style
        (64, 32, 128, 10000, True, None, 4, 2, 5),       # partial rotary
        # context-extension variants (rotary_embedding.py:173-414): the table covers max_pos * factor positions
        (64, 64, 64, 10000, True, {"rope_type": "linear", "factor": 4.0}, 4, 2, 6),
        (128, 128, 48, 500000, True, {"rope_type": "dynamic", "factor": 2.0}, 4, 1, 5),
        (64, 64, 96, 10000, True, {"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 32}, 4, 2, 6),
        (128, 128, 256, 1000000, True, {"rope_type": "yarn", "factor": 8.0, "original_max_position_embeddings": 24,
                                        "beta_fast": 16, "beta_slow": 2, "attn_factor": 1.25}, 8, 2, 5),
        (64, 32, 64, 10000, False, {"rope_type": "yarn", "factor": 2.0, "original_max_position_embeddings": 40,
                                    "extrapolation_factor": 0.5}, 4, 4, 5),
    ]
    for i, (hs, rd, mp, base, neox, sc, Hq, Hkv, T) in enumerate(cases):
        rope = get_rope(hs, rd, mp, base, neox, sc, dtype=torch.float32)
        span = rope.cos_sin_cache.shape[0]                   # = mp, or the extended length of a scaled variant
        pos = torch.randint(0, span, (T,), generator=g, dtype=torch.int64)
        pos[0] = 0
        pos[-1] = span - 1
        q = _randn(T, Hq * hs, generator=g)
        k = _randn(T, Hkv * hs, generator=g)
        q2, k2 = rope.forward_native(pos, q.clone(), k.clone())
        out.update({
            f"c{i}_head_size": np.int64(hs), f"c{i}_rotary_dim": np.int64(rd),
            f"c{i}_max_pos": np.int64(mp), f"c{i}_base": np.float64(base),
            f"c{i}_neox": np.int64(neox), f"c{i}_positions": pos, f"c{i}_q": q, f"c{i}_k": k,
            f"c{i}_q_out": q2, f"c{i}_k_out": k2,
            # the whole cache pins _compute_inv_freq/_compute_cos_sin_cache (incl. llama3)
            f"c{i}_cos_sin_cache": rope.cos_sin_cache,
        })
        if sc is not None and sc["rope_type"] == "llama3":
            out[f"c{i}_scaling"] = np.array([sc["factor"], sc["low_freq_factor"],
                                             sc["high_freq_factor"],
                                             sc["original_max_position_embeddings"]], np.float64)
        elif sc is not None:                                 # the other variants: the dict itself, as JSON text
            import json
            out[f"c{i}_scaling_json"] = np.array(json.dumps(sc))
    out["num_cases"] = np.int64(len(cases))
    _save("rotary", **out)


def gen_kv_pool():
    from types import SimpleNamespace
    from scratchpad.memory.pool import (MHATokenToKVPool, ReqToTokenPool,
                                        TokenToKVPoolAllocator)

    g = torch.Generator().manual_seed(104)
    size, Hkv, D, L = 40, 2, 16, 3
    pool = MHATokenToKVPool(size, 1, torch.float32, Hkv, D, L, "cpu", False)
    out = {"size": np.int64(size), "head_num": np.int64(Hkv), "head_dim": np.int64(D),
           "layer_num": np.int64(L)}
    for layer in range(L):
        T = 6 + layer
        loc = torch.randperm(size, generator=g)[:T] + 1
        if layer == 2:
            loc[0] = 0  # padded rows write the reserved dummy slot 0
        k = _randn(T, Hkv, D, generator=g)
        v = _randn(T, Hkv, D, generator=g)
        pool.set_kv_buffer(SimpleNamespace(layer_id=layer), loc, k, v)
        out.update({f"l{layer}_loc": loc, f"l{layer}_k": k, f"l{layer}_v": v})
    for layer in range(L):
        out[f"l{layer}_k_buffer"] = pool.get_key_buffer(layer)
        out[f"l{layer}_v_buffer"] = pool.get_value_buffer(layer)

    # allocator trace: alloc / free / free_group / backup-restore, recorded as int arrays
    alloc = TokenToKVPoolAllocator(16, torch.float32, "cpu", pool)
    a0 = alloc.alloc(5)
    a1 = alloc.alloc(4)
    avail0 = alloc.available_size()
    alloc.free(a0[1:3])
    a2 = alloc.alloc(8)
    too_many = alloc.alloc(100)
    state = alloc.backup_state()
    a3 = alloc.alloc(1)
    alloc.restore_state(state)
    a4 = alloc.alloc(1)
    alloc.free_group_begin()
    alloc.free(a1[:2])
    alloc.free(a2[:3])
    avail_in_group = alloc.available_size()
    alloc.free_group_end()
    out.update({"alloc_a0": a0, "alloc_a1": a1, "alloc_avail0": np.int64(avail0),
                "alloc_a2": a2, "alloc_too_many_is_none": np.int64(too_many is None),
                "alloc_a3": a3, "alloc_a4": a4,
                "alloc_avail_in_group": np.int64(avail_in_group),
                "alloc_free_final": alloc.free_slots})
    alloc.clear()
    out["alloc_free_after_clear"] = alloc.free_slots

    r2t = ReqToTokenPool(5, 12, "cpu", False)
    r0 = r2t.alloc(2)
    r1 = r2t.alloc(2)
    r2t.free(r0[0])
    r2 = r2t.alloc(2)
    none = r2t.alloc(3)
    r2t.write((torch.tensor([1, 3]), torch.tensor([4, 7])), torch.tensor([11, 13], dtype=torch.int32))
    r2t.write((2, slice(0, 3)), torch.tensor([5, 6, 7], dtype=torch.int32))
    out.update({"r2t_r0": r0, "r2t_r1": r1, "r2t_r2": r2, "r2t_none": np.int64(none is None),
                "r2t_table": r2t.req_to_token, "r2t_avail": np.int64(r2t.available_size())})
    _save("kv_pool", **out)


def gen_positions():
    from scratchpad.model_executor.forward_info import (compute_position_torch,
                                                        compute_position_triton)
    from scratchpad.scheduler.schedule_batch import write_req_to_token_pool_triton

    g = torch.Generator().manual_seed(105)
    out = {}
    prefix = torch.tensor([0, 3, 17, 0, 600], dtype=torch.int32)
    ext = torch.tensor([5, 1, 9, 1, 530], dtype=torch.int32)
    pos_t, start_t = compute_position_triton(prefix, ext, int(ext.sum()))
    pos_p, start_p = compute_position_torch(prefix, ext)
    assert torch.equal(pos_t, pos_p) and torch.equal(start_t, start_p)
    out.update({"prefix_lens": prefix, "extend_lens": ext, "positions": pos_t,
                "extend_start_loc": start_t})
    # decode positions: clamp(seq_lens - 1, min=0)  (forward_info.py clamp_position)
    seq = torch.tensor([1, 0, 7, 4096], dtype=torch.int64)
    out.update({"decode_seq_lens": seq, "decode_positions": torch.clamp(seq - 1, min=0).to(torch.int64)})

    # write_req_to_token_pool_triton
    bs, ctx = 5, 1200
    table = torch.zeros(8, ctx, dtype=torch.int32)
    req_idx = torch.tensor([6, 0, 3, 7, 2], dtype=torch.int64)
    seq_lens = (prefix + ext).to(torch.int64)
    total = int(ext.sum())
    out_loc = (torch.randperm(5000, generator=g)[:total] + 1).to(torch.int64)
    # cached-prefix slots are copied by the scheduler before the kernel runs
    pre_slots = []
    for i in range(bs):
        p = (torch.randperm(5000, generator=g)[: int(prefix[i])] + 6000).to(torch.int32)
        table[req_idx[i], : int(prefix[i])] = p
        pre_slots.append(p)
    table_in = table.clone()
    write_req_to_token_pool_triton[(bs,)](table, req_idx, prefix.to(torch.int64), seq_lens,
                                          ext.to(torch.int64), out_loc, ctx)
    out.update({"w_table_in": table_in, "w_req_pool_indices": req_idx, "w_pre_lens": prefix.to(torch.int64),
                "w_seq_lens": seq_lens, "w_extend_lens": ext.to(torch.int64),
                "w_out_cache_loc": out_loc, "w_table_out": table})
    _save("positions", **out)


def _paged_setup(g, bs, seq_lens, Hkv, D, pool_slots, n_req_rows, ctx):
    """random KV pool + req_to_token rows with a random slot permutation"""
    k_buf = _randn(pool_slots + 1, Hkv, D, generator=g)
    v_buf = _randn(pool_slots + 1, Hkv, D, generator=g)
    r2t = torch.zeros(n_req_rows, ctx, dtype=torch.int32)
    req_idx = torch.randperm(n_req_rows, generator=g)[:bs].to(torch.int64)
    perm = torch.randperm(pool_slots, generator=g) + 1
    off = 0
    for b in range(bs):
        L = int(seq_lens[b])
        r2t[req_idx[b], :L] = perm[off: off + L].to(torch.int32)
        off += L
    return k_buf, v_buf, r2t, req_idx


def gen_decode():
    from scratchpad.nn.attention.triton_attn.decode_attention import decode_attention_fwd

    g = torch.Generator().manual_seed(106)
    out = {}
    cases = [
        # bs, Hq, Hkv, D, seq_lens, logit_cap
        (4, 8, 2, 64, [1, 37, 64, 130], 0.0),      # GQA group 4, D=64 (Llama-3.2-1B head shape)
        (3, 8, 2, 128, [5, 129, 70], 0.0),         # GQA group 4, D=128 (Llama-3-8B head shape)
        (3, 8, 1, 128, [3, 65, 200], 0.0),         # group 8 (Llama-3-70B TP=8 rank shape)
        (3, 4, 4, 64, [2, 33, 90], 0.0),           # MHA path (group 1)
        (2, 8, 2, 64, [40, 77], 30.0),             # logit soft-cap
    ]
    for i, (bs, Hq, Hkv, D, lens, cap) in enumerate(cases):
        seq = torch.tensor(lens, dtype=torch.int64)
        total = int(seq.sum())
        k_buf, v_buf, r2t, req_idx = _paged_setup(g, bs, seq, Hkv, D, total + 9, bs + 3, max(lens) + 4)
        q = _randn(bs, Hq, D, generator=g)
        o = torch.zeros(bs, Hq, D)
        start = torch.zeros(bs, dtype=torch.int64)
        start[1:] = torch.cumsum(seq[:-1], 0)
        logits = torch.empty(Hq, total)
        scale = 1.0 / math.sqrt(D)
        decode_attention_fwd(q, k_buf, v_buf, o, r2t, req_idx, start, seq, logits,
                             int(seq.max()), scale, cap)
        out.update({f"c{i}_q": q, f"c{i}_k_buffer": k_buf, f"c{i}_v_buffer": v_buf,
                    f"c{i}_req_to_token": r2t, f"c{i}_req_pool_indices": req_idx,
                    f"c{i}_seq_lens": seq, f"c{i}_sm_scale": np.float64(scale),
                    f"c{i}_logit_cap": np.float64(cap), f"c{i}_o": o})
    out["num_cases"] = np.int64(len(cases))
    _save("decode_attention", **out)


def gen_extend():
    from scratchpad.nn.attention.triton_attn.extend_attention import extend_attention_fwd

    g = torch.Generator().manual_seed(107)
    out = {}
    cases = [
        # Hq, Hkv, D, prefix_lens, extend_lens, logit_cap
        (8, 2, 64, [0, 0, 0], [5, 70, 131], 0.0),          # pure prefill, ragged
        (8, 2, 128, [17, 0, 64], [9, 33, 1], 0.0),         # cached prefix + MIXED-like row (extend_len 1)
        (8, 1, 128, [130, 3], [66, 140], 0.0),             # group 8, prefix longer than a tile
        (4, 4, 64, [10, 0], [20, 45], 0.0),                # MHA
        (8, 2, 64, [12, 0], [30, 50], 30.0),               # soft-cap
    ]
    for i, (Hq, Hkv, D, pre, ext, cap) in enumerate(cases):
        bs = len(pre)
        pre_t = torch.tensor(pre, dtype=torch.int32)
        ext_t = torch.tensor(ext, dtype=torch.int32)
        seq = (pre_t + ext_t).to(torch.int64)
        total = int(seq.sum())
        k_buf, v_buf, r2t, req_idx = _paged_setup(g, bs, seq, Hkv, D, total + 5, bs + 2, int(seq.max()) + 4)
        T = int(ext_t.sum())
        q = _randn(T, Hq, D, generator=g)
        k_ext = torch.empty(T, Hkv, D)
        v_ext = torch.empty(T, Hkv, D)
        start = torch.zeros(bs, dtype=torch.int32)
        start[1:] = torch.cumsum(ext_t[:-1], 0)
        # the new tokens' K/V are ALSO in the pool already (KV store precedes the kernel:
        # triton_backend.py forward_extend); k_extend/v_extend are the contiguous copies.
        out_loc = torch.empty(T, dtype=torch.int64)
        for b in range(bs):
            s, p, e = int(start[b]), pre[b], ext[b]
            slots = r2t[req_idx[b], p: p + e].to(torch.int64)
            out_loc[s: s + e] = slots
            k_ext[s: s + e] = k_buf[slots]
            v_ext[s: s + e] = v_buf[slots]
        o = torch.zeros(T, Hq, D)
        scale = 1.0 / math.sqrt(D)
        extend_attention_fwd(q, k_ext, v_ext, o, k_buf, v_buf, r2t, req_idx, seq, ext_t, start,
                             int(ext_t.max()), scale, cap)
        out.update({f"c{i}_q": q,  # k_extend/v_extend == k/v_buffer[out_cache_loc]; not stored
                    f"c{i}_k_buffer": k_buf, f"c{i}_v_buffer": v_buf, f"c{i}_req_to_token": r2t,
                    f"c{i}_req_pool_indices": req_idx, f"c{i}_seq_lens": seq,
                    f"c{i}_extend_seq_lens": ext_t, f"c{i}_extend_prefix_lens": pre_t,
                    f"c{i}_extend_start_loc": start, f"c{i}_out_cache_loc": out_loc,
                    f"c{i}_sm_scale": np.float64(scale), f"c{i}_logit_cap": np.float64(cap),
                    f"c{i}_o": o})
    out["num_cases"] = np.int64(len(cases))
    _save("extend_attention", **out)


def gen_prefill_attention():
    """context_attention_fwd (nn/attention/triton_attn/prefill_attention.py:125-163) under the Triton interpreter:
    cache-less, CAUSAL, variable-length self-attention - sequence b's queries, keys and values are rows
    [b_start_loc[b], +b_seq_len[b]) of q / k / v; sm_scale = 1/sqrt(head dim) is fixed inside the function."""
    from scratchpad.nn.attention.triton_attn.prefill_attention import context_attention_fwd

    g = torch.Generator().manual_seed(131)
    out = {}
    cases = [
        # Hq, Hkv, D, seq_lens
        (8, 2, 64, [5, 70, 131]),          # GQA group 4, D=64; a sequence longer than the kernel's 128-row block
        (8, 2, 128, [1, 129, 33]),         # D=128, a one-token sequence, one that crosses a block by 1
        (8, 1, 128, [200, 3]),             # group 8
        (4, 4, 64, [64, 17, 128]),         # MHA, exact block multiples
    ]
    for i, (Hq, Hkv, D, lens) in enumerate(cases):
        seq = torch.tensor(lens, dtype=torch.int32)
        start = torch.zeros(len(lens), dtype=torch.int32)
        start[1:] = torch.cumsum(seq[:-1], 0)
        T = int(seq.sum())
        q = _randn(T, Hq, D, generator=g)
        k = _randn(T, Hkv, D, generator=g)
        v = _randn(T, Hkv, D, generator=g)
        o = torch.zeros(T, Hq, D)
        context_attention_fwd(q, k, v, o, start, seq, int(seq.max()))
        out.update({f"c{i}_q": q, f"c{i}_k": k, f"c{i}_v": v, f"c{i}_b_start_loc": start, f"c{i}_b_seq_len": seq,
                    f"c{i}_o": o})
        print("prefill case", i, float(o.abs().max()))
    out["num_cases"] = np.int64(len(cases))
    _save("prefill_attention", **out)


def gen_tiny_llama():
    """Full reference LlamaForCausalLM (2 layers) on CPU: ragged prefill then one decode step."""
    import transformers
    import scratchpad.nn.models.llama.llama as L
    from scratchpad.distributed import init_distributed_environment, initialize_model_parallel
    from scratchpad.memory.pool import MHATokenToKVPool, ReqToTokenPool
    from scratchpad.model_executor.cuda_graph_runner import _to_torch
    from scratchpad.model_executor.forward_info import (CaptureHiddenMode, ForwardBatch,
                                                        ForwardMode, compute_position_torch)
    from scratchpad.nn.attention.backend import AttentionBackend
    from scratchpad.nn.attention.triton_attn.decode_attention import decode_attention_fwd
    from scratchpad.nn.attention.triton_attn.extend_attention import extend_attention_fwd

    init_distributed_environment(world_size=1, rank=0, distributed_init_method="tcp://127.0.0.1:29517",
                                 local_rank=0, backend="gloo")
    initialize_model_parallel(1)

    out = {}
    variants = [
        # name, hidden, inter, layers, Hq, Hkv, vocab, theta, scaling, tie
        ("a", 256, 256, 2, 4, 1, 256, 500000.0, None, True),     # D=64, group 4, tied head
        ("b", 256, 320, 2, 2, 2, 192, 500000.0,                    # D=128, MHA, llama3 rope, untied
         {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
          "high_freq_factor": 4.0, "original_max_position_embeddings": 64}, False),
    ]
    for name, hidden, inter, nl, Hq, Hkv, vocab, theta, scaling, tie in variants:
        torch.manual_seed(108)
        cfg = transformers.LlamaConfig(
            vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=nl,
            num_attention_heads=Hq, num_key_value_heads=Hkv, max_position_embeddings=128,
            rms_norm_eps=1e-5, tie_word_embeddings=tie)
        # transformers>=5 keeps RoPE params only in cfg.rope_parameters; the reference reads attributes
        cfg.rope_theta = theta
        cfg.rope_scaling = scaling
        D = hidden // Hq
        model = L.LlamaForCausalLM(cfg).eval()
        for pname, p in model.named_parameters():
            if "norm" in pname:
                p.data = _grid(1.0 + 0.1 * torch.randn_like(p))
            else:
                # multiples of 1/1024, |k| <= 255: exact in bf16/fp16 (std 0.05)
                p.data = _grid(torch.randn_like(p) * 0.05, step=1024.0, lim=255.0)
        _to_torch(model, reverse=False, num_tokens=2)  # CustomOp -> forward_native
        kv = MHATokenToKVPool(96, 1, torch.float32, Hkv, D, nl, "cpu", False)
        r2t = ReqToTokenPool(4, 64, "cpu", False)

        class Probe(AttentionBackend):
            def init_forward_metadata(self, fb):
                pass

            def forward_extend(self, q, k, v, layer, fb, save_kv_cache=True):
                o = torch.empty_like(q)
                fb.token_to_kv_pool.set_kv_buffer(layer, fb.out_cache_loc, k, v)
                extend_attention_fwd(
                    q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), k.contiguous(), v.contiguous(),
                    o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                    *fb.token_to_kv_pool.get_kv_buffer(layer.layer_id),
                    fb.req_to_token_pool.req_to_token, fb.req_pool_indices, fb.seq_lens,
                    fb.extend_seq_lens, fb.extend_start_loc, int(fb.extend_seq_lens.max()),
                    layer.scaling, layer.logit_cap)
                return o

            def forward_decode(self, q, k, v, layer, fb, save_kv_cache=True):
                q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
                o = torch.empty_like(q)
                fb.token_to_kv_pool.set_kv_buffer(layer, fb.out_cache_loc, k, v)
                start = torch.zeros_like(fb.seq_lens)
                start[1:] = torch.cumsum(fb.seq_lens[:-1], 0)
                logits = torch.empty(layer.tp_q_head_num, int(fb.seq_lens.sum()))
                decode_attention_fwd(
                    q.view(-1, layer.tp_q_head_num, layer.qk_head_dim),
                    *fb.token_to_kv_pool.get_kv_buffer(layer.layer_id),
                    o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
                    fb.req_to_token_pool.req_to_token, fb.req_pool_indices, start, fb.seq_lens,
                    logits, int(fb.seq_lens.max()), layer.scaling, layer.logit_cap)
                return o

        backend = Probe()
        g = torch.Generator().manual_seed(109)
        ext = [5, 9, 1]
        bs = len(ext)
        ids = torch.randint(0, vocab, (sum(ext),), generator=g)
        req_idx = torch.tensor([2, 0, 3], dtype=torch.int64)
        slots = (torch.randperm(90, generator=g) + 1).to(torch.int64)
        out_loc = slots[: sum(ext)]
        off = 0
        for b in range(bs):
            r2t.req_to_token[req_idx[b], : ext[b]] = out_loc[off: off + ext[b]].to(torch.int32)
            off += ext[b]
        pre_t = torch.zeros(bs, dtype=torch.int32)
        ext_t = torch.tensor(ext, dtype=torch.int32)
        positions, start_loc = compute_position_torch(pre_t, ext_t)
        seq = torch.tensor(ext, dtype=torch.int64)
        fb = ForwardBatch(
            forward_mode=ForwardMode.EXTEND, batch_size=bs, input_ids=ids, req_pool_indices=req_idx,
            seq_lens=seq, out_cache_loc=out_loc, seq_lens_sum=int(seq.sum()), positions=positions,
            extend_num_tokens=sum(ext), extend_seq_lens=ext_t, extend_prefix_lens=pre_t,
            extend_start_loc=start_loc, extend_prefix_lens_cpu=[0] * bs, extend_seq_lens_cpu=ext,
            req_to_token_pool=r2t, token_to_kv_pool=kv, attn_backend=backend,
            capture_hidden_mode=CaptureHiddenMode.NULL)
        res = model.forward(ids, positions, fb)
        prefill_logits = res.next_token_logits.clone()
        next_ids = torch.argmax(prefill_logits, dim=-1)

        # decode step (ScheduleBatch.prepare_for_decode: seq_lens += 1, alloc, write req_to_token)
        dec_loc = slots[sum(ext): sum(ext) + bs]
        seq2 = seq + 1
        for b in range(bs):
            r2t.req_to_token[req_idx[b], int(seq[b])] = int(dec_loc[b])
        fb2 = ForwardBatch(
            forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=next_ids, req_pool_indices=req_idx,
            seq_lens=seq2, out_cache_loc=dec_loc, seq_lens_sum=int(seq2.sum()),
            positions=torch.clamp(seq2 - 1, min=0).to(torch.int64),
            req_to_token_pool=r2t, token_to_kv_pool=kv, attn_backend=backend,
            capture_hidden_mode=CaptureHiddenMode.NULL)
        res2 = model.forward(next_ids, fb2.positions, fb2)
        decode_logits = res2.next_token_logits.clone()

        # a second extend with a cached prefix: re-run request 1's last 4 tokens on top of its first 5
        sd = {k: v for k, v in model.state_dict().items()}
        pfx = f"{name}_"
        out.update({pfx + "w::" + k: v for k, v in sd.items()})
        out.update({
            pfx + "cfg": np.array([hidden, inter, nl, Hq, Hkv, vocab, int(tie)], np.int64),
            pfx + "rope_theta": np.float64(theta), pfx + "rms_eps": np.float64(1e-5),
            pfx + "max_pos": np.int64(128),
            pfx + "input_ids": ids, pfx + "extend_lens": ext_t, pfx + "req_pool_indices": req_idx,
            pfx + "out_cache_loc": out_loc, pfx + "positions": positions,
            pfx + "prefill_logits": prefill_logits, pfx + "next_ids": next_ids,
            pfx + "decode_out_cache_loc": dec_loc, pfx + "decode_logits": decode_logits,
            pfx + "k_buffer0_after": kv.get_key_buffer(0), pfx + "v_buffer1_after": kv.get_value_buffer(1),
        })
        if scaling is not None:
            out[pfx + "rope_scaling"] = np.array(
                [scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"],
                 scaling["original_max_position_embeddings"]], np.float64)
        from scratchpad.nn.layers import rotary_embedding as _re
        _re._ROPE_DICT.clear()
    _save("tiny_llama", **out)


def gen_tiny_mllama():
    """Reference MllamaForCausalLM (text model with one cross-attention layer) on CPU: a mixed batch
    (two requests with image tokens, one text-only), prefill then one decode step.

    The layer glue is the reference's own code (MllamaTextCrossAttention incl. per-head q/k RMSNorm,
    tanh gates, full_text_row_masked_out_mask, `hidden + residual` after each self-attention layer,
    the shared KV pool with encoder slots first).  The attention CALLS go through a probe backend
    written here from the flashinfer backend's documented semantics (flashinfer itself is absent):
    cross-attention = non-causal over req_to_token[req, 0:encoder_len], K/V stored at
    encoder_out_cache_loc; self-attention = causal over req_to_token[req, encoder_len:encoder_len+seq]
    (nn/attention/flashinfer_backend.py:400-417, 593-621, 792-828)."""
    import transformers.models.mllama.configuration_mllama as cm
    import scratchpad.nn.models.llama.mllama as M
    import torch.distributed as dist
    from scratchpad.distributed import init_distributed_environment, initialize_model_parallel
    from scratchpad.memory.pool import MHATokenToKVPool, ReqToTokenPool
    from scratchpad.model_executor.cuda_graph_runner import _to_torch
    from scratchpad.model_executor.forward_info import CaptureHiddenMode, ForwardBatch, ForwardMode
    from scratchpad.nn.attention.backend import AttentionBackend

    if not dist.is_initialized():
        init_distributed_environment(world_size=1, rank=0, distributed_init_method="tcp://127.0.0.1:29518",
                                     local_rank=0, backend="gloo")
        initialize_model_parallel(1)
    torch.manual_seed(110)
    hidden, inter, nl, Hq, Hkv, vocab = 256, 256, 3, 4, 2, 200
    cfg = cm.MllamaTextConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=nl,
                              num_attention_heads=Hq, num_key_value_heads=Hkv, intermediate_size=inter,
                              cross_attention_layers=[1], rms_norm_eps=1e-5, max_position_embeddings=128,
                              pad_token_id=0, bos_token_id=1, eos_token_id=2)
    cfg.rope_theta = 500000.0
    cfg.rope_scaling = None
    D = hidden // Hq
    from scratchpad.nn.layers.logits_processor import LogitsProcessor
    model = M.MllamaForCausalLM(cfg, quant_config=None).eval()
    logits_processor = LogitsProcessor(cfg)          # MllamaForConditionalGeneration.__init__ :801
    for pname, p in model.named_parameters():
        if "gate" in pname and p.numel() == 1:
            p.data = torch.tensor([0.5 if "attn_gate" in pname else -0.75])
        elif "norm" in pname:
            p.data = _grid(1.0 + 0.1 * torch.randn_like(p))
        else:
            p.data = _grid(torch.randn_like(p) * 0.05, step=1024.0, lim=255.0)
    _to_torch(model, reverse=False, num_tokens=2)
    kv = MHATokenToKVPool(96, 1, torch.float32, Hkv, D, nl, "cpu", False)
    r2t = ReqToTokenPool(4, 64, "cpu", False)

    def dense(q, kb, vb, idx, causal_offset, scale):
        """q [n,Hq,D] over pool rows idx; causal_offset = index of row 0's last visible key, or None"""
        k = kb[idx].float()
        v = vb[idx].float()
        g = q.shape[1] // k.shape[1]
        k = k.repeat_interleave(g, dim=1)
        v = v.repeat_interleave(g, dim=1)
        s = torch.einsum("nhd,lhd->hnl", q.float(), k) * scale
        if causal_offset is not None:
            n, L = q.shape[0], k.shape[0]
            col = torch.arange(L).view(1, L)
            row = torch.arange(n).view(n, 1) + causal_offset
            s = s.masked_fill(col > row, float("-inf"))
        return torch.einsum("hnl,lhd->nhd", torch.softmax(s, -1), v)

    class Probe(AttentionBackend):
        def init_forward_metadata(self, fb):
            pass

        def _run(self, q, k, v, layer, fb, starts, lens):
            q3 = q.reshape(-1, layer.tp_q_head_num, layer.qk_head_dim)
            o = torch.zeros_like(q3)
            cross = layer.is_cross_attention
            if k is not None:
                loc = fb.encoder_out_cache_loc if cross else fb.out_cache_loc
                fb.token_to_kv_pool.set_kv_buffer(layer, loc, k, v)
            kb, vb = fb.token_to_kv_pool.get_kv_buffer(layer.layer_id)
            for b in range(fb.batch_size):
                enc = int(fb.encoder_lens[b])
                row = fb.req_to_token_pool.req_to_token[int(fb.req_pool_indices[b])]
                s0, n = starts[b], lens[b]
                if n == 0:
                    continue
                if cross:
                    if enc == 0:
                        continue
                    o[s0:s0 + n] = dense(q3[s0:s0 + n], kb, vb, row[:enc].long(), None, layer.scaling)
                else:
                    L = int(fb.seq_lens[b])
                    o[s0:s0 + n] = dense(q3[s0:s0 + n], kb, vb, row[enc:enc + L].long(), L - n, layer.scaling)
            return o.reshape(-1, layer.tp_q_head_num * layer.v_head_dim)

        def forward_extend(self, q, k, v, layer, fb, save_kv_cache=True):
            return self._run(q, k, v, layer, fb, fb.extend_start_loc.tolist(), fb.extend_seq_lens.tolist())

        def forward_decode(self, q, k, v, layer, fb, save_kv_cache=True):
            return self._run(q, k, v, layer, fb, list(range(fb.batch_size)), [1] * fb.batch_size)

    backend = Probe()
    g = torch.Generator().manual_seed(111)
    enc = [10, 0, 7]
    text = [6, 5, 4]
    bs = 3
    req_idx = torch.tensor([1, 3, 0], dtype=torch.int64)
    slots = (torch.randperm(90, generator=g) + 1).to(torch.int64)
    # scheduler layout (prepare_for_extend + prepare_encoder_info_extend): per request the slots of
    # [encoder tokens | text tokens] are contiguous in out_cache_loc order
    pt, enc_loc, dec_loc = 0, [], []
    for b in range(bs):
        n = enc[b] + text[b]
        r2t.req_to_token[req_idx[b], :n] = slots[pt:pt + n].to(torch.int32)
        enc_loc.append(slots[pt:pt + enc[b]])
        dec_loc.append(slots[pt + enc[b]:pt + n])
        pt += n
    out_loc, enc_out_loc = torch.cat(dec_loc), torch.cat(enc_loc)
    ids = torch.randint(3, vocab, (sum(text),), generator=g)
    ext_t = torch.tensor(text, dtype=torch.int32)
    start = torch.zeros(bs, dtype=torch.int32)
    start[1:] = torch.cumsum(ext_t[:-1], 0)
    positions = torch.cat([torch.arange(n) for n in text]).to(torch.int64)
    cross_states = _randn(sum(enc), hidden, generator=g, scale=0.5)
    enc_t = torch.tensor(enc, dtype=torch.int64)
    seq = torch.tensor(text, dtype=torch.int64)
    fb = ForwardBatch(
        forward_mode=ForwardMode.EXTEND, batch_size=bs, input_ids=ids, req_pool_indices=req_idx, seq_lens=seq,
        out_cache_loc=out_loc, seq_lens_sum=int(seq.sum()), positions=positions,
        extend_num_tokens=sum(text), extend_seq_lens=ext_t, extend_prefix_lens=torch.zeros(bs, dtype=torch.int32),
        extend_start_loc=start, extend_prefix_lens_cpu=[0] * bs, extend_seq_lens_cpu=text,
        encoder_cached=[False, True, False], encoder_lens=enc_t, encoder_lens_cpu=enc,
        encoder_out_cache_loc=enc_out_loc, req_to_token_pool=r2t, token_to_kv_pool=kv, attn_backend=backend,
        capture_hidden_mode=CaptureHiddenMode.NULL)

    class Host:   # get_full_text_row_masked_out_mask is a method of the conditional-generation class
        pass
    mask = M.MllamaForConditionalGeneration.get_full_text_row_masked_out_mask(Host(), fb)
    hidden_states = model.model(input_ids=ids, positions=positions, cross_attention_states=cross_states,
                                cross_attention_mask=None, full_text_row_masked_out_mask=mask,
                                forward_batch=fb, skip_cross_attention=False)
    prefill_logits = logits_processor(ids, hidden_states, model.lm_head, fb).next_token_logits.clone()
    next_ids = prefill_logits.argmax(-1)

    # decode step: locs = encoder_lens + seq_lens (schedule_batch.py:1281-1283)
    dloc = slots[pt:pt + bs]
    for b in range(bs):
        r2t.req_to_token[req_idx[b], enc[b] + text[b]] = int(dloc[b])
    seq2 = seq + 1
    fb2 = ForwardBatch(
        forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=next_ids, req_pool_indices=req_idx,
        seq_lens=seq2, out_cache_loc=dloc, seq_lens_sum=int(seq2.sum()),
        positions=torch.clamp(seq2 - 1, min=0).to(torch.int64), encoder_cached=[True] * bs,
        encoder_lens=enc_t, encoder_lens_cpu=enc, encoder_out_cache_loc=torch.zeros(0, dtype=torch.int64),
        req_to_token_pool=r2t, token_to_kv_pool=kv, attn_backend=backend,
        capture_hidden_mode=CaptureHiddenMode.NULL)
    mask2 = M.MllamaForConditionalGeneration.get_full_text_row_masked_out_mask(Host(), fb2)
    hs2 = model.model(input_ids=next_ids, positions=fb2.positions, cross_attention_states=None,
                      cross_attention_mask=None, full_text_row_masked_out_mask=mask2, forward_batch=fb2,
                      skip_cross_attention=False)
    decode_logits = logits_processor(next_ids, hs2, model.lm_head, fb2).next_token_logits.clone()

    # per-head q/k RMSNorm on its own (MllamaTextRMSNorm): [T, H, D] -> normalised over D
    norm = model.model.layers[1].cross_attn.q_norm
    xn = _randn(7, Hq, D, generator=g)
    out = {"w::" + k: v for k, v in model.state_dict().items()}
    out.update({
        "cfg": np.array([hidden, inter, nl, Hq, Hkv, vocab], np.int64), "cross_layers": np.array([1], np.int64),
        "input_ids": ids, "text_lens": ext_t, "encoder_lens": enc_t, "req_pool_indices": req_idx,
        "out_cache_loc": out_loc, "encoder_out_cache_loc": enc_out_loc, "positions": positions,
        "cross_attention_states": cross_states, "row_mask_extend": mask.to(torch.int64),
        "row_mask_decode": mask2.to(torch.int64), "req_to_token_extend_rows": r2t.req_to_token.clone(),
        "prefill_logits": prefill_logits, "next_ids": next_ids, "decode_out_cache_loc": dloc,
        "decode_logits": decode_logits, "k_buffer1_after": kv.get_key_buffer(1),
        "qnorm_x": xn, "qnorm_w": norm.weight.data, "qnorm_y": norm(xn),
    })
    _save("tiny_mllama", **out)


def gen_radix_cache():
    """Seeded op trace through the reference's RadixCache (memory/radix_cache.py) with its own
    ReqToTokenPool / TokenToKVPoolAllocator on CPU; time.time() is replaced by a counter so the
    LRU order is a function of the op sequence only.  Stored as JSON: ops with inputs + outputs."""
    import json
    import random
    import time as _time
    from types import SimpleNamespace
    from scratchpad.memory.pool import ReqToTokenPool, TokenToKVPoolAllocator
    import scratchpad.memory.radix_cache as rc

    tick = [0.0]

    def fake_time():
        tick[0] += 1.0
        return tick[0]

    real_time = _time.time
    rc.time.time = fake_time
    try:
        rnd = random.Random(1234)
        pool_size, ctx = 400, 64
        r2t = ReqToTokenPool(16, ctx, "cpu", False)
        alloc = TokenToKVPoolAllocator(pool_size, torch.float32, "cpu", None)
        cache = rc.RadixCache(r2t, alloc, page_size=1)
        ops = []
        # a small alphabet and shared stems make deep splits likely
        stems = [[rnd.randrange(6) for _ in range(rnd.randrange(3, 14))] for _ in range(5)]

        def rand_key():
            k = list(rnd.choice(stems))[: rnd.randrange(1, 14)]
            k += [rnd.randrange(6) for _ in range(rnd.randrange(0, 12))]
            return k

        def snap():
            return {"evictable": cache.evictable_size(), "protected": cache.protected_size(),
                    "total": cache.total_size(), "available": alloc.available_size(),
                    "values_sorted": sorted(int(x) for x in cache.all_values_flatten().tolist())
                    if cache.total_size() else []}

        live = {}      # rid -> SimpleNamespace request holding locks
        next_rid = [0]
        for step in range(260):
            kind = rnd.choices(["insert", "match", "req_begin", "req_chunk", "req_finish", "evict"],
                               weights=[2, 3, 4, 2, 4, 2])[0]
            if kind == "insert":
                key = rand_key()
                slots = alloc.alloc(len(key))
                if slots is None:
                    continue
                n = cache.insert(key, slots.clone())
                alloc.free(slots[:n])
                ops.append({"op": "insert", "key": key, "slots": slots.tolist(), "ret": int(n), "after": snap()})
            elif kind == "match":
                key = rand_key()
                val, node = cache.match_prefix(key)
                ops.append({"op": "match", "key": key, "value": [int(x) for x in val.tolist()],
                            "node_key": list(node.key), "after": snap()})
            elif kind == "req_begin":
                if r2t.available_size() == 0:
                    continue
                ids = rand_key() + [rnd.randrange(6)]
                req = SimpleNamespace(rid=f"r{next_rid[0]}", origin_input_ids=ids, output_ids=[],
                                      fill_ids=None, prefix_indices=[], last_node=None, req_pool_idx=None)
                next_rid[0] += 1
                # Req.init_next_round_input (schedule_batch.py:472-492) with max prefix = len - 1
                chunk = rnd.randrange(1, len(ids) + 1)
                req.fill_ids = ids[:chunk] if chunk < len(ids) else list(ids)
                prefix, node = cache.match_prefix(req.fill_ids[: max(len(req.fill_ids) - 1, 0)])
                need = len(req.fill_ids) - len(prefix)
                if alloc.available_size() < need:
                    cache.evict(need)
                if alloc.available_size() < need:
                    continue
                req.prefix_indices, req.last_node = prefix, node
                cache.inc_lock_ref(node)
                req.req_pool_idx = r2t.alloc(1)[0]
                new = alloc.alloc(need)
                r2t.write((req.req_pool_idx, slice(0, len(prefix))), prefix.to(torch.int32))
                r2t.write((req.req_pool_idx, slice(len(prefix), len(req.fill_ids))), new.to(torch.int32))
                live[req.rid] = req
                ops.append({"op": "req_begin", "rid": req.rid, "ids": ids, "fill_len": len(req.fill_ids),
                            "prefix": [int(x) for x in prefix.tolist()], "new_slots": new.tolist(),
                            "req_pool_idx": int(req.req_pool_idx), "after": snap()})
            elif kind == "req_chunk":
                cands = [r for r in live.values() if len(r.fill_ids) < len(r.origin_input_ids)]
                if not cands:
                    continue
                req = rnd.choice(cands)
                cache.cache_unfinished_req(req)
                pre = [int(x) for x in req.prefix_indices.tolist()]
                # next chunk: extend fill_ids, allocate the new part
                grow = rnd.randrange(1, len(req.origin_input_ids) - len(req.fill_ids) + 1)
                old = len(req.fill_ids)
                if alloc.available_size() < grow:
                    cache.evict(grow)
                if alloc.available_size() < grow:
                    ops.append({"op": "req_chunk", "rid": req.rid, "prefix_after": pre, "grow": 0,
                                "new_slots": [], "row": r2t.req_to_token[req.req_pool_idx, :old].tolist(),
                                "after": snap()})
                    continue
                req.fill_ids = req.origin_input_ids[: old + grow]
                new = alloc.alloc(grow)
                r2t.write((req.req_pool_idx, slice(old, old + grow)), new.to(torch.int32))
                ops.append({"op": "req_chunk", "rid": req.rid, "prefix_after": pre, "grow": grow,
                            "new_slots": new.tolist(),
                            "row": r2t.req_to_token[req.req_pool_idx, : old + grow].tolist(), "after": snap()})
            elif kind == "req_finish":
                cands = [r for r in live.values() if len(r.fill_ids) == len(r.origin_input_ids)]
                if not cands:
                    continue
                req = rnd.choice(cands)
                n_out = rnd.randrange(1, 5)
                if alloc.available_size() < n_out:
                    cache.evict(n_out)
                if alloc.available_size() < n_out:
                    continue
                # decode steps: every output token but the last has its KV written
                outs = [rnd.randrange(6) for _ in range(n_out)]
                base = len(req.origin_input_ids)
                dec = alloc.alloc(n_out - 1) if n_out > 1 else torch.empty(0, dtype=torch.int64)
                if n_out > 1:
                    r2t.write((req.req_pool_idx, slice(base, base + n_out - 1)), dec.to(torch.int32))
                req.output_ids = outs
                cache.cache_finished_req(req)
                del live[req.rid]
                ops.append({"op": "req_finish", "rid": req.rid, "output_ids": outs,
                            "decode_slots": dec.tolist(), "after": snap()})
            else:
                n = rnd.randrange(1, 40)
                cache.evict(n)
                ops.append({"op": "evict", "n": n, "free_slots": alloc.free_slots.tolist(), "after": snap()})
        path = os.path.join(HERE, "radix_cache.json")
        with open(path, "w") as f:
            json.dump({"pool_size": pool_size, "context_len": ctx, "max_reqs": 16, "ops": ops}, f,
                      separators=(",", ":"))
        kinds = {}
        for o in ops:
            kinds[o["op"]] = kinds.get(o["op"], 0) + 1
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(ops)} ops {kinds}")
    finally:
        rc.time.time = real_time


def gen_sampling():
    """The reference's torch sampler functions (nn/layers/sampler.py:195-232) on seeded rows.
    ``torch.multinomial`` is intercepted so the filtered, sorted distribution the reference would
    sample from is recorded, and the draw is made by inverse CDF with a recorded uniform."""
    import scratchpad.nn.layers.sampler as ref_sampler

    g = torch.Generator().manual_seed(2718)
    bs, vocab = 24, 1000
    scales = torch.tensor([0.5, 1.0, 2.0, 4.0, 8.0, 1.5] * 4).view(-1, 1)
    logits = torch.randn(bs, vocab, generator=g) * scales
    temperatures = torch.tensor([1.0, 0.7, 1.3, 0.5] * 6).view(-1, 1)
    # rows 20..23: heavy exact ties (probabilities on a coarse grid)
    counts = torch.randint(1, 5, (4, vocab), generator=g).float()
    probs = torch.softmax(logits / temperatures, dim=-1)
    probs[20:] = counts / counts.sum(dim=-1, keepdim=True)
    top_ks = torch.tensor([1 << 30, 1, 5, 50, 200, 1 << 30, 1 << 30, 20, 3, 1 << 30, 7, 100,
                           1 << 30, 40, 1 << 30, 2, 64, 1 << 30, 10, 1 << 30, 1 << 30, 30, 1 << 30, 500],
                          dtype=torch.int32)
    top_ps = torch.tensor([1.0, 1.0, 0.9, 0.5, 0.95, 0.1, 0.8, 1.0, 0.3, 0.99, 1.0, 0.6,
                           0.7, 0.2, 0.999, 0.9, 1.0, 0.05, 0.85, 0.4, 1.0, 0.5, 0.25, 0.9])
    min_ps = torch.tensor([0.0, 0.0, 0.0, 0.05, 0.0, 0.0, 0.3, 0.1, 0.0, 0.02, 0.5, 0.0,
                           0.0, 0.0, 0.2, 0.0, 0.01, 0.0, 0.0, 0.9, 0.0, 0.0, 1.0, 0.0])
    uniform = torch.rand(bs, generator=g)
    captured = {}
    real_multinomial = torch.multinomial

    def recording_multinomial(weights, num_samples=1, **kw):
        captured["weights"] = weights.clone()
        cdf = torch.cumsum(weights.double(), dim=-1)
        target = uniform.double().view(-1, 1) * cdf[:, -1:]
        return (cdf <= target).sum(dim=-1, keepdim=True).clamp(max=weights.shape[-1] - 1)

    torch.multinomial = recording_multinomial
    try:
        out = {}
        for tag, need_min_p in (("minp", True), ("nominp", False)):
            ids = ref_sampler.top_k_top_p_min_p_sampling_from_probs_torch(
                probs.clone(), top_ks, top_ps, min_ps, need_min_p)
            w = captured["weights"]
            order = probs.sort(dim=-1, descending=True)[1]
            keep = torch.zeros(bs, vocab, dtype=torch.bool).scatter_(1, order, w > 0)
            out[f"{tag}_keep"] = keep
            out[f"{tag}_keep_count"] = (w > 0).sum(-1)
            out[f"{tag}_ids"] = ids.to(torch.int64)
            out[f"{tag}_sorted_weights"] = w
    finally:
        torch.multinomial = real_multinomial
    out["top_p_normalized"] = ref_sampler.top_p_normalize_probs_torch(probs.clone(), top_ps)
    lp = torch.log_softmax(logits, dim=-1)
    tv, ti = ref_sampler.get_top_logprobs(lp, [0, 3, 5, 1] * 6)
    out["top_logprobs_val"] = torch.tensor([v + [0.0] * (5 - len(v)) for v in tv])
    out["top_logprobs_idx"] = torch.tensor([i + [-1] * (5 - len(i)) for i in ti])
    _save("sampling", logits=logits, temperatures=temperatures, probs=probs, top_ks=top_ks, top_ps=top_ps,
          min_ps=min_ps, uniform=uniform, **out)


def gen_mllama_vision():
    """The reference's MllamaVisionModel (nn/models/llama/mllama.py:283-466) + multi_modal_projector
    shape on CPU, tiny config: 28x28 images, 14x14 patches (4 + class token = 5 patches, padded to 8),
    hidden 32, 2 heads (head size 16), 3 local + 2 global (gated) layers, up to 4 tiles.
    Two batches: all tiles real / some tiles padding, so both branches of the tile mask are recorded.
    (The cache-less varlen kernel behind VisionTritonAttention, context_attention_fwd, does run under the Triton
    interpreter - but it is causal only, and its vision caller passes an is_causal argument it does not accept
    (vision.py:315, 361): its own fixture is prefill_attention.npz, the CAUSAL cache-less path; the non-causal
    form VisionAttention needs has no reference output to record.)"""
    import transformers.models.mllama.configuration_mllama as cm
    import scratchpad.nn.models.llama.mllama as M
    import torch.distributed as dist
    from scratchpad.distributed import init_distributed_environment, initialize_model_parallel

    if not dist.is_initialized():
        init_distributed_environment(world_size=1, rank=0, distributed_init_method="tcp://127.0.0.1:29519",
                                     local_rank=0, backend="gloo")
        initialize_model_parallel(1)
    torch.manual_seed(271)
    cfg = cm.MllamaVisionConfig(hidden_size=32, hidden_act="gelu", num_hidden_layers=3, num_global_layers=2,
                                attention_heads=2, num_channels=3, intermediate_size=64, vision_output_dim=96,
                                image_size=28, patch_size=14, norm_eps=1e-5, max_num_tiles=4,
                                intermediate_layers_indices=[0, 2],
                                supported_aspect_ratios=[[1, 1], [1, 2], [1, 3], [1, 4], [2, 1], [2, 2], [3, 1], [4, 1]])
    model = M.MllamaVisionModel(cfg).eval()
    g = torch.Generator().manual_seed(272)
    weights = {}
    for name, prm in model.named_parameters():
        if name.endswith("gate") or "gate_attn" in name or "gate_ffn" in name:
            prm.data = torch.tensor([0.6 if "ffn" in name else -0.4]) if prm.numel() == 1 else prm.data
        elif "layernorm" in name and name.endswith("weight"):
            prm.data = _grid(1.0 + 0.1 * torch.randn(prm.shape, generator=g))
        elif name.endswith("bias"):
            prm.data = _grid(0.05 * torch.randn(prm.shape, generator=g), step=1024.0)
        else:
            prm.data = _grid(torch.randn(prm.shape, generator=g) * 0.08, step=1024.0, lim=255.0)
        weights["w." + name] = prm.data.clone()
    out = dict(weights)
    out["cfg"] = np.array([cfg.hidden_size, cfg.attention_heads, cfg.intermediate_size, cfg.num_hidden_layers,
                           cfg.num_global_layers, cfg.image_size, cfg.patch_size, cfg.max_num_tiles,
                           cfg.max_aspect_ratio_id, cfg.num_channels], dtype=np.int64)
    out["intermediate_layers_indices"] = np.array(cfg.intermediate_layers_indices, dtype=np.int64)
    for tag, ar_ids, ar_mask in (("full", [[6], [4]], [[[1, 1, 1, 1]], [[1, 1, 1, 1]]]),
                                 ("ragged", [[2], [1]], [[[1, 1, 0, 0]], [[1, 0, 0, 0]]])):
        pixels = _grid(torch.randn(2, 1, 4, 3, 28, 28, generator=g))
        ids = torch.tensor(ar_ids, dtype=torch.int64)
        mask = torch.tensor(ar_mask, dtype=torch.int64)
        pixels = pixels * mask.view(2, 1, 4, 1, 1, 1)        # the processor zero-fills unused tiles
        y = model(pixels, ids, mask)
        out.update({f"{tag}_pixel_values": pixels, f"{tag}_aspect_ratio_ids": ids,
                    f"{tag}_aspect_ratio_mask": mask, f"{tag}_out": y})
        print(tag, tuple(y.shape), float(y.abs().max()))
    _save("mllama_vision", **out)


def gen_input_logprobs():
    """The reference's input-logprob path: ScheduleBatch.prepare_for_extend (schedule_batch.py:952-1040: relative
    logprob start, ids whose logprob is reported) for requests with / without cached prefixes and chunked prompts, then
    LogitsProcessor.forward (logits_processor.py:148-340) with LogitsMetadata.from_forward_batch on seeded hidden
    states and an LM head: sampled logits, input token logprobs, top-k and requested-id logprobs."""
    import types
    from scratchpad.distributed import init_distributed_environment, initialize_model_parallel
    from scratchpad.distributed import parallel_state as ps
    from scratchpad.memory.pool import MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator
    from scratchpad.model_executor.forward_info import CaptureHiddenMode, ForwardBatch, ForwardMode
    from scratchpad.nn.layers.logits_processor import LogitsProcessor
    from scratchpad.sampling.sampling_params import SamplingParams
    from scratchpad.scheduler import schedule_batch as SB

    if ps._WORLD is None:
        init_distributed_environment(world_size=1, rank=0, distributed_init_method="tcp://127.0.0.1:29519",
                                     local_rank=0, backend="gloo")
        initialize_model_parallel(1)
    g = torch.Generator().manual_seed(131)
    vocab, vocab_padded, hidden = 200, 256, 64
    head = _grid(torch.randn(vocab_padded, hidden, generator=g) * 0.25, step=256.0)
    proc = LogitsProcessor(types.SimpleNamespace(vocab_size=vocab))
    sp = SamplingParams(max_new_tokens=4)
    # per case: list of (prompt length, cached prefix length, chunk end or None, logprob_start_len (-1: the default
    # "last token only"), top_logprobs_num, token_ids_logprob)
    cases = [
        [(6, 0, None, 0, 2, None), (5, 0, None, 2, 0, [7, 3, 150]), (4, 0, None, -1, 3, None)],
        [(9, 4, None, 0, 1, None), (7, 2, None, 5, 4, [0, 199]), (3, 0, None, 0, 0, None), (8, 0, 5, 1, 2, [11])],
        [(5, 0, None, -1, 0, None), (6, 3, None, -1, 2, None)],            # nobody wants input logprobs
        [(1, 0, None, 0, 5, [4, 5]), (12, 0, 12, 11, 1, None), (10, 6, 8, 2, 3, None)],
    ]
    out = {"num_cases": np.int64(len(cases)), "vocab": np.int64(vocab), "head": head}
    for ci, case in enumerate(cases):
        kv = MHATokenToKVPool(256, 1, torch.float32, 1, 8, 1, "cpu", False)
        alloc = TokenToKVPoolAllocator(256, torch.float32, "cpu", kv)
        r2t = ReqToTokenPool(8, 64, "cpu", False)
        reqs = []
        for i, (n, pre, chunk_end, start, k, ids) in enumerate(case):
            prompt = torch.randint(1, vocab, (n,), generator=g).tolist()
            r = SB.Req(str(i), "", prompt, sp, return_logprob=True, top_logprobs_num=k, token_ids_logprob=ids)
            r.logprob_start_len = n - 1 if start < 0 else start
            r.fill_ids = list(prompt) if chunk_end is None else list(prompt[:chunk_end])
            r.prefix_indices = alloc.alloc(pre) if pre else []
            r.extend_input_len = len(r.fill_ids) - len(r.prefix_indices)
            reqs.append(r)
            out[f"c{ci}_r{i}_prompt"] = np.array(prompt, np.int64)
            out[f"c{ci}_r{i}_ids"] = np.array([] if ids is None else ids, np.int64)
        mc = types.SimpleNamespace(is_encoder_decoder=False, vocab_size=vocab)
        b = SB.ScheduleBatch.init_new(reqs, r2t, alloc, None, mc, False, None, False)
        b.prepare_for_extend()
        ext = list(b.extend_lens)
        hidden_states = _randn(sum(ext), hidden, generator=g)
        fb = ForwardBatch(
            forward_mode=ForwardMode.EXTEND, batch_size=len(reqs), input_ids=b.input_ids,
            req_pool_indices=b.req_pool_indices, seq_lens=b.seq_lens, out_cache_loc=b.out_cache_loc,
            seq_lens_sum=b.seq_lens_sum, extend_num_tokens=sum(ext),
            extend_seq_lens=torch.tensor(ext, dtype=torch.int32), extend_seq_lens_cpu=ext,
            extend_prefix_lens_cpu=list(b.prefix_lens), return_logprob=True, top_logprobs_nums=b.top_logprobs_nums,
            token_ids_logprobs=b.token_ids_logprobs, extend_logprob_start_lens_cpu=b.extend_logprob_start_lens,
            extend_input_logprob_token_ids_gpu=b.extend_input_logprob_token_ids,
            capture_hidden_mode=CaptureHiddenMode.NULL)
        res = proc.forward(b.input_ids, hidden_states, types.SimpleNamespace(weight=head), fb)
        out.update({
            f"c{ci}_spec": np.array([[n, pre, -1 if ce is None else ce, st, k] for n, pre, ce, st, k, _ in case], np.int64),
            f"c{ci}_has_ids": np.array([ids is not None for *_, ids in case]),
            f"c{ci}_extend_lens": np.array(ext, np.int64), f"c{ci}_prefix_lens": np.array(b.prefix_lens, np.int64),
            f"c{ci}_extend_logprob_start_lens": np.array(b.extend_logprob_start_lens, np.int64),
            f"c{ci}_extend_input_logprob_token_ids": b.extend_input_logprob_token_ids.to(torch.int64),
            f"c{ci}_hidden": hidden_states, f"c{ci}_next_token_logits": res.next_token_logits,
            f"c{ci}_has_input": np.bool_(res.input_token_logprobs is not None),
        })
        if res.input_token_logprobs is not None:
            out[f"c{ci}_input_token_logprobs"] = res.input_token_logprobs
        for field in ("input_top_logprobs_val", "input_top_logprobs_idx", "input_token_ids_logprobs_val",
                      "input_token_ids_logprobs_idx"):
            v = getattr(res, field)
            out[f"c{ci}_has_{field}"] = np.bool_(v is not None)
            if v is not None:                       # ragged [request][position][k]: flattened + per-request shape
                for i, per_req in enumerate(v):
                    if field.startswith("input_token_ids") and case[i][5] is None:
                        continue                     # the reference indexes with None there: not a defined output
                    flat = [x for row in per_req for x in row]
                    dt = np.int64 if field.endswith("idx") else np.float32
                    out[f"c{ci}_{field}_r{i}"] = np.array(flat, dt)
                    out[f"c{ci}_{field}_r{i}_rows"] = np.int64(len(per_req))
    _save("input_logprobs", **out)


def gen_api_surface():
    """The CALL SURFACE of the reference's seams for this path, as data: dataclass field names / order / defaults,
    parameter names / order / kinds / defaults of the seam classes' public methods (inspect on the imported reference),
    and the call FORMS (positional count + keyword names) its own callers use on them (an ast walk over the caller
    files: scheduler.py, tp_worker*.py, model_runner.py, cuda_graph_runner.py, the attention backends and layers).
    Names, counts and default reprs only - no source text.  tests/test_api_surface.py binds every form on ours."""
    import ast
    import dataclasses
    import inspect
    import json

    import scratchpad.distributed.communication_op as comm
    import scratchpad.distributed.parallel_state as ps
    import scratchpad.memory.chunk_cache as cc
    import scratchpad.memory.pool as pool
    import scratchpad.memory.radix_cache as rc
    import scratchpad.model_executor.custom_op as cop
    import scratchpad.model_executor.forward_info as fi
    import scratchpad.nn.attention.backend as be
    import scratchpad.nn.attention.radix_attention as ra
    import scratchpad.scheduler.schedule_batch as sb

    def default_repr(v):
        if v is inspect.Parameter.empty or v is dataclasses.MISSING:
            return None
        if v is None or isinstance(v, (bool, int, float, str)):
            return repr(v)
        return "<" + type(v).__name__ + ">"

    def params(fn):
        out = []
        for p in inspect.signature(fn).parameters.values():
            if p.name in ("self", "cls"):
                continue
            out.append({"name": p.name, "kind": p.kind.name, "required": p.default is inspect.Parameter.empty
                        and p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD), "default": default_repr(p.default)})
        return out

    def fields(cls):
        return [{"name": f.name, "required": f.default is dataclasses.MISSING and f.default_factory is dataclasses.MISSING,
                 "default": default_repr(f.default)} for f in dataclasses.fields(cls)]

    def methods(cls, only=None):
        out = {}
        for name, member in cls.__dict__.items():
            if name.startswith("_") and name != "__init__":
                continue
            if only is not None and name not in only:
                continue
            kind = "method"
            if isinstance(member, classmethod):
                member, kind = member.__func__, "classmethod"
            elif isinstance(member, staticmethod):
                member, kind = member.__func__, "staticmethod"
            elif isinstance(member, property):
                out[name] = {"kind": "property", "params": []}
                continue
            if not inspect.isfunction(member):
                continue
            member = inspect.unwrap(member)
            out[name] = {"kind": kind, "params": params(member)}
        return out

    classes = {
        "AttentionBackend": be.AttentionBackend, "RadixAttention": ra.RadixAttention, "CustomOp": cop.CustomOp,
        "KVCache": pool.KVCache, "MHATokenToKVPool": pool.MHATokenToKVPool, "ReqToTokenPool": pool.ReqToTokenPool,
        "TokenToKVPoolAllocator": pool.TokenToKVPoolAllocator, "ScheduleBatch": sb.ScheduleBatch, "Req": sb.Req,
        "ForwardBatch": fi.ForwardBatch, "ForwardMode": fi.ForwardMode, "CaptureHiddenMode": fi.CaptureHiddenMode,
        "RadixCache": rc.RadixCache, "ChunkCache": cc.ChunkCache,
    }
    surface = {"dataclasses": {n: fields(c) for n, c in (("ForwardBatch", fi.ForwardBatch),
                                                         ("ModelWorkerBatch", sb.ModelWorkerBatch),
                                                         ("ScheduleBatch", sb.ScheduleBatch))},
               "enums": {n: {m.name: int(m.value) for m in c} for n, c in (("ForwardMode", fi.ForwardMode),
                                                                         ("CaptureHiddenMode", fi.CaptureHiddenMode))},
               "methods": {n: methods(c) for n, c in classes.items()},
               "functions": {}}
    surface["methods"]["GroupCoordinator"] = methods(ps.GroupCoordinator, only=("all_reduce", "all_gather", "graph_capture"))
    # the worker seam (managers/tp_worker*.py): only what scheduler/scheduler.py calls on it for this path
    import scratchpad.managers.tp_worker as tw
    import scratchpad.managers.tp_worker_client as twc
    worker_calls = ("get_worker_info", "get_pad_input_ids_func", "get_tp_cpu_group", "get_memory_pool", "forward_batch_generation")
    surface["methods"]["TpModelWorker"] = methods(tw.TpModelWorker, only=worker_calls)
    surface["methods"]["TpModelWorkerClient"] = methods(twc.TpModelWorkerClient, only=worker_calls + ("resolve_last_batch_result",))
    for mod, names in ((comm, ("tensor_model_parallel_all_reduce", "tensor_model_parallel_all_gather")),
                       (ps, ("graph_capture", "get_tp_group", "get_tensor_model_parallel_world_size",
                             "get_tensor_model_parallel_rank"))):
        for n in names:
            surface["functions"][n] = params(inspect.unwrap(getattr(mod, n)))

    # ---- call forms: every `<receiver>.<method>(...)` in the caller files whose method name belongs to a seam class
    seam_methods = set()
    for n in ("AttentionBackend", "RadixAttention", "KVCache", "MHATokenToKVPool", "ReqToTokenPool",
              "TokenToKVPoolAllocator", "ScheduleBatch", "Req", "GroupCoordinator", "RadixCache", "ChunkCache", "ForwardBatch",
              "TpModelWorker", "TpModelWorkerClient"):
        seam_methods |= {m for m in surface["methods"][n] if m != "__init__"}
    seam_methods |= set(surface["functions"])
    callers = ["scheduler/scheduler.py", "scheduler/schedule_batch.py", "scheduler/schedule_policy.py",
               "managers/tp_worker.py", "managers/tp_worker_client.py",
               "model_executor/model_runner.py", "model_executor/cuda_graph_runner.py", "model_executor/forward_info.py",
               "nn/attention/radix_attention.py", "nn/attention/triton_backend.py", "nn/attention/flashinfer_backend.py",
               "nn/models/llama/llama.py", "nn/models/llama/mllama.py", "nn/layers/linear.py",
               "nn/layers/logits_processor.py", "nn/layers/vocab_parallel_embedding.py", "memory/radix_cache.py",
               "memory/chunk_cache.py"]
    ctor_names = {"RadixAttention", "ScheduleBatch", "ModelWorkerBatch", "ForwardBatch", "ReqToTokenPool",
                  "TokenToKVPoolAllocator", "MHATokenToKVPool", "RadixCache", "ChunkCache", "Req"}

    def receiver_of(node):
        """last name of the receiver expression: self.last_batch.filter_batch -> 'last_batch'"""
        if isinstance(node, ast.Attribute):
            return node.attr
        if isinstance(node, ast.Name):
            return node.id
        if isinstance(node, ast.Call):
            return receiver_of(node.func) + "()"
        if isinstance(node, ast.Subscript):
            return receiver_of(node.value) + "[]"
        return type(node).__name__

    forms = []
    for rel in callers:
        path = os.path.join(_ref_loader.REF, rel)
        if not os.path.exists(path):
            continue
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            f = node.func
            if isinstance(f, ast.Attribute) and f.attr in seam_methods:
                recv, meth = receiver_of(f.value), f.attr
            elif isinstance(f, ast.Name) and (f.id in ctor_names or f.id in surface["functions"]):
                recv, meth = None, f.id
            else:
                continue
            forms.append({"file": rel, "line": node.lineno, "receiver": recv, "method": meth,
                          "n_positional": sum(1 for a in node.args if not isinstance(a, ast.Starred)),
                          "star_args": any(isinstance(a, ast.Starred) for a in node.args),
                          "keywords": [k.arg for k in node.keywords if k.arg is not None],
                          "star_kwargs": any(k.arg is None for k in node.keywords)})
    forms.sort(key=lambda d: (d["file"], d["line"], d["method"]))
    surface["call_forms"] = forms
    path = os.path.join(HERE, "api_surface.json")
    with open(path, "w") as fh:
        json.dump(surface, fh, indent=1, sort_keys=False)
        fh.write("\n")
    print(f"wrote {path}: {sum(len(v) for v in surface['methods'].values())} methods, {len(forms)} call forms")



GENERATORS = {
    "rmsnorm": gen_rmsnorm, "silu_mul": gen_silu_mul, "rotary": gen_rotary, "kv_pool": gen_kv_pool,
    "positions": gen_positions, "decode_attention": gen_decode, "extend_attention": gen_extend,
    "tiny_llama": gen_tiny_llama, "tiny_mllama": gen_tiny_mllama, "radix_cache": gen_radix_cache,
    "sampling": gen_sampling, "mllama_vision": gen_mllama_vision, "input_logprobs": gen_input_logprobs,
    "prefill_attention": gen_prefill_attention, "api_surface": gen_api_surface,
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
