"""CPU oracle: plain-torch restatement of the reference's hot-path operators.

TEST INFRASTRUCTURE ONLY.  Nothing under ``scratchpad_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline - never as the thing shipped.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against vectors produced by running the reference's own code in the build container
(``tests/golden/gen_golden.py``: ``forward_native`` methods, ``memory/pool.py`` classes and the
in-tree Triton kernels under the Triton interpreter).  The reference itself ships no golden
vectors for this path, so those generated fixtures are the pin.

All paths below are relative to ``/root/reference/scratchpad``.  Arithmetic is done in fp32 on
inputs of any float dtype, with the reference's rounding points reproduced where the
reference's torch path has them.
"""
import math
from typing import Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------- norm / act
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float,
            residual: Optional[torch.Tensor] = None):
    """RMSNorm.forward_native - nn/layers/layernorm.py:34-51.

    Without residual: returns y.  With residual: returns (y, residual') where
    residual' = (x + residual) rounded to x.dtype and y = norm(fp32 sum) (the fp32 sum, NOT the
    rounded residual', is what gets normalised - layernorm.py:41-47).
    """
    orig = x.dtype
    xf = x.to(torch.float32)
    if residual is not None:
        xf = xf + residual.to(torch.float32)
        residual = xf.to(orig)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    y = xf.to(orig) * weight
    return y if residual is None else (y, residual)


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """SiluAndMul.forward_native - nn/layers/activation.py:22-24 (F.silu(x[:d]) * x[d:])."""
    d = x.shape[-1] // 2
    return torch.nn.functional.silu(x[..., :d]) * x[..., d:]


# --------------------------------------------------------------------------- rotary
def rope_inv_freq(base: float, rotary_dim: int,
                  llama3: Optional[Sequence[float]] = None) -> torch.Tensor:
    """RotaryEmbedding._compute_inv_freq - nn/layers/rotary_embedding.py:77-90, and
    Llama3RotaryEmbedding._compute_inv_freq - 698-720.
    ``llama3`` = (factor, low_freq_factor, high_freq_factor, original_max_position)."""
    inv = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    if llama3 is None:
        return inv
    factor, low, high, orig_max = llama3
    low_wavelen = orig_max / low
    high_wavelen = orig_max / high
    wave_len = 2 * math.pi / inv
    if low != high:
        smooth = (orig_max / wave_len - low) / (high - low)
    else:
        smooth = 0
    return torch.where(
        wave_len < high_wavelen, inv,
        torch.where(wave_len > low_wavelen, inv / factor,
                    (1 - smooth) * inv / factor + smooth * inv))


def rope_cos_sin_cache(max_position: int, base: float, rotary_dim: int,
                       llama3: Optional[Sequence[float]] = None,
                       dtype: torch.dtype = torch.float32, scaling: Optional[dict] = None) -> torch.Tensor:
    """RotaryEmbedding._compute_cos_sin_cache - rotary_embedding.py:92-101, cast to the model
    dtype as __init__ does (72-75).  Layout [max_position, rotary_dim] = cos || sin.
    ``scaling``: a get_rope dict of type "linear" / "dynamic" / "yarn" (rotary_embedding.py:996-1036 ->
    LinearScalingRotaryEmbedding 214-245 with ONE factor, DynamicNTKScalingRotaryEmbedding 281-299,
    YaRNScalingRotaryEmbedding 373-414): the table then covers context * factor positions."""
    if scaling is None:
        inv = rope_inv_freq(base, rotary_dim, llama3)
        t = torch.arange(max_position, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)
    kind, factor = scaling["rope_type"], scaling["factor"]
    mscale = 1.0
    if kind == "linear":                                   # 232-236: t / factor over max_position * factor entries
        t = torch.arange(max_position * factor, dtype=torch.float) / factor
        inv = rope_inv_freq(base, rotary_dim)
    elif kind == "dynamic":                                # 286-292: the NTK base at the extended length
        n = max_position * factor
        ntk = base * ((factor * n / max_position) - (factor - 1)) ** (rotary_dim / (rotary_dim - 2))
        t = torch.arange(n, dtype=torch.float)
        inv = rope_inv_freq(ntk, rotary_dim)
    elif kind == "yarn":                                   # 302-345, 382-414; the context is the ORIGINAL one (1018-1036)
        ctx = scaling["original_max_position_embeddings"]
        fast, slow = scaling.get("beta_fast", 32), scaling.get("beta_slow", 1)
        pos_freqs = base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim)
        dim_of = lambda rot: (rotary_dim * math.log(ctx / (rot * 2 * math.pi))) / (2 * math.log(base))
        low, high = max(math.floor(dim_of(fast)), 0), min(math.ceil(dim_of(slow)), rotary_dim - 1)
        if low == high:
            high += 0.001
        ramp = torch.clamp((torch.arange(rotary_dim // 2, dtype=torch.float) - low) / (high - low), 0, 1)
        mask = (1 - ramp) * scaling.get("extrapolation_factor", 1)
        inv = 1.0 / (factor * pos_freqs) * (1 - mask) + 1.0 / pos_freqs * mask
        mscale = float((0.1 * math.log(factor) + 1.0 if factor > 1 else 1.0) * scaling.get("attn_factor", 1))
        t = torch.arange(ctx * factor, dtype=torch.float32)
    else:
        raise ValueError(kind)
    freqs = torch.einsum("i,j -> ij", t, inv)
    if kind == "yarn":
        return torch.cat((freqs.cos() * mscale, freqs.sin() * mscale), dim=-1).to(dtype)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                     head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool = True
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """RotaryEmbedding.forward_native + _apply_rotary_emb - rotary_embedding.py:23-49, 102-130.
    rotary_dim = cos_sin_cache.shape[1]; dims beyond it pass through.  Returns new (q, k).
    cos/sin are cast to the activation dtype before the multiply (37-38)."""
    rot = cos_sin_cache.shape[-1]
    positions = positions.flatten()
    cs = cos_sin_cache.index_select(0, positions)
    cos, sin = cs.chunk(2, dim=-1)

    def apply(x):
        shape = x.shape
        x = x.view(*shape[:-1], -1, head_size)
        xr, xp = x[..., :rot], x[..., rot:]
        c = cos.unsqueeze(-2).to(x.dtype)
        s = sin.unsqueeze(-2).to(x.dtype)
        if is_neox:
            x1, x2 = torch.chunk(xr, 2, dim=-1)
        else:
            x1, x2 = xr[..., ::2], xr[..., 1::2]
        o1 = x1 * c - x2 * s
        o2 = x2 * c + x1 * s
        if is_neox:
            o = torch.cat((o1, o2), dim=-1)
        else:
            o = torch.stack((o1, o2), dim=-1).flatten(-2)
        return torch.cat((o, xp), dim=-1).reshape(shape)

    return apply(query), apply(key)


# --------------------------------------------------------------------------- KV pool / indexing
def kv_store(k_buffer: torch.Tensor, v_buffer: torch.Tensor, loc: torch.Tensor,
             cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
    """MHATokenToKVPool.set_kv_buffer - memory/pool.py:392-424 (non-fp8 branch): in-place
    ``buffer[loc] = cache``.  Duplicate slots (padded rows all write slot 0): last writer wins
    in the reference's index_put; callers must not rely on slot 0's content."""
    if k_buffer.dtype == torch.uint8:
        # --kv-cache-dtype fp8_e5m2 (memory/pool.py:274-280, 401-412): cache.to(float8_e5m2), stored
        # as uint8 because index_put has no fp8 kernel
        k_buffer[loc] = cache_k.to(torch.float8_e5m2).view(torch.uint8)
        v_buffer[loc] = cache_v.to(torch.float8_e5m2).view(torch.uint8)
        return
    k_buffer[loc] = cache_k.to(k_buffer.dtype)
    v_buffer[loc] = cache_v.to(v_buffer.dtype)


def kv_rows(buffer: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Rows of a KV pool as fp32; a uint8 pool holds fp8 e5m2 bytes (get_key_buffer's
    ``.view(self.dtype)``, pool.py:366-371), widened exactly - flashinfer's treatment of fp8 KV
    (convert to the query type, fp32 accumulate); the in-tree Triton kernels' rounding of P to
    the KV dtype (``p.to(v.dtype)``) is not applied for fp8."""
    rows = buffer[idx]
    if rows.dtype == torch.uint8:
        rows = rows.view(torch.float8_e5m2)
    return rows.to(torch.float32)


def write_req_to_token(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                       pre_lens: torch.Tensor, seq_lens: torch.Tensor,
                       extend_lens: torch.Tensor, out_cache_loc: torch.Tensor) -> None:
    """write_req_to_token_pool_triton - scheduler/schedule_batch.py:1546-1580:
    req_to_token[req, pre:seq] = out_cache_loc[cumsum(extend_lens)[:b] + (0..seq-pre)]."""
    start = 0
    for b in range(req_pool_indices.shape[0]):
        pre, seq = int(pre_lens[b]), int(seq_lens[b])
        n = seq - pre
        req_to_token[int(req_pool_indices[b]), pre:seq] = out_cache_loc[start:start + n].to(
            req_to_token.dtype)
        start += int(extend_lens[b])


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor):
    """compute_position_torch - model_executor/forward_info.py:452-466.
    positions int64 [sum(extend)], extend_start_loc = exclusive cumsum (dtype of extend_seq_lens)."""
    pos = [torch.arange(int(p), int(p) + int(e)) for p, e in zip(extend_prefix_lens, extend_seq_lens)]
    positions = torch.cat(pos) if pos else torch.zeros(0, dtype=torch.int64)
    start = torch.zeros_like(extend_seq_lens)
    start[1:] = torch.cumsum(extend_seq_lens[:-1], dim=0)
    return positions.to(torch.int64), start


def clamp_position(seq_lens: torch.Tensor) -> torch.Tensor:
    """clamp_position - forward_info.py:469-471."""
    return torch.clamp(seq_lens - 1, min=0).to(torch.int64)


# --------------------------------------------------------------------------- attention
def _softcap(s: torch.Tensor, cap: float) -> torch.Tensor:
    # decode_attention.py:343-344 / extend_attention.py:136-137: cap * tanh(s / cap)
    return cap * torch.tanh(s / cap) if cap > 0 else s


def _softmax_parts(s: torch.Tensor, dtype: torch.dtype):
    """exp(s - rowmax) and its fp32 row sum.  As in the reference's kernels the probabilities are
    cast to the KV dtype before the P.V product while the denominator keeps the unrounded fp32 sum
    (decode_attention.py:421-428 `e_sum += sum(p); p = p.to(v.dtype)`; extend_attention.py:150-156).
    A no-op for fp32 inputs."""
    m = s.amax(dim=-1, keepdim=True)
    p = torch.exp(s - m)
    denom = p.sum(dim=-1, keepdim=True)
    if dtype in (torch.float16, torch.bfloat16):
        p = p.to(dtype).to(torch.float32)
    return p, denom


def decode_attention(q: torch.Tensor, k_buffer: torch.Tensor, v_buffer: torch.Tensor,
                     req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                     seq_lens: torch.Tensor, sm_scale: float, logit_cap: float = 0.0,
                     kv_start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """decode_attention_fwd - nn/attention/triton_attn/decode_attention.py:547-608 (stage 1
    250-354: logits = q.K[idx]^T * scale [soft-capped]; stage 2 357-434: softmax . V[idx]).

    q [bs, Hq, D]; buffers [P+1, Hkv, D]; idx = req_to_token[req_pool_indices[b],
    kv_start[b] : kv_start[b] + seq_lens[b]].  fp32 math, output in q.dtype.  A row with
    seq_len 0 yields zeros (the reference leaves it unwritten / NaN; callers never read it).
    ``kv_start`` (default 0) is the encoder offset used by cross/self attention on
    encoder-decoder models (flashinfer_backend.py:593-621)."""
    bs, Hq, D = q.shape
    Hkv = k_buffer.shape[1]
    Dv = v_buffer.shape[2]
    g = Hq // Hkv
    o = torch.zeros(bs, Hq, Dv, dtype=torch.float32)
    for b in range(bs):
        L = int(seq_lens[b])
        if L == 0:
            continue
        s0 = 0 if kv_start is None else int(kv_start[b])
        idx = req_to_token[int(req_pool_indices[b]), s0:s0 + L].long()
        k = kv_rows(k_buffer, idx)                   # [L, Hkv, D]
        v = kv_rows(v_buffer, idx)
        qb = q[b].to(torch.float32).view(Hkv, g, D)
        s = torch.einsum("hgd,lhd->hgl", qb, k) * sm_scale
        s = _softcap(s, logit_cap)
        p, denom = _softmax_parts(s, q.dtype)
        o[b] = (torch.einsum("hgl,lhd->hgd", p, v) / denom).reshape(Hq, Dv)
    return o.to(q.dtype)


def extend_attention(q: torch.Tensor, k_buffer: torch.Tensor, v_buffer: torch.Tensor,
                     req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                     seq_lens: torch.Tensor, extend_seq_lens: torch.Tensor,
                     extend_start_loc: torch.Tensor, sm_scale: float, logit_cap: float = 0.0,
                     k_extend: Optional[torch.Tensor] = None,
                     v_extend: Optional[torch.Tensor] = None,
                     causal: bool = True,
                     kv_start: Optional[torch.Tensor] = None,
                     window_left: int = -1) -> torch.Tensor:
    """extend_attention_fwd - nn/attention/triton_attn/extend_attention.py:16-327.

    Row t (0-based within the request's new tokens) of request b attends to the cached prefix
    (paged: req_to_token[req, :prefix]) and to new tokens <= t.  If ``k_extend`` is None the new
    tokens' K/V are read from the pool as well (they were stored before the call,
    triton_backend.py:131-134), which is numerically identical.  ``causal=False`` +
    ``kv_start`` give the encoder-decoder cross-attention form
    (flashinfer_backend.py:400-417): every row attends to kv [kv_start, kv_start+seq_len).
    ``window_left >= 0`` is flashinfer's sliding window (third-party v0.2.3, absent here; call site
    flashinfer_backend.py:408-417, documented semantics: the row at kv position p attends to
    [p - window_left, p]) - PARITY UNPINNED by the reference, which holds no test of it."""
    T, Hq, D = q.shape
    Hkv = k_buffer.shape[1]
    Dv = v_buffer.shape[2]
    g = Hq // Hkv
    o = torch.zeros(T, Hq, Dv, dtype=torch.float32)
    for b in range(req_pool_indices.shape[0]):
        L, E = int(seq_lens[b]), int(extend_seq_lens[b])
        if E == 0 or L == 0:   # no new rows / no visible keys (text-only row of cross-attention): zeros
            continue
        s0 = int(extend_start_loc[b])
        off = 0 if kv_start is None else int(kv_start[b])
        P = L - E if causal else L
        idx = req_to_token[int(req_pool_indices[b]), off:off + L].long()
        k = kv_rows(k_buffer, idx)
        v = kv_rows(v_buffer, idx)
        if causal and k_extend is not None:
            k = torch.cat((k[:P], k_extend[s0:s0 + E].to(torch.float32)), 0)
            v = torch.cat((v[:P], v_extend[s0:s0 + E].to(torch.float32)), 0)
        qb = q[s0:s0 + E].to(torch.float32).view(E, Hkv, g, D)
        s = torch.einsum("ehgd,lhd->hgel", qb, k) * sm_scale
        s = _softcap(s, logit_cap)
        if causal:
            col = torch.arange(L).view(1, L)
            row = torch.arange(E).view(E, 1) + P
            s = s.masked_fill(col > row, float("-inf"))
            if window_left >= 0:
                s = s.masked_fill(col < row - window_left, float("-inf"))
        p, denom = _softmax_parts(s, q.dtype)
        o[s0:s0 + E] = (torch.einsum("hgel,lhd->hged", p, v) / denom).permute(2, 0, 1, 3).reshape(E, Hq, Dv)
    return o.to(q.dtype)


def context_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, b_start_loc: torch.Tensor,
                      b_seq_len: torch.Tensor) -> torch.Tensor:
    """context_attention_fwd - nn/attention/triton_attn/prefill_attention.py:125-163 (kernel 18-122): cache-less
    CAUSAL variable-length self-attention.  Sequence b = rows [b_start_loc[b], + b_seq_len[b]) of q / k / v
    ([tokens, heads, D], GQA by head // group); sm_scale = 1 / sqrt(D) is fixed inside the function (131);
    probabilities are cast to V's dtype before P.V while the denominator keeps the fp32 sum (98-112).
    Pinned by tests/golden/prefill_attention.npz (the reference's kernel under the Triton interpreter)."""
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    g = Hq // Hkv
    o = torch.zeros(T, Hq, v.shape[2], dtype=torch.float32)
    for b in range(b_seq_len.shape[0]):
        s0, L = int(b_start_loc[b]), int(b_seq_len[b])
        if L == 0:
            continue
        qb = q[s0:s0 + L].to(torch.float32).view(L, Hkv, g, D)
        kb = k[s0:s0 + L].to(torch.float32)
        vb = v[s0:s0 + L].to(torch.float32)
        s = torch.einsum("ehgd,lhd->hgel", qb, kb) * (1.0 / math.sqrt(D))
        col = torch.arange(L).view(1, L)
        row = torch.arange(L).view(L, 1)
        s = s.masked_fill(col > row, float("-inf"))
        p, denom = _softmax_parts(s, q.dtype)
        o[s0:s0 + L] = (torch.einsum("hgel,lhd->hged", p, vb) / denom).permute(2, 0, 1, 3).reshape(L, Hq, -1)
    return o.to(q.dtype)


def merge_state(o1: torch.Tensor, lse1: torch.Tensor, o2: torch.Tensor, lse2: torch.Tensor):
    """flashinfer.cascade.merge_state (v0.2.3, third-party, absent here; call site
    nn/attention/flashinfer_backend.py:437-439): combine two partial attention results over
    disjoint key sets from their log-sum-exp.  o [T,H,D], lse [T,H] (natural log)."""
    m = torch.maximum(lse1, lse2)
    w1 = torch.exp(lse1 - m)
    w2 = torch.exp(lse2 - m)
    o = (o1.float() * w1.unsqueeze(-1) + o2.float() * w2.unsqueeze(-1)) / (w1 + w2).unsqueeze(-1)
    return o.to(o1.dtype), m + torch.log(w1 + w2)
