"""CPU oracle: Mllama text model (self-attention + tanh-gated cross-attention layers).

TEST INFRASTRUCTURE ONLY (see ``oracle/ops.py``).  Parity status: the layer glue is PINNED by
``tests/golden/tiny_mllama.npz`` (the reference's MllamaForCausalLM run on CPU); the cross-attention
call itself follows flashinfer's documented semantics (flashinfer is absent here), so attention
numerics for the encoder window are pinned only through ``ops.extend_attention`` /
``ops.decode_attention`` with ``causal=False`` / ``kv_start``.

Follows nn/models/llama/mllama.py: MllamaTextRMSNorm 469-483, MllamaTextCrossAttention.forward
543-571, MllamaCrossAttentionDecoderLayer.forward 609-634, MllamaTextModel.forward 681-716."""
from typing import Dict, List, Optional

import torch

from . import ops
from .llama import LlamaShape, OracleKV


def _linear(x, w):
    return torch.nn.functional.linear(x, w)


def forward(shape: LlamaShape, cross_layers: List[int], w: Dict[str, torch.Tensor], kv: OracleKV, *,
            mode: str, input_ids, positions, req_pool_indices, seq_lens, out_cache_loc, encoder_lens,
            row_mask, extend_seq_lens=None, extend_start_loc=None, cross_attention_states=None,
            encoder_out_cache_loc=None) -> torch.Tensor:
    dtype = w["model.embed_tokens.weight"].dtype
    D, Hq, Hkv = shape.D, shape.Hq, shape.Hkv
    cos_sin = ops.rope_cos_sin_cache(shape.max_pos, shape.rope_theta, D, shape.rope_scaling, dtype)
    scale = D ** -0.5
    mask = row_mask.to(dtype)
    h = torch.nn.functional.embedding(input_ids, w["model.embed_tokens.weight"])
    for i in range(shape.layers):
        p = f"model.layers.{i}."
        if i in cross_layers:
            residual = h
            x = ops.rmsnorm(h, w[p + "input_layernorm.weight"], shape.rms_eps)
            q = _linear(x, w[p + "cross_attn.qkv_proj.weight"])[:, :Hq * D].reshape(-1, Hq, D)
            q = ops.rmsnorm(q, w[p + "cross_attn.q_norm.weight"], shape.rms_eps)
            if cross_attention_states is not None:
                qkv_e = _linear(cross_attention_states, w[p + "cross_attn.qkv_proj.weight"])
                k = qkv_e[:, Hq * D:(Hq + Hkv) * D].reshape(-1, Hkv, D)
                v = qkv_e[:, (Hq + Hkv) * D:].reshape(-1, Hkv, D)
                k = ops.rmsnorm(k, w[p + "cross_attn.k_norm.weight"], shape.rms_eps)
                ops.kv_store(kv.k[i], kv.v[i], encoder_out_cache_loc, k, v)
            if mode == "decode":
                a = ops.decode_attention(q, kv.k[i], kv.v[i], kv.req_to_token, req_pool_indices,
                                         encoder_lens, scale)
            else:
                a = ops.extend_attention(q, kv.k[i], kv.v[i], kv.req_to_token, req_pool_indices, encoder_lens,
                                         extend_seq_lens, extend_start_loc, scale, causal=False)
            x = _linear(a.reshape(-1, Hq * D), w[p + "cross_attn.o_proj.weight"])
            x = mask * x
            h = residual + w[p + "cross_attn_attn_gate"].tanh() * x
            residual = h
            x = ops.rmsnorm(h, w[p + "post_attention_layernorm.weight"], shape.rms_eps)
            x = _linear(ops.silu_and_mul(_linear(x, w[p + "mlp.gate_up_proj.weight"])), w[p + "mlp.down_proj.weight"])
            x = mask * x
            h = residual + w[p + "cross_attn_mlp_gate"].tanh() * x
        else:
            residual = h
            x = ops.rmsnorm(h, w[p + "input_layernorm.weight"], shape.rms_eps)
            qkv = _linear(x, w[p + "self_attn.qkv_proj.weight"])
            q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
            q, k = ops.rotary_embedding(positions, q, k, D, cos_sin, True)
            ops.kv_store(kv.k[i], kv.v[i], out_cache_loc, k.reshape(-1, Hkv, D), v.reshape(-1, Hkv, D))
            q3 = q.reshape(-1, Hq, D)
            if mode == "decode":
                a = ops.decode_attention(q3, kv.k[i], kv.v[i], kv.req_to_token, req_pool_indices, seq_lens,
                                         scale, kv_start=encoder_lens)
            else:
                a = ops.extend_attention(q3, kv.k[i], kv.v[i], kv.req_to_token, req_pool_indices, seq_lens,
                                         extend_seq_lens, extend_start_loc, scale, kv_start=encoder_lens)
            x = _linear(a.reshape(-1, Hq * D), w[p + "self_attn.o_proj.weight"])
            x, residual = ops.rmsnorm(x, w[p + "post_attention_layernorm.weight"], shape.rms_eps, residual)
            x = _linear(ops.silu_and_mul(_linear(x, w[p + "mlp.gate_up_proj.weight"])), w[p + "mlp.down_proj.weight"])
            h = x + residual
    h = ops.rmsnorm(h, w["model.norm.weight"], shape.rms_eps)
    if mode == "extend":
        h = h[torch.cumsum(extend_seq_lens.long(), 0) - 1]
    return torch.matmul(h.to(dtype), w["lm_head.weight"].T)[:, : shape.vocab].float()
