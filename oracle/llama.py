"""CPU oracle: Llama decoder forward, op-for-op as the reference hosts it.

TEST INFRASTRUCTURE ONLY (see ``oracle/ops.py`` header).  Parity status: PINNED by
``tests/golden/tiny_llama.npz`` (logits of the reference's own ``LlamaForCausalLM`` run on CPU).

Follows nn/models/llama/llama.py: LlamaMLP.forward 62-66, LlamaAttention.forward 143-154,
LlamaDecoderLayer.forward 202-224 (residual protocol: first layer residual=None), LlamaModel
.forward 251-272, LlamaForCausalLM.forward 297-311; logits pruning + lm_head matmul:
nn/layers/logits_processor.py:179-203, 344-376.  Weight names are the reference's
``state_dict`` names (qkv_proj / gate_up_proj are the merged projections, linear.py:696-760).
"""
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch

from . import ops


@dataclass
class LlamaShape:
    hidden: int
    inter: int
    layers: int
    Hq: int
    Hkv: int
    vocab: int
    tie: bool = False
    rope_theta: float = 500000.0
    rope_scaling: Optional[Sequence[float]] = None   # (factor, low, high, orig_max) = llama3
    max_pos: int = 8192
    rms_eps: float = 1e-5

    @property
    def D(self):
        return self.hidden // self.Hq


class OracleKV:
    """Per-layer [P+1, Hkv, D] K/V buffers + req_to_token table (memory/pool.py:13-73, 258-424)."""

    def __init__(self, shape: LlamaShape, slots: int, req_rows: int, ctx: int, dtype=torch.float32):
        self.k = [torch.zeros(slots + 1, shape.Hkv, shape.D, dtype=dtype) for _ in range(shape.layers)]
        self.v = [torch.zeros(slots + 1, shape.Hkv, shape.D, dtype=dtype) for _ in range(shape.layers)]
        self.req_to_token = torch.zeros(req_rows, ctx, dtype=torch.int32)


def forward(shape: LlamaShape, w: Dict[str, torch.Tensor], kv: OracleKV, *, mode: str,
            input_ids: torch.Tensor, positions: torch.Tensor, req_pool_indices: torch.Tensor,
            seq_lens: torch.Tensor, out_cache_loc: torch.Tensor,
            extend_seq_lens: Optional[torch.Tensor] = None,
            extend_start_loc: Optional[torch.Tensor] = None,
            cos_sin_cache: Optional[torch.Tensor] = None, all_hidden: bool = False) -> torch.Tensor:
    """One forward pass; returns next-token logits [bs, vocab] fp32 (``all_hidden``: the final-norm hidden states
    of EVERY token instead - what the logits processor is handed, llama.py forward -> logits_processor.py:148).  ``mode`` is "extend" or
    "decode".  ``kv.req_to_token`` must already hold this step's slots (the scheduler writes
    them before the forward: schedule_batch.py:1046-1054, 1306-1308)."""
    dtype = w["model.embed_tokens.weight"].dtype
    D, Hq, Hkv = shape.D, shape.Hq, shape.Hkv
    if cos_sin_cache is None:
        cos_sin_cache = ops.rope_cos_sin_cache(shape.max_pos, shape.rope_theta, D,
                                               shape.rope_scaling, dtype)
    scale = D ** -0.5
    h = torch.nn.functional.embedding(input_ids, w["model.embed_tokens.weight"])
    residual = None
    for i in range(shape.layers):
        p = f"model.layers.{i}."
        if residual is None:
            residual = h
            h = ops.rmsnorm(h, w[p + "input_layernorm.weight"], shape.rms_eps)
        else:
            h, residual = ops.rmsnorm(h, w[p + "input_layernorm.weight"], shape.rms_eps, residual)
        qkv = torch.nn.functional.linear(h, w[p + "self_attn.qkv_proj.weight"])
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        q, k = ops.rotary_embedding(positions, q, k, D, cos_sin_cache, True)
        k3, v3 = k.reshape(-1, Hkv, D), v.reshape(-1, Hkv, D)
        ops.kv_store(kv.k[i], kv.v[i], out_cache_loc, k3, v3)
        q3 = q.reshape(-1, Hq, D)
        if mode == "decode":
            a = ops.decode_attention(q3, kv.k[i], kv.v[i], kv.req_to_token, req_pool_indices,
                                     seq_lens, scale)
        else:
            a = ops.extend_attention(q3, kv.k[i], kv.v[i], kv.req_to_token, req_pool_indices,
                                     seq_lens, extend_seq_lens, extend_start_loc, scale)
        h = torch.nn.functional.linear(a.reshape(-1, Hq * D), w[p + "self_attn.o_proj.weight"])
        h, residual = ops.rmsnorm(h, w[p + "post_attention_layernorm.weight"], shape.rms_eps, residual)
        gu = torch.nn.functional.linear(h, w[p + "mlp.gate_up_proj.weight"])
        h = torch.nn.functional.linear(ops.silu_and_mul(gu), w[p + "mlp.down_proj.weight"])
    h, _ = ops.rmsnorm(h, w["model.norm.weight"], shape.rms_eps, residual)
    if all_hidden:
        return h
    if mode == "extend":
        last = torch.cumsum(extend_seq_lens.long(), 0) - 1
        h = h[last]
    head = w["model.embed_tokens.weight"] if shape.tie else w["lm_head.weight"]
    return torch.matmul(h.to(head.dtype), head.T)[:, : shape.vocab].float()
