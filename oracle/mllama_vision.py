"""CPU oracle for the Mllama vision tower (TEST INFRASTRUCTURE ONLY - see oracle/ops.py header).

Functional fp32 restatement of nn/models/llama/mllama.py (paths relative to
``/root/reference/scratchpad``): ColumnParallelConv2dPatch 41-76, the precomputed aspect-ratio /
position embeddings 79-145, MllamaVisionMLP 148-178, MllamaVisionEncoderLayer 181-238,
MllamaVisionEncoder 241-280, MllamaVisionModel.forward 339-466, and of the attention it calls:
nn/attention/vision.py VisionAttention 98-180 with the SDPA back-end 257-321 and the additive
tile mask of transformers' ``_prepare_aspect_ratio_attention_mask`` (a (query, key) pair is masked
iff BOTH positions are padding).

Pinned by tests/golden/mllama_vision.npz = the reference's own MllamaVisionModel run on CPU
(tests/golden/gen_golden.py::gen_mllama_vision); weights are addressed by the reference's
parameter names.
"""
import math
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F


class VisionShape:
    def __init__(self, hidden, heads, inter, layers, global_layers, image_size, patch_size, max_num_tiles,
                 max_aspect_ratio_id, channels, intermediate_layers_indices: Sequence[int], norm_eps=1e-5):
        self.hidden, self.heads, self.inter = int(hidden), int(heads), int(inter)
        self.layers, self.global_layers = int(layers), int(global_layers)
        self.image_size, self.patch_size = int(image_size), int(patch_size)
        self.max_num_tiles, self.max_aspect_ratio_id = int(max_num_tiles), int(max_aspect_ratio_id)
        self.channels = int(channels)
        self.intermediate_layers_indices = [int(i) for i in intermediate_layers_indices]
        self.norm_eps = norm_eps
        self.num_patches = (self.image_size // self.patch_size) ** 2 + 1

    @classmethod
    def from_fixture(cls, g):
        return cls(*[int(x) for x in g["cfg"]], intermediate_layers_indices=g["intermediate_layers_indices"])


def masked_attention(q, k, v, pad_rows, scale):
    """q,k,v [b, s, H, D]; pad_rows [b, s] bool.  softmax over keys j with not(pad_i and pad_j)."""
    logits = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    both = pad_rows[:, None, :, None] & pad_rows[:, None, None, :]
    logits = logits.masked_fill(both, float("-inf"))
    return torch.einsum("bhij,bjhd->bihd", torch.softmax(logits, dim=-1), v)


def vision_attention(x, w: Dict[str, torch.Tensor], prefix: str, heads: int, pad_rows=None,
                     cu_seqlens: List[int] = None):
    """VisionAttention.forward, use_qkv_parallel=True (vision.py:110-117, 156-166).  With
    ``cu_seqlens`` the rows of the single batch entry are independent sequences (the
    VisionTritonAttention / context_attention_fwd form, vision.py:334-364)."""
    b, s, E = x.shape
    D = E // heads
    qkv = F.linear(x, w[prefix + "qkv_proj.weight"], w.get(prefix + "qkv_proj.bias"))
    q, k, v = (t.reshape(b, s, heads, D) for t in qkv.chunk(3, dim=-1))
    if cu_seqlens is not None:
        out = torch.zeros_like(q)
        flat = lambda t: t.reshape(1, b * s, heads, D)
        qf, kf, vf, of = flat(q), flat(k), flat(v), flat(out)
        none = torch.zeros(1, 1, dtype=torch.bool)
        for lo, hi in zip(cu_seqlens[:-1], cu_seqlens[1:]):
            if hi > lo:
                of[:, lo:hi] = masked_attention(qf[:, lo:hi], kf[:, lo:hi], vf[:, lo:hi],
                                                none.expand(1, hi - lo), D ** -0.5)
        out = of.reshape(b, s, heads, D)
    else:
        pad = pad_rows if pad_rows is not None else torch.zeros(b, s, dtype=torch.bool)
        out = masked_attention(q, k, v, pad, D ** -0.5)
    return F.linear(out.reshape(b, s, E), w[prefix + "proj.weight"], w.get(prefix + "proj.bias"))


def encoder_layer(x, w, prefix, sh: VisionShape, gated: bool, pad_rows):
    ln = lambda t, name: F.layer_norm(t, (sh.hidden,), w[prefix + name + ".weight"], w[prefix + name + ".bias"],
                                      sh.norm_eps)
    h = vision_attention(ln(x, "input_layernorm"), w, prefix + "self_attn.", sh.heads, pad_rows)
    x = x + (torch.tanh(w[prefix + "gate_attn"]) * h if gated else h)
    h = ln(x, "post_attention_layernorm")
    h = F.linear(F.gelu(F.linear(h, w[prefix + "mlp.fc1.weight"], w[prefix + "mlp.fc1.bias"])),
                 w[prefix + "mlp.fc2.weight"], w[prefix + "mlp.fc2.bias"])
    return x + (torch.tanh(w[prefix + "gate_ffn"]) * h if gated else h)


def pad_rows_of(aspect_ratio_mask: torch.Tensor, num_patches: int, padded: int) -> torch.Tensor:
    """[BM, T] tile validity -> [BM, T * padded] bool, True = padding position (an unused tile, or
    one of the padded - num_patches filler patches of a used tile)."""
    BM, T = aspect_ratio_mask.shape
    real = aspect_ratio_mask.bool().view(BM, T, 1).expand(BM, T, padded).clone()
    real[:, :, num_patches:] = False
    return ~real.reshape(BM, T * padded)


def forward(sh: VisionShape, w: Dict[str, torch.Tensor], pixel_values, aspect_ratio_ids, aspect_ratio_mask):
    w = {k: v.float() for k, v in w.items()}
    B, M, T, C, H, W = pixel_values.shape
    BM, E, P = B * M, sh.hidden, sh.num_patches
    ids = aspect_ratio_ids.reshape(BM)
    x = F.unfold(pixel_values.reshape(BM * T, C, H, W).float(), kernel_size=sh.patch_size, stride=sh.patch_size)
    x = F.linear(x.permute(0, 2, 1), w["patch_embedding._linear.weight"])             # [BM*T, P-1, E]
    x = x.reshape(BM, T, P - 1, E)
    x = x + (w["pre_tile_positional_embedding.embedding.weight"][ids].reshape(BM, sh.max_num_tiles, 1, E)
             * torch.tanh(w["pre_tile_positional_embedding.gate"]))
    x = torch.cat([w["class_embedding"].expand(BM, T, 1, E), x], dim=2)               # [BM, T, P, E]
    gate = torch.tanh(w["gated_positional_embedding.gate"])
    x = x + (1 - gate) * w["gated_positional_embedding.embedding"].view(1, 1, P, E)
    x = x + gate * w["gated_positional_embedding.tile_embedding.weight"][ids].reshape(BM, sh.max_num_tiles, P, E)
    x = F.layer_norm(x, (E,), w["layernorm_pre.weight"], w["layernorm_pre.bias"], 1e-5)
    Pp = P + (8 - P % 8) % 8
    x = F.pad(x, (0, 0, 0, Pp - P))
    pad_rows = pad_rows_of(aspect_ratio_mask.reshape(BM, T), P, Pp)
    x = x.reshape(BM, T * Pp, E)
    inter = []
    for i in range(sh.layers):
        if i in sh.intermediate_layers_indices:
            inter.append(x)
        x = encoder_layer(x, w, f"transformer.layers.{i}.", sh, False, pad_rows)
    if sh.layers - 1 in sh.intermediate_layers_indices:
        inter.append(x)
    x = F.layer_norm(x, (E,), w["layernorm_post.weight"], w["layernorm_post.bias"], 1e-5)
    x = x.reshape(BM, T, Pp, E)
    x = x + (w["post_tile_positional_embedding.embedding.weight"][ids].reshape(BM, sh.max_num_tiles, 1, E)
             * torch.tanh(w["post_tile_positional_embedding.gate"]))
    x = x.reshape(BM, T * Pp, E)
    for i in range(sh.global_layers):
        x = encoder_layer(x, w, f"global_transformer.layers.{i}.", sh, True, pad_rows)
    x = x.reshape(BM, T, Pp, E)[:, :, :P]
    inter = torch.stack(inter, dim=-1).reshape(BM, T, Pp, -1)[:, :, :P]
    return torch.cat([x, inter], dim=-1).reshape(B, M, T, P, -1)
