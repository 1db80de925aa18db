"""Mllama (Llama-3.2-Vision) vision tower on the HIP attention path.

Re-hosts nn/models/llama/mllama.py: ColumnParallelConv2dPatch 41-76, the precomputed aspect-ratio and
position embeddings 79-145, MllamaVisionMLP 148-178, MllamaVisionEncoderLayer 181-238,
MllamaVisionEncoder 241-280 and MllamaVisionModel 283-466, with the reference's parameter names so
a reference state_dict loads by name.

What runs where: the 40 self-attention layers (32 local + 8 global at the 11B size; 4 tiles x 1032
positions x 16 heads of 80) go through ``vision.VisionAttention`` = the MFMA extend kernel,
non-causal, heads zero-padded 80 -> 128 in the weights.  The tile mask is not materialised: the
reference builds a [b, 1, 4128, 4128] additive tensor per forward (mllama.py:403-410); here the same
semantics come from the list of padding positions (vision.py, ``pad_rows``).  LayerNorm, GELU, the
patch unfold and the embedding adds are torch ops; the linears are library GEMMs.

The patch projection is kept replicated on every TP rank (the reference shards it by columns and
all-gathers the result, mllama.py:367-371: same values, one small collective less).
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .vision import ColumnParallelLinear, RowParallelLinear, VisionAttention, VisionAttnPlan


class Conv2dPatch(nn.Module):
    """mllama.py:41-76: non-overlapping patches -> linear.  Input [n, C, H, W] -> [n, patches, out]."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int, dtype=None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self._linear = nn.Linear(in_channels * kernel_size * kernel_size, out_channels, bias=False, dtype=dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        cols = F.unfold(x, kernel_size=self.kernel_size, stride=self.stride)
        return self._linear(cols.transpose(1, 2))


class MllamaPrecomputedAspectRatioEmbedding(nn.Module):
    """mllama.py:79-103"""

    def __init__(self, config, is_gated: bool = True, dtype=None):
        super().__init__()
        self.max_num_tiles, self.hidden_size = config.max_num_tiles, config.hidden_size
        self.is_gated = is_gated
        self.embedding = nn.Embedding(config.max_aspect_ratio_id + 1, self.max_num_tiles * self.hidden_size,
                                      dtype=dtype)
        if is_gated:
            self.gate = nn.Parameter(torch.zeros(1, dtype=dtype))

    def forward(self, hidden_state: torch.Tensor, aspect_ratio_ids: torch.Tensor) -> torch.Tensor:
        emb = self.embedding(aspect_ratio_ids).reshape(-1, self.max_num_tiles, 1, self.hidden_size)
        if self.is_gated:
            emb = emb * self.gate.tanh()
        return hidden_state + emb


class MllamaPrecomputedPositionEmbedding(nn.Module):
    """mllama.py:106-145"""

    def __init__(self, config, dtype=None):
        super().__init__()
        self.max_num_tiles, self.hidden_size = config.max_num_tiles, config.hidden_size
        self.num_patches = (config.image_size // config.patch_size) ** 2 + 1
        self.gate = nn.Parameter(torch.zeros(1, dtype=dtype))
        self.embedding = nn.Parameter(torch.zeros(self.num_patches, self.hidden_size, dtype=dtype))
        self.tile_embedding = nn.Embedding(config.max_aspect_ratio_id + 1,
                                           self.max_num_tiles * self.num_patches * self.hidden_size, dtype=dtype)

    def forward(self, hidden_state: torch.Tensor, aspect_ratio_ids: torch.Tensor) -> torch.Tensor:
        gate = self.gate.tanh()
        hidden_state = hidden_state + (1 - gate) * self.embedding.view(1, 1, self.num_patches, self.hidden_size)
        tile = self.tile_embedding(aspect_ratio_ids).reshape(
            hidden_state.shape[0], self.max_num_tiles, self.num_patches, self.hidden_size)
        return hidden_state + gate * tile


class MllamaVisionMLP(nn.Module):
    """mllama.py:148-178 (fc1 column-parallel, fc2 row-parallel, both biased; exact GELU)."""

    def __init__(self, config, dtype=None):
        super().__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError(f"vision hidden_act {config.hidden_act}")
        self.fc1 = ColumnParallelLinear(config.hidden_size, config.intermediate_size, bias=True, dtype=dtype)
        self.fc2 = RowParallelLinear(config.intermediate_size, config.hidden_size, bias=True, dtype=dtype)

    def forward(self, x):
        x, _ = self.fc1(x)
        x, _ = self.fc2(F.gelu(x))
        return x


class MllamaVisionEncoderLayer(nn.Module):
    """mllama.py:181-238"""

    def __init__(self, config, is_gated: bool = False, dtype=None):
        super().__init__()
        self.is_gated = is_gated
        self.self_attn = VisionAttention(config.hidden_size, config.attention_heads, config.hidden_size,
                                         bias=True, dtype=dtype)
        self.mlp = MllamaVisionMLP(config, dtype)
        self.input_layernorm = nn.LayerNorm(config.hidden_size, eps=config.norm_eps, dtype=dtype)
        self.post_attention_layernorm = nn.LayerNorm(config.hidden_size, eps=config.norm_eps, dtype=dtype)
        if is_gated:
            self.gate_attn = nn.Parameter(torch.ones(1, dtype=dtype) * 0.7853981633974483)
            self.gate_ffn = nn.Parameter(torch.ones(1, dtype=dtype) * 0.7853981633974483)

    def forward(self, hidden_state: torch.Tensor, plan: Optional[VisionAttnPlan] = None):
        h = self.self_attn(self.input_layernorm(hidden_state), plan=plan)
        hidden_state = hidden_state + (self.gate_attn.tanh() * h if self.is_gated else h)
        h = self.mlp(self.post_attention_layernorm(hidden_state))
        return hidden_state + (self.gate_ffn.tanh() * h if self.is_gated else h)


class MllamaVisionEncoder(nn.Module):
    """mllama.py:241-280: returns (last hidden, hidden states entering the requested layers [+ the
    final one when the last layer index is requested])."""

    def __init__(self, config, num_layers: int, is_gated: bool = False, output_hidden_states=None, dtype=None):
        super().__init__()
        self.layers = nn.ModuleList(MllamaVisionEncoderLayer(config, is_gated, dtype) for _ in range(num_layers))
        self.output_hidden_states = list(output_hidden_states or [])

    def forward(self, hidden_states: torch.Tensor, plan: Optional[VisionAttnPlan] = None):
        taps = []
        for i, layer in enumerate(self.layers):
            if i in self.output_hidden_states:
                taps.append(hidden_states)
            hidden_states = layer(hidden_states, plan)
        if len(self.layers) - 1 in self.output_hidden_states:
            taps.append(hidden_states)
        return hidden_states, tuple(taps)


def padding_positions(aspect_ratio_mask: torch.Tensor, num_patches: int, padded_patches: int) -> torch.Tensor:
    """[n_images, tiles] tile validity -> [n_images, tiles * padded_patches] bool on the host, True
    where the position is padding: every position of an unused tile and the filler patches
    [num_patches, padded_patches) of a used tile - the positions _prepare_aspect_ratio_attention_mask
    (called at mllama.py:403-410) marks."""
    n, tiles = aspect_ratio_mask.shape
    real = aspect_ratio_mask.cpu().bool().view(n, tiles, 1).repeat(1, 1, padded_patches)
    real[:, :, num_patches:] = False
    return ~real.reshape(n, tiles * padded_patches)


class MllamaVisionModel(nn.Module):
    """mllama.py:283-466"""

    def __init__(self, config, dtype=None):
        super().__init__()
        self.image_size, self.patch_size = config.image_size, config.patch_size
        self.max_num_tiles, self.hidden_size = config.max_num_tiles, config.hidden_size
        self.num_patches = (self.image_size // self.patch_size) ** 2 + 1
        self.patch_embedding = Conv2dPatch(config.num_channels, self.hidden_size, self.patch_size,
                                           self.patch_size, dtype)
        self.class_embedding = nn.Parameter(torch.zeros(self.hidden_size, dtype=dtype))
        self.gated_positional_embedding = MllamaPrecomputedPositionEmbedding(config, dtype)
        self.pre_tile_positional_embedding = MllamaPrecomputedAspectRatioEmbedding(config, True, dtype)
        self.post_tile_positional_embedding = MllamaPrecomputedAspectRatioEmbedding(config, True, dtype)
        self.layernorm_pre = nn.LayerNorm(self.hidden_size, dtype=dtype)
        self.layernorm_post = nn.LayerNorm(self.hidden_size, dtype=dtype)
        self.transformer = MllamaVisionEncoder(config, config.num_hidden_layers, False,
                                               config.intermediate_layers_indices, dtype)
        self.global_transformer = MllamaVisionEncoder(config, config.num_global_layers, True, None, dtype)

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, aspect_ratio_ids: torch.Tensor,
                aspect_ratio_mask: torch.Tensor) -> torch.Tensor:
        B, M, T, C, H, W = pixel_values.shape
        n, E, P = B * M, self.hidden_size, self.num_patches
        dtype = self.layernorm_pre.weight.dtype
        ids = aspect_ratio_ids.reshape(n, -1).reshape(n)
        x = self.patch_embedding(pixel_values.reshape(n * T, C, H, W).to(dtype)).reshape(n, T, P - 1, E)
        x = self.pre_tile_positional_embedding(x, ids)
        x = torch.cat([self.class_embedding.expand(n, T, 1, E), x], dim=2)
        x = self.gated_positional_embedding(x, ids)
        x = self.layernorm_pre(x)
        Pp = P + (8 - P % 8) % 8                      # positions per tile, padded to a multiple of 8
        if Pp == P:
            raise NotImplementedError("patch count already a multiple of 8: the reference's mask marks "
                                      "every position as padding in that case (mllama.py:403-410)")
        x = F.pad(x, (0, 0, 0, Pp - P))
        # one attention plan for all 40 layers (the reference rebuilds its [n,1,S,S] mask per forward too)
        plan = VisionAttnPlan(n, T * Pp, x.device, pad_rows=padding_positions(aspect_ratio_mask.reshape(n, T), P, Pp))
        x, taps = self.transformer(x.reshape(n, T * Pp, E), plan)
        x = self.layernorm_post(x).reshape(n, T, Pp, E)
        x = self.post_tile_positional_embedding(x, ids).reshape(n, T * Pp, E)
        x, _ = self.global_transformer(x, plan)
        x = x.reshape(n, T, Pp, E)[:, :, :P]
        taps = torch.stack(taps, dim=-1).reshape(n, T, Pp, -1)[:, :, :P]
        return torch.cat([x, taps], dim=-1).reshape(B, M, T, P, -1)

    def load_full_state_dict(self, full: Dict[str, torch.Tensor]) -> None:
        """tp=1 state_dict of the reference's MllamaVisionModel, by parameter name."""
        own = dict(self.named_parameters())
        mods = dict(self.named_modules())
        done = set()
        for name, mod in mods.items():
            if isinstance(mod, VisionAttention):
                p = name + "."
                mod.load_reference_weights(full[p + "qkv_proj.weight"], full.get(p + "qkv_proj.bias"),
                                           full[p + "proj.weight"], full.get(p + "proj.bias"))
                done.update(k for k in own if k.startswith(p))
        for name, param in own.items():
            if name in done:
                continue
            src = full[name]
            mod = mods[name.rsplit(".", 1)[0]] if "." in name else None
            if mod is not None and hasattr(mod, "shard_from_full"):
                src = mod.shard_from_full(src)
            param.data.copy_(src.to(param.dtype).reshape(param.shape))
