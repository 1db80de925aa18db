"""RMSNorm / SiluAndMul / rotary embedding modules over the HIP kernels.

Mirrors nn/layers/layernorm.py:12-32, nn/layers/activation.py:21-31 and
nn/layers/rotary_embedding.py:52-170, 677-720, 918-1088 (same class names, constructor
arguments, return conventions and in-place contracts)."""
import math
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _native
from .custom_op import CustomOp


class RMSNorm(CustomOp):
    """layernorm.py:12-32.  With ``residual``: both tensors are updated IN PLACE and returned
    (flashinfer fused_add_rmsnorm contract); without: a new tensor is returned."""

    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward_hip(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                    ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if residual is not None:
            _native.fused_add_rmsnorm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        return _native.rmsnorm(x, self.weight.data, self.variance_epsilon)


class SiluAndMul(CustomOp):
    """activation.py:21-31: silu(x[..., :d]) * x[..., d:]."""

    def forward_hip(self, x: torch.Tensor) -> torch.Tensor:
        return _native.silu_and_mul(x)


class RotaryEmbedding(CustomOp):
    """rotary_embedding.py:52-170.  ``forward`` rotates query and key IN PLACE and returns them."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: int,
                 is_neox_style: bool, dtype: torch.dtype) -> None:
        super().__init__()
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        self.is_neox_style = is_neox_style
        self.dtype = dtype
        cache = self._compute_cos_sin_cache().to(dtype)
        self.cos_sin_cache: torch.Tensor
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def _compute_inv_freq(self, base: Union[int, float]) -> torch.Tensor:
        # rotary_embedding.py:77-90
        return 1.0 / (base ** (torch.arange(0, self.rotary_dim, 2, dtype=torch.float) / self.rotary_dim))

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        # rotary_embedding.py:92-101 (fp32 on the host, once at init)
        inv_freq = self._compute_inv_freq(self.base)
        t = torch.arange(self.max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1)

    def forward_hip(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                    offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.cos_sin_cache.device != query.device or self.cos_sin_cache.dtype != query.dtype:
            # rotary_embedding.py:141: the cache follows the activations' device and dtype
            self.cos_sin_cache = self.cos_sin_cache.to(query.device, dtype=query.dtype)
        if offsets is not None:
            positions = positions + offsets
        _native.rotary_embedding(positions, query, key, self.head_size, self.cos_sin_cache,
                                 self.is_neox_style)
        return query, key

    def extra_repr(self) -> str:
        return (f"head_size={self.head_size}, rotary_dim={self.rotary_dim}, "
                f"max_position_embeddings={self.max_position_embeddings}, base={self.base}, "
                f"is_neox_style={self.is_neox_style}")


class Llama3RotaryEmbedding(RotaryEmbedding):
    """rotary_embedding.py:677-720: wavelength-dependent rescale of inv_freq.

    Provenance: this is the public Llama-3.1 RoPE rescaling (Meta's reference implementation; the same formula in HF
    transformers `_compute_llama3_parameters` and in vLLM, from which the reference's class descends).  The cos/sin
    table has to equal the reference's bit for bit (tests/test_host_logic.py against rotary.npz), which fixes the order
    of the fp32 operations below; it runs once at init on the host and no kernel depends on it."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: int,
                 is_neox_style: bool, dtype: torch.dtype, scaling_factor: float,
                 low_freq_factor: float, high_freq_factor: float, orig_max_position: int) -> None:
        self.scaling_factor = scaling_factor
        self.low_freq_factor = low_freq_factor
        self.high_freq_factor = high_freq_factor
        self.orig_max_position = orig_max_position
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype)

    def _compute_inv_freq(self, base: Union[int, float]) -> torch.Tensor:
        inv_freqs = super()._compute_inv_freq(base)
        low_freq_wavelen = self.orig_max_position / self.low_freq_factor
        high_freq_wavelen = self.orig_max_position / self.high_freq_factor
        wave_len = 2 * math.pi / inv_freqs
        if self.low_freq_factor != self.high_freq_factor:
            smooth = (self.orig_max_position / wave_len - self.low_freq_factor) / (
                self.high_freq_factor - self.low_freq_factor)
        else:
            smooth = 0
        return torch.where(
            wave_len < high_freq_wavelen, inv_freqs,
            torch.where(wave_len > low_freq_wavelen, inv_freqs / self.scaling_factor,
                        (1 - smooth) * inv_freqs / self.scaling_factor + smooth * inv_freqs))


class ScaledRotaryEmbedding(RotaryEmbedding):
    """The context-extension variants get_rope hands out besides llama3 (rotary_embedding.py:173-414, 996-1036), as
    ONE table builder: a variant is (how the inverse frequencies are bent, what the positions are divided by, how long
    the table is, what cos / sin are multiplied by); the rotation kernel only ever sees the finished cos | sin table.

      "linear"  (position interpolation): positions / factor, table for max_pos * factor positions;
      "dynamic" (dynamic NTK): base * ((factor * len / max_pos) - (factor - 1)) ** (rot / (rot - 2)) with len =
                max_pos * factor - the reference evaluates the NTK base once, at the longest length;
      "yarn"    (Peng et al.): per-frequency blend of interpolated (1 / (factor * f)) and original (1 / f) inverse
                frequencies along a linear ramp between the dimensions that complete beta_fast and beta_slow
                rotations inside the ORIGINAL context, and cos / sin scaled by (0.1 ln(factor) + 1) * attn_factor.

    Public formulas (kaiokendev's interpolation; bloc97 / emozilla's NTK scaling; the YaRN paper and repository); the
    tables equal the reference classes' bit for bit (rotary.npz cases 6-10), which fixes the order of the fp32
    operations.  A LIST of linear factors (one table per LoRA adapter, rotary_embedding.py:173-258) is not built."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: int,
                 is_neox_style: bool, dtype: torch.dtype, kind: str, factor: float,
                 extrapolation_factor: float = 1, attn_factor: float = 1, beta_fast: int = 32,
                 beta_slow: int = 1) -> None:
        if kind not in ("linear", "dynamic", "yarn"):
            raise ValueError(f"Unknown RoPE scaling type {kind}")
        if isinstance(factor, (list, tuple)):
            raise NotImplementedError("several linear scaling factors in one table (per-LoRA tables) are not built")
        self.kind, self.scaling_factor = kind, factor
        self.yarn = (extrapolation_factor, attn_factor, beta_fast, beta_slow)
        self.mscale = float((0.1 * math.log(factor) + 1.0 if factor > 1 else 1.0) * attn_factor) if kind == "yarn" else 1.0
        super().__init__(head_size, rotary_dim, max_position_embeddings, base, is_neox_style, dtype)

    def _yarn_inv_freq(self) -> torch.Tensor:
        extrapolation_factor, _, beta_fast, beta_slow = self.yarn
        rot, base, ctx = self.rotary_dim, self.base, self.max_position_embeddings
        freq = base ** (torch.arange(0, rot, 2, dtype=torch.float) / rot)
        # the (fractional) dimension whose wavelength makes `turns` full rotations over the original context
        turn_dim = lambda turns: (rot * math.log(ctx / (turns * 2 * math.pi))) / (2 * math.log(base))
        lo = max(math.floor(turn_dim(beta_fast)), 0)
        hi = min(math.ceil(turn_dim(beta_slow)), rot - 1)
        if lo == hi:
            hi += 0.001
        ramp = torch.clamp((torch.arange(rot // 2, dtype=torch.float) - lo) / (hi - lo), 0, 1)
        keep = (1 - ramp) * extrapolation_factor                  # 1: the original frequency, 0: the interpolated one
        return (1.0 / (self.scaling_factor * freq)) * (1 - keep) + (1.0 / freq) * keep

    def _compute_cos_sin_cache(self) -> torch.Tensor:
        f, ctx = self.scaling_factor, self.max_position_embeddings
        length = ctx * f
        t = torch.arange(length, dtype=torch.float)
        if self.kind == "linear":
            inv_freq, t = self._compute_inv_freq(self.base), t / f
        elif self.kind == "dynamic":
            inv_freq = self._compute_inv_freq(
                self.base * ((f * length / ctx) - (f - 1)) ** (self.rotary_dim / (self.rotary_dim - 2)))
        else:
            inv_freq = self._yarn_inv_freq()
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        if self.kind == "yarn":
            return torch.cat((freqs.cos() * self.mscale, freqs.sin() * self.mscale), dim=-1)
        return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


_ROPE_DICT: Dict[Tuple, RotaryEmbedding] = {}


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: int,
             is_neox_style: bool = True, rope_scaling: Optional[Dict[str, Any]] = None,
             dtype: Optional[torch.dtype] = None,
             partial_rotary_factor: float = 1.0) -> RotaryEmbedding:
    """rotary_embedding.py:918-1088: "default", "llama3" (the hot-path configs), "linear", "dynamic" and "yarn"
    (ScaledRotaryEmbedding); "deepseek_yarn", "longrope" and mrope raise (MLA / Phi-3 / Qwen2-VL: model families out
    of scope)."""
    if dtype is None:
        dtype = torch.get_default_dtype()
    if rope_scaling is not None:
        rope_scaling_args = tuple(
            (k, tuple(v) if isinstance(v, list) else v) for k, v in rope_scaling.items())
    else:
        rope_scaling_args = None
    if partial_rotary_factor < 1.0:
        rotary_dim = int(rotary_dim * partial_rotary_factor)
    key = (head_size, rotary_dim, max_position, base, is_neox_style, rope_scaling_args, dtype)
    if key in _ROPE_DICT:
        return _ROPE_DICT[key]
    scaling_type = None if rope_scaling is None else rope_scaling.get(
        "rope_type", rope_scaling.get("type"))
    if scaling_type in (None, "default"):
        rope = RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype)
    elif scaling_type == "llama3":
        rope = Llama3RotaryEmbedding(
            head_size, rotary_dim, max_position, base, is_neox_style, dtype,
            rope_scaling["factor"], rope_scaling["low_freq_factor"],
            rope_scaling["high_freq_factor"], rope_scaling["original_max_position_embeddings"])
    elif scaling_type in ("linear", "dynamic"):
        rope = ScaledRotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype,
                                     scaling_type, rope_scaling["factor"])
    elif scaling_type == "yarn":
        # (the table is built over the ORIGINAL context times the factor: rotary_embedding.py:1018-1036)
        extra = {k: rope_scaling[k] for k in ("extrapolation_factor", "attn_factor", "beta_fast", "beta_slow")
                 if k in rope_scaling}
        rope = ScaledRotaryEmbedding(head_size, rotary_dim, rope_scaling["original_max_position_embeddings"], base,
                                     is_neox_style, dtype, "yarn", rope_scaling["factor"], **extra)
    else:
        raise ValueError(f"Unknown RoPE scaling type {scaling_type}")
    _ROPE_DICT[key] = rope
    return rope
