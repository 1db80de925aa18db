"""Attention without a KV cache (vision towers) on the HIP extend kernel.

Mirrors nn/attention/vision.py: ``VisionAttention`` 24-180 (the ``use_qkv_parallel=True`` form that
Mllama uses, mllama.py:195-206), ``VisionTritonAttention`` 324-364 (variable-length, non-causal,
``cu_seqlens``) and ``VisionSdpaAttention`` 182-321 (padded batch + mask).

One kernel serves all of them: ``sp_extend_attention`` with ``causal = 0`` and no cached prefix.
The freshly projected K/V tensors [tokens, heads, D] are handed to it as if they were the KV pool,
with an identity ``req_to_token`` table (row b lists the token indices of sequence b), so the
MFMA kernel written for ragged prefill runs the ViT as well - no second attention kernel.

Head sizes the kernel does not tile natively (Mllama's vision heads are 80 wide; the kernel takes
64 and 128) are zero-padded IN THE WEIGHTS, once at load time: the padded q/k columns add 0 to every
logit, the padded v columns produce zeros that the zero rows of the padded out-projection ignore.
"""
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from . import _native
from .distributed import (divide, get_tensor_model_parallel_rank,
                          get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce)

KERNEL_HEAD_DIMS = (64, 128)


def kernel_head_dim(head_size: int) -> int:
    for d in KERNEL_HEAD_DIMS:
        if head_size <= d:
            return d
    raise NotImplementedError(f"head size {head_size} > {KERNEL_HEAD_DIMS[-1]}")


class _Workspace:
    """Grow-only scratch shared by every cache-less attention call on a device."""
    _buf = {}

    @classmethod
    def get(cls, nbytes: int, device) -> torch.Tensor:
        key = str(device)
        cur = cls._buf.get(key)
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
            cls._buf[key] = cur
        return cur


class VarlenPlan:
    """Launch arguments of one cache-less attention problem, on the device.  A ViT runs the same
    problem in every layer: build the plan once per forward and hand it to each call (no per-layer
    host work or H2D copies)."""

    def __init__(self, seq_lens: Sequence[int], device, kv_rows: int,
                 key_index: Optional[torch.Tensor] = None, key_lens: Optional[Sequence[int]] = None):
        self.seq_lens = [int(n) for n in seq_lens]
        self.n_seq = len(self.seq_lens)
        self.total = sum(self.seq_lens)
        ext = torch.tensor(self.seq_lens, dtype=torch.int32)
        start = torch.zeros(self.n_seq, dtype=torch.int32)
        if self.n_seq > 1:
            start[1:] = torch.cumsum(ext, 0)[:-1]
        if key_index is None:      # self-attention: sequence b's keys are its own rows
            max_keys = max(self.seq_lens) if self.seq_lens else 0
            rows = torch.arange(max(max_keys, 1), dtype=torch.int32).unsqueeze(0) + start.unsqueeze(1)
            key_index = rows.clamp_(max=max(kv_rows - 1, 0))
            key_lens = self.seq_lens
        self.key_lens = [int(n) for n in key_lens]
        self.max_ext = max(self.seq_lens) if self.seq_lens else 0
        self.max_keys = max(self.key_lens) if self.key_lens else 0
        self.key_index = key_index.to(device=device, dtype=torch.int32).contiguous()
        self.ext = ext.to(device)
        self.start = start.to(device)
        self.klen = torch.tensor(self.key_lens, dtype=torch.int32).to(device)
        self.req = torch.arange(self.n_seq, dtype=torch.int32, device=device)


def varlen_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, seq_lens: Optional[Sequence[int]],
                     sm_scale: float, causal: bool = False,
                     key_index: Optional[torch.Tensor] = None,
                     key_lens: Optional[Sequence[int]] = None,
                     plan: Optional[VarlenPlan] = None) -> torch.Tensor:
    """q: [sum(seq_lens), H, D]; k, v: [tokens, Hkv, D] (16-bit, D in KERNEL_HEAD_DIMS).

    Sequence b's queries are rows [cu[b], cu[b+1]) of q.  By default its keys are the same rows
    of k/v (self-attention, what context_attention_fwd computes for VisionTritonAttention).
    ``key_index`` [n_seq, max_keys] int32 + ``key_lens`` select other key rows per sequence
    (used for the padded-patch rows of the Mllama tile mask).  ``plan`` = a prebuilt VarlenPlan."""
    if plan is None:
        plan = VarlenPlan(seq_lens, q.device, k.shape[0], key_index, key_lens)
    if q.shape[0] != plan.total:
        raise RuntimeError(f"varlen_attention: q has {q.shape[0]} rows, seq_lens sum to {plan.total}")
    if q.dtype not in (torch.float16, torch.bfloat16) or q.shape[-1] not in KERNEL_HEAD_DIMS:
        raise RuntimeError("varlen_attention: 16-bit q/k/v with head dim 64 or 128 expected "
                           "(pad the projection weights, see VisionAttention)")
    out = torch.empty_like(q)
    if plan.total == 0:
        return out
    ws = _Workspace.get(_native.extend_workspace_bytes(plan.total, plan.n_seq, q.shape[1], q.shape[2], q.dtype),
                        q.device)
    _native.extend_attention(out, q, k, v, plan.key_index, plan.req, plan.klen, plan.ext, plan.start,
                             sm_scale, 0.0, causal, plan.max_ext, plan.max_keys, ws)
    return out


class VisionAttnPlan:
    """Per-forward plan of VisionAttention: the main problem and, when padding positions exist, the
    small second problem that recomputes them over the real keys only."""

    def __init__(self, bsz: int, s: int, device, cu_seqlens: Optional[List[int]] = None,
                 pad_rows: Optional[torch.Tensor] = None):
        if cu_seqlens is not None:       # VisionTritonAttention: ragged sequences inside the rows
            seq_lens = [cu_seqlens[i + 1] - cu_seqlens[i] for i in range(len(cu_seqlens) - 1)]
        else:
            seq_lens = [s] * bsz
        self.main = VarlenPlan(seq_lens, device, bsz * s)
        self.redo = self.redo_rows = None
        if pad_rows is None:
            return
        pad_rows = pad_rows.cpu()
        if not bool(pad_rows.any()):
            return
        q_rows, q_lens, key_rows, key_lens = [], [], [], []
        for b in range(bsz):
            pad = torch.nonzero(pad_rows[b]).flatten() + b * s
            real = torch.nonzero(~pad_rows[b]).flatten() + b * s
            if pad.numel() == 0:
                continue
            if real.numel() == 0:
                raise RuntimeError("VisionAttention: a sequence made of padding only")
            q_rows.append(pad)
            q_lens.append(int(pad.numel()))
            key_rows.append(real)
            key_lens.append(int(real.numel()))
        table = torch.zeros(len(key_rows), max(key_lens), dtype=torch.int32)
        for i, rows in enumerate(key_rows):
            table[i, :rows.numel()] = rows.to(torch.int32)
        self.redo = VarlenPlan(q_lens, device, bsz * s, table, key_lens)
        self.redo_rows = torch.cat(q_rows).to(device)


class ColumnParallelLinear(nn.Module):
    """linear.py ColumnParallelLinear with bias: output features sharded over TP ranks."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, dtype=None):
        super().__init__()
        self.out_per_rank = divide(output_size, get_tensor_model_parallel_world_size())
        self.weight = nn.Parameter(torch.empty(self.out_per_rank, input_size, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(self.out_per_rank, dtype=dtype), requires_grad=False) if bias else None

    def shard_from_full(self, full: torch.Tensor) -> torch.Tensor:
        r = get_tensor_model_parallel_rank()
        return full[r * self.out_per_rank:(r + 1) * self.out_per_rank]

    def forward(self, x):
        return F.linear(x, self.weight, self.bias), None


class RowParallelLinear(nn.Module):
    """linear.py:1033-1155 with bias: input features sharded, SUM all-reduce, bias added once."""

    def __init__(self, input_size: int, output_size: int, bias: bool = True, dtype=None):
        super().__init__()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.in_per_rank = divide(input_size, self.tp_size)
        self.weight = nn.Parameter(torch.empty(output_size, self.in_per_rank, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(output_size, dtype=dtype), requires_grad=False) if bias else None

    def shard_from_full(self, full: torch.Tensor) -> torch.Tensor:
        if full.dim() == 1:
            return full
        r = get_tensor_model_parallel_rank()
        return full[:, r * self.in_per_rank:(r + 1) * self.in_per_rank]

    def forward(self, x):
        if self.tp_size == 1:
            return F.linear(x, self.weight, self.bias), None
        out = tensor_model_parallel_all_reduce(F.linear(x, self.weight))
        return (out if self.bias is None else out + self.bias), None


class VisionAttention(nn.Module):
    """vision.py:24-180, ``use_qkv_parallel=True``: x [b, s, E] -> [b, s, E].

    ``pad_rows`` (optional, [b, s] bool on the HOST, True = padding position) gives the Mllama
    tile mask its exact meaning (mllama.py:403-410 via _prepare_aspect_ratio_attention_mask): a pair
    (query i, key j) is masked iff BOTH are padding positions.  Real positions therefore attend to
    every key; padding positions attend to the real keys only - a second, small launch over the
    gathered padding rows with a key table that lists the real positions."""

    def __init__(self, embed_dim: int, num_heads: int, projection_size: int, bias: bool = True, dtype=None):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.head_size = embed_dim // num_heads
        self.kernel_head_size = kernel_head_dim(self.head_size)
        self.total_num_heads = num_heads
        self.num_heads = divide(num_heads, tp)
        self.embed_dim = embed_dim
        self.scaling = self.head_size ** -0.5
        Dp = self.kernel_head_size
        # [q | k | v] blocks of num_heads * Dp rows; rows d >= head_size of every head stay zero
        self.qkv_proj = ColumnParallelLinear(embed_dim, 3 * num_heads * Dp, bias=bias, dtype=dtype)
        self.proj = RowParallelLinear(num_heads * Dp, embed_dim, bias=bias, dtype=dtype)

    # ---- weight layout ------------------------------------------------------------------------
    def pack_qkv(self, full: torch.Tensor) -> torch.Tensor:
        """Reference layout [3 * H * D, ...] (q | k | v, vision.py:80-86) -> this rank's padded rows."""
        H, D, Dp, Hl = self.total_num_heads, self.head_size, self.kernel_head_size, self.num_heads
        r = get_tensor_model_parallel_rank()
        tail = full.shape[1:]
        blocks = full.reshape(3, H, D, *tail)[:, r * Hl:(r + 1) * Hl]
        out = full.new_zeros(3, Hl, Dp, *tail)
        out[:, :, :D] = blocks
        return out.reshape(3 * Hl * Dp, *tail)

    def pack_proj(self, full: torch.Tensor) -> torch.Tensor:
        """Reference layout [E, H * D] -> [E, Hl * Dp] with zero columns for the padded dims."""
        H, D, Dp, Hl = self.total_num_heads, self.head_size, self.kernel_head_size, self.num_heads
        r = get_tensor_model_parallel_rank()
        cols = full.reshape(full.shape[0], H, D)[:, r * Hl:(r + 1) * Hl]
        out = full.new_zeros(full.shape[0], Hl, Dp)
        out[:, :, :D] = cols
        return out.reshape(full.shape[0], Hl * Dp)

    def load_reference_weights(self, qkv_w, qkv_b, proj_w, proj_b):
        self.qkv_proj.weight.data.copy_(self.pack_qkv(qkv_w).to(self.qkv_proj.weight.dtype))
        if self.qkv_proj.bias is not None and qkv_b is not None:
            self.qkv_proj.bias.data.copy_(self.pack_qkv(qkv_b).to(self.qkv_proj.bias.dtype))
        self.proj.weight.data.copy_(self.pack_proj(proj_w).to(self.proj.weight.dtype))
        if self.proj.bias is not None and proj_b is not None:
            self.proj.bias.data.copy_(proj_b.to(self.proj.bias.dtype))

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, cu_seqlens: Optional[List[int]] = None,
                pad_rows: Optional[torch.Tensor] = None, plan: Optional[VisionAttnPlan] = None) -> torch.Tensor:
        """``plan``: a VisionAttnPlan built once by the caller for all layers (else built here from
        ``cu_seqlens`` / ``pad_rows``)."""
        bsz, s, _ = x.shape
        Hl, Dp = self.num_heads, self.kernel_head_size
        if plan is None:
            plan = VisionAttnPlan(bsz, s, x.device, cu_seqlens, pad_rows)
        qkv, _ = self.qkv_proj(x)
        q, k, v = (t.reshape(bsz * s, Hl, Dp) for t in qkv.chunk(3, dim=-1))
        out = varlen_attention(q, k, v, None, self.scaling, plan=plan.main)
        if plan.redo is not None:
            redo = varlen_attention(q.index_select(0, plan.redo_rows), k, v, None, self.scaling, plan=plan.redo)
            out.index_copy_(0, plan.redo_rows, redo)
        out, _ = self.proj(out.reshape(bsz, s, Hl * Dp))
        return out
