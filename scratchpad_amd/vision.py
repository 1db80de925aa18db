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
        self.workspace: Optional[torch.Tensor] = None


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
    nbytes = _native.extend_workspace_bytes(plan.total, plan.n_seq, q.shape[1], q.shape[2], q.dtype)
    if plan.workspace is None or plan.workspace.numel() < nbytes:     # per plan: plans may run concurrently
        plan.workspace = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=q.device)
    ws = plan.workspace
    _native.extend_attention(out, q, k, v, plan.key_index, plan.req, plan.klen, plan.ext, plan.start,
                             sm_scale, 0.0, causal, plan.max_ext, plan.max_keys, ws)
    return out


class VisionAttnPlan:
    """Per-forward plan of VisionAttention.

    ``main``: every query over every key of its sequence.  ``side`` (optional): a few rows that are
    better served by a second, tiny launch running on a side stream under the main one:
      * padding positions, which see the real keys only (the tile-mask semantics, see VisionAttention);
      * (MOVE_SPILL_ROWS, off) up to ROW_BLOCK/2 real rows that would otherwise open a nearly empty
        last 128-row block: Mllama's 4 x 1032 = 4128 positions are 32 blocks + 32 rows, i.e. 528
        workgroups for 512 resident slots (2 per CU).  Measured on the 11B tower: moving them made
        the forward SLOWER (43.6 vs 40.9 ms) - the gather/scatter of q and o around the main launch
        costs more than the 16 late workgroups, which run alone and fast.  Kept switchable."""
    ROW_BLOCK = 128          # query rows per workgroup of extend_mfma_kernel at G = 1
    MOVE_SPILL_ROWS = False

    def __init__(self, bsz: int, s: int, device, cu_seqlens: Optional[List[int]] = None,
                 pad_rows: Optional[torch.Tensor] = None):
        self.side = self.side_rows = self.main_rows = self.side_stream = None
        if cu_seqlens is not None:       # VisionTritonAttention: ragged sequences inside the rows
            seq_lens = [cu_seqlens[i + 1] - cu_seqlens[i] for i in range(len(cu_seqlens) - 1)]
            self.main = VarlenPlan(seq_lens, device, bsz * s)
            return
        pad_rows = torch.zeros(bsz, s, dtype=torch.bool) if pad_rows is None else pad_rows.cpu()
        main_rows, main_lens, side_q, side_lens, side_keys, side_key_lens = [], [], [], [], [], []
        for b in range(bsz):
            base = b * s
            pad = torch.nonzero(pad_rows[b]).flatten()
            real = torch.nonzero(~pad_rows[b]).flatten()
            if pad.numel() and real.numel() == 0:
                raise RuntimeError("VisionAttention: a sequence made of padding only")
            keep = s - pad.numel() if (self.MOVE_SPILL_ROWS and pad.numel() <= self.ROW_BLOCK // 2) else s
            spill = keep % self.ROW_BLOCK
            move_real = spill if (self.MOVE_SPILL_ROWS and keep >= 8 * self.ROW_BLOCK
                                  and 0 < spill <= self.ROW_BLOCK // 2 and keep == s - pad.numel()) else 0
            if keep == s:                # too many padding rows to move: they stay in main and are redone
                rows = torch.arange(s)
                if pad.numel():
                    side_q.append(pad + base); side_lens.append(int(pad.numel()))
                    side_keys.append(real + base); side_key_lens.append(int(real.numel()))
            else:
                moved = real[real.numel() - move_real:] if move_real else real[:0]
                rows = real[:real.numel() - move_real]
                if pad.numel():
                    side_q.append(pad + base); side_lens.append(int(pad.numel()))
                    side_keys.append(real + base); side_key_lens.append(int(real.numel()))
                if move_real:
                    side_q.append(moved + base); side_lens.append(int(moved.numel()))
                    side_keys.append(torch.arange(s) + base); side_key_lens.append(s)
            main_rows.append(rows + base)
            main_lens.append(int(rows.numel()))
        natural = all(n == s for n in main_lens)
        all_keys = (torch.arange(s, dtype=torch.int32).unsqueeze(0)
                    + (torch.arange(bsz, dtype=torch.int32) * s).unsqueeze(1))
        self.main = VarlenPlan(main_lens, device, bsz * s, None if natural else all_keys,
                               None if natural else [s] * bsz)
        if not natural:
            self.main_rows = torch.cat(main_rows).to(device)
        if side_q:
            table = torch.zeros(len(side_keys), max(side_key_lens), dtype=torch.int32)
            for i, rows in enumerate(side_keys):
                table[i, :rows.numel()] = rows.to(torch.int32)
            self.side = VarlenPlan(side_lens, device, bsz * s, table, side_key_lens)
            self.side_rows = torch.cat(side_q).to(device)
            self.side_stream = torch.cuda.Stream(device=device)


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
        if plan.side is None:
            out = varlen_attention(q, k, v, None, self.scaling, plan=plan.main)
        else:
            cur = torch.cuda.current_stream()
            plan.side_stream.wait_stream(cur)
            with torch.cuda.stream(plan.side_stream):
                o_side = varlen_attention(q.index_select(0, plan.side_rows), k, v, None, self.scaling,
                                          plan=plan.side)
            if plan.main_rows is None:
                out = varlen_attention(q, k, v, None, self.scaling, plan=plan.main)
            else:
                o_main = varlen_attention(q.index_select(0, plan.main_rows), k, v, None, self.scaling,
                                          plan=plan.main)
                out = o_main.new_empty(bsz * s, Hl, Dp)
                out.index_copy_(0, plan.main_rows, o_main)
            cur.wait_stream(plan.side_stream)
            o_side.record_stream(cur)
            out.index_copy_(0, plan.side_rows, o_side)
        out, _ = self.proj(out.reshape(bsz, s, Hl * Dp))
        return out
