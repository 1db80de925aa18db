"""Attention plugin seam: AttentionBackend ABC, RadixAttention layer, and the HIP backend.

Mirrors nn/attention/backend.py:11-105 (same abstract methods and dispatch),
nn/attention/radix_attention.py:6-57, and plays the role of TritonAttnBackend /
FlashInferAttnBackend (nn/attention/triton_backend.py:17-196, flashinfer_backend.py:49-496):
it reads ``req_to_token`` directly like the Triton backend (no kv_indices materialisation) and
implements the flashinfer backend's encoder-decoder dispatch (cross-attention reads kv slots
[0, encoder_len); self-attention reads [encoder_len, encoder_len + seq_len)).
"""
from abc import ABC, abstractmethod
from typing import TYPE_CHECKING, Optional

import torch
from torch import nn

from . import _native

if TYPE_CHECKING:
    from .forward_info import ForwardBatch, ForwardMode


class AttentionBackend(ABC):
    """The base class of attention backends (backend.py:11-105)."""

    @abstractmethod
    def init_forward_metadata(self, forward_batch: "ForwardBatch"):
        raise NotImplementedError()

    def init_cuda_graph_state(self, max_bs: int):
        raise NotImplementedError()

    def init_forward_metadata_capture_cuda_graph(self, bs: int, num_tokens: int,
                                                 req_pool_indices: torch.Tensor,
                                                 seq_lens: torch.Tensor,
                                                 encoder_lens: Optional[torch.Tensor],
                                                 forward_mode: "ForwardMode", spec_info=None):
        raise NotImplementedError()

    def init_forward_metadata_replay_cuda_graph(self, bs: int, req_pool_indices: torch.Tensor,
                                                seq_lens: torch.Tensor, seq_lens_sum: int,
                                                encoder_lens: Optional[torch.Tensor],
                                                forward_mode: "ForwardMode", spec_info=None,
                                                seq_lens_cpu: Optional[torch.Tensor] = None):
        raise NotImplementedError()

    def get_cuda_graph_seq_len_fill_value(self):
        raise NotImplementedError()

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, layer: "RadixAttention",
                forward_batch: "ForwardBatch", save_kv_cache: bool = True):
        if forward_batch.forward_mode.is_decode():
            return self.forward_decode(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache)
        return self.forward_extend(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache: bool = True):
        raise NotImplementedError()


class RadixAttention(nn.Module):
    """radix_attention.py:6-57: layer-local constants + delegation to the backend."""

    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int,
                 layer_id: int, logit_cap: float = 0.0, v_head_dim: int = -1,
                 sliding_window_size: int = -1, is_cross_attention: bool = False,
                 prefix: str = "", use_irope: bool = False):
        super().__init__()
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.head_dim = head_dim
        self.qk_head_dim = head_dim
        self.v_head_dim = v_head_dim if v_head_dim != -1 else head_dim
        self.scaling = scaling
        self.layer_id = layer_id
        self.logit_cap = logit_cap
        self.sliding_window_size = sliding_window_size or -1
        self.is_cross_attention = is_cross_attention
        self.k_scale = None
        self.v_scale = None
        # per-layer KV scales as floats: flashinfer_backend.py:400-401, 457-458 read these two names
        # (set by the checkpoint's kv-scale loader upstream; the reference's class never defines them)
        self.k_scale_float = None
        self.v_scale_float = None
        self.use_irope = use_irope

    def forward(self, q, k, v, forward_batch: "ForwardBatch", save_kv_cache: bool = True):
        if k is not None:
            assert v is not None
            k = k.view(-1, self.tp_k_head_num, self.qk_head_dim)
            v = v.view(-1, self.tp_v_head_num, self.v_head_dim)
        return forward_batch.attn_backend.forward(q, k, v, self, forward_batch, save_kv_cache)


def _pow2_floor(x: int) -> int:
    return 1 << (max(int(x), 1).bit_length() - 1)


class HipAttnBackend(AttentionBackend):
    """MI355X attention backend over ``sp_decode_attention`` / ``sp_extend_attention``.

    Constructed with a ``model_runner`` like the reference's backends (reads ``model_config``,
    ``tp_size``, ``token_to_kv_pool``, ``req_to_token_pool``, ``device``)."""

    # work items (request x split x head-group) we want per launch: one per CU.  Measured sweep
    # (tools/bench_decode_attn.py, bs 1-128 x ctx 1024/4096): fewer, longer workgroups win as soon
    # as every CU has one - the per-workgroup prologue (index -> gather -> first tile) is amortised
    # over more tiles - e.g. bs 32 x 1024: chunk 256 (256 items) 28.5 us vs chunk 64 (1024) 34.1 us
    TARGET_ITEMS = 256
    MIN_CHUNK, MAX_CHUNK = 64, 512
    # LlamaAttention may hand rotary + KV store to the backend as one kernel (sp_rotary_embedding
    # with pool arguments); set False to keep the reference's two-step order
    fused_rope_kv_store = True

    def __init__(self, model_runner):
        super().__init__()
        _native.load()
        cfg = model_runner.model_config
        self.num_head = cfg.num_attention_heads // model_runner.tp_size
        self.num_kv_head = cfg.get_num_kv_heads(model_runner.tp_size)
        self.head_dim = cfg.head_dim
        self.v_head_dim = model_runner.token_to_kv_pool.get_value_buffer(0).shape[-1]
        self.max_context_len = cfg.context_len
        self.device = model_runner.device
        self.kv_dtype = model_runner.token_to_kv_pool.dtype
        self.is_encoder_decoder = bool(getattr(cfg, "is_encoder_decoder", False))
        self.forward_metadata = None
        self._workspace = torch.empty(0, dtype=torch.uint8, device=self.device)
        self._plans = [torch.empty(0, dtype=torch.int32, device=self.device) for _ in range(3)]
        # Gemma-2 style models: some layers see only the last `sliding_window_size` keys + themselves
        # (the reference keeps a second flashinfer wrapper for them, flashinfer_backend.py:76-83)
        sw = getattr(model_runner, "sliding_window_size", None)
        self.sliding_window_size = sw if sw not in (None, -1) else None
        self._window = None            # (lens, kv_start) of the windowed layers, this step
        self._graph_window = None
        self._graph_state = {}         # bs bucket -> (chunk, workspace, plan buffers) of its captured graph
        self._extend_plan = None       # int32 work list of the current extend step (sp_extend_plan)

    # ---------------------------------------------------------------- launch planning
    def _head_groups(self, dtype: torch.dtype) -> int:
        vec = 4 if dtype == torch.float32 else 8
        rpl = 64 // (self.head_dim // vec)
        hh = 1
        while hh * 2 <= rpl and self.num_kv_head % (hh * 2) == 0:
            hh *= 2
        return self.num_kv_head // hh

    def _plan_chunk(self, kv_tokens: int, dtype: torch.dtype) -> int:
        groups = self._head_groups(dtype)
        chunk = _pow2_floor(max(kv_tokens, 1) * groups // self.TARGET_ITEMS)
        return max(self.MIN_CHUNK, min(self.MAX_CHUNK, chunk))

    def _ensure_workspace(self, nbytes: int) -> torch.Tensor:
        if self._workspace.numel() < nbytes:
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._workspace

    def _build_plans(self, plans, bs, seq_lens, encoder_lens, max_len, chunk):
        """One split plan per kv window (self-attention lens; encoder lens for cross-attention -
        the reference keeps two flashinfer wrappers for the same reason, flashinfer_backend.py:
        121-131).  Built once per step, read by every layer's launch."""
        need = _native.decode_plan_bytes(bs, max_len, chunk) // 4
        out = []
        window_lens = None if self._window is None else self._window[0]
        for i, lens in enumerate((seq_lens, encoder_lens, window_lens)):
            if lens is None:
                out.append(None)
                continue
            if plans[i].numel() < need:
                plans[i] = torch.empty(need, dtype=torch.int32, device=self.device)
            _native.decode_plan(plans[i], lens, max_len, chunk)
            out.append(plans[i])
        return tuple(out)

    # ---------------------------------------------------------------- metadata hooks
    def init_forward_metadata(self, forward_batch: "ForwardBatch"):
        """Per-step plan.  Decode: (chunk, max_seq_len, workspace); extend: (max_extend_len,
        max_seq_len, workspace).  No device sync: bounds come from host-side fields."""
        pool_dtype = forward_batch.token_to_kv_pool.dtype
        bs = forward_batch.batch_size
        enc_max = 0
        if forward_batch.encoder_lens_cpu:
            enc_max = max(forward_batch.encoder_lens_cpu)
        if forward_batch.forward_mode.is_decode():
            if forward_batch.seq_lens_cpu is not None:
                max_len = int(forward_batch.seq_lens_cpu.max())
            else:
                max_len = min(self.max_context_len, forward_batch.seq_lens_sum - (bs - 1))
            max_len = max(max_len, enc_max, 1)
            chunk = self._plan_chunk(forward_batch.seq_lens_sum, pool_dtype)
            ws = self._ensure_workspace(_native.decode_workspace_bytes(
                bs, self.num_head, self.v_head_dim, max_len, chunk))
            enc = forward_batch.encoder_lens if self.is_encoder_decoder else None
            self._window = self._window_of(forward_batch.seq_lens)
            plans = self._build_plans(self._plans, bs, forward_batch.seq_lens, enc, max_len, chunk)
            self.forward_metadata = (chunk, max_len, ws, plans)
        else:
            max_extend = max(forward_batch.extend_seq_lens_cpu)
            if forward_batch.seq_lens_cpu is not None:
                max_len = int(forward_batch.seq_lens_cpu.max())
            else:
                max_len = max(p + e for p, e in zip(forward_batch.extend_prefix_lens_cpu,
                                                    forward_batch.extend_seq_lens_cpu))
            max_len = max(max_len, enc_max, 1)
            ws = self._ensure_workspace(_native.extend_workspace_bytes(
                forward_batch.extend_num_tokens, bs, self.num_head, self.head_dim,
                pool_dtype if pool_dtype.itemsize > 1 else torch.bfloat16))
            # the step's (request, row block) work list, heaviest first: built once, read by every layer
            # (where the reference's backends run begin_forward, flashinfer_backend.py:672-830)
            self._extend_plan = _native.extend_plan(forward_batch.extend_seq_lens, forward_batch.seq_lens,
                                                    forward_batch.extend_num_tokens, self.num_head,
                                                    self.num_kv_head, True, self._extend_plan)
            self.forward_metadata = (max_extend, max_len, ws)

    def init_cuda_graph_state(self, max_bs: int):
        """Static split geometry + workspace for graph replay (triton_backend.py:70-80 allocates
        static attn_logits the same way)."""
        self.cuda_graph_max_seq_len = self.max_context_len
        # split geometry is part of the captured launches, so it is chosen per batch-size bucket at
        # capture time (a bs-1 graph wants 64-key splits, a bs-256 graph 512-key ones)
        self._graph_state = {}
        if self.sliding_window_size is not None:
            self._graph_window = tuple(torch.ones(max_bs, dtype=torch.int32, device=self.device)
                                       for _ in range(2))

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                 encoder_lens, forward_mode, spec_info=None):
        assert forward_mode.is_decode(), "only decode is captured"
        assert spec_info is None, "speculative decoding is out of scope"
        chunk = self._plan_chunk(bs * self.max_context_len // 2, self.kv_dtype)
        ws = torch.empty(_native.decode_workspace_bytes(bs, self.num_head, self.v_head_dim,
                                                        self.cuda_graph_max_seq_len, chunk),
                         dtype=torch.uint8, device=self.device)
        n = _native.decode_plan_bytes(bs, self.cuda_graph_max_seq_len, chunk) // 4
        plan_bufs = [torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(3)]
        self._graph_state[bs] = (chunk, ws, plan_bufs)
        self._window = self._window_of(seq_lens, self._graph_window, bs)
        plans = self._build_plans(plan_bufs, bs, seq_lens, encoder_lens, self.cuda_graph_max_seq_len, chunk)
        self.forward_metadata = (chunk, self.cuda_graph_max_seq_len, ws, plans)

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                encoder_lens, forward_mode, spec_info=None,
                                                seq_lens_cpu=None):
        # geometry is static and the kernels read seq_lens / req_pool_indices from the graph's
        # static input buffers; only the split plan (static buffer, fixed address) is rebuilt for
        # this step's lengths, ahead of the replay - where the reference recomputes start_loc /
        # kv_indices (triton_backend.py:103-113, flashinfer_backend.py:330-373)
        chunk, ws, plan_bufs = self._graph_state[bs]
        self._window = self._window_of(seq_lens[:bs], self._graph_window, bs)
        plans = self._build_plans(plan_bufs, bs, seq_lens[:bs],
                                  None if encoder_lens is None else encoder_lens[:bs],
                                  self.cuda_graph_max_seq_len, chunk)
        self.forward_metadata = (chunk, self.cuda_graph_max_seq_len, ws, plans)

    def get_cuda_graph_seq_len_fill_value(self):
        return 1  # padded rows attend to the dummy slot 0 only (triton_backend.py:115-116)

    # ---------------------------------------------------------------- forward
    def _window_of(self, seq_lens: torch.Tensor, static=None, bs: Optional[int] = None):
        """Decode kv range of the sliding-window layers (flashinfer_backend.py:559-577):
        lens = min(seq_lens, window + 1), kv_start = seq_lens - lens.  With ``static`` buffers (graph
        capture / replay) the values are written in place so the captured launches see them."""
        if self.sliding_window_size is None:
            return None
        lens = torch.clamp(seq_lens, max=self.sliding_window_size + 1)
        start = seq_lens - lens
        if static is None:
            return lens, start
        static[0][:bs].copy_(lens)
        static[1][:bs].copy_(start)
        return static[0][:bs], static[1][:bs]

    def _is_windowed(self, layer: RadixAttention) -> bool:
        return layer.sliding_window_size not in (-1, None) and not layer.is_cross_attention

    def _kv_window(self, layer: RadixAttention, forward_batch: "ForwardBatch"):
        """(seq_lens, kv_start) for this layer: flashinfer_backend.py:593-621, 792-828."""
        if self._is_windowed(layer) and forward_batch.forward_mode.is_decode():
            if self._window is None or layer.sliding_window_size != self.sliding_window_size:
                raise RuntimeError("sliding-window layer but the runner declares no (or another) "
                                   "sliding_window_size")
            return self._window
        if layer.is_cross_attention:
            return forward_batch.encoder_lens, None
        if self.is_encoder_decoder and forward_batch.encoder_lens is not None:
            return forward_batch.seq_lens, forward_batch.encoder_lens
        return forward_batch.seq_lens, None

    def rotary_and_store(self, rope, positions, q, k, v, layer: RadixAttention,
                         forward_batch: "ForwardBatch") -> None:
        """rotary_emb(positions, q, k) + set_kv_buffer(layer, out_cache_loc, k, v) in one launch
        (rotary_embedding.py:132-164 followed by pool.py:392-424).  q and k are rotated in place."""
        pool = forward_batch.token_to_kv_pool
        if rope.cos_sin_cache.device != q.device or rope.cos_sin_cache.dtype != q.dtype:
            rope.cos_sin_cache = rope.cos_sin_cache.to(q.device, dtype=q.dtype)
        if pool.dtype != q.dtype or layer.is_cross_attention:
            rope(positions, q, k)
            self._store(layer, forward_batch, k.view(-1, layer.tp_k_head_num, layer.qk_head_dim),
                        v.view(-1, layer.tp_v_head_num, layer.v_head_dim), True)
            return
        kb, vb = pool.get_kv_buffer(layer.layer_id)
        _native.rotary_embedding(positions, q, k, rope.head_size, rope.cos_sin_cache, rope.is_neox_style,
                                 value=v, k_buffer=kb, v_buffer=vb, out_cache_loc=forward_batch.out_cache_loc)

    def _kv_scales(self, layer: RadixAttention):
        """flashinfer_backend.py:400-401, 457-458: the layer's float scales, only for a non-"auto"
        (fp8) KV cache; they go to the store (which divides) and to the kernels (which multiply)."""
        if self.kv_dtype != torch.float8_e5m2:
            return None, None
        return layer.k_scale_float, layer.v_scale_float

    @staticmethod
    def _alloc_out(q: torch.Tensor, layer: RadixAttention) -> torch.Tensor:
        # The kernels leave a row with no visible key untouched (sp_decode_attention: seq_len 0).
        # Only cross-attention has such rows (text-only requests, encoder_len 0); they must read as
        # zeros because the model multiplies them by the row mask (mllama.py:621-622).
        # (allocated with spare rows behind it, like every activation that feeds a projection: _native.library_rows)
        return _native.empty_rows(q.shape[0], q.shape[1], q.dtype, q.device, zero=layer.is_cross_attention)

    def _store(self, layer, forward_batch, k, v, save_kv_cache):
        if k is None or not save_kv_cache:
            return
        assert v is not None
        cache_loc = (forward_batch.encoder_out_cache_loc if layer.is_cross_attention
                     else forward_batch.out_cache_loc)
        # flashinfer_backend.py:470-473: the layer's scales go to the store AND to the kernels
        k_scale, v_scale = self._kv_scales(layer)
        forward_batch.token_to_kv_pool.set_kv_buffer(layer, cache_loc, k, v, k_scale, v_scale)

    def forward_extend(self, q, k, v, layer: RadixAttention, forward_batch: "ForwardBatch",
                       save_kv_cache: bool = True):
        if layer.qk_head_dim != layer.v_head_dim:
            raise NotImplementedError("v_head_dim != head_dim (MLA) is out of scope")
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        o = self._alloc_out(q, layer)
        # KV store BEFORE the kernel: the kernel reads every key, new ones included, from the pool
        # (triton_backend.py:131-134)
        self._store(layer, forward_batch, k, v, save_kv_cache)
        max_extend, max_len, ws = self.forward_metadata
        k_scale, v_scale = self._kv_scales(layer)
        seq_lens, kv_start = self._kv_window(layer, forward_batch)
        kb, vb = forward_batch.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        _native.extend_attention(
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), kb, vb,
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices, seq_lens,
            forward_batch.extend_seq_lens, forward_batch.extend_start_loc, layer.scaling,
            layer.logit_cap, not layer.is_cross_attention, max_extend, max_len, ws, kv_start,
            window_left=layer.sliding_window_size if self._is_windowed(layer) else -1,
            k_scale=k_scale, v_scale=v_scale,
            plan=self._extend_plan if layer.tp_q_head_num == self.num_head
            and layer.tp_k_head_num == self.num_kv_head else None)
        return o

    def forward_decode(self, q, k, v, layer: RadixAttention, forward_batch: "ForwardBatch",
                       save_kv_cache: bool = True):
        if layer.qk_head_dim != layer.v_head_dim:
            raise NotImplementedError("v_head_dim != head_dim (MLA) is out of scope")
        q = q.reshape(-1, layer.tp_q_head_num * layer.qk_head_dim)
        o = self._alloc_out(q, layer)
        self._store(layer, forward_batch, k, v, save_kv_cache)
        chunk, max_len, ws, plans = self.forward_metadata
        k_scale, v_scale = self._kv_scales(layer)
        seq_lens, kv_start = self._kv_window(layer, forward_batch)
        plan = plans[2] if self._is_windowed(layer) else (plans[1] if layer.is_cross_attention else plans[0])
        kb, vb = forward_batch.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        _native.decode_attention(
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), kb, vb,
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices, seq_lens,
            layer.scaling, layer.logit_cap, max_len, chunk, ws, kv_start, plan,
            k_scale=k_scale, v_scale=v_scale)
        return o
