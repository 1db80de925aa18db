"""Attention plugin seam: AttentionBackend ABC, RadixAttention layer, and the HIP backend.

Mirrors nn/attention/backend.py:11-105 (same abstract methods and dispatch),
nn/attention/radix_attention.py:6-57, and plays the role of TritonAttnBackend /
FlashInferAttnBackend (nn/attention/triton_backend.py:17-196, flashinfer_backend.py:49-496):
it reads ``req_to_token`` directly like the Triton backend (no kv_indices materialisation) and
implements the flashinfer backend's encoder-decoder dispatch (cross-attention reads kv slots
[0, encoder_len); self-attention reads [encoder_len, encoder_len + seq_len)).
"""
import os
import threading
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

    # Launch-planning constants; the measurements behind each value are in profiles/NOTES.md ("attention.py constants").
    TARGET_ITEMS = int(os.environ.get("SP_DECODE_TARGET_ITEMS", "256"))     # work items per launch: about one per CU
    MIN_CHUNK = int(os.environ.get("SP_DECODE_MIN_CHUNK", "64"))            # smallest split (keys)
    MAX_CHUNK = int(os.environ.get("SP_DECODE_MAX_CHUNK", "768"))           # split cap: load balance of ragged batches
    # graph replay: a captured launch of bucket bs covers max(FLOOR, PER_REQ x bs) + bs work items / partial slots
    GRAPH_SLOTS_FLOOR, GRAPH_SLOTS_PER_REQ = 1024, 8
    fused_rope_kv_store = True      # rotary + KV store as one kernel; False keeps the reference's two-step order
    # True: every plan's overflow word is read back right after it is built (one device sync per step: tests,
    # debugging).  False: the 16-byte header is copied to pinned memory asynchronously and checked when the
    # NEXT plan is built, or by check_plans() - an understated seq_lens_sum raises one step late instead of never
    strict_plan_check = False

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
        self.pool_tokens = int(getattr(model_runner, "max_total_num_tokens", 0)) or None   # bounds sum(seq_lens)
        self._workspace = torch.empty(0, dtype=torch.uint8, device=self.device)
        self._plans = [torch.empty(0, dtype=torch.int32, device=self.device) for _ in range(3)]
        # Gemma-2 style models: some layers see only the last `sliding_window_size` keys + themselves
        # (the reference keeps a second flashinfer wrapper for them, flashinfer_backend.py:76-83)
        sw = getattr(model_runner, "sliding_window_size", None)
        self.sliding_window_size = sw if sw not in (None, -1) else None
        self._window = None            # (lens, kv_start) of the windowed layers, this step
        self._graph_window = None
        self._graph_ws = None          # graph replay: ONE partials workspace / plan-buffer triple for all buckets
        self._graph_plans = None
        self._extend_plan = None       # int32 work list of the current extend step (sp_extend_plan)
        self._plan_checks = []         # (pinned header copy, event, max_slots, step) of item plans not yet checked
        self.plan_step = None          # the overlap worker's step number while its forward thread builds that step's plans
        self._plan_hosts = []          # pinned buffers + events to reuse
        self._plan_lock = threading.Lock()   # the overlap worker checks from the scheduler thread (tp_worker_client.py)
        # Range geometry (include/scratchpad_hip.h): the pieces the step's keys are cut into, one wave per (piece, kv head),
        # two workgroups per CU; 0 where the range kernel does not take the shape (fp32, groups wider than 16).
        # SP_DECODE_RANGES=0 switches it off, =N forces N pieces (A/B runs).
        env = os.environ.get("SP_DECODE_RANGES", "")
        self.decode_ranges = int(env) if env else _native.decode_ranges(
            self.num_head, self.num_kv_head, self.head_dim, getattr(model_runner, "dtype", torch.bfloat16), self.kv_dtype)
        # The (request, split) items are planned only where a launch of this model can read them (round 6; the reference's
        # init_forward_metadata computes only what its one kernel reads, triton_backend.py:48-68): a shape the range kernel
        # refuses, or a layer with a logit soft-cap (Gemma-2 style; no Llama-3 / Mllama layer has one).  Without them a
        # plan is the range section alone and nothing of a step depends on the host's bound on sum(seq_lens).
        layers = [m for m in getattr(model_runner, "model", nn.Module()).modules() if isinstance(m, RadixAttention)]
        self.plan_items = self.decode_ranges <= 0 or any(m.logit_cap > 0 for m in layers)

    # ---------------------------------------------------------------- launch planning
    def _head_groups(self, dtype: torch.dtype) -> int:
        vec = 4 if dtype == torch.float32 else 8
        rpl = 64 // (self.head_dim // vec)
        hh = 1
        while hh * 2 <= rpl and self.num_kv_head % (hh * 2) == 0:
            hh *= 2
        return self.num_kv_head // hh

    def _plan_chunk(self, kv_tokens: int, dtype: torch.dtype) -> int:
        """Split size of the (request, split) items from the host's bound on sum(seq_lens): about one item per CU, between
        MIN_CHUNK and MAX_CHUNK.  (Until round 5 a near-uniform batch, known from an advisory hint of the longest request,
        was left unsplit; the range geometry made that the default path's property - equal lengths are cut between
        requests - and the hint left with it: profiles/NOTES.md, "Round 6".)"""
        groups = self._head_groups(dtype)
        chunk = _pow2_floor(max(kv_tokens, 1) * groups // self.TARGET_ITEMS)
        return max(self.MIN_CHUNK, min(self.MAX_CHUNK, chunk))

    def _ensure_workspace(self, nbytes: int) -> torch.Tensor:
        if self._workspace.numel() < nbytes:
            self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._workspace

    def _graph_slots(self, bs: int) -> int:
        """(request, split) items a captured launch of bucket `bs` covers; 0 where no launch of this model reads items"""
        if not (self.plan_items or self._ranges_for(bs, self.max_context_len) <= 0):
            return 0
        by_len = bs * -(-self.max_context_len // self.MIN_CHUNK)
        return min(by_len, max(self.GRAPH_SLOTS_FLOOR, self.GRAPH_SLOTS_PER_REQ * bs) + bs)

    def _fit_chunk(self, chunk: int, bs: int, kv_tokens: int, max_len: int, max_slots: int) -> int:
        """the split size doubles until the step's items fit the slots the launch covers"""
        while _native.decode_plan_slots(bs, max_len, chunk, kv_tokens) > max_slots:
            chunk *= 2
        return chunk

    def _ranges_for(self, bs: int, max_len: int) -> int:
        """pieces of the range geometry for a step (0: its line does not fit the plan's int32 positions)"""
        return self.decode_ranges if bs * (max_len + _native.RANGE_REQUEST_COST) < 2 ** 31 - 1 else 0

    def _build_plans(self, plans, bs, windows, max_len, max_slots=None, ranges=None):
        """One plan per kv window (self-attention lens; encoder lens for cross-attention - the reference keeps two
        flashinfer wrappers for the same reason, flashinfer_backend.py:121-131; the sliding-window layers' lens).
        windows: per plan (lens tensor or None, host bound on their sum).  Built once per step, read by every layer's
        launch.  A plan carries the range geometry (`ranges` pieces) where the backend's shape has one, and the (request,
        split) items - with their own split size - only where a launch can need them (self.plan_items, or a step whose
        line does not fit the range section).  Returns per plan (tensor, max_slots, smallest chunk, ranges) - the three
        numbers every launch on that plan is given - and the partial slots the workspace must hold."""
        out, need_slots = [], 1
        if ranges is None:
            ranges = self._ranges_for(bs, max_len)
        with_items = self.plan_items or ranges <= 0
        for i, (lens, kv_tokens) in enumerate(windows):
            if lens is None:
                out.append(None)
                continue
            if not with_items:
                chunk, slots = self.MIN_CHUNK, 0
            else:
                chunk = self._plan_chunk(kv_tokens, self.kv_dtype)
                if max_slots is None:          # eager: exactly what this step can need
                    slots = max(1, _native.decode_plan_slots(bs, max_len, chunk, kv_tokens))
                else:                          # graph replay: the captured launch's capacity is fixed
                    slots = max_slots
                    chunk = self._fit_chunk(chunk, bs, kv_tokens, max_len, slots)
            need = _native.decode_plan_bytes(bs, max_len, chunk, slots, ranges) // 4
            if plans[i].numel() < need:
                plans[i] = torch.empty(need, dtype=torch.int32, device=self.device)
            _native.decode_plan(plans[i], lens, max_len, chunk, slots, ranges)
            if with_items:
                self._watch_plan(plans[i], slots)
            # third field: the smallest split size this plan buffer may carry when the launch runs - the
            # step's own chunk (eager), MIN_CHUNK under graph replay (a later step's plan may use any size)
            out.append((plans[i], slots, chunk if max_slots is None else self.MIN_CHUNK, ranges))
            need_slots = max(need_slots, slots, bs + ranges if ranges else 0)
        return tuple(out), need_slots

    # ---------------------------------------------------------------- plan overflow (an understated seq_lens_sum)
    def _watch_plan(self, plan: torch.Tensor, slots: int) -> None:
        """The plan kernel records how many items the device-side lengths need; more than the launch covers means
        splits were dropped (wrong logits).  The header comes back through a 16-byte asynchronous copy: earlier
        plans whose copy has landed are checked here, this one at the next plan or in check_plans().
        With a `plan_step` (the overlap worker sets one per step, tp_worker_client.py) nothing is checked HERE: this is
        the forward thread, and an error raised in it would end the thread instead of reaching the scheduler with the
        step it belongs to - the entry carries the step and check_plans(upto=step) on the scheduler's side
        reports it when that step's results are resolved."""
        if self.plan_step is None:
            self.check_plans(wait=len(self._plan_checks) >= 64)      # (a bound on the copies in flight)
        with self._plan_lock:
            host, ev = self._plan_hosts.pop() if self._plan_hosts else (
                torch.empty(_native.PLAN_HEADER_WORDS, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        host.copy_(plan[:_native.PLAN_HEADER_WORDS], non_blocking=True)
        ev.record()
        with self._plan_lock:
            self._plan_checks.append((host, ev, slots, self.plan_step))
        if self.strict_plan_check and self.plan_step is None:
            self.check_plans(wait=True)

    def check_plans(self, wait: bool = True, upto: Optional[int] = None) -> None:
        """Raise RuntimeError if a decode plan built so far was cut short.  wait=False looks only at header copies
        that have already completed (no synchronisation).  upto=N: only the plans of steps <= N (entries tagged through
        `plan_step`); later steps' entries stay for their own check."""
        pending = []
        err = None
        with self._plan_lock:
            checks, self._plan_checks = self._plan_checks, []
        for entry in checks:
            host, ev, slots, step = entry
            if upto is not None and step is not None and step > upto:
                pending.append(entry)
                continue
            if wait:
                ev.synchronize()
            elif not ev.query():
                pending.append(entry)
                continue
            err = err or _native.decode_plan_overflow(host.tolist(), slots)
            with self._plan_lock:
                self._plan_hosts.append((host, ev))
        with self._plan_lock:
            self._plan_checks = pending + self._plan_checks
        if err:
            raise RuntimeError(err)

    def _windows(self, bs, seq_lens, seq_lens_sum, encoder_lens, encoder_sum):
        """(lens, bound on their sum) of the three kv windows of a decode step"""
        cap = bs * self.max_context_len if self.pool_tokens is None else min(bs * self.max_context_len,
                                                                             self.pool_tokens)
        seq_sum = max(1, min(int(seq_lens_sum), cap))
        enc = (None, 0)
        if encoder_lens is not None:
            enc = (encoder_lens, max(1, cap if encoder_sum is None else min(int(encoder_sum), cap)))
        win = (None, 0)
        if self._window is not None:
            win = (self._window[0], max(1, min(seq_sum, bs * (self.sliding_window_size + 1))))
        return (seq_lens, seq_sum), enc, win

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
            enc = forward_batch.encoder_lens if self.is_encoder_decoder else None
            enc_sum = sum(forward_batch.encoder_lens_cpu) if forward_batch.encoder_lens_cpu else None
            self._window = self._window_of(forward_batch.seq_lens)
            plans, slots = self._build_plans(self._plans, bs, self._windows(
                bs, forward_batch.seq_lens, forward_batch.seq_lens_sum, enc, enc_sum), max_len)
            ws = self._ensure_workspace(_native.decode_workspace_bytes(bs, self.num_head, self.v_head_dim, max_len,
                                                                       self.MIN_CHUNK, slots))
            self.forward_metadata = (self.MIN_CHUNK, max_len, ws, plans)
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
        """Static buffers for graph replay (triton_backend.py:70-80 allocates its static attn_logits the same
        way - but as [max_bs, heads, max_context_len]).  ONE partials workspace and ONE triple of plan buffers
        serve every batch-size bucket: a bucket's captured launches cover _graph_slots(bs) work items, and the
        split size is not part of the capture (it travels in the plan), so the scratch is bounded by the slots
        of the largest bucket whatever the model's context length is."""
        self.cuda_graph_max_seq_len = self.max_context_len
        # the piece count is launch geometry: ONE value for every bucket, decided for the largest (a smaller bucket's line fits too)
        self._graph_ranges = ranges = self._ranges_for(max_bs, self.cuda_graph_max_seq_len)
        slots = self._graph_slots(max_bs)
        self._graph_ws = torch.empty(_native.decode_workspace_bytes(max_bs, self.num_head, self.v_head_dim,
                                                                    self.cuda_graph_max_seq_len, self.MIN_CHUNK, slots, ranges),
                                     dtype=torch.uint8, device=self.device)
        n = _native.decode_plan_bytes(max_bs, self.cuda_graph_max_seq_len, self.MIN_CHUNK, slots, ranges) // 4
        self._graph_plans = [torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(3)]
        self._graph_max_bs = max_bs
        if self.sliding_window_size is not None:
            self._graph_window = tuple(torch.ones(max_bs, dtype=torch.int32, device=self.device)
                                       for _ in range(2))

    def graph_scratch_bytes(self) -> int:
        """attention scratch held for graph replay (workspace + plan buffers), all buckets together"""
        return self._graph_ws.numel() + sum(p.numel() * 4 for p in self._graph_plans)

    def _graph_metadata(self, bs, seq_lens, seq_lens_sum, encoder_lens):
        assert bs <= self._graph_max_bs
        plans, _ = self._build_plans(self._graph_plans, bs, self._windows(bs, seq_lens, seq_lens_sum, encoder_lens,
                                                                          None),
                                     self.cuda_graph_max_seq_len, self._graph_slots(bs), ranges=self._graph_ranges)
        # MIN_CHUNK: the smallest split size a replayed plan may carry (the merge launch is always captured)
        self.forward_metadata = (self.MIN_CHUNK, self.cuda_graph_max_seq_len, self._graph_ws, plans)

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens,
                                                 encoder_lens, forward_mode, spec_info=None):
        assert forward_mode.is_decode(), "only decode is captured"
        assert spec_info is None, "speculative decoding is out of scope"
        self._window = self._window_of(seq_lens, self._graph_window, bs)
        self._graph_metadata(bs, seq_lens, bs * self.get_cuda_graph_seq_len_fill_value(), encoder_lens)

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum,
                                                encoder_lens, forward_mode, spec_info=None,
                                                seq_lens_cpu=None):
        # the launches are static and read seq_lens / req_pool_indices from the graph's static input
        # buffers; only the split plans (static buffers, fixed addresses) are rebuilt for this step's
        # lengths, ahead of the replay - where the reference recomputes start_loc / kv_indices
        # (triton_backend.py:103-113, flashinfer_backend.py:330-373).  The split size is chosen HERE, per
        # step, from seq_lens_sum (it is plan data, not launch geometry).
        self._window = self._window_of(seq_lens[:bs], self._graph_window, bs)
        self._graph_metadata(bs, seq_lens[:bs], seq_lens_sum, None if encoder_lens is None else encoder_lens[:bs])

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
        entry = plans[2] if self._is_windowed(layer) else (plans[1] if layer.is_cross_attention else plans[0])
        plan, slots, chunk, ranges = entry if entry is not None else (None, None, chunk, 0)
        if plan is not None and slots == 0 and layer.logit_cap > 0:
            # (the library would refuse the launch anyway - SP_ERR_INVALID_ARG - this says why)
            raise RuntimeError(f"layer {layer.layer_id} carries a logit soft-cap, which the range kernel does not take, but this "
                               "backend plans no (request, split) items: it found no such layer in model_runner.model when "
                               "it was constructed (HipAttnBackend.plan_items) - construct it after the model")
        kb, vb = forward_batch.token_to_kv_pool.get_kv_buffer(layer.layer_id)
        _native.decode_attention(
            o.view(-1, layer.tp_q_head_num, layer.v_head_dim),
            q.view(-1, layer.tp_q_head_num, layer.qk_head_dim), kb, vb,
            forward_batch.req_to_token_pool.req_to_token, forward_batch.req_pool_indices, seq_lens,
            layer.scaling, layer.logit_cap, max_len, chunk, ws, kv_start, plan,
            k_scale=k_scale, v_scale=v_scale, max_slots=slots, ranges=ranges)
        return o
