"""ForwardMode / ForwardBatch / ModelWorkerBatch - the per-step argument pack.

Mirrors model_executor/forward_info.py:18-66 (ForwardMode), 69-81 (CaptureHiddenMode), 84-287
(ForwardBatch + init_new), 400-471 (positions) and scheduler/schedule_batch.py:1481-1543
(ModelWorkerBatch) for the fields the attention hot path reads, and the log-prob request fields
the logits processor / sampler read.  Fields that only feed out-of-scope subsystems (DP attention,
speculative decoding, toppings, mrope) are kept as inert attributes so call sites written against the reference still construct them."""
import threading
from dataclasses import dataclass
from enum import IntEnum, auto
from typing import Any, List, Optional

import torch

from . import _native


class ForwardMode(IntEnum):
    EXTEND = auto()        # prefill, possibly on top of a cached prefix
    DECODE = auto()        # one token per request
    MIXED = auto()         # chunked prefill + running decodes in one extend batch
    IDLE = auto()
    TARGET_VERIFY = auto()
    DRAFT_EXTEND = auto()
    DUMMY_FIRST = auto()

    def is_extend(self):
        return self == ForwardMode.EXTEND or self == ForwardMode.MIXED

    def is_decode(self):
        return self == ForwardMode.DECODE

    def is_mixed(self):
        return self == ForwardMode.MIXED

    def is_idle(self):
        return self == ForwardMode.IDLE

    def is_target_verify(self):
        return self == ForwardMode.TARGET_VERIFY

    def is_draft_extend(self):
        return self == ForwardMode.DRAFT_EXTEND

    def is_cuda_graph(self):
        return self.is_decode() or self.is_target_verify() or self.is_idle()

    def is_dummy_first(self):
        return self == ForwardMode.DUMMY_FIRST

    def is_decode_or_idle(self):
        return self.is_decode() or self.is_idle()


class CaptureHiddenMode(IntEnum):
    NULL = auto()
    FULL = auto()
    LAST = auto()

    def need_capture(self):
        return self != CaptureHiddenMode.NULL

    def is_full(self):
        return self == CaptureHiddenMode.FULL

    def is_last(self):
        return self == CaptureHiddenMode.LAST


@dataclass
class ModelWorkerBatch:
    """scheduler/schedule_batch.py:1481-1543 - what the scheduler hands the TP worker."""
    bid: int
    forward_mode: ForwardMode
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor
    seq_lens: torch.Tensor
    # (the reference's order; everything from here on is required upstream and passed by keyword there - here the
    # defaults let a caller leave out what the path does not read)
    seq_lens_cpu: Optional[torch.Tensor] = None
    out_cache_loc: torch.Tensor = None
    seq_lens_sum: int = None
    return_logprob: bool = False
    top_logprobs_nums: Optional[List[int]] = None
    token_ids_logprobs: Optional[List[List[int]]] = None
    global_num_tokens: Optional[List[int]] = None
    global_num_tokens_for_logprob: Optional[List[int]] = None
    can_run_dp_cuda_graph: bool = False
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[List[int]] = None
    extend_prefix_lens: Optional[List[int]] = None
    extend_logprob_start_lens: Optional[List[int]] = None
    extend_input_logprob_token_ids: Optional[torch.Tensor] = None
    multimodal_inputs: Optional[List[Any]] = None
    encoder_cached: Optional[List[bool]] = None
    encoder_lens: Optional[torch.Tensor] = None
    encoder_lens_cpu: Optional[List[int]] = None
    encoder_out_cache_loc: Optional[torch.Tensor] = None
    toppings_paths: Optional[List[str]] = None
    sampling_info: Any = None
    input_embeds: Optional[torch.Tensor] = None
    spec_algorithm: Any = None
    spec_info: Optional[Any] = None
    capture_hidden_mode: CaptureHiddenMode = CaptureHiddenMode.NULL
    launch_done: Optional[threading.Event] = None
    # stand-in for the (un-hosted) vision tower: projected + flattened vision states of the requests
    # whose encoder is not cached, [sum(encoder_lens_need), hidden] (mllama.py:966-979)
    encoder_states: Optional[torch.Tensor] = None


@dataclass
class ForwardBatch:
    """model_executor/forward_info.py:84-177: the reference's fields in the reference's order
    (tests/test_api_surface.py pins names, order and defaults against tests/golden/api_surface.json)."""
    forward_mode: ForwardMode
    batch_size: int
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor
    seq_lens: torch.Tensor
    out_cache_loc: torch.Tensor
    seq_lens_sum: int
    seq_lens_cpu: Optional[torch.Tensor] = None
    return_logprob: bool = False
    top_logprobs_nums: Optional[List[int]] = None
    token_ids_logprobs: Optional[List[List[int]]] = None
    # logprob post-processing switches (read by the logits processor: llama.py compute_logprobs)
    temp_scaled_logprobs: bool = False
    temperature: torch.Tensor = None
    top_p_normalized_logprobs: bool = False
    top_p: torch.Tensor = None
    positions: torch.Tensor = None
    extend_num_tokens: Optional[int] = None
    extend_seq_lens: Optional[torch.Tensor] = None
    extend_prefix_lens: Optional[torch.Tensor] = None
    extend_start_loc: Optional[torch.Tensor] = None
    extend_prefix_lens_cpu: Optional[List[int]] = None
    extend_seq_lens_cpu: Optional[List[int]] = None
    extend_logprob_start_lens_cpu: Optional[List[int]] = None
    extend_input_logprob_token_ids_gpu: Optional[torch.Tensor] = None
    mm_inputs: Optional[List[Any]] = None
    encoder_cached: Optional[List[bool]] = None
    encoder_lens: Optional[torch.Tensor] = None
    encoder_lens_cpu: Optional[List[int]] = None
    encoder_out_cache_loc: Optional[torch.Tensor] = None
    topping_paths: Optional[List[str]] = None
    input_embeds: Optional[torch.Tensor] = None
    sampling_info: Any = None
    req_to_token_pool: Any = None
    token_to_kv_pool: Any = None
    attn_backend: Any = None
    # DP attention (inert here: the reference's dp_size flag is inert too)
    global_num_tokens_cpu: Optional[List[int]] = None
    global_num_tokens_gpu: Optional[torch.Tensor] = None
    global_num_tokens_for_logprob_cpu: Optional[List[int]] = None
    global_num_tokens_for_logprob_gpu: Optional[torch.Tensor] = None
    dp_local_start_pos: Optional[torch.Tensor] = None
    dp_local_num_tokens: Optional[torch.Tensor] = None
    gathered_buffer: Optional[torch.Tensor] = None
    can_run_dp_cuda_graph: bool = False
    spec_info: Any = None
    spec_algorithm: Any = None
    capture_hidden_mode: CaptureHiddenMode = None
    padded_static_len: int = -1
    mrope_positions: torch.Tensor = None
    # ---- not in the reference (kept behind its fields)
    encoder_states: Optional[torch.Tensor] = None    # see ModelWorkerBatch

    @classmethod
    def init_new(cls, batch: ModelWorkerBatch, model_runner) -> "ForwardBatch":
        """forward_info.py:179-287 (decode: positions = clamp(seq_lens - 1); extend: positions and
        extend_start_loc from the HIP twin of compute_position_triton).

        Provenance: this is glue whose statement order is dictated by the field contract - which ModelWorkerBatch
        field lands in which ForwardBatch field, and what is derived when - so it follows the reference's init_new
        assignment by assignment, minus its DP-attention buffers, mrope and toppings branches (out of scope), and
        with the two positions helpers replaced by the HIP entry points."""
        device = model_runner.device
        ret = cls(
            forward_mode=batch.forward_mode, batch_size=len(batch.seq_lens), input_ids=batch.input_ids,
            req_pool_indices=batch.req_pool_indices, seq_lens=batch.seq_lens,
            out_cache_loc=batch.out_cache_loc, mm_inputs=batch.multimodal_inputs,
            encoder_cached=batch.encoder_cached, encoder_lens=batch.encoder_lens,
            encoder_lens_cpu=batch.encoder_lens_cpu, encoder_out_cache_loc=batch.encoder_out_cache_loc,
            seq_lens_sum=batch.seq_lens_sum,
            return_logprob=batch.return_logprob,
            top_logprobs_nums=batch.top_logprobs_nums, token_ids_logprobs=batch.token_ids_logprobs,
            can_run_dp_cuda_graph=batch.can_run_dp_cuda_graph, topping_paths=batch.toppings_paths,
            sampling_info=batch.sampling_info, req_to_token_pool=model_runner.req_to_token_pool,
            token_to_kv_pool=model_runner.token_to_kv_pool, attn_backend=model_runner.attn_backend,
            spec_algorithm=batch.spec_algorithm, spec_info=batch.spec_info,
            capture_hidden_mode=batch.capture_hidden_mode, input_embeds=batch.input_embeds,
            encoder_states=batch.encoder_states,
            extend_input_logprob_token_ids_gpu=(None if batch.extend_input_logprob_token_ids is None else
                                                batch.extend_input_logprob_token_ids.to(device, non_blocking=True)))
        if batch.global_num_tokens is not None:
            raise NotImplementedError("DP attention (global_num_tokens) is out of scope of this path")
        # nothing downstream of this seam reads these two (LogitsProcessor returns no hidden states, the model embeds
        # input_ids - the reference's own ModelRunner.forward_extend, model_runner.py:516-529, never hands input_embeds to
        # the model either): refuse them here instead of silently serving something else
        if batch.capture_hidden_mode not in (None, CaptureHiddenMode.NULL):
            raise NotImplementedError("returning hidden states (capture_hidden_mode != NULL) is out of scope of this path")
        if batch.input_embeds is not None:
            raise NotImplementedError("input_embeds is out of scope of this path: the model embeds input_ids")
        if ret.forward_mode.is_idle():
            ret.positions = torch.empty((0,), device=device)
            return ret
        if ret.spec_info is not None and getattr(ret.spec_info, "positions", None) is not None:
            ret.positions = ret.spec_info.positions
        if ret.seq_lens_cpu is None:
            ret.seq_lens_cpu = batch.seq_lens_cpu
        if ret.forward_mode.is_decode():
            if ret.positions is None:
                ret.positions = clamp_position(batch.seq_lens)
        else:
            ret.extend_seq_lens = torch.tensor(batch.extend_seq_lens, dtype=torch.int32).to(
                device, non_blocking=True)
            ret.extend_prefix_lens = torch.tensor(batch.extend_prefix_lens, dtype=torch.int32).to(
                device, non_blocking=True)
            ret.extend_num_tokens = batch.extend_num_tokens
            positions, ret.extend_start_loc = compute_position(
                ret.extend_prefix_lens, ret.extend_seq_lens, ret.extend_num_tokens)
            if ret.positions is None:
                ret.positions = positions
            ret.extend_prefix_lens_cpu = batch.extend_prefix_lens
            ret.extend_seq_lens_cpu = batch.extend_seq_lens
            ret.extend_logprob_start_lens_cpu = batch.extend_logprob_start_lens
        return ret

    def merge_mm_inputs(self):
        """forward_info.py:289-311: the batch's multimodal inputs folded into the first one (None: text only)"""
        have = [x for x in (self.mm_inputs or []) if x is not None]
        if not have:
            return None
        for other in have[1:]:
            have[0].merge(other)
        return have[0]

    def contains_image_inputs(self) -> bool:
        return any(x is not None and x.contains_image_inputs() for x in (self.mm_inputs or []))

    def contains_audio_inputs(self) -> bool:
        return any(x is not None and x.contains_audio_inputs() for x in (self.mm_inputs or []))

    def contains_mm_inputs(self) -> bool:
        return self.contains_audio_inputs() or self.contains_image_inputs()


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor,
                     extend_seq_lens_sum: int):
    """compute_position_triton - forward_info.py:400-449."""
    return _native.compute_position(extend_prefix_lens, extend_seq_lens, extend_seq_lens_sum)


def clamp_position(seq_lens: torch.Tensor) -> torch.Tensor:
    """clamp_position - forward_info.py:469-471."""
    return _native.clamp_position(seq_lens)
