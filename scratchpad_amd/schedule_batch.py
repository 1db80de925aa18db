"""Scheduler-side producer of the hot path's inputs (boundary, not the scheduler itself).

Mirrors the slot/table bookkeeping of scheduler/schedule_batch.py: ``prepare_for_extend``
(901-1071: request rows, cached-prefix copy, slot allocation, req_to_token scatter by the HIP twin
of write_req_to_token_pool_triton), ``prepare_for_decode`` (1230-1308: seq_lens += 1, alloc(bs),
req_to_token[req, seq_len-1] = slot), ``mix_with_running`` (1073-1101) and
``get_model_worker_batch`` (1399-1459), the prefix-cache hand-off (``Req.init_next_round_input``
472-492, ``alloc_token_slots`` 727-754 evicting through ``tree_cache``), ``check_decode_mem`` /
``retract_decode`` (1103-1211) and ``filter_batch`` / ``merge_batch`` (1309-1397).  Admission
policy, grammars, penalizers and detokenisation stay with the reference's scheduler (out of scope).
"""
import threading
from dataclasses import dataclass
from typing import Any, List, Optional, Set, Tuple, Union

import torch

from . import _native
from .forward_info import CaptureHiddenMode, ForwardMode, ModelWorkerBatch
from .pool import ReqToTokenPool, TokenToKVPoolAllocator
from .radix_cache import BasePrefixCache, ChunkCache

bid = 0


class BaseFinishReason:
    """scheduler/schedule_batch.py:33-90: why a request stopped (``to_json`` is what the detokenizer side reads)."""
    type = "base"

    def __init__(self, is_error: bool = False):
        self.is_error = is_error

    def _payload(self) -> dict:
        return {}

    def to_json(self):
        return {"type": self.type, **self._payload()}


class FINISH_MATCHED_TOKEN(BaseFinishReason):
    type = "stop"

    def __init__(self, matched: Union[int, List[int]]):
        super().__init__()
        self.matched = matched

    def _payload(self):
        return {"matched": self.matched}


class FINISH_MATCHED_STR(FINISH_MATCHED_TOKEN):
    def __init__(self, matched: str):
        super().__init__(matched)


class FINISH_LENGTH(BaseFinishReason):
    type = "length"

    def __init__(self, length: int):
        super().__init__()
        self.length = length

    def _payload(self):
        return {"length": self.length}


class FINISH_ABORT(BaseFinishReason):
    type = "abort"

    def __init__(self, message="Unknown error", status_code=None, err_type=None):
        super().__init__(is_error=True)
        self.message, self.status_code, self.err_type = message, status_code, err_type

    def _payload(self):
        return {"message": self.message, "status_code": self.status_code, "err_type": self.err_type}


class Req:
    """scheduler/schedule_batch.py:287-593, the constructor form and the attributes the producers below (and the
    reference's scheduler around them) read.  Parameters up to ``eos_token_ids`` are the reference's, in its order;
    the keyword-only ones behind them lay a request out directly (tests, benches)."""

    def __init__(self, rid: str, origin_input_text: str = "", origin_input_ids: Tuple[int] = (), sampling_params=None,
                 return_logprob: bool = False, top_logprobs_num: int = 0, token_ids_logprob: Optional[List[int]] = None,
                 stream: bool = False, origin_input_ids_unpadded: Optional[Tuple[int]] = None,
                 topping_path: Optional[str] = None, input_embeds=None, session_id: Optional[str] = None,
                 custom_logit_processor: Optional[str] = None, return_hidden_states: bool = False,
                 eos_token_ids: Optional[Set[int]] = None, *, output_ids: Optional[List[int]] = None,
                 prefix_indices=None, num_image_tokens: Optional[int] = None, multimodal_inputs=None,
                 logprob_start_len: int = 0):
        self.rid = rid
        self.origin_input_text = origin_input_text
        self.origin_input_ids = origin_input_ids
        self.origin_input_ids_unpadded = origin_input_ids_unpadded if origin_input_ids_unpadded else origin_input_ids
        self.output_ids: List[int] = [] if output_ids is None else output_ids
        self.session_id, self.input_embeds = session_id, input_embeds
        self.sampling_params = sampling_params           # sampler.SamplingParams; None = greedy
        self.custom_logit_processor = custom_logit_processor
        self.return_hidden_states = return_hidden_states
        self.topping_path = topping_path
        self.req_pool_idx: Optional[int] = None
        # finishing (check_finished)
        self.tokenizer = None
        self.finished_reason = None
        self.to_abort, self.to_abort_message = False, "Unknown error"
        self.stream = stream
        self.eos_token_ids = eos_token_ids
        self.decoded_text = ""
        self.grammar = None
        # prefix-cache hand-off: cached KV slots (radix-cache hit) and the node the request holds a lock on
        self.prefix_indices = [] if prefix_indices is None else prefix_indices
        self.last_node = None
        self.last_node_global = None
        self.fill_ids_override: Optional[List[int]] = None       # chunked prefill: the tokens of this round
        self.extend_input_len_override: Optional[int] = None    # pinned by the scheduler (1 for a running request)
        self.is_chunked = 0
        self.is_retracted = False
        self.cached_tokens = self.already_computed = 0
        # encoder-decoder models: the first num_image_tokens ids of origin_input_ids are the image pad ids
        # (mllama.py pad_input_ids 803-816; MultimodalInputs.num_image_tokens)
        self.num_image_tokens = num_image_tokens
        self.multimodal_inputs = multimodal_inputs
        # log-prob requests (schedule_batch.py:369-387): logprob_start_len is an index into origin_input_ids;
        # extend_logprob_start_len is the same, relative to this round's extend part (set by prepare_for_extend)
        self.return_logprob = return_logprob
        self.logprob_start_len = logprob_start_len
        self.top_logprobs_num = top_logprobs_num
        self.token_ids_logprob = token_ids_logprob
        self.extend_logprob_start_len = 0
        self.temp_scaled_logprobs = self.top_p_normalized_logprobs = False

    def __repr__(self):
        return f"Req(rid={self.rid}, input_ids={self.origin_input_ids}, output_ids={self.output_ids})"

    @property
    def seqlen(self):
        return len(self.origin_input_ids) + len(self.output_ids)

    @property
    def fill_ids(self) -> List[int]:
        if self.fill_ids_override is not None:
            return self.fill_ids_override
        return list(self.origin_input_ids) + self.output_ids

    @fill_ids.setter
    def fill_ids(self, ids: Optional[List[int]]):
        self.fill_ids_override = ids

    def extend_image_inputs(self, image_inputs):
        if self.multimodal_inputs is None:
            self.multimodal_inputs = image_inputs
        else:
            self.multimodal_inputs.merge(image_inputs)

    def finished(self) -> bool:
        return self.finished_reason is not None

    def adjust_max_prefix_ids(self):
        """schedule_batch.py:494-511: a cached prefix always leaves one token to compute (the step must produce
        logits), and may not swallow positions whose input logprobs are asked for."""
        self.fill_ids_override = None
        ids = self.fill_ids
        limit = len(ids) - 1
        if self.return_logprob:
            limit = min(limit, self.logprob_start_len)
        return ids[:max(limit, 0)]

    def init_next_round_input(self, tree_cache: Optional[BasePrefixCache] = None, enable_hierarchical_cache=False):
        """schedule_batch.py:472-492: look the prompt up in the prefix cache."""
        self.fill_ids_override = None
        self.extend_input_len_override = None
        if tree_cache is not None:
            if enable_hierarchical_cache:
                raise NotImplementedError("hierarchical (host-backed) prefix cache: out of scope (SURVEY section 8)")
            self.prefix_indices, self.last_node = tree_cache.match_prefix(rid=self.rid, key=self.adjust_max_prefix_ids())

    def check_finished(self):
        """schedule_batch.py:525-571: length, then stop tokens (sampling params / request / tokenizer), then stop strings."""
        if self.finished():
            return
        if self.to_abort:
            self.finished_reason = FINISH_ABORT(message=self.to_abort_message)
            return
        params = self.sampling_params
        max_new = getattr(params, "max_new_tokens", None)
        if max_new is not None and len(self.output_ids) >= max_new:
            self.finished_reason = FINISH_LENGTH(length=max_new)
            return
        last = self.output_ids[-1]
        if not getattr(params, "ignore_eos", False):
            stops = set(getattr(params, "stop_token_ids", None) or ()) | set(self.eos_token_ids or ())
            if self.tokenizer is not None:
                stops.add(self.tokenizer.eos_token_id)
                stops |= set(getattr(self.tokenizer, "additional_stop_token_ids", None) or ())
            if last in stops:
                self.finished_reason = FINISH_MATCHED_TOKEN(matched=last)
                return
        stop_strs = getattr(params, "stop_strs", None) or ()
        if stop_strs:
            tail = self.tokenizer.decode(self.output_ids[-(params.stop_str_max_len + 1):])
            for s in stop_strs:
                if s in tail or s in self.decoded_text:
                    self.finished_reason = FINISH_MATCHED_STR(matched=s)
                    return

    def reset_for_retract(self):
        """schedule_batch.py:573-584"""
        self.prefix_indices = []
        self.last_node = None
        self.fill_ids_override = None
        self.extend_input_len_override = None
        self.req_pool_idx = None
        self.extend_logprob_start_len = 0
        self.is_retracted = True
        self.is_chunked = 0
        self.already_computed = 0

    @property
    def extend_input_len(self) -> int:
        """tokens this request contributes to the extend step: fill_ids behind the cached prefix, unless the
        scheduler has pinned it (mix_with_running sets 1 for a running request, schedule_batch.py:1077-1079)"""
        if self.extend_input_len_override is not None:
            return self.extend_input_len_override
        return len(self.fill_ids) - self.prefix_len

    @extend_input_len.setter
    def extend_input_len(self, n: Optional[int]):
        self.extend_input_len_override = n

    @property
    def prefix_len(self) -> int:
        return 0 if self.prefix_indices is None else len(self.prefix_indices)


@dataclass
class ScheduleBatch:
    """scheduler/schedule_batch.py:595-680: the reference's fields in the reference's order (so its keyword and
    positional constructions bind), then the two this build adds.  Fields that only feed out-of-scope subsystems
    (DP attention, speculative decoding, custom logit processors) are inert."""
    reqs: List[Req]
    req_to_token_pool: ReqToTokenPool = None
    token_to_kv_pool_allocator: TokenToKVPoolAllocator = None
    tree_cache: BasePrefixCache = None
    model_config: Any = None
    forward_mode: ForwardMode = None
    enable_overlap: bool = False
    batch_is_full: bool = False
    launch_done: Optional[threading.Event] = None
    # None = every request greedy; sampler.SamplingBatchInfo.from_schedule_batch(batch, vocab) otherwise
    sampling_info: Any = None
    next_batch_sampling_info: Any = None
    input_ids: torch.Tensor = None
    input_embeds: torch.Tensor = None
    req_pool_indices: torch.Tensor = None
    seq_lens: torch.Tensor = None
    out_cache_loc: torch.Tensor = None
    output_ids: torch.Tensor = None
    seq_lens_sum: int = None
    global_num_tokens: Optional[List[int]] = None
    global_num_tokens_for_logprob: Optional[List[int]] = None
    can_run_dp_cuda_graph: bool = False
    # per-batch flags / per-request logprob requests: carried through merge_batch / mix_with_running
    return_logprob: bool = False
    top_logprobs_nums: Optional[List[int]] = None
    token_ids_logprobs: Optional[List[List[int]]] = None
    temp_scaled_logprobs: bool = False
    top_p_normalized_logprobs: bool = False
    prefix_lens: List[int] = None
    extend_lens: List[int] = None
    extend_num_tokens: int = None
    decoding_reqs: List[Req] = None
    extend_logprob_start_lens: List[int] = None
    extend_input_logprob_token_ids: Optional[torch.Tensor] = None
    encoder_cached: Optional[List[bool]] = None
    encoder_lens: Optional[torch.Tensor] = None
    encoder_lens_cpu: Optional[List[int]] = None
    encoder_out_cache_loc: Optional[torch.Tensor] = None
    has_stream: bool = False
    has_grammar: bool = False
    device: str = "cuda"
    spec_algorithm: Any = None
    spec_info: Optional[Any] = None
    enable_custom_logit_processor: bool = False
    return_hidden_states: bool = False
    # ---- not in the reference
    # None: taken from model_config.is_encoder_decoder, as the reference reads it (schedule_batch.py:1274, 1334)
    is_encoder_decoder: Optional[bool] = None

    def __post_init__(self):
        if self.is_encoder_decoder is None:
            self.is_encoder_decoder = bool(getattr(self.model_config, "is_encoder_decoder", False))
        if self.seq_lens_sum is None:
            self.seq_lens_sum = 0
        if self.extend_logprob_start_lens is None:
            self.extend_logprob_start_lens = []
        # a batch built directly (tests, benches) gets the flag init_new derives, schedule_batch.py:693
        self.return_logprob = self.return_logprob or any(r.return_logprob for r in self.reqs)

    @classmethod
    def init_new(cls, reqs: List[Req], req_to_token_pool: ReqToTokenPool,
                 token_to_kv_pool_allocator: TokenToKVPoolAllocator, tree_cache: BasePrefixCache, model_config,
                 enable_overlap: bool, spec_algorithm, enable_custom_logit_processor: bool = False):
        """schedule_batch.py:682-709.  (The last flag has a default here only: the reference's own idle-batch call,
        scheduler.py:1711-1719, leaves it out.)"""
        return cls(reqs=reqs, req_to_token_pool=req_to_token_pool, token_to_kv_pool_allocator=token_to_kv_pool_allocator,
                   tree_cache=tree_cache, model_config=model_config, enable_overlap=enable_overlap,
                   return_logprob=any(r.return_logprob for r in reqs),
                   has_stream=any(getattr(r, "stream", False) for r in reqs),
                   has_grammar=any(getattr(r, "grammar", None) for r in reqs),
                   device=req_to_token_pool.device, spec_algorithm=spec_algorithm,
                   enable_custom_logit_processor=enable_custom_logit_processor,
                   return_hidden_states=any(getattr(r, "return_hidden_states", False) for r in reqs))

    def batch_size(self):
        return len(self.reqs)

    def is_empty(self):
        return len(self.reqs) == 0

    def alloc_req_slots(self, num_reqs: int):
        idx = self.req_to_token_pool.alloc(num_reqs)
        if idx is None:
            raise RuntimeError("Out of memory. Please set a smaller number for "
                               "`--max-running-requests`.")
        return idx

    def alloc_token_slots(self, num_tokens: int, backup_state: bool = False):
        """schedule_batch.py:728-752; ``backup_state``: also return the allocator's free list as it was before."""
        if self.tree_cache is not None and self.token_to_kv_pool_allocator.available_size() < num_tokens:
            self.tree_cache.evict(num_tokens)
        state = self.token_to_kv_pool_allocator.backup_state() if backup_state else None
        out = self.token_to_kv_pool_allocator.alloc(num_tokens)
        if out is None:
            evictable = 0 if self.tree_cache is None else self.tree_cache.evictable_size()
            raise RuntimeError(f"Out of memory. Try to lower your batch size.\n"
                               f"Try to allocate {num_tokens} tokens.\n"
                               f"Avaliable tokens: {self.token_to_kv_pool_allocator.available_size() + evictable}\n")
        return (out, state) if backup_state else out

    def prepare_for_extend(self):
        self.forward_mode = ForwardMode.EXTEND
        reqs = self.reqs
        bs = len(reqs)
        req_pool_indices = self.alloc_req_slots(bs)
        input_ids = [r.fill_ids[r.prefix_len:] for r in reqs]
        extend_num_tokens = sum(len(ids) for ids in input_ids)
        seq_lens = [len(r.fill_ids) for r in reqs]
        prefix_lens = [r.prefix_len for r in reqs]
        extend_lens = [r.extend_input_len for r in reqs]

        dev = self.device
        req_pool_indices_tensor = torch.tensor(req_pool_indices, dtype=torch.int64).to(dev, non_blocking=True)
        input_ids_tensor = torch.tensor(sum(input_ids, []), dtype=torch.int64).to(dev, non_blocking=True)
        seq_lens_tensor = torch.tensor(seq_lens, dtype=torch.int64).to(dev, non_blocking=True)
        prefix_lens_tensor = torch.tensor(prefix_lens, dtype=torch.int64, device=dev)
        extend_lens_tensor = seq_lens_tensor - prefix_lens_tensor
        logprob_token_ids: List[int] = []
        for i, req in enumerate(reqs):
            req.req_pool_idx = req_pool_indices[i]
            if prefix_lens[i] > 0:
                self.req_to_token_pool.write((req.req_pool_idx, slice(0, prefix_lens[i])),
                                             req.prefix_indices.to(torch.int32))
            # schedule_batch.py:952-998: the request's logprob start relative to this round's extend part, and the
            # ids whose logprob each kept position is asked for (the NEXT input token; zero-padded past the prompt)
            if req.logprob_start_len >= prefix_lens[i]:
                req.extend_logprob_start_len = min(req.logprob_start_len - prefix_lens[i], req.extend_input_len,
                                                   len(req.origin_input_ids) + len(req.output_ids) - 1)
            else:
                req.extend_logprob_start_len = 0
            if self.return_logprob:
                first = max(prefix_lens[i], req.logprob_start_len)
                ids = req.origin_input_ids[first + 1:seq_lens[i] + 1]
                logprob_token_ids += ids + [0] * (req.extend_input_len - req.extend_logprob_start_len - len(ids))
        if self.return_logprob:
            self.top_logprobs_nums = [r.top_logprobs_num for r in reqs]
            self.token_ids_logprobs = [r.token_ids_logprob for r in reqs]
            self.extend_input_logprob_token_ids = torch.tensor(logprob_token_ids, dtype=torch.int64)
        else:
            self.extend_input_logprob_token_ids = None
        self.extend_logprob_start_lens = [r.extend_logprob_start_len for r in reqs]
        out_cache_loc = self.alloc_token_slots(extend_num_tokens)
        self.input_ids = input_ids_tensor
        self.req_pool_indices = req_pool_indices_tensor
        self.seq_lens = seq_lens_tensor
        self.out_cache_loc = out_cache_loc
        self.seq_lens_sum = sum(seq_lens)
        self.extend_num_tokens = extend_num_tokens
        self.prefix_lens = prefix_lens
        self.extend_lens = extend_lens
        _native.write_req_to_token(self.req_to_token_pool.req_to_token, req_pool_indices_tensor,
                                   prefix_lens_tensor, seq_lens_tensor, extend_lens_tensor,
                                   out_cache_loc)
        if self.is_encoder_decoder:
            self.prepare_encoder_info_extend(input_ids, seq_lens)

    def prepare_encoder_info_extend(self, input_ids: List[List[int]], seq_lens: List[int]):
        """schedule_batch.py:830-899: strip the encoder (image) tokens out of the decoder-side
        fields.  The request's req_to_token row keeps [encoder slots | text slots]; seq_lens,
        extend_lens/prefix_lens, input_ids and out_cache_loc describe the TEXT tokens only, and
        encoder_out_cache_loc receives the encoder slots (the encoder is all-or-nothing)."""
        self.encoder_lens_cpu, self.encoder_cached = [], []
        for req in self.reqs:
            if req.num_image_tokens is None:
                self.encoder_lens_cpu.append(0)
                self.encoder_cached.append(True)
            else:
                self.encoder_lens_cpu.append(req.num_image_tokens)
                self.encoder_cached.append(self.forward_mode.is_decode()
                                           or req.prefix_len >= req.num_image_tokens)
        self.encoder_lens = torch.tensor(self.encoder_lens_cpu, dtype=torch.int64).to(
            self.device, non_blocking=True)
        pt = 0
        decoder_out_cache_loc, encoder_out_cache_loc = [], []
        for i, req in enumerate(self.reqs):
            encoder_len = self.encoder_lens_cpu[i]
            seq_lens[i] -= encoder_len
            if req.prefix_len < encoder_len:
                assert req.prefix_len == 0, "the encoder part is cached as a whole"
                input_ids[i] = input_ids[i][encoder_len:]
                encoder_out_cache_loc.append(self.out_cache_loc[pt:pt + encoder_len])
                decoder_out_cache_loc.append(self.out_cache_loc[pt + encoder_len:pt + req.extend_input_len])
                self.extend_lens[i] -= encoder_len
                self.extend_num_tokens -= encoder_len
            else:
                decoder_out_cache_loc.append(self.out_cache_loc[pt:pt + req.extend_input_len])
                self.prefix_lens[i] -= encoder_len
            pt += req.extend_input_len
        self.input_ids = torch.tensor(sum(input_ids, []), dtype=torch.int64).to(self.device, non_blocking=True)
        self.seq_lens = torch.tensor(seq_lens, dtype=torch.int64).to(self.device, non_blocking=True)
        self.seq_lens_sum = sum(seq_lens)
        empty = torch.zeros(0, dtype=torch.int64).to(self.device)
        self.out_cache_loc = torch.cat(decoder_out_cache_loc) if decoder_out_cache_loc else empty
        self.encoder_out_cache_loc = torch.cat(encoder_out_cache_loc) if encoder_out_cache_loc else empty
        assert len(self.out_cache_loc) == self.extend_num_tokens

    def prepare_encoder_info_decode(self):
        """schedule_batch.py:1213-1215"""
        self.encoder_cached = [True] * len(self.reqs)

    def mix_with_running(self, running_batch: "ScheduleBatch", enable_overlap: Optional[bool] = None):
        """Chunked prefill + running decodes in one extend batch: the decode rows become extend rows
        of length 1 (schedule_batch.py:1073-1101).  Everything per-request that lives on the batch -
        sampling parameters, encoder lengths, pending output ids - is merged through ``merge_batch``
        exactly as the reference does; only input_ids / out_cache_loc are the concatenations built
        here (merge_batch resets out_cache_loc)."""
        # The reference's scheduler never mixes when either side returns logprobs (scheduler.py:944-949, "TODO: support
        # return_logprob + mixed chunked prefill"): its mix_with_running extends extend_logprob_start_lens by zeros but
        # not extend_input_logprob_token_ids, so the logits processor would index [kept positions, ids] with
        # mismatched lengths.  Here the id list is extended the way prepare_for_extend pads it ("the NEXT input token;
        # zero past the prompt"): a running row is one kept position whose next token is not known yet -> one zero.
        kept_own = sum(e - s for e, s in zip(self.extend_lens, self.extend_logprob_start_lens or [0] * len(self.extend_lens)))
        own_ids = self.extend_input_logprob_token_ids
        self.forward_mode = ForwardMode.MIXED
        running_bs = running_batch.batch_size()
        for req in running_batch.reqs:
            req.fill_ids = list(req.origin_input_ids) + req.output_ids
            req.extend_input_len = 1
        input_ids = torch.cat([self.input_ids, running_batch.input_ids])
        out_cache_loc = torch.cat([self.out_cache_loc, running_batch.out_cache_loc])
        self.merge_batch(running_batch)
        self.input_ids = input_ids
        self.out_cache_loc = out_cache_loc
        # with the overlap scheduler output_ids lags one step behind (schedule_batch.py:1088-1089)
        delta = 0 if (self.enable_overlap if enable_overlap is None else enable_overlap) else -1
        self.prefix_lens = self.prefix_lens + [len(r.origin_input_ids) + len(r.output_ids) + delta
                                               for r in running_batch.reqs]
        self.extend_lens = self.extend_lens + [1] * running_bs
        self.extend_num_tokens += running_bs
        self.extend_logprob_start_lens = list(self.extend_logprob_start_lens) + [0] * running_bs
        if self.return_logprob:            # (merge_batch has or-ed the two flags)
            if own_ids is None:
                own_ids = torch.zeros(kept_own, dtype=torch.int64)
            self.extend_input_logprob_token_ids = torch.cat([own_ids.cpu(), torch.zeros(running_bs, dtype=torch.int64)])

    def prepare_for_idle(self):
        """schedule_batch.py:1217-1228 (an idle batch has no rows: its sampling info stays "all greedy" = None)"""
        self.forward_mode = ForwardMode.IDLE
        self.input_ids = torch.empty(0, dtype=torch.int64, device=self.device)
        self.seq_lens = torch.empty(0, dtype=torch.int64, device=self.device)
        self.out_cache_loc = torch.empty(0, dtype=torch.int64, device=self.device)
        self.req_pool_indices = torch.empty(0, dtype=torch.int32, device=self.device)
        self.seq_lens_sum = 0
        self.extend_num_tokens = 0

    def prepare_for_decode(self, topping_manager=None):
        """schedule_batch.py:1230-1308.  ``topping_manager``: the reference's LoRA/delta manager; an ENABLED one
        re-orders the batch by adapter - toppings are out of scope, so that is refused rather than ignored."""
        if topping_manager is not None and getattr(topping_manager, "enabled", False):
            raise NotImplementedError("toppings (LoRA / delta adapters) are out of scope of this path")
        self.forward_mode = ForwardMode.DECODE
        bs = len(self.reqs)
        self.input_ids = self.output_ids
        self.output_ids = None
        if self.is_encoder_decoder:
            locs = self.encoder_lens + self.seq_lens      # text position behind the encoder slots
            self.prepare_encoder_info_decode()
        else:
            locs = self.seq_lens.clone()
        # overlap-safe (no in-place op): schedule_batch.py:1287-1292
        self.seq_lens = self.seq_lens + 1
        self.seq_lens_sum += bs
        self.out_cache_loc = self.alloc_token_slots(bs)
        self.req_to_token_pool.write((self.req_pool_indices, locs), self.out_cache_loc.to(torch.int32))

    def check_decode_mem(self, buf_multiplier: int = 1) -> bool:
        """schedule_batch.py:1109-1121: one new slot per running request, evicting if needed."""
        need = len(self.reqs) * buf_multiplier
        if self.token_to_kv_pool_allocator.available_size() >= need:
            return True
        if self.tree_cache is not None:
            self.tree_cache.evict(need)
        return self.token_to_kv_pool_allocator.available_size() >= need

    def retract_decode(self, server_args=20):
        """schedule_batch.py:1123-1211: give back the slots of the requests with the fewest output
        tokens (ties: longest prompt) until the rest can run ``retract_decode_steps`` more steps.
        ``server_args``: the reference's ServerArgs (``.retract_decode_steps`` is read) or the step count itself.
        Returns (retracted requests, the new token-ratio estimate) as the reference does."""
        if getattr(server_args, "speculative_algorithm", None):
            raise NotImplementedError("Speculative decoding is not supported yet.")
        retract_decode_steps = int(getattr(server_args, "retract_decode_steps", server_args))
        order = sorted(range(len(self.reqs)),
                       key=lambda i: (len(self.reqs[i].output_ids), -len(self.reqs[i].origin_input_ids)),
                       reverse=True)
        seq_lens_cpu = self.seq_lens.cpu().tolist()
        retracted = []
        first = True
        while first or self.token_to_kv_pool_allocator.available_size() < len(order) * retract_decode_steps:
            if len(order) == 1:
                assert self.token_to_kv_pool_allocator.available_size() > 0, "No space left for only one request"
                break
            first = False
            idx = order.pop()
            req = self.reqs[idx]
            retracted.append(req)
            row = self.req_to_token_pool.req_to_token[req.req_pool_idx]
            if self.tree_cache is None or isinstance(self.tree_cache, ChunkCache):
                self.token_to_kv_pool_allocator.free(row[:seq_lens_cpu[idx]].to(torch.int64))
                self.req_to_token_pool.free(req.req_pool_idx)
            else:
                # the cached prefix stays with the tree; only this request's own slots go back
                self.token_to_kv_pool_allocator.free(row[req.prefix_len:seq_lens_cpu[idx]].to(torch.int64))
                self.req_to_token_pool.free(req.req_pool_idx)
                self.tree_cache.dec_lock_ref(req.last_node)
                residual = len(order) * retract_decode_steps - self.token_to_kv_pool_allocator.available_size()
                self.tree_cache.evict(max(0, residual))
            req.reset_for_retract()
        self.filter_batch(keep_indices=order)
        decoded = sum(len(r.output_ids) for r in self.reqs)
        budget = sum(getattr(r.sampling_params, "max_new_tokens", 0) or 0 for r in self.reqs)
        ratio = min(1.0, (decoded + retract_decode_steps * len(self.reqs)) / budget) if budget else 1.0
        return retracted, ratio

    def filter_batch(self, chunked_req_to_exclude: Optional[Req] = None, keep_indices: Optional[List[int]] = None):
        """schedule_batch.py:1310-1359: keep the unfinished requests (and not the chunked request the scheduler
        is about to re-queue, scheduler.py:803), or exactly ``keep_indices``."""
        if keep_indices is None:
            keep_indices = [i for i, r in enumerate(self.reqs)
                            if not r.finished() and r is not chunked_req_to_exclude]
        if len(keep_indices) == 0:
            self.reqs = []
            return
        if len(keep_indices) == len(self.reqs):
            return
        keep = torch.tensor(keep_indices, dtype=torch.int64).to(self.device, non_blocking=True)
        if self.is_encoder_decoder:
            self.encoder_lens = self.encoder_lens[keep]
            self.encoder_lens_cpu = [self.encoder_lens_cpu[i] for i in keep_indices]
        self.reqs = [self.reqs[i] for i in keep_indices]
        self.req_pool_indices = self.req_pool_indices[keep]
        self.seq_lens = self.seq_lens[keep]
        self.out_cache_loc = None
        self.seq_lens_sum = self.seq_lens.sum().item()          # (schedule_batch.py:1334: the reference syncs here too)
        if self.output_ids is not None:
            self.output_ids = self.output_ids[keep]
        was_logprob = self.return_logprob
        self.return_logprob = any(r.return_logprob for r in self.reqs)
        if self.return_logprob and was_logprob:
            self.top_logprobs_nums = [self.top_logprobs_nums[i] for i in keep_indices]
            self.token_ids_logprobs = [self.token_ids_logprobs[i] for i in keep_indices]
        else:
            self.top_logprobs_nums = self.token_ids_logprobs = None
        self.has_stream = any(getattr(r, "stream", False) for r in self.reqs)
        self.has_grammar = any(getattr(r, "grammar", None) for r in self.reqs)
        if self.sampling_info is not None:
            self.sampling_info.filter_batch(keep_indices, keep)

    def merge_batch(self, other: "ScheduleBatch"):
        """schedule_batch.py:1361-1397.  The reference always carries a SamplingBatchInfo; here
        ``None`` stands for "every request greedy", so a side without one is given its greedy rows
        before the two are concatenated (row count must follow the merged request list)."""
        if self.sampling_info is not None or other.sampling_info is not None:
            from .sampler import SamplingBatchInfo
            have = self.sampling_info if self.sampling_info is not None else other.sampling_info
            vocab = have.vocab_size
            mine = (self.sampling_info if self.sampling_info is not None
                    else SamplingBatchInfo.from_schedule_batch(self, vocab))
            theirs = (other.sampling_info if other.sampling_info is not None
                      else SamplingBatchInfo.from_schedule_batch(other, vocab))
            mine.merge_batch(theirs)
            self.sampling_info = mine
        if self.is_encoder_decoder:
            self.encoder_lens = torch.cat([self.encoder_lens, other.encoder_lens])
            self.encoder_lens_cpu.extend(other.encoder_lens_cpu)
        self.req_pool_indices = torch.cat([self.req_pool_indices, other.req_pool_indices])
        self.seq_lens = torch.cat([self.seq_lens, other.seq_lens])
        self.out_cache_loc = None
        self.seq_lens_sum += other.seq_lens_sum
        if self.output_ids is not None:
            # as the reference: a side that carries pending output ids needs the other side's, or the rows
            # would no longer line up with reqs / seq_lens - fail loudly instead of keeping the old length
            if other.output_ids is None:
                raise RuntimeError("merge_batch: this batch carries output_ids but the merged one does not")
            self.output_ids = torch.cat([self.output_ids, other.output_ids])
        if self.return_logprob and other.return_logprob:
            self.top_logprobs_nums.extend(other.top_logprobs_nums)
            self.token_ids_logprobs.extend(other.token_ids_logprobs)
        elif self.return_logprob:
            self.top_logprobs_nums.extend([0] * len(other.reqs))
            self.token_ids_logprobs.extend([None] * len(other.reqs))
        elif other.return_logprob:
            self.top_logprobs_nums = [0] * len(self.reqs) + list(other.top_logprobs_nums)
            self.token_ids_logprobs = [None] * len(self.reqs) + list(other.token_ids_logprobs)
        self.reqs = self.reqs + other.reqs
        self.return_logprob |= other.return_logprob
        self.has_stream |= other.has_stream
        self.has_grammar |= other.has_grammar
        self.return_hidden_states |= other.return_hidden_states

    def get_model_worker_batch(self) -> ModelWorkerBatch:
        global bid
        bid += 1
        if self.forward_mode.is_decode_or_idle():
            extend_seq_lens = extend_prefix_lens = None
        else:
            extend_seq_lens, extend_prefix_lens = self.extend_lens, self.prefix_lens
        return ModelWorkerBatch(
            bid=bid, forward_mode=self.forward_mode, input_ids=self.input_ids,
            req_pool_indices=self.req_pool_indices, seq_lens=self.seq_lens,
            out_cache_loc=self.out_cache_loc, seq_lens_sum=self.seq_lens_sum,
            extend_num_tokens=self.extend_num_tokens, extend_seq_lens=extend_seq_lens,
            extend_prefix_lens=extend_prefix_lens,
            capture_hidden_mode=(CaptureHiddenMode.FULL if self.return_hidden_states else CaptureHiddenMode.NULL),
            global_num_tokens=self.global_num_tokens, global_num_tokens_for_logprob=self.global_num_tokens_for_logprob,
            can_run_dp_cuda_graph=self.can_run_dp_cuda_graph, input_embeds=self.input_embeds,
            toppings_paths=[getattr(r, "topping_path", None) for r in self.reqs],
            spec_algorithm=self.spec_algorithm, spec_info=self.spec_info, launch_done=self.launch_done,
            encoder_cached=self.encoder_cached, encoder_lens=self.encoder_lens,
            encoder_lens_cpu=self.encoder_lens_cpu, encoder_out_cache_loc=self.encoder_out_cache_loc,
            multimodal_inputs=[r.multimodal_inputs for r in self.reqs],
            sampling_info=self.sampling_info, return_logprob=self.return_logprob,
            top_logprobs_nums=self.top_logprobs_nums, token_ids_logprobs=self.token_ids_logprobs,
            extend_logprob_start_lens=(None if self.forward_mode.is_decode_or_idle() else self.extend_logprob_start_lens),
            extend_input_logprob_token_ids=(None if self.forward_mode.is_decode_or_idle()
                                            else self.extend_input_logprob_token_ids))

    def copy(self):
        """schedule_batch.py:1461-1472: only the fields process_batch_result reads"""
        return ScheduleBatch(reqs=self.reqs, model_config=self.model_config, forward_mode=self.forward_mode,
                             out_cache_loc=self.out_cache_loc, return_logprob=self.return_logprob,
                             decoding_reqs=self.decoding_reqs, spec_algorithm=self.spec_algorithm,
                             enable_custom_logit_processor=self.enable_custom_logit_processor,
                             device=self.device, is_encoder_decoder=self.is_encoder_decoder)

    def __str__(self):
        return f"ScheduleBatch(forward_mode={self.forward_mode.name if self.forward_mode else 'None'}, #req={len(self.reqs)})"
