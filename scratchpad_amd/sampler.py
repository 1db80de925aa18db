"""Next-token sampler on the HIP kernels of csrc/sampling.hip.

Mirrors nn/layers/sampler.py:24-163 (``Sampler.forward``), sampling/sampling_params.py:16-71
(``SamplingParams``: temperature < eps means greedy, top_k == -1 means the whole vocabulary) and the
tensor fields of sampling/sampling_batch_info.py:14-137 (``SamplingBatchInfo.from_schedule_batch``).
Penalizers, grammars / vocab masks and custom logit processors belong to the reference's request
front-end and are not built (out of scope).

Differences by design:
* the filter never sorts the vocabulary (see the kernel header) and carries mass as integers, so
  the drawn token is a pure function of (probabilities, top_k, top_p, min_p, uniform) - identical on
  every TP rank without the MIN all-reduce of sampler.py:146-157, which is still available as
  ``sync_token_ids_across_tp``;
* both reference back-ends ("flashinfer" rejection sampling and the "pytorch" sort) draw from the
  same filtered distribution; one kernel covers both.  With min-p the flashinfer branch
  renormalises between its top-k and top-p steps (sampler.py:91-96); the joint definition of
  sampler.py:195-221 is the one implemented, for every branch.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import nn

from . import _native

_SAMPLING_EPS = 1e-6
TOP_K_ALL = 1 << 30


class SamplingParams:
    """sampling/sampling_params.py:16-71 (the fields the sampler reads)."""

    def __init__(self, max_new_tokens: int = 128, temperature: float = 1.0, top_p: float = 1.0,
                 top_k: int = -1, min_p: float = 0.0, ignore_eos: bool = False):
        self.max_new_tokens = max_new_tokens
        self.temperature = temperature
        self.top_p = top_p
        self.top_k = top_k
        self.min_p = min_p
        self.ignore_eos = ignore_eos
        if 0 <= self.temperature < _SAMPLING_EPS:      # greedy
            self.temperature = 1.0
            self.top_k = 1
        if self.top_k == -1:
            self.top_k = TOP_K_ALL

    def verify(self):
        """sampling_params.py:73-103"""
        if self.temperature < 0.0:
            raise ValueError(f"temperature must be non-negative, got {self.temperature}.")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError(f"top_p must be in (0, 1], got {self.top_p}.")
        if not 0.0 <= self.min_p <= 1.0:
            raise ValueError(f"min_p must be in [0, 1], got {self.min_p}.")
        if self.top_k < 1:
            raise ValueError(f"top_k must be -1 (disable) or at least 1, got {self.top_k}.")


@dataclass
class SamplingBatchInfo:
    temperatures: torch.Tensor      # [bs, 1] fp32
    top_ps: torch.Tensor            # [bs] fp32
    top_ks: torch.Tensor            # [bs] int32
    min_ps: torch.Tensor            # [bs] fp32
    is_all_greedy: bool
    need_min_p_sampling: bool
    vocab_size: int
    device: str = "cuda"
    grammars: Optional[list] = None

    @classmethod
    def from_params(cls, params: List[SamplingParams], vocab_size: int, device: str):
        f32 = lambda xs: torch.tensor(xs, dtype=torch.float).to(device, non_blocking=True)
        return cls(
            temperatures=f32([p.temperature for p in params]).view(-1, 1),
            top_ps=f32([p.top_p for p in params]),
            top_ks=torch.tensor([p.top_k for p in params], dtype=torch.int32).to(device, non_blocking=True),
            min_ps=f32([p.min_p for p in params]),
            is_all_greedy=all(p.top_k <= 1 for p in params),
            need_min_p_sampling=any(p.min_p > 0 for p in params),
            vocab_size=vocab_size, device=device)

    @classmethod
    def from_schedule_batch(cls, batch, vocab_size: int):
        """sampling_batch_info.py:54-137; a request without sampling_params is greedy."""
        greedy = SamplingParams(temperature=0.0)
        return cls.from_params([getattr(r, "sampling_params", None) or greedy for r in batch.reqs],
                               vocab_size, batch.device)

    def __len__(self):
        return self.temperatures.shape[0]

    def filter_batch(self, keep_indices: List[int], keep_indices_device: torch.Tensor):
        """sampling_batch_info.py filter_batch: keep the rows of the surviving requests."""
        self.temperatures = self.temperatures[keep_indices_device]
        self.top_ps = self.top_ps[keep_indices_device]
        self.top_ks = self.top_ks[keep_indices_device]
        self.min_ps = self.min_ps[keep_indices_device]

    def merge_batch(self, other: "SamplingBatchInfo"):
        self.temperatures = torch.cat([self.temperatures, other.temperatures])
        self.top_ps = torch.cat([self.top_ps, other.top_ps])
        self.top_ks = torch.cat([self.top_ks, other.top_ks])
        self.min_ps = torch.cat([self.min_ps, other.min_ps])
        self.is_all_greedy = self.is_all_greedy and other.is_all_greedy
        self.need_min_p_sampling = self.need_min_p_sampling or other.need_min_p_sampling


def top_p_normalize_probs(probs: torch.Tensor, top_ps: torch.Tensor) -> torch.Tensor:
    """sampler.py:224-232"""
    return _native.top_k_top_p_min_p_renorm(probs, None, top_ps, None)


def top_k_renorm_prob(probs: torch.Tensor, top_ks: torch.Tensor) -> torch.Tensor:
    """nn/kernels/sampling.py:19-50"""
    return _native.top_k_top_p_min_p_renorm(probs, top_ks, None, None)


def top_p_renorm_prob(probs: torch.Tensor, top_ps: torch.Tensor) -> torch.Tensor:
    """nn/kernels/sampling.py:64-97"""
    return _native.top_k_top_p_min_p_renorm(probs, None, top_ps, None)


def get_top_logprobs(logprobs: torch.Tensor, top_logprobs_nums: List[int]):
    """sampler.py:235-250"""
    assert len(top_logprobs_nums) == logprobs.shape[0]
    ret = logprobs.topk(max(top_logprobs_nums), dim=1)
    values, indices = ret.values.tolist(), ret.indices.tolist()
    return ([values[i][:k] for i, k in enumerate(top_logprobs_nums)],
            [indices[i][:k] for i, k in enumerate(top_logprobs_nums)])


def get_token_ids_logprobs(logprobs: torch.Tensor, token_ids_logprobs: List[Optional[List[int]]]):
    """sampler.py:253-263"""
    vals, idxs = [], []
    for i, ids in enumerate(token_ids_logprobs):
        vals.append(logprobs[i, ids].tolist() if ids is not None else [])
        idxs.append(ids if ids is not None else [])
    return vals, idxs


class Sampler(nn.Module):
    def __init__(self, tp_group=None, use_nan_detection: bool = False):
        super().__init__()
        self.tp_group = tp_group
        self.use_nan_detection = use_nan_detection
        self.generator: Optional[torch.Generator] = None     # seedable source of the uniforms

    def forward(self, logits_output, sampling_info: SamplingBatchInfo, return_logprob: bool = False,
                top_logprobs_nums: Optional[List[int]] = None,
                token_ids_logprobs: Optional[List[Optional[List[int]]]] = None,
                uniform: Optional[torch.Tensor] = None) -> torch.Tensor:
        if sampling_info.is_all_greedy and not return_logprob and not self.use_nan_detection \
                and hasattr(logits_output, "greedy_token_ids"):
            # vocab-parallel greedy: no [bs, vocab] gather, no fp32 copy (LogitsProcessorOutput)
            ids = logits_output.greedy_token_ids()
            if sampling_info.grammars:
                self.sync_token_ids_across_tp(ids)
            return ids
        # every rank runs the same sampler on the same batch: the gather is an explicit collective here
        logits = (logits_output.gather_full_logits() if hasattr(logits_output, "gather_full_logits")
                  else logits_output.next_token_logits)
        if self.use_nan_detection and torch.any(torch.isnan(logits)):
            logits = torch.where(torch.isnan(logits), torch.full_like(logits, -1e5), logits)
        logprobs = None
        if sampling_info.is_all_greedy:
            ids = _native.argmax(logits)
            if return_logprob:
                logprobs = torch.nn.functional.log_softmax(logits, dim=-1)
        else:
            probs = _native.softmax_temperature_(logits, sampling_info.temperatures)
            if uniform is None:
                uniform = torch.rand(probs.shape[0], device=probs.device, generator=self.generator)
            ids = _native.top_k_top_p_min_p_sample(
                probs, sampling_info.top_ks, sampling_info.top_ps,
                sampling_info.min_ps if sampling_info.need_min_p_sampling else None, uniform)
            if return_logprob:
                logprobs = torch.log(top_p_normalize_probs(probs, sampling_info.top_ps)).clamp(
                    min=torch.finfo(probs.dtype).min)
        if return_logprob:
            if top_logprobs_nums and any(x > 0 for x in top_logprobs_nums):
                (logits_output.next_token_top_logprobs_val,
                 logits_output.next_token_top_logprobs_idx) = get_top_logprobs(logprobs, top_logprobs_nums)
            if token_ids_logprobs and any(x is not None for x in token_ids_logprobs):
                (logits_output.next_token_token_ids_logprobs_val,
                 logits_output.next_token_token_ids_logprobs_idx) = get_token_ids_logprobs(logprobs, token_ids_logprobs)
            logits_output.next_token_logprobs = logprobs[
                torch.arange(len(ids), device=ids.device), ids]
        if sampling_info.grammars:
            self.sync_token_ids_across_tp(ids)
        return ids

    def sync_token_ids_across_tp(self, ids: torch.Tensor):
        """sampler.py:146-157: MIN all-reduce of the drawn ids over the TP group."""
        if self.tp_group is not None and getattr(self.tp_group, "world_size", 1) > 1:
            torch.distributed.all_reduce(ids, op=torch.distributed.ReduceOp.MIN,
                                         group=self.tp_group.device_group)
