"""Llama decoder hosted on the HIP hot path.

Re-hosts nn/models/llama/llama.py (LlamaMLP 31-66, LlamaAttention 69-154, LlamaDecoderLayer
157-224, LlamaModel 227-272, LlamaForCausalLM 275-311): same module tree and parameter names
(so reference checkpoints / state_dicts map 1:1), same op order and residual protocol.  The
callers of the hot path keep the reference's tensor-parallel shard math:
QKVParallelLinear (nn/layers/linear.py:696-760), MergedColumnParallelLinear (423-470),
RowParallelLinear + all-reduce (1033-1155), VocabParallelEmbedding + all-reduce
(nn/layers/vocab_parallel_embedding.py:452-472), LogitsProcessor pruning + all-gather
(nn/layers/logits_processor.py:179-203, 344-376).  GEMMs stay on torch (hipBLASLt/rocBLAS):
they are outside the path this package re-kernels.
"""
from typing import Any, Dict, Iterable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _native
from .attention import RadixAttention
from .distributed import (divide, get_tensor_model_parallel_rank,
                          get_tensor_model_parallel_world_size, get_tp_group, tensor_model_parallel_all_gather,
                          tensor_model_parallel_all_reduce)
from .forward_info import ForwardBatch
from .layers import RMSNorm, SiluAndMul, get_rope


class LogitsProcessorOutput:
    """nn/layers/logits_processor.py LogitsProcessorOutput.  ``next_token_logits`` ([bs, vocab] fp32,
    all-gathered over the TP group, padding columns cut: logits_processor.py:362-369) is what the
    reference's callers read; here the forward leaves only this rank's vocab shard in the model dtype
    (``shard_logits``).  A greedy batch never needs more (``greedy_token_ids``): under TP it exchanges one
    (value, index) pair per row instead of [bs, vocab / tp] logits, and at TP = 1 it skips the fp32 copy of
    the logits (131 MB at bs 256).  Whoever needs the full matrix - the sampler of a non-greedy batch,
    logprobs, tests - calls ``gather_full_logits()``: under TP that is a COLLECTIVE, so it is an explicit
    method that every rank of the group calls at the same point, never a side effect of an attribute read
    (a rank-0-only log line or a debugger would otherwise deadlock or mis-order the group's collectives)."""

    def __init__(self, next_token_logits: Optional[torch.Tensor] = None,
                 hidden_states: Optional[torch.Tensor] = None,
                 shard_logits: Optional[torch.Tensor] = None, vocab_size: Optional[int] = None,
                 shard_offset: int = 0):
        self._full = next_token_logits
        self.hidden_states = hidden_states
        self.shard_logits = shard_logits
        self.vocab_size = vocab_size
        self.shard_offset = shard_offset
        # prefill-only part (logits_processor.py:40-50): logprobs of the INPUT tokens, filled when the batch asks for
        # them (ForwardBatch.return_logprob with extend_logprob_start_lens short of the extend lengths)
        self.input_token_logprobs: Optional[torch.Tensor] = None        # [#input tokens]
        self.input_top_logprobs_val = None                              # per request: [#tokens][k]
        self.input_top_logprobs_idx = None
        self.input_token_ids_logprobs_val = None                        # per request: [#tokens][n]
        self.input_token_ids_logprobs_idx = None

    def gather_full_logits(self) -> torch.Tensor:
        """[bs, vocab] fp32 (logits_processor.py:362-369).  Collective under TP: call on every rank."""
        if self._full is None:
            if self.shard_logits is None:
                raise RuntimeError("LogitsProcessorOutput holds neither a logits shard nor full logits")
            logits = self.shard_logits
            if get_tensor_model_parallel_world_size() > 1:
                logits = tensor_model_parallel_all_gather(logits)
            self._full = logits[:, : self.vocab_size].float()
        return self._full

    @property
    def next_token_logits(self) -> Optional[torch.Tensor]:
        """The gathered logits if ``gather_full_logits()`` has run (or they were given); at TP = 1, where no
        collective is involved, they are built on first read.  Under TP an ungathered read raises instead of
        starting a collective from an attribute access."""
        if self._full is None and self.shard_logits is not None:
            if get_tensor_model_parallel_world_size() > 1:
                raise RuntimeError("next_token_logits under tensor parallelism: call gather_full_logits() on "
                                   "EVERY rank first (it is a collective)")
            return self.gather_full_logits()
        return self._full

    @next_token_logits.setter
    def next_token_logits(self, value: Optional[torch.Tensor]) -> None:
        self._full = value

    def rows(self, n: int) -> "LogitsProcessorOutput":
        """The first n rows (graph replay hands back the live rows of a padded bucket).  The object this is
        called on is the graph runner's PERSISTENT output buffer: a full-logits matrix cached on it would be
        the previous replay's - it must never have been materialised there."""
        if self._full is not None:
            raise RuntimeError("full logits were materialised on a persistent graph output buffer (they would "
                               "be stale on the next replay): read them from the object rows() returns")
        return LogitsProcessorOutput(
            None, None if self.hidden_states is None else self.hidden_states[:n],
            None if self.shard_logits is None else self.shard_logits[:n], self.vocab_size, self.shard_offset)

    def greedy_token_ids(self) -> torch.Tensor:
        """torch.argmax(next_token_logits, -1) (sampler.py:63-65) without materialising it."""
        if self._full is not None:
            return _native.argmax(self._full)
        if self.shard_logits is None:
            raise RuntimeError("LogitsProcessorOutput holds neither a logits shard nor full logits")
        shard = self.shard_logits
        cols = max(0, min(shard.shape[1], self.vocab_size - self.shard_offset))   # padding columns never win
        tp = get_tensor_model_parallel_world_size()
        if tp == 1:
            return _native.argmax(shard[:, :cols])
        pairs = _native.argmax_shard(shard, cols, self.shard_offset)
        gathered = tensor_model_parallel_all_gather(pairs, dim=0)                 # rank-major [tp * bs, 2]
        return _native.argmax_merge(gathered.view(tp, shard.shape[0], 2))


# --------------------------------------------------------------------------- sharded linears
class _ShardedLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype),
                                   requires_grad=False)


class QKVParallelLinear(_ShardedLinear):
    """linear.py:696-760: per-rank output = [q (Hq/tp*D) | k (Hkv_l*D) | v (Hkv_l*D)]; KV heads
    are replicated when tp >= total_kv_heads."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: Optional[int] = None, dtype=None):
        tp = get_tensor_model_parallel_world_size()
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.num_heads = divide(total_num_heads, tp)
        if tp >= self.total_num_kv_heads:
            self.num_kv_heads = 1
            self.num_kv_head_replicas = divide(tp, self.total_num_kv_heads)
        else:
            self.num_kv_heads = divide(self.total_num_kv_heads, tp)
            self.num_kv_head_replicas = 1
        super().__init__(hidden_size, (self.num_heads + 2 * self.num_kv_heads) * head_size, dtype)

    def shard_from_full(self, full: torch.Tensor) -> torch.Tensor:
        """Slice this rank's rows out of the tp=1 merged [q|k|v] weight."""
        rank = get_tensor_model_parallel_rank()
        D = self.head_size
        qn, kn = self.total_num_heads * D, self.total_num_kv_heads * D
        q, k, v = full[:qn], full[qn:qn + kn], full[qn + kn:qn + 2 * kn]
        qs = q[rank * self.num_heads * D:(rank + 1) * self.num_heads * D]
        kv_rank = rank // self.num_kv_head_replicas
        ks = k[kv_rank * self.num_kv_heads * D:(kv_rank + 1) * self.num_kv_heads * D]
        vs = v[kv_rank * self.num_kv_heads * D:(kv_rank + 1) * self.num_kv_heads * D]
        return torch.cat((qs, ks, vs), 0)

    def forward(self, x):
        return _native.linear(x, self.weight), None


class MergedColumnParallelLinear(_ShardedLinear):
    """linear.py:423-470: gate and up projections merged, each sharded along its output dim."""

    def __init__(self, input_size: int, output_sizes, dtype=None):
        tp = get_tensor_model_parallel_world_size()
        self.output_sizes = list(output_sizes)
        self.shard_sizes = [divide(s, tp) for s in self.output_sizes]
        super().__init__(input_size, sum(self.shard_sizes), dtype)

    def shard_from_full(self, full: torch.Tensor) -> torch.Tensor:
        rank = get_tensor_model_parallel_rank()
        parts, off = [], 0
        for size, shard in zip(self.output_sizes, self.shard_sizes):
            parts.append(full[off + rank * shard: off + (rank + 1) * shard])
            off += size
        return torch.cat(parts, 0)

    def forward(self, x):
        return _native.linear(x, self.weight), None


class RowParallelLinear(_ShardedLinear):
    """linear.py:1033-1155: input dim sharded; SUM all-reduce of the partial outputs (in place)."""

    def __init__(self, input_size: int, output_size: int, dtype=None, reduce_results: bool = True):
        tp = get_tensor_model_parallel_world_size()
        self.tp_size = tp
        self.input_size_per_partition = divide(input_size, tp)
        self.reduce_results = reduce_results
        super().__init__(self.input_size_per_partition, output_size, dtype)

    def shard_from_full(self, full: torch.Tensor) -> torch.Tensor:
        rank = get_tensor_model_parallel_rank()
        n = self.input_size_per_partition
        return full[:, rank * n:(rank + 1) * n]

    def forward(self, x, reduce_output: bool = True):
        """reduce_output=False leaves this rank's PARTIAL sums: the caller owes them the all-reduce and
        pays it inside the next norm (norm_after_row_parallel)."""
        out = _native.linear(x, self.weight)
        if self.reduce_results and reduce_output and self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out, None


def norm_after_row_parallel(norm: RMSNorm, x_partial: torch.Tensor, residual: torch.Tensor):
    """RMSNorm(all_reduce(x_partial), residual): the all-reduce RowParallelLinear would have run
    (linear.py:1148-1149) and the fused-add RMSNorm that follows it in the decoder layer (llama.py:216,
    222, 268), as one kernel where the TP group's custom all-reduce offers it, else as the same two steps
    the unfused path runs - the results are bit-identical either way."""
    if get_tp_group().fused_all_reduce_add_rmsnorm(x_partial, residual, norm.weight.data, norm.variance_epsilon):
        return x_partial, residual
    return norm(tensor_model_parallel_all_reduce(x_partial), residual)


class VocabParallelEmbedding(nn.Module):
    """vocab_parallel_embedding.py:172, 452-472: vocab rows sharded; masked lookup + all-reduce."""

    def __init__(self, num_embeddings: int, embedding_dim: int, dtype=None, padding_size: int = 64):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.tp_size = tp
        self.org_vocab_size = num_embeddings
        padded = (num_embeddings + padding_size - 1) // padding_size * padding_size
        while padded % tp:
            padded += padding_size
        self.num_embeddings_padded = padded
        self.num_embeddings_per_partition = padded // tp
        rank = get_tensor_model_parallel_rank()
        self.vocab_start_index = rank * self.num_embeddings_per_partition
        self.vocab_end_index = self.vocab_start_index + self.num_embeddings_per_partition
        self.weight = nn.Parameter(
            torch.zeros(self.num_embeddings_per_partition, embedding_dim, dtype=dtype),
            requires_grad=False)

    def shard_from_full(self, full: torch.Tensor) -> torch.Tensor:
        out = torch.zeros_like(self.weight, device=full.device)
        hi = min(self.vocab_end_index, full.shape[0])
        if hi > self.vocab_start_index:
            out[: hi - self.vocab_start_index] = full[self.vocab_start_index:hi]
        return out

    def forward(self, input_):
        if self.tp_size > 1:
            mask = (input_ < self.vocab_start_index) | (input_ >= self.vocab_end_index)
            masked = (input_ - self.vocab_start_index).masked_fill(mask, 0)
            out = F.embedding(masked, self.weight)
            out.masked_fill_(mask.unsqueeze(-1), 0)
            return tensor_model_parallel_all_reduce(out)
        return F.embedding(input_, self.weight)


class ParallelLMHead(VocabParallelEmbedding):
    pass


class LogitsProcessor(nn.Module):
    """logits_processor.py:140-376.  Decode and extend-without-input-logprobs keep only the last token of every
    sequence, matmul with the (vocab-sharded) head, and defer the all-gather / slice / fp32 step of _get_logits
    (362-369) to LogitsProcessorOutput (a greedy batch never needs it).  When the batch asks for the logprobs of its
    INPUT tokens (179-340: return_logprob with extend_logprob_start_lens short of the extend lengths) the hidden states
    of every position from the request's logprob start on are projected, the full logits are gathered (an explicit
    collective: the decision depends on host-side batch fields only, so every rank takes it) and the input-token /
    top-k / requested-id logprobs are cut out exactly as the reference does."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.do_tensor_parallel_all_gather = get_tensor_model_parallel_world_size() > 1

    @staticmethod
    def input_logprob_plan(extend_seq_lens_cpu, extend_logprob_start_lens_cpu):
        """logits_processor.py:206-246 (host side): row ranges of hidden_states to project, the row of each request's
        SAMPLED token among them, the rows that carry an input logprob, and the pruned lengths."""
        spans, sample_indices, input_logprob_indices, pruned_lens = [], [], [], []
        pt, sample_pt, input_pt = 0, -1, 0
        for start_len, extend_len in zip(extend_logprob_start_lens_cpu, extend_seq_lens_cpu):
            # chunked prefill may ask for no input logprob of this chunk at all: one token is still sampled
            first = start_len - 1 if extend_len == start_len else start_len
            assert extend_len > first
            spans.append((pt + first, pt + extend_len))
            pt += extend_len
            sample_pt += extend_len - first
            sample_indices.append(sample_pt)
            input_logprob_indices.extend(input_pt + i for i in range(extend_len - start_len))
            input_pt += extend_len - first
            pruned_lens.append(extend_len - start_len)
        return spans, sample_indices, input_logprob_indices, pruned_lens

    def _full_logits(self, states, lm_head):
        """_get_logits, logits_processor.py:344-376: matmul, all-gather over the TP group, cut the padding, fp32"""
        logits = _native.linear(states.to(lm_head.weight.dtype), lm_head.weight)
        if self.do_tensor_parallel_all_gather:
            logits = tensor_model_parallel_all_gather(logits)
        return logits[:, : self.config.vocab_size].float()

    def forward(self, input_ids, hidden_states, lm_head, forward_batch: ForwardBatch):
        fb = forward_batch
        want_input = (fb.forward_mode.is_extend() and fb.return_logprob and fb.extend_logprob_start_lens_cpu is not None
                      and any(e - s > 0 for e, s in zip(fb.extend_seq_lens_cpu, fb.extend_logprob_start_lens_cpu)))
        if not want_input:
            if fb.forward_mode.is_decode_or_idle():
                pruned = hidden_states
            else:
                last_index = torch.cumsum(fb.extend_seq_lens, dim=0) - 1
                pruned = hidden_states[last_index]
            logits = _native.linear(pruned.to(lm_head.weight.dtype), lm_head.weight)
            return LogitsProcessorOutput(shard_logits=logits, vocab_size=self.config.vocab_size,
                                         shard_offset=getattr(lm_head, "vocab_start_index", 0))
        spans, sample_idx, input_idx, pruned_lens = self.input_logprob_plan(fb.extend_seq_lens_cpu,
                                                                            fb.extend_logprob_start_lens_cpu)
        dev = hidden_states.device
        pruned = torch.cat([hidden_states[a:b] for a, b in spans])
        logits = self._full_logits(pruned, lm_head)
        out = LogitsProcessorOutput(next_token_logits=logits[torch.tensor(sample_idx, device=dev, dtype=torch.int64)],
                                    vocab_size=self.config.vocab_size)
        input_logits = logits[torch.tensor(input_idx, device=dev, dtype=torch.int64)]
        del logits
        # logits_processor.py:292-306, 432-460: optional temperature scaling / top-p normalisation of the input logprobs
        info = fb.sampling_info
        lens = torch.tensor(pruned_lens, device=dev)
        if fb.temp_scaled_logprobs and info is not None:
            input_logits = input_logits / torch.repeat_interleave(info.temperatures.view(-1), lens).view(-1, 1)
        if fb.top_p_normalized_logprobs and info is not None and bool((info.top_ps != 1.0).any()):
            from .sampler import top_p_normalize_probs
            probs = torch.softmax(input_logits, dim=-1)
            logprobs = torch.log(top_p_normalize_probs(probs, torch.repeat_interleave(info.top_ps, lens)))
        else:
            logprobs = torch.nn.functional.log_softmax(input_logits, dim=-1)
        if fb.top_logprobs_nums and any(k > 0 for k in fb.top_logprobs_nums):
            # get_top_logprobs, 378-405: one top-k over all rows, cut per request
            top = logprobs.topk(max(fb.top_logprobs_nums), dim=1)
            vals, idxs = top.values.tolist(), top.indices.tolist()
            out.input_top_logprobs_val, out.input_top_logprobs_idx, pt = [], [], 0
            for k, n in zip(fb.top_logprobs_nums, pruned_lens):
                out.input_top_logprobs_val.append([vals[pt + j][:k] for j in range(max(n, 0))])
                out.input_top_logprobs_idx.append([idxs[pt + j][:k] for j in range(max(n, 0))])
                pt += max(n, 0)
        if fb.token_ids_logprobs and any(t is not None for t in fb.token_ids_logprobs):
            # get_token_ids_logprobs, 407-430
            out.input_token_ids_logprobs_val, out.input_token_ids_logprobs_idx, pt = [], [], 0
            for ids, n in zip(fb.token_ids_logprobs, pruned_lens):
                if n <= 0 or ids is None:
                    out.input_token_ids_logprobs_val.append([])
                    out.input_token_ids_logprobs_idx.append([])
                    pt += max(n, 0) if ids is None else 0
                    continue
                out.input_token_ids_logprobs_val.append([logprobs[pt + j, ids].tolist() for j in range(n)])
                out.input_token_ids_logprobs_idx.append([ids for _ in range(n)])
                pt += n
        out.input_token_logprobs = logprobs[torch.arange(logprobs.shape[0], device=dev),
                                            fb.extend_input_logprob_token_ids_gpu]
        return out


# --------------------------------------------------------------------------- decoder
class LlamaMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str, dtype=None):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, dtype)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, dtype)
        if hidden_act != "silu":
            raise ValueError(f"Unsupported activation: {hidden_act}. Only silu is supported for now.")
        self.act_fn = SiluAndMul()

    def forward(self, x, reduce_output: bool = True):
        # small decode steps (<= _native.SILU_FUSED_MAX_ROWS tokens): projection and activation in one weight-streaming launch, the bits of
        # the skinny projection followed by act_fn
        act = _native.linear_silu_mul(x, self.gate_up_proj.weight)
        if act is None:
            gate_up, _ = self.gate_up_proj(x)
            act = self.act_fn(gate_up)
        x, _ = self.down_proj(act, reduce_output)
        return x


class LlamaAttention(nn.Module):
    def __init__(self, config, hidden_size: int, num_heads: int, num_kv_heads: int,
                 layer_id: int = 0, rope_theta: float = 10000,
                 rope_scaling: Optional[Dict[str, Any]] = None, rope_is_neox_style: bool = True,
                 max_position_embeddings: int = 8192, dtype=None):
        super().__init__()
        self.hidden_size = hidden_size
        tp_size = get_tensor_model_parallel_world_size()
        self.total_num_heads = num_heads
        assert self.total_num_heads % tp_size == 0
        self.num_heads = self.total_num_heads // tp_size
        self.total_num_kv_heads = num_kv_heads
        if self.total_num_kv_heads >= tp_size:
            assert self.total_num_kv_heads % tp_size == 0
        else:
            assert tp_size % self.total_num_kv_heads == 0
        self.num_kv_heads = max(1, self.total_num_kv_heads // tp_size)
        self.head_dim = getattr(config, "head_dim", None) or self.hidden_size // self.total_num_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.rope_theta = rope_theta
        self.max_position_embeddings = max_position_embeddings
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, self.total_num_heads,
                                          self.total_num_kv_heads, dtype)
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, hidden_size, dtype)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim,
                                   max_position=max_position_embeddings, base=rope_theta,
                                   rope_scaling=rope_scaling, is_neox_style=rope_is_neox_style,
                                   dtype=dtype)
        self.attn = RadixAttention(self.num_heads, self.head_dim, self.scaling,
                                   num_kv_heads=self.num_kv_heads, layer_id=layer_id)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor,
                forward_batch: ForwardBatch, reduce_output: bool = True) -> torch.Tensor:
        qkv, _ = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        backend = forward_batch.attn_backend
        if getattr(backend, "fused_rope_kv_store", False):
            # one launch: rotate q,k in place AND scatter rotated k + v into the KV pool
            # (= rotary_emb followed by set_kv_buffer; the backend then skips its own store)
            backend.rotary_and_store(self.rotary_emb, positions, q, k, v, self.attn, forward_batch)
            attn_output = self.attn(q, k, v, forward_batch, save_kv_cache=False)
        else:
            q, k = self.rotary_emb(positions, q, k)       # in place on the qkv views
            attn_output = self.attn(q, k, v, forward_batch)
        output, _ = self.o_proj(attn_output, reduce_output)
        return output


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config, layer_id: int = 0, dtype=None):
        super().__init__()
        self.hidden_size = config.hidden_size
        rope_theta = getattr(config, "rope_theta", 10000)
        rope_scaling = getattr(config, "rope_scaling", None)
        rope_is_neox_style = getattr(config, "rope_is_neox_style", True)
        max_position_embeddings = getattr(config, "max_position_embeddings", 8192)
        self.self_attn = LlamaAttention(
            config=config, hidden_size=self.hidden_size, num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads, layer_id=layer_id, rope_theta=rope_theta,
            rope_scaling=rope_scaling, rope_is_neox_style=rope_is_neox_style,
            max_position_embeddings=max_position_embeddings, dtype=dtype)
        self.mlp = LlamaMLP(self.hidden_size, config.intermediate_size, config.hidden_act, dtype)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, forward_batch: ForwardBatch,
                residual: Optional[torch.Tensor], defer_reduce: bool = False
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """llama.py:202-224.  defer_reduce (tensor parallelism only): the all-reduces of o_proj and
        down_proj are paid inside the norm that follows each of them (norm_after_row_parallel), so the
        hidden_states that come IN (from the previous layer's down_proj) and go OUT are partial sums."""
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        elif defer_reduce:
            hidden_states, residual = norm_after_row_parallel(self.input_layernorm, hidden_states, residual)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions=positions, hidden_states=hidden_states,
                                       forward_batch=forward_batch, reduce_output=not defer_reduce)
        if defer_reduce:
            hidden_states, residual = norm_after_row_parallel(self.post_attention_layernorm, hidden_states,
                                                              residual)
        else:
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states, reduce_output=not defer_reduce)
        return hidden_states, residual


class LlamaModel(nn.Module):
    def __init__(self, config, dtype=None):
        super().__init__()
        self.config = config
        self.vocab_size = config.vocab_size
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size, dtype)
        self.layers = nn.ModuleList(
            [LlamaDecoderLayer(config, i, dtype) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, forward_batch: ForwardBatch,
                input_embeds: torch.Tensor = None) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids) if input_embeds is None else input_embeds
        residual = None
        # under TP every row-parallel all-reduce is handed to the norm behind it (2 x layers per step)
        defer = get_tensor_model_parallel_world_size() > 1
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual, defer)
        if defer:
            hidden_states, _ = norm_after_row_parallel(self.norm, hidden_states, residual)
        else:
            hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class LlamaForCausalLM(nn.Module):
    def __init__(self, config, dtype=None):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, dtype)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head = self.model.embed_tokens
        else:
            self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, dtype)
        self.logits_processor = LogitsProcessor(config)

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch: ForwardBatch,
                input_embeds: torch.Tensor = None) -> LogitsProcessorOutput:
        hidden_states = self.model(input_ids, positions, forward_batch, input_embeds)
        return self.logits_processor(input_ids, hidden_states, self.lm_head, forward_batch)

    # ---- weights ------------------------------------------------------------------------
    def load_full_state_dict(self, full: Dict[str, torch.Tensor]) -> None:
        """Load a tp=1 state_dict in the reference's parameter naming (merged qkv_proj /
        gate_up_proj), slicing this rank's shard of every tensor-parallel parameter."""
        own = dict(self.named_parameters())
        mods = dict(self.named_modules())
        for name, param in own.items():
            src = full[name]
            mod = mods[name.rsplit(".", 1)[0]]
            if hasattr(mod, "shard_from_full"):
                src = mod.shard_from_full(src)
            if src.shape != param.shape:
                raise RuntimeError(f"{name}: shard shape {tuple(src.shape)} != {tuple(param.shape)}")
            param.data.copy_(src.to(param.dtype))

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> None:
        """HF checkpoint names (q_proj/k_proj/v_proj, gate_proj/up_proj) -> merged parameters
        (llama.py:364-405); builds the tp=1 merged dict, then shards."""
        cfg = self.config
        D = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        staged: Dict[str, Dict[str, torch.Tensor]] = {}
        full: Dict[str, torch.Tensor] = {}
        for name, w in weights:
            if "rotary_emb.inv_freq" in name or "rotary_emb.cos_cached" in name or \
                    "rotary_emb.sin_cached" in name:
                continue
            for tag, merged in (("q_proj", "qkv_proj"), ("k_proj", "qkv_proj"), ("v_proj", "qkv_proj"),
                                ("gate_proj", "gate_up_proj"), ("up_proj", "gate_up_proj")):
                if "." + tag + "." in name:
                    staged.setdefault(name.replace(tag, merged), {})[tag] = w
                    break
            else:
                full[name] = w
        for name, parts in staged.items():
            order = ("q_proj", "k_proj", "v_proj") if "qkv_proj" in name else ("gate_proj", "up_proj")
            full[name] = torch.cat([parts[t] for t in order], 0)
        del D
        self.load_full_state_dict(full)
