"""Forward-batch runner: pools, attention backend, eager + HIP-graph decode, greedy sampling.

Mirrors the hot-path parts of model_executor/model_runner.py (init_memory_pool 360-442,
init_attention_backend 453-470, forward/forward_decode/forward_extend 506-537, sample 539-563),
model_executor/cuda_graph_runner.py:144-521 (static-buffer capture / padded replay) and
managers/tp_worker.py:164-169 (forward_batch_generation).  Weight loading, offload, toppings,
speculative decoding and the NVML/zmq control plane are out of scope.
"""
import bisect
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _native
from . import distributed as dist_
from .attention import HipAttnBackend
from .forward_info import CaptureHiddenMode, ForwardBatch, ForwardMode, ModelWorkerBatch
from .llama import LlamaForCausalLM, LogitsProcessorOutput
from .sampler import Sampler
from .pool import MHATokenToKVPool, ReqToTokenPool, TokenToKVPoolAllocator


@dataclass
class ModelConfig:
    """The head/shape math of config/model_config.py:47-99, 155-163 (read-only use)."""
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    context_len: int = 4096
    head_dim: Optional[int] = None
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[dict] = None
    max_position_embeddings: int = 8192
    tie_word_embeddings: bool = False
    hidden_act: str = "silu"
    # Mllama text model: indices of the cross-attention decoder layers (None = plain Llama)
    cross_attention_layers: Optional[List[int]] = None
    pad_token_id: Optional[int] = None
    # Mllama: the vision tower's config (MllamaVisionConfig fields); None = text side only
    vision_config: Optional[object] = None

    @property
    def is_encoder_decoder(self) -> bool:
        return bool(self.cross_attention_layers)

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads

    def get_num_kv_heads(self, tp_size: int) -> int:
        # model_config.py:155-163: replicate KV heads when tp > total_kv
        return max(1, self.num_key_value_heads // tp_size)

    @classmethod
    def llama3_8b(cls, context_len: int = 8192):
        return cls(4096, 14336, 32, 32, 8, 128256, context_len=context_len,
                   max_position_embeddings=max(8192, context_len))

    @classmethod
    def llama32_1b(cls, context_len: int = 4096):
        return cls(2048, 8192, 16, 32, 8, 128256, context_len=context_len, tie_word_embeddings=True,
                   rope_scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
                                 "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
                   max_position_embeddings=max(8192, context_len))

    @classmethod
    def llama3_70b_tp8_rank(cls, context_len: int = 8192):
        """ONE rank's shard of Llama-3-70B at TP=8 expressed as a TP=1 model (8 q heads, 1 kv head,
        inter/8, vocab/8): the per-rank compute of config 4 without its all-reduces."""
        return cls(8192, 28672 // 8, 80, 8, 1, 128256 // 8, context_len=context_len, head_dim=128,
                   max_position_embeddings=max(8192, context_len))

    @classmethod
    def mllama_11b_text(cls, context_len: int = 8192):
        """Llama-3.2-11B-Vision text model: Llama-3-8B widths, 40 layers of which 8 cross-attend."""
        return cls(4096, 14336, 40, 32, 8, 128256, context_len=context_len,
                   cross_attention_layers=[3, 8, 13, 18, 23, 28, 33, 38],
                   max_position_embeddings=max(8192, context_len))

    @classmethod
    def llama3_70b(cls, context_len: int = 8192):
        return cls(8192, 28672, 80, 64, 8, 128256, context_len=context_len,
                   max_position_embeddings=max(8192, context_len))


@dataclass
class ServerArgs:
    """The path-shaping flags of server/args.py (defaults listed in SURVEY.md section 5)."""
    attention_backend: str = "hip"
    mem_fraction_static: float = 0.8
    max_running_requests: Optional[int] = None
    max_total_tokens: Optional[int] = None
    disable_cuda_graph: bool = False
    # the reference caps graphs at bs 160 (args.py:186-187); bs=256 is the headline config here
    cuda_graph_max_bs: int = 256
    cuda_graph_bs: Optional[List[int]] = None
    kv_cache_dtype: str = "auto"


def get_batch_sizes_to_capture(server_args: ServerArgs, max_reqs: int) -> List[int]:
    """cuda_graph_runner.py:92-128 with the bucket list extended to cuda_graph_max_bs."""
    bs = server_args.cuda_graph_bs
    if bs is None:
        bs = [1, 2, 4, 8] + list(range(16, server_args.cuda_graph_max_bs + 1, 8))
    return sorted({b for b in bs if b <= max_reqs and b <= server_args.cuda_graph_max_bs})


class ModelRunner:
    def __init__(self, model_config: ModelConfig, server_args: Optional[ServerArgs] = None,
                 tp_rank: int = 0, tp_size: int = 1, device: str = "cuda",
                 dtype: torch.dtype = torch.bfloat16, gpu_id: int = 0, seed: int = 0,
                 init_weights: bool = True):
        self.model_config = model_config
        self.server_args = server_args or ServerArgs()
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self.dtype = dtype
        # model_runner.py:360-372: "auto" = the model dtype; "fp8_e5m2" = 1-byte KV
        kvd = self.server_args.kv_cache_dtype
        if kvd == "auto":
            self.kv_cache_dtype = dtype
        elif kvd == "fp8_e5m2":
            self.kv_cache_dtype = torch.float8_e5m2
        else:
            raise ValueError(f"Unsupported kv_cache_dtype: {kvd}.")
        self.gpu_id = gpu_id
        if device == "cuda":
            if not torch.cuda.is_available():
                raise RuntimeError("ModelRunner needs a GPU: the hot path has no CPU fallback")
            torch.cuda.set_device(gpu_id)
            self.device = f"cuda:{gpu_id}"
        else:
            raise RuntimeError(f"unsupported device {device!r} (reference: model_runner.py:136-140)")
        if not dist_.model_parallel_is_initialized():
            dist_.initialize_model_parallel(tp_size, local_rank=gpu_id)
        assert dist_.get_tensor_model_parallel_world_size() == tp_size
        if tp_size > 1 and getattr(dist_.get_tp_group(), "ca_comm", None) is None:
            from .custom_all_reduce import maybe_attach
            maybe_attach(dist_.get_tp_group())      # opt-in (SP_CUSTOM_ALLREDUCE=1): direct IPC all-reduce
        self.sliding_window_size = None
        self.load_model(seed, init_weights)
        self.init_memory_pool()
        self.init_attention_backend()
        self.graph_runner: Optional[HipGraphRunner] = None

    # ------------------------------------------------------------------ model
    def load_model(self, seed: int, init_weights: bool):
        with torch.device(self.device):
            if self.model_config.is_encoder_decoder:
                from .mllama import MllamaForConditionalGeneration
                self.model = MllamaForConditionalGeneration(self.model_config, dtype=self.dtype).eval()
            else:
                self.model = LlamaForCausalLM(self.model_config, dtype=self.dtype).eval()
        for p in self.model.parameters():
            if p.dtype != self.dtype:
                p.data = p.data.to(self.dtype)
        if init_weights:
            g = torch.Generator(device=self.device).manual_seed(seed + 1000 * self.tp_rank)
            for name, p in self.model.named_parameters():
                if "norm" in name:
                    p.data.fill_(1.0)
                elif p.numel() == 1:          # Mllama tanh gates
                    p.data.fill_(0.5)
                else:
                    p.data.normal_(0.0, 0.02, generator=g)

    # ------------------------------------------------------------------ memory
    def profile_max_num_token(self) -> int:
        """model_runner.py:325-358: tokens that fit in mem_fraction_static of the free memory."""
        free, _total = torch.cuda.mem_get_info()
        cell = (self.model_config.get_num_kv_heads(self.tp_size) * self.model_config.head_dim *
                self.model_config.num_hidden_layers * 2 * torch.finfo(self.kv_cache_dtype).bits // 8)
        return int(free * self.server_args.mem_fraction_static // cell)

    def init_memory_pool(self):
        a = self.server_args
        self.max_total_num_tokens = a.max_total_tokens or self.profile_max_num_token()
        if self.max_total_num_tokens <= 0:
            raise RuntimeError("Not enough memory. Please try to increase --mem-fraction-static.")
        max_reqs = a.max_running_requests
        if max_reqs is None:
            max_reqs = min(max(int(self.max_total_num_tokens / self.model_config.context_len * 512),
                               2048), 4096)
        self.max_running_requests = max_reqs
        self.req_to_token_pool = ReqToTokenPool(max_reqs + 1, self.model_config.context_len + 4,
                                                self.device)
        self.token_to_kv_pool = MHATokenToKVPool(
            self.max_total_num_tokens, 1, self.kv_cache_dtype,
            self.model_config.get_num_kv_heads(self.tp_size), self.model_config.head_dim,
            self.model_config.num_hidden_layers, self.device)
        self.token_to_kv_pool_allocator = TokenToKVPoolAllocator(
            self.max_total_num_tokens, self.kv_cache_dtype, self.device, self.token_to_kv_pool)

    def init_attention_backend(self):
        if self.server_args.attention_backend != "hip":
            raise ValueError(f"Invalid attention backend: {self.server_args.attention_backend}")
        self.attn_backend = HipAttnBackend(self)
        self.sampler = Sampler(tp_group=dist_.get_tp_group())

    def init_cuda_graphs(self):
        """model_runner.py:490-504."""
        if self.server_args.disable_cuda_graph:
            return
        if _native._LIBROWS_MODE == "auto":
            # SP_LIBRARY_ROWS=auto: measure, on this box and this library build, which graph buckets are
            # cheaper when the library is handed a few more rows (_native.calibrate_library_rows)
            self.library_rows_table = _native.calibrate_library_rows(
                self._projection_weights(),
                get_batch_sizes_to_capture(self.server_args, self.req_to_token_pool.size))
        self.graph_runner = HipGraphRunner(self)

    def _projection_weights(self):
        """{(N, K): [weights]} of every projection that goes through _native.linear (one list per shape: the
        layers' weights are cycled by the calibration so that no call finds its weights in a cache)."""
        by_shape = {}
        for mod in self.model.modules():
            w = getattr(mod, "weight", None)
            if isinstance(w, torch.nn.Parameter) and w.dim() == 2 and hasattr(mod, "shard_from_full") \
                    and not isinstance(mod, torch.nn.Embedding) and type(mod).__name__ != "VocabParallelEmbedding":
                by_shape.setdefault((w.shape[0], w.shape[1]), []).append(w.data)
        return by_shape

    # ------------------------------------------------------------------ forward
    def _run_model(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        if self.model_config.is_encoder_decoder:
            return self.model.forward(forward_batch.input_ids, forward_batch.positions, forward_batch,
                                      cross_attention_states=forward_batch.encoder_states)
        return self.model.forward(forward_batch.input_ids, forward_batch.positions, forward_batch)

    def forward_decode(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        self.attn_backend.init_forward_metadata(forward_batch)
        return self._run_model(forward_batch)

    def forward_extend(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        self.attn_backend.init_forward_metadata(forward_batch)
        return self._run_model(forward_batch)

    def forward(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        # every forward / graph-replay boundary: has a direct all-reduce of an EARLIER step timed out?
        # (a read of a pinned host word the kernels raise, no synchronisation; raises on every rank)
        dist_.get_tp_group().poll()
        if (forward_batch.forward_mode.is_cuda_graph() and self.graph_runner is not None
                and self.graph_runner.can_run(forward_batch)):
            return self.graph_runner.replay(forward_batch)
        if forward_batch.forward_mode.is_decode():
            return self.forward_decode(forward_batch)
        if forward_batch.forward_mode.is_extend():
            return self.forward_extend(forward_batch)
        raise ValueError(f"Invalid forward mode: {forward_batch.forward_mode}")

    def sample(self, logits_output: LogitsProcessorOutput, forward_batch: ForwardBatch) -> torch.Tensor:
        """model_runner.py sample -> nn/layers/sampler.py:24-163.  A batch without sampling_info is
        greedy (sampler.py:63-67)."""
        info = forward_batch.sampling_info
        if info is None:
            if not forward_batch.return_logprob:
                return logits_output.greedy_token_ids()
            info = _ALL_GREEDY
        return self.sampler(logits_output, info, forward_batch.return_logprob, forward_batch.top_logprobs_nums,
                            forward_batch.token_ids_logprobs)


class _AllGreedy:
    """what the sampler reads of a SamplingBatchInfo when no request of the batch carries sampling parameters"""
    is_all_greedy = True
    grammars = None


_ALL_GREEDY = _AllGreedy()


class HipGraphRunner:
    """HIP-graph capture/replay of the decode step (cuda_graph_runner.py:144-521): static input
    buffers, one graph per batch-size bucket, padded replay (padded rows: seq_len = fill value,
    out_cache_loc = 0 = the dummy slot)."""

    def __init__(self, model_runner: ModelRunner):
        self.model_runner = model_runner
        self.graphs: Dict[int, torch.cuda.CUDAGraph] = {}
        self.output_buffers: Dict[int, LogitsProcessorOutput] = {}
        self.capture_bs = get_batch_sizes_to_capture(model_runner.server_args,
                                                     model_runner.req_to_token_pool.size)
        self.max_bs = max(self.capture_bs)
        backend = model_runner.attn_backend
        backend.init_cuda_graph_state(self.max_bs)
        self.seq_len_fill_value = backend.get_cuda_graph_seq_len_fill_value()
        dev = model_runner.device
        # graph inputs (cuda_graph_runner.py:191-199): int32 indices under graph replay
        self.input_ids = torch.zeros((self.max_bs,), dtype=torch.int64, device=dev)
        self.req_pool_indices = torch.zeros((self.max_bs,), dtype=torch.int32, device=dev)
        self.seq_lens = torch.full((self.max_bs,), self.seq_len_fill_value, dtype=torch.int32, device=dev)
        self.out_cache_loc = torch.zeros((self.max_bs,), dtype=torch.int64, device=dev)
        self.positions = torch.zeros((self.max_bs,), dtype=torch.int64, device=dev)
        # encoder-decoder models (cuda_graph_runner.py:201-208): encoder_lens is a graph input too;
        # fill value 0 = "no encoder tokens" for padded rows (their cross-attention rows stay zero)
        self.is_encoder_decoder = model_runner.model_config.is_encoder_decoder
        self.encoder_lens = (torch.zeros((self.max_bs,), dtype=torch.int32, device=dev)
                             if self.is_encoder_decoder else None)
        self.pool = None
        self.stream = None
        self.capture()

    def can_run(self, forward_batch: ForwardBatch) -> bool:
        return (forward_batch.batch_size <= self.max_bs
                and (forward_batch.encoder_lens is None) == (self.encoder_lens is None))

    def capture(self):
        model = self.model_runner.model
        if self.is_encoder_decoder:
            model.capture_mode = True      # cross-attention is always part of the captured step
        try:
            # cuda_graph_runner.py:295-329: capture on the group's capture stream; graph_capture() also enters the
            # custom all-reduce's capture() when that slot is filled (parallel_state.py:257-302)
            with dist_.graph_capture() as graph_capture_context:
                self.stream = graph_capture_context.stream
                for bs in reversed(self.capture_bs):
                    graph, out = self.capture_one_batch_size(bs)
                    self.graphs[bs] = graph
                    self.output_buffers[bs] = out
        finally:
            if self.is_encoder_decoder:
                model.capture_mode = False

    def capture_one_batch_size(self, bs: int):
        mr = self.model_runner
        fb = ForwardBatch(
            forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=self.input_ids[:bs],
            req_pool_indices=self.req_pool_indices[:bs], seq_lens=self.seq_lens[:bs],
            out_cache_loc=self.out_cache_loc[:bs], seq_lens_sum=int(self.seq_len_fill_value) * bs,
            positions=self.positions[:bs], req_to_token_pool=mr.req_to_token_pool,
            token_to_kv_pool=mr.token_to_kv_pool, attn_backend=mr.attn_backend,
            capture_hidden_mode=CaptureHiddenMode.NULL)
        enc = None
        if self.is_encoder_decoder:
            enc = self.encoder_lens[:bs]
            fb.encoder_lens = enc
            fb.encoder_lens_cpu = [0] * bs
            fb.encoder_cached = [True] * bs
        mr.attn_backend.init_forward_metadata_capture_cuda_graph(
            bs, bs, fb.req_pool_indices, fb.seq_lens, enc, ForwardMode.DECODE, None)

        def run_once():
            return mr.model.forward(fb.input_ids, fb.positions, fb)

        # warm up on a side stream before capturing (cuda_graph_runner.py:400-412)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                run_once()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: under TP the RCCL process group's watchdog thread polls the events
        # of earlier collectives while this thread captures; in the default (global) mode any HIP call
        # of another thread during a capture is an error ("operation not permitted when stream is
        # capturing") and takes the process group down - found by
        # tests/test_gpu_tensor_parallel.py::test_rccl_all_reduce_inside_a_hip_graph_single_rank
        with torch.cuda.graph(graph, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
            out = run_once()
        self.pool = graph.pool()
        return graph, out

    def replay(self, forward_batch: ForwardBatch) -> LogitsProcessorOutput:
        raw_bs = forward_batch.batch_size
        index = bisect.bisect_left(self.capture_bs, raw_bs)
        bs = self.capture_bs[index]
        if bs != raw_bs:
            self.seq_lens.fill_(self.seq_len_fill_value)
            self.out_cache_loc.zero_()
            if self.encoder_lens is not None:
                self.encoder_lens.zero_()
        if self.encoder_lens is not None:
            self.encoder_lens[:raw_bs].copy_(forward_batch.encoder_lens)
        self.input_ids[:raw_bs].copy_(forward_batch.input_ids)
        self.req_pool_indices[:raw_bs].copy_(forward_batch.req_pool_indices)
        self.seq_lens[:raw_bs].copy_(forward_batch.seq_lens)
        self.out_cache_loc[:raw_bs].copy_(forward_batch.out_cache_loc)
        self.positions[:raw_bs].copy_(forward_batch.positions)
        self.model_runner.attn_backend.init_forward_metadata_replay_cuda_graph(
            bs, self.req_pool_indices, self.seq_lens,
            forward_batch.seq_lens_sum + (bs - raw_bs) * self.seq_len_fill_value, self.encoder_lens,
            ForwardMode.DECODE, None, forward_batch.seq_lens_cpu)
        self.graphs[bs].replay()
        return self.output_buffers[bs].rows(raw_bs)


class TpModelWorker:
    """managers/tp_worker.py:25-184, the forward entry point only."""

    def __init__(self, model_runner: ModelRunner):
        self.model_runner = model_runner
        a, cfg = model_runner.server_args, model_runner.model_config
        # the limits the scheduler reads at start-up (tp_worker.py:73-105)
        self.max_total_num_tokens = model_runner.max_total_num_tokens
        self.max_prefill_tokens = getattr(a, "max_prefill_tokens", 16384)
        self.max_running_requests = model_runner.max_running_requests
        self.max_req_len = min(cfg.context_len - 1, self.max_total_num_tokens - 1)
        self.max_req_input_len = self.max_req_len - 5
        self.random_seed = getattr(a, "random_seed", 0)
        self.device = model_runner.device

    # ---- the accessors the reference's Scheduler calls at start-up (scheduler.py:203-217, 333-336; tp_worker.py:107-162)
    def get_worker_info(self):
        mr = self.model_runner
        return (self.max_total_num_tokens, self.max_prefill_tokens, self.max_running_requests, self.max_req_len,
                self.max_req_input_len, self.random_seed, self.device, None,      # (global_args: server-side, not kept here)
                mr.req_to_token_pool.size, mr.req_to_token_pool.max_context_len, mr.token_to_kv_pool.size)

    def get_pad_input_ids_func(self):
        return getattr(self.model_runner.model, "pad_input_ids", None)

    def get_tp_cpu_group(self):
        return getattr(dist_.get_tp_group(), "cpu_group", None)

    def get_memory_pool(self):
        return self.model_runner.req_to_token_pool, self.model_runner.token_to_kv_pool_allocator

    def forward_batch_generation(self, model_worker_batch: ModelWorkerBatch, skip_sample: bool = False):
        forward_batch = ForwardBatch.init_new(model_worker_batch, self.model_runner)
        logits_output = self.model_runner.forward(forward_batch)
        if model_worker_batch.launch_done is not None:
            model_worker_batch.launch_done.set()
        if skip_sample:
            return logits_output, None
        return logits_output, self.model_runner.sample(logits_output, forward_batch)
