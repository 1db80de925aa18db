"""Tensor-parallel group + collectives: one process per GPU over RCCL / xGMI.

Mirrors the parts of distributed/parallel_state.py the hot path uses - GroupCoordinator
(122-221), all_reduce/_all_reduce (304-353) with the custom all-reduce slot ``ca_comm``
(266-267, 326-347), all_gather (355-381), init_distributed_environment /
initialize_model_parallel (881-997) - and distributed/communication_op.py:9-33.

``backend="nccl"`` IS RCCL on ROCm; ``backend="gloo"`` runs the same code on CPU (tests).
PP groups are not built: no reference model uses them.
"""
import contextlib
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class GraphCaptureContext:
    """parallel_state.py:38-40: the stream a graph is captured on."""
    stream: "torch.cuda.Stream"


@contextlib.contextmanager
def _capture_on(ca_comm, graph_capture_context: Optional[GraphCaptureContext]):
    """parallel_state.py:257-302 without pynccl: make the capture stream current (after everything already enqueued)
    and, when the custom all-reduce slot is filled, enter its ``capture()`` so that launches inside the graph
    register their buffers once (custom_all_reduce.py)."""
    if graph_capture_context is None:
        graph_capture_context = GraphCaptureContext(torch.cuda.Stream())
    stream = graph_capture_context.stream
    curr = torch.cuda.current_stream()
    if curr != stream:
        stream.wait_stream(curr)
    with torch.cuda.stream(stream), (ca_comm.capture() if ca_comm is not None else contextlib.nullcontext()):
        yield graph_capture_context


class GroupCoordinator:
    """A process group + its collectives (parallel_state.py:122-221, 304-381)."""

    def __init__(self, group_ranks, local_rank: int, backend: str):
        self.rank = dist.get_rank()
        self.local_rank = local_rank
        self.device_group = None
        self.cpu_group = None
        for ranks in group_ranks:
            device_group = dist.new_group(ranks, backend=backend)
            # a gloo twin for host-side object broadcast (parallel_state.py:176-187)
            cpu_group = dist.new_group(ranks, backend="gloo")
            if self.rank in ranks:
                self.ranks = ranks
                self.world_size = len(ranks)
                self.rank_in_group = ranks.index(self.rank)
                self.device_group = device_group
                self.cpu_group = cpu_group
        assert self.device_group is not None
        # custom all-reduce slot: an object with should_custom_ar(t) / custom_all_reduce(t)
        # (never constructed by the reference: parallel_state.py:174, 266); the direct xGMI
        # all-reduce plugs in here
        self.ca_comm = None

    @property
    def first_rank(self):
        return self.ranks[0]

    @property
    def is_first_rank(self):
        return self.rank == self.first_rank

    def graph_capture(self, graph_capture_context: Optional[GraphCaptureContext] = None):
        """Context manager around graph capture (parallel_state.py:257-302; entered by the graph runner,
        cuda_graph_runner.py:296): yields the GraphCaptureContext whose stream the capture runs on."""
        return _capture_on(self.ca_comm, graph_capture_context)

    def all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        """SUM all-reduce; applied in place or out of place - always use the return value."""
        if self.world_size == 1:
            return input_
        return self._all_reduce(input_)

    def _all_reduce(self, input_: torch.Tensor) -> torch.Tensor:
        ca_comm = self.ca_comm
        if ca_comm is not None and ca_comm.should_custom_ar(input_):
            out = ca_comm.custom_all_reduce(input_)
            if out is not None:
                return out
        dist.all_reduce(input_, group=self.device_group)
        return input_

    def fused_all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                                     eps: float) -> bool:
        """RowParallelLinear's all-reduce (linear.py:1148-1149) + the RMSNorm(x, residual) that follows it
        (llama.py:216, 222) as ONE kernel, when the custom all-reduce slot is filled and takes the shape:
        in place on x and residual, True.  False: nothing was done, the caller runs all_reduce() and the norm.
        The decision depends on shapes only, so every rank takes the same branch."""
        ca_comm = self.ca_comm
        if self.world_size == 1 or ca_comm is None or not hasattr(ca_comm, "fused_all_reduce_add_rmsnorm"):
            return False
        return ca_comm.fused_all_reduce_add_rmsnorm(x, residual, weight, eps)

    def poll(self) -> None:
        """Raise if the custom all-reduce has reported a failure (host-visible status word; no sync)."""
        if self.ca_comm is not None and hasattr(self.ca_comm, "poll"):
            self.ca_comm.poll()

    def all_gather(self, input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
        world_size = self.world_size
        if world_size == 1:
            return input_
        assert -input_.dim() <= dim < input_.dim(), f"Invalid dim ({dim}) for shape {input_.size()}"
        if dim < 0:
            dim += input_.dim()
        input_size = input_.size()
        # rank-major concatenation along dim 0 (a layout gloo and RCCL both accept), then viewed
        # as [world, *input] exactly like the reference's output tensor
        flat = torch.empty((world_size * input_size[0],) + tuple(input_size[1:]), dtype=input_.dtype,
                           device=input_.device)
        dist.all_gather_into_tensor(flat, input_.contiguous(), group=self.device_group)
        output = flat.view((world_size,) + tuple(input_size)).movedim(0, dim)
        return output.reshape(input_size[:dim] + (world_size * input_size[dim],) + input_size[dim + 1:])

    def broadcast_object(self, obj=None, src: int = 0):
        if self.world_size == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=self.ranks[src], group=self.cpu_group)
        return box[0]

    def barrier(self):
        dist.barrier(group=self.cpu_group)


_TP: Optional[GroupCoordinator] = None


def init_distributed_environment(world_size: int = 1, rank: int = 0,
                                 distributed_init_method: str = "env://", local_rank: int = 0,
                                 backend: str = "nccl"):
    """parallel_state.py:881-930."""
    if not dist.is_initialized():
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, init_method=distributed_init_method,
                                world_size=world_size, rank=rank, **kwargs)


def initialize_model_parallel(tensor_model_parallel_size: int = 1, backend: Optional[str] = None,
                              local_rank: int = 0) -> None:
    """parallel_state.py:933-997 (TP groups only): consecutive ranks form a TP group."""
    global _TP
    if tensor_model_parallel_size == 1:
        # TP=1 engines (also N independent replicas under one launcher) need no process group
        _TP = _SingleRankGroup()
        return
    assert dist.is_initialized(), "tensor parallelism needs init_distributed_environment() first"
    world_size = dist.get_world_size()
    backend = backend or dist.get_backend()
    assert world_size % tensor_model_parallel_size == 0
    n = world_size // tensor_model_parallel_size
    group_ranks = [list(range(i * tensor_model_parallel_size, (i + 1) * tensor_model_parallel_size))
                   for i in range(n)]
    _TP = GroupCoordinator(group_ranks, local_rank, backend)


class _SingleRankGroup:
    """TP=1 without a process group (single-GPU engine): every collective is the identity."""
    world_size = 1
    rank_in_group = 0
    ca_comm = None

    def graph_capture(self, graph_capture_context: Optional[GraphCaptureContext] = None):
        return _capture_on(None, graph_capture_context)

    def all_reduce(self, input_):
        return input_

    def all_gather(self, input_, dim: int = -1):
        return input_

    def fused_all_reduce_add_rmsnorm(self, x, residual, weight, eps) -> bool:
        return False

    def poll(self) -> None:
        pass

    def broadcast_object(self, obj=None, src: int = 0):
        return obj

    def barrier(self):
        pass


def model_parallel_is_initialized() -> bool:
    return _TP is not None


def destroy_model_parallel():
    global _TP
    _TP = None


def get_tp_group():
    assert _TP is not None, "tensor model parallel group is not initialized"
    return _TP


def graph_capture():
    """parallel_state.py:852-870 (TP group only: no PP groups are built)."""
    return get_tp_group().graph_capture()


def get_tensor_model_parallel_world_size() -> int:
    return get_tp_group().world_size


def get_tensor_model_parallel_rank() -> int:
    return get_tp_group().rank_in_group


def tensor_model_parallel_all_reduce(input_: torch.Tensor) -> torch.Tensor:
    """communication_op.py:9-11."""
    return get_tp_group().all_reduce(input_)


def tensor_model_parallel_all_gather(input_: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """communication_op.py:14-18."""
    return get_tp_group().all_gather(input_, dim)


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0, f"{numerator} is not divisible by {denominator}"
    return numerator // denominator
