"""Direct all-reduce for the TP group: the object GroupCoordinator.ca_comm expects
(distributed/parallel_state.py:266-267, 326-347: ``should_custom_ar(t)``, ``custom_all_reduce(t)``,
``capture()``).  The reference declares the slot and leaves it None; here it can be filled with the IPC
kernels of csrc/allreduce.hip (opt-in: ``SP_CUSTOM_ALLREDUCE=1``; RCCL stays the default and carries
large or unaligned messages).  On top of the reference's interface it offers the fusion of the
row-parallel all-reduce with the residual add + RMSNorm that follows it in every decoder layer
(``fused_all_reduce_add_rmsnorm``; linear.py:1148-1149 -> llama.py:216/222 -> layernorm.py:22-32).

Set-up: every rank allocates one fine-grained region (flags + call counters + status | data | reduced)
and one pinned host status word, exports the region's IPC handle, the handles travel over the group's
gloo twin, and every rank maps every region.  Per call: one kernel launch with call-independent arguments
(the epoch counters are device state), so the launch can be captured into a HIP graph and replayed - a
captured decode step needs no library collective.

Failure handling (a time-out is FATAL and COLLECTIVE): a peer barrier that waits longer than
``SP_CUSTOM_ALLREDUCE_TIMEOUT_S`` (default 30 s of wall clock) raises the status word of every rank's
region and the timing-out rank's host word; every later launch on any rank sees its region's word, raises
its own host word and stops waiting.  ``poll()`` reads the host word as plain memory - no synchronisation
- and raises; ModelRunner.forward() calls it at every forward / graph-replay boundary, ``check()`` (a
synchronising read of the region itself) runs in ``close()`` and after timed bench regions.  There is no
per-rank fallback to RCCL after a failure: a rank that changed transport alone would mismatch its peers'
collectives, so every later call raises as well."""
import contextlib
import ctypes
import os
from typing import Optional

import torch

from . import _native


class CustomAllReduce:
    MAX_BYTES = 8 << 20          # messages above this go to RCCL (prefill-sized)

    def __init__(self, group, max_bytes: int = MAX_BYTES):
        """group: a GroupCoordinator (uses .cpu_group, .ranks, .rank_in_group, .world_size)."""
        lib = _native.load()
        self.lib = lib
        self.world = group.world_size
        self.rank = group.rank_in_group
        if not 2 <= self.world <= 8:
            raise RuntimeError("CustomAllReduce: world size 2..8")
        self.data_bytes = int(max_bytes)
        self.flag_bytes = int(lib.sp_ar_flag_bytes())
        self.region_bytes = self.flag_bytes + 2 * self.data_bytes
        self.timeout_us = int(float(os.environ.get("SP_CUSTOM_ALLREDUCE_TIMEOUT_S", "30")) * 1e6)
        own = ctypes.c_void_p()
        _native._check(lib.sp_ar_alloc(ctypes.byref(own), self.region_bytes), "sp_ar_alloc")
        self._own = own.value
        host, dev = ctypes.c_void_p(), ctypes.c_void_p()
        _native._check(lib.sp_ar_host_status_alloc(ctypes.byref(host), ctypes.byref(dev)), "sp_ar_host_status_alloc")
        self._host_status, self._host_status_dev = host.value, dev.value
        self._host_word = ctypes.c_uint32.from_address(self._host_status)
        handle = ctypes.create_string_buffer(64)
        _native._check(lib.sp_ar_ipc_export(self._own, handle), "sp_ar_ipc_export")
        handles = [None] * self.world
        torch.distributed.all_gather_object(handles, (os.getpid(), handle.raw), group=group.cpu_group)
        self._mapped = []
        regions = (ctypes.c_void_p * self.world)()
        for r, (pid, raw) in enumerate(handles):
            if r == self.rank:
                regions[r] = self._own
                continue
            peer = ctypes.c_void_p()
            buf = ctypes.create_string_buffer(raw, 64)
            _native._check(lib.sp_ar_ipc_import(buf, ctypes.byref(peer)), "sp_ar_ipc_import")
            regions[r] = peer.value
            self._mapped.append(peer.value)
        self._regions = regions
        self._capturing = False
        self.failed = False          # a barrier timed out somewhere in the group: every later call raises
        self.calls = 0
        self.fused_calls = 0
        # the fused all-reduce + add + RMSNorm is a second opt-in (SP_CUSTOM_ALLREDUCE_FUSE_NORM=1): like the direct
        # all-reduce it has only run with all ranks on one device so far (no multi-GPU box in this pool); it becomes
        # the default once tests/test_gpu_tensor_parallel.py's one-device-per-rank cases have passed on a TP node
        self.fuse_norm = os.environ.get("SP_CUSTOM_ALLREDUCE_FUSE_NORM", "0") == "1"
        self._cast_weights = {}      # (data_ptr, version, dtype) -> the norm weight in the activation dtype
        self.debug = os.environ.get("SP_CUSTOM_ALLREDUCE_DEBUG", "0") == "1"
        torch.distributed.barrier(group=group.cpu_group)

    # ---- the ca_comm interface --------------------------------------------------------------
    def _eligible(self, t: torch.Tensor) -> bool:
        nbytes = t.numel() * t.element_size()
        return (t.is_cuda and t.is_contiguous() and nbytes % 16 == 0
                and 0 < nbytes <= self.data_bytes and t.data_ptr() % 16 == 0
                and t.dtype in (torch.float32, torch.float16, torch.bfloat16))

    def should_custom_ar(self, t: torch.Tensor) -> bool:
        # a pure function of the tensor's shape / dtype / alignment: the same answer on every rank (the
        # failure state is NOT part of it - a failed communicator raises instead of changing transport)
        return self._eligible(t)

    def _fail_if_failed(self):
        if self.failed:
            raise RuntimeError(f"direct all-reduce: the communicator failed earlier (rank {self.rank}); "
                               "the TP group must be torn down")

    def custom_all_reduce(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.should_custom_ar(t):
            return None
        self._fail_if_failed()
        out = torch.empty_like(t)
        _native._check(self.lib.sp_custom_all_reduce(out.data_ptr(), t.data_ptr(), t.numel(), _native._dt(t),
                                                     self._regions, self.rank, self.world, self.data_bytes,
                                                     self.timeout_us, self._host_status_dev,
                                                     _native._stream()), "sp_custom_all_reduce")
        self.calls += 1
        if self.debug and not self._capturing:
            self.check()
        return out

    def should_fuse_norm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor) -> bool:
        """Shape-only decision (identical on every rank) whether the fused kernel takes this call."""
        if not self.fuse_norm or x.dim() != 2 or residual.shape != x.shape or residual.dtype != x.dtype:
            return False
        vec = 16 // x.element_size()
        H = x.shape[1]
        # (everything a rank could decide differently from its peers is decided HERE, before any rank launches: the
        # weight's layout too - a weight in another dtype is cast once into a fresh, aligned buffer)
        return (self._eligible(x) and residual.is_cuda and residual.stride(1) == 1 and H % vec == 0 and H <= 8192
                and residual.stride(0) % vec == 0 and residual.data_ptr() % 16 == 0 and weight.numel() == H
                and weight.is_contiguous() and (weight.dtype != x.dtype or weight.data_ptr() % 16 == 0))

    def _norm_weight(self, weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        if weight.dtype == dtype:
            return weight
        key = (weight.data_ptr(), weight._version, dtype)
        w = self._cast_weights.get(key)
        if w is None:
            if len(self._cast_weights) > 1024:
                self._cast_weights.clear()
            w = self._cast_weights[key] = weight.to(dtype)
        return w

    def fused_all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                                     eps: float) -> bool:
        """x: this rank's partial sums [T, hidden].  In place: residual <- round(all_reduce(x) + residual),
        x <- RMSNorm(residual) * weight - bit for bit custom_all_reduce(x) followed by
        _native.fused_add_rmsnorm(x, residual, weight, eps).  False (nothing done) when the shape is not
        taken: the caller runs the two-step form."""
        if not self.should_fuse_norm(x, residual, weight):
            return False
        self._fail_if_failed()
        w = self._norm_weight(weight, x.dtype)
        _native._check(self.lib.sp_fused_allreduce_add_rmsnorm(
            x.data_ptr(), residual.data_ptr(), w.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
            residual.stride(0), float(eps), _native._dt(x), self._regions, self.rank, self.world,
            self.data_bytes, self.timeout_us, self._host_status_dev, _native._stream()),
            "sp_fused_allreduce_add_rmsnorm")
        self.fused_calls += 1
        if self.debug and not self._capturing:
            self.check()
        return True

    # ---- failure detection --------------------------------------------------------------------
    def poll(self) -> None:
        """Non-blocking: read the pinned host status word the kernels raise on a time-out (no stream
        synchronisation, no copy).  Cheap enough for every forward boundary."""
        if self._host_status is not None and self._host_word.value != 0:
            self._raise_failed("poll")

    def check(self) -> None:
        """Synchronising: read the status word of this rank's region itself (check points: close(), after a
        timed bench region, every call in debug mode)."""
        if self._own is None:
            return
        status = ctypes.c_int(0)
        _native._check(self.lib.sp_ar_status(self._own, ctypes.byref(status)), "sp_ar_status")
        if status.value != 0 or self._host_word.value != 0:
            self._raise_failed("check")

    def _raise_failed(self, where: str):
        self.failed = True
        raise RuntimeError(
            f"direct all-reduce ({where}): a peer barrier timed out in the TP group (seen on rank {self.rank} after "
            f"{self.calls} all-reduce / {self.fused_calls} fused calls, time-out {self.timeout_us / 1e6:.0f} s); "
            "results since then are invalid on EVERY rank - the failure is published to all regions, every rank "
            "raises at its next poll, and the group must be torn down")

    @contextlib.contextmanager
    def capture(self):
        """Graph capture (parallel_state.py:293-302 enters ca_comm.capture()): the launch is captured
        like any other kernel; only the per-call debug check is suspended (it synchronises)."""
        self._capturing = True
        try:
            yield
        finally:
            self._capturing = False

    def close(self):
        try:
            if self._own is not None and torch.cuda.is_available():
                torch.cuda.synchronize()
                self.check()
        finally:
            for p in self._mapped:
                self.lib.sp_ar_ipc_close(p)
            self._mapped = []
            if self._own:
                self.lib.sp_ar_free(self._own)
                self._own = None
            if self._host_status:
                self.lib.sp_ar_host_status_free(self._host_status)
                self._host_status = None


def maybe_attach(group) -> Optional[CustomAllReduce]:
    """Fill group.ca_comm when SP_CUSTOM_ALLREDUCE=1 and the group spans 2..8 ranks."""
    if os.environ.get("SP_CUSTOM_ALLREDUCE", "0") != "1" or getattr(group, "world_size", 1) < 2:
        return None
    group.ca_comm = CustomAllReduce(group)
    return group.ca_comm
