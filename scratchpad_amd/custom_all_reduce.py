"""Direct all-reduce for the TP group: the object GroupCoordinator.ca_comm expects
(distributed/parallel_state.py:266-267, 326-347: ``should_custom_ar(t)``, ``custom_all_reduce(t)``,
``capture()``).  The reference declares the slot and leaves it None; here it can be filled with the IPC
kernel of csrc/allreduce.hip (opt-in: ``SP_CUSTOM_ALLREDUCE=1``; RCCL stays the default and the
fallback for large or unaligned messages).

Set-up: every rank allocates one fine-grained region (flags + call counters + status | data | reduced),
exports its IPC handle, the handles travel over the group's gloo twin, and every rank maps every region.
Per call: one kernel launch with call-independent arguments (the epoch counters are device state), so
the launch can be captured into a HIP graph and replayed - a captured decode step needs no library
collective.  A peer barrier that times out raises the region's status word: ``check()`` reads it (after
every call with ``SP_CUSTOM_ALLREDUCE_DEBUG=1``, always in ``close()``; ``bench.py --mode tp`` calls it
after its timed regions) and raises; the communicator then hands every later call to RCCL."""
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
        own = ctypes.c_void_p()
        _native._check(lib.sp_ar_alloc(ctypes.byref(own), self.region_bytes), "sp_ar_alloc")
        self._own = own.value
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
        self.failed = False          # a barrier timed out: every later call goes to RCCL
        self.calls = 0
        self.debug = os.environ.get("SP_CUSTOM_ALLREDUCE_DEBUG", "0") == "1"
        torch.distributed.barrier(group=group.cpu_group)

    # ---- the ca_comm interface --------------------------------------------------------------
    def should_custom_ar(self, t: torch.Tensor) -> bool:
        nbytes = t.numel() * t.element_size()
        return (not self.failed and t.is_cuda and t.is_contiguous() and nbytes % 16 == 0
                and 0 < nbytes <= self.data_bytes and t.data_ptr() % 16 == 0
                and t.dtype in (torch.float32, torch.float16, torch.bfloat16))

    def custom_all_reduce(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.should_custom_ar(t):
            return None
        out = torch.empty_like(t)
        _native._check(self.lib.sp_custom_all_reduce(out.data_ptr(), t.data_ptr(), t.numel(), _native._dt(t),
                                                     self._regions, self.rank, self.world, self.data_bytes,
                                                     _native._stream()), "sp_custom_all_reduce")
        self.calls += 1
        if self.debug and not self._capturing:
            self.check()
        return out

    def check(self) -> None:
        """Read the status word of this rank's region (synchronises).  Raises if a peer barrier timed
        out since the communicator was created; the communicator is then disabled."""
        if self._own is None:
            return
        status = ctypes.c_int(0)
        _native._check(self.lib.sp_ar_status(self._own, ctypes.byref(status)), "sp_ar_status")
        if status.value != 0:
            self.failed = True
            raise RuntimeError(
                f"direct all-reduce: a peer barrier timed out on rank {self.rank} (after {self.calls} calls); "
                "results since then are invalid - falling back to RCCL for the rest of the run")

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


def maybe_attach(group) -> Optional[CustomAllReduce]:
    """Fill group.ca_comm when SP_CUSTOM_ALLREDUCE=1 and the group spans 2..8 ranks."""
    if os.environ.get("SP_CUSTOM_ALLREDUCE", "0") != "1" or getattr(group, "world_size", 1) < 2:
        return None
    group.ca_comm = CustomAllReduce(group)
    return group.ca_comm
