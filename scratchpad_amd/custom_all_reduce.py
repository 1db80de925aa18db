"""Direct all-reduce for the TP group: the object GroupCoordinator.ca_comm expects
(distributed/parallel_state.py:266-267, 326-347: ``should_custom_ar(t)``, ``custom_all_reduce(t)``,
``capture()``).  The reference declares the slot and leaves it None; here it can be filled with the IPC
kernel of csrc/allreduce.hip (opt-in: ``SP_CUSTOM_ALLREDUCE=1``; RCCL stays the default and the
fallback for large or unaligned messages and during graph capture).

Set-up: every rank allocates one fine-grained region (flags | data | reduced), exports its IPC handle,
the handles travel over the group's gloo twin, and every rank maps every region.  Per call: one
kernel launch, epochs advance by 3 on all ranks in lock step (calls are collective)."""
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
        self._epoch = 1
        self._capturing = False
        torch.distributed.barrier(group=group.cpu_group)

    # ---- the ca_comm interface --------------------------------------------------------------
    def should_custom_ar(self, t: torch.Tensor) -> bool:
        nbytes = t.numel() * t.element_size()
        return (not self._capturing and t.is_cuda and t.is_contiguous() and nbytes % 16 == 0
                and 0 < nbytes <= self.data_bytes and t.data_ptr() % 16 == 0
                and t.dtype in (torch.float32, torch.float16, torch.bfloat16))

    def custom_all_reduce(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        if not self.should_custom_ar(t):
            return None
        out = torch.empty_like(t)
        _native._check(self.lib.sp_custom_all_reduce(out.data_ptr(), t.data_ptr(), t.numel(), _native._dt(t),
                                                     self._regions, self.rank, self.world, self._epoch,
                                                     self.data_bytes, _native._stream()), "sp_custom_all_reduce")
        self._epoch = (self._epoch + 3) & 0xFFFFFFFF
        return out

    @contextlib.contextmanager
    def capture(self):
        """Graph capture: epochs cannot be baked into a replayed launch, so captured steps use RCCL."""
        self._capturing = True
        try:
            yield
        finally:
            self._capturing = False

    def close(self):
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
