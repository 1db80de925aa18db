"""KV memory: request->slot table, slot allocator, per-layer K/V pool.

Same seam as memory/pool.py (ReqToTokenPool 13-73, KVCache 150-186, TokenToKVPoolAllocator 189-255,
MHATokenToKVPool 258-424; page_size = 1 only, as the reference enforces at model_runner.py:431-432;
slot 0 is the reserved dummy slot for padded rows) with the bookkeeping re-designed: both free lists
are rings (host ring for request rows, device ring for KV slots) instead of lists that are rebuilt
on every alloc/free.

MI355X layout: the whole pool is ONE allocation [layers, P+1, 2, Hkv, D] (token-major; a token's K and V
rows of a layer adjacent - MHATokenToKVPool.interleave_kv, the default since the end of round 4; the
other layout is one arena [layers, P+1, Hkv, D] per side).  ``get_key_buffer(l)`` / ``get_value_buffer(l)``
are views of layer l with the reference's shape [P+1, Hkv, D] (NHD) and a token stride the kernels take.
"""
import abc
import os
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import _native


class _HostFifo:
    """Fixed-capacity FIFO of small integers on the host (numpy ring: head + count).  Pops come from
    the head in insertion order, pushes go to the tail - the hand-out order a list sliced at the front
    and appended at the back would give, without re-building the list on every call."""

    def __init__(self, capacity: int):
        self.capacity = max(int(capacity), 1)
        self.ring = np.zeros(self.capacity, dtype=np.int64)
        self.head = 0
        self.count = 0

    def load(self, values) -> None:
        values = np.asarray(values, dtype=np.int64).reshape(-1)
        if values.size > self.capacity:                     # a caller-supplied list may exceed the table
            self.capacity = int(values.size)
            self.ring = np.zeros(self.capacity, dtype=np.int64)
        self.ring[:values.size] = values
        self.head, self.count = 0, int(values.size)

    def pop(self, n: int) -> np.ndarray:
        idx = (self.head + np.arange(n)) % self.capacity
        self.head = (self.head + n) % self.capacity
        self.count -= n
        return self.ring[idx]

    def push(self, values) -> None:
        values = np.asarray(values, dtype=np.int64).reshape(-1)
        if self.count + values.size > self.capacity:
            raise RuntimeError("free list overflow: more rows returned than the pool has")
        idx = (self.head + self.count + np.arange(values.size)) % self.capacity
        self.ring[idx] = values
        self.count += int(values.size)

    def snapshot(self) -> np.ndarray:
        return self.ring[(self.head + np.arange(self.count)) % self.capacity].copy()


class ReqToTokenPool:
    """The request -> KV-slot table of memory/pool.py:13-73: ``req_to_token[size, max_context_len]``
    int32 on the device (row = request, column = position), plus the free request rows.  Public
    surface kept for the seam: ``req_to_token``, ``size``, ``max_context_len``, ``alloc`` (rows as a
    Python list, ``None`` when short), ``free`` (one row or a list), ``clear``, ``available_size``,
    ``write`` and the write journal (``get_write_records`` / ``apply_write_records``, used by the
    reference's overlap worker to mirror table writes).  Rows are handed out first-freed-first."""

    def __init__(self, size: int, max_context_len: int, device: str, use_records: bool = False):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.use_records = use_records
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self._rows = _HostFifo(size)
        self._journal: List[Tuple] = []
        self.clear()

    # the free rows in hand-out order; assignable (tests and benches lay out a specific order)
    @property
    def free_slots(self) -> List[int]:
        return self._rows.snapshot().tolist()

    @free_slots.setter
    def free_slots(self, rows) -> None:
        self._rows.load(list(rows))

    def available_size(self) -> int:
        return self._rows.count

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > self._rows.count:
            return None
        return self._rows.pop(need_size).tolist()

    def free(self, free_index: Union[int, List[int]]) -> None:
        self._rows.push([free_index] if isinstance(free_index, int) else free_index)

    def clear(self) -> None:
        self._rows.load(np.arange(self.size))
        self._journal = []

    def write(self, indices, values) -> None:
        """one of the two below, by ``use_records`` (the reference re-points the attribute in __init__, pool.py:28-31)"""
        if self.use_records:
            self.write_with_records(indices, values)
        else:
            self.write_without_records(indices, values)

    def write_without_records(self, indices, values) -> None:
        self.req_to_token[indices] = values

    def write_with_records(self, indices, values) -> None:
        self.req_to_token[indices] = values
        self._journal.append((indices, values))

    def get_write_records(self) -> List[Tuple]:
        out, self._journal = self._journal, []
        return out

    def apply_write_records(self, write_records: List[Tuple]) -> None:
        for indices, values in write_records:
            self.req_to_token[indices] = values


class KVCache(abc.ABC):
    """The cache interface of memory/pool.py:150-186 (what an attention backend may call)."""

    @abc.abstractmethod
    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        ...

    @abc.abstractmethod
    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        ...

    @abc.abstractmethod
    def get_kv_buffer(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        ...

    @abc.abstractmethod
    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor,
                      cache_v: torch.Tensor) -> None:
        ...

    @abc.abstractmethod
    def get_flat_data(self, indices):
        ...

    @abc.abstractmethod
    def transfer(self, indices, flat_data):
        ...

    @abc.abstractmethod
    def transfer_per_layer(self, indices, flat_data, layer_id):
        ...

    def register_layer_transfer_counter(self, layer_transfer_counter):
        self.layer_transfer_counter = layer_transfer_counter


class TokenToKVPoolAllocator:
    """KV slot allocator with the interface of memory/pool.py:189-255 (slot ids 1..size as int64 on
    ``device``, slot 0 reserved for padded rows, page_size 1), built as a **device-resident ring**:

    * ``_ring[capacity]`` holds the free slot ids; the host keeps ``_head`` and ``_count`` - every
      size it needs (``alloc(n)``, ``free(t)`` with ``t.numel()``) is host-known, so no call ever
      reads the device;
    * ``alloc(n)`` copies ``n`` ids out of the ring at the head (one small copy; a wrap makes it two
      pieces) - the returned tensor is the caller's own, later frees cannot touch it;
    * ``free(t)`` appends ``t`` at the tail (one small copy).  The reference re-materialises the whole
      free list with ``torch.cat`` on every free - 4.5 MB per decode step for a 560 k-slot pool;
      here a step's alloc + free moves ``2 x batch`` ids;
    * hand-out order is first-freed-first, which is what the reference's slice-the-head /
      append-at-the-tail list gives, so recorded reference traces replay bit for bit
      (``tests/test_host_logic.py``, ``tests/test_radix_cache.py``);
    * ``free_group_begin/end`` batches the frees of one scheduler event into a single append;
      ``backup_state`` / ``restore_state`` snapshot and reload the list (speculative paths)."""

    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache: KVCache):
        self.size = size
        self.dtype = dtype
        self.device = device
        self.page_size = 1
        self._kvcache = kvcache
        self._capacity = max(int(size), 1)
        self._ring = torch.empty(self._capacity, dtype=torch.int64, device=device)
        self._head = 0
        self._count = 0
        self._group: Optional[List[torch.Tensor]] = None     # open free-group, else None
        self.clear()

    # ---- the list view (tests, benches and backup/restore use it; the hot calls never do)
    @property
    def free_slots(self) -> torch.Tensor:
        return self._take(self._head, self._count)

    @free_slots.setter
    def free_slots(self, slots: torch.Tensor) -> None:
        slots = slots.to(device=self.device, dtype=torch.int64).reshape(-1)
        if slots.numel() > self._capacity:
            self._capacity = int(slots.numel())
            self._ring = torch.empty(self._capacity, dtype=torch.int64, device=self.device)
        self._ring[:slots.numel()] = slots
        self._head, self._count = 0, int(slots.numel())

    def _take(self, start: int, n: int) -> torch.Tensor:
        """n ring entries from `start` as a fresh tensor (two pieces across the wrap)."""
        first = min(n, self._capacity - start)
        if first == n:
            return self._ring[start:start + n].clone()
        return torch.cat((self._ring[start:], self._ring[:n - first]))

    def _append(self, ids: torch.Tensor) -> None:
        n = ids.numel()
        if self._count + n > self._capacity:
            raise RuntimeError("KV free list overflow: more slots freed than the pool has")
        tail = (self._head + self._count) % self._capacity
        first = min(n, self._capacity - tail)
        self._ring[tail:tail + first] = ids[:first]
        if first < n:
            self._ring[:n - first] = ids[first:]
        self._count += n

    # ---- the allocator interface
    def available_size(self) -> int:
        return self._count

    def get_kvcache(self):
        return self._kvcache

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > self._count:
            return None
        out = self._take(self._head, need_size)
        self._head = (self._head + need_size) % self._capacity
        self._count -= need_size
        return out

    def free(self, free_index: torch.Tensor) -> None:
        if free_index.numel() == 0:
            return
        ids = free_index.reshape(-1).to(device=self.device, dtype=torch.int64)
        if self._group is not None:
            self._group.append(ids)
        else:
            self._append(ids)

    def free_group_begin(self) -> None:
        self._group = []

    def free_group_end(self) -> None:
        pending, self._group = self._group, None
        if pending:
            self._append(torch.cat(pending) if len(pending) > 1 else pending[0])

    @property
    def is_not_in_free_group(self) -> bool:
        return self._group is None

    def backup_state(self) -> torch.Tensor:
        return self.free_slots

    def restore_state(self, free_slots: torch.Tensor) -> None:
        self.free_slots = free_slots

    def clear(self) -> None:
        # slot 0 stays out of the list: padded rows write and read it (the dummy slot)
        self.free_slots = torch.arange(1, self.size + 1, dtype=torch.int64)
        self._group = None


class MHATokenToKVPool(KVCache):
    """memory/pool.py:258-424: fp32 / fp16 / bf16 pools, and the fp8 e5m2 pool stored as uint8 (274-280, 401-412;
    DESIGN 4.7); e4m3 raises."""

    def __init__(self, size: int, page_size: int, dtype: torch.dtype, head_num: int, head_dim: int,
                 layer_num: int, device: str, enable_memory_saver: bool = False):
        if page_size != 1:
            raise NotImplementedError("page_size > 1 (reference: model_runner.py:431-432)")
        if dtype not in (torch.float32, torch.float16, torch.bfloat16, torch.float8_e5m2):
            raise NotImplementedError(f"KV cache dtype {dtype} is not built")
        self.size = size
        self.page_size = page_size
        self.dtype = dtype
        # pool.py:274-280: fp8 pools are stored as uint8 (index_put has no fp8 kernel in torch)
        self.store_dtype = torch.uint8 if dtype == torch.float8_e5m2 else dtype
        self.device = device
        self.head_num = head_num
        self.head_dim = head_dim
        self.layer_num = layer_num
        self._create_buffers()
        self.layer_transfer_counter = None
        self.capture_mode = False

    # Default: ONE arena [layers, P+1, 2, Hkv, D] - a token's K row and V row of a layer adjacent (one 4 KiB run at
    # Llama-3-8B's 8 KV heads, 512 B per rank at 70B / TP 8); k_buffer[l] / v_buffer[l] are its two strided views of the
    # reference's shape [P+1, Hkv, D] (memory/pool.py:295-318 allocates one tensor per layer and side); every kernel takes
    # the token stride, so nothing else changes.  SP_KV_INTERLEAVE=0: K and V each in one arena [layers, P+1, Hkv, D].
    # Measured twice in round 4 (profiles/r04_decode_variants.txt sections 1 and 7e): while the decode kernel's gathers
    # were plain loads the interleaved arena was 3 % SLOWER in the model at the headline; with non-temporal gathers
    # (the stream no longer thrashes the caches, DRAM locality is what is left) it is faster everywhere it matters -
    # headline +1.1 - 1.7 % tokens/s (attention 0.385 -> 0.372 ms per layer), ctx 1024 +1.3 - 1.6 %, ctx 4096 +1.3 %,
    # bs 32 / 128 and the fp8 pool +1.1 - 1.2 %; bs 8, the 70B rank shape and the TTFT pass equal.
    interleave_kv = os.environ.get("SP_KV_INTERLEAVE", "1") != "0"

    def _create_buffers(self):
        rows = self.size + self.page_size
        if self.interleave_kv:
            self._kv_arena = torch.zeros((self.layer_num, rows, 2, self.head_num, self.head_dim),
                                         dtype=self.store_dtype, device=self.device)
            self._k_arena = self._kv_arena[:, :, 0]
            self._v_arena = self._kv_arena[:, :, 1]
        else:
            shape = (self.layer_num, rows, self.head_num, self.head_dim)
            self._kv_arena = None
            self._k_arena = torch.zeros(shape, dtype=self.store_dtype, device=self.device)
            self._v_arena = torch.zeros(shape, dtype=self.store_dtype, device=self.device)
        self.k_buffer = [self._k_arena[i] for i in range(self.layer_num)]
        self.v_buffer = [self._v_arena[i] for i in range(self.layer_num)]

    def _clear_buffers(self):
        del self.k_buffer, self.v_buffer, self._k_arena, self._v_arena, self._kv_arena

    def get_kv_size_bytes(self):
        return (self._k_arena.numel() * self._k_arena.element_size(),
                self._v_arena.numel() * self._v_arena.element_size())

    def get_contiguous_buf_infos(self):
        """pool.py:329-346: (data pointers, byte lengths, bytes per token) of buffers in which token ``i`` lives at
        ``ptr + i * item_len`` - the contract the transfer engines upstream copy by.  Separate arenas: the K buffers
        then the V buffers, as the reference lists them.  Interleaved arena (the default): ONE buffer per layer whose
        item is the token's K row followed by its V row (2 x Hkv x D elements) - never overlapping K / V regions with
        an item length that is not the stride."""
        if self.interleave_kv:
            bufs = [self._kv_arena[i] for i in range(self.layer_num)]
        else:
            bufs = self.k_buffer + self.v_buffer
        assert all(b.is_contiguous() for b in bufs)
        return ([b.data_ptr() for b in bufs], [b.nbytes for b in bufs], [b[0].nbytes for b in bufs])

    def get_flat_data(self, indices):
        return torch.stack([self._k_arena[:, indices], self._v_arena[:, indices]])

    def transfer(self, indices, flat_data):
        flat_data = flat_data.to(device=self.device, non_blocking=False)
        self._k_arena[:, indices] = flat_data[0]
        self._v_arena[:, indices] = flat_data[1]

    def transfer_per_layer(self, indices, flat_data, layer_id):
        flat_data = flat_data.to(device=self.device, non_blocking=False)
        self.k_buffer[layer_id][indices] = flat_data[0]
        self.v_buffer[layer_id][indices] = flat_data[1]

    def get_key_buffer(self, layer_id: int):
        if self.layer_transfer_counter is not None:
            self.layer_transfer_counter.wait_until(layer_id)
        buf = self.k_buffer[layer_id]
        return buf.view(self.dtype) if self.store_dtype != self.dtype else buf     # pool.py:366-371

    def get_value_buffer(self, layer_id: int):
        if self.layer_transfer_counter is not None:
            self.layer_transfer_counter.wait_until(layer_id)
        buf = self.v_buffer[layer_id]
        return buf.view(self.dtype) if self.store_dtype != self.dtype else buf

    def get_kv_buffer(self, layer_id: int):
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor,
                      k_scale: Optional[float] = None, v_scale: Optional[float] = None):
        """pool.py:392-424: k_buffer[layer][loc] = cache_k (one HIP scatter for K and V)."""
        layer_id = layer.layer_id
        if self.dtype == torch.float8_e5m2:
            # divide by the scales and round to e5m2 inside the scatter (pool.py:401-412)
            _native.kv_store_fp8(self.k_buffer[layer_id], self.v_buffer[layer_id], loc, cache_k, cache_v,
                                 1.0 if k_scale is None else float(k_scale),
                                 1.0 if v_scale is None else float(v_scale))
            return
        if cache_k.dtype != self.dtype:
            if k_scale is not None:
                cache_k.div_(k_scale)
            if v_scale is not None:
                cache_v.div_(v_scale)
            cache_k = cache_k.to(self.dtype)
            cache_v = cache_v.to(self.dtype)
        _native.kv_store(self.k_buffer[layer_id], self.v_buffer[layer_id], loc, cache_k, cache_v)
