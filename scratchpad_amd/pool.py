"""KV memory: request->slot table, slot allocator, per-layer K/V pool.

Mirrors memory/pool.py: ReqToTokenPool 13-73, KVCache 150-186, TokenToKVPoolAllocator 189-255,
MHATokenToKVPool 258-424 (page_size = 1 only, as the reference enforces at
model_runner.py:431-432).  Slot 0 is the reserved dummy slot for padded rows.

MI355X layout: K and V each live in ONE allocation [layers, P+1, Hkv, D] (token-major, NHD);
``get_key_buffer(l)`` is a view of layer l, so addresses and strides are what the reference's
per-layer tensors would have, while a 288 GB HBM pool is a single contiguous arena.
"""
import abc
from typing import List, Optional, Tuple, Union

import torch

from . import _native


class ReqToTokenPool:
    """memory/pool.py:13-73: req_to_token[size, max_context_len] int32 + host free-list."""

    def __init__(self, size: int, max_context_len: int, device: str, use_records: bool = False):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)
        self.free_slots = list(range(size))
        self.write_records = []
        self.use_records = use_records
        self.write = self.write_with_records if use_records else self.write_without_records

    def available_size(self):
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        select_index = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return select_index

    def free(self, free_index: Union[int, List[int]]):
        if isinstance(free_index, int):
            self.free_slots.append(free_index)
        else:
            self.free_slots.extend(free_index)

    def clear(self):
        self.free_slots = list(range(self.size))
        self.write_records = []

    def write_without_records(self, indices, values):
        self.req_to_token[indices] = values

    def write_with_records(self, indices, values):
        self.req_to_token[indices] = values
        self.write_records.append((indices, values))

    def get_write_records(self):
        ret = self.write_records
        self.write_records = []
        return ret

    def apply_write_records(self, write_records: List[Tuple]):
        for indices, values in write_records:
            self.req_to_token[indices] = values


class KVCache(abc.ABC):
    """memory/pool.py:150-186."""

    @abc.abstractmethod
    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        raise NotImplementedError()

    @abc.abstractmethod
    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        raise NotImplementedError()

    @abc.abstractmethod
    def get_kv_buffer(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError()

    @abc.abstractmethod
    def set_kv_buffer(self, layer, loc: torch.Tensor, cache_k: torch.Tensor,
                      cache_v: torch.Tensor) -> None:
        raise NotImplementedError()

    @abc.abstractmethod
    def get_flat_data(self, indices):
        raise NotImplementedError()

    @abc.abstractmethod
    def transfer(self, indices, flat_data):
        raise NotImplementedError()

    @abc.abstractmethod
    def transfer_per_layer(self, indices, flat_data, layer_id):
        raise NotImplementedError()

    def register_layer_transfer_counter(self, layer_transfer_counter):
        self.layer_transfer_counter = layer_transfer_counter


class TokenToKVPoolAllocator:
    """memory/pool.py:189-255: free-list of slot ids 1..size (int64, on `device`)."""

    def __init__(self, size: int, dtype: torch.dtype, device: str, kvcache: KVCache):
        self.size = size
        self.dtype = dtype
        self.device = device
        self.page_size = 1
        self.free_slots = None
        self.is_not_in_free_group = True
        self.free_group = []
        self.clear()
        self._kvcache = kvcache

    def available_size(self):
        return len(self.free_slots)

    def get_kvcache(self):
        return self._kvcache

    def alloc(self, need_size: int):
        if need_size > len(self.free_slots):
            return None
        select_index = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return select_index

    def free(self, free_index: torch.Tensor):
        if free_index.numel() == 0:
            return
        if self.is_not_in_free_group:
            self.free_slots = torch.cat((self.free_slots, free_index))
        else:
            self.free_group.append(free_index)

    def free_group_begin(self):
        self.is_not_in_free_group = False
        self.free_group = []

    def free_group_end(self):
        self.is_not_in_free_group = True
        if self.free_group:
            self.free(torch.cat(self.free_group))

    def backup_state(self):
        return self.free_slots

    def restore_state(self, free_slots):
        self.free_slots = free_slots

    def clear(self):
        # slot 0 is reserved for dummy writes of padded tokens
        self.free_slots = torch.arange(1, self.size + 1, dtype=torch.int64, device=self.device)
        self.is_not_in_free_group = True
        self.free_group = []


class MHATokenToKVPool(KVCache):
    """memory/pool.py:258-424 (16/32-bit float pools; the fp8-as-uint8 branch is not built)."""

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

    def _create_buffers(self):
        shape = (self.layer_num, self.size + self.page_size, self.head_num, self.head_dim)
        self._k_arena = torch.zeros(shape, dtype=self.store_dtype, device=self.device)
        self._v_arena = torch.zeros(shape, dtype=self.store_dtype, device=self.device)
        self.k_buffer = [self._k_arena[i] for i in range(self.layer_num)]
        self.v_buffer = [self._v_arena[i] for i in range(self.layer_num)]

    def _clear_buffers(self):
        del self.k_buffer, self.v_buffer, self._k_arena, self._v_arena

    def get_kv_size_bytes(self):
        return (self._k_arena.numel() * self._k_arena.element_size(),
                self._v_arena.numel() * self._v_arena.element_size())

    def get_contiguous_buf_infos(self):
        bufs = self.k_buffer + self.v_buffer
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
