"""ctypes binding of ``libscratchpad_hip.so`` (C ABI: ``include/scratchpad_hip.h``).

There is no fallback: if the library is missing or a call fails, a ``RuntimeError`` is raised
(the reference's convention - exceptions in the forward thread are logged and the parent is
signalled, managers/tp_worker_client.py:110-116).  Tensors are passed as raw device pointers on
``torch.cuda.current_stream()``; nothing here synchronises.
"""
import ctypes
import os
import weakref
from typing import Optional

import torch

# SP_NATIVE_LIB names a diagnostic build under lib/ (A/B runs of a kernel variant through the whole model); unset = the product
_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                         os.environ.get("SP_NATIVE_LIB") or "libscratchpad_hip.so")
_lib = None

SP_F32, SP_F16, SP_BF16 = 0, 1, 2
_DTYPES = {torch.float32: SP_F32, torch.float16: SP_F16, torch.bfloat16: SP_BF16}
SP_FP8_E5M2 = 3
SP_OK, SP_ERR_INVALID_ARG, SP_ERR_UNSUPPORTED, SP_ERR_WORKSPACE, SP_ERR_LAUNCH = 0, -1, -2, -3, -4     # sp_status
_FP8_POOL_DTYPES = (torch.uint8, torch.float8_e5m2)


def _kv_layout(k_buffer: torch.Tensor, v_buffer: torch.Tensor, what: str, same_stride: bool = True) -> None:
    """The pool views every kernel assumes: [P+1, Hkv, D] with the heads of a token contiguous (stride(1) == D,
    stride(2) == 1) and - where the ABI carries ONE token stride for both sides - the same token stride on K and V.
    MHATokenToKVPool's views satisfy this under either arena layout; anything else would gather garbage silently."""
    for name, b in (("k_buffer", k_buffer), ("v_buffer", v_buffer)):
        if b.dim() != 3 or b.stride(2) != 1 or b.stride(1) != b.shape[2] or b.stride(0) < b.shape[1] * b.shape[2]:
            raise RuntimeError(f"{what}: {name} must be [tokens, Hkv, D] with contiguous heads per token, got shape "
                               f"{tuple(b.shape)} strides {tuple(b.stride())}")
    if k_buffer.shape[:2] != v_buffer.shape[:2]:
        raise RuntimeError(f"{what}: K and V pools disagree: {tuple(k_buffer.shape)} vs {tuple(v_buffer.shape)}")
    if same_stride and v_buffer.stride(0) != k_buffer.stride(0):
        raise RuntimeError(f"{what}: K and V pools need the same token stride ({k_buffer.stride(0)} vs {v_buffer.stride(0)})")


def _kv_dt(buf: torch.Tensor, q: torch.Tensor, what: str) -> int:
    """dtype code of a KV pool buffer next to 16/32-bit q: the same type, or an e5m2 byte pool."""
    if buf.dtype == q.dtype:
        return _DTYPES[q.dtype]
    if buf.dtype in _FP8_POOL_DTYPES and q.dtype in (torch.float16, torch.bfloat16):
        return SP_FP8_E5M2
    raise RuntimeError(f"{what}: KV pool dtype {buf.dtype} does not go with q dtype {q.dtype}")


_vp, _i64, _i32, _f32, _sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                              ctypes.c_size_t)

# name -> (restype, argtypes); must list every symbol include/scratchpad_hip.h declares
SIGNATURES = {
    "sp_abi_version": (_i32, []),
    "sp_status_string": (ctypes.c_char_p, [_i32]),
    "sp_debug_set": (_i32, [ctypes.c_char_p, _i32]),
    "sp_debug_get": (_i32, [ctypes.c_char_p]),
    "sp_rmsnorm": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _f32, _i32, _vp]),
    "sp_fused_add_rmsnorm": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _f32, _i32, _vp]),
    "sp_silu_and_mul": (_i32, [_vp, _vp, _i64, _i32, _i64, _i64, _i32, _vp]),
    "sp_rotary_embedding": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i64, _i64,
                                   _i32, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "sp_kv_store": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _i64,
                           _i64, _i32, _vp]),
    "sp_write_req_to_token": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "sp_compute_position": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "sp_clamp_position": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "sp_decode_plan_slots": (_i64, [_i32, _i64, _i64, _i32]),
    "sp_decode_attention_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sp_decode_ranges": (_i32, [_i32, _i32, _i32, _i32, _i32]),
    "sp_decode_plan_bytes": (_sz, [_i32, _i64, _i32]),
    "sp_decode_plan": (_i32, [_vp, _sz, _vp, _i32, _i32, _i64, _i32, _i64, _i32, _vp]),
    "sp_decode_attention": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32,
                                   _i32, _i32, _i32, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _i64, _i32, _i64,
                                   _i32, _vp, _sz, _vp, _sz, _i32, _i32, _vp]),
    "sp_extend_attention_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32, _i32]),
    "sp_extend_attention": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _vp,
                                   _i32, _i64, _i32, _i32, _i32, _i64, _i64, _i64, _f32, _f32, _f32, _f32,
                                   _i32, _i32, _i32, _i64, _vp, _sz, _vp, _sz, _i32, _i32, _vp]),
    "sp_extend_plan_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "sp_extend_plan": (_i32, [_vp, _sz, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "sp_kv_store_fp8": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _f32,
                               _i32, _vp]),
    "sp_argmax": (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "sp_argmax_shard": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
    "sp_argmax_merge": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "sp_softmax_temperature": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp]),
    "sp_top_k_top_p_min_p_sample": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "sp_top_k_top_p_min_p_renorm": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "sp_ar_flag_bytes": (_sz, []),
    "sp_ar_alloc": (_i32, [_vp, _sz]),
    "sp_ar_free": (_i32, [_vp]),
    "sp_ar_ipc_export": (_i32, [_vp, _vp]),
    "sp_ar_ipc_import": (_i32, [_vp, _vp]),
    "sp_ar_ipc_close": (_i32, [_vp]),
    "sp_ar_status": (_i32, [_vp, _vp]),
    "sp_ar_host_status_alloc": (_i32, [_vp, _vp]),
    "sp_ar_host_status_free": (_i32, [_vp]),
    "sp_custom_all_reduce": (_i32, [_vp, _vp, _i64, _i32, _vp, _i32, _i32, _sz, _i64, _vp, _vp]),
    "sp_fused_allreduce_add_rmsnorm": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _f32, _i32, _vp, _i32, _i32,
                                              _sz, _i64, _vp, _vp]),
    "sp_gemm_skinny": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _vp]),
}


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build it with `python -m scratchpad_amd.build` "
            "(or __graft_entry__.build()); there is no fallback path")
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.sp_abi_version() != 9:
        raise RuntimeError("libscratchpad_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(status: int, what: str):
    if status != 0:
        msg = load().sp_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} ({status})")


def debug_set(key: str, value: int) -> None:
    """Test / tuning switch of the library (sp_debug_set): "decode_kernel", "extend_defer_x10", "extend_dma" (0 = register-staged tiles),
    "extend_w64" (0 never / 1 where it pays (default) / 2 wherever it applies: the 4-wave x 64-row extend kernel),
    "extend_w64_persist" (the same three values for its persistent form on launches with a plan)."""
    _check(load().sp_debug_set(key.encode(), int(value)), f"sp_debug_set({key})")


EXTEND_KERNELS = {0: "none", 1: "extend_mfma_kernel (8 waves)", 2: "extend_w64_kernel (4 waves x 64 rows)",
                  3: "extend_w64p_kernel (persistent 4 waves x 64 rows)", 4: "row streams on the decode kernel"}


def debug_get(key: str) -> int:
    """sp_debug_get: "w64_descriptor_patched" (1 = the 4-wave x 64-row extend kernels may launch),
    "extend_last_kernel" (a key of EXTEND_KERNELS: what the last extend_attention call launched)."""
    v = int(load().sp_debug_get(key.encode()))
    if v < 0:
        raise RuntimeError(f"sp_debug_get({key}): unknown key")
    return v


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("scratchpad_amd HIP ops need device tensors (no CPU fallback)")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows(t: torch.Tensor) -> torch.Tensor:
    """View as [rows, inner] with a contiguous inner dimension (copy only if unavoidable)."""
    if t.dim() == 1:
        t = t.unsqueeze(0)
    if t.stride(-1) != 1:
        t = t.contiguous()
    if t.dim() > 2:
        t = t.reshape(-1, t.shape[-1])
    return t


# --------------------------------------------------------------------------- elementwise
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    _gpu(x, weight)
    x2 = _rows(x)
    out = empty_rows(x2.shape[0], x2.shape[1], x.dtype, x.device)      # spare rows behind it: see library_rows
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    _check(load().sp_rmsnorm(out.data_ptr(), x2.data_ptr(), w.data_ptr(), x2.shape[0], x2.shape[1],
                             x2.stride(0), out.stride(0), eps, _dt(x), _stream()), "sp_rmsnorm")
    return out.view(x.shape)


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """In place on both x and residual (flashinfer.norm.fused_add_rmsnorm contract)."""
    _gpu(x, residual, weight)
    if x.stride(-1) != 1 or residual.stride(-1) != 1 or x.dim() != 2 or residual.dim() != 2:
        raise RuntimeError("fused_add_rmsnorm needs 2-D row-contiguous x and residual (in-place op)")
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    _check(load().sp_fused_add_rmsnorm(x.data_ptr(), residual.data_ptr(), w.data_ptr(), x.shape[0],
                                       x.shape[1], x.stride(0), residual.stride(0), eps, _dt(x),
                                       _stream()), "sp_fused_add_rmsnorm")


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    _gpu(x)
    x2 = _rows(x)
    d = x2.shape[1] // 2
    out = empty_rows(x2.shape[0], d, x.dtype, x.device)      # spare rows behind it: see library_rows
    _check(load().sp_silu_and_mul(out.data_ptr(), x2.data_ptr(), x2.shape[0], d, x2.stride(0),
                                  out.stride(0), _dt(x), _stream()), "sp_silu_and_mul")
    return out.view(*x.shape[:-1], d)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                     head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool,
                     value: Optional[torch.Tensor] = None, k_buffer: Optional[torch.Tensor] = None,
                     v_buffer: Optional[torch.Tensor] = None,
                     out_cache_loc: Optional[torch.Tensor] = None) -> None:
    """In place on query/key ([T, H*head_size], row-contiguous views allowed).  With
    value/k_buffer/v_buffer/out_cache_loc also scatters rotated k and v into the KV pool."""
    _gpu(positions, query, key, cos_sin_cache, value, k_buffer, v_buffer, out_cache_loc)
    if query.dim() != 2 or key.dim() != 2 or query.stride(-1) != 1 or key.stride(-1) != 1:
        raise RuntimeError("rotary_embedding needs 2-D row-contiguous q and k (in-place op)")
    if cos_sin_cache.dtype != query.dtype or not cos_sin_cache.is_contiguous():
        raise RuntimeError("cos_sin_cache must be contiguous and in the activation dtype")
    if positions.dtype != torch.int64:
        positions = positions.to(torch.int64)
    positions = positions.contiguous()
    T = query.shape[0]
    Hq, Hkv = query.shape[1] // head_size, key.shape[1] // head_size
    fused = k_buffer is not None
    kv_stride = v_stride = 0
    if fused:
        if value.dim() != 2 or value.stride(-1) != 1:
            raise RuntimeError("fused KV store needs a 2-D row-contiguous value")
        if out_cache_loc.dtype != torch.int64:
            out_cache_loc = out_cache_loc.to(torch.int64)
        _kv_layout(k_buffer, v_buffer, "rotary_embedding (fused KV store)")
        kv_stride, v_stride = k_buffer.stride(0), value.stride(0)
    _check(load().sp_rotary_embedding(
        positions.data_ptr(), query.data_ptr(), key.data_ptr(), cos_sin_cache.data_ptr(), T, Hq, Hkv,
        head_size, cos_sin_cache.shape[1], query.stride(0), key.stride(0), int(is_neox),
        _ptr(value) if fused else None, v_stride, _ptr(k_buffer), _ptr(v_buffer),
        _ptr(out_cache_loc) if fused else None, kv_stride, _dt(query), _stream()),
        "sp_rotary_embedding")


def kv_store(k_buffer: torch.Tensor, v_buffer: torch.Tensor, loc: torch.Tensor,
             cache_k: torch.Tensor, cache_v: torch.Tensor) -> None:
    _gpu(k_buffer, v_buffer, loc, cache_k, cache_v)
    if cache_k.dtype != k_buffer.dtype or cache_v.dtype != v_buffer.dtype:
        raise RuntimeError("kv_store: cache dtype must equal the pool dtype")
    T = cache_k.shape[0]
    k2 = cache_k.reshape(T, -1)
    v2 = cache_v.reshape(T, -1)
    if k2.stride(-1) != 1:
        k2 = k2.contiguous()
    if v2.stride(-1) != 1:
        v2 = v2.contiguous()
    if loc.dtype != torch.int64:
        loc = loc.to(torch.int64)
    loc = loc.contiguous()
    _kv_layout(k_buffer, v_buffer, "kv_store", same_stride=False)
    Hkv, D, Dv = k_buffer.shape[1], k_buffer.shape[2], v_buffer.shape[2]
    _check(load().sp_kv_store(k_buffer.data_ptr(), v_buffer.data_ptr(), loc.data_ptr(), k2.data_ptr(),
                              v2.data_ptr(), T, Hkv, D, Dv, k2.stride(0), v2.stride(0),
                              k_buffer.stride(0), v_buffer.stride(0), _dt(k_buffer), _stream()),
           "sp_kv_store")


def kv_store_fp8(k_buffer: torch.Tensor, v_buffer: torch.Tensor, loc: torch.Tensor, cache_k: torch.Tensor,
                 cache_v: torch.Tensor, k_scale: float = 1.0, v_scale: float = 1.0) -> None:
    """pool[loc] = (cache / scale) rounded to fp8 e5m2; the pool is a uint8 (or float8_e5m2) tensor."""
    _gpu(k_buffer, v_buffer, loc, cache_k, cache_v)
    if k_buffer.dtype not in _FP8_POOL_DTYPES or v_buffer.dtype not in _FP8_POOL_DTYPES:
        raise RuntimeError("kv_store_fp8: the pool must be uint8 / float8_e5m2")
    if cache_k.dtype != cache_v.dtype or cache_k.dtype not in _DTYPES:
        raise RuntimeError("kv_store_fp8: k and v must share a 16/32-bit float dtype")
    T = cache_k.shape[0]
    k2, v2 = cache_k.reshape(T, -1), cache_v.reshape(T, -1)
    if k2.stride(-1) != 1:
        k2 = k2.contiguous()
    if v2.stride(-1) != 1:
        v2 = v2.contiguous()
    loc = loc.to(torch.int64).contiguous()
    _kv_layout(k_buffer, v_buffer, "kv_store_fp8", same_stride=False)
    Hkv, D = k_buffer.shape[1], k_buffer.shape[2]
    if v_buffer.shape[2] != D:
        raise RuntimeError("kv_store_fp8: v_head_dim must equal head_dim")
    _check(load().sp_kv_store_fp8(k_buffer.data_ptr(), v_buffer.data_ptr(), loc.data_ptr(), k2.data_ptr(),
                                  v2.data_ptr(), T, Hkv, D, k2.stride(0), v2.stride(0), k_buffer.stride(0),
                                  v_buffer.stride(0), k_scale, v_scale, _dt(cache_k), _stream()),
           "sp_kv_store_fp8")


# --------------------------------------------------------------------------- index kernels
def write_req_to_token(req_to_token: torch.Tensor, req_pool_indices: torch.Tensor,
                       pre_lens: torch.Tensor, seq_lens: torch.Tensor, extend_lens: torch.Tensor,
                       out_cache_loc: torch.Tensor) -> None:
    _gpu(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc)
    if req_to_token.dtype != torch.int32:
        raise RuntimeError("req_to_token must be int32")
    args = [t.to(torch.int64).contiguous() for t in
            (req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc)]
    _check(load().sp_write_req_to_token(req_to_token.data_ptr(), req_to_token.stride(0),
                                        *[t.data_ptr() for t in args], req_pool_indices.shape[0],
                                        _stream()), "sp_write_req_to_token")


def compute_position(extend_prefix_lens: torch.Tensor, extend_seq_lens: torch.Tensor,
                     extend_num_tokens: int):
    _gpu(extend_prefix_lens, extend_seq_lens)
    if extend_prefix_lens.dtype != torch.int32 or extend_seq_lens.dtype != torch.int32:
        raise RuntimeError("extend_prefix_lens / extend_seq_lens must be int32")
    bs = extend_seq_lens.shape[0]
    positions = torch.empty(extend_num_tokens, dtype=torch.int64, device=extend_seq_lens.device)
    start = torch.empty(bs, dtype=torch.int32, device=extend_seq_lens.device)
    _check(load().sp_compute_position(positions.data_ptr(), start.data_ptr(),
                                      extend_prefix_lens.contiguous().data_ptr(),
                                      extend_seq_lens.contiguous().data_ptr(), bs, _stream()),
           "sp_compute_position")
    return positions, start


def clamp_position(seq_lens: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _gpu(seq_lens)
    if seq_lens.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("seq_lens must be int32 or int64")
    seq_lens = seq_lens.contiguous()
    if out is None:
        out = torch.empty(seq_lens.shape[0], dtype=torch.int64, device=seq_lens.device)
    _check(load().sp_clamp_position(out.data_ptr(), seq_lens.data_ptr(),
                                    int(seq_lens.dtype == torch.int64), seq_lens.shape[0],
                                    _stream()), "sp_clamp_position")
    return out


# --------------------------------------------------------------------------- attention
def _idx_pair(req_pool_indices: torch.Tensor, seq_lens: torch.Tensor):
    if req_pool_indices.dtype != seq_lens.dtype:
        req_pool_indices = req_pool_indices.to(seq_lens.dtype)
    if seq_lens.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("seq_lens / req_pool_indices must be int32 or int64")
    return req_pool_indices.contiguous(), seq_lens.contiguous(), int(seq_lens.dtype == torch.int64)


def decode_plan_slots(bs: int, max_seq_len: int, chunk: int, kv_tokens: Optional[int] = None) -> int:
    """Partial slots (= work items) a decode step can need: min(bs * ceil(max_seq_len / chunk),
    kv_tokens // chunk + bs), kv_tokens = a bound on sum(seq_lens) (None: the first bound alone)."""
    return int(load().sp_decode_plan_slots(bs, -1 if kv_tokens is None else int(kv_tokens), max_seq_len, chunk))


def decode_workspace_bytes(bs: int, Hq: int, Dv: int, max_seq_len: int, chunk: int,
                           max_slots: Optional[int] = None, ranges: int = 0) -> int:
    """Split workspace for `max_slots` partial slots (default: the static bound bs * ceil(max_seq_len / chunk)) and,
    with `ranges`, for the bs + ranges slots of the range geometry - whichever is larger."""
    if max_slots is None:
        max_slots = decode_plan_slots(bs, max_seq_len, chunk)
    if ranges > 0:
        max_slots = max(max_slots, bs + ranges)
    return int(load().sp_decode_attention_workspace_bytes(max_slots, Hq, Dv))


RANGE_HEADER_WORDS = 4         # [pieces in use, piece length R, ranges, bs] in front of pos[bs + 1] and start[ranges]
RANGE_REQUEST_COST = 16        # positions a request takes on the line beyond its keys (attention_internal.h)


def decode_ranges(Hq: int, Hkv: int, D: int, dtype: torch.dtype, kv_dtype: Optional[torch.dtype] = None) -> int:
    """Pieces the range kernel wants for this shape (sp_decode_ranges): two workgroups per CU (three on a byte pool) over the
    kv heads, a wave per (piece, head); 0 where the range geometry does not apply (fp32, G > 16, D not 64 / 128)."""
    dt = _DTYPES.get(dtype)
    if dt is None:
        return 0
    kv = SP_FP8_E5M2 if kv_dtype in _FP8_POOL_DTYPES else dt
    return int(load().sp_decode_ranges(Hq, Hkv, D, dt, kv))


PLAN_HEADER_WORDS = 4          # [items listed, chunk, items the lengths need, keys the step gathers per kv head]


def _plan_slots(bs: int, max_seq_len: int, chunk: int, max_slots: Optional[int]) -> int:
    return decode_plan_slots(bs, max_seq_len, chunk) if max_slots is None else int(max_slots)


def decode_plan_bytes(bs: int, max_seq_len: int, chunk: int, max_slots: Optional[int] = None, ranges: int = 0) -> int:
    """max_slots = 0 with ranges > 0: a plan without the (request, split) items (range launches only)."""
    return int(load().sp_decode_plan_bytes(bs, _plan_slots(bs, max_seq_len, chunk, max_slots), ranges))


# What each plan buffer was last built with: data_ptr -> (batch size, max_slots, ranges).  The sections of a plan are
# located from these three numbers and a launch has to be given the same ones (include/scratchpad_hip.h, "Plan and launch
# must agree"): the range kernel would do nothing on a mismatch and the item kernels would read another section's words,
# so decode_attention() compares them here, on the host, and raises before anything is launched.  Buffers that were not
# built through decode_plan() (a copy made by hand) are not known and not checked; an entry whose tensor has died is stale
# (the allocator may have handed its address to something else) and is dropped, not trusted.
_BUILT_PLANS: dict = {}            # data_ptr -> (weakref to the plan tensor, (bs, max_slots, ranges))
_BUILT_PLANS_MAX = 4096


def _built_with(plan: torch.Tensor):
    entry = _BUILT_PLANS.get(plan.data_ptr())
    if entry is None:
        return None
    if entry[0]() is None:
        del _BUILT_PLANS[plan.data_ptr()]
        return None
    return entry[1]


def decode_plan(plan: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int, chunk: int,
                max_slots: Optional[int] = None, ranges: int = 0) -> None:
    """Fill `plan` (int32: [count, chunk, needed, keys | slot0[bs] | (request, split) x max_slots]) for this
    step's lengths.  `max_slots`: the item / partial-slot capacity the launches using this plan are given (default:
    the static bound bs * ceil(max_seq_len / chunk); 0 with ranges > 0: no items, the plan serves range launches only).
    plan[2] > max_slots afterwards means the capacity was too small (see decode_plan_overflow).
    `ranges` > 0 appends the range geometry ([pieces, R, ranges, bs | pos[bs + 1] | start[ranges]],
    include/scratchpad_hip.h) for launches given the same `ranges`."""
    _gpu(plan, seq_lens)
    if plan.dtype != torch.int32 or seq_lens.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("decode_plan: plan must be int32, seq_lens int32/int64")
    seq_lens = seq_lens.contiguous()
    bs = seq_lens.shape[0]
    max_slots = _plan_slots(bs, max_seq_len, chunk, max_slots)
    _check(load().sp_decode_plan(plan.data_ptr(), plan.numel() * 4, seq_lens.data_ptr(),
                                 int(seq_lens.dtype == torch.int64), bs, max_seq_len,
                                 chunk, max_slots, ranges, _stream()), "sp_decode_plan")
    key = plan.data_ptr()
    _BUILT_PLANS.pop(key, None)
    if len(_BUILT_PLANS) >= _BUILT_PLANS_MAX:
        del _BUILT_PLANS[next(iter(_BUILT_PLANS))]
    _BUILT_PLANS[key] = (weakref.ref(plan), (bs, max_slots, int(ranges)))


def decode_plan_overflow(header, max_slots: int) -> Optional[str]:
    """`header`: the first PLAN_HEADER_WORDS ints of a plan, on the host.  None if the plan lists every split the
    lengths need; otherwise the error text (splits were dropped: outputs of this step are wrong)."""
    listed, chunk, needed = int(header[0]), int(header[1]), int(header[2])
    if needed <= max_slots:
        return None
    return (f"decode split plan overflow: the step's seq_lens need {needed} (request, split) items at split size "
            f"{chunk} but the launch was sized for {max_slots} (listed {listed}): the bound on sum(seq_lens) handed to "
            "the attention backend (ForwardBatch.seq_lens_sum) is smaller than the lengths on the device")


def decode_attention(out: torch.Tensor, q: torch.Tensor, k_buffer: torch.Tensor,
                     v_buffer: torch.Tensor, req_to_token: torch.Tensor,
                     req_pool_indices: torch.Tensor, seq_lens: torch.Tensor, sm_scale: float,
                     logit_cap: float, max_seq_len: int, chunk: int, workspace: torch.Tensor,
                     kv_start: Optional[torch.Tensor] = None,
                     plan: Optional[torch.Tensor] = None, k_scale: Optional[float] = None,
                     v_scale: Optional[float] = None, max_slots: Optional[int] = None, ranges: int = 0) -> None:
    """q, out: [bs, Hq, D] (row stride free); buffers [P+1, Hkv, D].  k_scale / v_scale: the
    scales the store divided by (None = 1).  With a plan: `max_slots` = the capacity the plan was built
    for (0: it has no items), `chunk` = the smallest split size the plan may carry (the kernels read the actual one from
    it), `ranges` = the pieces its range section was built for (0: none; the workspace then holds max(max_slots, bs +
    ranges) slots).  Raises if they differ from what decode_plan() built this plan buffer with."""
    _gpu(out, q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens, workspace, kv_start, plan)
    bs, Hq, D = q.shape
    if q.stride(2) != 1 or q.stride(1) != D or out.stride(2) != 1 or out.stride(1) != D:
        raise RuntimeError("decode_attention: q/out must be [bs, Hq, D] with contiguous heads")
    kv_dt = _kv_dt(k_buffer, q, "decode_attention")
    _kv_layout(k_buffer, v_buffer, "decode_attention")
    if v_buffer.dtype != k_buffer.dtype:
        raise RuntimeError("decode_attention: K and V pools must share a dtype")
    req, seq, idx64 = _idx_pair(req_pool_indices, seq_lens)
    if kv_start is not None:
        kv_start = kv_start.to(seq.dtype).contiguous()
    max_slots = _plan_slots(bs, max_seq_len, chunk, max_slots)
    plan_bytes = 0
    if plan is not None:
        plan_bytes = plan.numel() * 4
        built = _built_with(plan)
        if built is not None and built != (bs, max_slots, int(ranges)):
            raise RuntimeError(
                f"decode_attention: the plan was built for (batch size, max_slots, ranges) = {built} and the launch is "
                f"given {(bs, max_slots, int(ranges))}: the plan's sections are located from these numbers, so a launch "
                "must pass the values its plan was built with (sp_decode_plan / sp_decode_attention)")
    _check(load().sp_decode_attention(
        out.data_ptr(), q.data_ptr(), k_buffer.data_ptr(), v_buffer.data_ptr(), req_to_token.data_ptr(),
        req_to_token.stride(0), req.data_ptr(), seq.data_ptr(), _ptr(kv_start), idx64, bs, Hq,
        k_buffer.shape[1], D, q.stride(0), out.stride(0), k_buffer.stride(0), sm_scale, logit_cap,
        1.0 if k_scale is None else float(k_scale), 1.0 if v_scale is None else float(v_scale),
        max_seq_len, chunk, max_slots, ranges if plan is not None else 0, workspace.data_ptr(),
        workspace.numel() * workspace.element_size(),
        _ptr(plan), plan_bytes, _dt(q), kv_dt, _stream()), "sp_decode_attention")


def extend_workspace_bytes(num_tokens: int, bs: int, Hq: int, D: int, dtype: torch.dtype) -> int:
    return int(load().sp_extend_attention_workspace_bytes(num_tokens, bs, Hq, D, _DTYPES[dtype]))


def extend_attention(out: torch.Tensor, q: torch.Tensor, k_buffer: torch.Tensor,
                     v_buffer: torch.Tensor, req_to_token: torch.Tensor,
                     req_pool_indices: torch.Tensor, seq_lens: torch.Tensor,
                     extend_seq_lens: torch.Tensor, extend_start_loc: torch.Tensor,
                     sm_scale: float, logit_cap: float, causal: bool, max_extend_len: int,
                     max_seq_len: int, workspace: torch.Tensor,
                     kv_start: Optional[torch.Tensor] = None, window_left: int = -1,
                     k_scale: Optional[float] = None, v_scale: Optional[float] = None,
                     plan: Optional[torch.Tensor] = None) -> None:
    """q, out: [T, Hq, D]; the new tokens' K/V must already be in the pool.  window_left >= 0:
    causal rows see only the window_left keys before their own position (and themselves)."""
    _gpu(out, q, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens, extend_seq_lens,
         extend_start_loc, workspace, kv_start, plan)
    if plan is not None and plan.dtype != torch.int32:
        raise RuntimeError("extend_attention: plan must be int32 (extend_plan)")
    T, Hq, D = q.shape
    if q.stride(2) != 1 or q.stride(1) != D or out.stride(2) != 1 or out.stride(1) != D:
        raise RuntimeError("extend_attention: q/out must be [T, Hq, D] with contiguous heads")
    kv_dt = _kv_dt(k_buffer, q, "extend_attention")
    _kv_layout(k_buffer, v_buffer, "extend_attention")
    if v_buffer.dtype != k_buffer.dtype:
        raise RuntimeError("extend_attention: K and V pools must share a dtype")
    if extend_seq_lens.dtype != torch.int32 or extend_start_loc.dtype != torch.int32:
        raise RuntimeError("extend_seq_lens / extend_start_loc must be int32")
    req, seq, idx64 = _idx_pair(req_pool_indices, seq_lens)
    if kv_start is not None:
        kv_start = kv_start.to(seq.dtype).contiguous()
    _check(load().sp_extend_attention(
        out.data_ptr(), q.data_ptr(), k_buffer.data_ptr(), v_buffer.data_ptr(), req_to_token.data_ptr(),
        req_to_token.stride(0), req.data_ptr(), seq.data_ptr(), _ptr(kv_start), idx64,
        extend_seq_lens.contiguous().data_ptr(), extend_start_loc.contiguous().data_ptr(),
        seq.shape[0], T, Hq, k_buffer.shape[1], D, q.stride(0), out.stride(0), k_buffer.stride(0),
        sm_scale, logit_cap, 1.0 if k_scale is None else float(k_scale),
        1.0 if v_scale is None else float(v_scale), int(causal), int(window_left), max_extend_len, max_seq_len, workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _ptr(plan), 0 if plan is None else plan.numel() * 4,
        _dt(q), kv_dt, _stream()), "sp_extend_attention")


def extend_plan(extend_seq_lens: torch.Tensor, seq_lens: torch.Tensor, num_tokens: int, Hq: int, Hkv: int,
                causal: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The (request, row block) work items of an extend step, heaviest first (int32; reuse `out` when it
    is large enough).  Built once per forward, shared by all layers with these head counts."""
    _gpu(extend_seq_lens, seq_lens, out)
    if extend_seq_lens.dtype != torch.int32 or seq_lens.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("extend_plan: extend_seq_lens int32, seq_lens int32/int64")
    bs = extend_seq_lens.shape[0]
    need = int(load().sp_extend_plan_bytes(num_tokens, bs, Hq, Hkv)) // 4
    if out is None or out.numel() < need or out.dtype != torch.int32:
        out = torch.empty(need, dtype=torch.int32, device=extend_seq_lens.device)
    seq = seq_lens.contiguous()
    _check(load().sp_extend_plan(out.data_ptr(), out.numel() * 4, extend_seq_lens.contiguous().data_ptr(),
                                 seq.data_ptr(), int(seq.dtype == torch.int64), bs, num_tokens, Hq, Hkv,
                                 int(causal), _stream()), "sp_extend_plan")
    return out


# --------------------------------------------------------------------------- sampler
def _prob_rows(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"{what}: expected [batch, vocab] with a contiguous vocab dimension")
    return t


def _opt(t: Optional[torch.Tensor], dtype: torch.dtype, bs: int, what: str):
    if t is None:
        return None
    t = t.reshape(-1)
    if t.shape[0] != bs:
        raise RuntimeError(f"{what}: expected {bs} entries, got {t.shape[0]}")
    return t.to(dtype).contiguous()


def argmax(logits: torch.Tensor) -> torch.Tensor:
    """First maximal index per row -> int64 [bs]."""
    _gpu(logits)
    logits = _prob_rows(logits, "argmax")
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    _check(load().sp_argmax(logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1],
                            out.data_ptr(), _dt(logits), _stream()), "sp_argmax")
    return out


def argmax_shard(logits: torch.Tensor, cols: int, index_offset: int) -> torch.Tensor:
    """This rank's vocab shard [bs, >= cols] -> int32 [bs, 2] = (fp32 bits of the row maximum over the
    first `cols` columns, global index of its first occurrence)."""
    _gpu(logits)
    logits = _prob_rows(logits, "argmax_shard")
    if not 0 <= cols <= logits.shape[1]:
        raise RuntimeError("argmax_shard: cols out of range")
    out = torch.empty((logits.shape[0], 2), dtype=torch.int32, device=logits.device)
    _check(load().sp_argmax_shard(logits.data_ptr(), logits.stride(0), logits.shape[0], int(cols),
                                  int(index_offset), out.data_ptr(), _dt(logits), _stream()), "sp_argmax_shard")
    return out


def argmax_merge(pairs: torch.Tensor) -> torch.Tensor:
    """pairs: int32 [shards, bs, 2] (rank-major all-gather of argmax_shard outputs) -> int64 [bs]."""
    _gpu(pairs)
    if pairs.dtype != torch.int32 or pairs.dim() != 3 or pairs.shape[2] != 2 or not pairs.is_contiguous():
        raise RuntimeError("argmax_merge: contiguous int32 [shards, bs, 2] expected")
    out = torch.empty(pairs.shape[1], dtype=torch.int64, device=pairs.device)
    _check(load().sp_argmax_merge(pairs.data_ptr(), pairs.shape[0], pairs.shape[1], out.data_ptr(), _stream()),
           "sp_argmax_merge")
    return out


def softmax_temperature_(logits: torch.Tensor, temperatures: Optional[torch.Tensor]) -> torch.Tensor:
    """In place: logits <- softmax(logits / T) (fp32 rows)."""
    _gpu(logits, temperatures)
    logits = _prob_rows(logits, "softmax_temperature_")
    if logits.dtype != torch.float32:
        raise RuntimeError("softmax_temperature_: fp32 logits expected (LogitsProcessor returns fp32)")
    temps = _opt(temperatures, torch.float32, logits.shape[0], "temperatures")
    _check(load().sp_softmax_temperature(logits.data_ptr(), logits.stride(0), _ptr(temps), logits.shape[0],
                                         logits.shape[1], _stream()), "sp_softmax_temperature")
    return logits


def top_k_top_p_min_p_sample(probs: torch.Tensor, top_ks: Optional[torch.Tensor],
                             top_ps: Optional[torch.Tensor], min_ps: Optional[torch.Tensor],
                             uniform: torch.Tensor, return_keep_count: bool = False):
    _gpu(probs, top_ks, top_ps, min_ps, uniform)
    probs = _prob_rows(probs, "top_k_top_p_min_p_sample")
    if probs.dtype != torch.float32:
        raise RuntimeError("top_k_top_p_min_p_sample: fp32 probabilities expected")
    bs, vocab = probs.shape
    ks, ps, ms = (_opt(top_ks, torch.int32, bs, "top_ks"), _opt(top_ps, torch.float32, bs, "top_ps"),
                  _opt(min_ps, torch.float32, bs, "min_ps"))
    u = _opt(uniform, torch.float32, bs, "uniform")
    out = torch.empty(bs, dtype=torch.int64, device=probs.device)
    cnt = torch.empty(bs, dtype=torch.int32, device=probs.device) if return_keep_count else None
    _check(load().sp_top_k_top_p_min_p_sample(probs.data_ptr(), probs.stride(0), _ptr(ks), _ptr(ps), _ptr(ms),
                                              u.data_ptr(), bs, vocab, out.data_ptr(), _ptr(cnt), _stream()),
           "sp_top_k_top_p_min_p_sample")
    return (out, cnt) if return_keep_count else out


def top_k_top_p_min_p_renorm(probs: torch.Tensor, top_ks: Optional[torch.Tensor] = None,
                             top_ps: Optional[torch.Tensor] = None, min_ps: Optional[torch.Tensor] = None,
                             return_keep_count: bool = False):
    _gpu(probs, top_ks, top_ps, min_ps)
    probs = _prob_rows(probs, "top_k_top_p_min_p_renorm")
    if probs.dtype != torch.float32:
        raise RuntimeError("top_k_top_p_min_p_renorm: fp32 probabilities expected")
    bs, vocab = probs.shape
    ks, ps, ms = (_opt(top_ks, torch.int32, bs, "top_ks"), _opt(top_ps, torch.float32, bs, "top_ps"),
                  _opt(min_ps, torch.float32, bs, "min_ps"))
    out = torch.empty_like(probs, memory_format=torch.contiguous_format)
    cnt = torch.empty(bs, dtype=torch.int32, device=probs.device) if return_keep_count else None
    _check(load().sp_top_k_top_p_min_p_renorm(probs.data_ptr(), probs.stride(0), _ptr(ks), _ptr(ps), _ptr(ms),
                                              bs, vocab, out.data_ptr(), out.stride(0), _ptr(cnt), _stream()),
           "sp_top_k_top_p_min_p_renorm")
    return (out, cnt) if return_keep_count else out


# --------------------------------------------------------------------------- small-batch projection
SKINNY_MAX_ROWS = 16
_SKINNY_ON = os.environ.get("SP_SKINNY_GEMM", "1") != "0"


def skinny_gemm_pays(M: int, N: int, K: int) -> bool:
    """Where the weight-streaming kernel measured faster than hipBLASLt on MI355X
    (tools/bench_gemv.py, Llama-3-8B shapes): every lane's x fragment is re-read from L2 per k-step,
    so the win shrinks as rows fill the 16-wide tile and with long rows (down_proj)."""
    if M <= 16 and N * K <= 4096 * 4096:
        return True                       # o_proj-sized: 11 vs 19 us at every M <= 16
    if M <= 2 and K <= 8192:
        return True                       # bs 1-2: qkv 13 vs 19, gate_up 44 vs 60, lm_head 166 vs 179 us
    return M <= 8 and K <= 8192 and N <= 65536


# ---- the row count the library is handed -----------------------------------------------------------------------
# hipBLASLt's kernel choice is erratic in the row count M (tools/bench_gemm_rows.py, device time under graph
# replay, cold weights, MI355X / ROCm 7.2): qkv_proj (N 6144, K 4096) takes 40 us at 256 rows and 28 us at 264;
# down_proj (N 4096, K 14336) 103 us at 192 rows and 59 us at 208.  GEMM rows are independent, so for those
# (shape, M) pairs the product of M rows is computed as the first M rows of an M' > M row product: the input is
# re-viewed over M' rows of its OWN storage (every activation this package allocates carries ROW_SLACK spare rows
# behind it; whatever they hold only reaches output rows that are sliced away), no copy, no extra launch.
#
# The table is a measurement of ONE library build, so it is gated on that build (SP_LIBRARY_ROWS):
#   "table" (default) - the table below, used only while torch / HIP report the versions it was measured on;
#   "auto"            - nothing is assumed: ModelRunner.init_cuda_graphs() times every (projection shape, graph
#                       bucket) against M' = M + 8 .. M + ROW_SLACK on the box it runs on and keeps a substitute
#                       only where it measured faster (calibrate_library_rows); on this build that reproduces the
#                       table, on another library it falls back to M wherever the table's entry does not pay;
#   "0"               - off: every product is run with the row count it was asked for.
ROW_SLACK = 64
_LIBRARY_ROWS_MEASURED_ON = {"torch": "2.10", "hip": "7.0"}      # version prefixes of the build the table is for
_LIBRARY_ROWS_TABLE = {
    (6144, 4096): {136: 176, 144: 176, 152: 176, 192: 200, 224: 232, 256: 264},      # qkv_proj  (8B)
    (4096, 4096): {96: 104, 192: 208},                                                # o_proj
    (28672, 4096): {16: 40, 48: 56, 80: 96},                                          # gate_up_proj
    (4096, 14336): {72: 80, 96: 112, 128: 136, 160: 208, 192: 208},                   # down_proj
    # Llama-3-70B shards at TP = 8 (profiles/r02_gemm_rows_70b_tp8.txt); nothing to gain at config 4's 128 rows
    (7168, 8192): {192: 200, 224: 232, 256: 264},                                     # gate_up_proj / 8
    (8192, 3584): {192: 208, 224: 232, 256: 264},                                     # down_proj / 8
    (1280, 8192): {64: 72},                                                           # qkv_proj / 8
}
SLACK_MAX_ROWS = 256          # decode-sized steps only; larger products are exactly F.linear


def library_versions_match() -> bool:
    hip = getattr(torch.version, "hip", None) or ""
    return (torch.__version__.startswith(_LIBRARY_ROWS_MEASURED_ON["torch"])
            and hip.startswith(_LIBRARY_ROWS_MEASURED_ON["hip"]))


def _library_rows_mode() -> str:
    m = os.environ.get("SP_LIBRARY_ROWS", "table").lower()
    return {"1": "table", "on": "table", "off": "0"}.get(m, m)


_LIBROWS_MODE = _library_rows_mode()
if _LIBROWS_MODE == "table" and library_versions_match():
    _LIBRARY_ROWS = {k: dict(v) for k, v in _LIBRARY_ROWS_TABLE.items()}
else:
    _LIBRARY_ROWS = {}        # off, another library build, or "auto" before calibrate_library_rows() has run
_LIBROWS_USED = {}            # (N, K, M) -> M' of every substitution made so far (reported by bench.py)


def library_rows(M: int, N: int, K: int) -> int:
    """rows to hand the library GEMM for an M-row product of shape (N, K): M, or the measured better M' > M"""
    return _LIBRARY_ROWS.get((N, K), {}).get(M, M)


def library_rows_report() -> dict:
    """What bench.py records: the mode, whether the version gate passed, and every substitution that was used."""
    return {"mode": _LIBROWS_MODE, "versions_match": library_versions_match(),
            "measured_on": dict(_LIBRARY_ROWS_MEASURED_ON),
            "running_on": {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)},
            "substituted": {f"N={n} K={k} M={m}": mp for (n, k, m), mp in sorted(_LIBROWS_USED.items())}}


def _time_mm(x: torch.Tensor, weights, reps: int = 3) -> float:
    """device microseconds of x @ W.T, one call per weight captured into a HIP graph (the weights of different
    layers are cycled so that no call finds its weights in L2 / the Infinity Cache, as in the decode step)"""
    outs = [torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device) for w in weights[:2]]
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        for i, w in enumerate(weights):
            torch.mm(x, w.t(), out=outs[i % 2])
        st.synchronize()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            for i, w in enumerate(weights):
                torch.mm(x, w.t(), out=outs[i % 2])
    torch.cuda.synchronize()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(weights)) * 1e3


def calibrate_library_rows(weights_by_shape, rows, min_gain: float = 0.05, step: int = 8) -> dict:
    """SP_LIBRARY_ROWS=auto.  weights_by_shape: {(N, K): [weight tensors of that shape]} (the model's own
    projections); rows: the batch-size buckets of the decode graphs.  Every (shape, M) is timed against
    M' = M + step .. M + ROW_SLACK; a substitute is kept only where it is at least `min_gain` faster than M
    itself ON THIS BOX.  Replaces the active table and returns it."""
    global _LIBRARY_ROWS
    found = {}
    for (N, K), ws in weights_by_shape.items():
        ws = list(ws)[:8]
        if not ws or ws[0].dtype not in (torch.float16, torch.bfloat16) or not ws[0].is_cuda:
            continue
        xbuf = torch.randn((max(rows) + ROW_SLACK, K), device=ws[0].device).to(ws[0].dtype) * 0.1
        cache = {}

        def t_of(m):
            if m not in cache:
                cache[m] = _time_mm(xbuf[:m], ws)
            return cache[m]

        for M in sorted(set(rows)):
            if not 16 < M <= SLACK_MAX_ROWS:        # <= 16 rows: the skinny kernel's territory
                continue
            base = t_of(M)
            best_t, best_m = base, M
            for Mp in range((M // step + 1) * step, M + ROW_SLACK + 1, step):
                t = t_of(Mp)
                if t < best_t:
                    best_t, best_m = t, Mp
            if best_m != M and best_t <= base * (1.0 - min_gain):
                found.setdefault((N, K), {})[M] = best_m
    _LIBRARY_ROWS = found
    return found


def empty_rows(rows: int, cols: int, dtype, device, zero: bool = False) -> torch.Tensor:
    """[rows, cols] activation with ROW_SLACK readable spare rows behind it in the same allocation"""
    full = (torch.zeros if zero else torch.empty)((rows + ROW_SLACK, cols), dtype=dtype, device=device)
    return full[:rows]


def extend_rows(x: torch.Tensor, rows: int) -> Optional[torch.Tensor]:
    """x re-viewed over `rows` >= x.shape[0] rows of its own storage, or None if the storage ends before"""
    if rows <= x.shape[0]:
        return x
    if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        return None
    last = x.storage_offset() + (rows - 1) * x.stride(0) + x.shape[1]
    if last * x.element_size() > x.untyped_storage().nbytes():
        return None
    return torch.as_strided(x, (rows, x.shape[1]), (x.stride(0), 1), x.storage_offset())


_SILU_LINEAR_ON = os.environ.get("SP_SKINNY_SILU", "1") != "0"
# rows up to which the fused form is used.  Measured (round 4, bench.py --ctx 1024, HIP-graph replay, ms/step fused vs
# projection + activation launch): bs 1 3.977 vs 4.006, bs 8 4.469 vs 4.579, bs 16 5.052 vs 4.698 - at 16 rows the
# merged projection is faster on the library (61 vs 68 us) and the activation launch it saves costs ~1 us under replay
SILU_FUSED_MAX_ROWS = 8


def linear_silu_mul(x: torch.Tensor, gate_up_weight: torch.Tensor, any_rows: bool = False) -> Optional[torch.Tensor]:
    """SiluAndMul(x @ gate_up_weight.T) as ONE launch (sp_gemm_skinny, epilogue 1) for a step of at most
    SILU_FUSED_MAX_ROWS tokens (any_rows: up to the kernel's 16): x [M, K], gate_up_weight [2 I, K] (gate rows first),
    result [M, I].  Bit for bit the skinny projection followed by silu_and_mul.  None when the shape is not taken (the
    caller runs the two steps)."""
    if not (_SKINNY_ON and _SILU_LINEAR_ON and x.is_cuda and x.dim() == 2 and gate_up_weight.dim() == 2
            and 0 < x.shape[0] <= (SKINNY_MAX_ROWS if any_rows else SILU_FUSED_MAX_ROWS)
            and gate_up_weight.shape[1] == x.shape[1]
            and gate_up_weight.shape[0] % 16 == 0
            and x.dtype in (torch.float16, torch.bfloat16) and gate_up_weight.dtype == x.dtype
            and x.shape[1] % 32 == 0 and x.stride(1) == 1 and gate_up_weight.stride(1) == 1
            and x.stride(0) % 8 == 0 and gate_up_weight.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and gate_up_weight.data_ptr() % 16 == 0):
        return None
    M, K = x.shape
    inter = gate_up_weight.shape[0] // 2
    out = empty_rows(M, inter, x.dtype, x.device)
    _check(load().sp_gemm_skinny(out.data_ptr(), x.data_ptr(), gate_up_weight.data_ptr(), M, inter, K, x.stride(0),
                                 gate_up_weight.stride(0), out.stride(0), 1, _dt(x), _stream()), "sp_gemm_skinny(silu)")
    return out


def linear(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """x @ weight.T.  Small-batch 16-bit products go to the weight-streaming kernel
    (sp_gemm_skinny) where it pays; everything else is the library GEMM - exactly F.linear, except
    that a product of at most SLACK_MAX_ROWS rows whose (shape, rows) is in the active library_rows
    table is computed over more rows than asked for (first M rows returned; rows are independent)."""
    if (_SKINNY_ON and x.is_cuda and x.dim() == 2 and 0 < x.shape[0] <= SKINNY_MAX_ROWS
            and skinny_gemm_pays(x.shape[0], weight.shape[0], x.shape[1])
            and x.dtype in (torch.float16, torch.bfloat16) and weight.dtype == x.dtype
            and x.shape[1] % 32 == 0 and x.stride(1) == 1 and weight.stride(1) == 1
            and x.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0):
        M, K = x.shape
        N = weight.shape[0]
        out = empty_rows(M, N, x.dtype, x.device)
        _check(load().sp_gemm_skinny(out.data_ptr(), x.data_ptr(), weight.data_ptr(), M, N, K, x.stride(0),
                                     weight.stride(0), out.stride(0), 0, _dt(x), _stream()), "sp_gemm_skinny")
        return out
    if (_LIBRARY_ROWS and x.is_cuda and x.dim() == 2 and 0 < x.shape[0] <= SLACK_MAX_ROWS and weight.dim() == 2
            and x.dtype in (torch.float16, torch.bfloat16) and weight.dtype == x.dtype and x.stride(1) == 1):
        M, K = x.shape
        N = weight.shape[0]
        Mp = library_rows(M, N, K)
        xe = extend_rows(x, Mp) if Mp > M else x
        if xe is None:                      # producer without spare rows (e.g. the embedding): run as asked
            xe, Mp = x, M
        # the output carries spare rows only where another measured projection may consume it (one whose input
        # width is this output's width); anything else - the LM head above all - is exactly F.linear
        if Mp > M or _feeds_a_measured_shape(N):
            if Mp > M:
                _LIBROWS_USED[(N, K, M)] = Mp
            out = torch.empty((Mp + ROW_SLACK, N), dtype=x.dtype, device=x.device)
            torch.mm(xe, weight.t(), out=out[:Mp])
            return out[:M]
    return torch.nn.functional.linear(x, weight)


def _feeds_a_measured_shape(width: int) -> bool:
    return any(k == width for (_n, k) in _LIBRARY_ROWS)
