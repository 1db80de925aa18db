"""CPU checks of the drop-in boundary: the C-ABI library loads and exports exactly the symbols
include/scratchpad_hip.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "scratchpad_hip.h")).read()
    return sorted(set(re.findall(r"^SP_API [^;(]*?\b(sp_\w+)\s*\(", text, flags=re.M)))


@pytest.fixture(scope="module")
def lib():
    from scratchpad_amd import _native, build
    build.build_native(verbose=False)
    return ctypes.CDLL(_native.lib_path())


def test_header_declares_the_whole_path():
    syms = header_symbols()
    for s in ["sp_rmsnorm", "sp_fused_add_rmsnorm", "sp_silu_and_mul", "sp_rotary_embedding",
              "sp_kv_store", "sp_write_req_to_token", "sp_compute_position", "sp_clamp_position",
              "sp_decode_attention", "sp_decode_attention_workspace_bytes", "sp_decode_plan",
              "sp_decode_plan_bytes", "sp_decode_ranges", "sp_extend_attention",
              "sp_extend_attention_workspace_bytes", "sp_abi_version", "sp_status_string",
              "sp_argmax", "sp_softmax_temperature", "sp_top_k_top_p_min_p_sample",
              "sp_top_k_top_p_min_p_renorm"]:
        assert s in syms


def test_library_exports_every_declared_symbol(lib):
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in the header but not exported"


def test_python_binding_covers_every_symbol():
    from scratchpad_amd import _native
    assert sorted(_native.SIGNATURES) == header_symbols()


def test_abi_version_and_status_strings(lib):
    lib.sp_abi_version.restype = ctypes.c_int
    assert lib.sp_abi_version() == 9
    lib.sp_status_string.restype = ctypes.c_char_p
    assert lib.sp_status_string(0) == b"ok"
    assert b"unsupported" in lib.sp_status_string(-2)


def test_host_side_argument_validation_needs_no_gpu(lib):
    # null pointers / bad sizes are rejected before any launch
    lib.sp_rmsnorm.restype = ctypes.c_int
    lib.sp_rmsnorm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                                      ctypes.c_int64, ctypes.c_float, ctypes.c_int,
                                                      ctypes.c_void_p]
    assert lib.sp_rmsnorm(None, None, None, 4, 64, 64, 64, 1e-5, 2, None) == -1
    # split geometry (ABI 4): slots = min(bs * ceil(max_seq_len / chunk), kv_tokens / chunk + bs); the workspace
    # holds [slots, Hq, D + 1] floats - bounded by the step's tokens, not by batch x context
    lib.sp_decode_plan_slots.restype = ctypes.c_int64
    lib.sp_decode_plan_slots.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    assert lib.sp_decode_plan_slots(256, -1, 4096, 512) == 256 * 8
    assert lib.sp_decode_plan_slots(256, 547000, 4096, 512) == 547000 // 512 + 256
    assert lib.sp_decode_plan_slots(8, 800000, 131072, 1024) == 800000 // 1024 + 8
    assert lib.sp_decode_plan_slots(4, 10 ** 9, 100, 512) == 4
    lib.sp_decode_attention_workspace_bytes.restype = ctypes.c_size_t
    lib.sp_decode_attention_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    assert lib.sp_decode_attention_workspace_bytes(256 * 8, 32, 128) == 256 * 32 * 8 * 129 * 4 + 16
    assert lib.sp_decode_attention_workspace_bytes(0, 8, 64) == 16
    lib.sp_decode_plan_bytes.restype = ctypes.c_size_t
    lib.sp_decode_plan_bytes.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    # a 4-word header (listed, chunk, needed, keys), slot0[bs], the item pairs (ABI 7: no arrival counters behind them)
    assert lib.sp_decode_plan_bytes(256, 2304, 0) == (4 + 256 + 2 * 2304) * 4
    # ABI 8: with ranges, the range geometry behind them: [pieces, R, ranges, bs], pos[bs + 1], start[ranges]
    assert lib.sp_decode_plan_bytes(256, 2304, 384) == (4 + 256 + 2 * 2304 + 4 + 257 + 384) * 4
    # ABI 9: max_slots = 0 - no items: the item header alone in front of the range section; neither section: nothing
    assert lib.sp_decode_plan_bytes(256, 0, 384) == (4 + 4 + 257 + 384) * 4
    assert lib.sp_decode_plan_bytes(256, 0, 0) == 16
    # ... and a launch is refused on the host, before anything touches a device, when its plan buffer is shorter than
    # the sections (batch_size, max_slots, ranges) describe, or when the plan has neither section
    from scratchpad_amd import _native
    lib.sp_decode_attention.restype, lib.sp_decode_attention.argtypes = _native.SIGNATURES["sp_decode_attention"]
    buf = (ctypes.c_char * 4096)()
    a = ctypes.addressof(buf)
    a += (-a) % 16

    def launch(max_slots, ranges, plan_bytes):
        return lib.sp_decode_attention(a, a, a, a, a, 64, a, a, None, 0, 256, 32, 8, 128, 4096, 4096, 2048, 0.1, 0.0, 1.0,
                                       1.0, 4096, 64, max_slots, ranges, a, 1 << 40, a, plan_bytes, 2, 2, None)
    assert launch(2304, 384, (4 + 256 + 2 * 2304 + 4 + 257 + 384) * 4 - 4) == -3       # SP_ERR_WORKSPACE
    assert launch(0, 384, (4 + 4 + 257 + 384) * 4 - 4) == -3
    assert launch(0, 0, 1 << 20) == -1                                                   # SP_ERR_INVALID_ARG


def test_ops_refuse_host_tensors():
    import torch
    from scratchpad_amd import _native
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.rmsnorm(torch.zeros(2, 64), torch.ones(64), 1e-5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.silu_and_mul(torch.zeros(2, 64))


def test_kv_pool_views_are_checked_before_a_launch():
    """the ABI carries ONE token stride for K and V and assumes contiguous heads per token; the wrappers
    refuse anything else instead of gathering garbage (host-side check, no launch)."""
    import torch
    from scratchpad_amd import _native
    from scratchpad_amd.pool import MHATokenToKVPool
    for interleave in (True, False):
        old = MHATokenToKVPool.interleave_kv
        MHATokenToKVPool.interleave_kv = interleave
        try:
            pool = MHATokenToKVPool(10, 1, torch.bfloat16, 2, 64, 2, "cpu")
        finally:
            MHATokenToKVPool.interleave_kv = old
        _native._kv_layout(pool.get_key_buffer(1), pool.get_value_buffer(1), "test")          # both layouts pass
    k = torch.zeros(11, 2, 64)
    with pytest.raises(RuntimeError, match="same token stride"):
        _native._kv_layout(k, torch.zeros(11, 4, 64)[:, :2], "test")
    _native._kv_layout(k, torch.zeros(11, 4, 64)[:, :2], "test", same_stride=False)           # the store takes two strides
    with pytest.raises(RuntimeError, match="contiguous heads"):
        _native._kv_layout(k, torch.zeros(11, 64, 2).transpose(1, 2), "test")
    with pytest.raises(RuntimeError, match="contiguous heads"):
        _native._kv_layout(torch.zeros(11, 2, 128)[:, :, :64], k, "test")
    with pytest.raises(RuntimeError, match="disagree"):
        _native._kv_layout(k, torch.zeros(12, 2, 64), "test")
