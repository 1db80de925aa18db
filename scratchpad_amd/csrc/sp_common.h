// Shared device helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/scratchpad_hip.h"

namespace sp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct f16_tag {};
struct bf16_tag {};
struct f32_tag {};

// ---- bit casts of 32-bit lanes.  NOTE: never apply __builtin_bit_cast directly to an
// ext_vector element expression (`v[i]`): hipcc (ROCm 7.2) then reads element 0 for every i.
// These helpers take the lane BY VALUE.
__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ bf16x2_t as_bf16x2(uint32_t u) { return __builtin_bit_cast(bf16x2_t, u); }
__device__ __forceinline__ f16x2_t as_f16x2(uint32_t u) { return __builtin_bit_cast(f16x2_t, u); }

// ---- scalar conversions -------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) {
  return __builtin_bit_cast(float, lo16 << 16);
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 b = (__bf16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t lo16) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)lo16);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

template <typename Tag>
struct Elem;
template <>
struct Elem<f32_tag> {
  static constexpr int kBytes = 4;
  static constexpr int kVec = 4;  // elements per 16-byte lane access
  typedef float storage;
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
template <>
struct Elem<f16_tag> {
  static constexpr int kBytes = 2;
  static constexpr int kVec = 8;
  typedef uint16_t storage;
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return f16_bits_to_f32(((const uint16_t*)p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) {
    ((uint16_t*)p)[i] = (uint16_t)f32_to_f16_bits(v);
  }
  static __device__ __forceinline__ float round(float v) { return (float)(_Float16)v; }
};
template <>
struct Elem<bf16_tag> {
  static constexpr int kBytes = 2;
  static constexpr int kVec = 8;
  typedef uint16_t storage;
  static __device__ __forceinline__ float load(const void* p, int64_t i) {
    return bf16_bits_to_f32(((const uint16_t*)p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) {
    ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16_bits(v);
  }
  static __device__ __forceinline__ float round(float v) { return (float)(__bf16)v; }
};

// ---- 16-byte vector <-> fp32 lanes --------------------------------------------------------
template <typename Tag>
__device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <>
__device__ __forceinline__ void unpack16<f32_tag>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = as_f32(v[i]);
}
template <>
__device__ __forceinline__ void unpack16<bf16_tag>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = as_f32(v[i] << 16);
    f[2 * i + 1] = as_f32(v[i] & 0xffff0000u);
  }
}
template <>
__device__ __forceinline__ void unpack16<f16_tag>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2_t h = as_f16x2(v[i]);
    f[2 * i] = (float)h[0];
    f[2 * i + 1] = (float)h[1];
  }
}

template <typename Tag>
__device__ __forceinline__ u32x4 pack16(const float* f);
template <>
__device__ __forceinline__ u32x4 pack16<f32_tag>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = as_u32(f[i]);
  return v;
}
template <>
__device__ __forceinline__ u32x4 pack16<bf16_tag>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x2_t b;
    b[0] = (__bf16)f[2 * i];
    b[1] = (__bf16)f[2 * i + 1];
    v[i] = __builtin_bit_cast(uint32_t, b);
  }
  return v;
}
template <>
__device__ __forceinline__ u32x4 pack16<f16_tag>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x2_t h;
    h[0] = (_Float16)f[2 * i];
    h[1] = (_Float16)f[2 * i + 1];
    v[i] = __builtin_bit_cast(uint32_t, h);
  }
  return v;
}

// SiLU as the reference's torch path evaluates it in fp32 (nn/layers/activation.py:22-24): a / (1 + exp(-a)), every
// operation rounded on its own whatever the translation unit's contraction mode (shared by sp_silu_and_mul and the
// fused prologue of sp_gemm_skinny, which must agree bit for bit)
__device__ __forceinline__ float silu_ref(float a) { return __fdiv_rn(a, __fadd_rn(1.0f, expf(-a))); }

__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *(u32x4*)p = v; }

// ---- fp8 e5m2 (1-5-2 = the top byte of an IEEE half) ---------------------------------------------
// 8 e5m2 bytes -> 8 halves, exactly: each byte becomes the high byte of a 16-bit lane (v_perm_b32)
__device__ __forceinline__ u32x4 expand_e5m2x8(const u32x2& w) {
  u32x4 o;
  o[0] = __builtin_amdgcn_perm(w[0], w[0], 0x010c000cu);   // [0, b0, 0, b1]
  o[1] = __builtin_amdgcn_perm(w[0], w[0], 0x030c020cu);   // [0, b2, 0, b3]
  o[2] = __builtin_amdgcn_perm(w[1], w[1], 0x010c000cu);
  o[3] = __builtin_amdgcn_perm(w[1], w[1], 0x030c020cu);
  return o;
}
// float -> e5m2 byte: one round-to-nearest-even from fp32 (normal and subnormal results), overflow
// -> inf, NaN -> NaN - the arithmetic of torch's .to(torch.float8_e5m2)
__device__ __forceinline__ uint32_t f32_to_e5m2_bits(float f) {
  uint32_t x = as_u32(f);
  const uint32_t sign = x & 0x80000000u;
  x ^= sign;
  uint32_t r;
  if (x >= (143u << 23)) {                       // |f| >= 2^16, inf, NaN
    r = x > (0xffu << 23) ? 0x7fu : 0x7cu;
  } else if (x < (113u << 23)) {                 // below 2^-14: the result is an e5m2 subnormal
    const float aligned = as_f32(x) + as_f32(134u << 23);   // ulp of (|f| + 128) is 2^-16
    r = as_u32(aligned) - (134u << 23);
  } else {
    const uint32_t odd = (x >> 21) & 1u;
    x += ((uint32_t)(15 - 127) << 23) + 0xfffffu + odd;     // rebias, round half to even
    r = x >> 21;
  }
  return (r & 0xffu) | (sign >> 24);
}
// 8 bf16 -> 8 halves (exact for |x| in the half's normal range)
__device__ __forceinline__ u32x4 bf16x8_to_f16x8(const u32x4& v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16x2_t h;
    h[0] = (_Float16)as_f32(v[i] << 16);
    h[1] = (_Float16)as_f32(v[i] & 0xffff0000u);
    o[i] = __builtin_bit_cast(uint32_t, h);
  }
  return o;
}

// ---- wave reductions ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// index arrays that are int64 in eager mode and int32 under graph replay
__device__ __forceinline__ int64_t load_idx(const void* p, int i, int idx64) {
  return idx64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}

// Host-side state that belongs to ONE device of the process (a kernel's dynamic-LDS limit raised by hipFuncSetAttribute,
// the occupancy the runtime reports for it): a slot per device id, filled once under a lock, by whichever thread gets
// there first - the overlap worker's forward thread and the scheduler thread both launch.  A function-local
// `static` flag would serve the device that was current on first use only.
template <typename V>
struct PerDevice {
  static constexpr int kMaxDevices = 64;
  std::mutex mu;
  bool known[kMaxDevices] = {};
  V value[kMaxDevices] = {};
  // value of the current device; `make(dev)` runs once per device.  Returns false when no device is current
  template <typename F>
  bool get(V& out, F make) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (!known[dev]) {
      value[dev] = make(dev);
      known[dev] = true;
    }
    out = value[dev];
    return true;
  }
};

}  // namespace sp

#define SP_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return SP_ERR_INVALID_ARG; \
  } while (0)

#define SP_LAUNCH_CHECK()                              \
  do {                                                 \
    if (hipGetLastError() != hipSuccess) return SP_ERR_LAUNCH; \
  } while (0)

#define SP_DISPATCH_DTYPE(dtype, ...)                                   \
  do {                                                                  \
    if ((dtype) == SP_F32) { typedef sp::f32_tag Tag; __VA_ARGS__; }    \
    else if ((dtype) == SP_F16) { typedef sp::f16_tag Tag; __VA_ARGS__; } \
    else if ((dtype) == SP_BF16) { typedef sp::bf16_tag Tag; __VA_ARGS__; } \
    else return SP_ERR_UNSUPPORTED;                                     \
  } while (0)
