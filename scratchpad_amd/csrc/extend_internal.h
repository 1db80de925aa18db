// Types and helpers of the MFMA extend kernel (extend_mfma.hip).  Internal to the library.
#pragma once
#include <type_traits>

#include "attention_internal.h"
#include "extend_api.h"

namespace sp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

struct ExtendArgs {
  void* out;
  const void* q;
  const char* kbuf;
  const char* vbuf;
  const int32_t* r2t;
  int64_t r2t_stride;
  const void* req_idx;
  const void* seq_lens;
  const void* kv_start;
  int idx64;
  const int32_t* ext_lens;
  const int32_t* ext_start;
  int bs, Hq, Hkv;
  int64_t q_stride, o_stride, kv_stride;  // elements
  float sm_scale, logit_cap, out_scale;   // out_scale = the pool's v_scale (1 for unscaled pools)
  int causal;
  int kv8;      // 1: fp8 e5m2 pool (kv_stride in bytes); tile math in fp16, see decode_mfma.hip
  int window;   // sliding window: a row at kv position p sees keys [p - window, p]; < 0 = unlimited
  float defer;  // the running maximum may trail the true row maximum by this much (log2 units)
  // optional work list from sp_extend_plan: header [count, BM, Hq, Hkv, num_tokens, bs, 0, 0], then
  // (request, row block) x count from word kExtPlanHeader.  A workgroup that finds the header built for
  // another block size / head counts / token total / batch size derives its item by walking the requests instead
  // (correct, merely unordered).  The header identifies a launch SHAPE, not a step: a plan of an earlier step with
  // the same shape but other per-request lengths is not detected - the host rebuilds the plan every step
  // (scratchpad_hip.h, sp_extend_plan).
  const int32_t* plan;
  int plan_items;       // grid rows when a plan is given (an upper bound of its count)
  int num_tokens;       // sum of the extend lengths (host-known)
};

static constexpr float kLog2eX = 1.4426950408889634f;
static constexpr float kNegBigX = -1.0e30f;
// the running row maximum trails the true one by at most this much (log2 units): P <= 2^6
static constexpr float kDeferLog2 = 6.0f;

template <typename Tag>
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c);
template <>
__device__ __forceinline__ f32x16 mfma32<bf16_tag>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<f16_tag>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename Tag>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2<bf16_tag>(float lo, float hi) {
  bf16x2_t b;
  b[0] = (__bf16)lo;
  b[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, b);
}
template <>
__device__ __forceinline__ uint32_t pack2<f16_tag>(float lo, float hi) {
  f16x2_t b;
  b[0] = (_Float16)lo;
  b[1] = (_Float16)hi;
  return __builtin_bit_cast(uint32_t, b);
}

// value held by lane ^ 32, by one v_permlane32_swap (VALU) instead of a ds_bpermute round trip:
// the swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second
__device__ __forceinline__ float xchg32(float x) {
  const uint32_t u = as_u32(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const uint32_t upper_gets = r[0], lower_gets = r[1];
  return as_f32((threadIdx.x & 32) ? upper_gets : lower_gets);
}

template <int D, int NT>
struct ExtCfg {
  static constexpr int BN = 64;                    // keys per tile
  static constexpr int ROW_B = D * 2;              // bytes per K/V row
  static constexpr int SK = ROW_B + 16;            // K row stride in LDS
  static constexpr int SV = ROW_B + 64;            // V row stride in LDS
  static constexpr int CPR = ROW_B / 16;           // 16-byte chunks per row
  static constexpr int RPP = NT / CPR;             // rows staged per pass of the NT threads
  static constexpr int PASSES = BN / RPP;
  static constexpr int KSTEPS = D / 16;            // MFMA k-steps of Q.K^T
  static constexpr int DBLK = D / 32;              // 32-wide d blocks of O^T
  static constexpr int kTileBytes = BN * (SK + SV);   // one K tile + one V tile
  static_assert(RPP <= BN && BN % RPP == 0, "staging geometry");
};

}  // namespace sp
