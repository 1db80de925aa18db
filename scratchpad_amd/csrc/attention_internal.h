// Internal (non-ABI) declarations shared by the attention translation units.
#pragma once
#include "sp_common.h"

namespace sp {

struct DecodeArgs {
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
  int bs, Hq, Hkv;
  int64_t q_stride, o_stride, kv_stride;  // elements
  float sm_scale, logit_cap;
  float out_scale;  // multiplies the normalised output (the pool's v_scale; 1 = none)
  int max_len;      // host-side bound of a request's kv length: device-side seq_lens are clamped to it
  int chunk, num_splits, hh_shift, head_groups;   // chunk / num_splits: the plan-less geometry (plan: plan[1])
  int max_slots;    // partial slots the workspace holds; work items the launch covers when planned
  float* part_o;    // [Hq, max_slots, D]   slot of (request b, split c): slot0[b] + c (a request's splits are adjacent)
  float* part_lse;  // [Hq, max_slots]      (log2 domain)
  int kv8;              // 1: the pool holds fp8 e5m2 bytes (kv_stride in bytes); 16-bit q/out only
  // optional, from sp_decode_plan: [count, chunk, slot0[bs], (b, c) x count].  The CHUNK is part of the
  // plan (device memory), so a captured launch follows whatever split size the step's plan was built
  // with; slot0[b] = first partial slot of request b (exclusive scan of its split count).
  const int32_t* plan;
};

// (request, split) of work item `item`, the split size, and the request's first partial slot
__device__ __forceinline__ bool decode_item(const DecodeArgs& a, int item, int& b, int& c, int& chunk, int& slot0) {
  if (a.plan) {
    if (item >= a.plan[0]) return false;
    chunk = a.plan[1];
    const int32_t* items = a.plan + 2 + a.bs;
    b = items[2 * item];
    c = items[2 * item + 1];
    slot0 = a.plan[2 + b];
  } else {
    chunk = a.chunk;
    c = item % a.num_splits;
    b = item / a.num_splits;
    slot0 = b * a.num_splits;
  }
  return true;
}

// decode_attention.hip: launch the split-KV decode kernel (+ merge when num_splits > 1)
int run_decode(const DecodeArgs& a, int head_dim, int group, int dtype, hipStream_t st);
// decode_attention.hip: merge of the split partials (shared by the VALU and MFMA decode kernels)
int run_decode_merge(const DecodeArgs& a, int head_dim, int dtype, hipStream_t st);
// decode_mfma.hip: matrix-core decode kernel for 16-bit dtypes, G <= 16 (attention kernel only)
int run_decode_mfma(const DecodeArgs& a, int head_dim, int dtype, hipStream_t st);
// heads per wave-load: the largest power of two <= rows-per-load that divides Hkv
int decode_heads_per_load_shift(int num_kv_heads, int head_dim, int dtype, int* head_groups);

// test / tuning hooks behind sp_debug_set
void set_decode_kernel(int which);

}  // namespace sp
