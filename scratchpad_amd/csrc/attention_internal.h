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
  int max_len;      // num_splits * chunk: device-side seq_lens are clamped to it
  int chunk, num_splits, hh_shift, head_groups;
  float* part_o;    // [bs, Hq, num_splits, D]
  float* part_lse;  // [bs, Hq, num_splits]  (log2 domain)
  int kv8;              // 1: the pool holds fp8 e5m2 bytes (kv_stride in bytes); 16-bit q/out only
  const int32_t* plan;  // optional: [count, chunk, (b, c) x count] from sp_decode_plan
};

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
