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

// extend_mfma.hip: MFMA tile kernel for 16-bit ragged extend; SP_ERR_UNSUPPORTED -> use row-streams
int run_extend_mfma(void* out, const void* q, const void* k_buffer, const void* v_buffer,
                    const int32_t* req_to_token, int64_t req_to_token_stride,
                    const void* req_pool_indices, const void* seq_lens, const void* kv_start,
                    int idx64, const int32_t* extend_seq_lens, const int32_t* extend_start_loc,
                    int batch_size, int num_q_heads, int num_kv_heads, int head_dim, int64_t q_stride,
                    int64_t out_stride, int64_t kv_buffer_stride, float sm_scale, float logit_cap,
                    float out_scale, int causal, int window_left, int max_extend_len, const int32_t* plan,
                    int plan_items, int dtype, int kv8, hipStream_t st);

// test / tuning hooks behind sp_debug_set
void set_decode_kernel(int which);
void set_extend_defer_x10(int tenths);
void set_extend_dma(int v);

}  // namespace sp
