// Ragged extend (prefill with cached prefix) attention for gfx950.
//
// Replaces extend_attention_fwd (nn/attention/triton_attn/extend_attention.py:16-327) and the
// flashinfer ragged + paged + merge_state path (nn/attention/flashinfer_backend.py:400-444).
//
// Path in this file (all dtypes, every supported head shape): "row-streams".  New token t of
// request b is an independent softmax stream over kv positions [0, prefix_b + t] (causal) or
// [0, seq_b) (cross-attention), i.e. exactly one decode-attention row whose request index,
// length and kv offset are per token.  A tiny expand kernel writes those three int32 arrays
// and the decode kernel (decode_attention.hip: coalesced full-row gathers, G query heads per
// KV read) does the rest.  KV traffic is O(sum L^2) instead of O(sum L^2 / BLOCK_M); the MFMA
// tile kernel for long 16-bit prompts replaces it where that matters.
#include <algorithm>
#include <cstring>

#include "attention_internal.h"
#include "extend_api.h"

namespace sp {

// per new token: request row, visible kv length, kv offset
__global__ __launch_bounds__(256) void expand_rows_kernel(
    int32_t* __restrict__ row_req, int32_t* __restrict__ row_len, int32_t* __restrict__ row_kv0,
    const void* __restrict__ req_pool_indices, const void* __restrict__ seq_lens,
    const void* __restrict__ kv_start, int idx64, const int32_t* __restrict__ extend_seq_lens,
    const int32_t* __restrict__ extend_start_loc, int causal, int window, int64_t num_tokens) {
  const int b = blockIdx.x;
  const int e = extend_seq_lens[b];
  const int64_t s0 = extend_start_loc[b];
  const int seq = (int)load_idx(seq_lens, b, idx64);
  const int req = (int)load_idx(req_pool_indices, b, idx64);
  const int kv0 = kv_start ? (int)load_idx(kv_start, b, idx64) : 0;
  const int prefix = seq - e;
  for (int i = threadIdx.x; i < e; i += 256) {
    const int64_t t = s0 + i;
    if (t >= num_tokens) break;  // never write beyond the caller's buffers
    row_req[t] = req;
    const int vis = causal ? prefix + i + 1 : seq;                 // keys [0, vis)
    const int len = (causal && window >= 0) ? min(vis, window + 1) : vis;   // ... the last `len` of them
    row_len[t] = len;
    row_kv0[t] = kv0 + (vis - len);
  }
}

// Work list of the MFMA extend kernel: the (request, row block) items of a step.  Requests are ordered
// longest first (the launch ends on its shortest rows), and all row blocks of one request stay
// together, last rows first: consecutive workgroups of a kv head (= of an XCD, see extend_mfma.hip) then
// walk the SAME request's keys and share them in that XCD's L2.  (Sorting the items themselves by cost
// interleaves the requests and measured 10 % slower: every XCD then streams many requests at once.)
// One workgroup: a histogram of item counts over the requests' cost classes (64-key tiles), a scan
// from the highest class down, and one ticket per request for its contiguous run of items.
constexpr int kPlanThreads = 1024;
constexpr int kPlanBins = 4096;         // requests above 262k keys share the top class
__global__ __launch_bounds__(kPlanThreads) void extend_plan_kernel(
    int32_t* __restrict__ plan, int max_items, const int32_t* __restrict__ extend_seq_lens,
    const void* __restrict__ seq_lens, int idx64, int bs, int block_rows, int num_q_heads, int num_kv_heads,
    int num_tokens) {
  __shared__ int s_bin[kPlanBins];
  __shared__ int s_total;
  for (int i = threadIdx.x; i < kPlanBins; i += kPlanThreads) s_bin[i] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < bs; b += kPlanThreads) {
    const int E = extend_seq_lens[b];
    const int L = (int)load_idx(seq_lens, b, idx64);
    const int nblk = E > 0 ? (E + block_rows - 1) / block_rows : 0;
    if (nblk > 0) atomicAdd(&s_bin[min((max(L, 0) + 63) / 64, kPlanBins - 1)], nblk);
  }
  __syncthreads();
  if (threadIdx.x == 0) {            // start offsets, longest class first (4096 adds: microseconds)
    int run = 0;
    for (int c = kPlanBins - 1; c >= 0; --c) {
      const int n = s_bin[c];
      s_bin[c] = run;
      run += n;
    }
    s_total = run;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bs; b += kPlanThreads) {
    const int E = extend_seq_lens[b];
    const int L = (int)load_idx(seq_lens, b, idx64);
    const int nblk = E > 0 ? (E + block_rows - 1) / block_rows : 0;
    if (nblk == 0) continue;
    const int pos = atomicAdd(&s_bin[min((max(L, 0) + 63) / 64, kPlanBins - 1)], nblk);
    for (int i = 0; i < nblk; ++i) {
      if (pos + i < max_items) {
        plan[kExtPlanHeader + 2 * (pos + i)] = b;
        plan[kExtPlanHeader + 1 + 2 * (pos + i)] = nblk - 1 - i;      // the request's last rows (most keys) first
      }
    }
  }
  // the persistent kernel's ticket and completion counters (extend_w64.hip): zero between launches
  for (int i = threadIdx.x; i < kExtPlanTicketWords; i += kPlanThreads) plan[kExtPlanHeader + 2 * max_items + i] = 0;
  if (threadIdx.x == 0) {
    // header: what the plan was built for (the attention kernel checks it against its own launch)
    plan[0] = min(s_total, max_items);
    plan[1] = block_rows;
    plan[2] = num_q_heads;
    plan[3] = num_kv_heads;
    plan[4] = num_tokens;
    plan[5] = bs;
    plan[6] = plan[7] = 0;
  }
}

int extend_block_rows(int num_q_heads, int num_kv_heads);   // extend_mfma.hip
void set_ar_fused_blocks(int n);                            // allreduce.hip
void set_skinny_nt(int v);                                  // gemm_skinny.hip
void set_skinny_unroll16(int v);                            // gemm_skinny.hip

}  // namespace sp

using namespace sp;

// items of a step: sum over requests of ceil(extend_len / block_rows) <= num_tokens / block_rows + bs
static inline int64_t extend_plan_items(int64_t num_tokens, int batch_size, int block_rows) {
  return num_tokens / block_rows + batch_size;
}

extern "C" size_t sp_extend_plan_bytes(int64_t num_tokens, int batch_size, int num_q_heads, int num_kv_heads) {
  if (num_tokens <= 0 || batch_size <= 0 || num_q_heads <= 0 || num_kv_heads <= 0) return 16;
  const int bm = extend_block_rows(num_q_heads, num_kv_heads);
  return (size_t)(kExtPlanHeader + 2 * extend_plan_items(num_tokens, batch_size, bm) + kExtPlanTicketWords) * sizeof(int32_t);
}

extern "C" int sp_extend_plan(int32_t* plan, size_t plan_bytes, const int32_t* extend_seq_lens,
                              const void* seq_lens, int idx64, int batch_size, int64_t num_tokens,
                              int num_q_heads, int num_kv_heads, int causal, void* stream) {
  SP_CHECK_ARG(plan && extend_seq_lens && seq_lens && batch_size >= 0 && num_tokens >= 0);
  SP_CHECK_ARG(num_q_heads > 0 && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0);
  if (plan_bytes < sp_extend_plan_bytes(num_tokens, batch_size, num_q_heads, num_kv_heads)) return SP_ERR_WORKSPACE;
  const int bm = extend_block_rows(num_q_heads, num_kv_heads);
  const int64_t items = extend_plan_items(num_tokens, batch_size, bm);
  if (items > 0x3fffffffLL || num_tokens > 0x7fffffffLL) return SP_ERR_INVALID_ARG;
  (void)causal;   // reserved: the item list does not depend on it (kept so that callers state what they plan for)
  extend_plan_kernel<<<dim3(1), kPlanThreads, 0, (hipStream_t)stream>>>(plan, (int)items, extend_seq_lens, seq_lens,
                                                                      idx64, batch_size, bm, num_q_heads,
                                                                      num_kv_heads, (int)num_tokens);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_debug_set(const char* key, int value) {
  SP_CHECK_ARG(key);
  if (!strcmp(key, "decode_kernel")) { set_decode_kernel(value); return SP_OK; }
  if (!strcmp(key, "decode_nt_min_mb")) { set_decode_nt_min_mb(value); return SP_OK; }
  if (!strcmp(key, "decode_ranges")) { set_decode_ranges(value); return SP_OK; }
  if (!strcmp(key, "extend_defer_x10")) { set_extend_defer_x10(value); return SP_OK; }
  if (!strcmp(key, "extend_dma")) { set_extend_dma(value); return SP_OK; }
  if (!strcmp(key, "extend_w64")) { set_extend_w64(value); return SP_OK; }
  if (!strcmp(key, "extend_w64_persist")) { set_extend_w64_persist(value); return SP_OK; }
  if (!strcmp(key, "ar_fused_blocks")) { set_ar_fused_blocks(value); return SP_OK; }
  if (!strcmp(key, "skinny_nt")) { set_skinny_nt(value); return SP_OK; }
  if (!strcmp(key, "skinny_unroll16")) { set_skinny_unroll16(value); return SP_OK; }
  return SP_ERR_INVALID_ARG;
}

// read-only counterpart: "w64_descriptor_patched" (1 = the staged build, build.py compile_w64, has sized the 4-wave x 64-row
// kernels' register allocation in this library: they may launch), "extend_last_kernel" (see extend_api.h);
// -1 for an unknown key
extern "C" int sp_debug_get(const char* key) {
  if (!key) return -1;
  if (!strcmp(key, "w64_descriptor_patched")) return w64_descriptor_patched();
  if (!strcmp(key, "extend_last_kernel")) return g_extend_last_kernel;
  if (!strcmp(key, "decode_last_kernel")) return g_decode_last_kernel;
  return -1;
}

#if defined(SP_EXTEND_STAMPS) || defined(SP_EXTEND_WGSTAMPS)
namespace sp { void set_extend_stamp_buffer(void* p); }
extern "C" SP_API int sp_debug_extend_stamp_buffer(void* device_u64x8) {
  sp::set_extend_stamp_buffer(device_u64x8);
  return SP_OK;
}
#endif

extern "C" size_t sp_extend_attention_workspace_bytes(int64_t num_tokens, int batch_size,
                                                      int num_q_heads, int head_dim, int dtype) {
  (void)batch_size; (void)num_q_heads; (void)head_dim; (void)dtype;
  if (num_tokens <= 0) return 16;
  return (size_t)num_tokens * 3 * sizeof(int32_t) + 64;
}

extern "C" int sp_extend_attention(void* out, const void* q, const void* k_buffer,
                                   const void* v_buffer, const int32_t* req_to_token,
                                   int64_t req_to_token_stride, const void* req_pool_indices,
                                   const void* seq_lens, const void* kv_start, int idx64,
                                   const int32_t* extend_seq_lens,
                                   const int32_t* extend_start_loc, int batch_size,
                                   int64_t num_tokens, int num_q_heads, int num_kv_heads,
                                   int head_dim, int64_t q_stride, int64_t out_stride,
                                   int64_t kv_buffer_stride, float sm_scale, float logit_cap,
                                   float k_scale, float v_scale, int causal, int window_left,
                                   int max_extend_len,
                                   int64_t max_seq_len, void* workspace, size_t workspace_bytes,
                                   int32_t* plan, size_t plan_bytes, int dtype, int kv_dtype,
                                   void* stream) {
  SP_CHECK_ARG(out && q && k_buffer && v_buffer && req_to_token && req_pool_indices && seq_lens);
  SP_CHECK_ARG(extend_seq_lens && extend_start_loc && batch_size >= 0 && num_tokens >= 0);
  SP_CHECK_ARG(num_q_heads > 0 && num_kv_heads > 0 && num_q_heads % num_kv_heads == 0);
  SP_CHECK_ARG(max_extend_len >= 0 && max_seq_len >= 0 && k_scale > 0.f && v_scale > 0.f);
  sm_scale *= k_scale;   // scaled pools (set_kv_buffer, memory/pool.py:401-412): see sp_decode_attention
  SP_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k_buffer & 15) == 0 &&
               ((uintptr_t)v_buffer & 15) == 0);
  if (dtype != SP_F32 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  const bool kv8 = kv_dtype == SP_FP8_E5M2;
  if (!kv8 && kv_dtype != dtype) return SP_ERR_UNSUPPORTED;
  if (kv8 && dtype == SP_F32) return SP_ERR_UNSUPPORTED;
  if (batch_size == 0 || num_tokens == 0) return SP_OK;
  if (head_dim != 64 && head_dim != 128) return SP_ERR_UNSUPPORTED;
  const int G = num_q_heads / num_kv_heads;
  // fp32 (row streams on the VALU decode kernel): 1/2/4/8 query heads per KV head; 16-bit: any width
  if (dtype == SP_F32 && G != 1 && G != 2 && G != 4 && G != 8) return SP_ERR_UNSUPPORTED;
  const int vec = dtype == SP_F32 ? 4 : 8;
  SP_CHECK_ARG(q_stride % vec == 0 && kv_buffer_stride % vec == 0);
  if (num_tokens > 0x7fffffffLL / 64) return SP_ERR_INVALID_ARG;
  const size_t need =
      sp_extend_attention_workspace_bytes(num_tokens, batch_size, num_q_heads, head_dim, dtype);
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return SP_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // a plan must be large enough for every grid row this launch reads an item for (its CONTENT is checked by
  // the kernel against the launch: block size, head counts, num_tokens, batch size)
  if (plan && plan_bytes < sp_extend_plan_bytes(num_tokens, batch_size, num_q_heads, num_kv_heads))
    return SP_ERR_WORKSPACE;

  // 16-bit dtypes run on the matrix cores (extend_mfma.hip); fp32 and unsupported shapes take the
  // row-stream path below
  {
    const int rc = run_extend_mfma(out, q, k_buffer, v_buffer, req_to_token, req_to_token_stride,
                                   req_pool_indices, seq_lens, kv_start, idx64, extend_seq_lens,
                                   extend_start_loc, batch_size, num_q_heads, num_kv_heads, head_dim,
                                   q_stride, out_stride, kv_buffer_stride, sm_scale, logit_cap, v_scale,
                                   causal, window_left, max_extend_len, max_seq_len, plan,
                                   (int)std::min<int64_t>(extend_plan_items(num_tokens, batch_size,
                                       extend_block_rows(num_q_heads, num_kv_heads)), 0x7fffffff),
                                   (int)num_tokens, dtype, kv8 ? 1 : 0, st);
    if (rc != SP_ERR_UNSUPPORTED || kv8) return rc;   // an fp8 pool has no row-stream path
  }

  int32_t* row_req = (int32_t*)workspace;
  int32_t* row_len = row_req + num_tokens;
  int32_t* row_kv0 = row_len + num_tokens;
  expand_rows_kernel<<<dim3(batch_size), 256, 0, st>>>(row_req, row_len, row_kv0, req_pool_indices,
                                                       seq_lens, kv_start, idx64, extend_seq_lens,
                                                       extend_start_loc, causal, window_left, num_tokens);
  SP_LAUNCH_CHECK();

  DecodeArgs a;
  a.out = out; a.q = q; a.kbuf = (const char*)k_buffer; a.vbuf = (const char*)v_buffer;
  a.r2t = req_to_token; a.r2t_stride = req_to_token_stride;
  a.req_idx = row_req; a.seq_lens = row_len; a.kv_start = row_kv0; a.idx64 = 0;
  a.bs = (int)num_tokens; a.Hq = num_q_heads; a.Hkv = num_kv_heads;
  a.q_stride = q_stride; a.o_stride = out_stride; a.kv_stride = kv_buffer_stride;
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.out_scale = v_scale;
  // one split per row: every row-stream is reduced inside one workgroup, no partials
  int64_t chunk = ((max_seq_len > 0 ? max_seq_len : 1) + 3) / 4 * 4;
  if (chunk > 0x7ffffff0LL) return SP_ERR_INVALID_ARG;
  a.chunk = (int)chunk; a.num_splits = 1; a.max_len = (int)chunk; a.max_slots = (int)num_tokens;
  a.hh_shift = decode_heads_per_load_shift(num_kv_heads, head_dim, dtype, &a.head_groups);
  a.part_o = nullptr; a.part_lse = nullptr; a.plan = nullptr; a.kv8 = 0; a.nt_min_keys = 0x7fffffff; a.rplan = nullptr; a.ranges = 0;
  g_extend_last_kernel = 4;
  return run_decode(a, head_dim, G, dtype, st);
}
