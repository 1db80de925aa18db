// Entry points of extend_mfma.hip used by the C-ABI file (extend_attention.hip).  Internal to the library.
// (Kept out of attention_internal.h so that the decode kernels' source hash - bench.py,
// profiles/*_decode_attn_pmc.json - does not move when only the extend kernel changes.)
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

namespace sp {

// words of an extend plan's header: [count, block rows, Hq, Hkv, num_tokens, bs, 0, 0]; the items follow
constexpr int kExtPlanHeader = 8;
// behind a plan's items (at word kExtPlanHeader + 2 * the item bound of sp_extend_plan_bytes): ticket and completion
// counters of the persistent extend kernel, one pair per kv-head group; zero whenever no launch is using the plan
constexpr int kExtPlanTicketGroups = 256;
constexpr int kExtPlanTicketWords = 2 * kExtPlanTicketGroups;

// extend_mfma.hip: MFMA tile kernel for 16-bit ragged extend; SP_ERR_UNSUPPORTED -> use row-streams
int run_extend_mfma(void* out, const void* q, const void* k_buffer, const void* v_buffer,
                    const int32_t* req_to_token, int64_t req_to_token_stride,
                    const void* req_pool_indices, const void* seq_lens, const void* kv_start,
                    int idx64, const int32_t* extend_seq_lens, const int32_t* extend_start_loc,
                    int batch_size, int num_q_heads, int num_kv_heads, int head_dim, int64_t q_stride,
                    int64_t out_stride, int64_t kv_buffer_stride, float sm_scale, float logit_cap,
                    float out_scale, int causal, int window_left, int max_extend_len, int64_t max_seq_len,
                    const int32_t* plan, int plan_items, int num_tokens, int dtype, int kv8, hipStream_t st);

// extend_w64.hip: the 4-wave x 64-row form of the same kernel (16-bit, D = 128, plain attention, query-head group a
// multiple of 4); returns 1 (one workgroup per item) or 2 (persistent workgroups) when it took the launch, 0 otherwise.
// set_extend_w64: 0 never, 1 where it pays (default), 2 always.
struct ExtendArgs;
int try_extend_w64(const ExtendArgs& a, int head_dim, int dtype, int max_extend_len, int64_t max_seq_len, hipStream_t st);
void set_extend_w64(int v);
void set_extend_w64_persist(int v);   // 0: never, 1 (default): where it pays, 2: every launch with a plan the 4 x 64-row kernel applies to

// which kernel the last sp_extend_attention call launched (sp_debug_get("extend_last_kernel")): 0 none yet, 1 the
// 8-wave matrix-core kernel, 2 the 4-wave x 64-row kernel (one workgroup per item), 3 its persistent form, 4 row streams
extern int g_extend_last_kernel;
int w64_descriptor_patched();

// test / tuning hooks behind sp_debug_set
void set_extend_defer_x10(int tenths);
void set_extend_dma(int v);   // 0: register-staged K/V tiles for every shape (the LDS-DMA ring is the default where it applies)

}  // namespace sp
