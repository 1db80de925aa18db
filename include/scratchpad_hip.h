/*
 * scratchpad_hip.h - C ABI of the MI355X (gfx950) hot-path library `libscratchpad_hip.so`.
 *
 * This is the drop-in boundary beneath Scratchpad's four Python seams.
 * The reference exports no C ABI itself - it calls third-party wheels (flashinfer v0.2.3,
 * triteia / triteia_cuda) - so every entry point below cites the reference call site it
 * replaces (paths relative to /root/reference/scratchpad).
 *
 * Conventions
 *  - plain pointers and sizes only; every buffer (inputs, outputs, workspaces, the KV pool) is
 *    allocated and owned by the caller (torch on the Python side); the library allocates nothing
 *    and keeps no state;
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and returns without
 *    synchronising; all are HIP-graph capturable (no host sync, no malloc, launch geometry
 *    depends only on host-visible arguments, never on device-side seq_lens values);
 *  - return 0 on success, a negative sp_status otherwise; nothing throws;
 *  - `dtype`: SP_F32 / SP_F16 / SP_BF16 - the activation & KV-pool element type;
 *  - strides are in ELEMENTS; innermost dimensions are contiguous;
 *  - `idx64`: 1 if req_pool_indices / seq_lens are int64 (eager mode), 0 if int32 (graph mode),
 *    as in ForwardBatch (model_executor/forward_info.py:84-104, cuda_graph_runner.py:193-199);
 *  - KV slot 0 is the reserved dummy slot (memory/pool.py:126-127, 249-253): padded rows carry
 *    out_cache_loc == 0 and seq_len == fill value and are harmless.
 */
#ifndef SCRATCHPAD_HIP_H
#define SCRATCHPAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_ABI_VERSION 9
#define SP_API __attribute__((visibility("default")))

typedef enum { SP_F32 = 0, SP_F16 = 1, SP_BF16 = 2, SP_FP8_E5M2 = 3 /* KV pool only */ } sp_dtype;

typedef enum {
  SP_OK = 0,
  SP_ERR_INVALID_ARG = -1, /* null pointer, negative size, misaligned buffer */
  SP_ERR_UNSUPPORTED = -2, /* head_dim / group size / dtype combination not built */
  SP_ERR_WORKSPACE = -3,   /* workspace too small */
  SP_ERR_LAUNCH = -4       /* hipLaunchKernel failed (see hipGetLastError) */
} sp_status;

SP_API int sp_abi_version(void);
SP_API const char* sp_status_string(int status);
/* Test / tuning hook (no reference counterpart): process-wide kernel-selection switches for A/B
 * measurements and parity tests of the non-default kernels.  Keys: "decode_kernel" (0 = default,
 * 1 = VALU kernel, 2 = matrix-core kernel), "extend_defer_x10" (how far the extend kernel's running
 * row maximum may trail, in tenths of a log2 unit; < 0 = the shipped value), "extend_dma" (1 = K/V tiles
 * by LDS-DMA into the swizzled ring where it applies - D 128, 16-bit pools - the default; 0 = register-
 * staged tiles for every shape: same bits), "extend_w64" (the 4-wave x 64-row form of the extend kernel -
 * D 128, 16-bit pools, plain attention, query-head group a multiple of 4: 1 = where it pays, the default;
 * 2 = wherever it applies; 0 = never), "extend_w64_persist" (its persistent form - one workgroup per compute
 * unit drawing the plan's items by ticket: 1 = where it pays, i.e. a plan with at least 32 items per
 * workgroup and row blocks of at most ~32 tiles, the default; 2 = every planned launch the kernel applies to;
 * 0 = never; without it mode 1 of "extend_w64" means mean extend length >= 768 or a cached prefix >= 1024
 * tokens; the two forms agree bit for bit), "decode_ranges" (0 = decode launches never take the range geometry of
 * their plan: the (request, split) items everywhere; -1 = the default, wherever the range kernel applies),
 * "decode_nt_min_mb" (see sp_decode_attention), "ar_fused_blocks", "skinny_nt".  Nothing on the call path reads the environment.  Returns
 * SP_ERR_INVALID_ARG for an unknown key.                                                          */
SP_API int sp_debug_set(const char* key, int value);
/* Read-only counterpart (ABI 6): "w64_descriptor_patched" (1 = extend_w64.hip was built in stages with its kernels'
 * register allocation fixed at assembly level - scratchpad_amd/build.py compile_w64 - and the
 * 4-wave x 64-row extend kernels may launch; anything else: sp_extend_attention keeps to the 8-wave kernel),
 * "extend_last_kernel" (what the last sp_extend_attention call launched: 1 = 8-wave matrix-core kernel, 2 = 4-wave
 * x 64-row kernel, 3 = its persistent form, 4 = row streams on the decode kernel, 0 = none yet), "decode_last_kernel"
 * (ABI 9; what the last sp_decode_attention call launched: 1 = VALU kernel, 2 = matrix-core kernel on (request, split)
 * items, 3 = range kernel, 0 = none yet).  -1: unknown key. */
SP_API int sp_debug_get(const char* key);

/* ---- RMSNorm: replaces flashinfer.norm.rmsnorm / fused_add_rmsnorm
 *      (nn/layers/layernorm.py:22-32; semantics of forward_native 34-51).
 * sp_rmsnorm:           out[t,:] = round(x[t,:] * rsqrt(mean(x^2)+eps)) * weight
 * sp_fused_add_rmsnorm: residual[t,:] = round(x + residual) (in place), then
 *                       x[t,:] = round(fp32(x+residual) * rsqrt(..)) * weight   (in place)       */
SP_API int sp_rmsnorm(void* out, const void* x, const void* weight, int64_t num_tokens, int hidden,
               int64_t x_stride, int64_t out_stride, float eps, int dtype, void* stream);
SP_API int sp_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t num_tokens,
                         int hidden, int64_t x_stride, int64_t res_stride, float eps, int dtype,
                         void* stream);

/* ---- SiLU-and-mul: replaces flashinfer.activation.silu_and_mul (nn/layers/activation.py:26-31).
 * out[t, j] = silu(x[t, j]) * x[t, d + j],  x: [T, 2d]                                           */
SP_API int sp_silu_and_mul(void* out, const void* x, int64_t num_tokens, int d, int64_t x_stride,
                    int64_t out_stride, int dtype, void* stream);

/* ---- Rotary embedding, in place on q and k: replaces triteia_cuda.rotary_embedding
 *      (nn/layers/rotary_embedding.py:132-164; semantics of forward_native 102-130).
 * cos_sin_cache: [max_position, rotary_dim] = cos || sin, in `dtype`; positions int64 [T].
 * q: [T, Hq*head_size], k: [T, Hkv*head_size].  is_neox: 1 = halves, 0 = GPT-J interleaved.
 * If k_buffer/v_buffer are non-null the rotated k rows and the v rows are ALSO scattered to
 * the KV pool at out_cache_loc (fused KV store, = sp_kv_store below); pass NULL to skip.        */
SP_API int sp_rotary_embedding(const int64_t* positions, void* q, void* k, const void* cos_sin_cache,
                        int64_t num_tokens, int num_q_heads, int num_kv_heads, int head_size,
                        int rotary_dim, int64_t q_stride, int64_t k_stride, int is_neox,
                        const void* v, int64_t v_stride, void* k_buffer, void* v_buffer,
                        const int64_t* out_cache_loc, int64_t kv_buffer_stride, int dtype,
                        void* stream);

/* ---- KV store: replaces `k_buffer[layer][loc] = cache_k` (MHATokenToKVPool.set_kv_buffer,
 *      memory/pool.py:392-424).  k,v: [T, Hkv, D] (token stride given); buffers [P+1, Hkv, D].  */
SP_API int sp_kv_store(void* k_buffer, void* v_buffer, const int64_t* loc, const void* k, const void* v,
                int64_t num_tokens, int num_kv_heads, int head_dim, int v_head_dim,
                int64_t k_stride, int64_t v_stride, int64_t k_buffer_stride,
                int64_t v_buffer_stride, int dtype, void* stream);

/* ---- KV store into an fp8 (e5m2) pool: the `--kv-cache-dtype fp8_e5m2` branch of set_kv_buffer
 *      (memory/pool.py:401-412: cache_k.div_(k_scale); cache_k.to(float8_e5m2); stored as uint8).
 * k,v: [T, Hkv, D] in `src_dtype` (fp16/bf16/fp32); buffers: uint8 [P+1, Hkv, D], strides in
 * elements (= bytes).  The reference's two steps with their two roundings: x / scale as an fp32
 * division rounded to `src_dtype` (cache_k.div_(k_scale)), then round-to-nearest-even to 1-5-2
 * (overflow -> inf), i.e. torch's .to(torch.float8_e5m2).  Pass 1.0 for an unscaled pool.        */
SP_API int sp_kv_store_fp8(void* k_buffer, void* v_buffer, const int64_t* loc, const void* k, const void* v,
                    int64_t num_tokens, int num_kv_heads, int head_dim, int64_t k_stride,
                    int64_t v_stride, int64_t k_buffer_stride, int64_t v_buffer_stride, float k_scale,
                    float v_scale, int src_dtype, void* stream);

/* ---- req_to_token scatter: replaces write_req_to_token_pool_triton
 *      (scheduler/schedule_batch.py:1546-1580).  All index arrays int64, table int32.           */
SP_API int sp_write_req_to_token(int32_t* req_to_token, int64_t row_stride,
                          const int64_t* req_pool_indices, const int64_t* pre_lens,
                          const int64_t* seq_lens, const int64_t* extend_lens,
                          const int64_t* out_cache_loc, int batch_size, void* stream);

/* ---- positions: replaces compute_position_triton (model_executor/forward_info.py:400-449)
 *      and clamp_position (469-471).                                                             */
SP_API int sp_compute_position(int64_t* positions, int32_t* extend_start_loc,
                        const int32_t* extend_prefix_lens, const int32_t* extend_seq_lens,
                        int batch_size, void* stream);
SP_API int sp_clamp_position(int64_t* positions, const void* seq_lens, int idx64, int batch_size,
                      void* stream);

/* ---- Paged decode attention: replaces decode_attention_fwd (nn/attention/triton_attn/
 *      decode_attention.py:547-608; call site triton_backend.py:183-195) and flashinfer
 *      BatchDecodeWithPagedKVCacheWrapper.forward (flashinfer_backend.py:475-482).
 * For request b and q head h:  o = softmax(q.K[idx]^T * sm_scale [soft-capped]) . V[idx],
 *   idx = req_to_token[req_pool_indices[b], kv_start[b] : kv_start[b] + seq_lens[b]]
 * q,o: [bs, Hq, D]; buffers [P+1, Hkv, D] with token stride kv_buffer_stride; page_size = 1.
 * Hq / Hkv (query heads per KV head): any value 1..16 for fp16/bf16, 1/2/4/8 for fp32; D 64 or 128.
 * A row with seq_lens[b] == 0 (no visible key) is left untouched: pre-fill `out` where that can
 * happen (cross-attention of text-only requests).
 * kv_start may be NULL (= 0); it is the encoder offset of encoder-decoder models
 * (flashinfer_backend.py:593-621).  `max_seq_len`: an upper bound on every seq_lens[b] (the context
 * length under graph capture); a device-side seq_lens[b] above it is clamped to it, never followed.
 *
 * Split-KV geometry (ABI 4).  A request is cut into splits of `chunk` keys; split c of request b writes
 * its partial (o, log-sum-exp) to SLOT slot0[b] + c of the workspace ([Hq, max_slots, D] + [Hq, max_slots]
 * floats) and one merge wave per (request, head) combines them in a second launch.  (ABI 6 could also merge inside
 * the attention kernel through arrival counters in the plan; measured slower wherever graphs replay, removed in
 * ABI 7: a plan is read-only for the launches that use it, so launches on several streams may share one.)
 *   - Without a plan the grid is the static (request, split) rectangle: slot0[b] = b * num_splits,
 *     num_splits = ceil(max_seq_len / chunk), max_slots is ignored (= batch_size * num_splits).
 *   - With a `plan` (sp_decode_plan: [count, chunk, needed, keys | slot0[bs] | (b, c) x max_slots], `plan_bytes` long, built once per step from
 *     the same seq_lens and shared by all layers - the counterpart of flashinfer's begin_forward()/plan,
 *     flashinfer_backend.py:623-670, and of TritonAttnBackend.init_forward_metadata,
 *     triton_backend.py:48-68) the launch covers `max_slots` work items, the kernels read the split size
 *     FROM THE PLAN (device memory) and slot0[b] is the exclusive scan of the requests' split counts.  So the
 *     launch geometry - what a HIP graph captures - is a function of (batch_size, heads, max_slots) only:
 *     one captured launch follows whatever split size each step's plan was built with, and the workspace is
 *     bounded by sum(seq_lens) / chunk + batch_size slots (sp_decode_plan_slots), not by the context
 *     length (the reference's static attn_logits buffer is [bs, heads, max_context_len],
 *     triton_backend.py:70-80).  `chunk` is then the SMALLEST split size the plan may carry (it only decides
 *     whether the merge launch is needed).  Items are listed longest first (XCD load balance on ragged
 *     batches); a plan never changes the result, only which workgroup computes which split.
 * workspace: sp_decode_attention_workspace_bytes(max_slots, ...); plan: sp_decode_plan_bytes().
 *
 * Range geometry (ABI 8; what a default decode step runs).  A plan built with `ranges` > 0 carries a second section,
 * [rcount, R, ranges, batch_size | pos[bs + 1] | start[ranges]], behind the items: the step's keys, request after request
 * in batch order, form one line (request b at pos[b] .. pos[b] + len_b, then 16 empty positions - what a request costs a
 * workgroup beyond its keys); the line is cut into rcount <= ranges equal pieces of R = ceil(length / ranges) positions
 * (at least 64) and start[j] is the first request with a key at or after j * R (-1: piece j holds none).  A launch
 * given the same `ranges` runs one WAVE per (piece, kv head) - with the kv heads in fours a workgroup is the four
 * heads of one piece: each walks its piece - the tail of one request, whole requests, the head of another - so every
 * wave gathers the same number of keys whatever the lengths are, there is no split size to choose, and a request is
 * written straight to the output unless a cut falls inside it.  A request whose keys lie in pieces jf .. jl > jf leaves
 * its partials in slots b + jf .. b + jl (b + j grows along the line: no two (request, piece) pairs share a slot) and
 * the merge launch combines them: the workspace then holds batch_size + ranges slots, whatever sum(seq_lens) is - the
 * overflow below cannot happen on this path.
 * sp_decode_ranges() is the piece count the library wants for a shape ON THE CURRENT DEVICE (the answer, and the
 * kernel's LDS limit behind it, are kept per device of the process: set the device before the first call, as for any
 * launch): two workgroups per CU (three on a byte pool), all resident at once, four waves each, over the kv heads (at
 * most 1024) - 256 pieces for 8 kv heads on MI355X, 1024 for a tensor-parallel rank's single head - or 0 where the range
 * kernel does not apply (fp32, groups wider than 16, head sizes other than 64 / 128).  Launches it does not take (those
 * shapes, a logit soft-cap, out rows not 8-byte aligned, sp_debug_set("decode_ranges", 0)) use the plan's (request, split)
 * items.  Requires batch_size * (max_seq_len + 16) < 2^31 (else SP_ERR_INVALID_ARG from sp_decode_plan; pass ranges = 0).
 * Results of the two geometries differ in the last bits (another split of the same sums), and under the range geometry
 * the cuts - hence a request's last bits - depend on the lengths of the whole batch, not on the request alone.
 *
 * Plan and launch must agree (ABI 9).  The sections of a plan are located from (batch_size, max_slots, ranges), so
 * sp_decode_attention must be given the values the plan was BUILT with - where the reference rebuilds indices and launch
 * from one begin_forward() call (flashinfer_backend.py:623-670) and they cannot disagree.  What is checked:
 *   - `plan_bytes` (sp_decode_attention) must cover sp_decode_plan_bytes(batch_size, max_slots, ranges), else
 *     SP_ERR_WORKSPACE: no kernel reads a header outside the caller's buffer;
 *   - the range section records the `ranges` and `batch_size` it was built for (its words 2, 3); a range launch (and its
 *     merge) that finds other values does NOTHING - no table read, no gather, `out` untouched - instead of following
 *     pos[] / start[] of another shape.  Nothing on the device reports it: pre-fill `out` to see it, or keep the pair on
 *     the host as scratchpad_amd/_native.py does (decode_plan() remembers what each plan buffer was built with and
 *     decode_attention() raises before launching on any difference);
 *   - ITEMS ARE OPTIONAL: max_slots = 0 (with ranges > 0) builds the range section alone - the item section is then its
 *     4-word header [0, chunk, 0, 0] - which is what a backend whose layers all take the range kernel wants (the plan
 *     kernel skips its six item passes).  A launch that would need the items of such a plan (soft-cap, fp32, ...) returns
 *     SP_ERR_INVALID_ARG.  Likewise ranges = 0 builds the items alone.  sp_debug_get("decode_last_kernel") says what the
 *     last call launched: 1 = VALU kernel (items), 2 = matrix-core kernel (items), 3 = range kernel.
 *
 * Overflow (ABI 6).  The plan's word 2 holds the number of items the lengths NEED; when it exceeds `max_slots`
 * (the caller's bound on sum(seq_lens) was too small) the surplus splits are not computed and the affected
 * output rows are wrong or unwritten.  Nothing on the device reports this by itself: the caller reads plan[2]
 * (a 16-byte copy of the header at a point where it synchronises anyway) and raises - HipAttnBackend does.
 *
 * Streaming gathers.  The plan's word 3 holds the keys the step gathers per kv head (the sum of the clamped lengths).
 * When a launch's K + V bytes (that sum x 2 x num_kv_heads x head_dim x element size) reach a threshold
 * (sp_debug_set("decode_nt_min_mb", n); 0 = always, -1 = never, -2 = the default) the matrix-core kernel gathers K/V rows with
 * NON-TEMPORAL loads: rows read once per step no longer displace the rest of the step's data from the caches.  It is plan
 * data, so a captured launch follows each step's own size; results are the same bits either way.  A plan-less launch
 * has no key count: it streams only when the threshold is 0 ("always", the default).
 *
 * `kv_dtype` = `dtype`, or SP_FP8_E5M2 for a uint8 pool written by sp_kv_store_fp8 (16-bit q
 * only; kv_buffer_stride then counts bytes): the kernels widen e5m2 to half exactly and compute
 * in fp16 with fp32 accumulation, as flashinfer does for fp8 KV (convert to the query type); the
 * in-tree Triton kernels' `p.to(v.dtype)` (probabilities rounded to e5m2) is NOT reproduced.
 *
 * `k_scale`, `v_scale` (> 0; 1.0 = none): the layer's KV scales, the same values the store divided
 * by (flashinfer_backend.py:470-482 passes layer.k_scale / layer.v_scale to both): the pool holds
 * k / k_scale and v / v_scale, so logits are multiplied by k_scale and the output by v_scale.    */
SP_API int64_t sp_decode_plan_slots(int batch_size, int64_t kv_tokens, int64_t max_seq_len, int chunk);
SP_API size_t sp_decode_attention_workspace_bytes(int64_t max_slots, int num_q_heads, int v_head_dim);
SP_API int sp_decode_ranges(int num_q_heads, int num_kv_heads, int head_dim, int dtype, int kv_dtype);
SP_API size_t sp_decode_plan_bytes(int batch_size, int64_t max_slots, int ranges);
SP_API int sp_decode_plan(int32_t* plan, size_t plan_bytes, const void* seq_lens, int idx64,
                   int batch_size, int64_t max_seq_len, int chunk, int64_t max_slots, int ranges, void* stream);
SP_API int sp_decode_attention(void* out, const void* q, const void* k_buffer, const void* v_buffer,
                        const int32_t* req_to_token, int64_t req_to_token_stride,
                        const void* req_pool_indices, const void* seq_lens, const void* kv_start,
                        int idx64, int batch_size, int num_q_heads, int num_kv_heads,
                        int head_dim, int64_t q_stride, int64_t out_stride,
                        int64_t kv_buffer_stride, float sm_scale, float logit_cap, float k_scale,
                        float v_scale, int64_t max_seq_len, int chunk, int64_t max_slots, int ranges,
                        void* workspace, size_t workspace_bytes, const int32_t* plan, size_t plan_bytes,
                        int dtype, int kv_dtype, void* stream);

/* ---- Ragged extend (prefill) attention: replaces extend_attention_fwd (nn/attention/
 *      triton_attn/extend_attention.py:229-327; call site triton_backend.py:137-154) and the
 *      flashinfer ragged+paged+merge_state / paged-only paths (flashinfer_backend.py:400-444).
 * New token t (0-based) of request b attends to kv positions [0, prefix_b + t] of
 * req_to_token[req_pool_indices[b], kv_start[b] + ...]; the new tokens' K/V must already be in
 * the pool (KV store precedes the kernel, triton_backend.py:131-134).  causal = 0 gives the
 * cross-attention form: every row attends to all seq_lens[b] kv positions
 * (flashinfer_backend.py:408-417).  window_left >= 0 (causal only) is the sliding window of
 * flashinfer_backend.py:413: the row at kv position p sees keys [p - window_left, p]; -1 = no
 * window.  (Decode needs no such argument: the caller passes kv_start = seq_len - min(seq_len,
 * window + 1) and that length, flashinfer_backend.py:559-577.)
 * q,o: [T, Hq, D]; extend_* are int32 [bs].  Hq / Hkv: any width for fp16/bf16, 1/2/4/8 for fp32.
 * num_tokens = sum(extend_seq_lens) (host-known: ForwardBatch.extend_num_tokens) and
 * max_extend_len >= max(extend_seq_lens), max_seq_len >= max(seq_lens) fix the launch geometry.
 * workspace: sp_extend_attention_workspace_bytes().                                              */
SP_API size_t sp_extend_attention_workspace_bytes(int64_t num_tokens, int batch_size, int num_q_heads,
                                           int head_dim, int dtype);
/* `plan` (optional, may be NULL): the step's (request, row block) work items, heaviest first, built once
 * per forward by sp_extend_plan() from the same extend_seq_lens / seq_lens / head counts / causal flag
 * and shared by every layer (the counterpart of flashinfer's begin_forward() for the prefill wrappers,
 * flashinfer_backend.py:400-444, 672-830).  It only decides which workgroup computes which rows (no
 * empty workgroups on ragged batches, longest rows first); results do not depend on it.  The plan carries
 * a header (block size, head counts, num_tokens, batch size): a launch of ANOTHER SHAPE - other head counts,
 * another token total or batch size - does not use its items and derives each workgroup's rows by walking the
 * requests instead.  That check identifies a launch shape, not a step: a plan left from an earlier step with
 * the same batch size and token total but other per-request lengths passes it and row blocks are then dropped
 * or computed twice.  REQUIREMENT: call sp_extend_plan for every step, with that step's extend_seq_lens, before
 * the step's first sp_extend_attention (HipAttnBackend.init_forward_metadata does).
 * sp_extend_plan_bytes() sizes the int32 buffer; `plan_bytes` (sp_extend_attention) must cover it, else
 * SP_ERR_WORKSPACE.  `causal` of sp_extend_plan is reserved (the item list does not depend on it).
 * The last 2 KiB of the buffer are counters of the attention kernel (the persistent form of the 4 x 64-row
 * kernel hands its workgroups the items by ticket): written by sp_extend_plan (zeros), counted up during
 * a launch that owns the plan and zero again when it ends - hence `plan` is not const, and one plan
 * buffer serves one launch at a time (the layers of a forward, one after the other on ONE stream: yes;
 * two streams at once: give each its own copy); a launch that was aborted leaves the counters dirty until
 * the next sp_extend_plan.  A plan that is not the launch's own is only read.                          */
SP_API size_t sp_extend_plan_bytes(int64_t num_tokens, int batch_size, int num_q_heads, int num_kv_heads);
SP_API int sp_extend_plan(int32_t* plan, size_t plan_bytes, const int32_t* extend_seq_lens, const void* seq_lens,
                   int idx64, int batch_size, int64_t num_tokens, int num_q_heads, int num_kv_heads,
                   int causal, void* stream);
SP_API int sp_extend_attention(void* out, const void* q, const void* k_buffer, const void* v_buffer,
                        const int32_t* req_to_token, int64_t req_to_token_stride,
                        const void* req_pool_indices, const void* seq_lens, const void* kv_start,
                        int idx64, const int32_t* extend_seq_lens,
                        const int32_t* extend_start_loc, int batch_size, int64_t num_tokens,
                        int num_q_heads, int num_kv_heads, int head_dim, int64_t q_stride,
                        int64_t out_stride, int64_t kv_buffer_stride, float sm_scale,
                        float logit_cap, float k_scale, float v_scale, int causal,
                        int window_left, int max_extend_len,
                        int64_t max_seq_len, void* workspace, size_t workspace_bytes,
                        int32_t* plan, size_t plan_bytes, int dtype, int kv_dtype, void* stream);

/* ---- Sampler.  Replaces nn/layers/sampler.py:63-75 (torch.argmax; logits.div_(T) + softmax),
 *      sampler.py:195-232 (top_k_top_p_min_p_sampling_from_probs_torch, top_p_normalize_probs_torch)
 *      and the flashinfer calls wrapped by nn/kernels/sampling.py:19-373 (top_k_renorm_probs,
 *      top_p_renorm_probs, top_k_top_p_sampling_from_probs "joint", min_p_sampling_from_probs).
 * Rows are [batch_size, vocab] with row_stride elements between rows; probabilities are fp32.
 * Filter (one definition for sampling and renormalising): rank tokens by (p descending, token id
 * ascending); a token is kept iff rank < top_k, the mass ranked above it is <= top_p, and
 * p >= p_max * min_p.  Mass is accumulated exactly as integers floor(p * 2^48), so the result
 * is independent of summation order.  NULL top_ks / top_ps / min_ps disable that filter.
 * sp_top_k_top_p_min_p_sample draws, per row, the token whose interval of the kept cumulative
 * mass (token-id order) contains floor(uniform * kept_mass); uniform[b] in [0,1) is supplied by
 * the caller (sampler.py:87-90 draws them with torch.rand).  keep_count (nullable) receives the
 * number of kept tokens per row.  A row with no mass yields token 0.
 * sp_argmax returns the first maximal index (torch.argmax).                                       */
SP_API int sp_argmax(const void* logits, int64_t row_stride, int batch_size, int vocab, int64_t* out_ids,
              int dtype, void* stream);
/* Vocab-parallel greedy (TP): instead of all-gathering [bs, vocab / tp] logits and taking the argmax
 * of the gathered row (nn/layers/logits_processor.py:362-369 followed by sampler.py:63-65), every
 * rank reduces its own shard to out_pairs[b] = {fp32 bits of the shard maximum, global index of its
 * first occurrence = local index + index_offset}; the caller all-gathers the [bs, 2] int32 words
 * rank-major and sp_argmax_merge returns, per row, the index of the largest value, ties to the
 * lowest global index - bit-identical to the gathered path.  `cols` = this rank's columns that are
 * real vocabulary (padding columns excluded; 0 = none: the shard never wins).                   */
SP_API int sp_argmax_shard(const void* logits, int64_t row_stride, int batch_size, int cols, int index_offset,
                    int32_t* out_pairs, int dtype, void* stream);
SP_API int sp_argmax_merge(const int32_t* pairs, int num_shards, int batch_size, int64_t* out_ids, void* stream);
SP_API int sp_softmax_temperature(float* logits_inout, int64_t row_stride, const float* temperatures,
                           int batch_size, int vocab, void* stream);
SP_API int sp_top_k_top_p_min_p_sample(const float* probs, int64_t row_stride, const int32_t* top_ks,
                                const float* top_ps, const float* min_ps, const float* uniform,
                                int batch_size, int vocab, int64_t* out_ids, int32_t* keep_count,
                                void* stream);
SP_API int sp_top_k_top_p_min_p_renorm(const float* probs, int64_t row_stride, const int32_t* top_ks,
                                const float* top_ps, const float* min_ps, int batch_size, int vocab,
                                float* out, int64_t out_stride, int32_t* keep_count, void* stream);

/* ---- Small-batch projection: out[M,N] = x[M,K] . w[N,K]^T, M <= 16, K % 32 == 0, fp16/bf16,
 *      fp32 accumulation, one rounding.  Serves F.linear inside QKVParallelLinear /
 *      MergedColumnParallelLinear / RowParallelLinear.forward (nn/layers/linear.py:423-470, 696-760,
 *      1033-1155) and the LM-head matmul (nn/layers/logits_processor.py:340-376) when the step has
 *      at most 16 tokens; a weight-streaming kernel (HBM-bound).  Returns SP_ERR_UNSUPPORTED for
 *      other shapes: the caller keeps the library GEMM.  Strides in elements.
 *      `epilogue` (ABI 6): 0 = out[M, N] as above.  1 = w is the merged gate|up matrix [2 N, K] of LlamaMLP
 *      (gate rows first; MergedColumnParallelLinear, linear.py:423-470) and out[M, N] = SiluAndMul of the projection
 *      (nn/layers/activation.py:21-31), bit for bit the plain call followed by sp_silu_and_mul: gate_up_proj +
 *      act_fn (nn/models/llama/llama.py:62-66) in one launch.                                      */
SP_API int sp_gemm_skinny(void* out, const void* x, const void* w, int M, int N, int K, int64_t x_stride,
                   int64_t w_stride, int64_t out_stride, int epilogue, int dtype, void* stream);

/* ---- Direct all-reduce through IPC-mapped peer regions: fills GroupCoordinator.ca_comm
 *      (distributed/parallel_state.py:266-267, 326-347; the reference never constructs one and always
 *      calls NCCL, linear.py:1148-1149).  Each rank sp_ar_alloc()s one fine-grained region of
 *      sp_ar_flag_bytes() + 2 * data_bytes, exports it (64-byte IPC handle), and imports every peer's.
 *      sp_custom_all_reduce sums `num_elems` elements over the `world` ranks (one-shot up to 256 KiB,
 *      reduce-scatter + all-gather above); `regions[r]` = rank r's region as mapped in this process;
 *      calls are collective (every rank issues the same sequence; the call counters live in the regions,
 *      so the launch is graph-capturable - the role of pynccl inside the reference's graphs,
 *      parallel_state.py:256-302).
 *      Failure is collective: a peer barrier that waits longer than `timeout_us` (<= 0: 30 s) raises the
 *      status word of EVERY rank's region and the caller's `host_status` word (device pointer of the pinned
 *      word from sp_ar_host_status_alloc, or NULL), which the host reads as plain memory at every forward
 *      boundary; a launch that finds its region's status raised raises its own host word and stops
 *      waiting.  sp_ar_status reads the region's status word through a synchronising copy (check points).
 *      Opt-in (see allreduce.hip: not yet run across xGMI).  sp_ar_alloc / sp_ar_ipc_import /
 *      sp_ar_host_status_alloc are the only entries that allocate; the caller's object owns the memory.  */
SP_API size_t sp_ar_flag_bytes(void);
SP_API int sp_ar_alloc(void** ptr, size_t bytes);
SP_API int sp_ar_free(void* ptr);
SP_API int sp_ar_ipc_export(void* ptr, void* handle64);
SP_API int sp_ar_ipc_import(const void* handle64, void** ptr);
SP_API int sp_ar_ipc_close(void* ptr);
SP_API int sp_ar_status(const void* own_region, int* status);
SP_API int sp_ar_host_status_alloc(void** host_ptr, void** device_ptr);
SP_API int sp_ar_host_status_free(void* host_ptr);
SP_API int sp_custom_all_reduce(void* out, const void* in, int64_t num_elems, int dtype, void* const* regions,
                         int rank, int world, size_t data_bytes, int64_t timeout_us, void* host_status,
                         void* stream);

/* ---- Fused all-reduce + residual add + RMSNorm on the same regions: RowParallelLinear's all-reduce
 *      (nn/layers/linear.py:1148-1149) followed by the next RMSNorm(x, residual) of the decoder layer
 *      (nn/models/llama/llama.py:216, 222, 268 -> nn/layers/layernorm.py:22-32), 2 x layers per step.
 *      x [T, hidden]: this rank's partial sums in, the normalised activations out; residual [T, hidden]
 *      in/out.  Bit for bit  x <- sp_custom_all_reduce(x); sp_fused_add_rmsnorm(x, residual, w, eps)
 *      (one rounding of the fp32 rank-order sum, then the norm kernel's rounding points and summation
 *      order).  One-shot up to 256 KiB, else reduce-scatter by rows + gather of the bf16 sums with the
 *      norm applied on arrival.  Collective, graph-capturable, same failure protocol as above.
 *      SP_ERR_UNSUPPORTED (caller runs the two-step form): hidden not a multiple of the 16-byte vector or
 *      above 8192 elements, unaligned pointers/strides.  Strides in elements.                           */
SP_API int sp_fused_allreduce_add_rmsnorm(void* x, void* residual, const void* weight, int64_t num_tokens,
                                   int hidden, int64_t x_stride, int64_t res_stride, float eps, int dtype,
                                   void* const* regions, int rank, int world, size_t data_bytes,
                                   int64_t timeout_us, void* host_status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCRATCHPAD_HIP_H */
