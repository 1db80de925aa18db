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
  // optional, from sp_decode_plan: [count, chunk, needed, keys | slot0[bs] | (b, c) x max_slots].
  // The CHUNK is part of the plan (device memory), so a captured launch follows whatever split size the step's
  // plan was built with; slot0[b] = first partial slot of request b (exclusive scan of its split count);
  // needed = the item count BEFORE the cut at max_slots (needed > count: the host's bound on sum(seq_lens) was
  // broken and items were dropped - the host checks this word, see sp_decode_plan); keys = sum of the (clamped)
  // lengths, i.e. the key rows this step's launches gather per kv head.  Read-only for every launch that uses it.
  const int32_t* plan;
  // K/V gathers of the matrix-core kernel are NON-TEMPORAL loads when the step reads at least this many keys in total
  // (plan[3], written by sp_decode_plan): a stream that is read once and is larger than the caches then no longer
  // displaces everything else in them, at the price of about a microsecond of latency per dependent round of gathers,
  // which only a launch of several rounds of workgroups hides.  0 = always (the default), INT_MAX = never; a plan-less
  // launch has no key count and streams only when the threshold is 0.
  int nt_min_keys;
  // RANGE geometry (the plan's second section, sp_decode_plan with ranges > 0; decode_mfma.hip's range kernel):
  // [rcount, R, ranges, bs | pos[bs + 1] | start[ranges]].  Words 2 and 3 say what the section was BUILT for (ABI 9): a
  // launch given another piece count or batch size finds the mismatch and does nothing (range_plan_matches).  The step's keys, request after request in batch order, form one
  // line on which request b takes pos[b] .. pos[b] + len_b and then kRangeReqCost empty positions (what a request costs
  // a wave beyond its keys); the line is cut into rcount <= ranges pieces of R positions, piece j is the work of
  // one wave per kv head, start[j] = the first request with a key at or after position j * R, or -1 if piece j
  // holds none.  A request whose keys lie in pieces jf .. jl > jf leaves jl - jf + 1 partials in slots b + jf .. b + jl
  // (b + j grows along the line, so no two (request, piece) pairs share a slot and bs + ranges slots always suffice),
  // a request inside one piece is written straight to the output.  Null: the launch uses the (request, split) items.
  const int32_t* rplan;
  int ranges;    // pieces the range section was sized for = waves per kv head of a range launch
};

static constexpr int kPlanHdr = 4;   // int32 words in front of slot0[]
static constexpr int kRangeHdr = 4;  // int32 words in front of pos[]
static constexpr int kRangeReqCost = 16;   // positions a request takes on the line beyond its keys
static constexpr int kRangeMin = 64;       // shortest piece (keys): one index register, four tiles

// words of a plan's (request, split) section; the range section follows it.  max_slots = 0 (ABI 9): a plan built for
// range launches only - the header alone, no slot0[] and no items
__host__ __device__ inline int64_t plan_item_words(int bs, int64_t max_slots) {
  return max_slots > 0 ? kPlanHdr + (int64_t)bs + 2 * max_slots : kPlanHdr;
}

// the range section was built for this launch's piece count and batch size (its words 2, 3): everything the range and
// merge kernels index with `ranges` and `bs` - pos[bs + 1], start[ranges], slot b + j - is only then what they assume
__device__ __forceinline__ bool range_plan_matches(const int32_t* rplan, int ranges, int bs) {
  return rplan[2] == ranges && rplan[3] == bs;
}

// request b of a range launch: its partial count and first slot (n <= 1: written straight to the output)
__device__ __forceinline__ void range_request(const int32_t* rplan, int b, int& len, int& nsplit, int& slot0) {
  const int R = rplan[1];
  const int p0 = rplan[kRangeHdr + b], w = rplan[kRangeHdr + b + 1] - p0;
  len = w > 0 ? w - kRangeReqCost : 0;
  const int jf = p0 / R, jl = len > 0 ? (p0 + len - 1) / R : jf;
  nsplit = jl - jf + 1;
  slot0 = b + jf;
}

// (request, split) of work item `item`, the split size, and the request's first partial slot
__device__ __forceinline__ bool decode_item(const DecodeArgs& a, int item, int& b, int& c, int& chunk, int& slot0) {
  if (a.plan) {
    if (item >= a.plan[0]) return false;
    chunk = a.plan[1];
    const int32_t* items = a.plan + kPlanHdr + a.bs;
    b = items[2 * item];
    c = items[2 * item + 1];
    slot0 = a.plan[kPlanHdr + b];
  } else {
    chunk = a.chunk;
    c = item % a.num_splits;
    b = item / a.num_splits;
    slot0 = b * a.num_splits;
  }
  return true;
}

// Combine the split partials of U (request, q head) pairs by their log2-sum-exp: one wave, lane = output elements
// lane, lane + 64 (D = 128).  The partials of up to 16 splits (log-sum-exp and the lane's output elements) are
// loaded in one go - a single memory round trip instead of a max pass followed by a dependent accumulate pass -
// and merged online across groups of 16 (every product-sum is an explicit fma).
// Partials are laid out [Hq][slot][D]: a request's splits are consecutive slots, so one (request, head)'s partials
// are one contiguous run (with [slot][Hq][D] the merge read 512-byte pieces 16 KiB apart: 10.5 us instead of 6.8).
// GRP = partials loaded per round trip: 16, or 4 for the requests of at most four partials (most of a range launch's: a
// piece cuts a request once or twice) - the same arithmetic on the live partials in the same order, a quarter of the loads.
template <typename Tag, int D, int U, int GRP = 16>
__device__ __forceinline__ void decode_merge_rows(const DecodeArgs& a, int b, const int (&h)[U], int lane,
                                                  int nsplit, int slot0) {
  typedef Elem<Tag> E;
  constexpr int PER = D / 64;
  constexpr float kFloor = -1.0e30f;
  const float* lse[U];
  const float* po[U];
  float o[U][PER], W[U], M[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    lse[u] = a.part_lse + (int64_t)h[u] * a.max_slots + slot0;
    po[u] = a.part_o + ((int64_t)h[u] * a.max_slots + slot0) * D;
    W[u] = 0.f;
    M[u] = kFloor;
#pragma unroll
    for (int e = 0; e < PER; ++e) o[u][e] = 0.f;
  }
  for (int c0 = 0; c0 < nsplit; c0 += GRP) {
    float ls[U][GRP], pv[U][GRP][PER];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < GRP; ++c) {
        const bool live = c0 + c < nsplit;
        const int cc = live ? c0 + c : c0;               // clamped: the load stays in bounds
        ls[u][c] = live ? lse[u][cc] : kFloor;
#pragma unroll
        for (int e = 0; e < PER; ++e) pv[u][c][e] = po[u][(int64_t)cc * D + e * 64 + lane];
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float Mg = M[u];
#pragma unroll
      for (int c = 0; c < GRP; ++c) Mg = fmaxf(Mg, ls[u][c]);
      const float rescale = __builtin_amdgcn_exp2f(M[u] - Mg);   // 0 on the first group (M = -1e30)
      W[u] *= rescale;
#pragma unroll
      for (int e = 0; e < PER; ++e) o[u][e] *= rescale;
#pragma unroll
      for (int c = 0; c < GRP; ++c) {
        const float w = c0 + c < nsplit ? __builtin_amdgcn_exp2f(ls[u][c] - Mg) : 0.f;
        W[u] += w;
#pragma unroll
        for (int e = 0; e < PER; ++e) o[u][e] = __builtin_fmaf(w, pv[u][c][e], o[u][e]);
      }
      M[u] = Mg;
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int e = 0; e < PER; ++e)
      E::store(a.out, (int64_t)b * a.o_stride + (int64_t)h[u] * D + e * 64 + lane, o[u][e] / W[u] * a.out_scale);
}

// decode_attention.hip: launch the split-KV decode kernel (+ merge when num_splits > 1)
int run_decode(const DecodeArgs& a, int head_dim, int group, int dtype, hipStream_t st);
// decode_attention.hip: merge of the split partials (shared by the VALU and MFMA decode kernels)
int run_decode_merge(const DecodeArgs& a, int head_dim, int dtype, hipStream_t st);
// decode_mfma.hip: matrix-core decode kernel for 16-bit dtypes, G <= 16 (attention kernel only)
int run_decode_mfma(const DecodeArgs& a, int head_dim, int dtype, hipStream_t st);
// heads per wave-load: the largest power of two <= rows-per-load that divides Hkv
int decode_heads_per_load_shift(int num_kv_heads, int head_dim, int dtype, int* head_groups);

// test / tuning hooks behind sp_debug_set / sp_debug_get
// which kernel the last sp_decode_attention call launched: 0 none yet, 1 the VALU kernel, 2 the matrix-core kernel on
// (request, split) items, 3 the range kernel
extern int g_decode_last_kernel;
void set_decode_kernel(int which);
void set_decode_nt_min_mb(int mb);
void set_decode_ranges(int n);
// decode_mfma.hip: pieces the range kernel wants for this shape (0: it does not take the shape)
int decode_mfma_ranges(int num_q_heads, int num_kv_heads, int head_dim, int dtype, int kv8);

}  // namespace sp
