// Sampler kernels for gfx950: greedy argmax, temperature softmax, and top-k / top-p / min-p
// filtering + sampling without a sort.
//
// Replaces nn/layers/sampler.py:63-75 (argmax / div_ + softmax), 195-232 (the torch top-k/top-p/
// min-p sampler and top_p_normalize_probs_torch) and the flashinfer entry points wrapped by
// nn/kernels/sampling.py (top_k_renorm_probs 19-50, top_p_renorm_probs 64-97,
// top_k_top_p_sampling_from_probs 209-292, min_p_sampling_from_probs 319-373).
//
// The reference sorts every row (128k floats) and takes an fp32 cumulative sum.  Here one
// workgroup owns a row and finds the cut of the (p desc, id asc) ranking by a 3-level radix
// descent over the fp32 bit pattern (11+11+10 bits): per level an LDS histogram of count, mass
// and minimum per digit, a scan from the top digit down, and the first digit whose smallest
// member is dropped is refined further.  Mass is carried as integers, fx(p) = floor(p * 2^48),
// so LDS atomics commute and the result does not depend on arrival order: all TP ranks draw the
// same token from the same probabilities (what sampler.py:146-157 asks of the kernels).
// The row (513 KB at vocab 128256) stays in the XCD's L2 between passes; HBM sees it once.
//
// tests/test_gpu_sampling.py holds the kernels to the CPU statement of the same definition,
// exactly (ids, keep counts).
#include "sp_common.h"

namespace sp {

constexpr int kSampThreads = 1024;
constexpr int kSampWaves = kSampThreads / 64;
constexpr int kBins = 2048;
constexpr double kFxOne = 281474976710656.0;  // 2^48

__device__ __forceinline__ uint32_t prob_bits(float p) { return p > 0.f ? as_u32(p) : 0u; }  // NaN, -x -> 0
__device__ __forceinline__ uint64_t fx_of_bits(uint32_t b) { return (uint64_t)((double)as_f32(b) * kFxOne); }

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
  const uint32_t lo = __shfl_up((uint32_t)v, d, 64), hi = __shfl_up((uint32_t)(v >> 32), d, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int d) {
  const uint32_t lo = __shfl_xor((uint32_t)v, d, 64), hi = __shfl_xor((uint32_t)(v >> 32), d, 64);
  return ((uint64_t)hi << 32) | lo;
}

struct SelectArgs {
  const float* probs;
  int64_t row_stride;
  const int32_t* top_ks;  // nullable: no rank limit
  const float* top_ps;    // nullable: no mass limit
  const float* min_ps;    // nullable: no floor
  const float* uniform;   // SAMPLE mode
  int vocab;
  int64_t* out_ids;       // SAMPLE mode
  float* renorm;          // RENORM mode
  int64_t renorm_stride;
  int32_t* keep_count;    // nullable
};

struct SelectShared {
  uint32_t cnt[kBins];
  uint64_t mass[kBins];
  uint32_t mn[kBins];
  uint32_t wave_cnt[kSampWaves];
  uint64_t wave_mass[kSampWaves];
  int sel;
  uint32_t sel_cnt;
  uint32_t base_cnt;   // tokens ranked above the current digit range
  uint64_t base_mass;  // their mass
  uint64_t grand_mass;
  uint32_t red_u32[kSampWaves];
  uint32_t seg_ties[kSampWaves];
  uint64_t seg_mass[kSampWaves];
};

// The cut, as every thread sees it after select_cut(): a token with pattern b is kept iff
//   b > pivot, or b == pivot and it is among the first `ties_kept` such tokens in id order.
struct Cut {
  uint32_t pivot;
  uint32_t ties_kept;
  uint32_t above;   // number of tokens with b > pivot
  uint64_t total;   // kept mass
};

__device__ Cut select_cut(const SelectArgs& a, const float* __restrict__ row, SelectShared& sh, int b) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vocab = a.vocab;
  const int64_t top_k = a.top_ks ? (int64_t)max(a.top_ks[b], 0) : (int64_t)1 << 40;
  const uint64_t top_p_fx = a.top_ps ? (uint64_t)((double)a.top_ps[b] * kFxOne) : ~(uint64_t)0;

  // p_max (only needed for the min-p floor)
  float thr = 0.f;
  if (a.min_ps) {
    uint32_t mx = 0;
    for (int i = tid; i < vocab; i += kSampThreads) mx = max(mx, prob_bits(row[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    if (lane == 0) sh.red_u32[wave] = mx;
    __syncthreads();
    mx = 0;
#pragma unroll
    for (int w = 0; w < kSampWaves; ++w) mx = max(mx, sh.red_u32[w]);
    thr = as_f32(mx) * a.min_ps[b];
  }
  if (tid == 0) { sh.base_cnt = 0; sh.base_mass = 0; }

  uint32_t prefix = 0;   // digits fixed so far
  bool keep_all = false;
#pragma unroll 1
  for (int level = 0; level < 3; ++level) {
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const int width = level == 2 ? 10 : 11;
    const uint32_t mask = (1u << width) - 1;
    for (int i = tid; i < kBins; i += kSampThreads) { sh.cnt[i] = 0; sh.mass[i] = 0; sh.mn[i] = 0xffffffffu; }
    if (tid == 0) sh.sel = -1;
    __syncthreads();
    for (int i = tid; i < vocab; i += kSampThreads) {
      const uint32_t bits = prob_bits(row[i]);
      if (level == 0 || (bits >> (shift + width)) == prefix) {
        const uint32_t d = (bits >> shift) & mask;
        atomicAdd(&sh.cnt[d], 1u);
        atomicAdd((unsigned long long*)&sh.mass[d], (unsigned long long)fx_of_bits(bits));
        atomicMin(&sh.mn[d], bits);
      }
    }
    __syncthreads();
    // scan digits from the top: thread t owns digits (2047-2t, 2046-2t)
    const int d0 = kBins - 1 - 2 * tid, d1 = d0 - 1;
    const uint32_t c0 = sh.cnt[d0], c1 = sh.cnt[d1];
    const uint64_t m0 = sh.mass[d0], m1 = sh.mass[d1];
    uint32_t ci = c0 + c1;
    uint64_t mi = m0 + m1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t cu = __shfl_up(ci, off, 64);
      const uint64_t mu = shfl_up_u64(mi, off);
      if (lane >= off) { ci += cu; mi += mu; }
    }
    if (lane == 63) { sh.wave_cnt[wave] = ci; sh.wave_mass[wave] = mi; }
    __syncthreads();
    uint32_t c_above = sh.base_cnt + ci - (c0 + c1);
    uint64_t m_above = sh.base_mass + mi - (m0 + m1);
    for (int w = 0; w < wave; ++w) { c_above += sh.wave_cnt[w]; m_above += sh.wave_mass[w]; }
    // does the smallest member of the digit fall outside the kept prefix?
    auto fails = [&](uint32_t c, uint64_t m, uint32_t mn, uint32_t ca, uint64_t ma) -> bool {
      if (c == 0) return false;
      return (int64_t)ca + (int64_t)c > top_k || ma + m - fx_of_bits(mn) > top_p_fx || as_f32(mn) < thr;
    };
    const bool f0 = fails(c0, m0, sh.mn[d0], c_above, m_above);
    const bool f1 = fails(c1, m1, sh.mn[d1], c_above + c0, m_above + m0);
    if (f0) atomicMax(&sh.sel, d0);
    else if (f1) atomicMax(&sh.sel, d1);
    if (level == 0 && tid == kSampThreads - 1) sh.grand_mass = m_above + m0 + m1;
    __syncthreads();
    const int sel = sh.sel;
    if (sel < 0) { keep_all = true; break; }   // only possible at level 0 (see header)
    __syncthreads();
    if (sel == d0) { sh.base_cnt = c_above; sh.base_mass = m_above; sh.sel_cnt = c0; }
    if (sel == d1) { sh.base_cnt = c_above + c0; sh.base_mass = m_above + m0; sh.sel_cnt = c1; }
    prefix = (prefix << width) | (uint32_t)sel;
    __syncthreads();
  }

  Cut cut;
  if (keep_all) {
    cut.pivot = 0;
    cut.ties_kept = 0xffffffffu;
    cut.above = 0;            // not used when everything is kept
    cut.total = sh.grand_mass;
    return cut;
  }
  const uint32_t m = sh.sel_cnt, above = sh.base_cnt;
  const uint64_t mass_above = sh.base_mass, w = fx_of_bits(prefix);
  uint64_t t = m;
  t = min(t, (uint64_t)max((int64_t)0, top_k - (int64_t)above));
  if (mass_above > top_p_fx) t = 0;
  else if (w > 0) t = min(t, (top_p_fx - mass_above) / w + 1);
  if (as_f32(prefix) < thr) t = 0;
  cut.pivot = prefix;
  cut.ties_kept = (uint32_t)t;
  cut.above = above;
  cut.total = mass_above + t * w;
  return cut;
}

// Per-wave id segments: mass of tokens above the pivot and number of pivot ties in each.
__device__ __forceinline__ int seg_len(int vocab) { return (((vocab + kSampWaves - 1) / kSampWaves) + 63) & ~63; }

__device__ void segment_totals(const float* __restrict__ row, int vocab, const Cut& cut, SelectShared& sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = seg_len(vocab), lo = wave * seg, hi = min(lo + seg, vocab);
  uint64_t msum = 0;
  uint32_t ties = 0;
  for (int i = lo + lane; i < hi; i += 64) {
    const uint32_t bits = prob_bits(row[i]);
    if (bits > cut.pivot) msum += fx_of_bits(bits);
    ties += bits == cut.pivot;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    msum += shfl_xor_u64(msum, off);
    ties += __shfl_xor(ties, off, 64);
  }
  if (lane == 0) { sh.seg_mass[wave] = msum; sh.seg_ties[wave] = ties; }
  __syncthreads();
}

template <bool SAMPLE>
__global__ __launch_bounds__(kSampThreads) void select_kernel(SelectArgs a) {
  __shared__ SelectShared sh;
  const int b = blockIdx.x;
  const float* __restrict__ row = a.probs + (int64_t)b * a.row_stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Cut cut = select_cut(a, row, sh, b);
  const uint64_t wpiv = fx_of_bits(cut.pivot);
  if (a.keep_count && threadIdx.x == 0) {
    // tokens kept; under keep-all every token counts (zeros included, they carry no mass)
    a.keep_count[b] = cut.ties_kept == 0xffffffffu ? a.vocab : (int32_t)(cut.above + cut.ties_kept);
  }
  segment_totals(row, a.vocab, cut, sh);
  const int seg = seg_len(a.vocab);

  if (SAMPLE) {
    if (cut.total == 0) {               // degenerate row (all zero / NaN): token 0, like an empty draw
      if (threadIdx.x == 0) a.out_ids[b] = 0;
      return;
    }
    uint64_t r = (uint64_t)((double)a.uniform[b] * (double)cut.total);
    r = min(r, cut.total - 1);
    // which segment holds r
    uint64_t run = 0;
    uint32_t ties_before = 0;
    int target = -1;
    for (int w = 0; w < kSampWaves; ++w) {
      const uint32_t tk = (uint32_t)min((uint64_t)sh.seg_ties[w],
                                        (uint64_t)(cut.ties_kept > ties_before ? cut.ties_kept - ties_before : 0));
      const uint64_t wm = sh.seg_mass[w] + (uint64_t)tk * wpiv;
      if (run + wm > r) { target = w; break; }
      run += wm;
      ties_before += sh.seg_ties[w];
    }
    if (wave != target) return;        // target >= 0 because r < total
    const int lo = wave * seg, hi = min(lo + seg, a.vocab);
    for (int base = lo; base < hi; base += 64) {
      const int i = base + lane;
      const uint32_t bits = i < hi ? prob_bits(row[i]) : 0u;
      const bool tie = i < hi && bits == cut.pivot;
      const uint64_t tmask = __ballot(tie);
      const uint32_t rank = ties_before + __popcll(tmask & ((1ull << lane) - 1));
      uint64_t wt = 0;
      if (i < hi) {
        if (bits > cut.pivot) wt = fx_of_bits(bits);
        else if (tie && rank < cut.ties_kept) wt = wpiv;
      }
      uint64_t inc = wt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint64_t u = shfl_up_u64(inc, off);
        if (lane >= off) inc += u;
      }
      const uint64_t hit = __ballot(run + inc > r);
      if (hit) {
        if (lane == 0) a.out_ids[b] = base + (__ffsll((unsigned long long)hit) - 1);
        return;
      }
      run += ((uint64_t)__shfl((uint32_t)(inc >> 32), 63, 64) << 32) | __shfl((uint32_t)inc, 63, 64);
      ties_before += __popcll(tmask);
    }
  } else {
    float* __restrict__ out = a.renorm + (int64_t)b * a.renorm_stride;
    const float denom = (float)((double)cut.total / kFxOne);
    uint32_t ties_before = 0;
    for (int w = 0; w < wave; ++w) ties_before += sh.seg_ties[w];
    const int lo = wave * seg, hi = min(lo + seg, a.vocab);
    for (int base = lo; base < hi; base += 64) {
      const int i = base + lane;
      const float p = i < hi ? row[i] : 0.f;
      const uint32_t bits = prob_bits(p);
      const bool tie = i < hi && bits == cut.pivot;
      const uint64_t tmask = __ballot(tie);
      const uint32_t rank = ties_before + __popcll(tmask & ((1ull << lane) - 1));
      const bool keep = bits > cut.pivot || (tie && rank < cut.ties_kept);
      if (i < hi) out[i] = (keep && cut.total) ? p / denom : 0.f;
      ties_before += __popcll(tmask);
    }
  }
}

// per-thread scan of a row for (maximum, first index): 16-byte loads (8 x 16-bit or 4 x fp32 values) when
// the row allows it - a 2-byte load per lane leaves the memory pipeline at a quarter of its rate
template <typename Tag>
__device__ __forceinline__ void thread_argmax(const void* __restrict__ logits, int64_t base, int cols, int tid,
                                              float& best, int& idx) {
  constexpr int V = Elem<Tag>::kVec;
  best = -INFINITY;
  idx = 0x7fffffff;
  const char* row = (const char*)logits + base * Elem<Tag>::kBytes;
  int done = 0;
  if ((((uintptr_t)row) & 15) == 0) {
    const int nvec = cols / V;
    for (int i = tid; i < nvec; i += kSampThreads) {
      float f[V];
      unpack16<Tag>(ld16(row + (int64_t)i * 16), f);
#pragma unroll
      for (int e = 0; e < V; ++e)
        if (f[e] > best || idx == 0x7fffffff) { best = f[e]; idx = i * V + e; }   // ids ascend per thread
    }
    done = nvec * V;
  }
  for (int i = done + tid; i < cols; i += kSampThreads) {
    const float v = Elem<Tag>::load(logits, base + i);
    if (v > best || idx == 0x7fffffff || (v == best && i < idx)) { best = v; idx = i; }
  }
}

// ---- greedy argmax (sampler.py:63-65): first maximal index, NaN-free input assumed -----------
template <typename Tag>
__global__ __launch_bounds__(kSampThreads) void argmax_kernel(const void* __restrict__ logits, int64_t row_stride,
                                                              int vocab, int64_t* __restrict__ out) {
  __shared__ float s_val[kSampWaves];
  __shared__ int s_idx[kSampWaves];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)b * row_stride;
  float best;
  int idx;
  thread_argmax<Tag>(logits, base, vocab, tid, best, idx);
  auto better = [](float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); };
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    if (better(ov, oi, best, idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kSampWaves; ++w)
      if (better(s_val[w], s_idx[w], best, idx)) { best = s_val[w]; idx = s_idx[w]; }
    out[b] = idx == 0x7fffffff ? 0 : idx;
  }
}

// ---- vocab-parallel greedy: every rank reduces its own vocab shard to one (value, global index)
// pair per row, the pairs ([bs, 2] words per rank instead of [bs, vocab / tp] logits) are
// all-gathered, and the merge picks the maximum with the lowest global index - the token
// torch.argmax would return on the gathered row (logits_processor.py:362-369 + sampler.py:63-65).
template <typename Tag>
__global__ __launch_bounds__(kSampThreads) void argmax_shard_kernel(const void* __restrict__ logits,
                                                                    int64_t row_stride, int cols,
                                                                    int index_offset,
                                                                    int32_t* __restrict__ pairs) {
  __shared__ float s_val[kSampWaves];
  __shared__ int s_idx[kSampWaves];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t)b * row_stride;
  float best;
  int idx;
  thread_argmax<Tag>(logits, base, cols, tid, best, idx);
  auto better = [](float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); };
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(idx, off, 64);
    if (better(ov, oi, best, idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kSampWaves; ++w)
      if (better(s_val[w], s_idx[w], best, idx)) { best = s_val[w]; idx = s_idx[w]; }
    // an empty shard (cols == 0: every column of this rank is vocabulary padding) never wins
    pairs[2 * b] = (int32_t)as_u32(idx == 0x7fffffff ? -INFINITY : best);
    pairs[2 * b + 1] = idx == 0x7fffffff ? 0x7fffffff : idx + index_offset;
  }
}

// pairs: [shards][bs][2]; one thread per row
__global__ __launch_bounds__(256) void argmax_merge_kernel(const int32_t* __restrict__ pairs, int shards,
                                                           int bs, int64_t* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= bs) return;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int r = 0; r < shards; ++r) {
    const float v = as_f32((uint32_t)pairs[((int64_t)r * bs + b) * 2]);
    const int i = pairs[((int64_t)r * bs + b) * 2 + 1];
    if (v > best || (v == best && i < idx) || idx == 0x7fffffff) { best = v; idx = i; }
  }
  out[b] = idx == 0x7fffffff ? 0 : idx;
}

// ---- probs = softmax(logits / T) in place, fp32 (sampler.py:71-73) ---------------------------
__global__ __launch_bounds__(kSampThreads) void softmax_temperature_kernel(float* __restrict__ x, int64_t row_stride,
                                                                          const float* __restrict__ temps, int vocab) {
  __shared__ float s_red[kSampWaves];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* __restrict__ row = x + (int64_t)b * row_stride;
  const float t = temps ? temps[b] : 1.f;
  float mx = -INFINITY;
  for (int i = tid; i < vocab; i += kSampThreads) mx = fmaxf(mx, row[i] / t);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = s_red[0];
#pragma unroll
  for (int w = 1; w < kSampWaves; ++w) mx = fmaxf(mx, s_red[w]);
  __syncthreads();
  float sum = 0.f;
  for (int i = tid; i < vocab; i += kSampThreads) sum += expf(row[i] / t - mx);
  sum = wave_sum(sum);
  if (lane == 0) s_red[wave] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int w = 0; w < kSampWaves; ++w) sum += s_red[w];
  const float inv = 1.f / sum;
  for (int i = tid; i < vocab; i += kSampThreads) row[i] = expf(row[i] / t - mx) * inv;
}

}  // namespace sp

extern "C" int sp_argmax(const void* logits, int64_t row_stride, int batch_size, int vocab, int64_t* out_ids,
                         int dtype, void* stream) {
  SP_CHECK_ARG(batch_size >= 0 && vocab > 0);
  if (batch_size == 0) return SP_OK;
  SP_CHECK_ARG(logits && out_ids && row_stride >= vocab);
  SP_DISPATCH_DTYPE(dtype, (sp::argmax_kernel<Tag><<<dim3(batch_size), sp::kSampThreads, 0, (hipStream_t)stream>>>(
                               logits, row_stride, vocab, out_ids)));
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_argmax_shard(const void* logits, int64_t row_stride, int batch_size, int cols, int index_offset,
                               int32_t* out_pairs, int dtype, void* stream) {
  SP_CHECK_ARG(batch_size >= 0 && cols >= 0 && index_offset >= 0);
  if (batch_size == 0) return SP_OK;
  SP_CHECK_ARG(logits && out_pairs && row_stride >= cols);
  SP_DISPATCH_DTYPE(dtype, (sp::argmax_shard_kernel<Tag><<<dim3(batch_size), sp::kSampThreads, 0,
                                                          (hipStream_t)stream>>>(logits, row_stride, cols,
                                                                                 index_offset, out_pairs)));
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_argmax_merge(const int32_t* pairs, int num_shards, int batch_size, int64_t* out_ids,
                               void* stream) {
  SP_CHECK_ARG(batch_size >= 0 && num_shards > 0);
  if (batch_size == 0) return SP_OK;
  SP_CHECK_ARG(pairs && out_ids);
  sp::argmax_merge_kernel<<<dim3((batch_size + 255) / 256), 256, 0, (hipStream_t)stream>>>(pairs, num_shards,
                                                                                            batch_size, out_ids);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_softmax_temperature(float* logits, int64_t row_stride, const float* temperatures, int batch_size,
                                      int vocab, void* stream) {
  SP_CHECK_ARG(batch_size >= 0 && vocab > 0);
  if (batch_size == 0) return SP_OK;
  SP_CHECK_ARG(logits && row_stride >= vocab);
  sp::softmax_temperature_kernel<<<dim3(batch_size), sp::kSampThreads, 0, (hipStream_t)stream>>>(
      logits, row_stride, temperatures, vocab);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_top_k_top_p_min_p_sample(const float* probs, int64_t row_stride, const int32_t* top_ks,
                                           const float* top_ps, const float* min_ps, const float* uniform,
                                           int batch_size, int vocab, int64_t* out_ids, int32_t* keep_count,
                                           void* stream) {
  SP_CHECK_ARG(batch_size >= 0 && vocab > 0);
  if (batch_size == 0) return SP_OK;
  SP_CHECK_ARG(probs && uniform && out_ids && row_stride >= vocab);
  sp::SelectArgs a{probs, row_stride, top_ks, top_ps, min_ps, uniform, vocab, out_ids, nullptr, 0, keep_count};
  sp::select_kernel<true><<<dim3(batch_size), sp::kSampThreads, 0, (hipStream_t)stream>>>(a);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_top_k_top_p_min_p_renorm(const float* probs, int64_t row_stride, const int32_t* top_ks,
                                           const float* top_ps, const float* min_ps, int batch_size, int vocab,
                                           float* out, int64_t out_stride, int32_t* keep_count, void* stream) {
  SP_CHECK_ARG(batch_size >= 0 && vocab > 0);
  if (batch_size == 0) return SP_OK;
  SP_CHECK_ARG(probs && out && row_stride >= vocab && out_stride >= vocab);
  sp::SelectArgs a{probs, row_stride, top_ks, top_ps, min_ps, nullptr, vocab, nullptr, out, out_stride, keep_count};
  sp::select_kernel<false><<<dim3(batch_size), sp::kSampThreads, 0, (hipStream_t)stream>>>(a);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
