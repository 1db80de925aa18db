// Paged (page_size = 1) decode attention for gfx950: one fused pass, split-KV + merge.
//
// Replaces the reference's two-pass Triton decode (nn/attention/triton_attn/decode_attention.py:
// stage 1 QK^T -> fp32 logits in HBM, stage 2 softmax.V) and flashinfer's paged decode wrapper.
//
// Shape of the problem: 4 flop per KV byte - strictly HBM-bound, MFMA is not needed (a group of
// 4-8 query heads per KV head is not a dense tile).  Everything is organised around reading the
// gathered KV rows once, with full-line coalesced 16-byte-per-lane loads straight to VGPRs
// (cdna_hip_programming.md 'GEMV / M<=16' row: no LDS round trip for once-read operands):
//
//   * a wave-instruction loads 1 KiB = RPL rows of D elements (RPL = 4 for 16-bit D=128); the
//     rows are `HH` adjacent KV heads of one token (contiguous in the [P+1, Hkv, D] pool) times
//     RPL/HH consecutive tokens;
//   * every lane owns 16 bytes (VEC dims) of its row and all G query heads of that KV head:
//     QK^T partials by v_dot2c (16-bit) / fma (fp32), then a DPP all-reduce over the LPR lanes of
//     the row; PV accumulates VEC dims x G heads in registers;
//   * each (wave, row) pair runs an independent online softmax over its tokens; the streams are
//     merged through LDS at the end of the workgroup, and split-KV partials (normalised o +
//     log2-sum-exp) are merged by a second tiny kernel;
//   * slot indices come from one coalesced read of req_to_token per 64 tokens per wave and are
//     handed to the lanes by ds_bpermute; K/V loads of the next batch are issued before the
//     current batch is consumed (two register sets).
//
// Launch geometry depends only on (batch, heads, max_seq_len, chunk): graph-capturable.
// Workgroups whose chunk starts beyond their request's seq_len exit at once.
#include <stdlib.h>

#include "sp_common.h"
#include "attention_internal.h"

namespace sp {


static constexpr float kLog2e = 1.4426950408889634f;
static constexpr float kNegBig = -1.0e30f;

// ---- all-reduce (sum) across the LPR consecutive lanes that share a KV row ---------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int LPR>
__device__ __forceinline__ float row_allreduce(float v) {
  static_assert(LPR == 8 || LPR == 16 || LPR == 32 || LPR == 64, "lanes per row");
  v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]  (xor 1)
  v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]  (xor 2)
  if constexpr (LPR == 8) {
    v += dpp_mov<0x141>(v);  // row_half_mirror: lane i <-> 7-i inside each 8 lanes
  } else {
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x128>(v);  // row_ror:8   -> every lane of the 16-lane row has the row sum
    if constexpr (LPR >= 32)
      v += __builtin_bit_cast(
          float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // xor 16
    if constexpr (LPR == 64) v += __shfl_xor(v, 32, 64);
  }
  return v;
}

template <typename Tag>
__device__ __forceinline__ float dot16(const u32x4& a, const u32x4& b, float acc);
template <>
__device__ __forceinline__ float dot16<f32_tag>(const u32x4& a, const u32x4& b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_fmaf(as_f32(a[i]), as_f32(b[i]), acc);
  return acc;
}
template <>
__device__ __forceinline__ float dot16<bf16_tag>(const u32x4& a, const u32x4& b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(a[i]), as_bf16x2(b[i]), acc, false);
  return acc;
}
template <>
__device__ __forceinline__ float dot16<f16_tag>(const u32x4& a, const u32x4& b, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2(as_f16x2(a[i]), as_f16x2(b[i]), acc, false);
  return acc;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename Tag, int D, int G>
struct DecodeCfg {
  typedef Elem<Tag> E;
  static constexpr int VEC = E::kVec;            // elements per lane
  static constexpr int LPR = D / VEC;            // lanes per KV row
  static constexpr int RPL = 64 / LPR;           // rows per wave-load
  static constexpr int NB = 4;                   // wave-loads per batch (per K and per V)
  static constexpr int WAVES = 4;
  static constexpr int kLdsFloats = WAVES * RPL * G * (D + 2);
};

template <typename Tag, int D, int G>
__global__ __launch_bounds__(256) void decode_attn_kernel(DecodeArgs a) {
  typedef DecodeCfg<Tag, D, G> C;
  typedef Elem<Tag> E;
  constexpr int VEC = C::VEC, LPR = C::LPR, RPL = C::RPL, NB = C::NB, WAVES = C::WAVES;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // blockIdx -> (item, head group); sibling head groups are adjacent so the two halves of a
  // token's KV row are fetched at about the same time.  With a plan (sp_decode_plan, built once
  // per step and shared by all layers) item -> (request b, split c) walks only the NON-EMPTY
  // splits, densely: the hardware deals workgroups to the 8 XCDs round-robin by blockIdx, each XCD
  // with its own queue, so a static (b, c) grid with its empty splits leaves whole XCDs idle on
  // ragged batches (measured on U[128,4096] contexts: 0.74x the fixed-length rate).
  const int hg = blockIdx.x % a.head_groups;
  const int item = blockIdx.x / a.head_groups;
  int b, c, chunk, slot0;
  if (!decode_item(a, item, b, c, chunk, slot0)) return;

  const int seq = min((int)load_idx(a.seq_lens, b, a.idx64), a.max_len);
  const int cs = c * chunk;
  if (cs >= seq || slot0 + c >= a.max_slots) return;  // (also covers seq == 0) uniform for the whole workgroup
  const int ce = min(cs + chunk, seq);
  const int nsplit = (seq + chunk - 1) / chunk;
  const int64_t req = load_idx(a.req_idx, b, a.idx64);
  const int64_t kv0 = a.kv_start ? load_idx(a.kv_start, b, a.idx64) : 0;
  const int32_t* idx_row = a.r2t + req * a.r2t_stride + kv0;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = lane / LPR, col = lane % LPR;
  const int HH = 1 << a.hh_shift;
  const int TPL = RPL >> a.hh_shift;  // tokens per wave-load
  const int head_local = row & (HH - 1);
  const int tok_in_load = row >> a.hh_shift;
  const int kv_head = hg * HH + head_local;

  // q fragment: G heads x VEC dims, kept packed
  u32x4 qf[G];
  {
    const char* qp = (const char*)a.q +
                     ((int64_t)b * a.q_stride + (int64_t)kv_head * G * D + col * VEC) * E::kBytes;
#pragma unroll
    for (int g = 0; g < G; ++g) qf[g] = ld16(qp + (int64_t)g * D * E::kBytes);
  }
  const int64_t lane_off = ((int64_t)kv_head * D + col * VEC) * E::kBytes;
  const int64_t tok_bytes = a.kv_stride * E::kBytes;

  float m[G], l[G], acc[G][VEC];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = kNegBig;
    l[g] = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[g][e] = 0.f;
  }
  const float cap = a.logit_cap;
  const float qk_scale = cap > 0.f ? a.sm_scale : a.sm_scale * kLog2e;

  // this wave's contiguous share of the chunk's ACTUAL tokens (a ragged last chunk is split
  // evenly too, otherwise wave 0 alone sets the workgroup's duration), in whole wave-loads
  const int sub = ((ce - cs + WAVES * TPL - 1) / (WAVES * TPL)) * TPL;
  const int ws = cs + wave * sub;
  const int we = min(ws + sub, ce);

  // pieces of <= 64 tokens: one index register each, fetched one piece ahead so the dependent
  // req_to_token -> KV row chain is paid once per workgroup, not once per piece
  int nextidx = (ws + lane < we) ? idx_row[ws + lane] : 0;
  for (int ps = ws; ps < we; ps += 64) {
    const int n = min(64, we - ps);
    const int myidx = nextidx;
    nextidx = (ps + 64 + lane < we) ? idx_row[ps + 64 + lane] : 0;
    const int nloads = (n + TPL - 1) / TPL;

    u32x4 kA[NB], vA[NB], kB[NB], vB[NB];
    auto issue = [&](u32x4(&kr)[NB], u32x4(&vr)[NB], int first_load) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int tl = (first_load + j) * TPL + tok_in_load;
        // out-of-range rows read the reserved dummy slot 0 (always mapped) and are masked below
        const int slot = __shfl(myidx, tl & 63, 64);
        const int64_t off = (tl < n ? (int64_t)slot : 0) * tok_bytes + lane_off;
        kr[j] = ld16(a.kbuf + off);
        vr[j] = ld16(a.vbuf + off);
      }
    };
    auto consume = [&](const u32x4(&kr)[NB], const u32x4(&vr)[NB], int first_load) {
      float s[NB][G];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) s[j][g] = dot16<Tag>(qf[g], kr[j], 0.f);
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) s[j][g] = row_allreduce<LPR>(s[j][g]);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const bool valid = (first_load + j) * TPL + tok_in_load < n;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float x = s[j][g] * qk_scale;
          if (cap > 0.f) x = cap * tanhf(x / cap) * kLog2e;
          s[j][g] = valid ? x : -INFINITY;
        }
      }
      float alpha[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float mx = m[g];
#pragma unroll
        for (int j = 0; j < NB; ++j) mx = fmaxf(mx, s[j][g]);
        alpha[g] = fast_exp2(m[g] - mx);
        m[g] = mx;
        float ps_ = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          s[j][g] = fast_exp2(s[j][g] - mx);
          ps_ += s[j][g];
        }
        l[g] = l[g] * alpha[g] + ps_;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[g][e] *= alpha[g];
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        float vf[VEC];
        unpack16<Tag>(vr[j], vf);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[g][e] = __builtin_fmaf(s[j][g], vf[e], acc[g][e]);
      }
    };

    issue(kA, vA, 0);
    for (int ld = 0; ld < nloads; ld += 2 * NB) {
      if (ld + NB < nloads) issue(kB, vB, ld + NB);
      consume(kA, vA, ld);
      if (ld + NB < nloads) {
        if (ld + 2 * NB < nloads) issue(kA, vA, ld + 2 * NB);
        consume(kB, vB, ld + NB);
      }
    }
  }

  // ---- merge the WAVES x RPL streams of this workgroup through LDS ---------------------------
  float* sm_acc = smem;                                // [WAVES][RPL][G][D]
  float* sm_ml = smem + WAVES * RPL * G * D;           // [WAVES][RPL][G][2]
  {
    const int stream = wave * RPL + row;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float* dst = sm_acc + ((stream * G + g) * D + col * VEC);
#pragma unroll
      for (int e = 0; e < VEC; e += 4)
        *(float4*)(dst + e) = make_float4(acc[g][e], acc[g][e + 1], acc[g][e + 2], acc[g][e + 3]);
      if (col == 0) {
        sm_ml[(stream * G + g) * 2] = m[g];
        sm_ml[(stream * G + g) * 2 + 1] = l[g];
      }
    }
  }
  __syncthreads();
  const int outs = HH * G * D;
  for (int i = threadIdx.x; i < outs; i += 256) {
    const int d = i % D;
    const int g = (i / D) % G;
    const int hl = i / (D * G);
    float M = kNegBig;
    for (int w = 0; w < WAVES; ++w)
      for (int t = 0; t < TPL; ++t)
        M = fmaxf(M, sm_ml[(((w * RPL) + (t << a.hh_shift) + hl) * G + g) * 2]);
    float L = 0.f, O = 0.f;
    for (int w = 0; w < WAVES; ++w)
      for (int t = 0; t < TPL; ++t) {
        const int stream = w * RPL + (t << a.hh_shift) + hl;
        const float wgt = fast_exp2(sm_ml[(stream * G + g) * 2] - M);
        L += sm_ml[(stream * G + g) * 2 + 1] * wgt;
        O += sm_acc[(stream * G + g) * D + d] * wgt;
      }
    const int h = (hg * HH + hl) * G + g;
    const float o = O / L;
    if (nsplit == 1) {
      E::store(a.out, (int64_t)b * a.o_stride + (int64_t)h * D + d, o * a.out_scale);
    } else {
      const int64_t pi = (int64_t)h * a.max_slots + (slot0 + c);
      a.part_o[pi * D + d] = o;
      if (d == 0) a.part_lse[pi] = M + __builtin_amdgcn_logf(L);  // v_log_f32 = log2
    }
  }
}

// one wave per (request, q head): combine the split partials by their log2-sum-exp (decode_merge_rows).  The launch
// is latency-bound at small batch (7 - 9 us for a few KB).
template <typename Tag, int D>
__global__ __launch_bounds__(256) void decode_merge_kernel(DecodeArgs a) {
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= a.bs * a.Hq) return;
  const int lane = threadIdx.x & 63;
  const int b = pair / a.Hq, h = pair - b * a.Hq;
  if (a.rplan) {               // range geometry: the request's pieces from its place on the line
    if (!range_plan_matches(a.rplan, a.ranges, a.bs)) return;   // (as the range kernel: not this launch's plan)
    int len, nsplit, slot0;
    range_request(a.rplan, b, len, nsplit, slot0);
    if (len <= 0 || nsplit <= 1) return;
    const int hs[1] = {h};
    if (nsplit <= 4) decode_merge_rows<Tag, D, 1, 4>(a, b, hs, lane, nsplit, slot0);
    else decode_merge_rows<Tag, D, 1>(a, b, hs, lane, nsplit, slot0);
    return;
  }
  const int seq = min((int)load_idx(a.seq_lens, b, a.idx64), a.max_len);
  const int chunk = a.plan ? a.plan[1] : a.chunk;
  const int slot0 = a.plan ? a.plan[kPlanHdr + b] : b * a.num_splits;
  int nsplit = (seq + chunk - 1) / chunk;
  if (nsplit <= 1) return;  // written directly by the attention kernel (or empty row)
  nsplit = min(nsplit, a.max_slots - slot0);   // never past the workspace (a plan cut short by a broken bound)
  if (nsplit < 1) return;
  const int hs[1] = {h};
  if (nsplit <= 4) decode_merge_rows<Tag, D, 1, 4>(a, b, hs, lane, nsplit, slot0);
  else decode_merge_rows<Tag, D, 1>(a, b, hs, lane, nsplit, slot0);
}

// Build the step's plan.  max_items = 0 (ABI 9: the caller's launches all take the range geometry): the item section is
// its header alone, [0, chunk, 0, 0], and only the range section behind it is built.  Otherwise the (request, split) items:
// plan[0] = number of non-empty (request, split) items the plan lists, plan[1] = chunk,
// plan[2] = the number the lengths NEED (> plan[0]: items were cut at max_items because the host's bound on
// sum(seq_lens) does not hold - an error the host reports, sp_decode_plan), plan[3] = 0,
// plan[4 + b] = first partial slot of request b (exclusive scan of its split count), the items (b, c) from
// plan[4 + bs].  One workgroup; requests in tiles of 256 with a running offset.  Items are
// emitted longest-first: all full splits, then the ragged last splits in four length classes (longest
// quarter first), so the launch ends on its shortest items.  The chunk travels IN the plan: the
// attention and merge kernels read it from there, so one captured launch serves any split size.
__global__ __launch_bounds__(256) void decode_plan_kernel(int32_t* __restrict__ plan,
                                                           const void* __restrict__ seq_lens,
                                                           int idx64, int bs, int chunk, int max_len,
                                                           int max_items, int ranges) {
  __shared__ int s_scan[256];
  __shared__ int s_base;
  __shared__ unsigned long long s_keys;
  if (threadIdx.x == 0) s_keys = 0;
  int32_t* items = plan + kPlanHdr + bs;
  // pass -1: slot0[b]; pass 0: full splits; passes 1..4: ragged tails by length class
  for (int pass = -1; pass < 5 && max_items > 0; ++pass) {
    if (pass <= 0) {
      __syncthreads();
      if (threadIdx.x == 0 && pass == -1) s_base = 0;
      if (threadIdx.x == 0 && pass == 0) s_base = 0;
      __syncthreads();
    }
    for (int t0 = 0; t0 < bs; t0 += 256) {
      const int b = t0 + threadIdx.x;
      int nfull = 0, tail = 0, rem = 0;
      if (b < bs) {
        // a length beyond the host-supplied bound would index past the plan and the partials
        const int seq = min((int)load_idx(seq_lens, b, idx64), max_len);
        nfull = seq > 0 ? seq / chunk : 0;
        rem = seq > 0 ? seq % chunk : 0;
        // tail class 3 = longest quarter of the chunk ... 0 = shortest; pass 1 takes class 3
        tail = rem > 0 && (4 - pass) == (int)(((int64_t)rem * 4 - 1) / chunk) ? 1 : 0;
      }
      const int mine = pass == -1 ? nfull + (rem > 0 ? 1 : 0) : (pass == 0 ? nfull : tail);
      if (pass == -1 && b < bs) atomicAdd(&s_keys, (unsigned long long)(nfull * (int64_t)chunk + rem));
      s_scan[threadIdx.x] = mine;
      __syncthreads();
      for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
        const int v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
      }
      const int base = s_base + s_scan[threadIdx.x] - mine;
      if (pass == -1) {
        if (b < bs) plan[kPlanHdr + b] = base;
      } else {
        // (the host sizes max_items from its own bound on sum(seq_lens); lengths that break that bound
        // lose their surplus items here, and the kernels skip slots >= max_items, instead of writing past
        // the plan and the partials)
        for (int i = 0; i < mine && base + i < max_items; ++i) {
          items[2 * (base + i)] = b;
          items[2 * (base + i) + 1] = pass == 0 ? i : nfull;
        }
      }
      __syncthreads();
      if (threadIdx.x == 255) s_base += s_scan[255];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    plan[0] = max_items > 0 ? min(s_base, max_items) : 0;
    plan[1] = chunk;
    plan[2] = max_items > 0 ? s_base : 0;
    plan[3] = max_items > 0 ? (int)min(s_keys, 0x7fffffffULL) : 0;   // keys this step gathers per kv head (DecodeArgs::nt_min_keys)
  }
  if (ranges <= 0) return;
  // ---- the range section (DecodeArgs::rplan): pos[] = exclusive scan of len + kRangeReqCost over the non-empty requests
  int32_t* rp = plan + plan_item_words(bs, max_items);
  int32_t* pos = rp + kRangeHdr;
  int32_t* start = pos + bs + 1;
  __syncthreads();
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int t0 = 0; t0 < bs; t0 += 256) {
    const int b = t0 + threadIdx.x;
    int cost = 0;
    if (b < bs) {
      const int seq = min((int)load_idx(seq_lens, b, idx64), max_len);
      cost = seq > 0 ? seq + kRangeReqCost : 0;
    }
    s_scan[threadIdx.x] = cost;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const int v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    if (b < bs) pos[b] = s_base + s_scan[threadIdx.x] - cost;
    __syncthreads();
    if (threadIdx.x == 255) s_base += s_scan[255];
    __syncthreads();
  }
  const int T = s_base;                               // (< 2^31: the host checks bs * (max_len + cost))
  // (not rounded to the tile: a batch of equal lengths whose size is a multiple of the piece count - decode steps of a
  // full graph bucket - then has its cuts exactly between requests: no partials at all)
  const int R = max(kRangeMin, (int)(((int64_t)T + ranges - 1) / ranges));
  const int rcount = T > 0 ? (T + R - 1) / R : 0;
  if (threadIdx.x == 0) {
    pos[bs] = T;
    rp[0] = rcount; rp[1] = R; rp[2] = ranges; rp[3] = bs;     // (2, 3: what the section was built for, range_plan_matches)
  }
  __syncthreads();                                    // pos[] is read back below
  // last request whose position is <= g (empty requests share the position of the next non-empty one, which is the last
  // of the equals: the one found)
  auto at_or_before = [&](int g) {
    int lo = 0, hi = bs;                              // first b in [0, bs] with pos[b] > g (pos[bs] = T > g)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (pos[mid] > g) hi = mid; else lo = mid + 1;
    }
    return lo - 1;
  };
  for (int j = threadIdx.x; j < ranges; j += 256) {
    int first = -1;
    if (j < rcount) {
      const int g = j * R;
      int b = at_or_before(g);
      if (g >= pos[b + 1] - kRangeReqCost)            // in the empty positions behind b's keys: the next request
        b = pos[b + 1] < T ? at_or_before(pos[b + 1]) : -1;
      if (b >= 0 && (int64_t)pos[b] < (int64_t)g + R) first = b;
    }
    start[j] = first;
  }
}

template <typename Tag, int D, int G>
static int launch_decode(const DecodeArgs& a, hipStream_t st) {
  typedef DecodeCfg<Tag, D, G> C;
  const size_t lds = (size_t)C::kLdsFloats * sizeof(float);
  const unsigned grid = (unsigned)((a.plan ? (int64_t)a.max_slots : (int64_t)a.bs * a.num_splits) * a.head_groups);
  if (lds > 64 * 1024) {         // the limit is the device's: raised once per device of the process
    static PerDevice<int> raised;
    int ok = 0;
    raised.get(ok, [&](int) {
      return hipFuncSetAttribute((const void*)decode_attn_kernel<Tag, D, G>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 1 : 0;
    });
    if (!ok) return SP_ERR_LAUNCH;
  }
  decode_attn_kernel<Tag, D, G><<<dim3(grid), 256, lds, st>>>(a);
  SP_LAUNCH_CHECK();
  g_decode_last_kernel = 1;
  if (a.num_splits > 1) {
    decode_merge_kernel<Tag, D><<<dim3((a.bs * a.Hq + 3) / 4), 256, 0, st>>>(a);
    SP_LAUNCH_CHECK();
  }
  return SP_OK;
}

template <typename Tag, int D>
static int dispatch_group(const DecodeArgs& a, int G, hipStream_t st) {
  switch (G) {
    case 1: return launch_decode<Tag, D, 1>(a, st);
    case 2: return launch_decode<Tag, D, 2>(a, st);
    case 4: return launch_decode<Tag, D, 4>(a, st);
    case 8: return launch_decode<Tag, D, 8>(a, st);
    default: return SP_ERR_UNSUPPORTED;
  }
}

template <typename Tag>
static int dispatch_dim(const DecodeArgs& a, int D, int G, hipStream_t st) {
  switch (D) {
    case 64: return dispatch_group<Tag, 64>(a, G, st);
    case 128: return dispatch_group<Tag, 128>(a, G, st);
    default: return SP_ERR_UNSUPPORTED;
  }
}

template <typename Tag, int D, int G>
static int occupancy_of() {
  typedef DecodeCfg<Tag, D, G> C;
  const size_t lds = (size_t)C::kLdsFloats * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)decode_attn_kernel<Tag, D, G>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decode_attn_kernel<Tag, D, G>, 256, lds) !=
      hipSuccess)
    return -1;
  return n;
}

int decode_occupancy(int head_dim, int group, int dtype) {
  if (dtype != SP_BF16 || head_dim != 128) return -1;
  switch (group) {
    case 1: return occupancy_of<bf16_tag, 128, 1>();
    case 2: return occupancy_of<bf16_tag, 128, 2>();
    case 4: return occupancy_of<bf16_tag, 128, 4>();
    case 8: return occupancy_of<bf16_tag, 128, 8>();
    default: return -1;
  }
}

template <typename Tag>
static int merge_dim(const DecodeArgs& a, int D, hipStream_t st) {
  const dim3 grid((a.bs * a.Hq + 3) / 4);
  if (D == 128) decode_merge_kernel<Tag, 128><<<grid, 256, 0, st>>>(a);
  else if (D == 64) decode_merge_kernel<Tag, 64><<<grid, 256, 0, st>>>(a);
  else return SP_ERR_UNSUPPORTED;
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int run_decode_merge(const DecodeArgs& a, int head_dim, int dtype, hipStream_t st) {
  if (a.num_splits <= 1 && !a.rplan) return SP_OK;
  SP_DISPATCH_DTYPE(dtype, return (merge_dim<Tag>(a, head_dim, st)));
}

// Kernel choice.  16-bit dtypes go to the matrix-core kernel (decode_mfma.hip; measured equal or
// faster than the VALU kernel on every shape tried, 2.3x at G = 8 where the VALU kernel needs 256
// VGPRs); fp32 and groups wider than 16 stay on the VALU kernel below.
// sp_debug_set("decode_kernel", 1 = valu | 2 = mfma | 0 = default) overrides the choice (A/B
// measurements and tests only; a process-wide variable, never the environment on the call path).
static int g_decode_kernel_forced = 0;
void set_decode_kernel(int which) { g_decode_kernel_forced = which; }
// sp_debug_set("decode_nt_min_mb", n): the K + V bytes a decode launch has to gather (over all its kv heads) before its
// gathers become non-temporal loads; 0 = always, -1 = never, -2 = back to the default.  See DecodeArgs::nt_min_keys and
// DESIGN 4.1 for the measurements behind the default.
static constexpr int kDecodeNtMinMbDefault = 0;
static int g_decode_nt_min_mb = kDecodeNtMinMbDefault;
void set_decode_nt_min_mb(int mb) { g_decode_nt_min_mb = mb == -2 ? kDecodeNtMinMbDefault : mb; }
// sp_debug_set("decode_ranges", n): 0 = launches never take the range geometry (DecodeArgs::rplan) even when their plan
// carries one, -1 = default (they do wherever the range kernel applies)
static int g_decode_ranges = -1;
void set_decode_ranges(int n) { g_decode_ranges = n < 0 ? -1 : n; }
int g_decode_last_kernel = 0;    // sp_debug_get("decode_last_kernel"), attention_internal.h
static int decode_kernel_choice(int group, int dtype) {
  if (dtype == SP_F32 || group > 16) return 1;
  if (g_decode_kernel_forced == 1 || g_decode_kernel_forced == 2) return g_decode_kernel_forced;
  return 2;
}

int run_decode(const DecodeArgs& a, int head_dim, int group, int dtype, hipStream_t st) {
  if (a.rplan) {                     // range geometry: the range kernel, whatever "decode_kernel" says
    const int rc = run_decode_mfma(a, head_dim, dtype, st);
    return rc == SP_OK ? run_decode_merge(a, head_dim, dtype, st) : rc;
  }
  if (a.kv8) {                       // fp8 pool: matrix-core kernel only
    if (a.Hq / a.Hkv > 16) return SP_ERR_UNSUPPORTED;
    const int rc = run_decode_mfma(a, head_dim, dtype, st);
    return rc == SP_OK ? run_decode_merge(a, head_dim, dtype, st) : rc;
  }
  if (group > 8 || decode_kernel_choice(a.Hq / a.Hkv, dtype) == 2) {   // groups 9..16: matrix-core kernel only
    const int rc = run_decode_mfma(a, head_dim, dtype, st);
    if (rc == SP_OK) return run_decode_merge(a, head_dim, dtype, st);
    if (rc != SP_ERR_UNSUPPORTED) return rc;
  }
  SP_DISPATCH_DTYPE(dtype, return (dispatch_dim<Tag>(a, head_dim, group, st)));
}

int decode_heads_per_load_shift(int num_kv_heads, int head_dim, int dtype, int* head_groups) {
  const int vec = dtype == SP_F32 ? 4 : 8;
  const int rpl = 64 / (head_dim / vec);
  int hh = 1, shift = 0;
  while (hh * 2 <= rpl && num_kv_heads % (hh * 2) == 0) { hh *= 2; ++shift; }
  *head_groups = num_kv_heads / hh;
  return shift;
}

}  // namespace sp

using namespace sp;

static inline int64_t num_splits_for(int64_t max_seq_len, int chunk) {
  int64_t s = (max_seq_len + chunk - 1) / chunk;
  return s < 1 ? 1 : s;
}

// Partial slots (= work items) a step can need: every request has ceil(seq / chunk) splits, so the sum is
// at most bs * ceil(max_seq_len / chunk) AND at most kv_tokens / chunk + bs, where kv_tokens bounds
// sum(seq_lens) (the step's seq_lens_sum, or the KV pool size).  The second bound is what keeps the split
// workspace independent of the model's context length.
extern "C" int64_t sp_decode_plan_slots(int batch_size, int64_t kv_tokens, int64_t max_seq_len, int chunk) {
  if (batch_size <= 0 || chunk <= 0) return 0;
  const int64_t by_len = (int64_t)batch_size * num_splits_for(max_seq_len, chunk);
  if (kv_tokens < 0) return by_len;
  const int64_t by_sum = kv_tokens / chunk + batch_size;
  return by_len < by_sum ? by_len : by_sum;
}

extern "C" size_t sp_decode_attention_workspace_bytes(int64_t max_slots, int num_q_heads, int v_head_dim) {
  if (max_slots <= 0 || num_q_heads <= 0 || v_head_dim <= 0) return 16;
  return (size_t)max_slots * num_q_heads * (v_head_dim + 1) * sizeof(float) + 16;
}

// diagnostic (not part of the forward path): resident workgroups per CU the runtime reports for the
// bf16 D=128 decode kernel of a given group size
extern "C" SP_API int sp_debug_decode_occupancy(int head_dim, int group, int dtype) {
  return decode_occupancy(head_dim, group, dtype);
}

// the line of a range plan (attention_internal.h) is indexed with int32
static inline bool range_line_fits(int batch_size, int64_t max_seq_len) {
  return (int64_t)batch_size * (max_seq_len + kRangeReqCost) < 0x7fffffffLL;
}

extern "C" size_t sp_decode_plan_bytes(int batch_size, int64_t max_slots, int ranges) {
  if (batch_size <= 0 || max_slots < 0 || (max_slots == 0 && ranges <= 0)) return 16;
  int64_t words = plan_item_words(batch_size, max_slots);
  if (ranges > 0) words += kRangeHdr + (int64_t)batch_size + 1 + ranges;
  return (size_t)words * sizeof(int32_t);
}

extern "C" int sp_decode_ranges(int num_q_heads, int num_kv_heads, int head_dim, int dtype, int kv_dtype) {
  if (num_q_heads <= 0 || num_kv_heads <= 0 || num_q_heads % num_kv_heads) return 0;
  return decode_mfma_ranges(num_q_heads, num_kv_heads, head_dim, dtype, kv_dtype == SP_FP8_E5M2);
}

extern "C" int sp_decode_plan(int32_t* plan, size_t plan_bytes, const void* seq_lens, int idx64,
                              int batch_size, int64_t max_seq_len, int chunk, int64_t max_slots,
                              int ranges, void* stream) {
  SP_CHECK_ARG(plan && seq_lens && batch_size >= 0 && chunk >= 4 && chunk % 4 == 0);
  SP_CHECK_ARG(max_seq_len >= 0 && max_seq_len <= 0x7fffffffLL && max_slots >= 0 && max_slots <= 0x3fffffffLL);
  SP_CHECK_ARG(ranges >= 0 && ranges <= 65536);
  SP_CHECK_ARG(max_slots > 0 || ranges > 0);           // (a plan with neither section plans nothing)
  SP_CHECK_ARG(ranges == 0 || range_line_fits(batch_size, max_seq_len));
  if (plan_bytes < sp_decode_plan_bytes(batch_size, max_slots, ranges)) return SP_ERR_WORKSPACE;
  decode_plan_kernel<<<dim3(1), 256, 0, (hipStream_t)stream>>>(plan, seq_lens, idx64, batch_size,
                                                               chunk, (int)max_seq_len, (int)max_slots, ranges);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_decode_attention(void* out, const void* q, const void* k_buffer,
                                   const void* v_buffer, const int32_t* req_to_token,
                                   int64_t req_to_token_stride, const void* req_pool_indices,
                                   const void* seq_lens, const void* kv_start, int idx64,
                                   int batch_size, int num_q_heads, int num_kv_heads, int head_dim,
                                   int64_t q_stride, int64_t out_stride, int64_t kv_buffer_stride,
                                   float sm_scale, float logit_cap, float k_scale, float v_scale,
                                   int64_t max_seq_len, int chunk, int64_t max_slots, int ranges,
                                   void* workspace, size_t workspace_bytes, const int32_t* plan, size_t plan_bytes,
                                   int dtype, int kv_dtype, void* stream) {
  SP_CHECK_ARG(out && q && k_buffer && v_buffer && req_to_token && req_pool_indices && seq_lens);
  SP_CHECK_ARG(batch_size >= 0 && num_q_heads > 0 && num_kv_heads > 0 && head_dim > 0);
  SP_CHECK_ARG(num_q_heads % num_kv_heads == 0 && max_seq_len >= 0);
  SP_CHECK_ARG(k_scale > 0.f && v_scale > 0.f);
  // each wave takes chunk/4 tokens in pieces of 64: keep the split a multiple of 4
  SP_CHECK_ARG(chunk >= 4 && chunk % 4 == 0);
  SP_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k_buffer & 15) == 0 &&
               ((uintptr_t)v_buffer & 15) == 0);
  if (batch_size == 0) return SP_OK;
  const bool kv8 = kv_dtype == SP_FP8_E5M2;
  if (!kv8 && kv_dtype != dtype) return SP_ERR_UNSUPPORTED;
  if (kv8 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  const int eb = dtype == SP_F32 ? 4 : 2;
  const int vec = 16 / eb;
  SP_CHECK_ARG(q_stride % vec == 0 && kv_buffer_stride % (kv8 ? 8 : vec) == 0);
  SP_CHECK_ARG(kv_buffer_stride > 0 && kv_buffer_stride * (kv8 ? 1 : eb) <= 0xffffffffLL);   // token stride in bytes: 32 bits
  SP_CHECK_ARG(!kv8 || (((uintptr_t)k_buffer & 7) == 0 && ((uintptr_t)v_buffer & 7) == 0));
  const int G = num_q_heads / num_kv_heads;
  // query heads per KV head: the matrix-core kernel tiles up to 16 of them as MFMA columns (16-bit
  // dtypes); the VALU kernel (fp32) holds up to 8 in registers
  if (G > 16 || (G > 8 && dtype == SP_F32)) return SP_ERR_UNSUPPORTED;
  const int Gk = G;
  if (head_dim != 64 && head_dim != 128) return SP_ERR_UNSUPPORTED;

  // `chunk`: the split size of the plan-less grid; with a plan, the SMALLEST split size the plan may carry
  // (the kernels read the actual one from the plan; the host value only decides whether a merge can be needed)
  const int64_t S = num_splits_for(max_seq_len, chunk);
  if (max_seq_len > 0x7fffffffLL) return SP_ERR_INVALID_ARG;
  SP_CHECK_ARG(ranges >= 0 && ranges <= 65536);
  if (!plan) max_slots = (int64_t)batch_size * S;        // static (request, split) grid
  SP_CHECK_ARG(max_slots >= 0 && max_slots <= 0x3fffffffLL);
  // the plan buffer holds what (batch_size, max_slots, ranges) say it holds: the sections are located from them
  if (plan && plan_bytes < sp_decode_plan_bytes(batch_size, max_slots, ranges)) return SP_ERR_WORKSPACE;
  SP_CHECK_ARG(!plan || max_slots > 0 || ranges > 0);
  DecodeArgs a;
  a.out = out; a.q = q; a.kbuf = (const char*)k_buffer; a.vbuf = (const char*)v_buffer;
  a.r2t = req_to_token; a.r2t_stride = req_to_token_stride; a.req_idx = req_pool_indices;
  a.seq_lens = seq_lens; a.kv_start = kv_start; a.idx64 = idx64; a.bs = batch_size;
  a.Hq = num_q_heads; a.Hkv = num_kv_heads; a.q_stride = q_stride; a.o_stride = out_stride;
  // the pool holds k / k_scale and v / v_scale (set_kv_buffer, memory/pool.py:401-412): q.k scales
  // with k_scale, which joins the softmax scale; the output scales with v_scale
  a.kv_stride = kv_buffer_stride; a.sm_scale = sm_scale * k_scale; a.logit_cap = logit_cap;
  a.out_scale = v_scale;
  a.max_len = (int)max_seq_len;
  a.chunk = chunk; a.num_splits = (int)S; a.max_slots = (int)max_slots; a.plan = plan; a.kv8 = kv8 ? 1 : 0;
  a.hh_shift = decode_heads_per_load_shift(num_kv_heads, head_dim, dtype, &a.head_groups);
  a.part_o = nullptr; a.part_lse = nullptr;
  {
    // bytes of K + V one key row costs this launch (all its kv heads)
    const int64_t key_bytes = 2LL * num_kv_heads * head_dim * (kv8 ? 1 : eb);
    a.nt_min_keys = g_decode_nt_min_mb < 0 ? 0x7fffffff
                                           : (int)(((int64_t)g_decode_nt_min_mb << 20) / key_bytes);   // (< 2^31: mb is an int)
  }
  // The range geometry (DecodeArgs::rplan) where the plan carries it and the range kernel takes the launch: the default
  // configuration (16-bit pool, always-streaming gathers, no soft-cap) on a shape sp_decode_ranges() accepts.  Everything
  // else uses the plan's (request, split) items.
  a.rplan = nullptr; a.ranges = 0;
  if (plan && ranges > 0 && g_decode_ranges != 0 && logit_cap <= 0.f && a.nt_min_keys == 0 &&
      range_line_fits(batch_size, max_seq_len) && out_stride % 4 == 0 &&
      sp_decode_ranges(num_q_heads, num_kv_heads, head_dim, dtype, kv_dtype) > 0) {
    a.rplan = plan + plan_item_words(batch_size, max_slots);
    a.ranges = ranges;
    max_slots = (int64_t)batch_size + ranges;       // partial slots of the range geometry: slot = request + piece
    a.max_slots = (int)max_slots;
  }
  // a plan built without the item section (max_slots = 0) serves range launches only: this one is not (soft-cap, fp32, a
  // shape or alignment the range kernel refuses, sp_debug_set("decode_ranges", 0)) - the caller must plan the items too
  if (!a.rplan && max_slots <= 0) return SP_ERR_INVALID_ARG;
  if (S > 1 || a.rplan) {       // (a range launch splits a request wherever a piece ends)
    const size_t need = sp_decode_attention_workspace_bytes(max_slots, num_q_heads, head_dim);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return SP_ERR_WORKSPACE;
    a.part_o = (float*)workspace;
    a.part_lse = a.part_o + (size_t)max_slots * num_q_heads * head_dim;
  }
  if (max_slots * a.head_groups > 0x7fffffffLL) return SP_ERR_INVALID_ARG;
  if (dtype != SP_F32 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  return run_decode(a, head_dim, Gk, dtype, (hipStream_t)stream);
}
