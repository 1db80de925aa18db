// Paged decode attention on the matrix cores, for wide GQA groups (16-bit dtypes, G <= 16).
//
// The VALU kernel (decode_attention.hip) keeps G x 8 fp32 accumulators and G packed q fragments per
// lane; at G = 8 that is 256 VGPRs (one wave per SIMD) and twice the vector work per KV byte, and it
// runs at 2-3 TB/s.  Here the G query heads of one KV head are the COLUMNS of a 16-wide MFMA tile, so
// the per-lane state is 32 accumulator registers whatever G is, and the vector unit only does the
// 4-values-per-lane softmax:
//
//   * workgroup = 4 waves = (request, split, ONE kv head); each wave walks its share of the split in
//     tiles of 16 keys;
//   * K/V rows are gathered exactly as in the VALU kernel - full 256-B lines, 16 B per lane, four rows
//     per wave-load, next tile's loads issued before the current tile is consumed - then written to a
//     wave-private 8 KiB LDS tile (16-B chunks XOR-swizzled by row, so the fragment reads below are
//     conflict-free); no barrier: a wave's own LDS accesses complete in order;
//   * S^T[16 keys x 16 cols] = K . Q^T with v_mfma_f32_16x16x32 (A = K rows by ds_read_b128, B = Q
//     fragments held in registers); the accumulator has the head column on lane&15 and 4 keys in
//     registers, which is exactly the B-operand layout of v_mfma_f32_16x16x16 (k = 4*(lane>>4)+j), so
//     O^T[16 d x 16 cols] += V^T . P^T takes P straight from registers; the V^T fragment is one
//     ds_read_b64_tr_b16 (hardware transpose) per 16-d block;
//   * online softmax per column: 4 in-lane values + two lane exchanges (xor 16, xor 32); the rescale
//     factor of O^T is lane-local;
//   * the 4 waves' (m, l, O) are merged through LDS (the tile area is reused), split partials go to
//     the same workspace / merge kernel as the VALU path.
#include <type_traits>

#include "attention_internal.h"

namespace sp {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_m __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_m __attribute__((ext_vector_type(8)));
typedef short s16x4_m __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_m __attribute__((ext_vector_type(4)));

static constexpr float kLog2eM = 1.4426950408889634f;
static constexpr float kNegBigM = -1.0e30f;

template <typename Tag>
__device__ __forceinline__ f32x4_t mfma_qk(const u32x4& a, const u32x4& b, const f32x4_t& c);
template <>
__device__ __forceinline__ f32x4_t mfma_qk<bf16_tag>(const u32x4& a, const u32x4& b, const f32x4_t& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_m, a),
                                                 __builtin_bit_cast(bf16x8_m, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma_qk<f16_tag>(const u32x4& a, const u32x4& b, const f32x4_t& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_m, a),
                                                __builtin_bit_cast(f16x8_m, b), c, 0, 0, 0);
}
template <typename Tag>
__device__ __forceinline__ f32x4_t mfma_pv(const u32x2& a, const u32x2& b, const f32x4_t& c);
template <>
__device__ __forceinline__ f32x4_t mfma_pv<bf16_tag>(const u32x2& a, const u32x2& b, const f32x4_t& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_m, a),
                                                   __builtin_bit_cast(s16x4_m, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma_pv<f16_tag>(const u32x2& a, const u32x2& b, const f32x4_t& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_m, a),
                                               __builtin_bit_cast(f16x4_m, b), c, 0, 0, 0);
}

template <typename Tag>
__device__ __forceinline__ uint32_t pack2m(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2m<bf16_tag>(float lo, float hi) {
  bf16x2_t b;
  b[0] = (__bf16)lo;
  b[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, b);
}
template <>
__device__ __forceinline__ uint32_t pack2m<f16_tag>(float lo, float hi) {
  f16x2_t b;
  b[0] = (_Float16)lo;
  b[1] = (_Float16)hi;
  return __builtin_bit_cast(uint32_t, b);
}

// values held by lane ^ 16 / lane ^ 32 through the gfx950 half-row / half-wave swaps (VALU) instead
// of ds_bpermute round trips: v_permlane16_swap exchanges the odd 16-lane rows of its first operand
// with the even rows of its second, v_permlane32_swap lanes 32-63 of the first with 0-31 of the second
__device__ __forceinline__ float xchg16m(float x) {
  const uint32_t u = as_u32(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const uint32_t odd_gets = r[0], even_gets = r[1];
  return as_f32((threadIdx.x & 16) ? odd_gets : even_gets);
}
__device__ __forceinline__ float xchg32m(float x) {
  const uint32_t u = as_u32(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const uint32_t upper_gets = r[0], lower_gets = r[1];
  return as_f32((threadIdx.x & 32) ? upper_gets : lower_gets);
}

template <int D>
struct DmCfg {
  static constexpr int TK = 16;                    // keys per tile
  static constexpr int ROW_B = D * 2;              // bytes per K/V row
  static constexpr int CPR = ROW_B / 16;           // 16-byte chunks per row (16 / 8)
  static constexpr int RPL = 64 / CPR;             // rows per wave-load (4 / 8)
  static constexpr int NLD = TK / RPL;             // wave-loads per tile, each for K and for V (4 / 2)
  static constexpr int KSTEPS = D / 32;            // 16x16x32 k-steps of Q.K^T
  static constexpr int DBLK = D / 16;              // 16-wide d blocks of O^T
  static constexpr int TILE_B = TK * ROW_B;        // one K (or V) tile
  static constexpr int WAVES = 4;
  static constexpr int kStageBytes = WAVES * 2 * TILE_B;
  static constexpr int kMergeBytes = WAVES * 16 * (D + 2) * 4;
  static constexpr int kLdsBytes = kStageBytes > kMergeBytes ? kStageBytes : kMergeBytes;
};
// persistent form: LDS per wave in which split partials wait for the end of the wave's work (with the 32 KiB of tiles
// at D = 128: 49 KiB per workgroup, three workgroups per CU)
static constexpr int kDmParkWaveB = 4352;

// HPW ("head per wave", Hkv % 4 == 0): the 4 waves take the 4 adjacent KV heads of the SAME keys - the
// workgroup then reads whole 1 KiB token half-rows, and each wave owns its heads outright: no merge,
// no barrier anywhere.  Otherwise (few KV heads per rank) the waves split the keys of one head.
//
// KV8: the pool holds fp8 e5m2 bytes (--kv-cache-dtype fp8_e5m2).  A lane gathers 8 bytes instead of
// 16, expands them to 8 halves on the way into the LDS tile (e5m2 is the top byte of a half: one
// v_perm_b32 per two elements, exact), and the tile math runs in fp16 whatever the model dtype is
// (bf16 q is converted once per workgroup; P is rounded to fp16); the output keeps the model dtype.
// HBM bytes per context token halve; everything after the LDS tile is unchanged.
template <typename Tag, int D, bool HPW, bool KV8>
#ifndef SP_DEC_WAVES
#define SP_DEC_WAVES 3
#endif
__global__ __launch_bounds__(256, SP_DEC_WAVES) void decode_mfma_kernel(DecodeArgs a) {
  typedef DmCfg<D> C;
  typedef Elem<Tag> E;
  typedef typename std::conditional<KV8, f16_tag, Tag>::type CT;      // dtype of the tile math
  typedef typename std::conditional<KV8, u32x2, u32x4>::type raw_t;   // one lane's gathered chunk
  constexpr int SRC_ROW_B = KV8 ? D : 2 * D;                          // bytes of one head row in the pool
  constexpr int SRC_CH_B = KV8 ? 8 : 16;                              // bytes of the 8 elements a lane gathers
  constexpr int TK = C::TK, ROW_B = C::ROW_B, CPR = C::CPR, RPL = C::RPL, NLD = C::NLD;
  constexpr int KSTEPS = C::KSTEPS, DBLK = C::DBLK, TILE_B = C::TILE_B, WAVES = C::WAVES;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  // blockIdx -> (item, kv head or head quad); the heads of one token row are adjacent in launch order
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hgroups = HPW ? a.Hkv / 4 : a.Hkv;
  const int hk = HPW ? (blockIdx.x % hgroups) * 4 + wave : blockIdx.x % hgroups;
  const int item = blockIdx.x / hgroups;
  int b, c, chunk, slot0;
  if (!decode_item(a, item, b, c, chunk, slot0)) return;
  const int seq = min((int)load_idx(a.seq_lens, b, a.idx64), a.max_len);
  const int cs = c * chunk;
  if (cs >= seq || slot0 + c >= a.max_slots) return;
  const int ce = min(cs + chunk, seq);
  const int nsplit = (seq + chunk - 1) / chunk;
  const int64_t req = load_idx(a.req_idx, b, a.idx64);
  const int64_t kv0 = a.kv_start ? load_idx(a.kv_start, b, a.idx64) : 0;
  const int32_t* idx_row = a.r2t + req * a.r2t_stride + kv0;
  const int G = a.Hq / a.Hkv;

  const int col = lane & 15, kq = lane >> 4;       // MFMA column (query head) / k quarter
  char* ldsK = lds + wave * 2 * TILE_B;
  char* ldsV = ldsK + TILE_B;

  // Q fragments: B operand of S^T = K.Q^T; lane (col, kq) holds Q[head col][32s + 8kq .. +7]
  u32x4 qf[KSTEPS];
  {
    const int hcol = min(col, G - 1);              // padding columns repeat the last head; never stored
    const char* qp = (const char*)a.q +
                     ((int64_t)b * a.q_stride + (int64_t)(hk * G + hcol) * D + 8 * kq) * 2;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      qf[s] = ld16(qp + s * 64);
      if constexpr (KV8 && std::is_same<Tag, bf16_tag>::value) qf[s] = bf16x8_to_f16x8(qf[s]);
    }
  }
  const float cap = a.logit_cap;
  const float qk_scale = cap > 0.f ? a.sm_scale : a.sm_scale * kLog2eM;

  f32x4_t oacc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = kNegBigM, l_run = 0.f;

  // gather mapping: lane -> (row within the wave-load, 16-byte chunk)
  const int ld_row = lane / CPR, ld_ch = lane % CPR;
  // Address of the chunk a lane gathers from K row `slot`:  kbase[i] + slot * tok_bytes, ONE v_mad_u64_u32 (the slot is a
  // non-negative int32 and the token stride fits 32 bits - the host checks it - so no 64 x 64-bit product, which the
  // compiler expands into three quarter-rate multiplies: PMC, round 5: the address arithmetic was 40 % of the loop's
  // vector instructions).  kbase[i] holds everything that does not depend on the key: pool base, head, and the source
  // chunk, XOR-swizzled by the tile row so that the LDS image is conflict-free.  V sits at a launch-uniform distance
  // from K (the pool's two views; one add).
  const uint32_t tok_bytes = (uint32_t)(a.kv_stride * (KV8 ? 1 : 2));
  const int64_t v_minus_k = a.vbuf - a.kbuf;
  uint64_t kbase[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int R = i * RPL + ld_row;
    kbase[i] = (uint64_t)(uintptr_t)a.kbuf + (uint64_t)hk * SRC_ROW_B + (uint64_t)((ld_ch ^ (R & (CPR - 1))) * SRC_CH_B);
  }
  // fragment-read addresses (constant per lane)
  const int i16 = lane & 15;
  const int tr_row = 4 * kq + (i16 >> 2);          // V row this lane addresses in a tr read

  // this wave's keys: the whole split (HPW) or a contiguous share of it in whole tiles
  const int sub = HPW ? ce - cs : ((ce - cs + WAVES * TK - 1) / (WAVES * TK)) * TK;
  const int ws = HPW ? cs : cs + wave * sub;
  const int we = min(ws + sub, ce);

  // The key loop exists in four copies: with plain and with NON-TEMPORAL K/V gathers (DecodeArgs::nt_min_keys; the
  // choice is uniform over the launch - plan data), with and without the logit soft-cap (a launch argument) - each
  // choice made once, here, so that every form is a loop of its own without a branch per score.
  auto key_loop = [&](auto nt_c, auto cap_c) {
  constexpr bool NT = decltype(nt_c)::value;
  constexpr bool CAP = decltype(cap_c)::value;
  int nextidx = (ws + lane < we) ? idx_row[ws + lane] : 0;
  for (int ps = ws; ps < we; ps += 64) {           // pieces of <= 64 keys: one index register
    const int n = min(64, we - ps);
    const int myidx = nextidx;
    nextidx = (ps + 64 + lane < we) ? idx_row[ps + 64 + lane] : 0;
    const int ntile = (n + TK - 1) / TK;

    raw_t kr[NLD], vr[NLD];
    raw_t kr2[NLD], vr2[NLD];   // KV8 only: a second set (see the loop below); unused and optimised away otherwise
    auto issue_from = [&](int tile, int idxreg, raw_t (&kd)[NLD], raw_t (&vd)[NLD]) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int key = tile * TK + i * RPL + ld_row;         // key within the piece
        // (lanes of the index register past the piece's keys hold slot 0, the pool's dummy row: no select needed)
        const uint32_t slot = (uint32_t)__shfl(idxreg, key & 63, 64);
        // (an integer turned into a pointer is a FLAT pointer to the compiler - flat loads also count in lgkmcnt and
        // take the aperture check: name the global address space)
        typedef const raw_t __attribute__((address_space(1)))* gptr_t;
        const uint64_t ka = kbase[i] + (uint64_t)slot * tok_bytes;
        const gptr_t kp = (gptr_t)ka, vp = (gptr_t)(ka + (uint64_t)v_minus_k);
        if constexpr (NT) {
          kd[i] = __builtin_nontemporal_load(kp);
          vd[i] = __builtin_nontemporal_load(vp);
        } else {
          kd[i] = *kp;
          vd[i] = *vp;
        }
      }
    };
    auto issue_to = [&](int tile, raw_t (&kd)[NLD], raw_t (&vd)[NLD]) { issue_from(tile, myidx, kd, vd); };
    auto issue = [&](int tile) { issue_to(tile, kr, vr); };
    auto stage_from = [&](const raw_t (&ks)[NLD], const raw_t (&vs)[NLD]) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int R = i * RPL + ld_row;
        if constexpr (KV8) {
          st16(ldsK + R * ROW_B + ld_ch * 16, expand_e5m2x8(ks[i]));
          st16(ldsV + R * ROW_B + ld_ch * 16, expand_e5m2x8(vs[i]));
        } else {
          st16(ldsK + R * ROW_B + ld_ch * 16, ks[i]);
          st16(ldsV + R * ROW_B + ld_ch * 16, vs[i]);
        }
      }
    };
    auto stage = [&]() {
      // registers -> this wave's LDS tile: position (row R, chunk ld_ch) holds source chunk
      // ld_ch ^ R, i.e. logical chunk cg of row R sits at chunk cg ^ R
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int R = i * RPL + ld_row;
        if constexpr (KV8) {
          st16(ldsK + R * ROW_B + ld_ch * 16, expand_e5m2x8(kr[i]));
          st16(ldsV + R * ROW_B + ld_ch * 16, expand_e5m2x8(vr[i]));
        } else {
          st16(ldsK + R * ROW_B + ld_ch * 16, kr[i]);
          st16(ldsV + R * ROW_B + ld_ch * 16, vr[i]);
        }
      }
    };
    auto consume = [&](int tile) {
      // ---- S^T = K . Q^T: lane (key = col index of A rows = lane&15, kq)
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
      {
        const int R = lane & 15;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const int cg = 4 * ks + kq;                          // logical chunk: dims 32ks + 8kq ..
          const u32x4 kf = ld16(ldsK + R * ROW_B + ((cg ^ (R & (CPR - 1))) * 16));
          s = mfma_qk<CT>(kf, qf[ks], s);
        }
      }
      // ---- scale, mask, online softmax for column `col`; this lane holds keys 4kq + j
      float x[4], mx = kNegBigM;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = s[j] * qk_scale;
        if constexpr (CAP) v = cap * tanhf(v / cap) * kLog2eM;
        x[j] = (tile * TK + 4 * kq + j < n) ? v : -INFINITY;
        mx = fmaxf(mx, x[j]);
      }
      mx = fmaxf(mx, xchg16m(mx));
      mx = fmaxf(mx, xchg32m(mx));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = __builtin_amdgcn_exp2f(x[j] - m_new);
        psum += x[j];
      }
      psum += xchg16m(psum);
      psum += xchg32m(psum);
      l_run = l_run * alpha + psum;
      u32x2 pf;  // B operand of O^T += V^T.P^T: P^T[k = 4kq + j][col]
      pf[0] = pack2m<CT>(x[0], x[1]);
      pf[1] = pack2m<CT>(x[2], x[3]);
      // ---- O^T[16 d x 16 cols] per d block; V^T fragment by one transposed read:
      //      lane i of a 16-lane group addresses row 4kq + (i>>2), elements 4(i&3)..+3 of the block
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        const int cg = 2 * db + ((i16 & 3) >> 1);             // logical 16-byte chunk of the row
        const char* vp = ldsV + tr_row * ROW_B + ((cg ^ (tr_row & (CPR - 1))) * 16) + 8 * (i16 & 1);
        const s16x4_m vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_m*)vp);
        const u32x2 vf = __builtin_bit_cast(u32x2, vt);
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[db][r] *= alpha;
        oacc[db] = mfma_pv<CT>(vf, pf, oacc[db]);
      }
    };

    // one register set: the tile's registers are free as soon as they are in LDS, so the next
    // tile's gathers are issued right there and fly during this tile's MFMAs and softmax (and under
    // the other waves of the SIMD: ~100 VGPRs => 4 waves per SIMD)
#ifdef SP_DEC_ONESET   // diagnostic build: one register set for byte pools too
    constexpr bool kTwoSets = false;
#elif defined(SP_DEC_TWOSETS)   // diagnostic build: two register sets for 16-bit pools too (with -DSP_DEC_WAVES=2: no spills)
    constexpr bool kTwoSets = true;
#else
    constexpr bool kTwoSets = KV8;
#endif
    if constexpr (kTwoSets) {
      // a byte pool moves half the bytes per gather: with one tile in flight per wave the CU has half the
      // bytes in flight of the 16-bit kernel (5.1 vs 5.7 TB/s).  Two register sets (8 B per lane and gather:
      // 16 registers more), tiles t+1 and t+2 in flight while tile t is consumed.
      issue_to(0, kr, vr);
      if (1 < ntile) issue_to(1, kr2, vr2);
      for (int t = 0; t < ntile; t += 2) {
        stage_from(kr, vr);
        if (t + 2 < ntile) issue_to(t + 2, kr, vr);
        consume(t);
        if (t + 1 < ntile) {
          stage_from(kr2, vr2);
          if (t + 3 < ntile) issue_to(t + 3, kr2, vr2);
          consume(t + 1);
        }
      }
    } else {
    issue(0);
    for (int t = 0; t < ntile; ++t) {
      stage();
      if (t + 1 < ntile) issue(t + 1);
      consume(t);
    }
    }
  }
  };
  // (a plan-less launch has no key count to compare: it streams only when the threshold is "always", the default)
  const bool nt = a.plan ? a.plan[3] >= a.nt_min_keys : a.nt_min_keys == 0;
  if (cap > 0.f) {
    if (nt) key_loop(std::true_type{}, std::true_type{}); else key_loop(std::false_type{}, std::true_type{});
  } else {
    if (nt) key_loop(std::true_type{}, std::false_type{}); else key_loop(std::false_type{}, std::false_type{});
  }

  if constexpr (HPW) {
    // ---- this wave owns heads hk*G .. hk*G+G-1: normalise and write straight from registers
    //      (lane (col, kq) holds d = 16db + 4kq .. +3 of head col)
    if (col < G) {
      const int h = hk * G + col;
      float inv = 1.0f / l_run;
      if (nsplit == 1) {
        inv *= a.out_scale;
        char* op = (char*)a.out + ((int64_t)b * a.o_stride + (int64_t)h * D + 4 * kq) * 2;
#pragma unroll
        for (int db = 0; db < DBLK; ++db) {
          u32x2 w;
          w[0] = pack2m<Tag>(oacc[db][0] * inv, oacc[db][1] * inv);
          w[1] = pack2m<Tag>(oacc[db][2] * inv, oacc[db][3] * inv);
          *(u32x2*)(op + db * 32) = w;
        }
      } else {
        const int64_t pi = (int64_t)h * a.max_slots + (slot0 + c);
        float* pp = a.part_o + pi * D + 4 * kq;
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
          *(float4*)(pp + 16 * db) = make_float4(oacc[db][0] * inv, oacc[db][1] * inv,
                                                 oacc[db][2] * inv, oacc[db][3] * inv);
        if (kq == 0) a.part_lse[pi] = m_run + __builtin_amdgcn_logf(l_run);
      }
    }
    return;
  }
  // ---- merge the 4 waves through LDS (reusing the tile area): O^T[d][col], m, l per column
  __syncthreads();  // every wave is done with its tiles
  float* sm_o = (float*)lds;                       // [WAVES][16 cols][D]
  float* sm_ml = sm_o + WAVES * 16 * D;            // [WAVES][16 cols][2]
  {
    float* dst = sm_o + (wave * 16 + col) * D + 4 * kq;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
      *(float4*)(dst + 16 * db) = make_float4(oacc[db][0], oacc[db][1], oacc[db][2], oacc[db][3]);
    if (kq == 0) {
      sm_ml[(wave * 16 + col) * 2] = m_run;
      sm_ml[(wave * 16 + col) * 2 + 1] = l_run;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G * D; i += 256) {
    const int d = i % D, g = i / D;
    float M = kNegBigM;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) M = fmaxf(M, sm_ml[(w * 16 + g) * 2]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float wgt = __builtin_amdgcn_exp2f(sm_ml[(w * 16 + g) * 2] - M);
      L += sm_ml[(w * 16 + g) * 2 + 1] * wgt;
      O += sm_o[(w * 16 + g) * D + d] * wgt;
    }
    const int h = hk * G + g;
    const float o = O / L;
    if (nsplit == 1) {
      E::store(a.out, (int64_t)b * a.o_stride + (int64_t)h * D + d, o * a.out_scale);
    } else {
      const int64_t pi = (int64_t)h * a.max_slots + (slot0 + c);
      a.part_o[pi * D + d] = o;
      if (d == 0) a.part_lse[pi] = M + __builtin_amdgcn_logf(L);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the head-per-wave kernel (round 5): planned launches of the default configuration (16-bit pool,
// non-temporal gathers, no soft-cap).  The launch has as many workgroups as the chip holds at once, and each takes
// (plan item, head quad) units one after the other WITHOUT draining its gathers in between: while a unit's last piece of
// 64 keys is consumed, the next unit's record and first index piece are fetched, and its first tile's gathers are issued
// where the current unit has none left to issue.  Which units a workgroup takes is STATIC - unit r * W + w in even rounds,
// r * W + (W - 1 - w) in odd ones: dealt in serpentine order, the plan's longest-first list gives every workgroup the
// load greedy longest-processing-time scheduling would (tools/sim_decode_split.py), whereas tickets drawn from one
// counter cost more than they balance (same-address device-scope atomics serialise at ~50 ns apiece).  The four waves of
// a workgroup compute the same sequence and take the four heads of each unit; they share nothing (wave-private LDS
// tile, no barrier).
//
// What this form buys is WHERE the split partials go (profiles/r05_decode_split_cost.txt): fp32 partial rows written
// to HBM in between the gathers cost ~1.8 us per MB - ten times their share of the bytes.  A wave PARKS the partial
// rows of its units in LDS the tiles do not use (kParkWaveB per wave: 2 units at G = 4, D = 128) and writes them, in
// whole 512-byte rows, when it has no unit left - which, the loads being level, is when the launch as a whole is
// running out of gathers.  (Holding the rows back until EVERY workgroup is done, behind a bounded device-wide barrier,
// was tried as well: no further gain, profiles/NOTES.md round 5.)  Units past the parking space, groups too wide to
// park, and unsplit requests' output rows are stored at once, as in the launch-per-item kernel.
// Same arithmetic, same order of the keys within a unit, same bits as the launch-per-item kernel.
template <typename Tag, int D>
__global__ __launch_bounds__(256, SP_DEC_WAVES) void decode_mfma_pw_kernel(DecodeArgs a) {
  typedef DmCfg<D> C;
  typedef u32x4 raw_t;
  constexpr int SRC_ROW_B = 2 * D, SRC_CH_B = 16;
  constexpr int TK = C::TK, ROW_B = C::ROW_B, CPR = C::CPR, RPL = C::RPL, NLD = C::NLD;
  constexpr int KSTEPS = C::KSTEPS, DBLK = C::DBLK, TILE_B = C::TILE_B;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int G = a.Hq / a.Hkv;
  const int col = lane & 15, kq = lane >> 4;
  char* ldsK = lds + wave * 2 * TILE_B;
  char* ldsV = ldsK + TILE_B;
  const float qk_scale = a.sm_scale * kLog2eM;
  const int ld_row = lane / CPR, ld_ch = lane % CPR;
  const uint32_t tok_bytes = (uint32_t)(a.kv_stride * 2);
  const int64_t v_minus_k = a.vbuf - a.kbuf;
  const int i16 = lane & 15;
  const int tr_row = 4 * kq + (i16 >> 2);

  const int hgroups = a.Hkv / 4;
  const int nunits = a.plan[0] * hgroups;            // (plan item, head quad) units, in the plan's longest-first order
  const int chunk = a.plan[1];
  const int32_t* items = a.plan + kPlanHdr + a.bs;
  const int W = (int)gridDim.x, w = (int)blockIdx.x;

  struct Unit { int b, c, seq, cs, ce, nsplit, slot0, hk; const int32_t* idx_row; };
  // A unit's record is read with SCALAR loads written as asm: inside the loop hipcc turns these reads into vector loads
  // (the kernel's own stores may alias them as far as it knows), each followed by s_waitcnt vmcnt(0) - drained round
  // trips per unit with no gather in flight.  A scalar load waits on lgkmcnt only: the tile in flight stays in flight.
  // Two dependent rounds: the item's (b, c), then everything indexed by b at once.
  auto load_item = [&](int item, int& b, int& c) {
    int64_t bc;
    const int64_t* q = (const int64_t*)items + item;       // (b, c) as one 8-byte record (items is 8-byte aligned:
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bc) : "s"(q) : "memory");   // bs + 4 words in)
    b = (int)(uint32_t)bc;
    c = (int)(bc >> 32);
  };
  const void* kv_or_seq = a.kv_start ? a.kv_start : a.seq_lens;   // (a valid address either way)
  auto load_request = [&](int b, int& slot0, int64_t& seq, int64_t& req, int64_t& kv0) {
    const int32_t* ps = a.plan + kPlanHdr + b;
    if (a.idx64) {
      const int64_t *p1 = (const int64_t*)a.seq_lens + b, *p2 = (const int64_t*)a.req_idx + b, *p3 = (const int64_t*)kv_or_seq + b;
      asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dwordx2 %1, %5, 0x0\n\ts_load_dwordx2 %2, %6, 0x0\n\t"
                   "s_load_dwordx2 %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(slot0), "=&s"(seq), "=&s"(req), "=&s"(kv0) : "s"(ps), "s"(p1), "s"(p2), "s"(p3) : "memory");
    } else {
      const int32_t *p1 = (const int32_t*)a.seq_lens + b, *p2 = (const int32_t*)a.req_idx + b, *p3 = (const int32_t*)kv_or_seq + b;
      int s32, r32, k32;
      asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\t"
                   "s_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(slot0), "=&s"(s32), "=&s"(r32), "=&s"(k32) : "s"(ps), "s"(p1), "s"(p2), "s"(p3) : "memory");
      seq = s32; req = r32; kv0 = k32;
    }
    if (!a.kv_start) kv0 = 0;
  };
  int round = 0;
  // this workgroup's unit of the next round (serpentine dealing), skipping empty ones (never listed by sp_decode_plan;
  // a cut plan); false when its list is exhausted
  auto next_unit = [&](Unit& u) -> bool {
    for (;;) {
      const int T = __builtin_amdgcn_readfirstlane(round * W + ((round & 1) ? W - 1 - w : w));
      ++round;
      if (T >= nunits) return false;
      const int item = T / hgroups;
      u.hk = (T - item * hgroups) * 4 + wave;
      load_item(item, u.b, u.c);
      int64_t seq, req, kv0;
      load_request(u.b, u.slot0, seq, req, kv0);
      u.seq = (int)min(seq, (int64_t)a.max_len);
      u.cs = u.c * chunk;
      u.ce = min(u.cs + chunk, u.seq);
      u.nsplit = (u.seq + chunk - 1) / chunk;
      u.idx_row = a.r2t + req * a.r2t_stride + kv0;
      if (u.cs < u.seq && u.slot0 + u.c < a.max_slots) return true;
    }
  };

  Unit cur;
  if (!next_unit(cur)) return;

  uint64_t kbase[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int R = i * RPL + ld_row;
    kbase[i] = (uint64_t)(uintptr_t)a.kbuf + (uint64_t)cur.hk * SRC_ROW_B + (uint64_t)((ld_ch ^ (R & (CPR - 1))) * SRC_CH_B);
  }
  u32x4 qf[KSTEPS];
  auto load_q = [&](const Unit& u) {
    const int hcol = min(col, G - 1);
    const char* qp = (const char*)a.q + ((int64_t)u.b * a.q_stride + (int64_t)(u.hk * G + hcol) * D + 8 * kq) * 2;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qf[s] = ld16(qp + s * 64);
  };
  load_q(cur);

  f32x4_t oacc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = kNegBigM, l_run = 0.f;

  raw_t kr[NLD], vr[NLD];
  auto issue = [&](int tile, int idxreg) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int key = tile * TK + i * RPL + ld_row;
      const uint32_t slot = (uint32_t)__shfl(idxreg, key & 63, 64);
      typedef const raw_t __attribute__((address_space(1)))* gptr_t;
      const uint64_t ka = kbase[i] + (uint64_t)slot * tok_bytes;
      kr[i] = __builtin_nontemporal_load((gptr_t)ka);
      vr[i] = __builtin_nontemporal_load((gptr_t)(ka + (uint64_t)v_minus_k));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int R = i * RPL + ld_row;
      st16(ldsK + R * ROW_B + ld_ch * 16, kr[i]);
      st16(ldsV + R * ROW_B + ld_ch * 16, vr[i]);
    }
  };
  auto consume = [&](int tile, int n) {
    f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
      const int R = lane & 15;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int cg = 4 * ks + kq;
        const u32x4 kf = ld16(ldsK + R * ROW_B + ((cg ^ (R & (CPR - 1))) * 16));
        s = mfma_qk<Tag>(kf, qf[ks], s);
      }
    }
    float x[4], mx = kNegBigM;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = s[j] * qk_scale;
      x[j] = (tile * TK + 4 * kq + j < n) ? v : -INFINITY;
      mx = fmaxf(mx, x[j]);
    }
    mx = fmaxf(mx, xchg16m(mx));
    mx = fmaxf(mx, xchg32m(mx));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = __builtin_amdgcn_exp2f(x[j] - m_new);
      psum += x[j];
    }
    psum += xchg16m(psum);
    psum += xchg32m(psum);
    l_run = l_run * alpha + psum;
    u32x2 pf;
    pf[0] = pack2m<Tag>(x[0], x[1]);
    pf[1] = pack2m<Tag>(x[2], x[3]);
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
      const int cg = 2 * db + ((i16 & 3) >> 1);
      const char* vp = ldsV + tr_row * ROW_B + ((cg ^ (tr_row & (CPR - 1))) * 16) + 8 * (i16 & 1);
      const s16x4_m vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_m*)vp);
      const u32x2 vf = __builtin_bit_cast(u32x2, vt);
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[db][r] *= alpha;
      oacc[db] = mfma_pv<Tag>(vf, pf, oacc[db]);
    }
  };
  // Parking space of this wave behind the tiles: records of [G rows of D floats | 16 log-sum-exps | kv head, slot]
  const int park_rows_b = G * D * 4;
  const int park_unit_b = park_rows_b + 64 + 16;
  const int park_cap = kDmParkWaveB / park_unit_b;          // 2 at G = 4, D = 128; 0: the group is too wide to park
  char* park = lds + 4 * 2 * TILE_B + wave * kDmParkWaveB;
  int parked = 0;
  auto flush_parked = [&]() {
    for (int k = 0; k < parked; ++k) {
      const char* src = park + k * park_unit_b;
      const int hk_ = __builtin_amdgcn_readfirstlane(*(const int*)(src + park_rows_b + 64));
      const int slot_ = __builtin_amdgcn_readfirstlane(*(const int*)(src + park_rows_b + 68));
      for (int off = lane * 16; off < park_rows_b; off += 1024) {     // whole rows, 16 B per lane
        const int row = off / (D * 4), within = off - row * (D * 4);
        const int64_t pi = (int64_t)(hk_ * G + row) * a.max_slots + slot_;
        *(u32x4*)((char*)(a.part_o + pi * D) + within) = ld16(src + off);
      }
      if (lane < G) a.part_lse[(int64_t)(hk_ * G + lane) * a.max_slots + slot_] = *(const float*)(src + park_rows_b + lane * 4);
    }
  };
  // a unit's result: straight to the output (an unsplit request) or as a partial for the merge launch
  auto write_unit = [&](const Unit& u) {
    if (u.nsplit > 1 && parked < park_cap) {
      char* dst = park + parked * park_unit_b;
      if (col < G) {
        const float inv = 1.0f / l_run;
#pragma unroll
        for (int db = 0; db < DBLK; ++db) {
          u32x4 w;
          w[0] = as_u32(oacc[db][0] * inv); w[1] = as_u32(oacc[db][1] * inv);
          w[2] = as_u32(oacc[db][2] * inv); w[3] = as_u32(oacc[db][3] * inv);
          st16(dst + col * (D * 4) + (16 * db + 4 * kq) * 4, w);
        }
        if (kq == 0) *(float*)(dst + park_rows_b + col * 4) = m_run + __builtin_amdgcn_logf(l_run);
      }
      if (lane == 0) {
        *(int*)(dst + park_rows_b + 64) = u.hk;
        *(int*)(dst + park_rows_b + 68) = u.slot0 + u.c;
      }
      ++parked;
      return;
    }
    if (col >= G) return;
    const int h = u.hk * G + col;
    float inv = 1.0f / l_run;
    if (u.nsplit == 1) {
      inv *= a.out_scale;
      char* op = (char*)a.out + ((int64_t)u.b * a.o_stride + (int64_t)h * D + 4 * kq) * 2;
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        u32x2 w;
        w[0] = pack2m<Tag>(oacc[db][0] * inv, oacc[db][1] * inv);
        w[1] = pack2m<Tag>(oacc[db][2] * inv, oacc[db][3] * inv);
        *(u32x2*)(op + db * 32) = w;
      }
    } else {
      const int64_t pi = (int64_t)h * a.max_slots + (u.slot0 + u.c);
      float* pp = a.part_o + pi * D + 4 * kq;
#pragma unroll
      for (int db = 0; db < DBLK; ++db)
        *(float4*)(pp + 16 * db) = make_float4(oacc[db][0] * inv, oacc[db][1] * inv, oacc[db][2] * inv, oacc[db][3] * inv);
      if (kq == 0) a.part_lse[pi] = m_run + __builtin_amdgcn_logf(l_run);
    }
  };

  int pos = cur.cs;
  int curidx = (pos + lane < cur.ce) ? cur.idx_row[pos + lane] : 0;
  issue(0, curidx);
  for (;;) {
    const int n = min(64, cur.ce - pos);
    const int ntile = (n + TK - 1) / TK;
    const bool last_piece = pos + 64 >= cur.ce;
    Unit nxt;
    bool have_next = false;
    int nextidx = 0;
    if (last_piece) {
      have_next = next_unit(nxt);
      if (have_next) nextidx = (nxt.cs + lane < nxt.ce) ? nxt.idx_row[nxt.cs + lane] : 0;
    } else {
      nextidx = (pos + 64 + lane < cur.ce) ? cur.idx_row[pos + 64 + lane] : 0;
    }
    for (int t = 0; t < ntile; ++t) {
      stage();
      if (t + 1 < ntile) {
        issue(t + 1, curidx);
      } else if (!last_piece) {
        issue(0, nextidx);                          // the next piece's first tile: no drain between pieces
      } else if (have_next) {                       // the next UNIT's first tile: no drain between units
        const int64_t dh = (int64_t)(nxt.hk - cur.hk) * SRC_ROW_B;
#pragma unroll
        for (int i = 0; i < NLD; ++i) kbase[i] += (uint64_t)dh;
        issue(0, nextidx);
      }
      consume(t, n);
    }
    if (last_piece) {
      write_unit(cur);
      if (!have_next) break;
      cur = nxt;
      load_q(cur);
#pragma unroll
      for (int db = 0; db < DBLK; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      m_run = kNegBigM;
      l_run = 0.f;
      pos = cur.cs;
    } else {
      pos += 64;
    }
    curidx = nextidx;
  }
  flush_parked();
}

template <typename Tag, int D, bool KV8>
static int launch_dm_kv(const DecodeArgs& a, hipStream_t st) {
  typedef DmCfg<D> C;
  // planned: one workgroup per (plan item, head group), the launch covers max_slots items (the surplus
  // exits at once); plan-less: the static (request, split) grid
  const int64_t items = a.plan ? (int64_t)a.max_slots : (int64_t)a.bs * a.num_splits;
  if (a.Hkv % 4 == 0 && a.o_stride % 4 == 0) {
    if constexpr (!KV8) {
      // the persistent form: planned launches of the default (always-streaming, no soft-cap) configuration
      if (a.plan && a.persist && a.logit_cap <= 0.f && a.nt_min_keys == 0) {
        constexpr int kLds = C::kStageBytes + C::WAVES * kDmParkWaveB;
        // as many workgroups as the chip holds at once (asked of the runtime once per instantiation), or the debug value
        static int resident = 0;
        if (!resident) {
          int dev = 0, per_cu = 0;
          hipDeviceProp_t prop;
          if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
              hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_mfma_pw_kernel<Tag, D>, 256, kLds) != hipSuccess ||
              per_cu < 1)
            return SP_ERR_LAUNCH;
          resident = per_cu * prop.multiProcessorCount;
        }
        // ... when the step can have more units than that: the lengths bound the items by bs x ceil(max_len / chunk)
        // as well (a graph's launch covers >= 1024 whatever the batch).  A launch of fewer units is a matter of
        // latency, not of bandwidth, and there the launch-per-item kernel is the faster one (bs 1, 1024 keys: 34 vs 41 us).
        const int64_t bound = (int64_t)a.bs * a.num_splits;
        const int64_t wgs = (items < bound ? items : bound) * (a.Hkv / 4);
        const int64_t want = a.persist > 0 ? a.persist : resident;
        if (wgs > want || a.persist > 0) {
          const unsigned grid = (unsigned)(wgs < want ? wgs : want);
          decode_mfma_pw_kernel<Tag, D><<<dim3(grid), 256, kLds, st>>>(a);
          SP_LAUNCH_CHECK();
          return SP_OK;
        }
      }
    }
    const unsigned grid = (unsigned)(items * (a.Hkv / 4));
    decode_mfma_kernel<Tag, D, true, KV8><<<dim3(grid), 256, C::kLdsBytes, st>>>(a);
  } else {
    const unsigned grid = (unsigned)(items * a.Hkv);
    decode_mfma_kernel<Tag, D, false, KV8><<<dim3(grid), 256, C::kLdsBytes, st>>>(a);
  }
  SP_LAUNCH_CHECK();
  return SP_OK;
}

template <typename Tag, int D>
static int launch_dm(const DecodeArgs& a, hipStream_t st) {
  return a.kv8 ? launch_dm_kv<Tag, D, true>(a, st) : launch_dm_kv<Tag, D, false>(a, st);
}

// 16-bit dtypes, D in {64,128}, G <= 16.  Launches the attention kernel only; the caller runs the
// split merge (shared with the VALU path).  (An in-kernel merge by arrival counters existed in round 4, ABI 6: it
// measured slower under graph replay at every batch size and was removed in round 5 - profiles/NOTES.md.)
int run_decode_mfma(const DecodeArgs& a, int head_dim, int dtype, hipStream_t st) {
  const int G = a.Hq / a.Hkv;
  if (G > 16 || (dtype != SP_BF16 && dtype != SP_F16)) return SP_ERR_UNSUPPORTED;
  if ((a.plan ? (int64_t)a.max_slots : (int64_t)a.bs * a.num_splits) * a.Hkv > 0x7fffffffLL) return SP_ERR_INVALID_ARG;
  if (head_dim == 128)
    return dtype == SP_BF16 ? launch_dm<bf16_tag, 128>(a, st) : launch_dm<f16_tag, 128>(a, st);
  if (head_dim == 64)
    return dtype == SP_BF16 ? launch_dm<bf16_tag, 64>(a, st) : launch_dm<f16_tag, 64>(a, st);
  return SP_ERR_UNSUPPORTED;
}

}  // namespace sp
