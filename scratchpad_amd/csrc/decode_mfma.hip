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
//
// Two kernels share this tile arithmetic.  decode_mfma_kernel runs one workgroup per (request, split) item - the
// plan's item list or the plan-less static grid; decode_mfma_range_kernel (below, round 5) runs one wave per (equal
// piece of the step's keys, kv head) and is what planned launches of the default configuration take.
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
// range kernel: LDS per wave in which the partials of cut requests wait for the end of the piece.  A record is G rows of
// D floats + 16 log-sum-exps + the slot: 2128 B at G = 4, D = 128, so the 8352 B below hold 3 records there, 2 at G = 8
// (4176 B each), 1 at G = 16 - a piece cuts at most two requests, so two are ever in use; a record that does not fit is
// stored at once.  With the 32 KiB of tiles at D = 128 that is 64.6 KiB per workgroup: the two workgroups per CU a range
// launch runs (decode_mfma_ranges) fit.
static constexpr int kDmParkWaveB = 2 * (8 * 128 * 4 + 64 + 16);
// ... and on a byte pool, where a tile in flight is half the bytes, THREE workgroups per CU: 4256 B per wave = 2 records
// at G = 4, 1 at G = 8, none (stored at once) at G = 16; 49 KiB per workgroup at D = 128
static constexpr int kDmParkWaveB8 = 2 * (4 * 128 * 4 + 64 + 16);

// The state and the tile arithmetic of ONE wave, shared by the two kernels below: the gather of a 16-key tile into
// registers, its image in the wave-private LDS tile, S^T = K.Q^T, the online softmax of the wave's G columns, O^T += V^T.P^T,
// and the two ways a result leaves the wave.  KV8: the pool holds fp8 e5m2 bytes - a lane gathers 8 bytes instead of 16,
// expands them to 8 halves on the way into the LDS tile (e5m2 is the top byte of a half: one v_perm_b32 per two elements,
// exact), and the tile math runs in fp16 whatever the model dtype is (bf16 q is converted once per request; P is rounded to
// fp16); the output keeps the model dtype.  HBM bytes per context token halve; everything after the LDS tile is unchanged.
template <typename Tag, int D, bool KV8>
struct DmWave {
  typedef DmCfg<D> C;
  typedef typename std::conditional<KV8, f16_tag, Tag>::type CT;      // dtype of the tile math
  typedef typename std::conditional<KV8, u32x2, u32x4>::type raw_t;   // one lane's gathered chunk
  static constexpr int SRC_ROW_B = KV8 ? D : 2 * D;                   // bytes of one head row in the pool
  static constexpr int SRC_CH_B = KV8 ? 8 : 16;                       // bytes of the 8 elements a lane gathers
  static constexpr int TK = C::TK, ROW_B = C::ROW_B, CPR = C::CPR, RPL = C::RPL, NLD = C::NLD;
  static constexpr int KSTEPS = C::KSTEPS, DBLK = C::DBLK, TILE_B = C::TILE_B;

  int lane, col, kq;               // MFMA column (query head) / k quarter
  int ld_row, ld_ch;               // gather mapping: lane -> (row within the wave-load, 16-byte chunk)
  int i16, tr_row;                 // fragment-read addresses (constant per lane): V row this lane addresses in a tr read
  char *ldsK, *ldsV;               // this wave's tile
  // Address of the chunk a lane gathers from K row `slot`:  kbase[i] + slot * tok_bytes, ONE v_mad_u64_u32 (the slot is a
  // non-negative int32 and the token stride fits 32 bits - the host checks it - so no 64 x 64-bit product, which the
  // compiler expands into three quarter-rate multiplies: PMC, round 5: the address arithmetic was 40 % of the loop's
  // vector instructions).  kbase[i] holds everything that does not depend on the key: pool base, head, and the source
  // chunk, XOR-swizzled by the tile row so that the LDS image is conflict-free.  V sits at a launch-uniform distance
  // from K (the pool's two views; one add).
  uint32_t tok_bytes;
  int64_t v_minus_k;
  uint64_t kbase[NLD];
  u32x4 qf[KSTEPS];                // Q fragments: B operand of S^T = K.Q^T; lane (col, kq) holds Q[head col][32s + 8kq .. +7]
  f32x4_t oacc[DBLK];              // O^T: lane (col, kq) holds d = 16db + 4kq .. +3 of head col
  float m_run, l_run;
  float qk_scale, cap;

  __device__ __forceinline__ void init(const DecodeArgs& a, char* tile, int lane_, int hk) {
    lane = lane_;
    col = lane & 15; kq = lane >> 4;
    ld_row = lane / CPR; ld_ch = lane % CPR;
    i16 = lane & 15;
    tr_row = 4 * kq + (i16 >> 2);
    ldsK = tile;
    ldsV = tile + TILE_B;
    tok_bytes = (uint32_t)(a.kv_stride * (KV8 ? 1 : 2));
    v_minus_k = a.vbuf - a.kbuf;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int R = i * RPL + ld_row;
      kbase[i] = (uint64_t)(uintptr_t)a.kbuf + (uint64_t)hk * SRC_ROW_B + (uint64_t)((ld_ch ^ (R & (CPR - 1))) * SRC_CH_B);
    }
    cap = a.logit_cap;
    qk_scale = cap > 0.f ? a.sm_scale : a.sm_scale * kLog2eM;
  }
  // request b's query rows of kv head hk; padding columns repeat the last head and are never stored
  __device__ __forceinline__ void load_q(const DecodeArgs& a, int b, int hk, int G) {
    const int hcol = min(col, G - 1);
    const char* qp = (const char*)a.q + ((int64_t)b * a.q_stride + (int64_t)(hk * G + hcol) * D + 8 * kq) * 2;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      qf[s] = ld16(qp + s * 64);
      if constexpr (KV8 && std::is_same<Tag, bf16_tag>::value) qf[s] = bf16x8_to_f16x8(qf[s]);
    }
  }
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int db = 0; db < DBLK; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    m_run = kNegBigM;
    l_run = 0.f;
  }
  // gather tile `tile` of the 64 keys whose slots sit in `idxreg` (lanes past the keys hold slot 0, the pool's dummy row:
  // no select needed); NT: non-temporal loads
  template <bool NT>
  __device__ __forceinline__ void issue(int tile, int idxreg, raw_t (&kd)[NLD], raw_t (&vd)[NLD]) const {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int key = tile * TK + i * RPL + ld_row;
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
  }
  // registers -> this wave's LDS tile: position (row R, chunk ld_ch) holds source chunk ld_ch ^ R, i.e. logical chunk cg of
  // row R sits at chunk cg ^ R
  __device__ __forceinline__ void stage(const raw_t (&ks)[NLD], const raw_t (&vs)[NLD]) const {
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
  }
  // the staged tile `tile` of a 64-key span with n keys; CAP: logit soft-cap
  template <bool CAP>
  __device__ __forceinline__ void consume(int tile, int n) {
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
      const s16x4_m vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_m*)vp);
      const u32x2 vf = __builtin_bit_cast(u32x2, vt);
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[db][r] *= alpha;
      oacc[db] = mfma_pv<CT>(vf, pf, oacc[db]);
    }
  }
  // the wave owns heads hk*G .. hk*G+G-1 of request b outright: normalise and write the output rows from registers
  __device__ __forceinline__ void store_out(const DecodeArgs& a, int b, int hk, int G) const {
    if (col >= G) return;
    const float inv = 1.0f / l_run * a.out_scale;
    char* op = (char*)a.out + ((int64_t)b * a.o_stride + (int64_t)(hk * G + col) * D + 4 * kq) * 2;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
      u32x2 w;
      w[0] = pack2m<Tag>(oacc[db][0] * inv, oacc[db][1] * inv);
      w[1] = pack2m<Tag>(oacc[db][2] * inv, oacc[db][3] * inv);
      *(u32x2*)(op + db * 32) = w;
    }
  }
  // ... or its partial (normalised rows + log2-sum-exp) to slot `slot` of the workspace
  __device__ __forceinline__ void store_partial(const DecodeArgs& a, int hk, int G, int slot) const {
    if (col >= G) return;
    const float inv = 1.0f / l_run;
    const int64_t pi = (int64_t)(hk * G + col) * a.max_slots + slot;
    float* pp = a.part_o + pi * D + 4 * kq;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
      *(float4*)(pp + 16 * db) = make_float4(oacc[db][0] * inv, oacc[db][1] * inv, oacc[db][2] * inv, oacc[db][3] * inv);
    if (kq == 0) a.part_lse[pi] = m_run + __builtin_amdgcn_logf(l_run);
  }
};

// HPW ("head per wave", Hkv % 4 == 0): the 4 waves take the 4 adjacent KV heads of the SAME keys - the
// workgroup then reads whole 1 KiB token half-rows, and each wave owns its heads outright: no merge,
// no barrier anywhere.  Otherwise (few KV heads per rank) the waves split the keys of one head.
template <typename Tag, int D, bool HPW, bool KV8>
#ifndef SP_DEC_WAVES
#define SP_DEC_WAVES 3
#endif
__global__ __launch_bounds__(256, SP_DEC_WAVES) void decode_mfma_kernel(DecodeArgs a) {
  typedef DmCfg<D> C;
  typedef Elem<Tag> E;
  typedef DmWave<Tag, D, KV8> W;
  typedef typename W::raw_t raw_t;
  constexpr int TK = C::TK, NLD = C::NLD, TILE_B = C::TILE_B, WAVES = C::WAVES;
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

  W w;
  w.init(a, lds + wave * 2 * TILE_B, lane, hk);
  w.load_q(a, b, hk, G);
  w.reset();

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
      w.template issue<NT>(0, myidx, kr, vr);
      if (1 < ntile) w.template issue<NT>(1, myidx, kr2, vr2);
      for (int t = 0; t < ntile; t += 2) {
        w.stage(kr, vr);
        if (t + 2 < ntile) w.template issue<NT>(t + 2, myidx, kr, vr);
        w.template consume<CAP>(t, n);
        if (t + 1 < ntile) {
          w.stage(kr2, vr2);
          if (t + 3 < ntile) w.template issue<NT>(t + 3, myidx, kr2, vr2);
          w.template consume<CAP>(t + 1, n);
        }
      }
    } else {
      // one register set: the tile's registers are free as soon as they are in LDS, so the next
      // tile's gathers are issued right there and fly during this tile's MFMAs and softmax (and under
      // the other waves of the SIMD)
      w.template issue<NT>(0, myidx, kr, vr);
      for (int t = 0; t < ntile; ++t) {
        w.stage(kr, vr);
        if (t + 1 < ntile) w.template issue<NT>(t + 1, myidx, kr, vr);
        w.template consume<CAP>(t, n);
      }
    }
  }
  };
  // (a plan-less launch has no key count to compare: it streams only when the threshold is "always", the default)
  const bool nt = a.plan ? a.plan[3] >= a.nt_min_keys : a.nt_min_keys == 0;
  if (w.cap > 0.f) {
    if (nt) key_loop(std::true_type{}, std::true_type{}); else key_loop(std::false_type{}, std::true_type{});
  } else {
    if (nt) key_loop(std::true_type{}, std::false_type{}); else key_loop(std::false_type{}, std::false_type{});
  }

  if constexpr (HPW) {
    // ---- this wave owns heads hk*G .. hk*G+G-1: straight from registers
    if (nsplit == 1) w.store_out(a, b, hk, G);
    else w.store_partial(a, hk, G, slot0 + c);
    return;
  }
  // ---- merge the 4 waves through LDS (reusing the tile area): O^T[d][col], m, l per column
  const int col = w.col, kq = w.kq;
  constexpr int DBLK = C::DBLK;
  __syncthreads();  // every wave is done with its tiles
  float* sm_o = (float*)lds;                       // [WAVES][16 cols][D]
  float* sm_ml = sm_o + WAVES * 16 * D;            // [WAVES][16 cols][2]
  {
    float* dst = sm_o + (wave * 16 + col) * D + 4 * kq;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
      *(float4*)(dst + 16 * db) = make_float4(w.oacc[db][0], w.oacc[db][1], w.oacc[db][2], w.oacc[db][3]);
    if (kq == 0) {
      sm_ml[(wave * 16 + col) * 2] = w.m_run;
      sm_ml[(wave * 16 + col) * 2 + 1] = w.l_run;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G * D; i += 256) {
    const int d = i % D, g = i / D;
    float M = kNegBigM;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) M = fmaxf(M, sm_ml[(wv * 16 + g) * 2]);
    float Lsum = 0.f, O = 0.f;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) {
      const float wgt = __builtin_amdgcn_exp2f(sm_ml[(wv * 16 + g) * 2] - M);
      Lsum += sm_ml[(wv * 16 + g) * 2 + 1] * wgt;
      O += sm_o[(wv * 16 + g) * D + d] * wgt;
    }
    const int h = hk * G + g;
    const float o = O / Lsum;
    if (nsplit == 1) {
      E::store(a.out, (int64_t)b * a.o_stride + (int64_t)h * D + d, o * a.out_scale);
    } else {
      const int64_t pi = (int64_t)h * a.max_slots + (slot0 + c);
      a.part_o[pi * D + d] = o;
      if (d == 0) a.part_lse[pi] = M + __builtin_amdgcn_logf(Lsum);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// RANGE kernel (round 5): launches of the default configuration (16-bit or e5m2 byte pool, non-temporal
// gathers, no soft-cap) whose plan carries the range geometry (DecodeArgs::rplan).  The step's keys form one line,
// request after request; the plan cuts it into equal pieces - as many as two workgroups per CU have waves per kv head -
// and wave (piece j, kv head) walks piece j: the tail of the request the piece starts in, whole requests, the head
// of the request it ends in.  Every wave gathers the same number of keys - whatever the lengths are, there is no
// split size to choose and no list of items to deal out - and they all finish together.
//
// What that buys, measured on the (request, split) items of this same kernel (profiles/r05_decode_split_cost.txt):
//   * balance: batches whose items deal out evenly over the resident workgroups stream at 6.4 TB/s, the ragged headline
//     batch at 6.15;
//   * WHERE the partials go: fp32 partial rows written to HBM in between the gathers cost ~1.8 us per MB, ten times
//     their share of the bytes.  A piece cuts at most two requests (its first and its last); the wave PARKS those
//     partial rows in LDS behind the tiles and writes them, in whole 512-byte rows, when the piece is done - which
//     is when the launch as a whole runs out of gathers.  Requests inside a piece are written straight to the output.
// The walk never drains the wave's gathers: while a request's last 64 keys are consumed, the next request's record
// and first index register are fetched, and its first tile's gathers are issued where the current request has none
// left to issue.  The four waves of a workgroup walk the same piece for the four heads of the quad and share nothing
// (wave-private LDS tile, no barrier).  The arithmetic per key is that of decode_mfma_kernel; the cuts differ, so the
// bits are those of another - equally valid - split of the same sums.
template <typename Tag, int D, bool KV8>
__global__ __launch_bounds__(256, SP_DEC_WAVES) void decode_mfma_range_kernel(DecodeArgs a) {
  typedef DmCfg<D> C;
  typedef DmWave<Tag, D, KV8> W;
  typedef typename W::raw_t raw_t;
  constexpr int TK = C::TK, NLD = C::NLD, DBLK = C::DBLK, TILE_B = C::TILE_B;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // wave -> (piece, kv head): the heads of one piece are adjacent in launch order.  With the kv heads in fours a
  // workgroup is the four heads of ONE piece - its waves walk the same requests and read whole 1 KiB token half-rows
  // between them; with fewer (a tensor-parallel rank's one or two heads) its waves walk different pieces.  Nothing
  // here depends on which: the waves share nothing.
  const int gw = (int)blockIdx.x * 4 + wave;
  const int piece = gw / a.Hkv;
  const int hk = gw - piece * a.Hkv;
  const int32_t* rp = a.rplan;
  if (piece >= a.ranges) return;                            // (the launch's last workgroup when ranges * Hkv is not in fours)
  if (!range_plan_matches(rp, a.ranges, a.bs)) return;      // a plan built for another launch: nothing is read or written
  if (piece >= rp[0]) return;
  const int R = rp[1];
  const int32_t* posv = rp + kRangeHdr;
  const int lo = piece * R, hi = lo + R;                    // this piece of the line
  const int G = a.Hq / a.Hkv;

  // a request's share of the piece: keys cs .. ce of request b; `whole`: all of its keys (no partial)
  struct Seg { int b, cs, ce, end; bool whole; const int32_t* idx_row; };
  // A request's record is read with SCALAR loads written as asm: inside the loop hipcc turns such reads into vector
  // loads (the kernel's own stores may alias them as far as it knows), each followed by s_waitcnt vmcnt(0) - drained
  // round trips with no gather in flight.  A scalar load waits on lgkmcnt only: the tile in flight stays in flight.
  const void* kv_or_req = a.kv_start ? a.kv_start : a.req_idx;   // (a valid address either way)
  auto load_request = [&](int b, int& p0, int& p1, int64_t& req, int64_t& kv0) {
    const int32_t* pp = posv + b;
    if (a.idx64) {
      const int64_t *p2 = (const int64_t*)a.req_idx + b, *p3 = (const int64_t*)kv_or_req + b;
      asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x4\n\ts_load_dwordx2 %2, %5, 0x0\n\t"
                   "s_load_dwordx2 %3, %6, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(p0), "=&s"(p1), "=&s"(req), "=&s"(kv0) : "s"(pp), "s"(p2), "s"(p3) : "memory");
    } else {
      const int32_t *p2 = (const int32_t*)a.req_idx + b, *p3 = (const int32_t*)kv_or_req + b;
      int r32, k32;
      asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x4\n\ts_load_dword %2, %5, 0x0\n\t"
                   "s_load_dword %3, %6, 0x0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(p0), "=&s"(p1), "=&s"(r32), "=&s"(k32) : "s"(pp), "s"(p2), "s"(p3) : "memory");
      req = r32; kv0 = k32;
    }
    if (!a.kv_start) kv0 = 0;
  };
  // the share of request b (and of the ones behind it while they are empty); false: the piece holds no more keys
  auto segment_at = [&](int b, Seg& u) -> bool {
    for (b = __builtin_amdgcn_readfirstlane(b); b < a.bs; ++b) {     // (uniform by construction; the asm needs SGPRs)
      int p0, p1;
      int64_t req, kv0;
      load_request(b, p0, p1, req, kv0);
      if (p0 >= hi) return false;
      const int len = p1 - p0 - kRangeReqCost;
      if (p1 <= p0) continue;                               // an empty request takes no room on the line
      u.b = b;
      u.cs = max(lo - p0, 0);
      u.ce = min(hi - p0, len);
      u.end = len;
      u.whole = u.cs == 0 && u.ce == len;
      u.idx_row = a.r2t + req * a.r2t_stride + kv0;
      if (u.cs < u.ce) return true;                         // (cs >= ce: only the room behind b's keys is in the piece)
    }
    return false;
  };

  Seg cur;
  {
    int first;
    const int32_t* q = posv + a.bs + 1 + piece;             // start[piece]
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(first) : "s"(q) : "memory");
    if (first < 0 || !segment_at(first, cur)) return;
  }

  W w;
  w.init(a, lds + wave * 2 * TILE_B, lane, hk);
  w.load_q(a, cur.b, hk, G);
  w.reset();
  const int col = w.col, kq = w.kq;

  // Parking space of this wave behind the tiles: records of [G rows of D floats | 16 log-sum-exps | slot]
  const int park_rows_b = G * D * 4;
  const int park_unit_b = park_rows_b + 64 + 16;
  constexpr int kParkWaveB = KV8 ? kDmParkWaveB8 : kDmParkWaveB;
  const int park_cap = kParkWaveB / park_unit_b;            // (16-bit pool: at least 1 - G <= 16, D <= 128)
  char* park = lds + 4 * 2 * TILE_B + wave * kParkWaveB;
  int parked = 0;
  auto flush_parked = [&]() {
    for (int k = 0; k < parked; ++k) {
      const char* src = park + k * park_unit_b;
      const int slot_ = __builtin_amdgcn_readfirstlane(*(const int*)(src + park_rows_b + 64));
      for (int off = lane * 16; off < park_rows_b; off += 1024) {     // whole rows, 16 B per lane
        const int row = off / (D * 4), within = off - row * (D * 4);
        const int64_t pi = (int64_t)(hk * G + row) * a.max_slots + slot_;
        *(u32x4*)((char*)(a.part_o + pi * D) + within) = ld16(src + off);
      }
      if (lane < G) a.part_lse[(int64_t)(hk * G + lane) * a.max_slots + slot_] = *(const float*)(src + park_rows_b + lane * 4);
    }
  };
  // a request's result: straight to the output (all of its keys were in this piece) or as the partial of slot b + piece
  auto write_seg = [&](const Seg& u) {
    if (u.whole) {
      w.store_out(a, u.b, hk, G);
      return;
    }
    const int slot = u.b + piece;
    if (parked >= park_cap) {
      w.store_partial(a, hk, G, slot);
      return;
    }
    char* dst = park + parked * park_unit_b;
    if (col < G) {
      const float inv = 1.0f / w.l_run;
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        u32x4 v;
        v[0] = as_u32(w.oacc[db][0] * inv); v[1] = as_u32(w.oacc[db][1] * inv);
        v[2] = as_u32(w.oacc[db][2] * inv); v[3] = as_u32(w.oacc[db][3] * inv);
        st16(dst + col * (D * 4) + (16 * db + 4 * kq) * 4, v);
      }
      if (kq == 0) *(float*)(dst + park_rows_b + col * 4) = w.m_run + __builtin_amdgcn_logf(w.l_run);
    }
    if (lane == 0) *(int*)(dst + park_rows_b + 64) = slot;
    ++parked;
  };

  raw_t kr[NLD], vr[NLD];
  int pos = cur.cs;
  int curidx = (pos + lane < cur.ce) ? cur.idx_row[pos + lane] : 0;
  w.template issue<true>(0, curidx, kr, vr);
  for (;;) {
    const int n = min(64, cur.ce - pos);
    const int ntile = (n + TK - 1) / TK;
    const bool last = pos + 64 >= cur.ce;            // the last 64 keys of this request's share
    Seg nxt;
    bool have_next = false;
    int nextidx = 0;
    if (last) {
      // (a share that stops short of its request's end is the piece's last: nothing behind it)
      have_next = cur.ce == cur.end && segment_at(cur.b + 1, nxt);
      if (have_next) nextidx = (nxt.cs + lane < nxt.ce) ? nxt.idx_row[nxt.cs + lane] : 0;
    } else {
      nextidx = (pos + 64 + lane < cur.ce) ? cur.idx_row[pos + 64 + lane] : 0;
    }
    for (int t = 0; t < ntile; ++t) {
      w.stage(kr, vr);
      if (t + 1 < ntile) w.template issue<true>(t + 1, curidx, kr, vr);
      else if (!last || have_next) w.template issue<true>(0, nextidx, kr, vr);   // the next 64 keys' first tile: no drain in between
      w.template consume<false>(t, n);
    }
    if (last) {
      write_seg(cur);
      if (!have_next) break;
      cur = nxt;
      w.load_q(a, cur.b, hk, G);
      w.reset();
      pos = cur.cs;
    } else {
      pos += 64;
    }
    curidx = nextidx;
  }
  flush_parked();
}

template <int D, bool KV8>
static constexpr int dm_range_lds() { return DmCfg<D>::kStageBytes + DmCfg<D>::WAVES * (KV8 ? kDmParkWaveB8 : kDmParkWaveB); }

template <typename Tag, int D, bool KV8>
static int launch_dm_kv(const DecodeArgs& a, hipStream_t st) {
  typedef DmCfg<D> C;
  const bool hpw = a.Hkv % 4 == 0 && a.o_stride % 4 == 0;
  if (a.rplan) {                                  // the range geometry: one wave per (piece, kv head)
    if (a.o_stride % 4 || a.ranges <= 0 || a.logit_cap > 0.f || a.nt_min_keys != 0) return SP_ERR_INVALID_ARG;
    // (the kernel's dynamic-LDS limit on THIS device was raised by dm_range_workgroups: run_decode_mfma asked it)
    const unsigned grid = (unsigned)(((int64_t)a.ranges * a.Hkv + 3) / 4);
    decode_mfma_range_kernel<Tag, D, KV8><<<dim3(grid), 256, dm_range_lds<D, KV8>(), st>>>(a);
    SP_LAUNCH_CHECK();
    g_decode_last_kernel = 3;
    return SP_OK;
  }
  // planned: one workgroup per (plan item, head group), the launch covers max_slots items (the surplus
  // exits at once); plan-less: the static (request, split) grid
  const int64_t items = a.plan ? (int64_t)a.max_slots : (int64_t)a.bs * a.num_splits;
  if (hpw) {
    const unsigned grid = (unsigned)(items * (a.Hkv / 4));
    decode_mfma_kernel<Tag, D, true, KV8><<<dim3(grid), 256, C::kLdsBytes, st>>>(a);
  } else {
    const unsigned grid = (unsigned)(items * a.Hkv);
    decode_mfma_kernel<Tag, D, false, KV8><<<dim3(grid), 256, C::kLdsBytes, st>>>(a);
  }
  SP_LAUNCH_CHECK();
  g_decode_last_kernel = 2;
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
  if (a.rplan && !decode_mfma_ranges(a.Hq, a.Hkv, head_dim, dtype, a.kv8)) return SP_ERR_INVALID_ARG;
  if ((a.plan ? (int64_t)a.max_slots : (int64_t)a.bs * a.num_splits) * a.Hkv > 0x7fffffffLL) return SP_ERR_INVALID_ARG;
  if (head_dim == 128)
    return dtype == SP_BF16 ? launch_dm<bf16_tag, 128>(a, st) : launch_dm<f16_tag, 128>(a, st);
  if (head_dim == 64)
    return dtype == SP_BF16 ? launch_dm<bf16_tag, 64>(a, st) : launch_dm<f16_tag, 64>(a, st);
  return SP_ERR_UNSUPPORTED;
}

// Pieces the range kernel wants: TWO workgroups per CU (three on a byte pool), four waves each, over the kv heads - all
// resident at once (the runtime is asked, once per instantiation), every wave's piece the same length.  Measured on MI355X, bs 256, contexts
// U[128, 4096] (profiles/r05_decode_range.txt): 1 / 2 / 3 workgroups per CU 340.7 / 339.7 / 348.1 us, two rounds of
// shorter pieces 366 - 392: a piece pays its start, its cut requests' partials and its tail once, so few long streams
// beat many short ones as soon as they keep HBM busy, and 8 waves per CU with one 8 KiB tile in flight each do
// (6.6 TB/s).  0: the shape is not the range kernel's (16-bit q, G <= 16, D in {64, 128}).
template <typename Tag, int D, bool KV8>
static int dm_range_workgroups() {
  // per device of the process: the attribute below is the device's, and so is the occupancy
  static PerDevice<int> cache;
  int workgroups = 0;
  cache.get(workgroups, [](int dev) {
    int per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (dm_range_lds<D, KV8>() > 64 * 1024 &&
        hipFuncSetAttribute((const void*)decode_mfma_range_kernel<Tag, D, KV8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            dm_range_lds<D, KV8>()) != hipSuccess)
      return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_mfma_range_kernel<Tag, D, KV8>, 256, dm_range_lds<D, KV8>()) !=
            hipSuccess || per_cu < 1)
      return 0;
    const int want = KV8 ? 3 : 2;
    return (per_cu < want ? per_cu : want) * prop.multiProcessorCount;
  });
  return workgroups;
}

int decode_mfma_ranges(int num_q_heads, int num_kv_heads, int head_dim, int dtype, int kv8) {
  if (num_q_heads / num_kv_heads > 16 || (dtype != SP_BF16 && dtype != SP_F16)) return 0;
  int wgs = 0;
  auto of = [&](auto tag, auto dim) {
    typedef decltype(tag) T;
    constexpr int Dd = decltype(dim)::value;
    return kv8 ? dm_range_workgroups<T, Dd, true>() : dm_range_workgroups<T, Dd, false>();
  };
  if (head_dim == 128) wgs = dtype == SP_BF16 ? of(bf16_tag{}, std::integral_constant<int, 128>{}) : of(f16_tag{}, std::integral_constant<int, 128>{});
  else if (head_dim == 64) wgs = dtype == SP_BF16 ? of(bf16_tag{}, std::integral_constant<int, 64>{}) : of(f16_tag{}, std::integral_constant<int, 64>{});
  // a wave per (piece, kv head); at most 1024 pieces - a rank's single head: its pieces are short already, and every cut
  // costs a partial (bs 128, one head: 2048 / 1024 pieces 32.9 / 31.5 us, traffic 1.10 x the algorithmic bytes at 2048)
  const int pieces = wgs * 4 / num_kv_heads;
  return pieces > 1024 ? 1024 : pieces > 0 ? pieces : (wgs > 0 ? 1 : 0);
}

}  // namespace sp
