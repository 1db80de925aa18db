// Ragged extend (prefill with cached prefix / cross-attention) on the matrix cores - 16-bit dtypes.
//
// Replaces extend_attention_fwd (nn/attention/triton_attn/extend_attention.py:16-327) and the
// flashinfer ragged + paged + merge_state path (nn/attention/flashinfer_backend.py:400-444): one
// kernel, every key (cached prefix AND the new tokens, which the KV store wrote just before) is
// gathered from the paged pool through req_to_token.
//
// Tiling (cdna_hip_programming.md section 3 and Appendix B 'Fused attention prefill'):
//   * workgroup = 4 waves = (request, kv head, block of BM new tokens); a wave owns 32 query rows of
//     ONE query head; the 4 waves cover Gk = min(G,4) heads x 4/Gk row blocks, so a K/V tile staged
//     once in LDS is shared by the whole GQA group;
//   * S^T = K . Q^T with v_mfma_f32_32x32x16 (A = K rows from LDS by ds_read_b128, B = Q fragments held
//     in registers for the whole kernel): the accumulator then has the query row on the LANE and
//     the 16 keys of the lane half in registers, so the softmax row max / sum are in-lane plus one
//     exchange with lane^32, and the rescale factor of O^T is lane-local;
//   * O^T = V^T . P^T: P^T is taken straight from the S^T accumulator registers as the B operand
//     ("an accumulator tile as the next MFMA's operand": registers 8s..8s+7 -> k-step s, key order
//     16s + 8(j>>2) + 4h + (j&3)); the matching V^T fragments come from the row-major V tile by
//     ds_read_b64_tr_b16 (hardware transpose), two reads per k-step;
//   * K/V tiles of 64 keys: gathered rows (full 256-B lines, 16 B per lane) -> registers ->
//     LDS with padded row strides (K +16 B: conflict-free ds_read_b128; V +64 B: conflict-free
//     tr reads); the next tile's global loads are issued before the current tile is consumed.
#include <type_traits>

#include "attention_internal.h"

namespace sp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

struct ExtendArgs {
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
  const int32_t* ext_lens;
  const int32_t* ext_start;
  int bs, Hq, Hkv;
  int64_t q_stride, o_stride, kv_stride;  // elements
  float sm_scale, logit_cap;
  int causal;
  int kv8;      // 1: fp8 e5m2 pool (kv_stride in bytes); tile math in fp16, see decode_mfma.hip
  int window;   // sliding window: a row at kv position p sees keys [p - window, p]; < 0 = unlimited
};

static constexpr float kLog2eX = 1.4426950408889634f;
static constexpr float kNegBigX = -1.0e30f;

template <typename Tag>
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c);
template <>
__device__ __forceinline__ f32x16 mfma32<bf16_tag>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<f16_tag>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <typename Tag>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2<bf16_tag>(float lo, float hi) {
  bf16x2_t b;
  b[0] = (__bf16)lo;
  b[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, b);
}
template <>
__device__ __forceinline__ uint32_t pack2<f16_tag>(float lo, float hi) {
  f16x2_t b;
  b[0] = (_Float16)lo;
  b[1] = (_Float16)hi;
  return __builtin_bit_cast(uint32_t, b);
}

// value held by lane ^ 32, by one v_permlane32_swap (VALU) instead of a ds_bpermute round trip:
// the swap exchanges lanes 32-63 of its first operand with lanes 0-31 of its second
__device__ __forceinline__ float xchg32(float x) {
  const uint32_t u = as_u32(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const uint32_t upper_gets = r[0], lower_gets = r[1];
  return as_f32((threadIdx.x & 32) ? upper_gets : lower_gets);
}

template <int D>
struct ExtCfg {
  static constexpr int BN = 64;                    // keys per tile
  static constexpr int ROW_B = D * 2;              // bytes per K/V row
  static constexpr int SK = ROW_B + 16;            // K row stride in LDS
  static constexpr int SV = ROW_B + 64;            // V row stride in LDS
  static constexpr int CPR = ROW_B / 16;           // 16-byte chunks per row
  static constexpr int RPP = 256 / CPR;            // rows staged per pass of the 256 threads
  static constexpr int PASSES = BN / RPP;
  static constexpr int KSTEPS = D / 16;            // MFMA k-steps of Q.K^T
  static constexpr int DBLK = D / 32;              // 32-wide d blocks of O^T
  static constexpr int kTileBytes = BN * (SK + SV);   // one K tile + one V tile
  static constexpr int kLdsBytes = 2 * kTileBytes;    // double-buffered: one barrier per tile
};

template <typename Tag, int D, int GK, bool KV8>
__global__ __launch_bounds__(256, 2) void extend_mfma_kernel(ExtendArgs a) {
  typedef ExtCfg<D> C;
  typedef typename std::conditional<KV8, f16_tag, Tag>::type CT;      // dtype of the tile math
  typedef typename std::conditional<KV8, u32x2, u32x4>::type raw_t;   // one thread's gathered chunk
  constexpr int BN = C::BN, SK = C::SK, SV = C::SV, CPR = C::CPR, RPP = C::RPP, PASSES = C::PASSES;
  constexpr int KSTEPS = C::KSTEPS, DBLK = C::DBLK;
  constexpr int BM = 128 / GK;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // two (K,V) tile buffers: tile t lives in buffer t & 1

  const int b = blockIdx.z;
  const int E = a.ext_lens[b];
  const int row0 = blockIdx.x * BM;
  if (row0 >= E) return;
  const int G = a.Hq / a.Hkv;
  const int halves = G / GK;                       // query-head blocks per kv head (2 when G = 8)
  const int hk = blockIdx.y / halves;
  const int hh = blockIdx.y - hk * halves;
  const int L = (int)load_idx(a.seq_lens, b, a.idx64);
  const int P = a.causal ? L - E : 0;              // cached prefix length
  const int64_t req = load_idx(a.req_idx, b, a.idx64);
  const int64_t kv0 = a.kv_start ? load_idx(a.kv_start, b, a.idx64) : 0;
  const int32_t* idx_row = a.r2t + req * a.r2t_stride + kv0;
  const int64_t t0 = a.ext_start[b];
  // keys this workgroup can see: the prefix and new tokens up to its last row (causal), else all
  const int kv_len = a.causal ? min(L, P + min(row0 + BM, E)) : L;
  const int ntiles = (kv_len + BN - 1) / BN;
  // sliding window (causal only): the first row of the block reaches furthest back
  const bool windowed = a.causal && a.window >= 0;
  const int tbeg = windowed ? max(0, P + row0 - a.window) / BN : 0;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 31, h = lane >> 5;
  const int g = wave % GK, rb = wave / GK;
  const int head = hk * G + hh * GK + g;
  const int r0 = row0 + rb * 32;                   // this wave's first query row
  const int my_row = r0 + c;                       // this lane's query row (may be >= E: masked out)
  const bool wave_live = r0 < E;

  // ---- Q fragments: B operand of S^T = K.Q^T, lane (c,h) holds Q[row c][16ks + 8h .. +7]
  u32x4 qf[KSTEPS];
  {
    const int qrow = min(my_row, E - 1);
    const char* qp = (const char*)a.q + ((t0 + qrow) * a.q_stride + (int64_t)head * D + 8 * h) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      qf[ks] = ld16(qp + ks * 32);
      if constexpr (KV8 && std::is_same<Tag, bf16_tag>::value) qf[ks] = bf16x8_to_f16x8(qf[ks]);
    }
  }
  const float cap = a.logit_cap;
  const float qk_scale = cap > 0.f ? a.sm_scale : a.sm_scale * kLog2eX;
  const int row_limit = a.causal ? P + my_row : 0x7fffffff;   // last visible key index
  const int row_first = windowed ? P + my_row - a.window : 0;  // first visible key index (may be < 0)

  f32x16 oacc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = kNegBigX, l_run = 0.f;

  // ---- staging: thread -> (row, 16-byte chunk) of the tile, PASSES rows each for K and V
  const int st_row = tid / CPR, st_ch = tid % CPR;
  const int64_t tok_bytes = a.kv_stride * (KV8 ? 1 : 2);
  const int64_t head_off = (int64_t)hk * D * (KV8 ? 1 : 2) + st_ch * (KV8 ? 8 : 16);
  raw_t kreg[PASSES], vreg[PASSES];
  // slot indices run one tile ahead of the row gathers: the dependent req_to_token -> row chain is
  // then never waited for inside the loop (the wave is in-order: a wait on a fresh index load at
  // the top of an iteration stalled the whole tile's compute behind an L2/HBM round trip)
  int slot_next[PASSES];
  auto fetch_slots = [&](int tile) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int key = tile * BN + p * RPP + st_row;
      slot_next[p] = key < kv_len ? idx_row[key] : 0;   // rows past the end read the dummy slot 0
    }
  };
  auto prefetch = [&](int tile) {   // gathers of `tile` from slot_next, then the indices of tile+1
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t off = (int64_t)slot_next[p] * tok_bytes + head_off;
      kreg[p] = *(const raw_t*)(a.kbuf + off);
      vreg[p] = *(const raw_t*)(a.vbuf + off);
    }
    fetch_slots(tile + 1);
  };
  auto stage = [&](int buf) {
    char* dK = lds + buf * C::kTileBytes;
    char* dV = dK + BN * SK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int row = p * RPP + st_row;
      if constexpr (KV8) {
        st16(dK + row * SK + st_ch * 16, expand_e5m2x8(kreg[p]));
        st16(dV + row * SV + st_ch * 16, expand_e5m2x8(vreg[p]));
      } else {
        st16(dK + row * SK + st_ch * 16, kreg[p]);
        st16(dV + row * SV + st_ch * 16, vreg[p]);
      }
    }
  };

  // tr-read address pieces: lane i of a 16-lane group supplies row (i>>2), columns 4*(i&3)..+3
  const int i16 = lane & 15, g16 = lane >> 4;
  const int tr_rowq = i16 >> 2, tr_col = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;  // bytes

  // pipeline: tile t+1's global gathers fly during compute(t); they are written to the OTHER LDS
  // buffer right after compute(t), and one barrier per tile both publishes tile t+1 and retires
  // every wave's reads of tile t (whose buffer is overwritten only in iteration t+1)
  fetch_slots(tbeg);
  prefetch(tbeg);
  stage(tbeg & 1);
  __syncthreads();
  for (int t = tbeg; t < ntiles; ++t) {
    if (t + 1 < ntiles) prefetch(t + 1);
    const char* ldsK = lds + (t & 1) * C::kTileBytes;
    const char* ldsV = ldsK + BN * SK;
    const int key0 = t * BN;
    // a wave skips tiles that lie entirely above its rows' diagonal (wave-uniform)
    const bool visible = wave_live && (!a.causal || key0 <= P + min(r0 + 31, E - 1)) &&
                         (!windowed || key0 + BN - 1 >= P + r0 - a.window);
    if (visible) {
      // ---- S^T = K . Q^T for the two 32-key blocks of the tile
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        const char* kp = ldsK + (kb * 32 + c) * SK + h * 16;
        // all fragment reads of the block are issued before its first MFMA (distinct registers), so
        // one LDS latency is exposed per block instead of one per MFMA
        u32x4 kf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) kf[ks] = ld16(kp + ks * 32);
        // keep the scheduler from sinking each read next to its MFMA again (it then reuses one
        // register quad and waits lgkmcnt(0) before every MFMA)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) s[kb] = mfma32<CT>(kf[ks], qf[ks], s[kb]);
      }
      // ---- scale, mask, online softmax (query row on the lane; keys in registers + lane^32)
      float mx = kNegBigX;
      // interior tiles (entirely below every row's diagonal and inside the key range) need no mask
      const bool need_mask = key0 + BN > kv_len || (a.causal && key0 + BN - 1 > P + r0) ||
                             (windowed && key0 < P + min(r0 + 31, E - 1) - a.window);
      if (need_mask) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float x = s[kb][r] * qk_scale;
            if (cap > 0.f) x = cap * tanhf(x / cap) * kLog2eX;
            x = (key < kv_len && key <= row_limit && key >= row_first) ? x : -INFINITY;
            s[kb][r] = x;
            mx = fmaxf(mx, x);
          }
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float x = s[kb][r] * qk_scale;
            if (cap > 0.f) x = cap * tanhf(x / cap) * kLog2eX;
            s[kb][r] = x;
            mx = fmaxf(mx, x);
          }
      }
      mx = fmaxf(mx, xchg32(mx));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
          s[kb][r] = p;
          psum += p;
        }
      psum += xchg32(psum);
      l_run = l_run * alpha + psum;
      if (!__all(alpha == 1.0f)) {  // the row maxima settle after the first tiles: usually skipped
#pragma unroll
        for (int db = 0; db < DBLK; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      }
      // ---- O^T += V^T . P^T
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        // V^T fragments of the whole 32-key block first (2 k-steps x DBLK d-blocks, two transposed
        // reads each), then the MFMAs
        u32x4 vf[2][DBLK];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const int keyA = kb * 32 + 16 * sidx + 4 * h + tr_rowq;       // rows for elements 0..3
          const char* vp = ldsV + keyA * SV + tr_col;
#pragma unroll
          for (int db = 0; db < DBLK; ++db) {
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(vp + db * 64));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(vp + 8 * SV + db * 64));
            const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
            vf[sidx][db][0] = lo2[0];
            vf[sidx][db][1] = lo2[1];
            vf[sidx][db][2] = hi2[0];
            vf[sidx][db][3] = hi2[1];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          u32x4 pf;  // B operand: registers 8s..8s+7 of the S^T block, rounded to the KV dtype
#pragma unroll
          for (int j = 0; j < 4; ++j)
            pf[j] = pack2<CT>(s[kb][8 * sidx + 2 * j], s[kb][8 * sidx + 2 * j + 1]);
#pragma unroll
          for (int db = 0; db < DBLK; ++db) oacc[db] = mfma32<CT>(vf[sidx][db], pf, oacc[db]);
        }
      }
    }
    if (t + 1 < ntiles) stage((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: O[row][head][d] = O^T[d][row] / l ; lane holds 4 consecutive d per register quad
  if (my_row < E && wave_live) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;  // no visible key (empty encoder): zeros
    char* op = (char*)a.out + ((t0 + my_row) * a.o_stride + (int64_t)head * D) * 2;
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int quad = 0; quad < 4; ++quad) {
        const int d = db * 32 + 8 * quad + 4 * h;
        u32x2 w;
        w[0] = pack2<Tag>(oacc[db][4 * quad] * inv, oacc[db][4 * quad + 1] * inv);
        w[1] = pack2<Tag>(oacc[db][4 * quad + 2] * inv, oacc[db][4 * quad + 3] * inv);
        *(u32x2*)(op + d * 2) = w;
      }
  }
}

template <typename Tag, int D, int GK>
static int launch_extend(const ExtendArgs& a, int max_extend_len, int halves, hipStream_t st) {
  typedef ExtCfg<D> C;
  constexpr int BM = 128 / GK;
  const dim3 grid((max_extend_len + BM - 1) / BM, a.Hkv * halves, a.bs);
  if (a.kv8) extend_mfma_kernel<Tag, D, GK, true><<<grid, 256, C::kLdsBytes, st>>>(a);
  else extend_mfma_kernel<Tag, D, GK, false><<<grid, 256, C::kLdsBytes, st>>>(a);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

template <typename Tag, int D>
static int dispatch_extend_group(const ExtendArgs& a, int G, int max_extend_len, hipStream_t st) {
  // a workgroup's 4 waves take Gk = 4, 2 or 1 query heads of the KV head (the largest that divides G)
  // and the remaining G / Gk head blocks become extra workgroups: any group width works
  if (G < 1 || G > 64) return SP_ERR_UNSUPPORTED;
  if (G % 4 == 0) return launch_extend<Tag, D, 4>(a, max_extend_len, G / 4, st);
  if (G % 2 == 0) return launch_extend<Tag, D, 2>(a, max_extend_len, G / 2, st);
  return launch_extend<Tag, D, 1>(a, max_extend_len, G, st);
}

// 16-bit dtypes, D in {64,128}; anything else returns SP_ERR_UNSUPPORTED and the caller takes the
// row-stream path.
int run_extend_mfma(void* out, const void* q, const void* k_buffer, const void* v_buffer,
                    const int32_t* req_to_token, int64_t req_to_token_stride,
                    const void* req_pool_indices, const void* seq_lens, const void* kv_start,
                    int idx64, const int32_t* extend_seq_lens, const int32_t* extend_start_loc,
                    int batch_size, int num_q_heads, int num_kv_heads, int head_dim, int64_t q_stride,
                    int64_t out_stride, int64_t kv_buffer_stride, float sm_scale, float logit_cap,
                    int causal, int window_left, int max_extend_len, int dtype, int kv8, hipStream_t st) {
  if (dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  if (head_dim != 64 && head_dim != 128) return SP_ERR_UNSUPPORTED;
  if (batch_size > 65535 || num_q_heads > 65535) return SP_ERR_UNSUPPORTED;   // grid.z, grid.y
  if (q_stride % 8 || out_stride % 4 || kv_buffer_stride % 8) return SP_ERR_UNSUPPORTED;
  if (kv8 && (((uintptr_t)k_buffer | (uintptr_t)v_buffer) & 7)) return SP_ERR_UNSUPPORTED;
  ExtendArgs a;
  a.out = out; a.q = q; a.kbuf = (const char*)k_buffer; a.vbuf = (const char*)v_buffer;
  a.r2t = req_to_token; a.r2t_stride = req_to_token_stride; a.req_idx = req_pool_indices;
  a.seq_lens = seq_lens; a.kv_start = kv_start; a.idx64 = idx64; a.ext_lens = extend_seq_lens;
  a.ext_start = extend_start_loc; a.bs = batch_size; a.Hq = num_q_heads; a.Hkv = num_kv_heads;
  a.q_stride = q_stride; a.o_stride = out_stride; a.kv_stride = kv_buffer_stride;
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.causal = causal;
  a.window = causal ? window_left : -1;
  a.kv8 = kv8;
  const int G = num_q_heads / num_kv_heads;
  if (max_extend_len <= 0) return SP_OK;
  if (dtype == SP_BF16) {
    return head_dim == 128 ? dispatch_extend_group<bf16_tag, 128>(a, G, max_extend_len, st)
                           : dispatch_extend_group<bf16_tag, 64>(a, G, max_extend_len, st);
  }
  return head_dim == 128 ? dispatch_extend_group<f16_tag, 128>(a, G, max_extend_len, st)
                         : dispatch_extend_group<f16_tag, 64>(a, G, max_extend_len, st);
}

}  // namespace sp
