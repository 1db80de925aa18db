// out[M, N] = x[M, K] . w[N, K]^T for M <= 16 (decode at small batch): a weight-streaming kernel.
//
// Call sites it serves: the four per-layer projections and the LM head at decode time -
// QKVParallelLinear / RowParallelLinear / MergedColumnParallelLinear.forward (nn/layers/linear.py:
// 696-760, 423-470, 1033-1155 -> F.linear) and LogitsProcessor._get_logits
// (nn/layers/logits_processor.py:340-376).  Above 16 rows the library GEMM (hipBLASLt) stays.
//
// At M <= 16 the product is a pure stream of the weight matrix (2 flop per weight byte per row):
// HBM-bound.  hipBLASLt runs these shapes at 2.5-5.0 TB/s of weight bytes (bs 1, Llama-3-8B:
// o_proj 13.6 us for 33.5 MB, qkv 15.5 us for 50 MB); a kernel shaped for the stream does better:
//   * one workgroup = 16 output columns (16 rows of W), its 8 waves take the k-steps round-robin,
//     so at any moment the workgroup reads 512 contiguous bytes of each of its 16 rows;
//   * per k-step (32 elements) a lane loads 16 B of W and 16 B of x straight into the A / B operand
//     registers of v_mfma_f32_16x16x32 (rows of W are the A rows, rows of x the B columns) - no LDS
//     on the way in; x (<= 16 x K, <= 460 KB) is re-read from L2 by every workgroup;
//   * 8 k-steps of loads are in flight per wave before the first MFMA consumes one;
//   * the 8 waves' 16x16 partial tiles are summed through LDS in a fixed order (deterministic),
//     rounded once, and stored.
//
// Epilogue fusion (round 4): with `epilogue` = 1 the weight is the merged gate|up matrix [2 I, K] of LlamaMLP
// (nn/models/llama/llama.py:62-66) and the output is SiluAndMul (nn/layers/activation.py:21-31) of the projection,
// [M, I]: a workgroup streams 8 gate rows and the 8 matching up rows (still 16 rows of W, still one workgroup per
// 16 rows), and the lanes that finish the gate columns take the up sums of the same columns from the LDS reduction
// and write round(silu(round(gate))) * round(up), rounded - bit for bit the plain projection followed by
// sp_silu_and_mul, with the activation launch, its boundary and the [M, 2 I] round trip gone.  (The other way round -
// the activation in the PROLOGUE of the down projection, every workgroup recomputing the M x K activations it
// streams - was built first and measured 2 x SLOWER than the two launches: 76 vs 38 us at 1 row; the exponentials
// of all 16 lane rows of every k-step are vector work the stream cannot hide.  profiles/NOTES.md, round 4.)
#include "sp_common.h"

namespace sp {

typedef float f32x4_g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_g __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_g __attribute__((ext_vector_type(8)));

template <typename Tag>
__device__ __forceinline__ f32x4_g mfma_g(const u32x4& a, const u32x4& b, const f32x4_g& c);
template <>
__device__ __forceinline__ f32x4_g mfma_g<bf16_tag>(const u32x4& a, const u32x4& b, const f32x4_g& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_g, a), __builtin_bit_cast(bf16x8_g, b),
                                                 c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_g mfma_g<f16_tag>(const u32x4& a, const u32x4& b, const f32x4_g& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_g, a), __builtin_bit_cast(f16x8_g, b),
                                                c, 0, 0, 0);
}

struct SkinnyArgs {
  const char* x;
  const char* w;
  void* out;
  int M, N, K;
  int64_t x_stride, w_stride, out_stride;   // elements
};

constexpr int kSkWaves = 8;

// NB = 16-column blocks per workgroup: one x fragment feeds NB MFMAs (x traffic out of L2 drops to
// 1/NB of the weight stream).  Measured on MI355X it is SLOWER than NB = 1 wherever N allows it
// (gate_up 28672x4096 at 1 row: NB 2 53.9 us vs NB 1 44.1 us; LM head 128256x4096: NB 4 206 us vs
// NB 1 166 us; same sign at 8 rows) - fewer, fatter workgroups lose more than the x reads cost -
// so launch_skinny always takes NB = 1; the parameter stays for re-measurement.
// NT (sp_debug_set("skinny_nt", 1); OFF by default): non-temporal weight loads - the weights are read exactly once by
// exactly one workgroup (MI355X_MICROARCH.md, 'nt-weights').  Measured on this kernel it LOSES: 1 row qkv 14.5 vs 13.6 us,
// gate_up 46.5 vs 43.3, LM head 186 vs 166; bench.py --bs 1: 4.13 vs 3.98 ms/step (round 3, tools/bench_gemv.py).
__device__ __forceinline__ u32x4 ld16_nt(const void* p) { return __builtin_nontemporal_load((const u32x4*)p); }

// EPI 1 (NB = 1 only): a.N = I output columns; the 16 W rows of a workgroup are gate rows n0 .. n0+7 (tile rows 0-7) and
// up rows I + n0 .. I + n0 + 7 (tile rows 8-15)
template <typename Tag, int NB, int UNROLL, bool NT, int EPI>
__global__ __launch_bounds__(kSkWaves * 64) void gemm_skinny_kernel(SkinnyArgs a) {
  typedef Elem<Tag> E;
  static_assert(EPI == 0 || NB == 1, "the SiLU-mul epilogue pairs the two halves of one 16-row tile");
  __shared__ float red[kSkWaves][NB][16 * 17];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r16 = lane & 15, q = lane >> 4;            // operand row / k-quarter of the lane
  const int n0 = EPI == 1 ? blockIdx.x * 8 : blockIdx.x * 16 * NB;
  const bool mrow = r16 < a.M;
  const char* wp[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    int n = min(n0 + 16 * nb + r16, a.N - 1);          // clamp: columns past N are computed, never stored
    if constexpr (EPI == 1) n = (r16 < 8 ? 0 : a.N) + min(n0 + (r16 & 7), a.N - 1);
    wp[nb] = a.w + ((int64_t)n * a.w_stride + 8 * q) * 2;
  }
  const char* xp = a.x + ((int64_t)min(r16, a.M - 1) * a.x_stride + 8 * q) * 2;
  const int ksteps = a.K / 32;
  f32x4_g acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_g{0.f, 0.f, 0.f, 0.f};
  const u32x4 zero = {0u, 0u, 0u, 0u};
  // this wave's k-steps: wave, wave + 8, ...; processed in groups of UNROLL with all loads first
  int ks = wave;
  for (; ks + (UNROLL - 1) * kSkWaves < ksteps; ks += UNROLL * kSkWaves) {
    u32x4 wf[UNROLL][NB], xf[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t off = (int64_t)(ks + u * kSkWaves) * 64;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wf[u][nb] = NT ? ld16_nt(wp[nb] + off) : ld16(wp[nb] + off);
      xf[u] = mrow ? ld16(xp + off) : zero;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_g<Tag>(wf[u][nb], xf[u], acc[nb]);
  }
  for (; ks < ksteps; ks += kSkWaves) {
    const int64_t off = (int64_t)ks * 64;
    const u32x4 xf = mrow ? ld16(xp + off) : zero;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_g<Tag>(NT ? ld16_nt(wp[nb] + off) : ld16(wp[nb] + off), xf, acc[nb]);
  }
  // acc[nb][r] = partial of out[m = r16][n0 + 16 nb + 4q + r]; sum the waves in wave order
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][nb][(4 * q + r) * 17 + r16] = acc[nb][r];
  __syncthreads();
  for (int nb = wave; nb < NB; nb += kSkWaves) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 4 * q + r;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kSkWaves; ++w) s += red[w][nb][col * 17 + r16];
      if constexpr (EPI == 1) {
        // tile columns 0-7 are gate sums, 8-15 the up sums of the same output columns: the lanes of the gate half
        // finish both (same summation order as above) and apply silu_mul_vec_kernel's arithmetic to the ROUNDED sums
        float su = 0.f;
#pragma unroll
        for (int w = 0; w < kSkWaves; ++w) su += red[w][nb][((col & 7) + 8) * 17 + r16];
        const int n = n0 + col;
        if (mrow && q < 2 && n < a.N)
          E::store(a.out, (int64_t)r16 * a.out_stride + n, __fmul_rn(E::round(silu_ref(E::round(s))), E::round(su)));
      } else {
        const int n = n0 + 16 * nb + col;
        if (mrow && n < a.N) E::store(a.out, (int64_t)r16 * a.out_stride + n, s);
      }
    }
  }
}

static int g_skinny_nt = 0;     // sp_debug_set("skinny_nt", 0 / 1): A/B switch of the non-temporal weight loads
void set_skinny_nt(int v) { g_skinny_nt = v; }
// sp_debug_set("skinny_unroll16", 0 / 1): 16 k-steps of loads in flight per wave where a wave's share is a multiple of
// 16 (K = 4096: the whole share in ONE round of loads instead of two)
static int g_skinny_u16 = 0;
void set_skinny_unroll16(int v) { g_skinny_u16 = v; }

template <typename Tag>
static void launch_skinny(const SkinnyArgs& a, int epilogue, hipStream_t st) {
  const dim3 block(kSkWaves * 64);
  const bool u16 = g_skinny_u16 && (a.K / 32) % (16 * kSkWaves) == 0;
  if (epilogue == 1 && u16) gemm_skinny_kernel<Tag, 1, 16, false, 1><<<dim3((a.N + 7) / 8), block, 0, st>>>(a);
  else if (epilogue == 1) gemm_skinny_kernel<Tag, 1, 8, false, 1><<<dim3((a.N + 7) / 8), block, 0, st>>>(a);
  else if (u16) gemm_skinny_kernel<Tag, 1, 16, false, 0><<<dim3((a.N + 15) / 16), block, 0, st>>>(a);
  else if (g_skinny_nt) gemm_skinny_kernel<Tag, 1, 8, true, 0><<<dim3((a.N + 15) / 16), block, 0, st>>>(a);
  else gemm_skinny_kernel<Tag, 1, 8, false, 0><<<dim3((a.N + 15) / 16), block, 0, st>>>(a);
}

}  // namespace sp

extern "C" int sp_gemm_skinny(void* out, const void* x, const void* w, int M, int N, int K, int64_t x_stride,
                              int64_t w_stride, int64_t out_stride, int epilogue, int dtype, void* stream) {
  SP_CHECK_ARG(M >= 0 && N >= 0 && K > 0 && (epilogue == 0 || epilogue == 1));
  if (M == 0 || N == 0) return SP_OK;
  SP_CHECK_ARG(out && x && w);
  if (M > 16 || K % 32 != 0) return SP_ERR_UNSUPPORTED;
  if (dtype != SP_BF16 && dtype != SP_F16) return SP_ERR_UNSUPPORTED;
  SP_CHECK_ARG(x_stride % 8 == 0 && w_stride % 8 == 0 && x_stride >= K && w_stride >= K && out_stride >= N);
  SP_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0);
  sp::SkinnyArgs a{(const char*)x, (const char*)w, out, M, N, K, x_stride, w_stride, out_stride};
  if (dtype == SP_BF16) sp::launch_skinny<sp::bf16_tag>(a, epilogue, (hipStream_t)stream);
  else sp::launch_skinny<sp::f16_tag>(a, epilogue, (hipStream_t)stream);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
