// HBM-bound row kernels of the hot path: RMSNorm (+fused add), SiLU-mul, rotary (+fused KV
// store), KV store, req_to_token scatter, positions.  All vectorised to 16 B per lane
// (cdna_hip_programming.md Guideline 13); each has a scalar twin for odd shapes / alignments.
//
// Rounding points follow the reference's torch path (forward_native) so that 16-bit results
// are bit-comparable with torch's tensor arithmetic: see the per-kernel comments.
#include "sp_common.h"

// The rounding points below must survive to the ISA.  With the default -ffp-contract=fast hipcc
// rewrites `round16(a*b) - round16(c*d)` on fp16 data into v_fma_f16 (one rounding fewer than
// torch's tensor arithmetic), and a source pragma does not stop the backend: this translation
// unit is compiled with -ffp-contract=off (scratchpad_amd/build.py PER_FILE_FLAGS).
#pragma clang fp contract(off)

namespace sp {

static constexpr int kBlock = 256;

__device__ __forceinline__ float block_sum(float v, float* smem) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < kBlock / 64; ++i) t += smem[i];
  __syncthreads();
  return t;
}

// ------------------------------------------------------------------------------------ RMSNorm
// nn/layers/layernorm.py:34-51.  y = round(xf * rsqrt(mean(xf^2)+eps)) * w  (second product
// rounded again: torch multiplies two `dtype` tensors).  Fused: xf = x + residual in fp32,
// residual <- round(xf), and the UNROUNDED xf is normalised.
template <typename Tag, bool FUSED, int MAXIT>
__global__ __launch_bounds__(kBlock) void rmsnorm_vec_kernel(void* out,  // may alias x_io (fused)
                                                              void* x_io, void* residual,
                                                              const void* __restrict__ weight,
                                                              int hidden, int64_t x_stride,
                                                              int64_t r_stride, int64_t o_stride,
                                                              float eps) {
  typedef Elem<Tag> E;
  constexpr int V = E::kVec;
  __shared__ float smem[kBlock / 64];
  const int64_t row = blockIdx.x;
  const char* xrow = (const char*)x_io + row * x_stride * E::kBytes;
  char* rrow = FUSED ? (char*)residual + row * r_stride * E::kBytes : nullptr;
  char* orow = (char*)out + row * o_stride * E::kBytes;
  const int nvec = hidden / V;
  float cache[MAXIT][V];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int v = threadIdx.x + it * kBlock;
    if (v < nvec) {
      unpack16<Tag>(ld16(xrow + (int64_t)v * 16), cache[it]);
      if (FUSED) {
        float r[V];
        unpack16<Tag>(ld16(rrow + (int64_t)v * 16), r);
#pragma unroll
        for (int e = 0; e < V; ++e) cache[it][e] = __fadd_rn(cache[it][e], r[e]);
        st16(rrow + (int64_t)v * 16, pack16<Tag>(cache[it]));
      }
#pragma unroll
      for (int e = 0; e < V; ++e) ss += cache[it][e] * cache[it][e];
    }
  }
  const float mean = block_sum(ss, smem) / (float)hidden;
  const float rs = 1.0f / sqrtf(mean + eps);
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int v = threadIdx.x + it * kBlock;
    if (v < nvec) {
      float w[V], y[V];
      unpack16<Tag>(ld16((const char*)weight + (int64_t)v * 16), w);
#pragma unroll
      for (int e = 0; e < V; ++e) y[e] = __fmul_rn(E::round(__fmul_rn(cache[it][e], rs)), w[e]);
      st16(orow + (int64_t)v * 16, pack16<Tag>(y));
    }
  }
}

// any hidden / stride / alignment; two passes over the row (second pass re-reads)
template <typename Tag, bool FUSED>
__global__ __launch_bounds__(kBlock) void rmsnorm_scalar_kernel(void* out,  // may alias x_io
                                                                 void* x_io, void* residual,
                                                                 const void* __restrict__ weight,
                                                                 int hidden, int64_t x_stride,
                                                                 int64_t r_stride,
                                                                 int64_t o_stride, float eps) {
  typedef Elem<Tag> E;
  __shared__ float smem[kBlock / 64];
  const int64_t row = blockIdx.x;
  float ss = 0.f;
  for (int i = threadIdx.x; i < hidden; i += kBlock) {
    float v = E::load(x_io, row * x_stride + i);
    // pass two recomputes the fp32 sum from x and the OLD residual, so nothing is written yet
    if (FUSED) v = __fadd_rn(v, E::load(residual, row * r_stride + i));
    ss += v * v;
  }
  const float mean = block_sum(ss, smem) / (float)hidden;
  const float rs = 1.0f / sqrtf(mean + eps);
  for (int i = threadIdx.x; i < hidden; i += kBlock) {
    float v = E::load(x_io, row * x_stride + i);
    if (FUSED) {
      v = __fadd_rn(v, E::load(residual, row * r_stride + i));
      E::store(residual, row * r_stride + i, v);
    }
    const float w = E::load(weight, i);
    E::store(out, row * o_stride + i, __fmul_rn(E::round(__fmul_rn(v, rs)), w));
  }
}

// ------------------------------------------------------------------------------ SiLU-and-mul
// nn/layers/activation.py:22-24: F.silu(a) * b with both factors in `dtype` (silu rounded
// before the product, as torch does).
__device__ __forceinline__ float silu_f32(float a) { return silu_ref(a); }   // sp_common.h

template <typename Tag>
__global__ __launch_bounds__(kBlock) void silu_mul_vec_kernel(void* __restrict__ out,
                                                               const void* __restrict__ x,
                                                               int64_t num_tokens, int d,
                                                               int64_t x_stride,
                                                               int64_t o_stride) {
  typedef Elem<Tag> E;
  constexpr int V = E::kVec;
  const int vec_per_row = d / V;
  const int64_t total = num_tokens * vec_per_row;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / vec_per_row;
    const int v = (int)(i - t * vec_per_row);
    const char* row = (const char*)x + t * x_stride * E::kBytes;
    float a[V], b[V], y[V];
    unpack16<Tag>(ld16(row + (int64_t)v * 16), a);
    unpack16<Tag>(ld16(row + ((int64_t)d * E::kBytes) + (int64_t)v * 16), b);
#pragma unroll
    for (int e = 0; e < V; ++e) y[e] = __fmul_rn(E::round(silu_f32(a[e])), b[e]);
    st16((char*)out + t * o_stride * E::kBytes + (int64_t)v * 16, pack16<Tag>(y));
  }
}

template <typename Tag>
__global__ __launch_bounds__(kBlock) void silu_mul_scalar_kernel(void* __restrict__ out,
                                                                  const void* __restrict__ x,
                                                                  int64_t num_tokens, int d,
                                                                  int64_t x_stride,
                                                                  int64_t o_stride) {
  typedef Elem<Tag> E;
  const int64_t total = num_tokens * d;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / d;
    const int j = (int)(i - t * d);
    const float a = E::load(x, t * x_stride + j), b = E::load(x, t * x_stride + d + j);
    E::store(out, t * o_stride + j, __fmul_rn(E::round(silu_f32(a)), b));
  }
}

// ------------------------------------------------------------------- rotary (+ fused KV store)
// nn/layers/rotary_embedding.py:23-49: o1 = x1*cos - x2*sin, o2 = x2*cos + x1*sin with every
// product and the sum rounded to `dtype` (torch tensor arithmetic).  One workgroup per token.
struct RotaryArgs {
  const int64_t* positions;
  void* q;
  void* k;
  const void* v;
  const void* cache;
  void* k_buffer;
  void* v_buffer;
  const int64_t* loc;
  int64_t q_stride, k_stride, v_stride, kv_buffer_stride;
  int Hq, Hkv, head_size, rot, is_neox;
};

template <typename Tag>
__device__ __forceinline__ void rotate_pair(float x1, float x2, float c, float s, float& o1,
                                            float& o2) {
  typedef Elem<Tag> E;
  o1 = E::round(__fsub_rn(E::round(__fmul_rn(x1, c)), E::round(__fmul_rn(x2, s))));
  o2 = E::round(__fadd_rn(E::round(__fmul_rn(x2, c)), E::round(__fmul_rn(x1, s))));
}

template <typename Tag, int V>  // V = Elem::kVec (16-byte path) or 1 (scalar path)
__global__ __launch_bounds__(kBlock) void rotary_kernel(RotaryArgs a) {
  typedef Elem<Tag> E;
  const int64_t t = blockIdx.x;
  const int64_t pos = a.positions[t];
  const int half = a.rot / 2;
  const char* cos_row = (const char*)a.cache + pos * a.rot * E::kBytes;
  const char* sin_row = cos_row + (int64_t)half * E::kBytes;
  const bool store = a.k_buffer != nullptr;
  const int64_t slot = store ? a.loc[t] : 0;
  char* kdst = store ? (char*)a.k_buffer + slot * a.kv_buffer_stride * E::kBytes : nullptr;
  const int nheads = a.Hq + a.Hkv;

  if (a.is_neox) {
    // unit = (head, vector j of the first half); partner vector sits `half` elements later
    const int vec_per_head = half / V;
    for (int u = threadIdx.x; u < nheads * vec_per_head; u += kBlock) {
      const int h = u / vec_per_head, j = u - h * vec_per_head;
      const bool is_k = h >= a.Hq;
      const int hh = is_k ? h - a.Hq : h;
      char* base = is_k ? (char*)a.k + (t * a.k_stride + (int64_t)hh * a.head_size) * E::kBytes
                        : (char*)a.q + (t * a.q_stride + (int64_t)hh * a.head_size) * E::kBytes;
      float x1[V], x2[V], c[V], s[V], o1[V], o2[V];
      if constexpr (V > 1) {
        unpack16<Tag>(ld16(base + (int64_t)j * 16), x1);
        unpack16<Tag>(ld16(base + ((int64_t)half * E::kBytes) + (int64_t)j * 16), x2);
        unpack16<Tag>(ld16(cos_row + (int64_t)j * 16), c);
        unpack16<Tag>(ld16(sin_row + (int64_t)j * 16), s);
      } else {
        x1[0] = E::load(base, j);
        x2[0] = E::load(base, half + j);
        c[0] = E::load(cos_row, j);
        s[0] = E::load(sin_row, j);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) rotate_pair<Tag>(x1[e], x2[e], c[e], s[e], o1[e], o2[e]);
      if constexpr (V > 1) {
        const u32x4 p1 = pack16<Tag>(o1), p2 = pack16<Tag>(o2);
        st16(base + (int64_t)j * 16, p1);
        st16(base + ((int64_t)half * E::kBytes) + (int64_t)j * 16, p2);
        if (store && is_k) {
          char* d = kdst + (int64_t)hh * a.head_size * E::kBytes;
          st16(d + (int64_t)j * 16, p1);
          st16(d + ((int64_t)half * E::kBytes) + (int64_t)j * 16, p2);
        }
      } else {
        E::store(base, j, o1[0]);
        E::store(base, half + j, o2[0]);
        if (store && is_k) {
          char* d = kdst + (int64_t)hh * a.head_size * E::kBytes;
          E::store(d, j, o1[0]);
          E::store(d, half + j, o2[0]);
        }
      }
    }
  } else {
    // GPT-J interleaved: pairs (2i, 2i+1); unit = V elements = V/2 pairs (scalar: one pair)
    constexpr int W = V > 1 ? V : 2;
    const int vec_per_head = a.rot / W;
    for (int u = threadIdx.x; u < nheads * vec_per_head; u += kBlock) {
      const int h = u / vec_per_head, j = u - h * vec_per_head;
      const bool is_k = h >= a.Hq;
      const int hh = is_k ? h - a.Hq : h;
      char* base = is_k ? (char*)a.k + (t * a.k_stride + (int64_t)hh * a.head_size) * E::kBytes
                        : (char*)a.q + (t * a.q_stride + (int64_t)hh * a.head_size) * E::kBytes;
      float x[W], o[W];
      if constexpr (V > 1) {
        unpack16<Tag>(ld16(base + (int64_t)j * 16), x);
      } else {
        x[0] = E::load(base, 2 * j);
        x[1] = E::load(base, 2 * j + 1);
      }
#pragma unroll
      for (int e = 0; e < W / 2; ++e) {
        const int ci = j * (W / 2) + e;
        rotate_pair<Tag>(x[2 * e], x[2 * e + 1], E::load(cos_row, ci), E::load(sin_row, ci),
                         o[2 * e], o[2 * e + 1]);
      }
      char* d = (store && is_k) ? kdst + (int64_t)hh * a.head_size * E::kBytes : nullptr;
      if constexpr (V > 1) {
        const u32x4 p = pack16<Tag>(o);
        st16(base + (int64_t)j * 16, p);
        if (d) st16(d + (int64_t)j * 16, p);
      } else {
        E::store(base, 2 * j, o[0]);
        E::store(base, 2 * j + 1, o[1]);
        if (d) {
          E::store(d, 2 * j, o[0]);
          E::store(d, 2 * j + 1, o[1]);
        }
      }
    }
  }
  if (!store) return;
  // pass-through tail of k (rotary_dim < head_size) and the v rows go to the pool unchanged
  const int tail = a.head_size - a.rot;
  if (tail > 0) {
    const int per_head = tail / V;
    for (int u = threadIdx.x; u < a.Hkv * per_head; u += kBlock) {
      const int h = u / per_head, j = u - h * per_head;
      const char* src = (const char*)a.k + (t * a.k_stride + (int64_t)h * a.head_size + a.rot) * E::kBytes;
      char* d = kdst + ((int64_t)h * a.head_size + a.rot) * E::kBytes;
      if constexpr (V > 1) st16(d + (int64_t)j * 16, ld16(src + (int64_t)j * 16));
      else E::store(d, j, E::load(src, j));
    }
  }
  {
    char* vdst = (char*)a.v_buffer + slot * a.kv_buffer_stride * E::kBytes;
    const char* vsrc = (const char*)a.v + t * a.v_stride * E::kBytes;
    const int n = a.Hkv * a.head_size / V;
    for (int u = threadIdx.x; u < n; u += kBlock) {
      if constexpr (V > 1) st16(vdst + (int64_t)u * 16, ld16(vsrc + (int64_t)u * 16));
      else E::store(vdst, u, E::load(vsrc, u));
    }
  }
}

// ----------------------------------------------------------------------------------- KV store
// memory/pool.py:414-424: buffer[loc[t]] = cache[t].  Rows are Hkv*D contiguous elements.
template <int BYTES>  // 16 (vector) or the element size (scalar)
__global__ __launch_bounds__(kBlock) void kv_store_kernel(char* __restrict__ kbuf,
                                                           char* __restrict__ vbuf,
                                                           const int64_t* __restrict__ loc,
                                                           const char* __restrict__ k,
                                                           const char* __restrict__ v,
                                                           int64_t num_tokens, int k_row_bytes,
                                                           int v_row_bytes, int64_t k_stride_b,
                                                           int64_t v_stride_b, int64_t kb_stride_b,
                                                           int64_t vb_stride_b) {
  const int ku = k_row_bytes / BYTES, vu = v_row_bytes / BYTES;
  const int per_tok = ku + vu;
  const int64_t total = num_tokens * per_tok;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / per_tok;
    int u = (int)(i - t * per_tok);
    const int64_t slot = loc[t];
    const char* src;
    char* dst;
    if (u < ku) {
      src = k + t * k_stride_b + (int64_t)u * BYTES;
      dst = kbuf + slot * kb_stride_b + (int64_t)u * BYTES;
    } else {
      u -= ku;
      src = v + t * v_stride_b + (int64_t)u * BYTES;
      dst = vbuf + slot * vb_stride_b + (int64_t)u * BYTES;
    }
    if constexpr (BYTES == 16) *(u32x4*)dst = *(const u32x4*)src;
    else if constexpr (BYTES == 4) *(uint32_t*)dst = *(const uint32_t*)src;
    else *(uint16_t*)dst = *(const uint16_t*)src;
  }
}

// e5m2 pool: one thread converts 8 consecutive elements of a K or V row and stores 8 bytes
template <typename Tag>
__global__ __launch_bounds__(kBlock) void kv_store_fp8_kernel(char* __restrict__ kbuf, char* __restrict__ vbuf,
                                                               const int64_t* __restrict__ loc,
                                                               const void* __restrict__ k,
                                                               const void* __restrict__ v, int64_t num_tokens,
                                                               int row_elems, int64_t k_stride, int64_t v_stride,
                                                               int64_t kb_stride, int64_t vb_stride,
                                                               float k_scale, float v_scale) {
  typedef Elem<Tag> E;
  const int units = row_elems / 8;                     // per row
  const int64_t total = num_tokens * units * 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / (2 * units);
    int u = (int)(i - t * 2 * units);
    const bool is_v = u >= units;
    if (is_v) u -= units;
    const int64_t slot = loc[t];
    const void* src = is_v ? v : k;
    const int64_t sbase = t * (is_v ? v_stride : k_stride) + (int64_t)u * 8;
    // the reference's two steps and two roundings (memory/pool.py:401-412): cache_k.div_(k_scale) in
    // the activation dtype (an fp32 division rounded back to that dtype), then .to(float8_e5m2)
    const float scale = is_v ? v_scale : k_scale;
    u32x2 out;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      uint32_t word = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        word |= f32_to_e5m2_bits(E::round(E::load(src, sbase + 4 * w + e) / scale)) << (8 * e);
      out[w] = word;
    }
    char* dst = (is_v ? vbuf + slot * vb_stride : kbuf + slot * kb_stride) + (int64_t)u * 8;
    *(u32x2*)dst = out;
  }
}

// --------------------------------------------------------------- req_to_token write / positions
__device__ __forceinline__ int64_t block_prefix_i64(const int64_t* a32or64, const int32_t* a32,
                                                    int n, int64_t* smem) {
  int64_t s = 0;
  for (int i = threadIdx.x; i < n; i += kBlock) s += a32 ? (int64_t)a32[i] : a32or64[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
  __syncthreads();
  int64_t t = 0;
#pragma unroll
  for (int i = 0; i < kBlock / 64; ++i) t += smem[i];
  return t;
}

// scheduler/schedule_batch.py:1546-1580
__global__ __launch_bounds__(kBlock) void write_req_to_token_kernel(
    int32_t* __restrict__ table, int64_t row_stride, const int64_t* __restrict__ req_pool_indices,
    const int64_t* __restrict__ pre_lens, const int64_t* __restrict__ seq_lens,
    const int64_t* __restrict__ extend_lens, const int64_t* __restrict__ out_cache_loc) {
  __shared__ int64_t smem[kBlock / 64];
  const int b = blockIdx.x;
  const int64_t start = block_prefix_i64(extend_lens, nullptr, b, smem);
  const int64_t pre = pre_lens[b], n = seq_lens[b] - pre;
  int32_t* row = table + req_pool_indices[b] * row_stride + pre;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) row[i] = (int32_t)out_cache_loc[start + i];
}

// model_executor/forward_info.py:423-449
__global__ __launch_bounds__(kBlock) void compute_position_kernel(
    int64_t* __restrict__ positions, int32_t* __restrict__ extend_start_loc,
    const int32_t* __restrict__ prefix_lens, const int32_t* __restrict__ extend_lens) {
  __shared__ int64_t smem[kBlock / 64];
  const int b = blockIdx.x;
  const int64_t start = block_prefix_i64(nullptr, extend_lens, b, smem);
  const int64_t pre = prefix_lens[b];
  const int n = extend_lens[b];
  for (int i = threadIdx.x; i < n; i += kBlock) positions[start + i] = pre + i;
  if (threadIdx.x == 0) extend_start_loc[b] = (int32_t)start;
}

// model_executor/forward_info.py:469-471
__global__ void clamp_position_kernel(int64_t* __restrict__ positions,
                                      const void* __restrict__ seq_lens, int idx64, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int64_t v = load_idx(seq_lens, i, idx64) - 1;
    positions[i] = v < 0 ? 0 : v;
  }
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace sp

using namespace sp;

extern "C" int sp_abi_version(void) { return SP_ABI_VERSION; }

extern "C" const char* sp_status_string(int status) {
  switch (status) {
    case SP_OK: return "ok";
    case SP_ERR_INVALID_ARG: return "invalid argument";
    case SP_ERR_UNSUPPORTED: return "unsupported shape/dtype";
    case SP_ERR_WORKSPACE: return "workspace too small";
    case SP_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}

template <typename Tag, bool FUSED>
static int launch_rmsnorm(void* out, void* x, void* residual, const void* weight, int64_t T,
                          int hidden, int64_t xs, int64_t rs, int64_t os, float eps,
                          hipStream_t st) {
  typedef Elem<Tag> E;
  constexpr int V = E::kVec;
  constexpr int MAXIT = (E::kBytes == 4) ? 8 : 4;
  const bool vec = hidden % V == 0 && xs % V == 0 && os % V == 0 && (!FUSED || rs % V == 0) &&
                   aligned16(out) && aligned16(x) && aligned16(weight) &&
                   (!FUSED || aligned16(residual)) && hidden / V <= kBlock * MAXIT;
  if (vec)
    rmsnorm_vec_kernel<Tag, FUSED, MAXIT><<<dim3((unsigned)T), kBlock, 0, st>>>(
        out, x, residual, weight, hidden, xs, rs, os, eps);
  else
    rmsnorm_scalar_kernel<Tag, FUSED><<<dim3((unsigned)T), kBlock, 0, st>>>(
        out, x, residual, weight, hidden, xs, rs, os, eps);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_rmsnorm(void* out, const void* x, const void* weight, int64_t num_tokens,
                          int hidden, int64_t x_stride, int64_t out_stride, float eps, int dtype,
                          void* stream) {
  SP_CHECK_ARG(num_tokens >= 0 && hidden > 0);
  if (num_tokens == 0) return SP_OK;
  SP_CHECK_ARG(out && x && weight);
  SP_DISPATCH_DTYPE(dtype, return (launch_rmsnorm<Tag, false>(out, (void*)x, nullptr, weight,
                                                               num_tokens, hidden, x_stride, 0,
                                                               out_stride, eps,
                                                               (hipStream_t)stream)));
}

extern "C" int sp_fused_add_rmsnorm(void* x, void* residual, const void* weight,
                                    int64_t num_tokens, int hidden, int64_t x_stride,
                                    int64_t res_stride, float eps, int dtype, void* stream) {
  SP_CHECK_ARG(num_tokens >= 0 && hidden > 0);
  if (num_tokens == 0) return SP_OK;
  SP_CHECK_ARG(x && residual && weight);
  SP_DISPATCH_DTYPE(dtype, return (launch_rmsnorm<Tag, true>(x, x, residual, weight, num_tokens,
                                                              hidden, x_stride, res_stride,
                                                              x_stride, eps,
                                                              (hipStream_t)stream)));
}

template <typename Tag>
static int launch_silu(void* out, const void* x, int64_t T, int d, int64_t xs, int64_t os,
                       hipStream_t st) {
  typedef Elem<Tag> E;
  constexpr int V = E::kVec;
  const bool vec = d % V == 0 && xs % V == 0 && os % V == 0 && aligned16(out) && aligned16(x);
  const int64_t work = vec ? T * (d / V) : T * (int64_t)d;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (vec) silu_mul_vec_kernel<Tag><<<dim3((unsigned)blocks), kBlock, 0, st>>>(out, x, T, d, xs, os);
  else silu_mul_scalar_kernel<Tag><<<dim3((unsigned)blocks), kBlock, 0, st>>>(out, x, T, d, xs, os);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_silu_and_mul(void* out, const void* x, int64_t num_tokens, int d,
                               int64_t x_stride, int64_t out_stride, int dtype, void* stream) {
  SP_CHECK_ARG(num_tokens >= 0 && d > 0);
  if (num_tokens == 0) return SP_OK;
  SP_CHECK_ARG(out && x);
  SP_DISPATCH_DTYPE(dtype, return (launch_silu<Tag>(out, x, num_tokens, d, x_stride, out_stride,
                                                     (hipStream_t)stream)));
}

template <typename Tag>
static int launch_rotary(const RotaryArgs& a, int64_t T, hipStream_t st) {
  typedef Elem<Tag> E;
  constexpr int V = E::kVec;
  const bool store = a.k_buffer != nullptr;
  bool vec = a.head_size % V == 0 && a.q_stride % V == 0 && a.k_stride % V == 0 &&
             aligned16(a.q) && aligned16(a.k) && aligned16(a.cache) &&
             (a.is_neox ? (a.rot / 2) % V == 0 : a.rot % V == 0);
  if (store)
    vec = vec && a.v_stride % V == 0 && a.kv_buffer_stride % V == 0 && aligned16(a.v) &&
          aligned16(a.k_buffer) && aligned16(a.v_buffer) && (a.head_size - a.rot) % V == 0;
  if (vec) rotary_kernel<Tag, V><<<dim3((unsigned)T), kBlock, 0, st>>>(a);
  else rotary_kernel<Tag, 1><<<dim3((unsigned)T), kBlock, 0, st>>>(a);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_rotary_embedding(const int64_t* positions, void* q, void* k,
                                   const void* cos_sin_cache, int64_t num_tokens, int num_q_heads,
                                   int num_kv_heads, int head_size, int rotary_dim,
                                   int64_t q_stride, int64_t k_stride, int is_neox, const void* v,
                                   int64_t v_stride, void* k_buffer, void* v_buffer,
                                   const int64_t* out_cache_loc, int64_t kv_buffer_stride,
                                   int dtype, void* stream) {
  SP_CHECK_ARG(positions && q && k && cos_sin_cache && num_tokens >= 0);
  SP_CHECK_ARG(num_q_heads >= 0 && num_kv_heads >= 0 && head_size > 0);
  SP_CHECK_ARG(rotary_dim > 0 && rotary_dim <= head_size && rotary_dim % 2 == 0);
  const bool store = k_buffer != nullptr || v_buffer != nullptr;
  if (store) SP_CHECK_ARG(k_buffer && v_buffer && v && out_cache_loc && kv_buffer_stride > 0);
  if (num_tokens == 0) return SP_OK;
  RotaryArgs a;
  a.positions = positions; a.q = q; a.k = k; a.v = v; a.cache = cos_sin_cache;
  a.k_buffer = store ? k_buffer : nullptr; a.v_buffer = store ? v_buffer : nullptr;
  a.loc = out_cache_loc; a.q_stride = q_stride; a.k_stride = k_stride; a.v_stride = v_stride;
  a.kv_buffer_stride = kv_buffer_stride; a.Hq = num_q_heads; a.Hkv = num_kv_heads;
  a.head_size = head_size; a.rot = rotary_dim; a.is_neox = is_neox ? 1 : 0;
  SP_DISPATCH_DTYPE(dtype, return (launch_rotary<Tag>(a, num_tokens, (hipStream_t)stream)));
}

extern "C" int sp_kv_store(void* k_buffer, void* v_buffer, const int64_t* loc, const void* k,
                           const void* v, int64_t num_tokens, int num_kv_heads, int head_dim,
                           int v_head_dim, int64_t k_stride, int64_t v_stride,
                           int64_t k_buffer_stride, int64_t v_buffer_stride, int dtype,
                           void* stream) {
  SP_CHECK_ARG(k_buffer && v_buffer && loc && k && v && num_tokens >= 0);
  SP_CHECK_ARG(num_kv_heads > 0 && head_dim > 0 && v_head_dim > 0);
  if (dtype != SP_F32 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  if (num_tokens == 0) return SP_OK;
  const int eb = dtype == SP_F32 ? 4 : 2;
  const int krow = num_kv_heads * head_dim * eb, vrow = num_kv_heads * v_head_dim * eb;
  const int64_t ks = k_stride * eb, vs = v_stride * eb, kbs = k_buffer_stride * eb,
                vbs = v_buffer_stride * eb;
  const bool vec = krow % 16 == 0 && vrow % 16 == 0 && ks % 16 == 0 && vs % 16 == 0 &&
                   kbs % 16 == 0 && vbs % 16 == 0 && aligned16(k_buffer) && aligned16(v_buffer) &&
                   aligned16(k) && aligned16(v);
  const int unit = vec ? 16 : eb;
  const int64_t work = num_tokens * ((krow + vrow) / unit);
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = (hipStream_t)stream;
#define SP_KV_LAUNCH(B)                                                                        \
  kv_store_kernel<B><<<dim3((unsigned)blocks), kBlock, 0, st>>>(                              \
      (char*)k_buffer, (char*)v_buffer, loc, (const char*)k, (const char*)v, num_tokens, krow, \
      vrow, ks, vs, kbs, vbs)
  if (vec) SP_KV_LAUNCH(16);
  else if (eb == 4) SP_KV_LAUNCH(4);
  else SP_KV_LAUNCH(2);
#undef SP_KV_LAUNCH
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_kv_store_fp8(void* k_buffer, void* v_buffer, const int64_t* loc, const void* k,
                               const void* v, int64_t num_tokens, int num_kv_heads, int head_dim,
                               int64_t k_stride, int64_t v_stride, int64_t k_buffer_stride,
                               int64_t v_buffer_stride, float k_scale, float v_scale, int src_dtype,
                               void* stream) {
  SP_CHECK_ARG(k_buffer && v_buffer && loc && k && v && num_tokens >= 0);
  SP_CHECK_ARG(num_kv_heads > 0 && head_dim > 0 && k_scale > 0.f && v_scale > 0.f);
  if (num_tokens == 0) return SP_OK;
  const int row = num_kv_heads * head_dim;
  if (row % 8 || k_buffer_stride % 8 || v_buffer_stride % 8 || ((uintptr_t)k_buffer & 7) ||
      ((uintptr_t)v_buffer & 7))
    return SP_ERR_UNSUPPORTED;
  const int64_t work = num_tokens * (row / 8) * 2;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  SP_DISPATCH_DTYPE(src_dtype, (kv_store_fp8_kernel<Tag><<<dim3((unsigned)blocks), kBlock, 0, (hipStream_t)stream>>>(
                                   (char*)k_buffer, (char*)v_buffer, loc, k, v, num_tokens, row, k_stride, v_stride,
                                   k_buffer_stride, v_buffer_stride, k_scale, v_scale)));
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_write_req_to_token(int32_t* req_to_token, int64_t row_stride,
                                     const int64_t* req_pool_indices, const int64_t* pre_lens,
                                     const int64_t* seq_lens, const int64_t* extend_lens,
                                     const int64_t* out_cache_loc, int batch_size, void* stream) {
  SP_CHECK_ARG(req_to_token && req_pool_indices && pre_lens && seq_lens && extend_lens &&
               out_cache_loc && batch_size >= 0 && row_stride > 0);
  if (batch_size == 0) return SP_OK;
  write_req_to_token_kernel<<<dim3(batch_size), kBlock, 0, (hipStream_t)stream>>>(
      req_to_token, row_stride, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_compute_position(int64_t* positions, int32_t* extend_start_loc,
                                   const int32_t* extend_prefix_lens,
                                   const int32_t* extend_seq_lens, int batch_size, void* stream) {
  SP_CHECK_ARG(positions && extend_start_loc && extend_prefix_lens && extend_seq_lens &&
               batch_size >= 0);
  if (batch_size == 0) return SP_OK;
  compute_position_kernel<<<dim3(batch_size), kBlock, 0, (hipStream_t)stream>>>(
      positions, extend_start_loc, extend_prefix_lens, extend_seq_lens);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" int sp_clamp_position(int64_t* positions, const void* seq_lens, int idx64,
                                 int batch_size, void* stream) {
  SP_CHECK_ARG(positions && seq_lens && batch_size >= 0);
  if (batch_size == 0) return SP_OK;
  clamp_position_kernel<<<dim3((batch_size + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      positions, seq_lens, idx64, batch_size);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
