// Direct all-reduce through IPC-mapped peer buffers: the `ca_comm` seam of GroupCoordinator
// (distributed/parallel_state.py:266-267, 326-347: should_custom_ar / custom_all_reduce), which the
// reference declares and never fills - its per-layer [T, hidden] SUM all-reduce (linear.py:1148-1149)
// always goes through NCCL - and the fusion of that all-reduce with what follows it in every decoder
// layer: residual add + RMSNorm (nn/models/llama/llama.py:202-224 -> nn/layers/layernorm.py:22-32).
//
// On an MI355X node the 8 GPUs are a full xGMI mesh (7 links per GPU), so for the small per-layer
// message (2 MiB at T = 128, hidden = 8192) a ring is the wrong shape: every rank should talk to its 7
// peers at once.  Two forms, chosen by message size:
//   * one-shot (<= 256 KiB): every rank reads all peers' inputs and sums them;
//   * two-shot: rank r sums slice r of every peer's input (reduce-scatter by reads over 7 links at
//     once), publishes it, and gathers the other ranks' reduced slices.
// Mechanics (the scheme of vLLM's custom all-reduce, restated for gfx950): each rank owns one
// fine-grained (cross-device coherent) region [flags | data | reduced]; every rank maps every region by
// hipIpcOpenMemHandle.  A workgroup synchronises only with the SAME workgroup index on the other ranks:
// lane p stores this call's epoch into flag slot [block][my rank] of peer p (system-scope release) and
// spins until slot [block][p] of its own region shows the epoch (system-scope acquire).  Epochs
// increase monotonically per communicator, so no flag is ever reset.  The epoch is DEVICE state: every
// workgroup keeps its own counter word in its rank's region (read at entry, advanced by 3 at exit;
// only that workgroup of that rank ever touches it, and launches of one stream are serialised), so
// the launch arguments are the same for every call and a captured launch replays correctly - the
// all-reduces of a HIP-graph decode step need no library call (the reference runs pynccl inside its
// graphs, distributed/parallel_state.py:256-302, device_communicators/pynccl.py:108-130).
//
// FAILURE IS COLLECTIVE AND FATAL.  A barrier that waits longer than the caller's time-out (wall
// clock, s_memrealtime; default 30 s) raises the status word of EVERY rank's region and this rank's
// host-visible status word (pinned host memory, mapped into the device): the host reads that word
// without any synchronisation at every forward boundary (CustomAllReduce.poll()).  A launch that finds
// its own region's status word raised (a peer timed out earlier) raises its host word too and stops
// waiting in its barriers, so every rank learns of the failure at its next launch and none is left
// spinning against a peer that has given up.  Results after a time-out are garbage by definition; the
// host turns the status into an exception on every rank (no per-rank fallback: a rank that switched to
// RCCL alone would mismatch its peers' collectives).
//
// STATUS: functional tests run with all ranks on ONE GPU (IPC within a device), eager and inside
// HIP-graph replay; it has not run across xGMI (no multi-GPU box available to the build), hence opt-in:
// SP_CUSTOM_ALLREDUCE=1.
#include <cstring>

#include "sp_common.h"

// the fused kernel reproduces sp_fused_add_rmsnorm (elementwise.hip) bit for bit: same rounding points,
// same summation order, no contraction of its products into fmas (this file is also built with
// -ffp-contract=off, scratchpad_amd/build.py)
#pragma clang fp contract(off)

namespace sp {

constexpr int kArMaxRanks = 8;
constexpr int kArMaxBlocks = 128;      // flag rows per region; a launch uses the first gridDim.x of them
constexpr int kArBlocks = 32;          // workgroups of the plain all-reduce (each syncs with its twin on the peers)
constexpr int kArThreads = 512;
constexpr int kFusedThreads = 256;     // one row per workgroup pass, the mapping of rmsnorm_vec_kernel
constexpr int kFusedBlocks = 64;       // default workgroup cap of the fused kernel (see launch_fused)
// words of a region's flag area: [kArMaxBlocks][kArMaxRanks] arrival flags, then one epoch counter per
// workgroup, then the status word
constexpr int kArEpochWord0 = kArMaxBlocks * kArMaxRanks;
constexpr int kArStatusWord = kArEpochWord0 + kArMaxBlocks;

struct ArComm {
  char* region[kArMaxRanks];           // every rank's region as mapped in THIS process
  int rank, world;
  int64_t flag_bytes, data_bytes;      // region layout: [flags | epoch counters | status][data][reduced]
  int64_t timeout_ticks;               // s_memrealtime ticks (100 MHz) a barrier may wait
  uint32_t* host_status;               // pinned host word mapped into the device (may be null)
};

struct ArArgs {
  ArComm c;
  const void* in;
  void* out;
  int64_t n;                           // elements
};

struct ArFusedArgs {
  ArComm c;
  void* x;                             // [T, H] in: this rank's partial sums; out: RMSNorm(residual') * w
  void* residual;                      // [T, H] in/out
  const void* weight;                  // [H]
  int T, H;
  int64_t x_stride, r_stride;          // elements
  float eps;
};

__device__ __forceinline__ void ar_raise(const ArComm& c) {
  // tell everybody: every rank's region (their next launch sees it at entry) and my host
  for (int q = 0; q < c.world; ++q)
    __hip_atomic_store((uint32_t*)c.region[q] + kArStatusWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (c.host_status)
    __hip_atomic_store(c.host_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// entry of every collective kernel: this workgroup's epoch (device state, see the header) and whether the
// communicator has already failed (then no barrier of this launch waits)
__device__ __forceinline__ uint32_t ar_enter(const ArComm& c, uint32_t* s_epoch, int* s_failed) {
  volatile uint32_t* my_counter = (volatile uint32_t*)c.region[c.rank] + kArEpochWord0 + blockIdx.x;
  if (threadIdx.x == 0) {
    *s_epoch = *my_counter + 1u;
    const uint32_t st = __hip_atomic_load((const uint32_t*)c.region[c.rank] + kArStatusWord, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_SYSTEM);
    *s_failed = st != 0u;
    if (st != 0u && c.host_status)
      __hip_atomic_store(c.host_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  return *s_epoch;
}

__device__ __forceinline__ void ar_leave(const ArComm& c, uint32_t epoch) {
  if (threadIdx.x == 0)
    *((volatile uint32_t*)c.region[c.rank] + kArEpochWord0 + blockIdx.x) = epoch + 2u;
}

__device__ __forceinline__ void ar_barrier(const ArComm& c, uint32_t epoch, int* s_failed) {
  // all of this workgroup's earlier writes must be visible system-wide before the flag goes out
  __syncthreads();
  if (threadIdx.x < c.world) {
    const int p = threadIdx.x;
    __threadfence_system();
    uint32_t* theirs = (uint32_t*)c.region[p] + (blockIdx.x * kArMaxRanks + c.rank);
    __hip_atomic_store(theirs, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = (const uint32_t*)c.region[c.rank] + (blockIdx.x * kArMaxRanks + p);
    // epochs only grow: a later epoch from a fast peer also releases us.  The wait is bounded in wall
    // time (a lost peer must not park waves on the GPU for ever); on expiry the failure is published to
    // every rank (ar_raise) and the rest of this launch stops waiting.
    if (!*(volatile int*)s_failed) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) > c.timeout_ticks) {
          ar_raise(c);
          *(volatile int*)s_failed = 1;
          break;
        }
      }
    }
  }
  __syncthreads();
}

template <typename Tag>
__device__ __forceinline__ void accumulate16(const u32x4& v, float* acc) {
  float f[Elem<Tag>::kVec];
  unpack16<Tag>(v, f);
#pragma unroll
  for (int i = 0; i < Elem<Tag>::kVec; ++i) acc[i] += f[i];
}

// n is a multiple of the 16-byte vector; buffers are 16-byte aligned
template <typename Tag, bool TWO_SHOT>
__global__ __launch_bounds__(kArThreads) void all_reduce_kernel(ArArgs a) {
  constexpr int V = Elem<Tag>::kVec;
  const ArComm& c = a.c;
  const int64_t nvec = a.n / V;
  const int64_t tid = (int64_t)blockIdx.x * kArThreads + threadIdx.x, nthr = (int64_t)kArBlocks * kArThreads;
  char* my_data = c.region[c.rank] + c.flag_bytes;
  __shared__ uint32_t s_epoch;
  __shared__ int s_failed;
  const uint32_t epoch = ar_enter(c, &s_epoch, &s_failed);   // the call uses epochs e, e+1, e+2
  // 1. publish my input
  for (int64_t i = tid; i < nvec; i += nthr) st16(my_data + i * 16, ld16((const char*)a.in + i * 16));
  ar_barrier(c, epoch, &s_failed);
  if (!TWO_SHOT) {
    // 2. every rank sums all inputs (fixed rank order: identical bits on every rank)
    for (int64_t i = tid; i < nvec; i += nthr) {
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
      for (int r = 0; r < c.world; ++r) accumulate16<Tag>(ld16(c.region[r] + c.flag_bytes + i * 16), acc);
      st16((char*)a.out + i * 16, pack16<Tag>(acc));
    }
    ar_barrier(c, epoch + 1, &s_failed);   // nobody overwrites its data region while a peer still reads it
    ar_leave(c, epoch);
    return;
  }
  // 2. reduce-scatter: I own vectors [lo, hi)
  const int64_t per = (nvec + c.world - 1) / c.world;
  const int64_t lo = min(per * c.rank, nvec), hi = min(lo + per, nvec);
  char* my_red = c.region[c.rank] + c.flag_bytes + c.data_bytes;
  // vector i is always handled by global thread i % nthr, in every phase and on every rank: the twin-
  // workgroup barrier then orders exactly the accesses that depend on each other
  auto first_at = [&](int64_t from) { return from + ((tid - from) % nthr + nthr) % nthr; };
  for (int64_t i = first_at(lo); i < hi; i += nthr) {
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int r = 0; r < c.world; ++r) accumulate16<Tag>(ld16(c.region[r] + c.flag_bytes + i * 16), acc);
    const u32x4 v = pack16<Tag>(acc);
    st16(my_red + i * 16, v);
    st16((char*)a.out + i * 16, v);
  }
  ar_barrier(c, epoch + 1, &s_failed);
  // 3. all-gather the other ranks' reduced slices
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;
    const int64_t rlo = min(per * r, nvec), rhi = min(rlo + per, nvec);
    const char* red = c.region[r] + c.flag_bytes + c.data_bytes;
    for (int64_t i = first_at(rlo); i < rhi; i += nthr) st16((char*)a.out + i * 16, ld16(red + i * 16));
  }
  ar_barrier(c, epoch + 2, &s_failed);
  ar_leave(c, epoch);
}

// ---------------------------------------------------------------------------------------------------------
// Fused all-reduce + residual add + RMSNorm: what every decoder layer does after o_proj and down_proj under
// tensor parallelism (RowParallelLinear's all-reduce, linear.py:1148-1149, then RMSNorm(x, residual),
// llama.py:222 / 216, layernorm.py:22-32).  Contract, bit for bit the sequence
//     x <- all_reduce(x)  (sp_custom_all_reduce: fp32 sum in rank order, ONE rounding to the dtype)
//     sp_fused_add_rmsnorm(x, residual, w, eps)
// i.e.  s = round(sum_r x_r);  xf = s + residual (fp32);  residual <- round(xf);
//       x <- round(xf * rsqrt(mean(xf^2) + eps)) * w  (rounded), with the norm kernel's summation order.
// What it saves: the all-reduced x is never written to HBM and read back, one launch per layer half goes,
// and in the two-shot form the gather phase moves the bf16 sums while the norm runs on them.
//
// Rows are the unit: row t belongs to workgroup t % gridDim.x on EVERY rank and in EVERY phase (so the
// twin-workgroup barrier orders exactly the dependent accesses), and in the two-shot form to owner rank
// t / ceil(T / world).  One 256-thread workgroup handles a row at a time with the thread -> vector mapping
// and the block reduction of rmsnorm_vec_kernel.
__device__ __forceinline__ float fused_block_sum(float v, float* smem) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < kFusedThreads / 64; ++i) t += smem[i];
  __syncthreads();
  return t;
}

template <typename Tag, bool TWO_SHOT, int MAXIT>
__global__ __launch_bounds__(kFusedThreads) void all_reduce_add_rmsnorm_kernel(ArFusedArgs a) {
  typedef Elem<Tag> E;
  constexpr int V = E::kVec;
  const ArComm& c = a.c;
  const int nvec = a.H / V;                    // 16-byte vectors per row (<= kFusedThreads * MAXIT)
  const int NB = gridDim.x, k = blockIdx.x;
  const int64_t row_b = (int64_t)nvec * 16;    // bytes of a row in the regions (dense)
  char* my_data = c.region[c.rank] + c.flag_bytes;
  char* my_red = my_data + c.data_bytes;
  __shared__ uint32_t s_epoch;
  __shared__ int s_failed;
  __shared__ float smem[kFusedThreads / 64];
  const uint32_t epoch = ar_enter(c, &s_epoch, &s_failed);

  // 1. publish my partial sums (rows of this workgroup)
  for (int t = k; t < a.T; t += NB) {
    const char* src = (const char*)a.x + (int64_t)t * a.x_stride * E::kBytes;
    for (int v = threadIdx.x; v < nvec; v += kFusedThreads)
      st16(my_data + t * row_b + (int64_t)v * 16, ld16(src + (int64_t)v * 16));
  }
  ar_barrier(c, epoch, &s_failed);

  // row t, given where its all-reduced value comes from: residual add + RMSNorm exactly as
  // rmsnorm_vec_kernel<Tag, true> does it, in place on x and residual
  auto finish_row = [&](int t, bool reduce_here, const char* sum_row, char* publish_row) {
    char* xrow = (char*)a.x + (int64_t)t * a.x_stride * E::kBytes;
    char* rrow = (char*)a.residual + (int64_t)t * a.r_stride * E::kBytes;
    float cache[MAXIT][V];
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int v = threadIdx.x + it * kFusedThreads;
      if (v < nvec) {
        u32x4 s16;
        if (reduce_here) {
          float acc[V];
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] = 0.f;
          for (int r = 0; r < c.world; ++r)
            accumulate16<Tag>(ld16(c.region[r] + c.flag_bytes + t * row_b + (int64_t)v * 16), acc);
          s16 = pack16<Tag>(acc);                      // the all-reduce's one rounding
          if (publish_row) st16(publish_row + (int64_t)v * 16, s16);
        } else {
          s16 = ld16(sum_row + (int64_t)v * 16);
        }
        unpack16<Tag>(s16, cache[it]);
        float r[V];
        unpack16<Tag>(ld16(rrow + (int64_t)v * 16), r);
#pragma unroll
        for (int e = 0; e < V; ++e) cache[it][e] = __fadd_rn(cache[it][e], r[e]);
        st16(rrow + (int64_t)v * 16, pack16<Tag>(cache[it]));
#pragma unroll
        for (int e = 0; e < V; ++e) ss = __fadd_rn(ss, __fmul_rn(cache[it][e], cache[it][e]));
      }
    }
    const float mean = fused_block_sum(ss, smem) / (float)a.H;
    const float rs = 1.0f / sqrtf(mean + a.eps);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int v = threadIdx.x + it * kFusedThreads;
      if (v < nvec) {
        float w[V], y[V];
        unpack16<Tag>(ld16((const char*)a.weight + (int64_t)v * 16), w);
#pragma unroll
        for (int e = 0; e < V; ++e) y[e] = __fmul_rn(E::round(__fmul_rn(cache[it][e], rs)), w[e]);
        st16(xrow + (int64_t)v * 16, pack16<Tag>(y));
      }
    }
  };

  if (!TWO_SHOT) {
    // 2. every rank sums every row itself (fixed rank order: identical bits on every rank)
    for (int t = k; t < a.T; t += NB) finish_row(t, true, nullptr, nullptr);
    ar_barrier(c, epoch + 1, &s_failed);   // nobody overwrites its data region while a peer still reads it
    ar_leave(c, epoch);
    return;
  }
  // 2. reduce-scatter by rows: I own rows [lo, hi); their sums go to my `reduced` area for the peers
  const int per = (a.T + c.world - 1) / c.world;
  const int lo = min(per * c.rank, a.T), hi = min(lo + per, a.T);
  for (int t = lo + ((k - lo) % NB + NB) % NB; t < hi; t += NB) finish_row(t, true, nullptr, my_red + t * row_b);
  ar_barrier(c, epoch + 1, &s_failed);
  // 3. the other ranks' rows: gather the sum from its owner and finish the row locally
  for (int t = k; t < a.T; t += NB) {
    if (t >= lo && t < hi) continue;
    const int owner = t / per;
    finish_row(t, false, c.region[owner] + c.flag_bytes + c.data_bytes + t * row_b, nullptr);
  }
  ar_barrier(c, epoch + 2, &s_failed);
  ar_leave(c, epoch);
}

}  // namespace sp

using namespace sp;

// flags [blocks][ranks] + epoch counters [blocks] + one 'a barrier timed out' word, padded to 256 bytes
extern "C" size_t sp_ar_flag_bytes(void) { return (((size_t)kArStatusWord + 1) * sizeof(uint32_t) + 255) / 256 * 256; }

// status word of a rank's OWN region: 0 = every barrier so far completed, 1 = one timed out somewhere in
// the group (results since then are garbage).  Synchronises with the device (a small copy): for check
// points; the call path reads the host status word instead (sp_ar_host_status_alloc).
extern "C" int sp_ar_status(const void* own_region, int* status) {
  SP_CHECK_ARG(own_region && status);
  uint32_t w = 0;
  if (hipMemcpy(&w, (const uint32_t*)own_region + kArStatusWord, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
    return SP_ERR_LAUNCH;
  *status = (int)w;
  return SP_OK;
}

// one pinned, device-mapped host word: the kernels raise it on a time-out, the host reads *host_ptr as
// plain memory (no stream synchronisation) at every forward boundary
extern "C" int sp_ar_host_status_alloc(void** host_ptr, void** device_ptr) {
  SP_CHECK_ARG(host_ptr && device_ptr);
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return SP_ERR_LAUNCH;
  memset(h, 0, 64);
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return SP_ERR_LAUNCH; }
  *host_ptr = h;
  *device_ptr = d;
  return SP_OK;
}

extern "C" int sp_ar_host_status_free(void* host_ptr) {
  return hipHostFree(host_ptr) == hipSuccess ? SP_OK : SP_ERR_LAUNCH;
}

extern "C" int sp_ar_alloc(void** ptr, size_t bytes) {
  SP_CHECK_ARG(ptr && bytes > 0);
  // fine-grained: stores become visible to peer devices without a kernel boundary
  if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained) != hipSuccess) return SP_ERR_LAUNCH;
  if (hipMemset(*ptr, 0, bytes) != hipSuccess) return SP_ERR_LAUNCH;
  if (hipDeviceSynchronize() != hipSuccess) return SP_ERR_LAUNCH;
  return SP_OK;
}

extern "C" int sp_ar_free(void* ptr) { return hipFree(ptr) == hipSuccess ? SP_OK : SP_ERR_LAUNCH; }

extern "C" int sp_ar_ipc_export(void* ptr, void* handle64) {
  SP_CHECK_ARG(ptr && handle64);
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, ptr) != hipSuccess) return SP_ERR_LAUNCH;
  memcpy(handle64, &h, sizeof(h));
  return SP_OK;
}

extern "C" int sp_ar_ipc_import(const void* handle64, void** ptr) {
  SP_CHECK_ARG(ptr && handle64);
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return SP_ERR_LAUNCH;
  return SP_OK;
}

extern "C" int sp_ar_ipc_close(void* ptr) { return hipIpcCloseMemHandle(ptr) == hipSuccess ? SP_OK : SP_ERR_LAUNCH; }

static int fill_comm(ArComm& c, void* const* regions, int rank, int world, size_t data_bytes,
                     int64_t timeout_us, void* host_status) {
  SP_CHECK_ARG(regions && world >= 2 && world <= kArMaxRanks && rank >= 0 && rank < world);
  for (int r = 0; r < world; ++r) {
    SP_CHECK_ARG(regions[r] != nullptr);
    c.region[r] = (char*)regions[r];
  }
  c.rank = rank; c.world = world;
  c.flag_bytes = (int64_t)sp_ar_flag_bytes(); c.data_bytes = (int64_t)data_bytes;
  if (timeout_us <= 0) timeout_us = 30ll * 1000 * 1000;
  c.timeout_ticks = timeout_us * 100;            // s_memrealtime: 100 MHz
  c.host_status = (uint32_t*)host_status;
  return SP_OK;
}

// regions: host array of `world` pointers (every rank's region as mapped here; regions[rank] is my own).
// Calls are collective: every rank issues the same sequence (the device-side epoch counters advance
// in lock step).  Graph-capturable.  timeout_us <= 0: 30 s; host_status: device pointer of the word from
// sp_ar_host_status_alloc, or null.
extern "C" int sp_custom_all_reduce(void* out, const void* in, int64_t num_elems, int dtype,
                                    void* const* regions, int rank, int world, size_t data_bytes,
                                    int64_t timeout_us, void* host_status, void* stream) {
  SP_CHECK_ARG(out && in && num_elems >= 0);
  ArArgs a;
  const int rc = fill_comm(a.c, regions, rank, world, data_bytes, timeout_us, host_status);
  if (rc != SP_OK) return rc;
  if (num_elems == 0) return SP_OK;
  const int eb = dtype == SP_F32 ? 4 : 2;
  if (dtype != SP_F32 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  if ((num_elems * eb) % 16 || ((uintptr_t)out & 15) || ((uintptr_t)in & 15)) return SP_ERR_UNSUPPORTED;
  if ((size_t)num_elems * eb > data_bytes) return SP_ERR_WORKSPACE;
  a.in = in; a.out = out; a.n = num_elems;
  const bool two_shot = (size_t)num_elems * eb > (256u << 10);
  hipStream_t st = (hipStream_t)stream;
#define SP_AR_LAUNCH(TWO)                                                                    \
  SP_DISPATCH_DTYPE(dtype, (all_reduce_kernel<Tag, TWO><<<dim3(kArBlocks), kArThreads, 0, st>>>(a)))
  if (two_shot) { SP_AR_LAUNCH(true); } else { SP_AR_LAUNCH(false); }
#undef SP_AR_LAUNCH
  SP_LAUNCH_CHECK();
  return SP_OK;
}

static int g_fused_blocks = 0;     // sp_debug_set("ar_fused_blocks", n): tuning knob, 0 = default
namespace sp { void set_ar_fused_blocks(int n) { g_fused_blocks = n < 0 ? 0 : (n > kArMaxBlocks ? kArMaxBlocks : n); } }

template <typename Tag>
static int launch_fused(const ArFusedArgs& a, bool two_shot, hipStream_t st) {
  constexpr int MAXIT = (Elem<Tag>::kBytes == 4) ? 8 : 4;
  // one workgroup per row up to kFusedBlocks, then several rows per workgroup.  Measured with the ranks of ONE
  // GPU (tools/bench_allreduce.py, [128, 8192] bf16, 2 / 4 ranks, us per call): 128 workgroups 73 / 207,
  // 64: 37, 32: 39 / 95, 16: 55 - every barrier is a system-scope fence per workgroup, and those serialise;
  // all-reduce + norm as two launches: 33 / 122
  // (4 ranks: 32 workgroups 91 - 95, 64: 140 - the fences of more ranks serialise harder - so the cap halves above 2 ranks)
  const int cap = g_fused_blocks > 0 ? g_fused_blocks : (a.c.world <= 2 ? kFusedBlocks : kFusedBlocks / 2);
  const int blocks = a.T < cap ? a.T : cap;
  if (two_shot)
    all_reduce_add_rmsnorm_kernel<Tag, true, MAXIT><<<dim3(blocks), kFusedThreads, 0, st>>>(a);
  else
    all_reduce_add_rmsnorm_kernel<Tag, false, MAXIT><<<dim3(blocks), kFusedThreads, 0, st>>>(a);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// x [T, H]: this rank's partial sums in, RMSNorm(residual') * w out; residual [T, H] in/out (see the kernel).
// Every rank passes the same T, H, dtype, eps and weights.  SP_ERR_UNSUPPORTED for shapes the vector
// kernel does not take (the caller then runs all-reduce + sp_fused_add_rmsnorm): H not a multiple of the
// 16-byte vector or above 8192 (16-bit) / 8192 (fp32) elements, unaligned rows.
extern "C" int sp_fused_allreduce_add_rmsnorm(void* x, void* residual, const void* weight, int64_t num_tokens,
                                              int hidden, int64_t x_stride, int64_t res_stride, float eps,
                                              int dtype, void* const* regions, int rank, int world,
                                              size_t data_bytes, int64_t timeout_us, void* host_status,
                                              void* stream) {
  SP_CHECK_ARG(num_tokens >= 0 && hidden > 0);
  ArFusedArgs a;
  const int rc = fill_comm(a.c, regions, rank, world, data_bytes, timeout_us, host_status);
  if (rc != SP_OK) return rc;
  if (num_tokens == 0) return SP_OK;
  SP_CHECK_ARG(x && residual && weight);
  if (dtype != SP_F32 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  const int eb = dtype == SP_F32 ? 4 : 2, V = 16 / eb, maxit = dtype == SP_F32 ? 8 : 4;
  if (hidden % V || x_stride % V || res_stride % V || hidden / V > kFusedThreads * maxit) return SP_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 15) || ((uintptr_t)residual & 15) || ((uintptr_t)weight & 15)) return SP_ERR_UNSUPPORTED;
  if (num_tokens > 0x7fffffffLL / hidden) return SP_ERR_INVALID_ARG;
  const size_t bytes = (size_t)num_tokens * hidden * eb;
  if (bytes > data_bytes) return SP_ERR_WORKSPACE;
  a.x = x; a.residual = residual; a.weight = weight; a.T = (int)num_tokens; a.H = hidden;
  a.x_stride = x_stride; a.r_stride = res_stride; a.eps = eps;
  // one-shot while the whole message is small or the batch has fewer rows than ranks
  const bool two_shot = bytes > (256u << 10) && num_tokens >= world;
  SP_DISPATCH_DTYPE(dtype, return (launch_fused<Tag>(a, two_shot, (hipStream_t)stream)));
}
