// Direct all-reduce through IPC-mapped peer buffers: the `ca_comm` seam of GroupCoordinator
// (distributed/parallel_state.py:266-267, 326-347: should_custom_ar / custom_all_reduce), which the
// reference declares and never fills - its per-layer [T, hidden] SUM all-reduce (linear.py:1148-1149)
// always goes through NCCL.
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
// A barrier that times out (lost or slow peer) raises the region's status word; the host reads it
// through sp_ar_status() (CustomAllReduce.check(): in debug mode after every call, always at close()
// and after a timed run) and then abandons the communicator for RCCL.
//
// STATUS: functional tests run with all ranks on ONE GPU (IPC within a device), eager and inside
// HIP-graph replay; it has not run across xGMI (no multi-GPU box available to the build), hence opt-in:
// SP_CUSTOM_ALLREDUCE=1.
#include <cstring>

#include "sp_common.h"

namespace sp {

constexpr int kArMaxRanks = 8;
constexpr int kArBlocks = 32;          // workgroups per launch (each syncs with its twin on the peers)
constexpr int kArThreads = 512;
// words of a region's flag area: [kArBlocks][kArMaxRanks] arrival flags, then one epoch counter per
// workgroup, then the status word
constexpr int kArEpochWord0 = kArBlocks * kArMaxRanks;
constexpr int kArStatusWord = kArEpochWord0 + kArBlocks;

struct ArArgs {
  char* region[kArMaxRanks];           // every rank's region as mapped in THIS process
  const void* in;
  void* out;
  int64_t n;                           // elements
  int rank, world;
  int64_t flag_bytes, data_bytes;      // region layout: [flags | epoch counters | status][data][reduced]
};

__device__ __forceinline__ void ar_barrier(const ArArgs& a, uint32_t epoch) {
  // all of this workgroup's earlier writes must be visible system-wide before the flag goes out
  __syncthreads();
  if (threadIdx.x < a.world) {
    const int p = threadIdx.x;
    __threadfence_system();
    uint32_t* theirs = (uint32_t*)a.region[p] + (blockIdx.x * kArMaxRanks + a.rank);
    __hip_atomic_store(theirs, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* mine = (const uint32_t*)a.region[a.rank] + (blockIdx.x * kArMaxRanks + p);
    // epochs only grow: a later epoch from a fast peer also releases us.  The spin is bounded (a lost
    // peer must not park waves on the GPU for ever): on expiry the result is garbage and the last flag
    // word of the region is raised for the host to see.
    long spins = 0;
    while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1l << 24)) {        // ~ seconds
        ((volatile uint32_t*)a.region[a.rank])[kArStatusWord] = 1u;
        break;
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
  const int64_t nvec = a.n / V;
  const int64_t tid = (int64_t)blockIdx.x * kArThreads + threadIdx.x, nthr = (int64_t)kArBlocks * kArThreads;
  char* my_data = a.region[a.rank] + a.flag_bytes;
  // this workgroup's call counter (device state, see the header): the call uses epochs e, e+1, e+2
  __shared__ uint32_t s_epoch;
  volatile uint32_t* my_counter = (volatile uint32_t*)a.region[a.rank] + kArEpochWord0 + blockIdx.x;
  if (threadIdx.x == 0) s_epoch = *my_counter + 1u;
  __syncthreads();
  const uint32_t epoch = s_epoch;
  // 1. publish my input
  for (int64_t i = tid; i < nvec; i += nthr) st16(my_data + i * 16, ld16((const char*)a.in + i * 16));
  ar_barrier(a, epoch);
  if (!TWO_SHOT) {
    // 2. every rank sums all inputs (fixed rank order: identical bits on every rank)
    for (int64_t i = tid; i < nvec; i += nthr) {
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
      for (int r = 0; r < a.world; ++r) accumulate16<Tag>(ld16(a.region[r] + a.flag_bytes + i * 16), acc);
      st16((char*)a.out + i * 16, pack16<Tag>(acc));
    }
    ar_barrier(a, epoch + 1);          // nobody overwrites its data region while a peer still reads it
    if (threadIdx.x == 0) *my_counter = epoch + 2u;
    return;
  }
  // 2. reduce-scatter: I own vectors [lo, hi)
  const int64_t per = (nvec + a.world - 1) / a.world;
  const int64_t lo = min(per * a.rank, nvec), hi = min(lo + per, nvec);
  char* my_red = a.region[a.rank] + a.flag_bytes + a.data_bytes;
  // vector i is always handled by global thread i % nthr, in every phase and on every rank: the twin-
  // workgroup barrier then orders exactly the accesses that depend on each other
  auto first_at = [&](int64_t from) { return from + ((tid - from) % nthr + nthr) % nthr; };
  for (int64_t i = first_at(lo); i < hi; i += nthr) {
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int r = 0; r < a.world; ++r) accumulate16<Tag>(ld16(a.region[r] + a.flag_bytes + i * 16), acc);
    const u32x4 v = pack16<Tag>(acc);
    st16(my_red + i * 16, v);
    st16((char*)a.out + i * 16, v);
  }
  ar_barrier(a, epoch + 1);
  // 3. all-gather the other ranks' reduced slices
  for (int r = 0; r < a.world; ++r) {
    if (r == a.rank) continue;
    const int64_t rlo = min(per * r, nvec), rhi = min(rlo + per, nvec);
    const char* red = a.region[r] + a.flag_bytes + a.data_bytes;
    for (int64_t i = first_at(rlo); i < rhi; i += nthr) st16((char*)a.out + i * 16, ld16(red + i * 16));
  }
  ar_barrier(a, epoch + 2);
  if (threadIdx.x == 0) *my_counter = epoch + 2u;
}

}  // namespace sp

using namespace sp;

// flags [blocks][ranks] + epoch counters [blocks] + one 'a barrier timed out' word, padded to 256 bytes
extern "C" size_t sp_ar_flag_bytes(void) { return (((size_t)kArStatusWord + 1) * sizeof(uint32_t) + 255) / 256 * 256; }

// status word of a rank's OWN region: 0 = every barrier so far completed, 1 = one timed out (results
// since then are garbage).  Synchronises with the device (a small copy): not for the call path.
extern "C" int sp_ar_status(const void* own_region, int* status) {
  SP_CHECK_ARG(own_region && status);
  uint32_t w = 0;
  if (hipMemcpy(&w, (const uint32_t*)own_region + kArStatusWord, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
    return SP_ERR_LAUNCH;
  *status = (int)w;
  return SP_OK;
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

// regions: host array of `world` pointers (every rank's region as mapped here; regions[rank] is my own).
// Calls are collective: every rank issues the same sequence (the device-side epoch counters advance
// in lock step).  Graph-capturable.
extern "C" int sp_custom_all_reduce(void* out, const void* in, int64_t num_elems, int dtype,
                                    void* const* regions, int rank, int world, size_t data_bytes,
                                    void* stream) {
  SP_CHECK_ARG(out && in && regions && num_elems >= 0 && world >= 2 && world <= kArMaxRanks);
  SP_CHECK_ARG(rank >= 0 && rank < world);
  if (num_elems == 0) return SP_OK;
  const int eb = dtype == SP_F32 ? 4 : 2;
  if (dtype != SP_F32 && dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  if ((num_elems * eb) % 16 || ((uintptr_t)out & 15) || ((uintptr_t)in & 15)) return SP_ERR_UNSUPPORTED;
  if ((size_t)num_elems * eb > data_bytes) return SP_ERR_WORKSPACE;
  ArArgs a;
  for (int r = 0; r < world; ++r) a.region[r] = (char*)regions[r];
  a.in = in; a.out = out; a.n = num_elems; a.rank = rank; a.world = world;
  a.flag_bytes = (int64_t)sp_ar_flag_bytes(); a.data_bytes = (int64_t)data_bytes;
  const bool two_shot = (size_t)num_elems * eb > (256u << 10);
  hipStream_t st = (hipStream_t)stream;
#define SP_AR_LAUNCH(TWO)                                                                    \
  SP_DISPATCH_DTYPE(dtype, (all_reduce_kernel<Tag, TWO><<<dim3(kArBlocks), kArThreads, 0, st>>>(a)))
  if (two_shot) { SP_AR_LAUNCH(true); } else { SP_AR_LAUNCH(false); }
#undef SP_AR_LAUNCH
  SP_LAUNCH_CHECK();
  return SP_OK;
}
