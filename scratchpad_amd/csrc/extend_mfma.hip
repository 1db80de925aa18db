// Ragged extend (prefill with cached prefix / cross-attention) on the matrix cores - 16-bit dtypes.
//
// Replaces extend_attention_fwd (nn/attention/triton_attn/extend_attention.py:16-327) and the
// flashinfer ragged + paged + merge_state path (nn/attention/flashinfer_backend.py:400-444): one
// kernel, every key (cached prefix AND the new tokens, which the KV store wrote just before) is
// gathered from the paged pool through req_to_token.
//
// Tiling (cdna_hip_programming.md section 3 and Appendix B 'Fused attention prefill'):
//   * workgroup = NW waves = (request, kv head, block of BM new tokens); a wave owns 32 query rows of
//     ONE query head; the waves cover Gk query heads x NW/Gk row blocks, so a K/V tile staged once in
//     LDS is shared by the whole GQA group (8 waves: 256 (row, head) pairs per staged tile);
//   * S^T = K . Q^T with v_mfma_f32_32x32x16 (A = K rows from LDS by ds_read_b128, B = Q fragments held
//     in registers for the whole kernel): the accumulator then has the query row on the LANE and
//     the 16 keys of the lane half in registers, so the softmax row max is in-lane plus one exchange
//     with lane^32, the row sum stays a per-lane partial until the epilogue, and the rescale factor
//     of O^T is lane-local;
//   * O^T = V^T . P^T: P^T is taken straight from the S^T accumulator registers as the B operand
//     ("an accumulator tile as the next MFMA's operand": registers 8s..8s+7 -> k-step s, key order
//     16s + 8(j>>2) + 4h + (j&3)); the matching V^T fragments come from the row-major V tile by
//     ds_read_b64_tr_b16 (hardware transpose), two reads per k-step;
//   * K/V tiles of 64 keys: gathered rows (full 256-B lines, 16 B per lane) -> registers ->
//     LDS with padded row strides (K +16 B: conflict-free ds_read_b128; V +64 B: conflict-free
//     tr reads); the next tile's global loads are issued before the current tile is consumed;
//   * online softmax in the exp2 domain with the scale folded into the exponent's fma; the running
//     maximum is only advanced when some row's maximum grew by more than kDeferLog2 (the O^T rescale
//     is then a rare, wave-uniform branch; P stays <= 2^kDeferLog2);
//   * grid: x = kv head (fastest: the hardware deals consecutive workgroups round-robin to the 8
//     XCDs, so with 8 kv heads every row block of one (request, kv head) runs on ONE XCD and its
//     K/V re-reads hit that XCD's L2: FETCH_SIZE per launch 5.7 -> 1.8 GB), y = the (request, row
//     block) items of sp_extend_plan, heaviest first (without a plan: every row block of every request,
//     last rows of the prompt first).
#include "extend_internal.h"

namespace sp {

// Diagnostic build only (-DSP_EXTEND_STAMPS, tools/build_stamps.sh; never in the shipped library):
// s_memtime stamps between the segments of a tile iteration, summed per wave and added to a
// caller-supplied buffer, to see where an iteration spends its cycles (cdna_hip_programming.md
// section 7 'In-kernel stamps').
#if defined(SP_EXTEND_STAMPS) || defined(SP_EXTEND_WGSTAMPS)
__device__ unsigned long long* g_stamp_buf = nullptr;
#endif
// Second diagnostic form (-DSP_EXTEND_WGSTAMPS): the timeline of a whole workgroup - entry, metadata decoded,
// first tile landed, tile loop done, output written - summed over workgroups (wave 0), same buffer.
#ifdef SP_EXTEND_WGSTAMPS
#define SP_WGSTAMP(var)                                                                     \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#else
#define SP_WGSTAMP(var) do { } while (0)
#endif
#ifdef SP_EXTEND_STAMPS
#define SP_STAMP(var)                                                                       \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define SP_STAMP_ACC(slot, from, to) stamp_sum[slot] += (to) - (from)
#else
#define SP_STAMP(var) do { } while (0)
#define SP_STAMP_ACC(slot, from, to) do { } while (0)
#endif

// PLAIN: no logit cap and no sliding window (both compiled out of the tile loop)
template <typename Tag, int D, int GK, int NW, bool KV8, bool PLAIN, bool DMA>
__global__ __launch_bounds__(NW * 64, 2) void extend_mfma_kernel(ExtendArgs a) {
  static_assert(!DMA || ((D == 128 || D == 64) && !KV8 && NW == 8), "the LDS-DMA images are laid out for 256- / 128-byte rows and 8 waves");
  constexpr int NT = NW * 64;
  constexpr int NB = 2;                // LDS tile buffers: tile t lives in buffer t & 1
  typedef ExtCfg<D, NT> C;
  typedef typename std::conditional<KV8, f16_tag, Tag>::type CT;      // dtype of the tile math
  typedef typename std::conditional<KV8, u32x2, u32x4>::type raw_t;   // one thread's gathered chunk
  constexpr int BN = C::BN, SK = C::SK, SV = C::SV, CPR = C::CPR, RPP = C::RPP, PASSES = C::PASSES;
  constexpr int KSTEPS = C::KSTEPS, DBLK = C::DBLK;
  constexpr int BM = 32 * (NW / GK);
  extern __shared__ __attribute__((aligned(16))) char lds[];
#ifdef SP_EXTEND_WGSTAMPS
  unsigned long long wg0 = 0, wg1 = 0, wg2 = 0, wg3 = 0, wg4 = 0;
  SP_WGSTAMP(wg0);
#endif

  int b, row0;
  if (a.plan) {                                    // planned: items are sorted heaviest first
    const bool mine = a.plan[1] == BM && a.plan[2] == a.Hq && a.plan[3] == a.Hkv && a.plan[4] == a.num_tokens &&
                      a.plan[5] == a.bs;
    if (mine) {
      if ((int)blockIdx.y >= a.plan[0]) return;
      b = a.plan[kExtPlanHeader + 2 * blockIdx.y];
      row0 = a.plan[kExtPlanHeader + 1 + 2 * blockIdx.y] * BM;
    } else {
      // a plan built for another block size, other head counts or another step: walk the requests for
      // grid row blockIdx.y (the grid has a row for every item of THIS step); same results, no ordering
      int rem = blockIdx.y, nblk = 0;
      for (b = 0; b < a.bs; ++b) {
        const int e = a.ext_lens[b];
        nblk = e > 0 ? (e + BM - 1) / BM : 0;
        if (rem < nblk) break;
        rem -= nblk;
      }
      if (b >= a.bs) return;
      row0 = (nblk - 1 - rem) * BM;
    }
  } else {
    b = blockIdx.z;
    row0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * BM;    // heaviest row blocks first
  }
  const int E = a.ext_lens[b];
  if (row0 >= E) return;
  const int G = a.Hq / a.Hkv;
  const int halves = G / GK;                       // query-head blocks per kv head
  const int hk = blockIdx.x / halves;
  const int hh = blockIdx.x - hk * halves;
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
  const bool windowed = !PLAIN && a.causal && a.window >= 0;
  const int tbeg = windowed ? max(0, P + row0 - a.window) / BN : 0;
  SP_WGSTAMP(wg1);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c = lane & 31, h = lane >> 5;
  const int g = wave % GK, rb = wave / GK;
  const int head = hk * G + hh * GK + g;
  const int r0 = row0 + rb * 32;                   // this wave's first query row
  const int my_row = r0 + c;                       // this lane's query row (may be >= E: masked out)
  const bool wave_live = r0 < E;
  if (ntiles <= tbeg) {
    // no key at all (a request without encoder tokens): zeros, and no look at req_to_token, whose row
    // need not hold a valid slot then.  Workgroup-uniform: taken before the first barrier.
    if (my_row < E && wave_live) {
      char* op = (char*)a.out + ((t0 + my_row) * a.o_stride + (int64_t)head * D) * 2;
      for (int d = 4 * h; d < D; d += 8) *(u32x2*)(op + d * 2) = u32x2{0u, 0u};
    }
    return;
  }

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
  // DMA: the Q loads stay ahead of the first DMA pieces, so that the prologue's counted wait covers them
  if constexpr (DMA) __builtin_amdgcn_sched_barrier(0);
  const float cap = PLAIN ? 0.f : a.logit_cap;
  // scores enter the exponent as exp2(s * sc - m): sc carries the softmax scale (the capped form
  // rewrites s into the log2 domain first, so its sc is 1)
  const float sc = cap > 0.f ? 1.0f : a.sm_scale * kLog2eX;
  const int row_limit = a.causal ? P + my_row : 0x7fffffff;   // last visible key index
  const int row_first = windowed ? P + my_row - a.window : 0;  // first visible key index (may be < 0)

  f32x16 oacc[DBLK];
#pragma unroll
  for (int db = 0; db < DBLK; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = kNegBigX;   // running row maximum (log2 domain), identical in both lane halves
  float l_part = 0.f;       // this lane's part of the row sum (its 32 keys of every tile)

  // ---- staging: thread -> (row, 16-byte chunk) of the tile, PASSES rows each for K and V
  const int st_row = tid / CPR, st_ch = tid % CPR;
  const int64_t tok_bytes = a.kv_stride * (KV8 ? 1 : 2);
  const int64_t head_off = (int64_t)hk * D * (KV8 ? 1 : 2) + st_ch * (KV8 ? 8 : 16);
  // Two register sets of gathered rows, two tiles in flight: set X holds tile t+1 while tile t is
  // consumed from LDS and set Y is being filled with tile t+2 (in-kernel stamps showed a gather that
  // is issued one tile ahead still waited for ~3,300 cycles per tile: the 32 workgroups of an XCD
  // walk the same keys in step, so every tile's first touch is an HBM miss for all of them at once).
  // Slot indices of a set's NEXT tile are fetched right after its gathers are issued, so the
  // dependent req_to_token -> row chain is never waited for inside the loop.
  struct RegSet {
    raw_t k[PASSES], v[PASSES];
    int slot[PASSES];
  };
  RegSet setA, setB;
  auto fetch_slots = [&](RegSet& r, int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      // rows past the end re-read the last valid key (masked in the softmax; real, finite K/V so a
      // zero probability times its V row is zero).  No conditional load and no select on the loaded
      // value: a load the wave may skip makes the compiler's count of outstanding loads ambiguous
      // (every later wait degrades to vmcnt(0), which also waits for the youngest gathers), and a
      // select on the index is scheduled at the END of the iteration that issued the load.
      const int key = tile * BN + p * RPP + st_row;
      r.slot[p] = idx_row[max(min(key, kv_len - 1), 0)];
    }
  };
  auto prefetch = [&](RegSet& r, int tile) __attribute__((always_inline)) {   // gathers of `tile` from r.slot, then the indices of tile+2
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int64_t off = (int64_t)r.slot[p] * tok_bytes + head_off;
      r.k[p] = *(const raw_t*)(a.kbuf + off);
      r.v[p] = *(const raw_t*)(a.vbuf + off);
    }
    fetch_slots(r, tile + 2);
  };
  auto stage = [&](const RegSet& r, int buf) __attribute__((always_inline)) {
    char* dK = lds + buf * C::kTileBytes;
    char* dV = dK + BN * SK;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int row = p * RPP + st_row;
      if constexpr (KV8) {
        st16(dK + row * SK + st_ch * 16, expand_e5m2x8(r.k[p]));
        st16(dV + row * SV + st_ch * 16, expand_e5m2x8(r.v[p]));
      } else {
        st16(dK + row * SK + st_ch * 16, r.k[p]);
        st16(dV + row * SV + st_ch * 16, r.v[p]);
      }
    }
  };

  // tr-read address pieces: lane i of a 16-lane group supplies row (i>>2), columns 4*(i&3)..+3
  const int i16 = lane & 15, g16 = lane >> 4;
  const int tr_rowq = i16 >> 2, tr_col = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;  // bytes
  // LDS-DMA image (DMA): rows of ROW_B = 256 B (D = 128) or 128 B (D = 64) without padding (a DMA piece is 1 KiB of
  // consecutive LDS = 4 or 8 rows), made conflict-free by the SOURCE address each lane asks for, i.e. row r keeps its
  // 16-byte chunk c at position c ^ f(r):
  //   D = 128   K: f = r & 15 (a ds_read_b128 serves 16 lanes = 16 different rows mod 16 at a time)
  //             V: f = (r & 3) << 2 (a transposed read serves 4 rows x 64 B per 32 lanes)
  //   D = 64    rows r and r + 2 share their banks (256 B wrap), so the row PAIR index drives the swizzle:
  //             K: f = (r >> 1) & 7 (the 16 lanes of a ds_read_b128 group hold 8 even and 8 odd rows whose pair
  //                indices are all different), V: f = ((r >> 1) & 1) << 2 (the 4 rows of a transposed read land on
  //                the four 64-byte quarters of the bank space)
  // Four K and four V buffers (D = 128: 128 KiB; D = 64: 64 KiB).
  constexpr int kRowB = 2 * D;                       // bytes of a K / V row in LDS
  constexpr int kDmaTile = 64 * kRowB;
  constexpr int kDmaBufs = 4;   // tile t is consumed from buffer t % 4 while t+1, t+2 and t+3 are in flight
  constexpr int kCPR = kRowB / 16;                   // 16-byte chunks per row (16 / 8)
  constexpr int kRPP = 64 / kCPR;                    // rows per DMA piece (4 / 8)
  constexpr int kPPW = 8 / kRPP;                     // pieces per wave and tile, each for K and for V (2 / 1)
  const int k_swz = D == 128 ? ((c & 15) ^ h) : (((c >> 1) & 7) ^ h);
  // pipeline: the gathers of tiles t+1 and t+2 fly during compute(t); tile t+1 is written to the other
  // LDS buffer right after compute(t), and one barrier per tile both publishes tile t+1 and retires
  // every wave's reads of the buffer that tile t+2 will overwrite.
#ifdef SP_EXTEND_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
#endif
  // S^T, mask, online softmax of tile t from LDS buffer `buf`; leaves P^T packed in pf (B operands of
  // the four 16-key k-steps).  Returns false when the wave has no visible key in the tile.
  auto qk_softmax = [&](int t, int buf, u32x4 (&pf)[4]) __attribute__((always_inline)) -> bool {
    const char* ldsK = DMA ? lds + buf * kDmaTile : lds + buf * C::kTileBytes;
    const int key0 = t * BN;
    // a wave skips tiles that lie entirely above its rows' diagonal (wave-uniform)
    const bool visible = wave_live && (!a.causal || key0 <= P + min(r0 + 31, E - 1)) &&
                         (!windowed || key0 + BN - 1 >= P + r0 - a.window);
    if (!visible) return false;
    // ---- S^T = K . Q^T for the two 32-key blocks of the tile
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      const char* kp = DMA ? ldsK + kb * 32 * kRowB + c * kRowB : ldsK + (kb * 32 + c) * SK + h * 16;
      // all fragment reads of the block are issued before its first MFMA (distinct registers), so
      // one LDS latency is exposed per block instead of one per MFMA; the sched_barrier keeps the
      // scheduler from sinking each read next to its MFMA again (it then reuses one register quad
      // and waits lgkmcnt(0) before every MFMA)
      u32x4 kf[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) kf[ks] = ld16(DMA ? kp + (((2 * ks) ^ k_swz) << 4) : kp + ks * 32);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) s[kb] = mfma32<CT>(kf[ks], qf[ks], s[kb]);
    }
    SP_STAMP(ts2);
    // ---- mask (edge tiles only), online softmax (query row on the lane; keys in registers + lane^32)
    if constexpr (!PLAIN) {
      if (cap > 0.f) {   // wave-uniform: rewrite the scores as cap * tanh(s * scale / cap) in log2 units
        const float pre = a.sm_scale / cap, post = cap * kLog2eX;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] = post * tanhf(s[kb][r] * pre);
      }
    }
    // interior tiles (entirely below every row's diagonal and inside the key range) need no mask
    const bool need_mask = key0 + BN > kv_len || (a.causal && key0 + BN - 1 > P + r0) ||
                           (windowed && key0 < P + min(r0 + 31, E - 1) - a.window);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool vis = key < kv_len && key <= row_limit && (PLAIN || key >= row_first);
          s[kb][r] = vis ? s[kb][r] : -INFINITY;
        }
    }
#ifdef SP_X_NOMAX          // DIAGNOSTIC (wrong results): what the row-maximum tree of a tile costs
    float mx = fmaxf(s[0][0], s[1][15]);
#else
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
#endif
    mx = fmaxf(mx, xchg32(mx));
    const float m_cand = mx * sc;               // sc > 0: the maximum commutes with the scale
    if (__any(m_cand > m_run + a.defer)) {      // rare once the row maxima have settled
      const float m_new = fmaxf(m_run, m_cand);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_part *= alpha;
#pragma unroll
      for (int db = 0; db < DBLK; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#if defined(SP_X_NOEXP)   // DIAGNOSTIC (wrong results): what the 32 v_exp of a tile cost
        const float p = __builtin_fmaf(s[kb][r], sc, -m_run);
#elif defined(SP_X_NOFMAEXP)   // DIAGNOSTIC (wrong results): neither the fma nor the exp
        const float p = s[kb][r];
#else
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], sc, -m_run));
#endif
        s[kb][r] = p;
#ifndef SP_X_NOADD        // DIAGNOSTIC (wrong results): what the 32 row-sum adds of a tile cost
        psum += p;
#endif
      }
    l_part += psum;
    // B operands of O^T += V^T . P^T: registers 8s..8s+7 of an S^T block, rounded to the KV dtype
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          pf[kb * 2 + sidx][j] = pack2<CT>(s[kb][8 * sidx + 2 * j], s[kb][8 * sidx + 2 * j + 1]);
    SP_STAMP(ts3);
    return true;
  };
  // ---- O^T += V^T . P^T with the V tile of LDS buffer `buf`
  auto pv = [&](int buf, const u32x4 (&pf)[4]) __attribute__((always_inline)) {
    if constexpr (DMA) {
      // The transposed reads are issued as asm text: hipcc puts s_waitcnt vmcnt(0) in front of every
      // ds_read_tr INTRINSIC while LDS-DMA is in flight (it cannot tell that the pieces in flight go to
      // another buffer), which would drain the two tiles of prefetch once per tile.  Their arrival
      // is therefore counted by hand: the reads of k-step s+1 are issued before the wait for k-step s
      // (8 reads per k-step, LDS returns in order), and the wait is tied to the registers it covers so
      // that no MFMA can be scheduled above it.
      uint32_t va[DBLK];
      {
        const uint32_t vbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds +
                               (kDmaBufs + buf) * kDmaTile + (4 * h + tr_rowq) * kRowB + tr_col;
        const int vsw = D == 128 ? tr_rowq : (tr_rowq >> 1);   // 64-byte unit swizzle of this lane's row
#pragma unroll
        for (int db = 0; db < DBLK; ++db) va[db] = vbase + ((db ^ vsw) << 6);
      }
      u32x2 lo[2][DBLK], hi[2][DBLK];
#define SP_TR_ISSUE(SET, STEP)                                                                       \
  _Pragma("unroll") for (int db = 0; db < DBLK; ++db) {                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[SET][db]) : "v"(va[db]), "n"((STEP) * 16 * kRowB));            \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[SET][db]) : "v"(va[db]), "n"((STEP) * 16 * kRowB + 8 * kRowB)); \
  }
// (the waits name every register of the set they cover; N = reads still allowed in flight = 2 * DBLK of the other set)
#define SP_TR_WAIT(SET, LATER)                                                                       \
  do {                                                                                               \
    if constexpr (DBLK == 4) {                                                                       \
      if constexpr (LATER)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(8)"                                                          \
                     : "+v"(lo[SET][0]), "+v"(lo[SET][1]), "+v"(lo[SET][2]), "+v"(lo[SET][3]),       \
                       "+v"(hi[SET][0]), "+v"(hi[SET][1]), "+v"(hi[SET][2]), "+v"(hi[SET][3]));      \
      else                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                          \
                     : "+v"(lo[SET][0]), "+v"(lo[SET][1]), "+v"(lo[SET][2]), "+v"(lo[SET][3]),       \
                       "+v"(hi[SET][0]), "+v"(hi[SET][1]), "+v"(hi[SET][2]), "+v"(hi[SET][3]));      \
    } else {                                                                                         \
      if constexpr (LATER)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[SET][0]), "+v"(lo[SET][1]), "+v"(hi[SET][0]), "+v"(hi[SET][1])); \
      else                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[SET][0]), "+v"(lo[SET][1]), "+v"(hi[SET][0]), "+v"(hi[SET][1])); \
    }                                                                                                \
  } while (0)
#define SP_TR_MFMA(SET, STEP)                                                                        \
  _Pragma("unroll") for (int db = 0; db < DBLK; ++db) {                                              \
    u32x4 vf;                                                                                        \
    vf[0] = lo[SET][db][0]; vf[1] = lo[SET][db][1]; vf[2] = hi[SET][db][0]; vf[3] = hi[SET][db][1];  \
    oacc[db] = mfma32<CT>(vf, pf[STEP], oacc[db]);                                                   \
  }
      static_assert(!DMA || DBLK == 4 || DBLK == 2, "wait lists are written for four or two d blocks");
      SP_TR_ISSUE(0, 0)
      SP_TR_ISSUE(1, 1)
      SP_TR_WAIT(0, true);
      SP_TR_MFMA(0, 0)
      SP_TR_ISSUE(0, 2)
      SP_TR_WAIT(1, true);
      SP_TR_MFMA(1, 1)
      SP_TR_ISSUE(1, 3)
      SP_TR_WAIT(0, true);
      SP_TR_MFMA(0, 2)
      SP_TR_WAIT(1, false);
      SP_TR_MFMA(1, 3)
#undef SP_TR_ISSUE
#undef SP_TR_WAIT
#undef SP_TR_MFMA
    } else {
    const char* ldsV = lds + buf * C::kTileBytes + BN * SK;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        // V^T fragments of one 16-key k-step (DBLK d-blocks, two transposed reads each)
        u32x4 vf[DBLK];
        const int keyA = kb * 32 + 16 * sidx + 4 * h + tr_rowq;       // rows for elements 0..3
        const char* vp = ldsV + keyA * SV + tr_col;
#pragma unroll
        for (int db = 0; db < DBLK; ++db) {
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(vp + db * 64));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(vp + 8 * SV + db * 64));
          const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
          vf[db][0] = lo2[0];
          vf[db][1] = lo2[1];
          vf[db][2] = hi2[0];
          vf[db][3] = hi2[1];
        }
#pragma unroll
        for (int db = 0; db < DBLK; ++db) oacc[db] = mfma32<CT>(vf[db], pf[kb * 2 + sidx], oacc[db]);
      }
    }
    }
  };

  if constexpr (DMA) {
    const int dR = lane / kCPR, dp = lane % kCPR;     // row within a piece / 16-byte chunk position of this lane
    const int64_t row_off = (int64_t)hk * D * 2;
    struct Slots { int s[kPPW]; };
    Slots s0, s1, s2, s3;   // slot indices of tiles x with (x - tbeg) % 4 == 0, 1, 2, 3
    auto load_slots = [&](Slots& r, int tile) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < kPPW; ++j) {
        const int key = tile * BN + wave * 8 + kRPP * j + dR;
        r.s[j] = idx_row[max(min(key, kv_len - 1), 0)];   // see fetch_slots
      }
    };
    auto dma = [&](const Slots& r, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < kPPW; ++j) {
        const int R = wave * 8 + kRPP * j + dR;
        const int64_t off = (int64_t)r.s[j] * tok_bytes + row_off;
        const int fk = D == 128 ? (R & 15) : ((R >> 1) & 7);
        const int fv = D == 128 ? ((R & 3) << 2) : (((R >> 1) & 1) << 2);
        const char* ks = a.kbuf + off + ((dp ^ fk) << 4);
        const char* vs = a.vbuf + off + ((dp ^ fv) << 4);
        char* kd = lds + buf * kDmaTile + (wave * 8 + kRPP * j) * kRowB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ks,
                                         (__attribute__((address_space(3))) void*)kd, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vs,
                                         (__attribute__((address_space(3))) void*)(kd + kDmaBufs * kDmaTile), 16, 0, 0);
      }
    };
    // The compiler does not track LDS-DMA arrivals: the waits are placed by hand.  A tile step issues
    // kPPW index loads (tile +5) and the 2 kPPW DMA pieces of tile +3 (D = 128: 2 + 4; D = 64: 1 + 2); before the
    // barrier that publishes tile +1, everything up to ITS pieces must have landed, i.e. all but the pieces of tiles
    // +2 and +3.  The count must not rely on the index loads: hipcc hoists them ahead of the prologue's pieces and
    // deletes them from the tail steps (unused values), so the wait is "all but the 4 kPPW youngest operations"
    // (8 / 4), which leaves at most the pieces of tiles +2 and +3 in flight whatever else was issued between them.
    // A raw s_barrier: __syncthreads() would drain every DMA in flight (its fence waits vmcnt(0)).
    load_slots(s0, tbeg);
    load_slots(s1, tbeg + 1);
    load_slots(s2, tbeg + 2);
    dma(s0, 0);
    // every index load of the prologue is issued AHEAD of the last eight pieces, so that the counted wait
    // below covers it: a load that hipcc believes may still be pending at the loop head costs a
    // s_waitcnt vmcnt(0) inside the loop (its merge of the two paths into the loop is not exact) - checked
    // on the assembly by tests/test_extend_isa.py
    load_slots(s3, tbeg + 3);
    load_slots(s0, tbeg + 4);
    __builtin_amdgcn_sched_barrier(0);
    dma(s1, 1);
    dma(s2, 2);
    // (the s_waitcnt builtin, not asm text: hipcc's own wait insertion sees it and learns that the older
    // loads - the Q fragments, the index loads - have landed; an asm wait it cannot see leaves it
    // believing they may still be pending at the loop head and it drains vmcnt(0) in every iteration)
    if constexpr (kPPW == 2) __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): tile tbeg has landed
    else __builtin_amdgcn_s_waitcnt(0x0F74);                          // vmcnt(4)
    asm volatile("s_barrier" ::: "memory");
    SP_WGSTAMP(wg2);
    auto step = [&](int t, int buf, Slots& refill, const Slots& issue) __attribute__((always_inline)) {
      SP_STAMP(ts0);
#ifndef SP_EXTEND_NOGATHER   // diagnostic: the matrix pipeline alone (tiles hold whatever the prologue left)
      load_slots(refill, t + 5);
      dma(issue, (buf + 3) % 4);
#endif
      SP_STAMP(ts1);
      u32x4 pf[4];
#ifdef SP_EXTEND_NOCOMPUTE   // diagnostic: the gather pipeline alone
      const bool visible = false;
#else
      const bool visible = qk_softmax(t, buf, pf);
      if (visible) pv(buf, pf);
#endif
#ifdef SP_EXTEND_STAMPS
      if (!visible) ts2 = ts3 = ts1;
      stamp_sum[6] += visible ? 1 : 0;
      stamp_sum[7] += 1;
#endif
      SP_STAMP(ts4);
      if constexpr (kPPW == 2) __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0)
      else __builtin_amdgcn_s_waitcnt(0x0074);                          // vmcnt(4) lgkmcnt(0)
      SP_STAMP(ts5);
      asm volatile("s_barrier" ::: "memory");
      SP_STAMP(ts6);
      SP_STAMP_ACC(0, ts0, ts1);   // index loads + issue of the DMA pieces three tiles ahead
      SP_STAMP_ACC(1, ts1, ts2);
      SP_STAMP_ACC(2, ts2, ts3);
      SP_STAMP_ACC(3, ts3, ts4);
      SP_STAMP_ACC(4, ts4, ts5);   // wait for the next tile's pieces
      SP_STAMP_ACC(5, ts5, ts6);   // barrier
    };
    int t = tbeg;
    for (; t + 3 < ntiles; t += 4) {
      step(t, 0, s1, s3);
      step(t + 1, 1, s2, s0);
      step(t + 2, 2, s3, s1);
      step(t + 3, 3, s0, s2);
    }
    if (t < ntiles) {
      step(t, 0, s1, s3);
      if (t + 1 < ntiles) {
        step(t + 1, 1, s2, s0);
        if (t + 2 < ntiles) step(t + 2, 2, s3, s1);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no DMA may outlive the workgroup's LDS
    SP_WGSTAMP(wg3);
  } else {
  fetch_slots(setA, tbeg);
  fetch_slots(setB, tbeg + 1);
  prefetch(setA, tbeg);
  prefetch(setB, tbeg + 1);
  stage(setA, tbeg & 1);
  __syncthreads();
  // one tile: `fill` is the free register set (it receives tile t+2), `next` holds tile t+1
  auto tile_step = [&](int t, RegSet& fill, const RegSet& next) __attribute__((always_inline)) {
    SP_STAMP(ts0);
    // unconditional, also past the last tile (rows of the last valid key, never consumed): see fetch_slots
    prefetch(fill, t + 2);
    SP_STAMP(ts1);
    u32x4 pf[4];
    const bool visible = qk_softmax(t, t & 1, pf);
    if (visible) pv(t & 1, pf);
#ifdef SP_EXTEND_STAMPS
    if (!visible) ts2 = ts3 = ts1;
#endif
    SP_STAMP(ts4);
    stage(next, (t + 1) & 1);   // past the last tile: dummy rows into the buffer nobody reads again
    SP_STAMP(ts5);
    __syncthreads();
    SP_STAMP(ts6);
    SP_STAMP_ACC(0, ts0, ts1);   // issue of the next tile's gathers (+ wait for its slot indices)
    SP_STAMP_ACC(1, ts1, ts2);   // K fragment reads + S^T MFMAs (issue)
    SP_STAMP_ACC(2, ts2, ts3);   // softmax (starts by waiting for the S^T results)
    SP_STAMP_ACC(3, ts3, ts4);   // V^T reads + O^T MFMAs (issue)
    SP_STAMP_ACC(4, ts4, ts5);   // wait for the gathers + LDS writes
    SP_STAMP_ACC(5, ts5, ts6);   // barrier
#ifdef SP_EXTEND_STAMPS
    stamp_sum[6] += visible ? 1 : 0;
    stamp_sum[7] += 1;
#endif
  };
  // whole pairs in the loop, an odd last tile after it: with the second step under a condition the
  // compiler sees a path from the first step straight to the loop head, where set A's indices would
  // be the youngest loads, and waits vmcnt(0) at the top of EVERY iteration
  int t = tbeg;
  for (; t + 1 < ntiles; t += 2) {
    tile_step(t, setA, setB);
    tile_step(t + 1, setB, setA);
  }
  if (t < ntiles) tile_step(t, setA, setB);
  }
#ifdef SP_EXTEND_STAMPS
  if (lane == 0 && g_stamp_buf) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_stamp_buf[i], stamp_sum[i]);
  }
#endif

  // ---- epilogue: O[row][head][d] = O^T[d][row] / l.  In the accumulator a lane holds 4 consecutive d of ITS row
  // per register quad: stored straight from there, every store instruction writes 64 pieces of 8 bytes
  // into 32 different rows, and the workgroup's last microseconds are store ISSUE (a workgroup's fixed
  // cost measured 10 - 13 us, a tile step 1.7 us).  The wave's 32 x D tile goes through LDS instead (the
  // tile ring is free by now: barrier first, every wave has left it and drained its DMA) and leaves as
  // whole rows, 16 bytes per lane, 4 rows of 256 B per store instruction.
  const float l_run = l_part + xchg32(l_part);
  __syncthreads();
  {
    constexpr int EPS = C::ROW_B + 8;                   // padded row: the 8-byte writes of 16 lanes hit 32 different banks
    char* ep = lds + wave * (32 * EPS);
    const float inv = l_run > 0.f ? a.out_scale / l_run : 0.f;  // no visible key (empty encoder): zeros
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int quad = 0; quad < 4; ++quad) {
        u32x2 w;
        w[0] = pack2<Tag>(oacc[db][4 * quad] * inv, oacc[db][4 * quad + 1] * inv);
        w[1] = pack2<Tag>(oacc[db][4 * quad + 2] * inv, oacc[db][4 * quad + 3] * inv);
        *(u32x2*)(ep + c * EPS + (db * 32 + 8 * quad + 4 * h) * 2) = w;
      }
    // wave-private region: no barrier, the compiler orders the wave's own LDS writes before its reads
    constexpr int LPR = C::ROW_B / 16;                  // lanes per output row
    constexpr int RPI = 64 / LPR;                       // rows per store instruction
    const int rr0 = lane / LPR, ch = lane % LPR;
    char* ob = (char*)a.out + ((t0 + r0) * a.o_stride + (int64_t)head * D) * 2 + ch * 16;
    if (wave_live) {
#pragma unroll
      for (int k = 0; k < 32 / RPI; ++k) {
        const int rr = k * RPI + rr0;
        // 8-byte LDS reads (the padded stride is not a multiple of 16)
        u32x4 v;
        const u32x2 lo = *(const u32x2*)(ep + rr * EPS + ch * 16), hi = *(const u32x2*)(ep + rr * EPS + ch * 16 + 8);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        if (r0 + rr < E) st16(ob + (int64_t)rr * a.o_stride * 2, v);
      }
    }
  }
#ifdef SP_EXTEND_WGSTAMPS
  SP_WGSTAMP(wg4);
  if (threadIdx.x == 0 && g_stamp_buf && wg2) {
    atomicAdd(&g_stamp_buf[0], wg1 - wg0);   // entry -> metadata decoded
    atomicAdd(&g_stamp_buf[1], wg2 - wg1);   // -> first tile landed and published (Q loads, index loads, first pieces)
    atomicAdd(&g_stamp_buf[2], wg3 - wg2);   // tile loop
    atomicAdd(&g_stamp_buf[3], wg4 - wg3);   // barrier + output through LDS + stores issued
    atomicAdd(&g_stamp_buf[7], 1ULL);
  }
#endif
}

// test hook (set through sp_debug_set, never read from the environment on the call path): how far the
// running maximum may trail (log2 units); < 0 restores the shipped value.  tests compare 0 (rescale at
// every growth of a row maximum) with the shipped threshold.
static float g_extend_defer = kDeferLog2;
void set_extend_defer_x10(int tenths) { g_extend_defer = tenths < 0 ? kDeferLog2 : 0.1f * tenths; }
#if defined(SP_EXTEND_STAMPS) || defined(SP_EXTEND_WGSTAMPS)
void set_extend_stamp_buffer(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &q, sizeof(q));
}
#endif

static int g_extend_dma = 1;
void set_extend_dma(int v) { g_extend_dma = v; }
int g_extend_last_kernel = 0;

constexpr int kExtendWaves = 8;

// row blocks per workgroup tile for a query-head group of width G (any G: Gk = the largest of 4, 2, 1
// that divides it; the remaining G / Gk head blocks become extra workgroups)
static int extend_gk(int G) { return G % 4 == 0 ? 4 : (G % 2 == 0 ? 2 : 1); }
int extend_block_rows(int num_q_heads, int num_kv_heads) {
  return 32 * (kExtendWaves / extend_gk(num_q_heads / num_kv_heads));
}

template <typename Tag, int D, int GK>
static int launch_extend(const ExtendArgs& a, int max_extend_len, int halves, hipStream_t st) {
  constexpr int NW = kExtendWaves;
  typedef ExtCfg<D, NW * 64> C;
  constexpr int BM = 32 * (NW / GK);
  constexpr int kLds = 2 * C::kTileBytes;
  // with a plan: one grid row per (request, row block) item, heaviest first; without: every
  // request x every possible row block (workgroups past a request's end exit at once)
  const dim3 grid(a.Hkv * halves, a.plan ? a.plan_items : (max_extend_len + BM - 1) / BM, a.plan ? 1 : a.bs);
  const bool plain = !(a.logit_cap > 0.f) && a.window < 0;
#define SP_EXT_LAUNCH(KV8_, PLAIN_, DMA_)                                                            \
  do {                                                                                               \
    constexpr int lds_bytes = DMA_ ? 8 * 64 * 2 * D : kLds;                                          \
    static bool attr_set = false; /* benign race: idempotent */                                      \
    if (!attr_set && lds_bytes > 64 * 1024) {                                                        \
      (void)hipFuncSetAttribute((const void*)extend_mfma_kernel<Tag, D, GK, NW, KV8_, PLAIN_, DMA_>, \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);              \
      attr_set = true;                                                                               \
    }                                                                                                \
    extend_mfma_kernel<Tag, D, GK, NW, KV8_, PLAIN_, DMA_><<<grid, NW * 64, lds_bytes, st>>>(a);     \
  } while (0)
  if (a.kv8) {
    if (plain) SP_EXT_LAUNCH(true, true, false); else SP_EXT_LAUNCH(true, false, false);
  } else if (g_extend_dma) {       // 16-bit pools, D 128 or 64: K/V tiles by LDS-DMA into the swizzled ring
    if (plain) SP_EXT_LAUNCH(false, true, true); else SP_EXT_LAUNCH(false, false, true);
  } else {
    if (plain) SP_EXT_LAUNCH(false, true, false); else SP_EXT_LAUNCH(false, false, false);
  }
#undef SP_EXT_LAUNCH
  SP_LAUNCH_CHECK();
  return SP_OK;
}

template <typename Tag, int D>
static int dispatch_extend_group(const ExtendArgs& a, int G, int max_extend_len, hipStream_t st) {
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
                    float out_scale, int causal, int window_left, int max_extend_len, int64_t max_seq_len,
                    const int32_t* plan, int plan_items, int num_tokens, int dtype, int kv8, hipStream_t st) {
  if (dtype != SP_F16 && dtype != SP_BF16) return SP_ERR_UNSUPPORTED;
  if (head_dim != 64 && head_dim != 128) return SP_ERR_UNSUPPORTED;
  if (batch_size > 65535 || num_q_heads > 65535) return SP_ERR_UNSUPPORTED;   // grid.z, grid.y
  if (max_extend_len > 65535 * 32) return SP_ERR_UNSUPPORTED;
  if (q_stride % 8 || out_stride % 8 || kv_buffer_stride % 8 || ((uintptr_t)out & 15)) return SP_ERR_UNSUPPORTED;
  if (kv8 && (((uintptr_t)k_buffer | (uintptr_t)v_buffer) & 7)) return SP_ERR_UNSUPPORTED;
  ExtendArgs a;
  a.out = out; a.q = q; a.kbuf = (const char*)k_buffer; a.vbuf = (const char*)v_buffer;
  a.r2t = req_to_token; a.r2t_stride = req_to_token_stride; a.req_idx = req_pool_indices;
  a.seq_lens = seq_lens; a.kv_start = kv_start; a.idx64 = idx64; a.ext_lens = extend_seq_lens;
  a.ext_start = extend_start_loc; a.bs = batch_size; a.Hq = num_q_heads; a.Hkv = num_kv_heads;
  a.q_stride = q_stride; a.o_stride = out_stride; a.kv_stride = kv_buffer_stride;
  a.sm_scale = sm_scale; a.logit_cap = logit_cap; a.out_scale = out_scale; a.causal = causal;
  a.window = causal ? window_left : -1;
  a.kv8 = kv8;
  a.defer = g_extend_defer;
  a.plan = plan; a.plan_items = plan_items; a.num_tokens = num_tokens;
  if (plan && (plan_items <= 0 || plan_items > 65535)) a.plan = nullptr;   // grid.y limit: unplanned launch
  const int G = num_q_heads / num_kv_heads;
  if (max_extend_len <= 0) return SP_OK;
  if (const int form = try_extend_w64(a, head_dim, dtype, max_extend_len, max_seq_len, st)) {
    g_extend_last_kernel = form == 2 ? 3 : 2;
    SP_LAUNCH_CHECK();
    return SP_OK;
  }
  g_extend_last_kernel = 1;
#ifdef SP_EXTEND_ONLY_HEADLINE   // ISA experiments only: one instantiation family, seconds to compile
  if (dtype == SP_BF16 && head_dim == 128 && G % 4 == 0)
    return launch_extend<bf16_tag, 128, 4>(a, max_extend_len, G / 4, st);
  return SP_ERR_UNSUPPORTED;
#else
  if (dtype == SP_BF16) {
    return head_dim == 128 ? dispatch_extend_group<bf16_tag, 128>(a, G, max_extend_len, st)
                           : dispatch_extend_group<bf16_tag, 64>(a, G, max_extend_len, st);
  }
  return head_dim == 128 ? dispatch_extend_group<f16_tag, 128>(a, G, max_extend_len, st)
                         : dispatch_extend_group<f16_tag, 64>(a, G, max_extend_len, st);
#endif
}

}  // namespace sp
