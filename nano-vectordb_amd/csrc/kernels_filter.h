// kernels_filter.h -- the hot kernels: MFMA filter scan of the fp16 / int8 corpus (gfx950, CDNA4).
//
// Dataflow ("The filter kernel -- CDNA4 mapping"):
//   * a workgroup = 4 wavefronts = one per SIMD, 512-register budget each.  Every wave keeps the B-operand
//     fragments of its queries for the WHOLE K = DIM in registers for the lifetime of the kernel (384 registers:
//     64 queries at DIM <= 768, 32 at DIM <= 1536) -- queries are the stationary operand, they never touch LDS and
//     are read from memory once per launch.
//   * the corpus is the streamed operand: tiles of 32 (or 16) rows x DIM, 48 KB, go HBM -> LDS with direct-to-LDS
//     loads (global_load_lds_dwordx4, 1 KB per wave-instruction), three stages deep, one s_barrier per tile,
//     counted vmcnt so two tiles stay in flight; the workgroups that stream the same rows keep together through a
//     sibling rendezvous so that the XCD's L2 serves all but one of them.
//   * per tile and wave: ds_read_b128 A fragments (XOR-swizzled image, conflict-free) feed the MFMAs from inline
//     asm (B operands read in place from AGPRs); the accumulator has the query on the lane, so a threshold is one
//     VGPR and "does anything reach it" is a v_max3 tree + one compare.
//   * survivors (rare) are logged per wave with plain stores and filed under their queries when the wave has
//     finished its stream (scatter_own_log).  Nothing else is written: the B x N score matrix never exists.
//
// Kernels: filter_f16_m16_kernel (production fp16, v_mfma_f32_16x16x32_f16; MB/NQB pick the tile shape),
// filter_f16_k2_kernel (1536 < DIM <= 3072: a tile streamed as two half-K stages),
// filter_f16_kernel (32x32x16 build: batches <= 128, the bootstrap build VAR 7, timing ablations),
// filter_i8w_kernel (int8 two-stage: hi plane always, lo plane on demand), filter_i8_kernel (int8 two-plane:
// bootstrap build, A/B reference), prep_q16 / prep_q8 (query scaling, error bounds), shadow / generator / norm kernels.
//
// The MFMA result is only a FILTER: |filter - reference score| <= ebound[q] (prep kernels), every
// survivor is re-scored in the reference's exact fp32 order afterwards (kernels_exact.h).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_exact.h"

namespace nvdbhip {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx16 __attribute__((ext_vector_type(16)));
typedef int int4_t __attribute__((ext_vector_type(4)));

#define NVDB_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NVDB_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int FILTER_ROWS = 32;          // corpus rows per tile (MFMA M)
constexpr int FILTER_STAGES = 3;
constexpr int FILTER_STAGES_I8 = 5;      // int8 stages are half the bytes: 5 stages keep 4 tiles (96 KB at d=768) in flight per workgroup

// ------------------------------------------------------------------------------------------------
// query preparation for the fp16 filter.
//   q16[q][i]  = half( q[i] * 2^e ),  e chosen so that max|q| * 2^e is in [2^14, 2^15)
//                (keeps small elements out of the half subnormal range; exact power-of-two scale)
//   qscale[q]  = 2^e,   qinv[q] = 2^-e
//   ebound[q]  = REL * ||q|| * max_row_norm  >= |filter - reference score|
//   slack[q]   = 2 * ebound[q]
// grid = nq_pad (multiple of 256), block = 256.  Pad queries get zeros.
// ------------------------------------------------------------------------------------------------
// What init_search_kernel resets per search, folded into the prep launch of the filter path (one launch fewer per search; the
// exact and any-k paths have no prep kernel and keep init_search_kernel).  cnt == nullptr: nothing.  q_src != nullptr: the
// queries are read from THERE (pinned host memory the device reads over PCIe: the host API's small calls enqueue no H2D copy)
// and written to the search's device copy `q32`, which every later kernel reads.
struct PrepInit {
  uint32_t* cnt;
  float* thr;
  uint32_t* misc;
  uint32_t* prog;
  uint32_t prog_words;
  uint32_t* tickets;               // FUSE_TICKETS words: "last workgroup" ticket of the fused rescore + final select launch
  const float* q_src;
  float* q32_copy;
};
constexpr uint32_t FUSE_TICKETS = 1;

// block q of the prep grid (nq_pad blocks): its own query's words; the shared words strided over the grid.  Returns where the block
// reads its query from.
__device__ __forceinline__ const float* prep_init(const PrepInit& pi, const float* q32, uint32_t q, uint32_t tid, uint32_t nq, uint32_t dim, uint32_t* overflow) {
  if (pi.cnt != nullptr) {
    if (tid == 0) { pi.cnt[q] = 0; overflow[q] = 0; pi.thr[q] = NEG_INF; }
    if (q == 0 && tid < 8) pi.misc[tid] = 0;
    if (q == 0 && tid >= 64 && tid < 64 + FUSE_TICKETS) pi.tickets[tid - 64] = 0;
    for (uint32_t j = q * 256 + tid; j < pi.prog_words; j += gridDim.x * 256) pi.prog[j] = 0xFFFFFFFFu;
  }
  if (pi.q_src != nullptr && q < nq) {
    // each thread later re-reads exactly the elements it copies here (same stride): its own global writes, always visible to it
    float* dst = pi.q32_copy + static_cast<uint64_t>(q) * dim;
    const float* srch = pi.q_src + static_cast<uint64_t>(q) * dim;
    for (uint32_t i = tid; i < dim; i += 256) dst[i] = srch[i];
    return dst;
  }
  return q32 + static_cast<uint64_t>(q) * dim;
}

static __global__ __launch_bounds__(256) void prep_q16_kernel(const float* __restrict__ q32, uint32_t nq, uint32_t dim, uint32_t sdim,
                                                       float max_row_norm, float rel, float abs_per_norm, uint32_t* __restrict__ overflow, _Float16* __restrict__ q16,
                                                       float* __restrict__ qscale, float* __restrict__ qinv,
                                                       float* __restrict__ ebound, float* __restrict__ slack, PrepInit pi) {
  __shared__ float red_max[4], red_ss[4];
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const float* src = prep_init(pi, q32, q, tid, nq, dim, overflow);
  // sdim >= dim: row stride of q16 = the (zero-padded) dim the filter kernel is instantiated for
  if (q >= nq) {
    for (uint32_t i = tid; i < sdim; i += 256) q16[static_cast<uint64_t>(q) * sdim + i] = static_cast<_Float16>(0.f);
    if (tid == 0) { qscale[q] = 1.f; qinv[q] = 1.f; ebound[q] = 0.f; slack[q] = 0.f; }
    return;
  }
  float mx = 0.f, ss = 0.f;
  for (uint32_t i = tid; i < dim; i += 256) { const float v = src[i]; mx = fmaxf(mx, fabsf(v)); ss = __builtin_fmaf(v, v, ss); }
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); ss += __shfl_xor(ss, o); }
  if ((tid & 63) == 0) { red_max[tid >> 6] = mx; red_ss[tid >> 6] = ss; }
  __syncthreads();
  mx = fmaxf(fmaxf(red_max[0], red_max[1]), fmaxf(red_max[2], red_max[3]));
  ss = (red_ss[0] + red_ss[1]) + (red_ss[2] + red_ss[3]);
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) {
    int ex; (void)frexpf(mx, &ex);          // mx = m * 2^ex, m in [0.5,1)  -> mx in [2^(ex-1), 2^ex)
    e = 15 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  const float sc = ldexpf(1.f, e);
  for (uint32_t i = tid; i < sdim; i += 256) q16[static_cast<uint64_t>(q) * sdim + i] = static_cast<_Float16>(i < dim ? src[i] * sc : 0.f);
  if (tid == 0) {
    const float nrm = sqrtf(ss) * 1.0001f;
    // abs_per_norm: absolute rounding of the corpus side (fp16 shadow of an fp32 corpus: subnormal halves)
    const float eb = rel * nrm * max_row_norm + abs_per_norm * nrm + 1e-30f;
    qscale[q] = sc; qinv[q] = ldexpf(1.f, -e); ebound[q] = eb; slack[q] = 2.f * eb;
    // a query with a NaN / infinite element (or a norm beyond fp32) has no usable bound: flag it like a list overflow,
    // the host redoes its sub-batch on the exact path, whose comparisons treat such scores as the CPU path does
    if (!(ss < 3.0e38f)) { overflow[q] = 1u; ebound[q] = -1.f; }     // negative bound = "do not self-check this query"
  }
}

// ------------------------------------------------------------------------------------------------
// fp16 MFMA filter scan of rows [row_lo,row_hi), (row_hi-row_lo) % 32 == 0 (the host gives the
// ragged tail to the exact scan).
//   grid   = S * QT workgroups (S row streams x QT query tiles of 256), block = 256
//   LDS    = 3 stages x 32 rows x DIM x 2 bytes (dynamic)
//   DIM % 128 == 0 (bank-swizzle arithmetic below), rows 16-byte aligned.
// Workgroup b: XCD label b % 8; the QT workgroups that stream the same rows against different
// query tiles are consecutive on the same XCD label so that they share that XCD's L2.
//
// Register plan (one wave per SIMD, 512 registers): the 64 queries' B fragments take DIM/2
// registers; the first 64 fragments live in AGPRs, the rest in VGPRs, and the MFMAs are issued
// from inline asm so that they read the AGPR-resident fragments in place (the compiler's builtin
// only takes VGPR sources and would copy 4 registers per MFMA).  What the compiler cannot see
// inside the asm is handled here: the first MFMA of a tile takes the literal 0 as C (no VALU
// write feeding an MFMA), accumulator chains reuse exactly the same 16 registers, and 32 wait
// states separate the last MFMA from the first VALU read of the accumulators.
// ------------------------------------------------------------------------------------------------
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

#define NVDB_MFMA_F16_ZERO_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b))
#define NVDB_MFMA_F16_ZERO_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b))
#define NVDB_MFMA_F16_ACC_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define NVDB_MFMA_F16_ACC_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// Direct-to-LDS 16-byte load issued from inline asm: LDS address = M0 (wave-uniform byte offset)
// + lane*16, global address = sbase + voff.  Issued from asm on purpose: hipcc treats a
// __builtin_amdgcn_global_load_lds as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front
// of the next ds_read, which would drain the two tiles this kernel keeps in flight.  The kernel
// counts these loads itself (s_waitcnt vmcnt(PPW) + s_barrier before a stage is read).
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_off) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_off), "s"(sbase) : "memory");
}

// max of three without the canonicalising v_max the compiler wraps around fmaxf of values it did not produce itself
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// One survivor of the filter, logged by the wave that found it (16 bytes, one dwordx4 store).
struct Hit { float score; uint32_t row; uint32_t qid; uint32_t pad; };
// int8: bound on the L1 norm of a query's hi plane (prep_q8_kernel keeps it): |<row, hi>| <= 128 * I8_HI_L1_MAX < 2^23, the range in
// which an int32 accumulator started at I8_ACC_BIAS (the bits of 2^23 as a float) reads, as a float, 2^23 + H for H >= 0
constexpr uint32_t I8_HI_L1_MAX = 65500;
constexpr int I8_ACC_BIAS = 0x4B000000;
// H * 2^bits for a possibly negative H (a left shift of a negative int is undefined before C++20; the range is checked: |H| < 2^23, bits <= 7)
__device__ __forceinline__ int shl_i32(int v, uint32_t bits) { return static_cast<int>(static_cast<uint32_t>(v) << bits); }
constexpr uint32_t FILTER_LOGCAP = 4096;           // entries per wave and launch (64 KB of log per wave; typical use: < 100)

// where the logged survivors go when the wave has finished its stream
struct ScatterArgs {
  Cand* cand;                // [nq][cap] candidate lists
  uint32_t* cnt;             // [nq] list lengths (atomic slot allocation)
  uint32_t* overflow;        // [nq] per-query list overflow flags
  uint32_t* log_overflow;    // one word: some wave's log overflowed (unknown queries lost entries)
  uint32_t cap;
  uint32_t n_rows;           // rows >= n_rows are the zero rows that pad the corpus to whole tiles
  // Tile order.  The chunk scheme assumes that a chunk is a fair sample of the corpus (then k * chunk / rows_seen rows
  // clear the previous threshold, whatever the data); a corpus stored in topic / cluster order breaks that and floods
  // the lists.  So logical tile g (what row_lo, row_hi and the streams count in) is physical tile perm_tile(g): an odd
  // multiplier modulo the next power of two, cycle-walked back into [0, perm_T) -- a bijection of the corpus' tiles
  // that costs a handful of scalar instructions per tile.  perm_mask == 0: identity.
  uint32_t perm_mul, perm_mask, perm_T;
  // XCD balance (null: equal shares).  The chip's 8 XCDs do not run at the same speed under this load (their tile loops differ by
  // +-3.5 %, workgroups inside one XCD by 0.3 %: profiles/r02_xcd_balance_clock.txt) and a launch ends with its slowest workgroup.
  // xcdw[0..7]: relative speed of XCD label x (mean 1) -- the row streams of label x get a share of the tiles proportional to it;
  // xcdw[8..15], [16..23]: tile-loop microseconds and tiles per label, added up by the big launches, turned into new weights by
  // the select kernel that follows every filter launch (device-side only, nothing for the host to read).
  float* xcdw;
  // int8: a query is quantised to t = (hi << lo_bits) + lo with hi in [-127, 127], lo in [-2^(lo_bits-1), 2^(lo_bits-1));
  // the filter value of a row is (H << lo_bits) + L, H / L its integer dot products with the two planes (prep_q8_kernel)
  uint32_t lo_bits;
};

// tiles [lo, hi) of row stream `stream` out of S streams over T tiles.  Streams s with equal s & 7 share an XCD label when the
// XCD-aware block mapping is active (xcd_map); f(s) is the same monotone expression for every workgroup, f(0) = 0, f(S) = T.
__device__ __forceinline__ void stream_tile_range(uint32_t T, uint32_t S, uint32_t stream, bool xcd_map, const float* xcdw, uint32_t& lo, uint32_t& hi) {
  if (!xcd_map || xcdw == nullptr || (S & 7u) != 0) {
    lo = static_cast<uint32_t>(static_cast<uint64_t>(T) * stream / S);
    hi = static_cast<uint32_t>(static_cast<uint64_t>(T) * (stream + 1) / S);
    return;
  }
  double pre[9];
  pre[0] = 0.0;
#pragma unroll
  for (int x = 0; x < 8; ++x) pre[x + 1] = pre[x] + static_cast<double>(xcdw[x]);
  const double total = pre[8] * static_cast<double>(S >> 3);
  auto f = [&](uint32_t s) -> uint32_t {
    if (s >= S) return T;
    double part = 0.0;
#pragma unroll
    for (int x = 0; x < 8; ++x) part = (static_cast<uint32_t>(x) == (s & 7u)) ? pre[x] : part;
    const double cum = pre[8] * static_cast<double>(s >> 3) + part;
    const uint32_t v = static_cast<uint32_t>(static_cast<double>(T) * (cum / total));
    return v > T ? T : v;
  };
  lo = f(stream);
  hi = f(stream + 1);
}

// end of a big launch: this workgroup's tile-loop time and tile count, filed under its XCD label
__device__ __forceinline__ void record_xcd_speed(float* xcdw, bool xcd_map, uint32_t label, uint32_t ntiles, uint32_t ticks_100mhz) {
  if (xcdw == nullptr || !xcd_map || ntiles < 128) return;
  atomicAdd(xcdw + 8 + label, static_cast<float>(ticks_100mhz) * 0.01f);
  atomicAdd(xcdw + 16 + label, static_cast<float>(ntiles));
}

__host__ __device__ __forceinline__ uint32_t perm_tile_raw(uint32_t g, uint32_t mul, uint32_t mask, uint32_t T) {
  if (mask == 0) return g;
  uint32_t x = g;
  do { x = (x * mul + 0x9E3779B1u) & mask; } while (x >= T);
  return x;
}
__device__ __forceinline__ uint32_t perm_tile(uint32_t g, const ScatterArgs& a) { return perm_tile_raw(g, a.perm_mul, a.perm_mask, a.perm_T); }

// the multiplier / mask for T tiles (T < 64: identity)
__host__ __device__ __forceinline__ void perm_params(uint32_t T, uint32_t& mul, uint32_t& mask) {
  mul = 1u; mask = 0u;
  if (T < 64) return;
  uint32_t m = 1;
  while (m < T) m <<= 1;
  mask = m - 1;
  mul = (static_cast<uint32_t>(0.6180339887 * m) | 1u) & mask;    // odd: a bijection modulo the power of two
}

// File this wave's logged survivors under their queries (cand[qid][slot], slot from an atomic counter).  Runs
// once, after the tile loop: nothing is in flight any more, so the returning atomics cost nothing in the loop.
__device__ __forceinline__ void scatter_own_log(const Hit* mylog, uint32_t wcnt, const ScatterArgs& a, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own log stores (and the speculative stages) have completed
  uint32_t n = wcnt;
  if (n > FILTER_LOGCAP) { if (lane == 0) *a.log_overflow = 1u; n = FILTER_LOGCAP; }
  for (uint32_t i = lane; i < n; i += 64) {
    // L2-served loads: the entries were written by this wave in this launch
    const uint32_t* w = reinterpret_cast<const uint32_t*>(mylog + i);
    const uint32_t sbits = __hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t row = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t qid = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (row >= a.n_rows) continue;
    const uint32_t slot = atomicAdd(&a.cnt[qid], 1u);
    if (slot < a.cap) a.cand[static_cast<uint64_t>(qid) * a.cap + slot] = Cand{__builtin_bit_cast(float, sbits), row};
    else a.overflow[qid] = 1u;
  }
}

// VAR selects timing-only ablation builds (results are wrong for VAR != 0; used by
// nvdb_hip_debug_filter_variant): 1 = no direct-to-LDS loads in the loop, 2 = 1 + no barrier,
// 3 = no MFMA (loads + LDS reads only), 4 = no epilogue compare, 5 = no LDS reads (MFMA on a constant).
// NB = 32-query blocks per wave: 2 -> 256 queries per workgroup (MFMA-bound batches), 1 -> 128 queries
// per workgroup (half the MFMA work per streamed byte: the HBM-bound regime, nq <= 128).
// RING = A fragments in flight LDS -> VGPR.
// same, with a full 64-bit per-lane source address (row gathers whose offsets exceed 32 bits)
__device__ __forceinline__ void glds16_v(const void* gptr, uint32_t lds_off) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}

template <int DIM, int NB, int VAR = 0, int RING = 6>
__global__ __launch_bounds__(256, 1) void filter_f16_kernel(
    const _Float16* __restrict__ rows, uint32_t row_lo, uint32_t row_hi, const _Float16* __restrict__ q16,
    uint32_t nq, uint32_t QT, const float* __restrict__ thr, const float* __restrict__ qscale,
    const float* __restrict__ qinv, Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t aux) {
  // aux: VAR 7 (bootstrap build) only -- hitlog then points at the candidate lists and aux is their stride
  constexpr int KSTEPS = DIM / 16;                 // MFMA k-steps per tile
  constexpr int ROW_BYTES = DIM * 2;
  constexpr int STAGE_BYTES = FILTER_ROWS * ROW_BYTES;
  constexpr int PIECES = STAGE_BYTES / 1024;       // 1 KB direct-to-LDS pieces per stage
  constexpr int PPW = PIECES / 4;                  // pieces per wave
  constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;
  constexpr int NFRAG = NB * KSTEPS;               // B fragments per wave
  constexpr int NFRAG_A = NFRAG < 64 ? NFRAG : 64; // ... of which this many live in AGPRs
  constexpr int NFRAG_V = NFRAG - NFRAG_A;
  constexpr int QPW = 32 * NB, QPB = 4 * QPW;      // queries per wave / per workgroup
  static_assert(NB == 1 || NB == 2, "one or two 32-query blocks per wave");
  static_assert(DIM % 128 == 0, "swizzle assumes row stride is a multiple of 256 bytes");
  static_assert(PIECES % 4 == 0, "pieces must split evenly over 4 waves");
  static_assert(KSTEPS <= 64, "query block 0 must fit the AGPR-resident fragments");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, hsel = lane >> 5;
  const uint32_t wave_gid = blockIdx.x * 4 + wave;

  // ---- workgroup -> (row stream, query tile) ---------------------------------------------------
  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) {
    const uint32_t xcd = b & 7u, i = b >> 3;       // i-th workgroup of this XCD label
    qt = i % QT;
    stream = (i / QT) * 8u + xcd;
  } else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / FILTER_ROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // ---- stationary operand: this wave's queries, all of K, in registers -------------------------
  // fragment f = nb*KSTEPS + s : query block nb (32 queries), k-step s.  Lane (r31,hsel) holds
  // q16[query r31 of the block][16 s + 8 hsel .. +8] -- the same k-slice the A fragment holds, so
  // the MFMA's internal k order is irrelevant.
  const uint32_t qbase = qt * QPB + wave * QPW;
  float4_t bqa[NFRAG_A];
  float4_t bqv[NFRAG_V > 0 ? NFRAG_V : 1];
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) {
    const int nb = f / KSTEPS, s = f % KSTEPS;
    const float4_t v = *reinterpret_cast<const float4_t*>(q16 + static_cast<uint64_t>(qbase + nb * 32 + r31) * DIM + 16 * s + 8 * hsel);
    if (f < NFRAG_A) bqa[f] = v; else bqv[f - NFRAG_A] = v;
  }
  // Make the compiler retire these loads HERE (an empty asm that reads every fragment): left to
  // itself it defers its `s_waitcnt vmcnt(n)` for them to the first use inside the tile loop,
  // where they would also drain the direct-to-LDS loads this kernel keeps in flight.
#pragma unroll
  for (int f = 0; f < NFRAG_A; ++f) asm volatile("" ::"a"(bqa[f]));
#pragma unroll
  for (int f = 0; f < NFRAG_V; ++f) asm volatile("" ::"v"(bqv[f]));
  float thr_s[NB], inv_s[NB];
  uint32_t qid[NB];
  bool wave_has_queries = false;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    qid[nb] = qbase + nb * 32 + r31;
    const bool real = qid[nb] < nq;
    thr_s[nb] = real ? thr[qid[nb]] * qscale[qid[nb]] : __builtin_huge_valf();
    inv_s[nb] = real ? qinv[qid[nb]] : 0.f;
    asm volatile("" ::"v"(thr_s[nb]), "v"(inv_s[nb]));
  }
  wave_has_queries = qbase < nq;                   // wave-uniform: padding-only waves skip their MFMAs

  // ---- per-lane source offsets of this wave's direct-to-LDS pieces -----------------------------
  // LDS image of a stage: [32 rows][CHUNKS_PER_ROW 16-byte chunks], chunk c of row r stored at
  // chunk position c ^ (r & 15) (XOR stays inside a 16-chunk = 256-byte bank row).  The LDS side of
  // a direct-to-LDS load is linear (M0 base + lane*16), so the permutation is applied to the
  // per-lane GLOBAL address: LDS position P = piece*64 + lane  <-  row P / CPR, chunk (P % CPR) ^ (row & 15).
  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CHUNKS_PER_ROW, cpos = P % CHUNKS_PER_ROW;
    src_off[i] = r * ROW_BYTES + ((cpos ^ (r & 15u)) << 4);
  }
  // A-fragment read offset: lane (r31,hsel) reads chunk 2s+hsel of row r31, stored at position
  // (2s+hsel) ^ (r31&15); since 2s+hsel == 2s ^ hsel this is a_base ^ ((s&7) << 5) + 256*(s>>3).
  const uint32_t a_base = r31 * ROW_BYTES + ((static_cast<uint32_t>(hsel) ^ (r31 & 15u)) << 4);

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  // tile t_rel of this stream (clamped: past-the-end tiles re-load the last one, harmlessly)
  const uint32_t g_lo = row_lo / FILTER_ROWS + t_lo;          // logical index of my first tile in the whole corpus
  auto tile_phys = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa); };
  auto tile_ptr = [&](uint32_t t_rel) -> const char* {
    if constexpr (VAR == 6) return gbase + static_cast<uint64_t>(row_lo + (t_lo + (t_rel & 7u)) * FILTER_ROWS) * ROW_BYTES;   // ablation: 8 tiles per stream, L2-resident
    return gbase + static_cast<uint64_t>(tile_phys(t_rel)) * FILTER_ROWS * ROW_BYTES;
  };
  auto issue_piece = [&](const char* tile, uint32_t buf, int i) {
    glds16(src_off[i], tile, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };

#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_piece(tile_ptr(0), 0, i);
#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_piece(tile_ptr(1), 1, i);

  constexpr int PIECE_EVERY = KSTEPS / PPW;        // one direct-to-LDS piece per this many k-steps
  static_assert(KSTEPS % PPW == 0 && KSTEPS >= RING, "schedule assumes KSTEPS is a multiple of PPW");
  uint32_t wcnt = 0;                               // survivors logged by this wave (uniform)
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;

  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  for (uint32_t t = 0; t < NT; ++t) {
    // my pieces of tile t have landed once all but the newest stage's PPW loads are complete
    if constexpr (VAR != 1 && VAR != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    if constexpr (VAR != 2) __builtin_amdgcn_s_barrier();   // everyone's pieces landed; buffer (t+2)%3 is free

    const char* next_tile = tile_ptr(t + 2);
    const uint32_t next_buf = (t + 2) % FILTER_STAGES;
    const char* stage = smem + (t % FILTER_STAGES) * STAGE_BYTES;
    if (!wave_has_queries) {                       // padding-only wave: keep streaming, skip the math
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(next_tile, next_buf, i);
      continue;
    }
    auto read_a = [&](int s) -> float4_t {
      return *reinterpret_cast<const float4_t*>(stage + (a_base ^ ((s & 7) << 5)) + (s >> 3) * 256);
    };
    float4_t ar[RING];
    if constexpr (VAR == 5) {
#pragma unroll
      for (int s = 0; s < RING; ++s) ar[s] = float4_t{1.f, 2.f, 3.f, 4.f};
    }
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) if constexpr (VAR != 5) ar[s] = read_a(s);
    floatx16 acc[NB];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      // keep RING-1 fragment reads ahead of the MFMAs; the slot written here was consumed by step s-1
      if constexpr (VAR != 5) { if (s + RING - 1 < KSTEPS) ar[(s + RING - 1) % RING] = read_a(s + RING - 1); }
      const float4_t a = ar[s % RING];
      if constexpr (VAR == 3) {
        asm volatile("" ::"v"(a));
        if (s == 0) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = floatx16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int f = nb * KSTEPS + s;
          if (s == 0) {
            if (f < NFRAG_A) NVDB_MFMA_F16_ZERO_A(acc[nb], a, bqa[f < NFRAG_A ? f : 0]);
            else NVDB_MFMA_F16_ZERO_V(acc[nb], a, bqv[f >= NFRAG_A ? f - NFRAG_A : 0]);
          } else {
            if (f < NFRAG_A) NVDB_MFMA_F16_ACC_A(acc[nb], a, bqa[f < NFRAG_A ? f : 0]);
            else NVDB_MFMA_F16_ACC_V(acc[nb], a, bqv[f >= NFRAG_A ? f - NFRAG_A : 0]);
          }
        }
      }
      // stream tile t+2 in behind the MFMAs, one 1 KB piece every PIECE_EVERY k-steps
      if constexpr (VAR != 1 && VAR != 2) { if (s % PIECE_EVERY == PIECE_EVERY - 1) issue_piece(next_tile, next_buf, s / PIECE_EVERY); }
    }
    // 32 wait states: MFMA result -> VALU read (the compiler does not see inside the asm)
    if constexpr (NB == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
    else asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]));

    if constexpr (VAR == 7) {
      // ---- bootstrap epilogue: best row of this tile for every query -> cand[qid][tile] -----------
      // (a lower bound generator for the first thresholds: the k-th largest of T tile maxima is <= the
      //  k-th largest score overall; the host discards these entries after the threshold is taken and
      //  the same rows are scanned again by the normal build, so nothing is lost or duplicated)
      const uint32_t tile = t_lo + t;                          // list slot: logical
      const uint32_t row0b = tile_phys(t) * FILTER_ROWS;       // rows: physical
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        // rows past the end of the corpus (zero padding of the last tile, which the permuted order can bring here) are
        // not rows: they must not stand in for one of the T rows the threshold argument counts
        float best = -__builtin_huge_valf();
        uint32_t brow = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool real = row0b + (r & 3) + 8 * (r >> 2) + 4 * hsel < sa.n_rows;
          const bool gt = real && acc[nb][r] > best;
          best = gt ? acc[nb][r] : best; brow = gt ? static_cast<uint32_t>(r) : brow;
        }
        uint32_t grow = row0b + (brow & 3) + 8 * (brow >> 2) + 4 * hsel;
        const float obest = __shfl_xor(best, 32);
        const uint32_t orow = static_cast<uint32_t>(__shfl_xor(static_cast<int>(grow), 32));
        if (obest > best || (obest == best && orow < grow)) { best = obest; grow = orow; }
        if (hsel == 0 && qid[nb] < nq) reinterpret_cast<Cand*>(hitlog)[static_cast<uint64_t>(qid[nb]) * aux + tile] = Cand{best * inv_s[nb], grow};
      }
      continue;
    }
    // ---- epilogue: threshold filter ------------------------------------------------------------
    bool any = false;
    if constexpr (VAR == 4) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(acc[nb]));
    } else {
      // max tree per query block, then one compare (see filter_f16_m16_kernel)
      float d[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float m = vmax3(acc[nb][0], acc[nb][1], acc[nb][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m = vmax3(m, acc[nb][r], acc[nb][r + 1]);
        m = vmax3(m, acc[nb][15], acc[nb][15]);
        d[nb] = m - thr_s[nb];
      }
      any = (NB == 2 ? vmax3(d[0], d[NB - 1], d[NB - 1]) : d[0]) >= 0.f;
    }
    if (__builtin_amdgcn_ballot_w64(any)) {
      // rare path: log the survivors in this wave's own region (plain 16-byte stores, no atomics,
      // nothing to wait for); scatter_own_log files them under their queries when the stream is done.
      const uint32_t row0 = tile_phys(t) * FILTER_ROWS;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[nb][r];
          const bool hit = v >= thr_s[nb];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m) {
            const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
            if (hit && idx < FILTER_LOGCAP) {
              const uint32_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hsel;
              mylog[idx] = Hit{v * inv_s[nb], row, qid[nb], 0u};
            }
            wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
          }
        }
      }
    }
  }
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  if constexpr (VAR == 0) scatter_own_log(mylog, wcnt, sa, lane);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the two speculative stages before exit
}

// ------------------------------------------------------------------------------------------------
// Same kernel on v_mfma_f32_16x16x32_f16 (64 queries per wave = 4 blocks of 16, tile = 2 blocks of 16
// rows, 24 k-steps of 32): identical bytes, registers (384 for B, 32 accumulators) and LDS reads, twice
// as many half-size MFMAs.  The chip holds a higher clock on this shape under matrix load
// (MI355X_MICROARCH.md "DVFS give-back" item 7), which is the only reason this variant exists.
// ------------------------------------------------------------------------------------------------
#define NVDB_MFMA16_ZERO_A(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b))
#define NVDB_MFMA16_ZERO_V(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b))
#define NVDB_MFMA16_ACC_A(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define NVDB_MFMA16_ACC_V(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// SYNC: the QT workgroups that stream the same rows (different query tiles, same XCD label) keep within
// SYNC_LEAD tiles of each other, so that the XCD's L2 serves QT-1 of the QT reads of every tile.  Speed
// only: a leader that does not see its siblings advance (other XCD, not resident) gives up after a bounded
// spin; nothing is ever read through this channel except the progress counters themselves.
constexpr uint32_t SYNC_MAX_SPINS = 400;

// Sibling rendezvous (see filter_f16_m16_kernel): wave 0 publishes its tile index and waits while it is more than
// `lead` tiles ahead of the slowest workgroup of its row stream; bounded, and abandoned after 3 time-outs.
__device__ __forceinline__ void sibling_rendezvous(uint32_t* myprog, uint32_t qt, uint32_t t, uint32_t lead, uint32_t& strikes, int lane) {
  if (lane == 0) __hip_atomic_store(myprog + qt, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (uint32_t spin = 0; strikes < 3; ++spin) {
    if (spin == SYNC_MAX_SPINS) { ++strikes; break; }
    uint32_t p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile("s_load_dwordx8 s[88:95], %8, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 %0, s88\n\ts_mov_b32 %1, s89\n\ts_mov_b32 %2, s90\n\ts_mov_b32 %3, s91\n\t"
                 "s_mov_b32 %4, s92\n\ts_mov_b32 %5, s93\n\ts_mov_b32 %6, s94\n\ts_mov_b32 %7, s95"
                 : "=s"(p0), "=s"(p1), "=s"(p2), "=s"(p3), "=s"(p4), "=s"(p5), "=s"(p6), "=s"(p7)
                 : "s"(myprog)
                 : "memory", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
    uint32_t lo = p0 < p1 ? p0 : p1;
    lo = lo < p2 ? lo : p2; lo = lo < p3 ? lo : p3; lo = lo < p4 ? lo : p4;
    lo = lo < p5 ? lo : p5; lo = lo < p6 ? lo : p6; lo = lo < p7 ? lo : p7;
    if (lo >= t || t - lo <= lead) break;             // nobody is more than `lead` tiles behind me
    __builtin_amdgcn_s_sleep(8);
  }
}


// STAMP: diagnostic build only (nvdb_hip_debug_clock): wave 0 stamps s_memtime / s_memrealtime around the tile
// loop and stores the two differences behind the progress counters, where nothing else reads them.
// VAR (with STAMP only; results are wrong): 1 = no direct-to-LDS loads in the loop, 5 = no LDS reads (MFMA on whatever
// the ring registers hold), 15 = neither (bare MFMA stream + barrier + epilogue), 16 = 15 without the epilogue compares, 17 = 16 without the barrier.
// MB x NQB = 16-row blocks per tile x 16-query blocks per wave: 2 x 4 (32-row tiles, 64 queries per wave) for
// DIM <= 768; 1 x 2 (16-row tiles, 32 queries per wave) for DIM up to 1536 -- the wave's B fragments are NQB*DIM/8
// registers either way (384 at the two corners), and a stage stays 48 KB.
// WPB = waves per workgroup: 4 (one per SIMD, 512 registers each) or 8 (two per SIMD, 256 registers each, NQB = 2:
// the second wave of a SIMD fills the gaps the first leaves at tile boundaries; twice the LDS reads per MFMA)
template <int DIM, int RING = 6, bool SYNC = false, bool STAMP = false, int VAR = 0, int MB = 2, int NQB = 4, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, 1) void filter_f16_m16_kernel(
    const _Float16* __restrict__ rows, uint32_t row_lo, uint32_t row_hi, const _Float16* __restrict__ q16,
    uint32_t nq, uint32_t QT, const float* __restrict__ thr, const float* __restrict__ qscale,
    const float* __restrict__ qinv, Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t* __restrict__ prog,
    uint32_t sync_mask, uint32_t sync_lead) {
  constexpr int KS = DIM / 32;                     // k-steps of 32
  constexpr int ROW_BYTES = DIM * 2;
  constexpr int TROWS = 16 * MB;                   // corpus rows per tile
  constexpr int STAGE_BYTES = TROWS * ROW_BYTES;
  constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / WPB;
  constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;
  constexpr int NFRAG = NQB * KS, AMAX = WPB == 8 ? 32 : 64;      // two waves per SIMD: 128 AGPRs + 128 VGPRs each
  constexpr int NFRAG_A = NFRAG < AMAX ? NFRAG : AMAX, NFRAG_V = NFRAG - NFRAG_A;
  constexpr int NREAD = MB * KS;                   // A fragments per tile (MB row blocks x KS)
  static_assert(DIM % 128 == 0 && PIECES % WPB == 0 && NREAD % PPW == 0 && (WPB == 4 || WPB == 8), "shape");
  static_assert((MB == 1 || MB == 2 || MB == 4) && (NQB == 1 || NQB == 2 || NQB == 4) && NQB * KS * 4 <= 384 && STAGE_BYTES * FILTER_STAGES <= 160 * 1024, "registers / LDS");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x15 = lane & 15, g4 = lane >> 4;
  const uint32_t wave_gid = blockIdx.x * WPB + wave;

  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) { const uint32_t xcd = b & 7u, i = b >> 3; qt = i % QT; stream = (i / QT) * 8u + xcd; }
  else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / TROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // B fragment f = nb*KS + s: query block nb (16 queries), k-step s; lane (x15,g4) holds
  // q16[query x15 of the block][32 s + 8 g4 .. +8]
  const uint32_t qbase = qt * (16u * NQB * WPB) + wave * (16u * NQB);
  float4_t bqa[NFRAG_A];
  float4_t bqv[NFRAG_V > 0 ? NFRAG_V : 1];
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) {
    const int nb = f / KS, s = f % KS;
    const float4_t v = *reinterpret_cast<const float4_t*>(q16 + static_cast<uint64_t>(qbase + nb * 16 + x15) * DIM + 32 * s + 8 * g4);
    if (f < NFRAG_A) bqa[f] = v; else bqv[f - NFRAG_A] = v;
  }
#pragma unroll
  for (int f = 0; f < NFRAG_A; ++f) asm volatile("" ::"a"(bqa[f]));
#pragma unroll
  for (int f = 0; f < NFRAG_V; ++f) asm volatile("" ::"v"(bqv[f]));
  float thr_s[NQB], inv_s[NQB];
  uint32_t qid[NQB];
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb) {
    qid[nb] = qbase + nb * 16 + x15;
    const bool real = qid[nb] < nq;
    thr_s[nb] = real ? thr[qid[nb]] * qscale[qid[nb]] : __builtin_huge_valf();
    inv_s[nb] = real ? qinv[qid[nb]] : 0.f;
    asm volatile("" ::"v"(thr_s[nb]), "v"(inv_s[nb]));
  }
  const bool wave_has_queries = qbase < nq;

  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CHUNKS_PER_ROW, cpos = P % CHUNKS_PER_ROW;
    src_off[i] = r * ROW_BYTES + ((cpos ^ (r & 15u)) << 4);
  }
  // A fragment (mb,s): lane (x15,g4) reads chunk 4s+g4 of row 16mb+x15, stored at position (4s+g4)^x15:
  //   a16 ^ ((s&3) << 6)  +  256*(s>>2)  +  16*ROW_BYTES*mb
  const uint32_t a16 = static_cast<uint32_t>(x15) * ROW_BYTES + ((static_cast<uint32_t>(g4) ^ static_cast<uint32_t>(x15)) << 4);

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  const uint32_t g_lo = row_lo / TROWS + t_lo;                // logical index of my first tile in the whole corpus
  auto tile_phys = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa); };
  auto tile_ptr = [&](uint32_t t_rel) -> const char* {
    return gbase + static_cast<uint64_t>(tile_phys(t_rel)) * TROWS * ROW_BYTES;
  };
  auto issue_piece = [&](const char* tile, uint32_t buf, int i) {
    glds16(src_off[i], tile, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_piece(tile_ptr(0), 0, i);
#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_piece(tile_ptr(1), 1, i);

  constexpr int PIECE_EVERY = NREAD / PPW;
  uint32_t wcnt = 0;
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;
  // progress counters of this stream's workgroups: prog[stream*8 + qt], 32-byte aligned group, unused slots 0xFFFFFFFF
  uint32_t* myprog = prog + static_cast<uint64_t>(stream) * 8;

  uint32_t sync_strikes = 0;                       // rendezvous that timed out; after 3 this workgroup stops waiting
  uint64_t stamp_c = 0, stamp_r = 0;
  if constexpr (STAMP) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  // 8-wave build: the two waves of a SIMD leave the per-tile barrier together, multiply together and would test together, with
  // the matrix pipe idle meanwhile.  Waves 4..7 therefore test a tile's accumulators AFTER the next barrier (LATE): their test
  // runs beside the partner wave's ring priming and first MFMAs, the partner's test beside their last MFMAs.
  const bool late = (WPB == 8) && (VAR == 0 || VAR == 21) && wave >= 4;
  float4_t acc[MB][NQB];
  auto epilogue = [&](uint32_t te) {
    // "does any of my 32 scores reach its query's threshold": a max tree per query block (v_max3_f32), one subtract
    // per block, one compare in all -- 23 vector instructions.  (32 compares into SGPR pairs + 32 s_or_b64 cost
    // ~770 cycles per tile, a quarter of the tile's MFMA time: profiles/r01d_clock_ablation.txt.)
    float dmax[NQB];
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb) {
      float v[4 * MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * mb + r] = acc[mb][nb][r];
      float m = vmax3(v[0], v[1], v[2]);
#pragma unroll
      for (int x = 3; x + 1 < 4 * MB; x += 2) m = vmax3(m, v[x], v[x + 1]);
      m = vmax3(m, v[4 * MB - 1], v[4 * MB - 1]);                  // 4*MB is even: one value is left over
      dmax[nb] = m - thr_s[nb];                      // >= 0 iff m >= thr (a difference of floats never rounds across 0)
    }
    const bool any = (NQB == 4 ? vmax3(vmax3(dmax[0], dmax[1], dmax[NQB / 2]), dmax[NQB - 1], dmax[NQB - 1])
                               : NQB == 2 ? vmax3(dmax[0], dmax[NQB - 1], dmax[NQB - 1]) : dmax[0]) >= 0.f;
    if (__builtin_amdgcn_ballot_w64(any)) {
      const uint32_t row0 = tile_phys(te) * TROWS;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[mb][nb][r];
            const bool hit = v >= thr_s[nb];
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            if (m) {
              const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
              if (hit && idx < FILTER_LOGCAP) mylog[idx] = Hit{v * inv_s[nb], row0 + 16 * mb + 4 * g4 + r, qid[nb], 0u};
              wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
            }
          }
    }
  };
  for (uint32_t t = 0; t < NT; ++t) {
    if constexpr (SYNC) {
      // wave 0 only; the per-tile barrier holds the other waves back
      if (wave == 0 && (t & sync_mask) == 0) sibling_rendezvous(myprog, qt, t, sync_lead, sync_strikes, lane);
    }
    constexpr bool NO_GLDS = (VAR == 1 || VAR >= 15), NO_READ = (VAR == 5 || VAR >= 15), NO_EPI = (VAR >= 16), NO_BAR = (VAR >= 17);
    if constexpr (!NO_GLDS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    if constexpr (!NO_BAR) __builtin_amdgcn_s_barrier();
    const char* next_tile = tile_ptr(t + 2);
    const uint32_t next_buf = (t + 2) % FILTER_STAGES;
    const char* stage = smem + (t % FILTER_STAGES) * STAGE_BYTES;
    if (!wave_has_queries) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(next_tile, next_buf, i);
      continue;
    }
    if (late && t > 0) epilogue(t - 1);
    auto read_a = [&](int u) -> float4_t {        // u = MB*s + mb
      const int s = u / MB, mb = u % MB;
      return *reinterpret_cast<const float4_t*>(stage + (a16 ^ ((s & 3) << 6)) + (s >> 2) * 256 + mb * 16 * ROW_BYTES);
    };
    float4_t ar[RING];
    if constexpr (NO_READ) {
#pragma unroll
      for (int u = 0; u < RING; ++u) ar[u] = bqv[u];          // random, non-trivial operand bits
    }
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) if constexpr (!NO_READ) ar[u] = read_a(u);
#pragma unroll
    for (int u = 0; u < NREAD; ++u) {
      if constexpr (!NO_READ) { if (u + RING - 1 < NREAD) ar[(u + RING - 1) % RING] = read_a(u + RING - 1); }
      const float4_t a = ar[u % RING];
      const int s = u / MB, mb = u % MB;
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) {
        const int f = nb * KS + s;
        if (s == 0) {
          if (f < NFRAG_A) NVDB_MFMA16_ZERO_A(acc[mb][nb], a, bqa[f < NFRAG_A ? f : 0]);
          else NVDB_MFMA16_ZERO_V(acc[mb][nb], a, bqv[f >= NFRAG_A ? f - NFRAG_A : 0]);
        } else {
          if (f < NFRAG_A) NVDB_MFMA16_ACC_A(acc[mb][nb], a, bqa[f < NFRAG_A ? f : 0]);
          else NVDB_MFMA16_ACC_V(acc[mb][nb], a, bqv[f >= NFRAG_A ? f - NFRAG_A : 0]);
        }
      }
      if constexpr (!NO_GLDS) { if (u % PIECE_EVERY == PIECE_EVERY - 1) issue_piece(next_tile, next_buf, u / PIECE_EVERY); }
    }
    // 32 wait states: MFMA result -> VALU read (the compiler does not see inside the asm)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) asm volatile("" : "+v"(acc[mb][nb]));
    if constexpr (NO_EPI) continue;
    if (!late) epilogue(t);
  }
  if (late && wave_has_queries) epilogue(NT - 1);
  if constexpr (STAMP) {
    const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c, dr = __builtin_amdgcn_s_memrealtime() - stamp_r;
    if (wave == 0 && lane == 0) {
      uint64_t* out = reinterpret_cast<uint64_t*>(prog + static_cast<uint64_t>(gridDim.x) * 8) + static_cast<uint64_t>(blockIdx.x) * 2;
      out[0] = dc; out[1] = dr;
    }
  }
  if constexpr (SYNC) { if (wave == 0 && lane == 0) __hip_atomic_store(myprog + qt, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  scatter_own_log(mylog, wcnt, sa, lane);
}

// ------------------------------------------------------------------------------------------------
// 1536 < DIM <= 3072: the K-split build.  A wave cannot hold more than 384 registers of B fragments, so it keeps ONE
// 16-query block (DIM/8 registers) and a 16-row tile is streamed as two stages, one per half of K (each again
// <= 48 KB); the accumulators live across the two stages and the threshold test runs after the second.  One MFMA per
// A fragment read: LDS-bound at about half the MFMA rate of the 768 build, 64 queries per workgroup -- still far from
// the fp32 VALU path these dims would otherwise take.  Same logging / scatter / rendezvous as filter_f16_m16_kernel.
// ------------------------------------------------------------------------------------------------
template <int DIM, bool SYNC = false>
__global__ __launch_bounds__(256, 1) void filter_f16_k2_kernel(
    const _Float16* __restrict__ rows, uint32_t row_lo, uint32_t row_hi, const _Float16* __restrict__ q16,
    uint32_t nq, uint32_t QT, const float* __restrict__ thr, const float* __restrict__ qscale,
    const float* __restrict__ qinv, Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t* __restrict__ prog,
    uint32_t sync_mask, uint32_t sync_lead) {
  constexpr int RING = 6;
  constexpr int DIMS = DIM / 2;                    // dims per stage
  constexpr int KS = DIMS / 32;                    // k-steps of 32 per stage
  constexpr int ROW_BYTES = DIM * 2, SROW = DIMS * 2;
  constexpr int TROWS = 16;
  constexpr int STAGE_BYTES = TROWS * SROW;
  constexpr int PIECES = STAGE_BYTES / 1024, PPW = PIECES / 4;
  constexpr int CPR = SROW / 16;                   // 16-byte chunks per stage row
  constexpr int NFRAG = 2 * KS, NFRAG_A = NFRAG < 64 ? NFRAG : 64, NFRAG_V = NFRAG - NFRAG_A;
  static_assert(DIMS % 128 == 0 && PIECES % 4 == 0 && KS % PPW == 0 && NFRAG * 4 <= 384 && STAGE_BYTES * FILTER_STAGES <= 160 * 1024, "shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x15 = lane & 15, g4 = lane >> 4;
  const uint32_t wave_gid = blockIdx.x * 4 + wave;

  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) { const uint32_t xcd = b & 7u, i = b >> 3; qt = i % QT; stream = (i / QT) * 8u + xcd; }
  else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / TROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // B fragment f = kh*KS + s: K-half kh, k-step s; lane (x15,g4) holds q16[query x15][kh*DIMS + 32 s + 8 g4 .. +8]
  const uint32_t qbase = qt * 64u + wave * 16u;
  float4_t bqa[NFRAG_A];
  float4_t bqv[NFRAG_V > 0 ? NFRAG_V : 1];
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) {
    const float4_t v = *reinterpret_cast<const float4_t*>(q16 + static_cast<uint64_t>(qbase + x15) * DIM + 32 * f + 8 * g4);
    if (f < NFRAG_A) bqa[f] = v; else bqv[f - NFRAG_A] = v;
  }
#pragma unroll
  for (int f = 0; f < NFRAG_A; ++f) asm volatile("" ::"a"(bqa[f]));
#pragma unroll
  for (int f = 0; f < NFRAG_V; ++f) asm volatile("" ::"v"(bqv[f]));
  const uint32_t qid = qbase + x15;
  const bool real = qid < nq;
  float thr_s = real ? thr[qid] * qscale[qid] : __builtin_huge_valf();
  float inv_s = real ? qinv[qid] : 0.f;
  asm volatile("" ::"v"(thr_s), "v"(inv_s));
  const bool wave_has_queries = qbase < nq;

  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CPR, cpos = P % CPR;
    src_off[i] = r * ROW_BYTES + ((cpos ^ (r & 15u)) << 4);
  }
  const uint32_t a16 = static_cast<uint32_t>(x15) * SROW + ((static_cast<uint32_t>(g4) ^ static_cast<uint32_t>(x15)) << 4);

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  const uint32_t g_lo = row_lo / TROWS + t_lo;
  auto tile_phys = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa); };
  auto stage_ptr = [&](uint32_t t_rel, int kh) -> const char* {
    return gbase + static_cast<uint64_t>(tile_phys(t_rel)) * TROWS * ROW_BYTES + kh * SROW;
  };
  auto issue_piece = [&](const char* src, uint32_t buf, int i) {
    glds16(src_off[i], src, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_piece(stage_ptr(0, 0), 0, i);
#pragma unroll
  for (int i = 0; i < PPW; ++i) issue_piece(stage_ptr(0, 1), 1, i);

  constexpr int PIECE_EVERY = KS / PPW;
  uint32_t wcnt = 0;
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;
  uint32_t* myprog = prog + static_cast<uint64_t>(stream) * 8;
  uint32_t sync_strikes = 0;
  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  for (uint32_t t = 0; t < NT; ++t) {
    if constexpr (SYNC) {
      if (wave == 0 && (t & sync_mask) == 0) sibling_rendezvous(myprog, qt, t, sync_lead, sync_strikes, lane);
    }
    float4_t acc;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const uint32_t st = 2 * t + kh;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
      __builtin_amdgcn_s_barrier();
      const char* next_src = stage_ptr(t + 1, kh);               // stage st+2 = the same half of the next tile
      const uint32_t next_buf = (st + 2) % FILTER_STAGES;
      const char* stage = smem + (st % FILTER_STAGES) * STAGE_BYTES;
      if (!wave_has_queries) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(next_src, next_buf, i);
        continue;
      }
      auto read_a = [&](int s) -> float4_t {
        return *reinterpret_cast<const float4_t*>(stage + (a16 ^ ((s & 3) << 6)) + (s >> 2) * 256);
      };
      float4_t ar[RING];
#pragma unroll
      for (int s = 0; s < RING - 1; ++s) ar[s] = read_a(s);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + RING - 1 < KS) ar[(s + RING - 1) % RING] = read_a(s + RING - 1);
        const float4_t a = ar[s % RING];
        const int f = kh * KS + s;
        if (f == 0) NVDB_MFMA16_ZERO_A(acc, a, bqa[0]);
        else if (f < NFRAG_A) NVDB_MFMA16_ACC_A(acc, a, bqa[f < NFRAG_A ? f : 0]);
        else NVDB_MFMA16_ACC_V(acc, a, bqv[f >= NFRAG_A ? f - NFRAG_A : 0]);
        if (s % PIECE_EVERY == PIECE_EVERY - 1) issue_piece(next_src, next_buf, s / PIECE_EVERY);
      }
    }
    if (!wave_has_queries) continue;
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
    const float m = vmax3(vmax3(acc[0], acc[1], acc[2]), acc[3], acc[3]);
    if (__builtin_amdgcn_ballot_w64(m - thr_s >= 0.f)) {
      const uint32_t row0 = tile_phys(t) * TROWS;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[r];
        const bool hit = v >= thr_s;
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(hit);
        if (mm) {
          const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(mm & ((1ull << lane) - 1ull)));
          if (hit && idx < FILTER_LOGCAP) mylog[idx] = Hit{v * inv_s, row0 + 4 * g4 + r, qid, 0u};
          wcnt += static_cast<uint32_t>(__builtin_popcountll(mm));
        }
      }
    }
  }
  if constexpr (SYNC) { if (wave == 0 && lane == 0) __hip_atomic_store(myprog + qt, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  scatter_own_log(mylog, wcnt, sa, lane);
}

// ================================================================================================
// int8(+scale) corpus: the same streaming structure on the integer matrix cores.
//
// The reference dequantises on the fly and keeps the query in fp32 (src/simd_dot.cpp:160-199).  Here the
// filter quantises the QUERY to 15 bits as two int8 planes, q_i ~ s_q * (128*hi_i + lo_i) with
// hi in [-127,127], lo in [-64,63], and runs two v_mfma_i32_32x32x32_i8 per A fragment (hi plane, lo
// plane); integer accumulation is exact, so the only filter error is the query quantisation:
//   |filter - cpu_score| <= (sqrt(dim) * s_q / 2) * max_row(||x_int8|| * scale)  (+ fp32 rounding),
// which prep_q8_kernel turns into ebound/slack exactly like the fp16 path.  One wave = 32 queries
// (two planes x 24 k-steps = 48 fragments = 192 AGPRs), one workgroup = 128 queries.
// Per stage: 32 rows x DIM bytes of corpus + the tile's 32 row scales (one extra 1 KB piece per wave,
// into the wave's own LDS slot, so that no ordinary global load -- whose compiler-inserted vmcnt wait
// would drain the stream -- is needed in the loop).
// ================================================================================================
static __global__ __launch_bounds__(256) void prep_q8_kernel(const float* __restrict__ q32, uint32_t nq, uint32_t dim, uint32_t sdim,
                                                      float max_row_norm, float corpus_resid, signed char* __restrict__ qhi,
                                                      signed char* __restrict__ qlo, float* __restrict__ qscale,
                                                      float* __restrict__ qinv, float* __restrict__ ebound,
                                                      float* __restrict__ slack, float* __restrict__ qdelta, uint32_t* __restrict__ overflow, uint32_t lo_bits, PrepInit pi) {
  __shared__ float red_max[4], red_ss[4];
  __shared__ uint32_t red_lo[4];
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const float* src = prep_init(pi, q32, q, tid, nq, dim, overflow);
  // sdim >= dim: row stride of the two planes = the (zero-padded) dim the filter kernel is instantiated for
  if (q >= nq) {
    for (uint32_t i = tid; i < sdim; i += 256) { qhi[static_cast<uint64_t>(q) * sdim + i] = 0; qlo[static_cast<uint64_t>(q) * sdim + i] = 0; }
    if (tid == 0) { qscale[q] = 1.f; qinv[q] = 1.f; ebound[q] = 0.f; slack[q] = 0.f; qdelta[q] = 0.f; }
    return;
  }
  float mx = 0.f, ss = 0.f;
  for (uint32_t i = tid; i < dim; i += 256) { const float v = src[i]; mx = fmaxf(mx, fabsf(v)); ss = __builtin_fmaf(v, v, ss); }
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); ss += __shfl_xor(ss, o); }
  if ((tid & 63) == 0) { red_max[tid >> 6] = mx; red_ss[tid >> 6] = ss; }
  __syncthreads();
  mx = fmaxf(fmaxf(red_max[0], red_max[1]), fmaxf(red_max[2], red_max[3]));
  ss = (red_ss[0] + red_ss[1]) + (red_ss[2] + red_ss[3]);
  const int tmax = 127 << lo_bits, half = 1 << (lo_bits - 1);   // lo_bits = 7: 15-bit queries, |t| <= 16256
  const float sq0 = (mx > 0.f && mx < 3.0e38f) ? mx / static_cast<float>(tmax) : 1.f;
  const float isq0 = 1.0f / sq0;
  float sq_adj = 1.f;
  // |H| = |<row, hi plane>| must stay below 2^23 for the biased accumulators of filter_i8p_kernel (I8_HI_L1_MAX * 128 < 2^23).
  // Only a query with nearly all components at full magnitude exceeds that (mean |hi| > 84 at d = 768): it is quantised
  // more coarsely (the error terms below follow sq).
  if (static_cast<uint64_t>(dim) * 127u > I8_HI_L1_MAX) {
    __shared__ uint32_t red_l1[4];
    uint32_t l1 = 0;
    for (uint32_t i = tid; i < dim; i += 256) {
      int t = static_cast<int>(rintf(src[i] * isq0));
      t = t > tmax ? tmax : (t < -tmax ? -tmax : t);
      const int hi = (t + half) >> lo_bits;
      l1 += static_cast<uint32_t>(hi < 0 ? -hi : hi);
    }
    for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o);
    if ((tid & 63) == 0) red_l1[tid >> 6] = l1;
    __syncthreads();
    l1 = (red_l1[0] + red_l1[1]) + (red_l1[2] + red_l1[3]);
    if (l1 > I8_HI_L1_MAX - dim) {                 // (- dim: every |hi| may round up by one at the coarser scale)
      const float shrink = static_cast<float>(l1) / static_cast<float>(I8_HI_L1_MAX - dim) * 1.001f;
      sq_adj = shrink;
    }
  }
  const float sq = sq0 * sq_adj;
  const float isq = 1.0f / sq;
  uint32_t lo2 = 0;                               // sum of lo^2 (exact: <= dim * 4096)
  for (uint32_t i = tid; i < dim; i += 256) {
    int t = static_cast<int>(rintf(src[i] * isq));
    t = t > tmax ? tmax : (t < -tmax ? -tmax : t);
    const int hi = (t + half) >> lo_bits;         // floor((t + half) / 2^lo_bits), arithmetic shift
    const int lo = t - (hi << lo_bits);           // in [-half, half)
    qhi[static_cast<uint64_t>(q) * sdim + i] = static_cast<signed char>(hi);
    qlo[static_cast<uint64_t>(q) * sdim + i] = static_cast<signed char>(lo);
    lo2 += static_cast<uint32_t>(lo * lo);
  }
  for (uint32_t i = dim + tid; i < sdim; i += 256) { qhi[static_cast<uint64_t>(q) * sdim + i] = 0; qlo[static_cast<uint64_t>(q) * sdim + i] = 0; }
  for (int o = 32; o > 0; o >>= 1) lo2 += __shfl_xor(lo2, o);
  if ((tid & 63) == 0) red_lo[tid >> 6] = lo2;
  __syncthreads();
  if (tid == 0) {
    // what the lo plane can add to a row's filter value, in units of s_q (Cauchy-Schwarz):
    //   |sum lo_i x_i| * scale <= ||lo|| * max_row(||x_int8|| * scale)       (filter_i8w_kernel's first stage)
    qdelta[q] = sqrtf(static_cast<float>(red_lo[0] + red_lo[1] + red_lo[2] + red_lo[3])) * max_row_norm * 1.0001f;

    const float nrm = sqrtf(ss) * 1.0001f;
    // quantisation: |dq_i| <= 0.5 s_q (+ the rounding of q*isq: <= 2^-23 |q_i|); fp32 chains: 1e-5 ||q||
    // corpus_resid > 0: the rows streamed are an int8 SHADOW of an fp16 / fp32 corpus (shadow_q8_kernel): |<q, x> - <q, x_shadow>| <= ||q|| * max ||x - x_shadow||
    const float eb = (0.5005f * sqrtf(static_cast<float>(dim)) * sq + 1.0e-5f * nrm) * max_row_norm * 1.001f + nrm * corpus_resid * 1.001f + 1e-30f;
    qscale[q] = isq; qinv[q] = sq; ebound[q] = eb; slack[q] = 2.f * eb;
    if (!(ss < 3.0e38f)) { overflow[q] = 1u; ebound[q] = -1.f; }     // NaN / infinite query: exact path (see prep_q16_kernel)
  }
}

// LDS image of an int8 tile: chunk c (16 bytes) of row r sits at chunk position swz_chunk(c, r) of its row, so that the 16 rows a
// ds_read_b128 group touches at one k-offset fall into 16 different 16-byte bank groups.  Row strides that are multiples of 256
// bytes start every row in the same bank group: c ^ (r & 15).  Odd multiples of 128 bytes (d = 384, the reference's own data
// dimension) start consecutive rows 8 bank groups apart -- (r & 1) already picks the half, (r >> 1) & 7 the place inside it.
template <int ROW_BYTES> __device__ constexpr bool swz16() { return (ROW_BYTES / 16) % 16 == 0; }
template <int ROW_BYTES> __device__ inline uint32_t swz_chunk(uint32_t c, uint32_t r) {
  static_assert(ROW_BYTES % 128 == 0, "int8 row stride: a multiple of 128 bytes");
  return swz16<ROW_BYTES>() ? c ^ (r & 15u) : c ^ ((r >> 1) & 7u);
}
// v_mfma_i32_32x32x32_i8 A fragment of k-step s: chunk 2 s + hsel of row r31; 2 s + hsel == 2 s ^ hsel, so the swizzled position is
// a_base ^ (the k-step's bits inside the swizzle group) + the byte offset of the group
template <int ROW_BYTES> __device__ inline uint32_t a_base_i8(uint32_t r31, uint32_t hsel) { return r31 * ROW_BYTES + (swz_chunk<ROW_BYTES>(hsel, r31) << 4); }
template <int ROW_BYTES> __device__ inline uint32_t a_addr_i8(uint32_t a_base, int s) {
  return swz16<ROW_BYTES>() ? (a_base ^ ((s & 7) << 5)) + (s >> 3) * 256 : (a_base ^ ((s & 3) << 5)) + (s >> 2) * 128;
}

#define NVDB_MFMA_I8_ZERO(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b))
#define NVDB_MFMA_I8_ACC(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
// first k-step of a block whose accumulators start at a constant (16 VGPRs holding it: srcC and vDst share a register class)
#define NVDB_MFMA_I8_FROM(acc, a, b, c0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c0))

// BOOT = true: bootstrap build (see filter_f16_kernel VAR 7): best (score,row) per tile and query -> cand lists
// (hitlog then points at the candidate lists, aux is their stride).
template <int DIM, bool BOOT = false, int RING = 6, bool SYNC = false>
__global__ __launch_bounds__(256, 1) void filter_i8_kernel(
    const signed char* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const signed char* __restrict__ qhi, const signed char* __restrict__ qlo, uint32_t nq, uint32_t QT,
    const float* __restrict__ thr, const float* __restrict__ qscale, const float* __restrict__ qinv,
    Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t aux, uint32_t* __restrict__ prog, uint32_t sync_mask, uint32_t sync_lead) {
  constexpr int KSTEPS = DIM / 32;                 // v_mfma_i32_32x32x32_i8: K = 32
  constexpr int ROW_BYTES = DIM;
  constexpr int DATA_BYTES = FILTER_ROWS * ROW_BYTES;
  constexpr int STAGE_BYTES = DATA_BYTES + 4 * 1024;   // + one 1 KB scales slot per wave
  constexpr int PIECES = DATA_BYTES / 1024;
  constexpr int PPW = PIECES / 4;                  // corpus pieces per wave
  constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;
  static_assert(DIM % 128 == 0, "int8 row stride: a multiple of 128 bytes (swz_chunk)");
  static_assert(PIECES % 4 == 0 && KSTEPS % PPW == 0 && 2 * KSTEPS <= 64, "shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, hsel = lane >> 5;
  const uint32_t wave_gid = blockIdx.x * 4 + wave;

  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) { const uint32_t xcd = b & 7u, i = b >> 3; qt = i % QT; stream = (i / QT) * 8u + xcd; }
  else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / FILTER_ROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // stationary operand: 32 queries x two int8 planes x all of K, in AGPRs.  Lane (r31,hsel) holds bytes
  // [32 s + 16 hsel, +16) of its query -- the same k-slice the A fragment holds.
  const uint32_t qbase = qt * 128u + wave * 32u;
  float4_t bq[2 * KSTEPS];
#pragma unroll
  for (int f = 0; f < 2 * KSTEPS; ++f) {
    const signed char* plane = (f < KSTEPS) ? qhi : qlo;
    const int s = f % KSTEPS;
    bq[f] = *reinterpret_cast<const float4_t*>(plane + static_cast<uint64_t>(qbase + r31) * DIM + 32 * s + 16 * hsel);
  }
#pragma unroll
  for (int f = 0; f < 2 * KSTEPS; ++f) asm volatile("" ::"a"(bq[f]));
  const uint32_t qid = qbase + r31;
  const bool real = qid < nq;
  float thr_s = real ? thr[qid] * qscale[qid] : __builtin_huge_valf();   // threshold in units of s_q
  float inv_s = real ? qinv[qid] : 0.f;
  asm volatile("" ::"v"(thr_s), "v"(inv_s));
  const bool wave_has_queries = qbase < nq;

  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CHUNKS_PER_ROW, cpos = P % CHUNKS_PER_ROW;
    src_off[i] = r * ROW_BYTES + (swz_chunk<ROW_BYTES>(cpos, r) << 4);
  }
  const uint32_t sc_off = (lane & 7) * 16;         // 32 row scales = 128 bytes; lanes >= 8 re-load the same chunks
  const uint32_t a_base = a_base_i8<ROW_BYTES>(static_cast<uint32_t>(r31), static_cast<uint32_t>(hsel));

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  const uint32_t g_lo = row_lo / FILTER_ROWS + t_lo;          // logical index of my first tile in the whole corpus
  auto tile_row0 = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa) * FILTER_ROWS; };
  auto issue_piece = [&](uint32_t row0, uint32_t buf, int i) {
    glds16(src_off[i], gbase + static_cast<uint64_t>(row0) * ROW_BYTES, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
  auto issue_scales = [&](uint32_t row0, uint32_t buf) {
    glds16(sc_off, reinterpret_cast<const char*>(scales + row0), lds_base + buf * STAGE_BYTES + DATA_BYTES + wave * 1024);
  };
#pragma unroll
  for (int st = 0; st < FILTER_STAGES_I8 - 1; ++st) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(tile_row0(st), st, i);
    issue_scales(tile_row0(st), st);
  }

  constexpr int PIECE_EVERY = KSTEPS / PPW;
  uint32_t wcnt = 0;
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;

  uint32_t sync_strikes = 0;
  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  for (uint32_t t = 0; t < NT; ++t) {
    if constexpr (SYNC) {
      if (wave == 0 && (t & sync_mask) == 0) sibling_rendezvous(prog + static_cast<uint64_t>(stream) * 8, qt, t, sync_lead, sync_strikes, lane);
    }
    // my pieces of tile t have landed once all but the newest FILTER_STAGES_I8-2 tiles' loads are complete
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((FILTER_STAGES_I8 - 2) * (PPW + 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    const uint32_t next_row0 = tile_row0(t + FILTER_STAGES_I8 - 1), next_buf = (t + FILTER_STAGES_I8 - 1) % FILTER_STAGES_I8;
    const char* stage = smem + (t % FILTER_STAGES_I8) * STAGE_BYTES;
    if (!wave_has_queries) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(next_row0, next_buf, i);
      issue_scales(next_row0, next_buf);
      continue;
    }
    auto read_a = [&](int s) -> float4_t {
      return *reinterpret_cast<const float4_t*>(stage + a_addr_i8<ROW_BYTES>(a_base, s));
    };
    // this tile's 16 row scales for my lanes: read now, used in the epilogue (their latency hides behind the MFMAs)
    const float* sc_lds = reinterpret_cast<const float*>(stage + DATA_BYTES + wave * 1024);
    float4 sc4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc4[j] = *reinterpret_cast<const float4*>(sc_lds + 8 * j + 4 * hsel);
    float4_t ar[RING];
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) ar[s] = read_a(s);
    intx16 acc_hi, acc_lo;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + RING - 1 < KSTEPS) ar[(s + RING - 1) % RING] = read_a(s + RING - 1);
      const float4_t a = ar[s % RING];
      if (s == 0) { NVDB_MFMA_I8_ZERO(acc_hi, a, bq[0]); NVDB_MFMA_I8_ZERO(acc_lo, a, bq[KSTEPS]); }
      else { NVDB_MFMA_I8_ACC(acc_hi, a, bq[s]); NVDB_MFMA_I8_ACC(acc_lo, a, bq[KSTEPS + s]); }
      if (s % PIECE_EVERY == PIECE_EVERY - 1) issue_piece(next_row0, next_buf, s / PIECE_EVERY);
      if (s == 1) issue_scales(next_row0, next_buf);
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc_hi), "+v"(acc_lo));

    // epilogue: f = (128*H + L) * scale_row  compared with thr/s_q ; rows (r&3) + 8*(r>>2) + 4*hsel
    // 128*H + L in int32 (|.| < 768*127*(127*128 + 64) < 2^31), one conversion: RN(128 H + L), the value the
    // fp32 fma(float(H), 128, float(L)) gives as well (float(H) is exact below 2^24)
    static_assert(DIM <= 768, "int32 range of 128*H + L");
    float fv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float scv[4] = {sc4[j].x, sc4[j].y, sc4[j].z, sc4[j].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * j + i;
        fv[r] = static_cast<float>(shl_i32(acc_hi[r], sa.lo_bits) + acc_lo[r]) * scv[i];
      }
    }
    float fmx = vmax3(fv[0], fv[1], fv[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) fmx = vmax3(fmx, fv[r], fv[r + 1]);
    const bool any = vmax3(fmx, fv[15], fv[15]) >= thr_s;
    if constexpr (BOOT) {
      const uint32_t tile = t_lo + t;                          // list slot: logical
      const uint32_t row0b = tile_row0(t);                     // rows: physical
      float best = -__builtin_huge_valf();                     // padding rows are not rows (see filter_f16_kernel)
      uint32_t brow = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool real = row0b + (r & 3) + 8 * (r >> 2) + 4 * hsel < sa.n_rows;
        const bool gt = real && fv[r] > best;
        best = gt ? fv[r] : best; brow = gt ? static_cast<uint32_t>(r) : brow;
      }
      uint32_t grow = row0b + (brow & 3) + 8 * (brow >> 2) + 4 * hsel;
      const float obest = __shfl_xor(best, 32);
      const uint32_t orow = static_cast<uint32_t>(__shfl_xor(static_cast<int>(grow), 32));
      if (obest > best || (obest == best && orow < grow)) { best = obest; grow = orow; }
      if (hsel == 0 && qid < nq) reinterpret_cast<Cand*>(hitlog)[static_cast<uint64_t>(qid) * aux + tile] = Cand{best * inv_s, grow};
      continue;
    }
    if (__builtin_amdgcn_ballot_w64(any)) {
      const uint32_t row0 = tile_row0(t);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool hit = fv[r] >= thr_s;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (m) {
          const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
          if (hit && idx < FILTER_LOGCAP) mylog[idx] = Hit{fv[r] * inv_s, row0 + (r & 3) + 8 * (r >> 2) + 4 * hsel, qid, 0u};
          wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
        }
      }
    }
  }
  if constexpr (SYNC) { if (wave == 0 && lane == 0) __hip_atomic_store(prog + static_cast<uint64_t>(stream) * 8 + qt, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  if constexpr (!BOOT) scatter_own_log(mylog, wcnt, sa, lane);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define NVDB_MFMA_I8_ZERO_V(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b))
#define NVDB_MFMA_I8_ACC_V(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// sum of an int over the 64 lanes, uniform result: two quad permutes, half-row mirror, row mirror (DPP, no LDS round trips),
// then the four row sums through SGPRs
__device__ __forceinline__ int wave_sum_i32(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);   // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror: every lane of a 16-lane row holds the row's sum
  return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48);
}

__device__ __forceinline__ int imax3(int a, int b, int c) {
  int r;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ------------------------------------------------------------------------------------------------
// int8 two-stage build.  One wave = NB x 32 queries (NB = 2 for batches > 128), but only their HI plane is resident
// (NB x 24 fragments = up to 192 AGPRs) and only the hi plane is multiplied for every tile: half the matrix work of
// filter_i8_kernel per query.  The lo plane can change a row's filter value by at most
//   delta_q = ||lo_q|| * max_row(||x_int8|| * scale)        (units of s_q; prep_q8_kernel)
// so a row can only reach its threshold T if its hi-plane value reaches T - delta_q.  Per tile:
//   stage 1 (always): max_r(128 * H_r * scale_r) >= T - delta ?   per 32-query block, a v_max3 tree
//   stage 2 (rare): the lo plane of that 32-query block is multiplied after all -- its fragments come from
//            global memory (L2), the A fragments are re-read from the LDS stage, which is still intact --
//            and the full value (128 H + L) * scale is compared with T exactly as filter_i8_kernel does.
// The survivors logged are therefore exactly those of filter_i8_kernel, with the same filter scores.
// ------------------------------------------------------------------------------------------------
// MB = 32-row blocks per tile: 2 -> 64-row tiles (a 52 KB stage, three of them) halve the per-tile costs (barrier,
// loop-top scalar code, rendezvous, MFMA drain) per row; 1 -> 32-row tiles, five stages (debug / A-B only).
// STAMP / VAR: diagnostic builds only (libnvdb_hip_dev.so, nvdb_hip_debug_clock_i8; results are wrong for VAR != 0):
// STAMP = s_memtime / s_memrealtime around the tile loop (written behind the rendezvous counters); VAR 1 = never run
// stage 2 (no lo-plane pass), 2 = no stage-1 test either (MFMAs + stream only), 3 = 2 + no per-tile barrier.
template <int DIM, int NB = 2, int RING = 6, bool SYNC = false, int MB = 2, bool STAMP = false, int VAR = 0>
__global__ __launch_bounds__(256, 1) void filter_i8w_kernel(
    const signed char* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const signed char* __restrict__ qhi, const signed char* __restrict__ qlo, uint32_t nq, uint32_t QT,
    const float* __restrict__ thr, const float* __restrict__ qscale, const float* __restrict__ qinv,
    const float* __restrict__ qdelta, Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t* __restrict__ prog,
    uint32_t sync_mask, uint32_t sync_lead, uint32_t* __restrict__ stage_counts) {
  constexpr int KSTEPS = DIM / 32;
  constexpr int ROW_BYTES = DIM;
  constexpr int TROWS = FILTER_ROWS * MB;          // corpus rows per tile
  constexpr int NSTAGE = (MB == 2 || DIM > 768) ? 3 : FILTER_STAGES_I8;
  constexpr int DATA_BYTES = TROWS * ROW_BYTES;
  constexpr int STAGE_BYTES = DATA_BYTES + 4 * 1024;
  constexpr int PIECES = DATA_BYTES / 1024;
  constexpr int PPW = PIECES / 4;
  constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;
  // dims up to 1536 (768 < DIM: NB = 1, MB = 1 -- 32 queries per wave, 192 AGPRs of hi-plane fragments, 32-row tiles in three
  // 52 KB stages).  int32 range of (H << 7) + L at any dim: prep_q8_kernel keeps the L1 norm of a query's hi plane below
  // I8_HI_L1_MAX, so |H| < 2^23 whatever the dim.
  static_assert(DIM % 128 == 0 && DIM <= 1536 && (DIM <= 768 || (NB == 1 && MB == 1)), "row stride multiple of 128 bytes (swz_chunk); fragments of one wave <= 192 registers");
  static_assert(PIECES % 4 == 0 && (MB * KSTEPS) % PPW == 0 && NB * KSTEPS <= 64 && KSTEPS % 2 == 0, "shape");
  static_assert(NB == 1 || NB == 2, "one or two 32-query blocks per wave");
  static_assert((MB == 1 || MB == 2) && NSTAGE * STAGE_BYTES <= 160 * 1024 && (NSTAGE - 2) * (PPW + 1) < 64, "LDS / vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, hsel = lane >> 5;
  const uint32_t wave_gid = blockIdx.x * 4 + wave;

  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) { const uint32_t xcd = b & 7u, i = b >> 3; qt = i % QT; stream = (i / QT) * 8u + xcd; }
  else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / TROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // stationary operand: hi plane of this wave's NB blocks of 32 queries, all of K, in AGPRs
  const uint32_t qbase = qt * (128u * NB) + wave * (32u * NB);
  float4_t bq[NB * KSTEPS];
#pragma unroll
  for (int f = 0; f < NB * KSTEPS; ++f) {
    const int nb = f / KSTEPS, s = f % KSTEPS;
    bq[f] = *reinterpret_cast<const float4_t*>(qhi + static_cast<uint64_t>(qbase + nb * 32 + r31) * DIM + 32 * s + 16 * hsel);
  }
#pragma unroll
  for (int f = 0; f < NB * KSTEPS; ++f) asm volatile("" ::"a"(bq[f]));
  uint32_t qid[NB];
  float thr_s[NB], t1q[NB], inv_s[NB];
  const float lo_unit = __builtin_bit_cast(float, (127u - sa.lo_bits) << 23);   // 2^-lo_bits: the first stage compares H alone
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    qid[nb] = qbase + nb * 32 + r31;
    const bool real = qid[nb] < nq;
    thr_s[nb] = real ? thr[qid[nb]] * qscale[qid[nb]] : __builtin_huge_valf();   // threshold in units of s_q
    inv_s[nb] = real ? qinv[qid[nb]] : 0.f;
    // first-stage threshold for H * scale (the factor 128 moved to this side, exactly): a little below
    // T - delta, the margin covering the fp32 roundings of both stages' values
    const float T = thr_s[nb];
    t1q[nb] = real ? (T - (1.001f * qdelta[qid[nb]] + 2e-6f * fabsf(T) + 1e-5f)) * lo_unit : __builtin_huge_valf();
    asm volatile("" ::"v"(thr_s[nb]), "v"(inv_s[nb]), "v"(t1q[nb]));
  }
  const bool wave_has_queries = qbase < nq;

  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CHUNKS_PER_ROW, cpos = P % CHUNKS_PER_ROW;
    src_off[i] = r * ROW_BYTES + (swz_chunk<ROW_BYTES>(cpos, r) << 4);
  }
  const uint32_t sc_off = (lane & (8 * MB - 1)) * 16;   // TROWS row scales = 128 * MB bytes; the other lanes re-load the same chunks
  const uint32_t a_base = a_base_i8<ROW_BYTES>(static_cast<uint32_t>(r31), static_cast<uint32_t>(hsel));

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  const uint32_t g_lo = row_lo / TROWS + t_lo;                // logical index of my first tile in the whole corpus
  auto tile_row0 = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa) * TROWS; };
  auto issue_piece = [&](uint32_t row0, uint32_t buf, int i) {
    glds16(src_off[i], gbase + static_cast<uint64_t>(row0) * ROW_BYTES, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
  auto issue_scales = [&](uint32_t row0, uint32_t buf) {         // only the lanes that carry distinct bytes take part
    if (lane < 8 * MB) glds16(sc_off, reinterpret_cast<const char*>(scales + row0), lds_base + buf * STAGE_BYTES + DATA_BYTES + wave * 1024);
  };
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(tile_row0(st), st, i);
    issue_scales(tile_row0(st), st);
  }

  uint32_t wcnt = 0;
  uint32_t n_stage1 = 0, n_stage2 = 0;             // diagnostics (uniform): tiles of this wave that went past stage 0 / 1
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;

  // Deferred single value (the common stage-2 case: exactly ONE of the wave's values of a tile passes stage 1).  Instead
  // of multiplying the whole lo plane of its 32-query block right away -- 24 MFMAs behind an L2 round trip for the
  // fragments, with the other three waves waiting at the next barrier -- the wave copies that value's corpus row out of
  // the LDS stage into 4 registers, asks for the query's lo-plane row (768 bytes) by a direct-to-LDS load into its own
  // 1-KB scratch slot and carries on; one tile later the bytes are there and the exact lo-plane dot product of that one
  // (row, query) pair is 4 x v_dot4_i32_i8 per lane and a wave reduction.  Same integer, same float expression, same
  // hit as the block path below.
  constexpr bool DEFER = (MB == 2) && (NSTAGE * STAGE_BYTES + 4096 <= 160 * 1024) && (DIM % 16 == 0) && (DIM / 16 <= 64);
  char* scratch = smem + NSTAGE * STAGE_BYTES + wave * 1024;
  const uint32_t lds_scratch = lds_base + NSTAGE * STAGE_BYTES + wave * 1024;
  bool pend = false;                               // uniform
  uint4 pend_x = make_uint4(0, 0, 0, 0);           // lanes < DIM/16: 16 bytes of the pending pair's corpus row
  int pend_H = 0;
  float pend_scale = 0.f, pend_thr = 0.f, pend_inv = 0.f;
  uint32_t pend_qid = 0, pend_row = 0;
  auto consume_pending = [&](int still_in_flight) {
    // the scratch load is older than the newest `still_in_flight` loads of this wave
    if (still_in_flight == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");
    int part = 0;
    if (lane < DIM / 16) {
      const uint4 lo = *reinterpret_cast<const uint4*>(scratch + lane * 16);
      part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x.x), static_cast<int>(lo.x), part, false);
      part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x.y), static_cast<int>(lo.y), part, false);
      part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x.z), static_cast<int>(lo.z), part, false);
      part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x.w), static_cast<int>(lo.w), part, false);
    }
    part = wave_sum_i32(part);                                             // exact: integer sum
    const float fv = static_cast<float>(shl_i32(pend_H, sa.lo_bits) + part) * pend_scale;
    if (fv >= pend_thr) {
      if (lane == 0 && wcnt < FILTER_LOGCAP) mylog[wcnt] = Hit{fv * pend_inv, pend_row, pend_qid, 0u};
      ++wcnt;
    }
    pend = false;
  };

  uint32_t sync_strikes = 0;
  uint64_t stamp_c = 0, stamp_r = 0;
  if constexpr (STAMP) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  for (uint32_t t = 0; t < NT; ++t) {
    if constexpr (SYNC) {
      if (wave == 0 && (t & sync_mask) == 0) sibling_rendezvous(prog + static_cast<uint64_t>(stream) * 8, qt, t, sync_lead, sync_strikes, lane);
    }
    // my pieces of tile t have landed once all but the newest NSTAGE-2 tiles' loads are complete
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (PPW + 1)) : "memory");
    if constexpr (VAR < 3) __builtin_amdgcn_s_barrier();
    const uint32_t next_row0 = tile_row0(t + NSTAGE - 1), next_buf = (t + NSTAGE - 1) % NSTAGE;
    const char* stage = smem + (t % NSTAGE) * STAGE_BYTES;
    if (!wave_has_queries) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(next_row0, next_buf, i);
      issue_scales(next_row0, next_buf);
      continue;
    }
    auto read_a = [&](int u) -> float4_t {        // u = MB*s + mb: k-step s of row block mb
      const int s = u / MB, mb = u % MB;
      return *reinterpret_cast<const float4_t*>(stage + a_addr_i8<ROW_BYTES>(a_base, s) + mb * FILTER_ROWS * ROW_BYTES);
    };
    // this tile's 16 row scales per row block for my lanes (rows 32*mb + (r&3) + 8*(r>>2) + 4*hsel)
    const float* sc_lds = reinterpret_cast<const float*>(stage + DATA_BYTES + wave * 1024);
    float scv[MB][16];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(sc_lds + 32 * mb + 8 * j + 4 * hsel);
        scv[mb][4 * j] = v.x; scv[mb][4 * j + 1] = v.y; scv[mb][4 * j + 2] = v.z; scv[mb][4 * j + 3] = v.w;
      }
    constexpr int NREAD = MB * KSTEPS;
    float4_t ar[RING];
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) ar[u] = read_a(u);
    intx16 acc[MB][NB];
#pragma unroll
    for (int u = 0; u < NREAD; ++u) {
      if (u + RING - 1 < NREAD) ar[(u + RING - 1) % RING] = read_a(u + RING - 1);
      const float4_t a = ar[u % RING];
      const int s = u / MB, mb = u % MB;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (s == 0) NVDB_MFMA_I8_ZERO(acc[mb][nb], a, bq[nb * KSTEPS]);
        else NVDB_MFMA_I8_ACC(acc[mb][nb], a, bq[nb * KSTEPS + s]);
      }
      if (u % (NREAD / PPW) == NREAD / PPW - 1) issue_piece(next_row0, next_buf, u / (NREAD / PPW));
      if (u == 1) issue_scales(next_row0, next_buf);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(acc[mb][nb]));

    if constexpr (VAR >= 2) continue;
    // ---- stage 1: can any of my NB x 16 hi-plane values H_r * scale_r reach its first-stage threshold? ----------
    // (exact per value: cvt + mul, then a max tree per block and one compare; a cheaper bound such as
    //  max(H) * max(scale) lets a quarter of the tiles through, and a tile costs what its slowest wave costs)
    float d1[MB][NB];
    float dall = -__builtin_huge_valf();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float fh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) fh[r] = static_cast<float>(acc[mb][nb][r]) * scv[mb][r];
        float m = vmax3(fh[0], fh[1], fh[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m = vmax3(m, fh[r], fh[r + 1]);
        d1[mb][nb] = vmax3(m, fh[15], fh[15]) - t1q[nb];                 // >= 0 iff some value reaches the threshold
        dall = vmax3(dall, d1[mb][nb], d1[mb][nb]);
      }
    if constexpr (DEFER) { if (pend) consume_pending(1); }             // issued one tile ago: PPW + 1 younger loads in flight
    if (!__builtin_amdgcn_ballot_w64(dall >= 0.f)) continue;
    ++n_stage1;
    if constexpr (VAR == 1) continue;
    const uint32_t row0 = tile_row0(t);
    if constexpr (DEFER) {
      // rare path: which values pass?  (the same comparison as the max tree, value by value)
      uint32_t nflag = 0;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if (!__builtin_amdgcn_ballot_w64(d1[mb][nb] >= 0.f)) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(static_cast<float>(acc[mb][nb][r]) * scv[mb][r] >= t1q[nb]);
            if (m) {
              nflag += static_cast<uint32_t>(__builtin_popcountll(m));
              const int L = __builtin_ctzll(m);
              pend_H = __builtin_amdgcn_readlane(acc[mb][nb][r], L);
              pend_scale = readlane_f(scv[mb][r], L);
              pend_thr = readlane_f(thr_s[nb], L);
              pend_inv = readlane_f(inv_s[nb], L);
              pend_qid = readlane_u(qid[nb], L);
              pend_row = 32u * mb + (r & 3) + 8u * (r >> 2) + 4u * (static_cast<uint32_t>(L) >> 5);      // row inside the tile
            }
          }
        }
      if (nflag == 1) {
        // its corpus row: chunk l of row i sits at position l ^ (i & 15) of the stage's row image
        if (lane < DIM / 16) pend_x = *reinterpret_cast<const uint4*>(stage + pend_row * ROW_BYTES + (swz_chunk<ROW_BYTES>(static_cast<uint32_t>(lane), pend_row) << 4));
        if (lane < DIM / 16) glds16(static_cast<uint32_t>(lane) * 16u, qlo + static_cast<uint64_t>(pend_qid) * DIM, lds_scratch);
        pend_row += row0;
        pend = true;
        continue;
      }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (!__builtin_amdgcn_ballot_w64(d1[mb][nb] >= 0.f)) continue;
        ++n_stage2;
        // ---- stage 2: the lo plane of this (row block, query block), fragments from global memory in two halves ----
        const signed char* ql = qlo + static_cast<uint64_t>(qbase + nb * 32 + r31) * DIM + 16 * hsel;
        intx16 lo;
        constexpr int HALF = KSTEPS / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4_t bl[HALF];
#pragma unroll
          for (int s = 0; s < HALF; ++s) bl[s] = *reinterpret_cast<const float4_t*>(ql + 32 * (h * HALF + s));   // all in flight at once
#pragma unroll
          for (int s = 0; s < RING - 1; ++s) ar[s] = read_a(MB * (h * HALF + s) + mb);
#pragma unroll
          for (int s = 0; s < HALF; ++s) {
            if (s + RING - 1 < HALF) ar[(s + RING - 1) % RING] = read_a(MB * (h * HALF + s + RING - 1) + mb);
            const float4_t a = ar[s % RING];
            if (h == 0 && s == 0) NVDB_MFMA_I8_ZERO_V(lo, a, bl[s]); else NVDB_MFMA_I8_ACC_V(lo, a, bl[s]);
          }
        }
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(lo));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float fv = static_cast<float>(shl_i32(acc[mb][nb][r], sa.lo_bits) + lo[r]) * scv[mb][r];
          const bool hit = fv >= thr_s[nb];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m) {
            const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
            if (hit && idx < FILTER_LOGCAP) mylog[idx] = Hit{fv * inv_s[nb], row0 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * hsel, qid[nb], 0u};
            wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
          }
        }
      }
  }
  if constexpr (DEFER) { if (pend) consume_pending(0); }
  if constexpr (STAMP) {
    const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c, dr = __builtin_amdgcn_s_memrealtime() - stamp_r;
    if (wave == 0 && lane == 0) {
      uint64_t* out = reinterpret_cast<uint64_t*>(prog + static_cast<uint64_t>(gridDim.x) * 8) + static_cast<uint64_t>(blockIdx.x) * 2;
      out[0] = dc; out[1] = dr;
    }
  }
  if constexpr (SYNC) { if (wave == 0 && lane == 0) __hip_atomic_store(prog + static_cast<uint64_t>(stream) * 8 + qt, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (stage_counts && lane == 0 && (n_stage1 | n_stage2)) { atomicAdd(stage_counts, n_stage1); atomicAdd(stage_counts + 1, n_stage2); }
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  scatter_own_log(mylog, wcnt, sa, lane);
}

// int8 two-stage kernel, DEFER = false: the tile loop logs every value that passes the FIRST stage (hi plane only) with its row
// scale and H; here, after the stream, the wave finishes them: L = <row, lo plane of the query> by v_dot4 (4 lanes per entry,
// 16 entries per step, rows re-read from L2 / HBM), fv = ((H << lo_bits) + L) * scale, and files those with fv >= T under their
// queries -- the same survivors with the same filter scores as the in-loop second stage, without stalling the workgroup's
// barrier for every flagged value.
template <int DIM>
__device__ __forceinline__ void verify_and_scatter_i8(const Hit* mylog, uint32_t wcnt, const ScatterArgs& a, int lane, const signed char* __restrict__ rows,
                                                      const signed char* __restrict__ qlo, const float* __restrict__ thr, const float* __restrict__ qscale,
                                                      const float* __restrict__ qinv) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own log stores (and the speculative stages) have completed
  uint32_t n = wcnt;
  if (n > FILTER_LOGCAP) { if (lane == 0) *a.log_overflow = 1u; n = FILTER_LOGCAP; }
  constexpr int CPL = DIM / 64;                      // 16-byte chunks per lane (4 lanes per entry)
  constexpr int NE = 2;                              // entries per lane group and step: the loads of both are in flight together
  const int part4 = lane & 3;
  for (uint32_t base = 0; base < n; base += 16 * NE) {
    uint32_t sbits[NE], row[NE], qid[NE];
    int H[NE], part[NE];
    bool valid[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const uint32_t e = base + 16u * j + (static_cast<uint32_t>(lane) >> 2);
      valid[j] = e < n;
      const uint32_t* w = reinterpret_cast<const uint32_t*>(mylog + (valid[j] ? e : 0u));
      // L2-served loads: the entries were written by this wave in this launch
      sbits[j] = __hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      row[j] = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      qid[j] = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      H[j] = static_cast<int>(__hip_atomic_load(w + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    uint4 x[NE][CPL], y[NE][CPL];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      valid[j] = valid[j] && row[j] < a.n_rows;
      const uint4* xr = reinterpret_cast<const uint4*>(rows + static_cast<uint64_t>(valid[j] ? row[j] : 0u) * DIM);
      const uint4* qr = reinterpret_cast<const uint4*>(qlo + static_cast<uint64_t>(valid[j] ? qid[j] : 0u) * DIM);
#pragma unroll
      for (int c = 0; c < CPL; ++c) { x[j][c] = xr[part4 + 4 * c]; y[j][c] = qr[part4 + 4 * c]; }   // (an invalid slot reads row 0 / query 0: in bounds, unused)
    }
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      part[j] = 0;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        part[j] = __builtin_amdgcn_sdot4(static_cast<int>(x[j][c].x), static_cast<int>(y[j][c].x), part[j], false);
        part[j] = __builtin_amdgcn_sdot4(static_cast<int>(x[j][c].y), static_cast<int>(y[j][c].y), part[j], false);
        part[j] = __builtin_amdgcn_sdot4(static_cast<int>(x[j][c].z), static_cast<int>(y[j][c].z), part[j], false);
        part[j] = __builtin_amdgcn_sdot4(static_cast<int>(x[j][c].w), static_cast<int>(y[j][c].w), part[j], false);
      }
      part[j] += __builtin_amdgcn_update_dpp(0, part[j], 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
      part[j] += __builtin_amdgcn_update_dpp(0, part[j], 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]: the entry's four lanes hold L
      if (valid[j] && part4 == 0) {
        const float fv = static_cast<float>(shl_i32(H[j], a.lo_bits) + part[j]) * __builtin_bit_cast(float, sbits[j]);
        if (fv >= thr[qid[j]] * qscale[qid[j]]) {
          const uint32_t slot = atomicAdd(&a.cnt[qid[j]], 1u);
          if (slot < a.cap) a.cand[static_cast<uint64_t>(qid[j]) * a.cap + slot] = Cand{fv * qinv[qid[j]], row[j]};
          else a.overflow[qid[j]] = 1u;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// int8 two-stage build, software-pipelined (batches > 128: NB = 2 query blocks per wave, 64-row tiles).
//
// filter_i8w_kernel spends ~950 of its ~5200 cycles per tile in the stage-1 test AFTER the tile's MFMAs
// (profiles/r02_i8_clock_ablation.txt: 64 cvt + 64 mul + a max tree per wave and tile, nothing else running on the SIMD).
// Here the two 32-row blocks of a tile are multiplied one after the other and each block's test runs in the shadow of
// the OTHER block's MFMAs -- an MFMA holds the SIMD's issue port for 8 of its 32 cycles, the rest takes VALU work:
//
//   tile t:  barrier A | MFMAs(block 0, tile t)  ||  test(block 1, tile t-1), its rare path | barrier B |
//                        MFMAs(block 1, tile t)  ||  loads of tile t+2, test(block 0, tile t), its rare path
//
// No second accumulator set: block 0's accumulators are tested while block 1's are written and vice versa.
//
// Two builds of the loop (template DEFER; every build logs the same survivors with the same filter scores,
// test_int8_two_stage_kernel_matches_two_plane_kernel):
//   DEFER = false, the default: a value that passes the first stage is only LOGGED in the loop (row scale, row, query, H) and
//     finished exactly after the stream by verify_and_scatter_i8.  Nothing in the loop reads the previous tile's LDS stage
//     (block 1's row scales are kept in registers), so there is ONE barrier per tile (A), the tile's LDS-DMA issues are spread
//     over both halves (the first three in the bubble while the fragment ring fills), the ring runs through both halves, and the
//     accumulators start at the bits of 2^23 so that the test is one v_fma per value (BIASED, see below).
//   DEFER = true (option i8_defer, and by itself for a corpus with a negative / NaN row scale): the in-loop second stage of
//     filter_i8w_kernel -- one flagged value -> deferred exact dot product (v_dot4), several -> lo-plane MFMAs of the block.
//     Barrier B is what it costs: block 1 of tile t-1 is tested during tile t, and its rare path still reads that tile's LDS
//     stage, so the loads that overwrite it (tile t+2, same buffer) are issued only after every wave has passed B.
// ------------------------------------------------------------------------------------------------
// VAR (diagnostic builds, STAMP only; wrong results): 1 = no test and no rare path (the pipelined structure alone), 2 = test but no rare path,
// 3 = rare path without consuming the deferred values, 4 = rare path entered and left at once; 5 = the production loop (right results) reporting
// the cycles wave 0 spent inside rare_path (high word) and consume_slots (low word) instead of the loop's cycle count
// WPB = 8: the same 256 queries per workgroup on 8 waves of 32 (NB = 1), two waves per SIMD with 256 registers each: a wave's
// LDS-DMA issue, ring priming, barrier waits and rare path run beside its SIMD partner's MFMAs.
// DEFER = false (the default build): a value that passes the first stage is only logged in the loop and finished after the stream
// (verify_and_scatter_i8); DEFER = true: the in-loop second stage (deferred v_dot4 slots, lo-plane MFMAs for dense blocks).
template <int DIM, bool SYNC = false, bool STAMP = false, int RING = 6, int VAR = 0, int WPB = 4, bool DEFER = true>
__global__ __launch_bounds__(64 * WPB, 1) void filter_i8p_kernel(
    const signed char* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const signed char* __restrict__ qhi, const signed char* __restrict__ qlo, uint32_t nq, uint32_t QT,
    const float* __restrict__ thr, const float* __restrict__ qscale, const float* __restrict__ qinv,
    const float* __restrict__ qdelta, Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t* __restrict__ prog,
    uint32_t sync_mask, uint32_t sync_lead, uint32_t* __restrict__ stage_counts) {
  static_assert(WPB == 4 || WPB == 8, "4 waves x 64 queries or 8 waves x 32 queries");
  constexpr int NB = 8 / WPB, MB = 2;
  constexpr int NV = 16 * NB;                                // values a lane tests per row block
  constexpr int KSTEPS = DIM / 32;
  constexpr int ROW_BYTES = DIM;
  constexpr int TROWS = FILTER_ROWS * MB;
  constexpr int NSTAGE = 3;
  constexpr int DATA_BYTES = TROWS * ROW_BYTES;
  constexpr int SC_COPIES = WPB == 4 ? 4 : 1;                // 8 waves: every wave loads the same 256 bytes to the same place (the LDS is full)
  constexpr int STAGE_BYTES = DATA_BYTES + SC_COPIES * 256;  // the tile + 256-byte copies of its 64 row scales
  constexpr int NSLOT_DEFER = 16 / WPB, SCRATCH_BYTES = DEFER ? NSLOT_DEFER * DIM : 0;   // per wave: lo-plane rows of up to 4 (2) deferred values
  constexpr int PIECES = DATA_BYTES / 1024;
  constexpr int PPW = PIECES / WPB;
  constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;
  static_assert(DIM % 128 == 0 && DIM <= 768, "row stride multiple of 128 bytes (swz_chunk); int32 range of 128*H + L");
  static_assert(PIECES % WPB == 0 && KSTEPS % PPW == 0 && KSTEPS % 2 == 0 && KSTEPS >= 8, "shape");
  static_assert(NSTAGE * STAGE_BYTES + WPB * SCRATCH_BYTES <= 160 * 1024 && PPW + 1 < 64 && DIM % 16 == 0 && DIM / 16 <= 64, "LDS / vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, hsel = lane >> 5;
  const uint32_t wave_gid = blockIdx.x * WPB + wave;

  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) { const uint32_t xcd = b & 7u, i = b >> 3; qt = i % QT; stream = (i / QT) * 8u + xcd; }
  else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / TROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // stationary operand: hi plane of this wave's 2 blocks of 32 queries, all of K, in AGPRs
  const uint32_t qbase = qt * 256u + wave * (32u * NB);
  float4_t bq[NB * KSTEPS];
#pragma unroll
  for (int f = 0; f < NB * KSTEPS; ++f) {
    const int nb = f / KSTEPS, s = f % KSTEPS;
    bq[f] = *reinterpret_cast<const float4_t*>(qhi + static_cast<uint64_t>(qbase + nb * 32 + r31) * DIM + 32 * s + 16 * hsel);
  }
#pragma unroll
  for (int f = 0; f < NB * KSTEPS; ++f) asm volatile("" ::"a"(bq[f]));
  // !DEFER: the accumulators start at I8_ACC_BIAS, the bits of 2^23 as a float.  |H| < 2^23 (I8_HI_L1_MAX), so the int32 sum read
  // AS A FLOAT is 2^23 + H for H >= 0 and 2^23 + H / 2 for H < 0: the first-stage test needs no int -> float conversion, one v_fma
  // per value does it (32 values: 32 fma + 16 max3 instead of 32 cvt + 32 mul + 16 max3; the packed v_pk_add / v_pk_mul pair was
  // slower than the conversions, profiles/r02g_i8_biased_ab.txt).  Row scales are >= 0 here (a corpus with a negative one takes the DEFER build), so a
  // negative H only yields a negative value -- halved, i.e. never smaller than the true one: no flag is lost.
  constexpr bool BIASED = !DEFER;
  intx16 bias0;
#pragma unroll
  for (int r = 0; r < 16; ++r) bias0[r] = I8_ACC_BIAS;
  if constexpr (BIASED) asm volatile("" : "+v"(bias0));
  auto hval = [&](int bits) -> int { return BIASED ? bits - I8_ACC_BIAS : bits; };
  uint32_t qid[NB];
  float thr_s[NB], t1q[NB], inv_s[NB];
  const float lo_unit = __builtin_bit_cast(float, (127u - sa.lo_bits) << 23);   // 2^-lo_bits: the first stage compares H alone
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    qid[nb] = qbase + nb * 32 + r31;
    const bool real = qid[nb] < nq;
    thr_s[nb] = real ? thr[qid[nb]] * qscale[qid[nb]] : __builtin_huge_valf();
    inv_s[nb] = real ? qinv[qid[nb]] : 0.f;
    const float T = thr_s[nb];
    t1q[nb] = real ? (T - (1.001f * qdelta[qid[nb]] + 2e-6f * fabsf(T) + 1e-5f)) * lo_unit : __builtin_huge_valf();   // as filter_i8w_kernel
    asm volatile("" ::"v"(thr_s[nb]), "v"(inv_s[nb]), "v"(t1q[nb]));
  }
  const bool wave_has_queries = qbase < nq;

  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CHUNKS_PER_ROW, cpos = P % CHUNKS_PER_ROW;
    src_off[i] = r * ROW_BYTES + (swz_chunk<ROW_BYTES>(cpos, r) << 4);
  }
  const uint32_t sc_off = (lane & (8 * MB - 1)) * 16;
  const uint32_t a_base = a_base_i8<ROW_BYTES>(static_cast<uint32_t>(r31), static_cast<uint32_t>(hsel));

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  const uint32_t g_lo = row_lo / TROWS + t_lo;
  auto tile_row0 = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa) * TROWS; };
  auto issue_piece = [&](uint32_t row0, uint32_t buf, int i) {
    glds16(src_off[i], gbase + static_cast<uint64_t>(row0) * ROW_BYTES, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
  auto issue_scales = [&](uint32_t row0, uint32_t buf) {
    if (lane < 8 * MB) glds16(sc_off, reinterpret_cast<const char*>(scales + row0), lds_base + buf * STAGE_BYTES + DATA_BYTES + (wave % SC_COPIES) * 256);
  };
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(tile_row0(st), st, i);
    issue_scales(tile_row0(st), st);
  }

  uint32_t wcnt = 0;
  uint32_t n_stage1 = 0, n_stage2 = 0;
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;

  // Deferred values (see filter_i8w_kernel): a value that passes stage 1 waits in one of 4 slots -- its corpus row in 4
  // registers, its query's lo-plane row on the way into this wave's LDS scratch -- and is finished by an exact v_dot4
  // product at the end of a tile's second half, when 13 younger loads stand behind it.  Only a block with more flagged values
  // than free slots takes the lo-plane MFMAs (dense cases: first chunks, near-duplicate corpora).
  char* scratch = smem + NSTAGE * STAGE_BYTES + wave * SCRATCH_BYTES;
  const uint32_t lds_scratch = lds_base + NSTAGE * STAGE_BYTES + wave * SCRATCH_BYTES;
  uint32_t pend_mask = 0;                          // uniform: occupied slots
  uint32_t pend_old = 0;                           // ... of them, deferred during the PREVIOUS tile: those are consumed at the end of this one
  uint32_t young_loads = 0;                        // deferred loads issued during this tile's first half (younger than every `pend_old` load)
  uint4 pend_x[NSLOT_DEFER];
  int pend_H[NSLOT_DEFER];
  float pend_scale[NSLOT_DEFER], pend_thr[NSLOT_DEFER], pend_inv[NSLOT_DEFER];
  uint32_t pend_qid[NSLOT_DEFER], pend_row[NSLOT_DEFER];
#pragma unroll
  for (int i = 0; i < NSLOT_DEFER; ++i) { pend_x[i] = make_uint4(0, 0, 0, 0); pend_H[i] = 0; pend_scale[i] = pend_thr[i] = pend_inv[i] = 0.f; pend_qid[i] = pend_row[i] = 0; }
  // finish the slots in `which`: their lo-plane rows have landed once all but the `younger` newest loads are complete
  auto consume_slots = [&](uint32_t which, uint32_t younger) {
    if constexpr (VAR == 3) { pend_mask &= ~which; return; }
    switch (younger) {                             // s_waitcnt takes an immediate
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case PPW + 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory"); break;
      case PPW + 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2) : "memory"); break;
      case PPW + 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 3) : "memory"); break;
      case PPW + 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 4) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 5) : "memory"); break;
    }
#pragma unroll
    for (int i = 0; i < NSLOT_DEFER; ++i) {
      if (!(which & (1u << i))) continue;
      int part = 0;
      if (lane < DIM / 16) {
        const uint4 lo = *reinterpret_cast<const uint4*>(scratch + i * DIM + lane * 16);
        part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x[i].x), static_cast<int>(lo.x), part, false);
        part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x[i].y), static_cast<int>(lo.y), part, false);
        part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x[i].z), static_cast<int>(lo.z), part, false);
        part = __builtin_amdgcn_sdot4(static_cast<int>(pend_x[i].w), static_cast<int>(lo.w), part, false);
      }
      part = wave_sum_i32(part);                                           // exact: integer sum
      const float fv = static_cast<float>(shl_i32(pend_H[i], sa.lo_bits) + part) * pend_scale[i];
      if (fv >= pend_thr[i]) {                     // (a slot given up by the block path carries thr = +inf)
        if (lane == 0 && wcnt < FILTER_LOGCAP) mylog[wcnt] = Hit{fv * pend_inv[i], pend_row[i], pend_qid[i], 0u};
        ++wcnt;
      }
    }
    pend_mask &= ~which;
  };

  intx16 acc0[NB], acc1[NB];                       // row block 0 / 1 of the tile in flight
  // row scales of the block under test.  DEFER: one set, re-read from LDS at the top of each half.  Otherwise two: [0] block 0 and
  // [1] block 1 of the current tile, both read in the tile's second half -- block 1 is tested during the NEXT tile's first half,
  // which then touches nothing of the previous tile's LDS stage: its buffer can be refilled right after barrier A, and barrier B goes.
  float scv2[DEFER ? 1 : 2][16];
  float scc2[DEFER ? 1 : 2][16];                   // biased build: -2^23 * scale, the constant of the test's v_fma
  float (&scv)[16] = scv2[0];
  float mx[NB][4];                                 // running max of H * scale per query block and group of 4 accumulator registers
  float4_t ar[RING];

  // one 32-row block's MFMAs (KSTEPS k-steps x NB query blocks) with `slot(s)` called after every k-step
  auto read_a = [&](const char* stage, int s, int mb) -> float4_t {
    return *reinterpret_cast<const float4_t*>(stage + a_addr_i8<ROW_BYTES>(a_base, s) + mb * FILTER_ROWS * ROW_BYTES);
  };
  auto load_scales = [&](const char* stage, int mb, float (&dst)[16]) {
    const float* sc_lds = reinterpret_cast<const float*>(stage + DATA_BYTES + (wave % SC_COPIES) * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(sc_lds + 32 * mb + 8 * j + 4 * hsel);
      dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
    }
    if constexpr (!DEFER) {
      float (&dc)[16] = (&dst == &scv2[0]) ? scc2[0] : scc2[DEFER ? 0 : 1];
#pragma unroll
      for (int i = 0; i < 16; ++i) dc[i] = dst[i] * -8388608.f;
    }
  };
  // !DEFER: the same in the shadow of the MFMAs -- slot 0 reads the 16 scales of block mb of `stage` into set `set`, the slots from the 6th on
  // multiply one of them each by -2^23 (the test's v_fma constant).  Block 0's set is filled during the tile's first half (used in
  // the second), block 1's during the second half (used in the next tile's first half).
  auto scale_step = [&](const char* stage, int mb, int set, int w) {
    if constexpr (VAR == 1) return;
    if (w == 0) {
      const float* sc_lds = reinterpret_cast<const float*>(stage + DATA_BYTES + (wave % SC_COPIES) * 256);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(sc_lds + 32 * mb + 8 * j + 4 * hsel);
        scv2[set][4 * j] = v.x; scv2[set][4 * j + 1] = v.y; scv2[set][4 * j + 2] = v.z; scv2[set][4 * j + 3] = v.w;
      }
    }
    constexpr int NSL = NB * KSTEPS, SW0 = NSL >= 32 ? 6 : 4, SPS = (16 + NSL - SW0 - 1) / (NSL - SW0);   // constants per slot: 1 at d >= 512
#pragma unroll
    for (int i = (w - SW0) * SPS; i < (w - SW0 + 1) * SPS; ++i)
      if (i >= 0 && i < 16) scc2[set][i] = scv2[set][i] * -8388608.f;
  };
  // Stage-1 test of a block (32 values: query block v / 16, accumulator register v % 16), software-pipelined over the MFMA
  // slots so that the (at most 3) VALU instructions behind one MFMA never depend on each other: slot j converts the values
  // of step j, multiplies those of step j - 1 by their row scales and folds those of step j - 2 into the running maxima.
  float tc[NV], tm[NV];
  auto test_step = [&](const intx16 (&a)[NB], const float (&sc)[16], int j, int vps) {
    if constexpr (VAR == 1) return;
    if constexpr (BIASED) {
      // one v_fma per value: (2^23 + H) * scale - 2^23 * scale = H * scale, rounded once like float(H) * scale (the constant is
      // exact: a power of two times the scale); slot j multiplies the values of step j, folds those of step j - 1
      const float (&scc)[16] = (&sc == &scv2[0]) ? scc2[0] : scc2[BIASED ? 1 : 0];
#pragma unroll
      for (int v = j * vps; v < (j + 1) * vps; ++v)
        if (v >= 0 && v < NV) {
          const int b0 = a[v / 16][v % 16];                                  // (a copy: bit_cast of a vector ELEMENT reads element 0)
          tm[v] = __builtin_fmaf(__builtin_bit_cast(float, b0), sc[v % 16], scc[v % 16]);
        }
#pragma unroll
      for (int v = (j - 1) * vps; v < j * vps; ++v)
        if (v >= 0 && v < NV && (v & 1)) mx[v / 16][(v % 16) / 4] = vmax3(mx[v / 16][(v % 16) / 4], tm[v - 1], tm[v]);
      return;
    }
#pragma unroll
    for (int v = j * vps; v < (j + 1) * vps; ++v) if (v >= 0 && v < NV) tc[v] = static_cast<float>(a[v / 16][v % 16]);
#pragma unroll
    for (int v = (j - 1) * vps; v < j * vps; ++v) if (v >= 0 && v < NV) tm[v] = tc[v] * sc[v % 16];
#pragma unroll
    for (int v = (j - 2) * vps; v < (j - 1) * vps; ++v) if (v >= 0 && v < NV && (v & 1)) mx[v / 16][(v % 16) / 4] = vmax3(mx[v / 16][(v % 16) / 4], tm[v - 1], tm[v]);
  };
  auto reset_max = [&]() {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) mx[nb][g] = -__builtin_huge_valf();
  };
  float flagv = -1.f;                              // >= 0 iff some value of the tested block reaches its first-stage threshold
  auto combine_flags = [&]() {                     // behind the half's last MFMA
    if constexpr (VAR == 1) return;
    float m[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) m[nb] = vmax3(vmax3(mx[nb][0], mx[nb][1], mx[nb][2]), mx[nb][3], mx[nb][3]) - t1q[nb];
    flagv = NB == 2 ? vmax3(m[0], m[NB - 1], m[NB - 1]) : m[0];
  };
  auto any_flag = [&]() -> bool { return __builtin_amdgcn_ballot_w64(flagv >= 0.f) != 0; };
  // rare path of one tested block: `a` its accumulators, `stage` / `row0` its tile, mb its row block.  Returns the number of
  // deferred loads it issued.  Per query block: the flagged group(s) of 4 registers -> the flagged values -> one slot each;
  // a block with more flagged values than free slots gives its slots back and takes the lo-plane MFMAs instead.
  auto rare_path = [&](const intx16 (&a)[NB], const char* stage, uint32_t row0, int mb) -> uint32_t {
    ++n_stage1;
    uint32_t issued = 0;
    if constexpr (VAR == 4) return issued;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const uint32_t before = pend_mask;
      bool overflow = false;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (!__builtin_amdgcn_ballot_w64(mx[nb][g] >= t1q[nb])) continue;
        // the group's 4 values, branch-free: per lane a 4-bit mask of the values that pass and (H, scale) of the last one
        uint32_t bits = 0;
        int hsel_v = 0;
        float ssel_v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const bool f = static_cast<float>(a[nb][r]) * scv[r] >= t1q[nb];
          bits |= f ? (1u << j) : 0u;
          hsel_v = f ? a[nb][r] : hsel_v;
          ssel_v = f ? scv[r] : ssel_v;
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(bits != 0);
        while (m) {                                // flagged lanes (queries) of this group: almost always one
          const int L = __builtin_ctzll(m);
          m &= m - 1;
          const uint32_t lb = readlane_u(bits, L);
          if ((lb & (lb - 1)) != 0 || pend_mask == (1u << NSLOT_DEFER) - 1) { overflow = true; continue; }   // two values of one query in one group, or no slot left
          const uint32_t r = 4u * g + (31u - static_cast<uint32_t>(__builtin_clz(lb)));
          const uint32_t rowi = 32u * mb + (r & 3) + 8u * (r >> 2) + 4u * (static_cast<uint32_t>(L) >> 5);
          bool placed = false;
#pragma unroll
          for (int i = 0; i < NSLOT_DEFER; ++i) {
            if (placed || (pend_mask & (1u << i))) continue;
            placed = true;
            pend_H[i] = __builtin_amdgcn_readlane(hsel_v, L);
            pend_scale[i] = readlane_f(ssel_v, L);
            pend_thr[i] = readlane_f(thr_s[nb], L);
            pend_inv[i] = readlane_f(inv_s[nb], L);
            pend_qid[i] = readlane_u(qid[nb], L);
            pend_row[i] = rowi + row0;
            // its corpus row: chunk l of row i sits at position l ^ (i & 15) of the stage's row image
            if (lane < DIM / 16) pend_x[i] = *reinterpret_cast<const uint4*>(stage + rowi * ROW_BYTES + (swz_chunk<ROW_BYTES>(static_cast<uint32_t>(lane), rowi) << 4));
            if (lane < DIM / 16) glds16(static_cast<uint32_t>(lane) * 16u, qlo + static_cast<uint64_t>(pend_qid[i]) * DIM, lds_scratch + i * DIM);
            pend_mask |= 1u << i;
            ++issued;
          }
        }
      }
      if (!overflow) continue;
      // ---- too many for the slots: this block's deferrals are void (their slots stay busy until consumed, with a threshold
      //      nothing reaches) and the lo plane of this (row block, query block) goes to the matrix cores ----
#pragma unroll
      for (int i = 0; i < NSLOT_DEFER; ++i) if ((pend_mask & ~before) & (1u << i)) pend_thr[i] = __builtin_huge_valf();
      ++n_stage2;
      const signed char* ql = qlo + static_cast<uint64_t>(qbase + nb * 32 + r31) * DIM + 16 * hsel;
      intx16 lo;
      constexpr int NCH = WPB == 8 ? 4 : 2, HALF = KSTEPS / NCH;      // query lo-plane fragments held at a time (8 waves: 256 registers per wave)
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        float4_t bl[HALF];
#pragma unroll
        for (int s = 0; s < HALF; ++s) bl[s] = *reinterpret_cast<const float4_t*>(ql + 32 * (h * HALF + s));
#pragma unroll
        for (int s = 0; s < HALF; ++s) {
          const float4_t av = read_a(stage, h * HALF + s, mb);
          if (h == 0 && s == 0) NVDB_MFMA_I8_ZERO_V(lo, av, bl[s]); else NVDB_MFMA_I8_ACC_V(lo, av, bl[s]);
        }
      }
      asm volatile("s_nop 15\n\ts_nop 15" : "+v"(lo));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float fv = static_cast<float>(shl_i32(a[nb][r], sa.lo_bits) + lo[r]) * scv[r];
        const bool hit = fv >= thr_s[nb];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (m) {
          const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
          if (hit && idx < FILTER_LOGCAP) mylog[idx] = Hit{fv * inv_s[nb], row0 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * hsel, qid[nb], 0u};
          wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
        }
      }
    }
    return issued;
  };

  // DEFER = false: log the values of a tested block that pass the first stage -- row scale, row, query, H -- for the exact
  // finish after the stream.  Per flagged group of 4 accumulator registers one ballot per value; lanes compact into the log.
  auto rare_log = [&](const intx16 (&a)[NB], const float (&sc)[16], uint32_t row0, int mb) {
    ++n_stage1;
    if constexpr (VAR == 4) return;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (!__builtin_amdgcn_ballot_w64(mx[nb][g] >= t1q[nb])) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const bool hit = static_cast<float>(hval(a[nb][r])) * sc[r] >= t1q[nb];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m) {
            const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
            if (hit && idx < FILTER_LOGCAP)
              mylog[idx] = Hit{sc[r], row0 + 32u * mb + (r & 3) + 8 * (r >> 2) + 4 * hsel, qid[nb], static_cast<uint32_t>(hval(a[nb][r]))};
            wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
          }
        }
      }
    }
  };

  uint32_t sync_strikes = 0;
  uint64_t stamp_c = 0, stamp_r = 0;
  [[maybe_unused]] uint64_t rare_acc = 0, cons_acc = 0;     // VAR 5: cycles this wave spent inside rare_path / consume_slots
  if constexpr (STAMP) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  for (uint32_t t = 0; t < NT; ++t) {
    if constexpr (SYNC) {
      if (wave == 0 && (t & sync_mask) == 0) sibling_rendezvous(prog + static_cast<uint64_t>(stream) * 8, qt, t, sync_lead, sync_strikes, lane);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");         // tile t has landed (tile t+1 may still be in flight)
    __builtin_amdgcn_s_barrier();                                          // A
    const uint32_t next_row0 = tile_row0(t + 2), next_buf = (t + 2) % NSTAGE;
    const char* stage = smem + (t % NSTAGE) * STAGE_BYTES;
    const char* prev_stage = smem + ((t + NSTAGE - 1) % NSTAGE) * STAGE_BYTES;
    if (!wave_has_queries) {
      if constexpr (DEFER) __builtin_amdgcn_s_barrier();                   // B
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(next_row0, next_buf, i);
      issue_scales(next_row0, next_buf);
      continue;
    }
    // Behind every MFMA at most one value of the other block's test (slots W0 .. W0 + 31 of the half's 2 KSTEPS MFMAs:
    // <= 3 VALU instructions inside the MFMA's 24 free issue cycles).  The A-fragment ring is primed per half: keeping it
    // alive across the rare path between the halves costs more registers than the file has.
    constexpr int NUNIT = NV;                                              // test units: values
    constexpr int NSLOT = NB * KSTEPS, W0 = NB == 1 ? (NSLOT / 6 < 4 ? NSLOT / 6 : 4) : (NSLOT < 32 ? 4 : 8), VPS = (NUNIT + NSLOT - W0 - 4) / (NSLOT - W0 - 3);   // units per slot: 1 at d = 768
    static_assert(VPS >= 1 && W0 + (NUNIT + VPS - 1) / VPS + 1 <= NSLOT - 2 && (VPS == 1 || VPS % 2 == 0 || BIASED), "the three test stages end before the half's last MFMA");
    // LDS-DMA issue of tile t+2.  DEFER: all of it in the second half, behind barrier B.  Otherwise spread over both halves (one
    // piece costs ~60 issue cycles; beside the second half's MFMAs, test and ring reads a wave has room for half of them).
    constexpr int HP = DEFER ? 0 : PPW / 2;                                // pieces issued in the first half ...
    constexpr int HP0 = DEFER ? 0 : (HP < 3 ? HP : 3);                     // ... of them in the bubble while the A-fragment ring fills after barrier A
    constexpr int STEP1 = (HP - HP0) ? KSTEPS / (HP - HP0) : 1, STEP2 = KSTEPS / (PPW - HP);
    static_assert(DEFER || (PPW % 2 == 0 && KSTEPS % (PPW / 2) == 0 && (HP == HP0 || KSTEPS % (HP - HP0) == 0)), "pieces spread evenly over the k-steps of both halves");
    // !DEFER: the ring runs through both halves of a tile (the last RING - 1 k-steps of block 0 fetch the first fragments of block 1):
    // one priming bubble per tile instead of two, and the first LDS-DMA issues of the tile sit inside it.
    constexpr bool THROUGH = !DEFER;
    // ---- first half: block 0 of tile t  ||  test of block 1 of tile t-1 (t == 0: garbage, tested and ignored) ----------
    if constexpr (DEFER) load_scales(prev_stage, 1, scv2[0]);
    const float (&sc_first)[16] = scv2[DEFER ? 0 : 1];   // (DEFER: the one set)
    reset_max();
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) ar[s] = read_a(stage, s, 0);
    if constexpr (HP0 > 0) {
#pragma unroll
      for (int i = 0; i < HP0; ++i) issue_piece(next_row0, next_buf, i);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + RING - 1 < KSTEPS) ar[(s + RING - 1) % RING] = read_a(stage, s + RING - 1, 0);
      else if constexpr (THROUGH) ar[(s + RING - 1) % RING] = read_a(stage, s + RING - 1 - KSTEPS, 1);
      const float4_t av = ar[s % RING];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (s == 0) { if constexpr (BIASED) NVDB_MFMA_I8_FROM(acc0[nb], av, bq[nb * KSTEPS], bias0); else NVDB_MFMA_I8_ZERO(acc0[nb], av, bq[nb * KSTEPS]); }
        else NVDB_MFMA_I8_ACC(acc0[nb], av, bq[nb * KSTEPS + s]);
        const int w = NB * s + nb;
        if constexpr (!DEFER) scale_step(stage, 0, 0, w);
        test_step(acc1, sc_first, w - W0, VPS);
        if (w == NSLOT - 1) combine_flags();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (HP > HP0) {
        if (s % STEP1 == STEP1 - 1) issue_piece(next_row0, next_buf, HP0 + s / STEP1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    young_loads = 0;
    if ((VAR == 0 || VAR >= 3) && t > 0 && any_flag()) {
      const uint64_t r0 = VAR == 5 ? __builtin_amdgcn_s_memtime() : 0;
      if constexpr (DEFER) young_loads = rare_path(acc1, prev_stage, tile_row0(t - 1), 1); else rare_log(acc1, sc_first, tile_row0(t - 1), 1);
      if constexpr (VAR == 5) rare_acc += __builtin_amdgcn_s_memtime() - r0;
    }
    if constexpr (VAR == 1 || VAR == 2) { asm volatile("" ::"v"(flagv)); }   // keep this half's test alive
    if constexpr (DEFER) __builtin_amdgcn_s_barrier();                     // B: nobody reads the stage of tile t-1 any more
    // ---- second half: block 1 of tile t  ||  loads of tile t+2, test of block 0 of tile t ---------------------------
    if constexpr (DEFER) load_scales(stage, 0, scv2[0]);                   // (!DEFER: block 0's set was filled during the first half)
    reset_max();
    constexpr int R0 = THROUGH ? KSTEPS % RING : 0;                        // ring slot of block 1's first fragment
    if constexpr (!THROUGH) {
#pragma unroll
      for (int s = 0; s < RING - 1; ++s) ar[s] = read_a(stage, s, 1);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + RING - 1 < KSTEPS) ar[(R0 + s + RING - 1) % RING] = read_a(stage, s + RING - 1, 1);
      const float4_t av = ar[(R0 + s) % RING];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (s == 0) { if constexpr (BIASED) NVDB_MFMA_I8_FROM(acc1[nb], av, bq[nb * KSTEPS], bias0); else NVDB_MFMA_I8_ZERO(acc1[nb], av, bq[nb * KSTEPS]); }
        else NVDB_MFMA_I8_ACC(acc1[nb], av, bq[nb * KSTEPS + s]);
        const int w = NB * s + nb;
        if constexpr (!DEFER) scale_step(stage, 1, DEFER ? 0 : 1, w);        // block 1's scales: tested in the next tile's first half
        test_step(acc0, scv2[0], w - W0, VPS);
        if (w == NSLOT - 1) combine_flags();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s % STEP2 == STEP2 - 1) issue_piece(next_row0, next_buf, HP + s / STEP2);
      if (s == 0) issue_scales(next_row0, next_buf);
      __builtin_amdgcn_sched_barrier(0);
    }
    // what was deferred during the PREVIOUS tile is at least one tile old now; behind it: this tile's deferred loads and its PPW + 1 pieces
    {
      const uint64_t r0 = VAR == 5 ? __builtin_amdgcn_s_memtime() : 0;
      if (pend_old) consume_slots(pend_old, PPW + 1 + young_loads);
      if constexpr (VAR == 5) { const uint64_t r1 = __builtin_amdgcn_s_memtime(); cons_acc += r1 - r0; }
    }
    if ((VAR == 0 || VAR >= 3) && any_flag()) {
      const uint64_t r0 = VAR == 5 ? __builtin_amdgcn_s_memtime() : 0;
      if constexpr (DEFER) rare_path(acc0, stage, tile_row0(t), 0); else rare_log(acc0, scv2[0], tile_row0(t), 0);
      if constexpr (VAR == 5) rare_acc += __builtin_amdgcn_s_memtime() - r0;
    }
    pend_old = pend_mask;                                                  // everything deferred during this tile: due at the end of the next
    if constexpr (VAR == 1 || VAR == 2) { asm volatile("" ::"v"(flagv)); }
  }
  if (wave_has_queries) {                          // block 1 of the last tile
    const char* last_stage = smem + ((NT - 1) % NSTAGE) * STAGE_BYTES;
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    load_scales(last_stage, 1, scv2[0]);
    reset_max();
#pragma unroll
    for (int j = 0; j < NV + 2; ++j) test_step(acc1, scv2[0], j, 1);       // (biased build: NV / 2 + 2 of them do something)
    combine_flags();
    if (pend_mask) consume_slots(pend_mask, 0);
    if ((VAR == 0 || VAR >= 3) && any_flag()) { if constexpr (DEFER) rare_path(acc1, last_stage, tile_row0(NT - 1), 1); else rare_log(acc1, scv2[0], tile_row0(NT - 1), 1); }
    if (pend_mask) consume_slots(pend_mask, 0);
  }
  if constexpr (STAMP) {
    uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c;
    const uint64_t dr = __builtin_amdgcn_s_memrealtime() - stamp_r;
    if constexpr (VAR == 5) dc = (rare_acc << 32) | (cons_acc & 0xFFFFFFFFull);   // this build reports those instead of the loop's cycles
    if (wave == 0 && lane == 0) {
      uint64_t* out = reinterpret_cast<uint64_t*>(prog + static_cast<uint64_t>(gridDim.x) * 8) + static_cast<uint64_t>(blockIdx.x) * 2;
      out[0] = dc; out[1] = dr;
    }
  }
  if constexpr (SYNC) { if (wave == 0 && lane == 0) __hip_atomic_store(prog + static_cast<uint64_t>(stream) * 8 + qt, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (stage_counts && lane == 0 && (n_stage1 | n_stage2)) { atomicAdd(stage_counts, n_stage1); atomicAdd(stage_counts + 1, n_stage2); }
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  if constexpr (DEFER) scatter_own_log(mylog, wcnt, sa, lane);
  else verify_and_scatter_i8<DIM>(mylog, wcnt, sa, lane, rows, qlo, thr, qscale, qinv);
}

}  // namespace nvdbhip
