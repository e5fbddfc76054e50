// kernels_exact_mfma.h -- the reference's fp32 score arithmetic on the fp32 matrix cores.
//
// The reference's AVX2 kernels (src/simd_dot.cpp:26-49, 102-124, 160-199) keep EIGHT fp32 accumulators: lane j of the
// 8-wide register sums q[8i+j] * x[8i+j] over i = 0, 1, 2, ... with one fused multiply-add per step, and the result is
// ((a0+a4)+(a1+a5))+((a2+a6)+(a3+a7)).  `v_mfma_f32_16x16x4_f32` computes D = A(16x4) * B(4x16) + C with the four
// products of a K-step folded into C one after the other by fused multiply-adds (MI355X_MICROARCH.md: "exact f32
// (== fmaf chain, bitwise)"), so ONE accumulator tile per reference lane j reproduces that lane's chain bit for bit when its
// K operands are the elements 8(4t+k)+j, k = 0..3, of step t:
//
//     acc_j[row m][query n]  =  fma chain over i = 0 .. dim/8-1 of  q_n[8i+j] * x_m[8i+j]        (i = 4t + k)
//
// 8 accumulator tiles (32 registers), 8 MFMAs per 32 elements of K, then the reference's reduction tree on the VALU.
// Rate: 64 flop / clk / SIMD = 157 TFLOP/s peak, against ~29 TFLOP/s of the VALU kernels of kernels_exact.h
// (scan_exact_kernel / scores_exact_kernel), which stay for every shape this file does not take.
//
// Mapping (one wave = 16 queries x 16-row tiles):
//   * B operand = queries, STATIONARY in registers: lane (n = lane % 16, kq = lane / 16) keeps q_n[8(4t+kq) + j] for all
//     t < DIM/32, j < 8: DIM/4 registers (192 at d = 768; hence DIM <= 768 and one wave per SIMD).
//   * A operand = corpus rows, streamed straight from global memory into registers: lane (m = lane % 16, kq) loads the 8
//     consecutive elements [8(4t+kq), +8) of row m -- one 16-byte load for fp16 rows, 8 bytes for int8, 32 for fp32 --
//     converts them to 8 floats (exact conversions, like vcvtph2ps / vpmovsxbd+vcvtdq2ps) and feeds element j to chain j.
//     The load of the NEXT tile's step t is issued into the same registers right after step t's bytes are converted: one
//     whole tile (6144 MFMA cycles) of prefetch distance without a second register set (fp32 rows: half a tile).
//   * D: lane holds rows 4 * (lane / 16) + v, v = 0..3, of query lane % 16.
//   * workgroup = 4 waves = up to 4 blocks of 16 queries over the SAME rows (HBM sees a row once; the other waves hit
//     L2), or -- for fewer queries -- the waves split the workgroup's tiles.
//   * top-k (scan kernel): per query a best-first list of <= 64 entries in LDS, entry e handled by lane e; the k-th best
//     (score, row) of the lane's own query sits in two registers, so the common case "nothing in this tile beats it" is
//     four compares per lane and one ballot per wave.  Order: (score desc, id asc), as everywhere.
// Shapes taken: DIM in {128, 256, 384, 512, 768} with dim % 32 == 0 (no scalar tail; the tails' special orders stay with
// the VALU kernels), all three dtypes.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_exact.h"
#include "kernels_filter.h"        // glds16_v, NVDB_LPTR: direct-to-LDS loads issued from inline asm

namespace nvdbhip {

typedef float floatx4_t __attribute__((ext_vector_type(4)));

template <int DT> struct ExactRaw;                       // raw bytes of one K-step of one lane (8 elements)
template <> struct ExactRaw<DT_F16> { uint4 v; };
template <> struct ExactRaw<DT_I8> { uint2 v; };
template <> struct ExactRaw<DT_F32> { float4 a, b; };

template <int DT>
__device__ __forceinline__ ExactRaw<DT> exact_raw_load(const char* p) {
  ExactRaw<DT> r;
  if constexpr (DT == DT_F16) r.v = *reinterpret_cast<const uint4*>(p);
  else if constexpr (DT == DT_I8) r.v = *reinterpret_cast<const uint2*>(p);
  else { r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 16); }
  return r;
}
template <int DT>
__device__ __forceinline__ void exact_raw_to_f32(const ExactRaw<DT>& r, float (&x)[8]) {
  if constexpr (DT == DT_F16) {
    x[0] = half_bits_to_float(r.v.x & 0xFFFFu); x[1] = half_bits_to_float(r.v.x >> 16);
    x[2] = half_bits_to_float(r.v.y & 0xFFFFu); x[3] = half_bits_to_float(r.v.y >> 16);
    x[4] = half_bits_to_float(r.v.z & 0xFFFFu); x[5] = half_bits_to_float(r.v.z >> 16);
    x[6] = half_bits_to_float(r.v.w & 0xFFFFu); x[7] = half_bits_to_float(r.v.w >> 16);
  } else if constexpr (DT == DT_I8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = static_cast<float>(static_cast<int>(r.v.x << (24 - 8 * j)) >> 24);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[4 + j] = static_cast<float>(static_cast<int>(r.v.y << (24 - 8 * j)) >> 24);
  } else {
    x[0] = r.a.x; x[1] = r.a.y; x[2] = r.a.z; x[3] = r.a.w; x[4] = r.b.x; x[5] = r.b.y; x[6] = r.b.z; x[7] = r.b.w;
  }
}

template <int DT> constexpr int exact_bpe() { return DT == DT_F32 ? 4 : (DT == DT_F16 ? 2 : 1); }
// K-steps of prefetch distance: a whole tile, half a tile for fp32 rows (8 registers per step instead of 4 / 2)
template <int DT, int DIM> constexpr int exact_ring() { return DT == DT_F32 ? DIM / 64 : DIM / 32; }

// The stationary operand: this lane's share of query `qi` (zeros for a padding query).
template <int DIM>
__device__ __forceinline__ void exact_load_bq(const float* __restrict__ q32, uint32_t qi, uint32_t nq, int kq, float (&bq)[DIM / 32][8]) {
#pragma unroll
  for (int t = 0; t < DIM / 32; ++t) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (qi < nq) {
      const float* p = q32 + static_cast<uint64_t>(qi) * DIM + 8 * (4 * t + kq);
      a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4);
    }
    bq[t][0] = a.x; bq[t][1] = a.y; bq[t][2] = a.z; bq[t][3] = a.w; bq[t][4] = b.x; bq[t][5] = b.y; bq[t][6] = b.z; bq[t][7] = b.w;
  }
}

// One 16-row tile against the wave's 16 queries.  `raw` holds this tile's K-steps (ring of RING steps, the rest is fetched as
// the loop goes); while they are consumed the same registers receive the steps of the tile at `next` (lane's row pointer
// + its kq offset; pass the current pointer again for the last tile: the bytes are loaded and never used).
// Returns s[v] = reference score of (row 4 * (lane / 16) + v of the tile, query lane % 16), int8: before the row scale.
template <int DT, int DIM>
__device__ __forceinline__ void exact_tile(const char* cur, const char* next, ExactRaw<DT> (&raw)[exact_ring<DT, DIM>()],
                                           const float (&bq)[DIM / 32][8], float (&s)[4]) {
  constexpr int T = DIM / 32, RING = exact_ring<DT, DIM>(), STEP_BYTES = 32 * exact_bpe<DT>();
  floatx4_t acc[8];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float x[8];
    exact_raw_to_f32<DT>(raw[t % RING], x);
    // refill the slot: step t + RING of this tile, or step t + RING - T of the next one
    if (t + RING < T) raw[t % RING] = exact_raw_load<DT>(cur + (t + RING) * STEP_BYTES);
    else raw[t % RING] = exact_raw_load<DT>(next + (t + RING - T) * STEP_BYTES);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (t == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], bq[0][j], floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], bq[t][j], acc[j], 0, 0, 0);
    }
    // keep step t's refill load inside step t: left alone, the scheduler gathers all of a tile's loads behind its last MFMA
    // and the prefetch distance collapses to the epilogue
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    // (lo + hi), hadd, hadd  (simd_dot.cpp:38-44)
    const float s0 = acc[0][v] + acc[4][v], s1 = acc[1][v] + acc[5][v], s2 = acc[2][v] + acc[6][v], s3 = acc[3][v] + acc[7][v];
    s[v] = (s0 + s1) + (s2 + s3);
  }
}

constexpr uint32_t EXACT_MFMA_ROWS = 16;                 // rows per tile (MFMA M)
constexpr uint32_t EXACT_MFMA_QB = 16;                   // queries per wave (MFMA N)

// wave -> (query block, row slice) of a workgroup that serves `nqb` (1..4) blocks of 16 queries
__device__ __forceinline__ void exact_wave_role(uint32_t nqb, uint32_t wave, uint32_t& qb, uint32_t& slice, uint32_t& nslice) {
  const uint32_t QW = nqb >= 3 ? 4u : nqb;               // 1, 2 or 4 query blocks side by side
  qb = wave % QW; slice = wave / QW; nslice = 4u / QW;
}

// ------------------------------------------------------------------------------------------------
// scan: grid = (row splits P, ceil(nq / 64)), block = 256.  Same contract as scan_exact_kernel: every wave appends the <= k
// best rows of its tiles (those that also clear the query's global threshold) to its queries' candidate lists; a query
// receives at most P * nslice * k entries.
// ------------------------------------------------------------------------------------------------
template <int DT, int DIM>
__global__ __launch_bounds__(256, 1) void scan_exact_mfma_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const float* __restrict__ q32, uint32_t nq, uint32_t k, const float* __restrict__ thr,
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t* __restrict__ overflow) {
  static_assert(DIM % 32 == 0 && DIM <= 768, "whole MFMA K-steps; 16 queries x DIM fp32 in registers");
  constexpr int BPE = exact_bpe<DT>(), RING = exact_ring<DT, DIM>(), STEP_BYTES = 32 * BPE;
  constexpr uint64_t ROW_BYTES = static_cast<uint64_t>(DIM) * BPE;
  __shared__ float l_s[4][EXACT_MFMA_QB][64];            // per wave and query: best-first list, entry e <-> lane e
  __shared__ uint32_t l_id[4][EXACT_MFMA_QB][64];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const uint32_t qg0 = blockIdx.y * 64u;
  const uint32_t nqb_all = (nq - qg0 + EXACT_MFMA_QB - 1) / EXACT_MFMA_QB, nqb = nqb_all < 4u ? nqb_all : 4u;
  uint32_t qb, slice, nslice;
  exact_wave_role(nqb, wave, qb, slice, nslice);
  if (qb >= nqb) return;                                  // three blocks on four waves: the fourth has no queries
  const uint32_t qi = qg0 + qb * EXACT_MFMA_QB + n16;     // this lane's query

  float bq[DIM / 32][8];
  exact_load_bq<DIM>(q32, qi, nq, kq, bq);

  // this workgroup's tiles, this wave's share of them
  const uint32_t P = gridDim.x, p = blockIdx.x;
  const uint32_t tiles = (row_hi - row_lo + EXACT_MFMA_ROWS - 1) / EXACT_MFMA_ROWS;
  const uint32_t t_lo = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * p / P), t_hi = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * (p + 1) / P);

  // per lane: the k-th best (score, row) of ITS query so far -- what a new row must beat -- and the list length
  float thr_s = NEG_INF;
  uint32_t thr_id = 0xFFFFFFFFu, my_cnt = 0;
  const float gthr = (thr != nullptr && qi < nq) ? thr[qi] : NEG_INF;
  const bool real = qi < nq;
  float quick = gthr;

  const char* gbase = static_cast<const char*>(rows);
  auto lane_ptr = [&](uint32_t tile) -> const char* {     // row (tile, m = lane % 16), clamped into the range; + this lane's K offset
    uint32_t r = row_lo + tile * EXACT_MFMA_ROWS + static_cast<uint32_t>(n16);
    r = r < row_hi ? r : row_hi - 1;
    return gbase + static_cast<uint64_t>(r) * ROW_BYTES + static_cast<uint32_t>(kq) * (8 * BPE);
  };
  uint32_t tile = t_lo + slice;
  if (tile >= t_hi) return;
  // int8: the row scales of this lane's four rows travel one tile ahead, issued BEFORE that tile's row loads -- a load issued
  // behind them would have to wait for all of them (vmcnt counts in issue order) and drain the prefetch once per tile
  float sc_cur[4] = {1.f, 1.f, 1.f, 1.f};
  auto load_scales = [&](uint32_t tl, float (&dst)[4]) {
    if constexpr (DT == DT_I8) {
      const uint32_t r0 = row_lo + tl * EXACT_MFMA_ROWS + 4u * static_cast<uint32_t>(kq);
#pragma unroll
      for (int v = 0; v < 4; ++v) dst[v] = scales[r0 + v < row_hi ? r0 + v : row_hi - 1];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_scales(tile, sc_cur);
  ExactRaw<DT> raw[RING];
  const char* cur = lane_ptr(tile);
#pragma unroll
  for (int t = 0; t < RING; ++t) { raw[t] = exact_raw_load<DT>(cur + t * STEP_BYTES); __builtin_amdgcn_sched_barrier(0); }   // in step order, as the loop refills them: the counted vmcnt waits need one issue order on both paths into the loop

  for (; tile < t_hi; tile += nslice) {
    const uint32_t nxt = tile + nslice < t_hi ? tile + nslice : tile;
    const char* next = lane_ptr(nxt);
    float sc_nxt[4] = {1.f, 1.f, 1.f, 1.f};
    load_scales(nxt, sc_nxt);
    float s[4];
    exact_tile<DT, DIM>(cur, next, raw, bq, s);
    cur = next;
    const uint32_t row0 = row_lo + tile * EXACT_MFMA_ROWS + 4u * static_cast<uint32_t>(kq);     // this lane's rows: row0 + v
    if constexpr (DT == DT_I8) {
#pragma unroll
      for (int v = 0; v < 4; ++v) { s[v] = s[v] * sc_cur[v]; sc_cur[v] = sc_nxt[v]; }                           // simd_dot.cpp:198
    }
    // one compare per tile and lane in front of the full test: a row can only enter with a score >= quick (= the bar handed in and,
    // once this query's list is full, its k-th score; NaN scores never enter)
    const float smax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    if (!__ballot(real && smax >= quick)) continue;
    bool pass[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const uint32_t row = row0 + v;
      pass[v] = real && row < row_hi && s[v] >= gthr && (my_cnt < k || better(s[v], row, thr_s, thr_id));
    }
    // rare: file the passing (row, query) pairs, one at a time, into the queries' lists
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      unsigned long long m = __ballot(pass[v]);
      while (m) {
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t g = static_cast<uint32_t>(L) & 15u;                                  // the pair's query (block-relative)
        const float cs = readlane_f(s[v], L);
        const uint32_t cid = row_lo + tile * EXACT_MFMA_ROWS + 4u * (static_cast<uint32_t>(L) >> 4) + v;
        const uint32_t c = readlane_u(my_cnt, static_cast<int>(g));                         // (lanes g, g + 16, .. hold query g's state)
        const float ts = readlane_f(thr_s, static_cast<int>(g));
        const uint32_t tid = readlane_u(thr_id, static_cast<int>(g));
        if (!(c < k || better(cs, cid, ts, tid))) continue;                                  // an earlier pair of this tile raised the bar
        float es = l_s[wave][g][lane];
        uint32_t eid = l_id[wave][g][lane];
        const bool ahead = static_cast<uint32_t>(lane) < c && better(es, eid, cs, cid);
        const uint32_t pos = static_cast<uint32_t>(__builtin_popcountll(__ballot(ahead)));
        const float up_s = __shfl_up(es, 1);
        const uint32_t up_id = __shfl_up(eid, 1);
        if (static_cast<uint32_t>(lane) > pos) { es = up_s; eid = up_id; }
        else if (static_cast<uint32_t>(lane) == pos) { es = cs; eid = cid; }
        const uint32_t c2 = c < k ? c + 1 : k;
        if (static_cast<uint32_t>(lane) < c2) { l_s[wave][g][lane] = es; l_id[wave][g][lane] = eid; }
        const float nts = readlane_f(es, static_cast<int>(k) - 1);
        const uint32_t ntid = readlane_u(eid, static_cast<int>(k) - 1);
        if (static_cast<uint32_t>(n16) == g) {
          my_cnt = c2;
          if (c2 == k) { thr_s = nts; thr_id = ntid; }
        }
      }
    }
    quick = my_cnt < k ? gthr : fmaxf(gthr, thr_s);
  }
  // append this wave's lists (global candidate lists; select_kernel orders them)
  for (uint32_t g = 0; g < EXACT_MFMA_QB; ++g) {
    const uint32_t q = qg0 + qb * EXACT_MFMA_QB + g;
    const uint32_t c = readlane_u(my_cnt, static_cast<int>(g));
    if (q >= nq || c == 0) continue;
    uint32_t slot0 = 0;
    if (lane == 0) slot0 = atomicAdd(&cnt[q], c);
    slot0 = readlane_u(slot0, 0);
    if (static_cast<uint32_t>(lane) < c) {
      const uint32_t slot = slot0 + lane;
      if (slot < cap) cand[static_cast<uint64_t>(q) * cap + slot] = Cand{l_s[wave][g][lane], l_id[wave][g][lane]};
      else overflow[q] = 1u;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// scores: out[q][row] for every (query, row) -- the any-k path's score matrix (kernels_largek.h).
// grid = (row splits, ceil(nq / 64)), block = 256; ld = row stride of `out` (a multiple of 4: 16-byte stores of 4 rows).
// ------------------------------------------------------------------------------------------------
template <int DT, int DIM>
__global__ __launch_bounds__(256, 1) void scores_exact_mfma_kernel(const void* __restrict__ rows, const float* __restrict__ scales, uint32_t n,
                                                                   const float* __restrict__ q32, uint32_t nq, float* __restrict__ out, uint64_t ld) {
  static_assert(DIM % 32 == 0 && DIM <= 768, "whole MFMA K-steps; 16 queries x DIM fp32 in registers");
  constexpr int BPE = exact_bpe<DT>(), RING = exact_ring<DT, DIM>(), STEP_BYTES = 32 * BPE;
  constexpr uint64_t ROW_BYTES = static_cast<uint64_t>(DIM) * BPE;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  const uint32_t qg0 = blockIdx.y * 64u;
  const uint32_t nqb_all = (nq - qg0 + EXACT_MFMA_QB - 1) / EXACT_MFMA_QB, nqb = nqb_all < 4u ? nqb_all : 4u;
  uint32_t qb, slice, nslice;
  exact_wave_role(nqb, wave, qb, slice, nslice);
  if (qb >= nqb) return;
  const uint32_t qi = qg0 + qb * EXACT_MFMA_QB + n16;
  float bq[DIM / 32][8];
  exact_load_bq<DIM>(q32, qi, nq, kq, bq);
  const uint32_t P = gridDim.x, p = blockIdx.x;
  const uint32_t tiles = (n + EXACT_MFMA_ROWS - 1) / EXACT_MFMA_ROWS;
  const uint32_t t_lo = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * p / P), t_hi = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * (p + 1) / P);
  const char* gbase = static_cast<const char*>(rows);
  auto lane_ptr = [&](uint32_t tile) -> const char* {
    uint32_t r = tile * EXACT_MFMA_ROWS + static_cast<uint32_t>(n16);
    r = r < n ? r : n - 1;
    return gbase + static_cast<uint64_t>(r) * ROW_BYTES + static_cast<uint32_t>(kq) * (8 * BPE);
  };
  uint32_t tile = t_lo + slice;
  if (tile >= t_hi) return;
  float sc_cur[4] = {1.f, 1.f, 1.f, 1.f};              // int8 row scales, one tile ahead (see scan_exact_mfma_kernel)
  auto load_scales = [&](uint32_t tl, float (&dst)[4]) {
    if constexpr (DT == DT_I8) {
      const uint32_t r0 = tl * EXACT_MFMA_ROWS + 4u * static_cast<uint32_t>(kq);
#pragma unroll
      for (int v = 0; v < 4; ++v) dst[v] = scales[r0 + v < n ? r0 + v : n - 1];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  load_scales(tile, sc_cur);
  ExactRaw<DT> raw[RING];
  const char* cur = lane_ptr(tile);
#pragma unroll
  for (int t = 0; t < RING; ++t) { raw[t] = exact_raw_load<DT>(cur + t * STEP_BYTES); __builtin_amdgcn_sched_barrier(0); }
  for (; tile < t_hi; tile += nslice) {
    const uint32_t nxt = tile + nslice < t_hi ? tile + nslice : tile;
    const char* next = lane_ptr(nxt);
    float sc_nxt[4] = {1.f, 1.f, 1.f, 1.f};
    load_scales(nxt, sc_nxt);
    float s[4];
    exact_tile<DT, DIM>(cur, next, raw, bq, s);
    cur = next;
    const uint32_t row0 = tile * EXACT_MFMA_ROWS + 4u * static_cast<uint32_t>(kq);
    if constexpr (DT == DT_I8) {
#pragma unroll
      for (int v = 0; v < 4; ++v) { s[v] = s[v] * sc_cur[v]; sc_cur[v] = sc_nxt[v]; }
    }
    if (qi < nq) {
      float* o = out + static_cast<uint64_t>(qi) * ld + row0;
      if (row0 + 3 < n) *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
      else {
#pragma unroll
        for (int v = 0; v < 4; ++v) if (row0 + v < n) o[v] = s[v];
      }
    }
  }
}

// ================================================================================================
// The same tiles with the rows STAGED THROUGH LDS once per workgroup (full groups of 64 queries: four waves = four blocks of 16
// queries over the same 16-row tile).
//
// The register-direct kernels above let every wave fetch its own copy of a tile in 64-byte-per-row pieces: four times the
// L2 -> CU traffic in half-line requests, and 60-80 TFLOP/s (profiles/r03_exact_mfma_bench.txt).  Here the four waves bring a tile
// in together as 1-KB direct-to-LDS pieces (whole 128-byte lines, each row byte crosses the L2 -> CU path once), three stages
// deep (two for fp32 rows of 768: 48 KB stages), and read their A operand back with ds_read_b128 / b64.
//   LDS image: chunk c (16 bytes) of row r at position c ^ (r & 15) of the row -- applied on the SOURCE address of the
//   direct-to-LDS load, whose LDS side is linear.  Lane (x15, kq) reads chunk 4t + kq (fp16), chunks 2(4t + kq), +1 (fp32) or
//   half kq & 1 of chunk 2t + kq / 2 (int8) of row x15: 16 distinct 16-byte bank slots per lane group in every case.
//   One s_barrier per tile; loads counted by hand (s_waitcnt vmcnt) because they are invisible to the compiler, which is also why
//   the int8 row scales come in through LDS: an ordinary load's compiler-made wait would drain the prefetch.
// Results are bit-identical to the register-direct kernels and to the VALU kernels (same chains, same order).
// ================================================================================================
template <int DT, int DIM> constexpr bool exact_lds_shape() {
  return (16 * DIM * exact_bpe<DT>()) % 4096 == 0 && (DIM * exact_bpe<DT>()) % 256 == 0;    // whole pieces per wave; XOR stays inside a row
}
template <int DT, int DIM> constexpr int exact_lds_stages() { return 3 * 16 * DIM * exact_bpe<DT>() + 3 * 256 + 33 * 1024 <= 160 * 1024 ? 3 : 2; }
template <int DT, int DIM> constexpr int exact_lds_bytes() { return exact_lds_stages<DT, DIM>() * (16 * DIM * exact_bpe<DT>() + 256); }

// 4 bytes per lane HBM -> LDS (the int8 row scales): LDS address = lds_off + lane * 4
__device__ __forceinline__ void glds4_v(const void* gptr, uint32_t lds_off) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gptr), "s"(lds_off) : "memory");
}

// SCORES = false: the scan (top-k lists, contract of scan_exact_kernel); true: the score matrix (contract of scores_exact_kernel).
// grid = (row splits, nq / 64): ONLY full groups of 64 queries (the caller sends the rest to the register-direct kernels).
template <int DT, int DIM, bool SCORES>
__global__ __launch_bounds__(256, 1) void exact_mfma_lds_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const float* __restrict__ q32, uint32_t nq, uint32_t k, const float* __restrict__ thr,
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t* __restrict__ overflow,
    float* __restrict__ out, uint64_t ld) {
  static_assert(DIM % 32 == 0 && DIM <= 768 && exact_lds_shape<DT, DIM>(), "whole MFMA K-steps; whole 1-KB pieces per wave");
  constexpr int BPE = exact_bpe<DT>(), T = DIM / 32, ROW_BYTES = DIM * BPE, CHUNKS_PER_ROW = ROW_BYTES / 16;
  constexpr int DATA_BYTES = 16 * ROW_BYTES, STAGE_BYTES = DATA_BYTES + 256;       // + four 64-byte copies of the tile's row scales (int8)
  constexpr int NSTAGE = exact_lds_stages<DT, DIM>(), PIECES = DATA_BYTES / 1024, PPW = PIECES / 4;
  constexpr int LPT = PPW + (DT == DT_I8 ? 1 : 0);                                 // loads per wave and tile
  constexpr int RING = 4;                                                          // K-steps of LDS reads in flight
  static_assert(T % PPW == 0 || PPW % T == 0, "pieces spread evenly over the K-steps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float l_s[SCORES ? 1 : 4][SCORES ? 1 : EXACT_MFMA_QB][64];
  __shared__ uint32_t l_id[SCORES ? 1 : 4][SCORES ? 1 : EXACT_MFMA_QB][64];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x15 = lane & 15, kq = lane >> 4;
  const uint32_t qg0 = blockIdx.y * 64u;
  const uint32_t qi = qg0 + wave * EXACT_MFMA_QB + x15;          // < nq: full groups only

  float bq[T][8];
  exact_load_bq<DIM>(q32, qi, nq, kq, bq);
  // consume the fragments HERE: hipcc otherwise defers its wait for these loads to their first use inside the tile loop, and that
  // s_waitcnt vmcnt(0) would also wait for the direct-to-LDS prefetches it cannot see -- once per tile
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(bq[t][j]));
  const float gthr = (!SCORES && thr != nullptr) ? thr[qi] : NEG_INF;
  asm volatile("" ::"v"(gthr));

  const uint32_t P = gridDim.x, p = blockIdx.x;
  const uint32_t tiles = (row_hi - row_lo + EXACT_MFMA_ROWS - 1) / EXACT_MFMA_ROWS;
  const uint32_t t_lo = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * p / P), t_hi = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * (p + 1) / P);
  if (t_lo >= t_hi) return;

  // loader: piece i of this wave = linear 16-byte slots [(wave * PPW + i) * 64, +64) of the stage; slot L = row L / CHUNKS_PER_ROW,
  // position L % CHUNKS_PER_ROW, filled from source chunk position ^ (row & 15) of that row (clamped into the row range)
  uint32_t piece_row[PPW], piece_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t L = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = L / CHUNKS_PER_ROW, cpos = L % CHUNKS_PER_ROW;
    piece_row[i] = r;
    piece_off[i] = (cpos ^ (r & 15u)) << 4;
  }
  const char* gbase = static_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  auto issue_piece = [&](uint32_t tile, int i) {
    const uint32_t tl = tile < t_hi ? tile : t_hi - 1;           // beyond my range: the last tile again (never read)
    uint32_t r = row_lo + tl * EXACT_MFMA_ROWS + piece_row[i];
    r = r < row_hi ? r : row_hi - 1;
    glds16_v(gbase + static_cast<uint64_t>(r) * ROW_BYTES + piece_off[i], lds_base + (tl % NSTAGE) * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
  auto issue_scales = [&](uint32_t tile) {
    if constexpr (DT == DT_I8) {
      const uint32_t tl = tile < t_hi ? tile : t_hi - 1;
      uint32_t r = row_lo + tl * EXACT_MFMA_ROWS + static_cast<uint32_t>(x15);
      r = r < row_hi ? r : row_hi - 1;
      if (lane < 16) glds4_v(scales + r, lds_base + (tl % NSTAGE) * STAGE_BYTES + DATA_BYTES + wave * 64);
    }
  };
  // NOTE: a stage is chosen by the TILE index (tile % NSTAGE), so prefetches beyond t_hi land in the stage of the last tile --
  // only after that tile has been consumed?  No: they are issued while it is being read.  Keep them out: see `have` below.
  auto issue_tile = [&](uint32_t tile) {
    if (tile < t_hi) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(tile, i);
      issue_scales(tile);
    }
  };

  // this lane's read offsets inside a stage.  Chunk index c of step t XOR x15 splits into a part that depends on (t mod 2^m) only
  // through an XOR of address bits and a part that is a plain multiple of 256 bytes, so a handful of base registers plus immediate
  // offsets address every step (written out so that the compiler keeps it that way: 24 computed addresses spilled into AGPRs, and
  // every v_accvgpr_read / v_add between two fp32 MFMAs costs ~10 cycles -- profiles/r03_mfma_f32_rate.txt)
  const uint32_t row_off = static_cast<uint32_t>(x15) * ROW_BYTES;
  const uint32_t a_f16 = row_off + ((static_cast<uint32_t>(kq) ^ static_cast<uint32_t>(x15)) << 4);                       // chunk 4t + kq
  const uint32_t a_f32_0 = row_off + (((2u * kq) ^ static_cast<uint32_t>(x15)) << 4);                                     // chunks 8t + 2kq, + 1
  const uint32_t a_f32_1 = row_off + (((2u * kq + 1u) ^ static_cast<uint32_t>(x15)) << 4);
  const uint32_t a_i8 = row_off + (((static_cast<uint32_t>(kq) >> 1) ^ static_cast<uint32_t>(x15)) << 4) + 8u * (kq & 1);  // half kq & 1 of chunk 2t + kq / 2
  auto read_step = [&](const char* stage, int t) -> ExactRaw<DT> {
    ExactRaw<DT> r;
    if constexpr (DT == DT_F16) {
      r.v = *reinterpret_cast<const uint4*>(stage + (a_f16 ^ ((t & 3) << 6)) + (t >> 2) * 256);
    } else if constexpr (DT == DT_F32) {
      r.a = *reinterpret_cast<const float4*>(stage + (a_f32_0 ^ ((t & 1) << 7)) + (t >> 1) * 256);
      r.b = *reinterpret_cast<const float4*>(stage + (a_f32_1 ^ ((t & 1) << 7)) + (t >> 1) * 256);
    } else {
      r.v = *reinterpret_cast<const uint2*>(stage + (a_i8 ^ ((t & 7) << 5)) + (t >> 3) * 256);
    }
    return r;
  };

  float thr_s = NEG_INF;
  uint32_t thr_id = 0xFFFFFFFFu, my_cnt = 0;
  float quick = gthr;

  // prologue: the first NSTAGE - 1 tiles
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) issue_tile(t_lo + st);
  // tiles whose loads were actually issued count in vmcnt; near the end fewer are outstanding, and a too-generous count would
  // let a stage be read before it has landed: wait for everything once the prefetch has run dry
  for (uint32_t tile = t_lo; tile < t_hi; ++tile) {
    if (tile + NSTAGE - 1 <= t_hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * (NSTAGE - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // tile `tile` has landed for every wave; nobody still reads the stage of tile - 1
    const char* stage = smem + (tile % NSTAGE) * STAGE_BYTES;
    const uint32_t nxt = tile + NSTAGE - 1;
    ExactRaw<DT> ring[RING];
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) ring[t] = read_step(stage, t);
    floatx4_t acc[8];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + RING - 1 < T) ring[(t + RING - 1) % RING] = read_step(stage, t + RING - 1);
      float x[8];
      exact_raw_to_f32<DT>(ring[t % RING], x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (t == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], bq[0][j], floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], bq[t][j], acc[j], 0, 0, 0);
      }
      // the next tile-but-one's pieces, spread over the K-steps
      if (nxt < t_hi) {
        if constexpr (T >= PPW) { if (t % (T / PPW) == 0) issue_piece(nxt, t / (T / PPW)); }
        else {
#pragma unroll
          for (int i = 0; i < PPW / T; ++i) issue_piece(nxt, t * (PPW / T) + i);
        }
        if (t == 1) issue_scales(nxt);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float s[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float s0 = acc[0][v] + acc[4][v], s1 = acc[1][v] + acc[5][v], s2 = acc[2][v] + acc[6][v], s3 = acc[3][v] + acc[7][v];
      s[v] = (s0 + s1) + (s2 + s3);
    }
    const uint32_t row0 = row_lo + tile * EXACT_MFMA_ROWS + 4u * static_cast<uint32_t>(kq);
    if constexpr (DT == DT_I8) {
      const float4 sc = *reinterpret_cast<const float4*>(stage + DATA_BYTES + wave * 64 + 16 * kq);
      s[0] *= sc.x; s[1] *= sc.y; s[2] *= sc.z; s[3] *= sc.w;                      // simd_dot.cpp:198
    }
    if constexpr (SCORES) {
      float* o = out + static_cast<uint64_t>(qi) * ld + row0;
      if (row0 + 3 < row_hi) *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
      else {
#pragma unroll
        for (int v = 0; v < 4; ++v) if (row0 + v < row_hi) o[v] = s[v];
      }
    } else {
      const float smax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));       // (the quick test of scan_exact_mfma_kernel)
      if (!__ballot(smax >= quick)) continue;
      bool pass[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const uint32_t row = row0 + v;
        pass[v] = row < row_hi && s[v] >= gthr && (my_cnt < k || better(s[v], row, thr_s, thr_id));
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        unsigned long long m = __ballot(pass[v]);
        while (m) {
          const int L = __builtin_ctzll(m);
          m &= m - 1;
          const uint32_t g = static_cast<uint32_t>(L) & 15u;
          const float cs = readlane_f(s[v], L);
          const uint32_t cid = row_lo + tile * EXACT_MFMA_ROWS + 4u * (static_cast<uint32_t>(L) >> 4) + v;
          const uint32_t c = readlane_u(my_cnt, static_cast<int>(g));
          const float ts = readlane_f(thr_s, static_cast<int>(g));
          const uint32_t tid = readlane_u(thr_id, static_cast<int>(g));
          if (!(c < k || better(cs, cid, ts, tid))) continue;
          float es = l_s[SCORES ? 0 : wave][SCORES ? 0 : g][lane];
          uint32_t eid = l_id[SCORES ? 0 : wave][SCORES ? 0 : g][lane];
          const bool ahead = static_cast<uint32_t>(lane) < c && better(es, eid, cs, cid);
          const uint32_t pos = static_cast<uint32_t>(__builtin_popcountll(__ballot(ahead)));
          const float up_s = __shfl_up(es, 1);
          const uint32_t up_id = __shfl_up(eid, 1);
          if (static_cast<uint32_t>(lane) > pos) { es = up_s; eid = up_id; }
          else if (static_cast<uint32_t>(lane) == pos) { es = cs; eid = cid; }
          const uint32_t c2 = c < k ? c + 1 : k;
          if (static_cast<uint32_t>(lane) < c2) { l_s[SCORES ? 0 : wave][SCORES ? 0 : g][lane] = es; l_id[SCORES ? 0 : wave][SCORES ? 0 : g][lane] = eid; }
          const float nts = readlane_f(es, static_cast<int>(k) - 1);
          const uint32_t ntid = readlane_u(eid, static_cast<int>(k) - 1);
          if (static_cast<uint32_t>(x15) == g) {
            my_cnt = c2;
            if (c2 == k) { thr_s = nts; thr_id = ntid; }
          }
        }
      }
      quick = my_cnt < k ? gthr : fmaxf(gthr, thr_s);
    }
  }
  if constexpr (!SCORES) {
    for (uint32_t g = 0; g < EXACT_MFMA_QB; ++g) {
      const uint32_t q = qg0 + wave * EXACT_MFMA_QB + g;
      const uint32_t c = readlane_u(my_cnt, static_cast<int>(g));
      if (c == 0) continue;
      uint32_t slot0 = 0;
      if (lane == 0) slot0 = atomicAdd(&cnt[q], c);
      slot0 = readlane_u(slot0, 0);
      if (static_cast<uint32_t>(lane) < c) {
        const uint32_t slot = slot0 + lane;
        if (slot < cap) cand[static_cast<uint64_t>(q) * cap + slot] = Cand{l_s[SCORES ? 0 : wave][SCORES ? 0 : g][lane], l_id[SCORES ? 0 : wave][SCORES ? 0 : g][lane]};
        else overflow[q] = 1u;
      }
    }
  }
}

// ================================================================================================
// fp16 / int8 rows through an fp32 LDS IMAGE (round 4).
//
// The kernels above convert an A operand right in front of each MFMA, every wave for itself: four times the conversions, all of
// them inside the MFMA stream -- 86 / 81 TFLOP/s for fp16 / int8 rows where fp32 rows, which need no conversion, reach 101.
// Here a tile is converted ONCE per workgroup: each of the four waves brings a quarter of the raw 16-row tile in (direct-to-LDS
// 16-byte loads, 1 KB per wave-instruction, into a per-lane landing slot), converts it -- the exact conversions of the reference,
// vcvtph2ps / vpmovsxbd + vcvtdq2ps (src/simd_dot.cpp:102-124, 160-199) -- and writes it into an fp32 image of the tile in LDS;
// all four waves then feed their MFMAs from the image with ds_read_b128 and immediate offsets, exactly like fp32 rows.
//   per tile and wave:  192 MFMAs on image[t % 2]; behind them, spread over the K-steps: my quarter of tile t+1 out of its landing
//                       slots -> image[(t+1) % 2], each slot refilled with tile t+2 as soon as it is empty;  epilogue;  s_barrier
//   image layout: row r at r * DIM * 4, 16-byte chunk p at position p ^ (r & 15) (the fp32-row build's, conflict-free for the
//   A-operand reads); the lane -> (row, raw chunk) mapping of the conversion spreads the 16 lanes of a write phase over 2 (fp16) /
//   4 (int8) rows, so that their fp32 chunks fall into 16 different bank groups without any reordering.
//   LDS at d = 768: 2 x 48 KB image + 24 KB (fp16) / 12 KB (int8) of landing slots + 256 B of row scales (+ 32 KB of top-k lists in
//   the scan build): 152.3 KB of 160.
// How it got here (profiles/r04_exact_img_ablation.txt; 256 queries x 4M fp16 rows): quarter in registers, converted between two
// tiles 97 TFLOP/s -> landing slots in LDS 103 -> conversion behind the MFMAs 111 -> one compare in front of the top-k test 113
// (int8 rows 96 -> 115).  What is left against the MFMA stream alone (133): ~7 % the six direct-to-LDS loads per tile (~40 cycles
// of the matrix pipe each, more when split into dword loads), ~8 % the conversion, ~5 % epilogue, ~4 % the barrier.
// Same chains, same order: bit-identical to every other exact kernel (test_exact_scores_on_the_fp32_matrix_cores).
// ================================================================================================
template <int DT, int DIM> constexpr bool exact_img_shape() { return DT != DT_F32 && (16 * DIM * exact_bpe<DT>()) % 4096 == 0; }   // whole 1-KB pieces per wave
template <int DT, int DIM> constexpr int exact_img_bytes() { return 2 * 16 * DIM * 4 + 16 * DIM * exact_bpe<DT>() + 256; }   // two images + the raw tile in flight + row scales

template <int DT, int DIM, bool SCORES>
__global__ __launch_bounds__(256, 1) void exact_mfma_img_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const float* __restrict__ q32, uint32_t nq, uint32_t k, const float* __restrict__ thr,
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t* __restrict__ overflow,
    float* __restrict__ out, uint64_t ld) {
  static_assert(DT != DT_F32 && DIM % 32 == 0 && DIM <= 768 && exact_img_shape<DT, DIM>(), "fp16 / int8 rows; whole 1-KB pieces per wave");
  static_assert(exact_img_bytes<DT, DIM>() + (SCORES ? 1 : 4 * EXACT_MFMA_QB) * 64 * 8 <= 160 * 1024, "images + landing slots + top-k lists fit the 160 KB of LDS");
  constexpr int BPE = exact_bpe<DT>(), T = DIM / 32, ROW_BYTES = DIM * BPE, CHUNKS_PER_ROW = ROW_BYTES / 16;
  constexpr int IMG_ROW = DIM * 4, IMG_BYTES = 16 * IMG_ROW;
  constexpr int PIECES = 16 * ROW_BYTES / 1024, PPW = PIECES / 4;                   // 16-byte raw chunks per lane and tile
  constexpr int RING = 4;                                                          // K-steps of LDS reads in flight
  constexpr int EPC = 16 / BPE;                                                    // elements per raw chunk: 8 halves / 16 bytes
  constexpr int WPC = EPC / 4;                                                     // fp32 chunks per raw chunk: 2 / 4
  extern __shared__ __attribute__((aligned(16))) char smem[];                      // [image 0 | image 1 | raw tile: 4 waves x PPW KB]
  __shared__ float l_s[SCORES ? 1 : 4][SCORES ? 1 : EXACT_MFMA_QB][64];
  __shared__ uint32_t l_id[SCORES ? 1 : 4][SCORES ? 1 : EXACT_MFMA_QB][64];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int x15 = lane & 15, kq = lane >> 4;
  const uint32_t qg0 = blockIdx.y * 64u;
  const uint32_t qi = qg0 + wave * EXACT_MFMA_QB + x15;          // < nq: full groups only

  float bq[T][8];
  exact_load_bq<DIM>(q32, qi, nq, kq, bq);
#pragma unroll
  for (int t = 0; t < T; ++t)          // consumed HERE (see exact_mfma_lds_kernel: a wait deferred into the loop would drain the asm loads)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(bq[t][j]));
  const float gthr = (!SCORES && thr != nullptr) ? thr[qi] : NEG_INF;
  asm volatile("" ::"v"(gthr));

  const uint32_t P = gridDim.x, p = blockIdx.x;
  const uint32_t tiles = (row_hi - row_lo + EXACT_MFMA_ROWS - 1) / EXACT_MFMA_ROWS;
  const uint32_t t_lo = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * p / P), t_hi = static_cast<uint32_t>(static_cast<uint64_t>(tiles) * (p + 1) / P);
  if (t_lo >= t_hi) return;

  // my quarter of a tile = rows 4 * wave .. + 3.  Raw chunk i of this lane is chunk `pc(i)` of row `pr(i)`, chosen so that the 16 lanes
  // of one LDS write phase cover WPC rows x (16 / WPC) consecutive chunks: their fp32 chunks WPC * c + j then fall, for every j, into
  // 16 different bank groups of the XOR-swizzled image ((WPC * c + j) ^ (r & 15): the chunks differ in the upper bits, the rows in the
  // low log2(WPC) bits) -- no conflict and no data shuffling.  (A linear lane -> chunk mapping needed a rotated write order to avoid
  // the conflicts: 146 v_cndmask per tile for int8 rows, three times the conversions themselves.)
  //   fp16 (WPC 2; lane = 16 g + l): row 4w + 2 (g & 1) + (l & 1), chunk 16 i + 8 (g >> 1) + (l >> 1)     (a 16-lane group: 2 rows x 8 chunks)
  //   int8 (WPC 4):                  row 4w + (lane & 3),           chunk 16 i + (lane >> 2)               (a 16-lane group: 4 rows x 4 chunks)
  // Either way piece i of a wave covers chunks [16 i, 16 i + 16) of its four rows; global side: whole 128-byte lines per 16-lane group.
  const uint32_t lg = static_cast<uint32_t>(lane) >> 4, l15 = static_cast<uint32_t>(lane) & 15u;
  auto pr = [&](int) -> uint32_t { return WPC == 2 ? 4u * wave + 2u * (lg & 1u) + (l15 & 1u) : 4u * wave + (static_cast<uint32_t>(lane) & 3u); };
  auto pc = [&](int i) -> uint32_t { return WPC == 2 ? 16u * static_cast<uint32_t>(i) + 8u * (lg >> 1) + (l15 >> 1) : 16u * static_cast<uint32_t>(i) + (static_cast<uint32_t>(lane) >> 2); };
  static_assert(CHUNKS_PER_ROW == 16 * PPW, "piece i covers chunks [16 i, 16 i + 16) of the wave's four rows");
  const char* gbase = static_cast<const char*>(rows);
  // The raw quarter-tile travels HBM -> LDS directly (global_load_lds_dwordx4: lane l's 16 bytes land at piece base + 16 l) and
  // is read back by the SAME lane when it is converted one iteration later: LDS as the landing zone of an asynchronous load, no
  // register is held across the MFMA stream and only this wave's own vmcnt orders it (no barrier).  With the quarter in registers
  // instead (24 VGPRs live across the loop of a kernel that already uses all 256) hipcc split a live range right behind the loads
  // -- `s_waitcnt vmcnt(0); v_mov` after the fifth of six -- and every tile paid a full memory latency: 18 % of the kernel
  // (profiles/r04_exact_img_ablation.txt).  The loads are issued from inline asm and counted by hand, as in the kernel above; the
  // int8 row scales take the same road (an ordinary load's compiler-made wait would drain the prefetch).
  //   LPT loads per wave and tile, in this order: pieces 0 .. PPW-1, then (int8) the 16 row scales.  Slot i is refilled -- with the
  //   tile after next -- right after it has been read, so whenever slot i is about to be read exactly LPT - 1 younger loads may
  //   still be in flight: s_waitcnt vmcnt(LPT - 1).  (A store of the score build in between only makes that wait stricter.)
  constexpr int LPT = PPW + (DT == DT_I8 ? 1 : 0);
  const uint32_t lds_raw = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem))) + 2u * IMG_BYTES + wave * (PPW * 1024u);
  const uint32_t lds_sc = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem))) + 2u * IMG_BYTES + 4u * (PPW * 1024u) + wave * 64u;
  const char* my_raw = smem + 2 * IMG_BYTES + wave * (PPW * 1024) + lane * 16;
  const char* my_sc = smem + 2 * IMG_BYTES + 4 * (PPW * 1024) + wave * 64 + kq * 16;       // scales of rows 4 kq .. + 3
  auto fetch_piece = [&](uint32_t tile, int i) {                   // (a tile beyond my range: the last one again, never converted)
    const uint32_t tl = tile < t_hi ? tile : t_hi - 1;
    uint32_t r = row_lo + tl * EXACT_MFMA_ROWS + pr(i);
    r = r < row_hi ? r : row_hi - 1;
    glds16_v(gbase + static_cast<uint64_t>(r) * ROW_BYTES + (pc(i) << 4), lds_raw + static_cast<uint32_t>(i) * 1024u);
  };
  auto fetch_scales = [&](uint32_t tile) {
    if constexpr (DT == DT_I8) {
      const uint32_t tl = tile < t_hi ? tile : t_hi - 1;
      uint32_t r = row_lo + tl * EXACT_MFMA_ROWS + static_cast<uint32_t>(x15);
      r = r < row_hi ? r : row_hi - 1;
      if (lane < 16) glds4_v(scales + r, lds_sc);
    }
  };
  // slot i, read into registers one K-step earlier (landed: the reader has waited) -> fp32 -> image of `tile`
  auto convert_piece = [&](uint32_t tile, int i, const uint4& raw) {
    char* img = smem + (tile & 1u) * IMG_BYTES;
    const uint32_t r = pr(i), c = pc(i);
    float x[EPC];
    if constexpr (DT == DT_F16) {
      ExactRaw<DT_F16> rw; rw.v = raw;
      exact_raw_to_f32<DT_F16>(rw, x);
    } else {
      ExactRaw<DT_I8> lo, hi; lo.v = make_uint2(raw.x, raw.y); hi.v = make_uint2(raw.z, raw.w);
      float xl[8], xh[8];
      exact_raw_to_f32<DT_I8>(lo, xl); exact_raw_to_f32<DT_I8>(hi, xh);
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[e] = xl[e]; x[8 + e] = xh[e]; }
    }
#pragma unroll
    for (int j = 0; j < WPC; ++j) {
      const uint32_t pos = (WPC * c + static_cast<uint32_t>(j)) ^ (r & 15u);
      *reinterpret_cast<float4*>(img + r * IMG_ROW + (pos << 4)) = make_float4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
    }
  };
  auto read_piece = [&](int i) -> uint4 { return *reinterpret_cast<const uint4*>(my_raw + i * 1024); };
  float sc_nxt[4] = {1.f, 1.f, 1.f, 1.f};
  auto read_scales = [&]() {
    if constexpr (DT == DT_I8) {
      const float4 v = *reinterpret_cast<const float4*>(my_sc);
      sc_nxt[0] = v.x; sc_nxt[1] = v.y; sc_nxt[2] = v.z; sc_nxt[3] = v.w;
    }
  };
  // Inside the tile loop (K-step t and MFMA j are compile-time constants after unrolling): piece i of tile + 1 is read out of its slot
  // behind the MFMAs of K-step conv_step(i) - 1, and converted one element (fp16) / two (int8) behind EACH of the eight MFMAs of
  // K-step conv_step(i): a vector instruction issued while an MFMA runs costs ~3 cycles of the matrix pipe
  // (profiles/r03_mfma_f32_rate.txt), a serial conversion phase between two tiles cost ~1600 cycles per tile
  // (profiles/r04_exact_img_ablation.txt).  The slot is refilled with tile + 2 as soon as its bytes are in registers.  All of it is
  // unconditional: behind my last tile the (clamped) prefetch is converted into the image nobody reads any more.
  auto conv_step = [](int i) constexpr { return ((2 * i + 1) * T) / (2 * PPW); };
  static_assert(T / PPW >= 4, "a piece's read and conversion steps do not collide with its neighbours'");
  // image write addresses: fp32 chunk jj of raw chunk 16 i + cl of row r lands at 16-byte position WPC * 16 * i + ((WPC * cl + jj) ^ (r & 15))
  // (the XOR stays inside the low four bits): one base per jj, the piece is an immediate
  uint32_t wbase[WPC];
#pragma unroll
  for (int jj = 0; jj < WPC; ++jj) wbase[jj] = pr(0) * IMG_ROW + (((WPC * pc(0) + static_cast<uint32_t>(jj)) ^ (pr(0) & 15u)) << 4);
  uint4 rawv = make_uint4(0u, 0u, 0u, 0u);
  float cx[EPC];
  auto raw_dword = [&](int d) -> uint32_t { return d == 0 ? rawv.x : d == 1 ? rawv.y : d == 2 ? rawv.z : rawv.w; };
  auto micro_convert = [&](uint32_t tile, int i, int j) {         // behind MFMA j of K-step conv_step(i) of `tile`: piece i of tile + 1
    char* wimg = smem + ((tile + 1u) & 1u) * IMG_BYTES + i * (WPC * 256);
    const uint32_t d = raw_dword(j >> 1);
    if constexpr (DT == DT_F16) {
      cx[j] = half_bits_to_float((j & 1) ? d >> 16 : d & 0xFFFFu);
      if ((j & 3) == 3) *reinterpret_cast<float4*>(wimg + wbase[j >> 2]) = make_float4(cx[j - 3], cx[j - 2], cx[j - 1], cx[j]);
    } else {
      const int b = 2 * (j & 1);
      cx[2 * j] = static_cast<float>(static_cast<int>(d << (24 - 8 * b)) >> 24);
      cx[2 * j + 1] = static_cast<float>(static_cast<int>(d << (16 - 8 * b)) >> 24);
      if (j & 1) *reinterpret_cast<float4*>(wimg + wbase[j >> 1]) = make_float4(cx[2 * j - 2], cx[2 * j - 1], cx[2 * j], cx[2 * j + 1]);
    }
    if (j == 7) fetch_piece(tile + 2, i);                           // (the slot's old bytes are in registers)
  };
  auto staged_reads = [&](uint32_t tile, int t) {                  // behind the MFMAs of K-step t
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (t == conv_step(i) - 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT - 1) : "memory");
        rawv = read_piece(i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (DT == DT_I8) {
      if (t == T - 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT - 1) : "memory");
        read_scales();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (read before the slot is refilled)
        fetch_scales(tile + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // A-operand reads out of the image: chunks 8t + 2kq, + 1 of row x15 at position ^ x15 (the fp32-row build's addressing:
  // two base registers plus immediates)
  const uint32_t row_off = static_cast<uint32_t>(x15) * IMG_ROW;
  const uint32_t a0 = row_off + (((2u * kq) ^ static_cast<uint32_t>(x15)) << 4);
  const uint32_t a1 = row_off + (((2u * kq + 1u) ^ static_cast<uint32_t>(x15)) << 4);
  auto read_step = [&](const char* img, int t) -> ExactRaw<DT_F32> {
    ExactRaw<DT_F32> r;
    r.a = *reinterpret_cast<const float4*>(img + (a0 ^ ((t & 1) << 7)) + (t >> 1) * 256);
    r.b = *reinterpret_cast<const float4*>(img + (a1 ^ ((t & 1) << 7)) + (t >> 1) * 256);
    return r;
  };

  float thr_s = NEG_INF;
  uint32_t thr_id = 0xFFFFFFFFu, my_cnt = 0;
  float quick = gthr;
  float sc_cur[4] = {1.f, 1.f, 1.f, 1.f};

  // prologue: image of the first tile; my quarter of the second on its way
#pragma unroll
  for (int i = 0; i < PPW; ++i) fetch_piece(t_lo, i);
  fetch_scales(t_lo);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < PPW; ++i) convert_piece(t_lo, i, read_piece(i));
  read_scales();
#pragma unroll
  for (int v = 0; v < 4; ++v) sc_cur[v] = sc_nxt[v];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // slots read, my image writes performed before I signal
#pragma unroll
  for (int i = 0; i < PPW; ++i) fetch_piece(t_lo + 1, i);
  fetch_scales(t_lo + 1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // one tile: MFMAs out of image tile & 1 with my quarter of tile + 1 converted into the other image (and tile + 2 fetched) behind
  // them, epilogue, barrier
  auto one_tile = [&](uint32_t tile) {
    const char* img = smem + (tile & 1u) * IMG_BYTES;
    ExactRaw<DT_F32> ring[RING];
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) ring[t] = read_step(img, t);
    floatx4_t acc[8];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + RING - 1 < T) ring[(t + RING - 1) % RING] = read_step(img, t + RING - 1);
      float x[8];
      exact_raw_to_f32<DT_F32>(ring[t % RING], x);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (t == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], bq[0][j], floatx4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], bq[t][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
          if (t == conv_step(i)) {
            micro_convert(tile, i, j);
            __builtin_amdgcn_sched_barrier(0);                     // pinned behind MFMA j
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      staged_reads(tile, t);
    }
    float s[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float s0 = acc[0][v] + acc[4][v], s1 = acc[1][v] + acc[5][v], s2 = acc[2][v] + acc[6][v], s3 = acc[3][v] + acc[7][v];
      s[v] = (s0 + s1) + (s2 + s3);
    }
    const uint32_t row0 = row_lo + tile * EXACT_MFMA_ROWS + 4u * static_cast<uint32_t>(kq);
    if constexpr (DT == DT_I8) {
#pragma unroll
      for (int v = 0; v < 4; ++v) s[v] *= sc_cur[v];                               // simd_dot.cpp:198
    }
    if constexpr (SCORES) {
      float* o = out + static_cast<uint64_t>(qi) * ld + row0;
      if (row0 + 3 < row_hi) *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
      else {
#pragma unroll
        for (int v = 0; v < 4; ++v) if (row0 + v < row_hi) o[v] = s[v];
      }
    } else {
      // one compare per tile and lane in front of the full test: a row can only enter with a score >= quick (= the bar handed in
      // and, once my list is full, its k-th score; NaN scores never enter, as before)
      const float smax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      if (__ballot(smax >= quick)) {
        bool pass[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const uint32_t row = row0 + v;
          pass[v] = row < row_hi && s[v] >= gthr && (my_cnt < k || better(s[v], row, thr_s, thr_id));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          unsigned long long m = __ballot(pass[v]);
          while (m) {
            const int L = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t g = static_cast<uint32_t>(L) & 15u;
            const float cs = readlane_f(s[v], L);
            const uint32_t cid = row_lo + tile * EXACT_MFMA_ROWS + 4u * (static_cast<uint32_t>(L) >> 4) + v;
            const uint32_t c = readlane_u(my_cnt, static_cast<int>(g));
            const float ts = readlane_f(thr_s, static_cast<int>(g));
            const uint32_t tid = readlane_u(thr_id, static_cast<int>(g));
            if (!(c < k || better(cs, cid, ts, tid))) continue;
            float es = l_s[SCORES ? 0 : wave][SCORES ? 0 : g][lane];
            uint32_t eid = l_id[SCORES ? 0 : wave][SCORES ? 0 : g][lane];
            const bool ahead = static_cast<uint32_t>(lane) < c && better(es, eid, cs, cid);
            const uint32_t pos = static_cast<uint32_t>(__builtin_popcountll(__ballot(ahead)));
            const float up_s = __shfl_up(es, 1);
            const uint32_t up_id = __shfl_up(eid, 1);
            if (static_cast<uint32_t>(lane) > pos) { es = up_s; eid = up_id; }
            else if (static_cast<uint32_t>(lane) == pos) { es = cs; eid = cid; }
            const uint32_t c2 = c < k ? c + 1 : k;
            if (static_cast<uint32_t>(lane) < c2) { l_s[SCORES ? 0 : wave][SCORES ? 0 : g][lane] = es; l_id[SCORES ? 0 : wave][SCORES ? 0 : g][lane] = eid; }
            const float nts = readlane_f(es, static_cast<int>(k) - 1);
            const uint32_t ntid = readlane_u(eid, static_cast<int>(k) - 1);
            if (static_cast<uint32_t>(x15) == g) {
              my_cnt = c2;
              if (c2 == k) { thr_s = nts; thr_id = ntid; }
            }
          }
        }
        quick = my_cnt < k ? gthr : fmaxf(gthr, thr_s);
      }
    }
    if constexpr (DT == DT_I8) {
#pragma unroll
      for (int v = 0; v < 4; ++v) sc_cur[v] = sc_nxt[v];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // image (tile + 1) & 1 complete; everybody is done reading image tile & 1
    asm volatile("" ::: "memory");
  };
  for (uint32_t tile = t_lo; tile < t_hi; ++tile) one_tile(tile);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the prefetch that ran past my range)
  if constexpr (!SCORES) {
    for (uint32_t g = 0; g < EXACT_MFMA_QB; ++g) {
      const uint32_t q = qg0 + wave * EXACT_MFMA_QB + g;
      const uint32_t c = readlane_u(my_cnt, static_cast<int>(g));
      if (c == 0) continue;
      uint32_t slot0 = 0;
      if (lane == 0) slot0 = atomicAdd(&cnt[q], c);
      slot0 = readlane_u(slot0, 0);
      if (static_cast<uint32_t>(lane) < c) {
        const uint32_t slot = slot0 + lane;
        if (slot < cap) cand[static_cast<uint64_t>(q) * cap + slot] = Cand{l_s[SCORES ? 0 : wave][SCORES ? 0 : g][lane], l_id[SCORES ? 0 : wave][SCORES ? 0 : g][lane]};
        else overflow[q] = 1u;
      }
    }
  }
}

}  // namespace nvdbhip
