// kernels_refine.h -- exact squared-L2 rerank of R candidates per query (the stage that follows
// IVF-PQ candidate generation).  HIP replacement for the reference's CUDA kernels
// (src/cuda_refine.cu:405-502 and its warp-merge variants :505-838), designed for wave64:
//
//   * one workgroup (256 threads = 4 waves) per query, one lane per candidate row;
//   * the query is wave-uniform, so its elements come from SGPRs (scalar loads), the candidate row
//     is streamed with 16-byte loads;
//   * distance arithmetic follows the reference kernel's fp32 order exactly
//     (l2_fp16_base_half2, cuda_refine.cu:326-382: pair p -> accumulator p&3, two fmaf per pair,
//      (a0+a1)+(a2+a3); l2_fp32_base, :383-392: one accumulator, one fma per element);
//   * top-K is wavefront-resident (entry j in lane j, K <= 64) instead of the reference's
//     per-thread register lists + thread-0 serial merge, which its own profile shows to be 58 % of
//     the kernel (Performance_CUDA.md:267); the four waves' lists are merged through LDS.
//   * ties: (distance asc, id asc).  Output padded with id 0xFFFFFFFF / dist 1e30 (:248-252, :892-894).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_exact.h"
#include "kernels_filter.h"

namespace nvdbhip {

__device__ __forceinline__ bool closer(float d1, uint32_t i1, float d2, uint32_t i2) {
  return (d1 < d2) || (d1 == d2 && i1 < i2);
}

struct WaveTopKMin { float d; uint32_t id; uint32_t cnt; float thr_d; uint32_t thr_id; };

__device__ __forceinline__ void wmin_insert(WaveTopKMin& t, uint32_t K, float d, uint32_t id, int lane) {
  const bool b = (static_cast<uint32_t>(lane) < t.cnt) && closer(t.d, t.id, d, id);
  const uint32_t pos = static_cast<uint32_t>(__builtin_popcountll(__ballot(b)));
  if (pos >= K) return;
  const float up_d = __shfl_up(t.d, 1);
  const uint32_t up_id = __shfl_up(t.id, 1);
  if (static_cast<uint32_t>(lane) > pos) { t.d = up_d; t.id = up_id; }
  else if (static_cast<uint32_t>(lane) == pos) { t.d = d; t.id = id; }
  if (t.cnt < K) ++t.cnt;
  if (t.cnt == K) { t.thr_d = readlane_f(t.d, static_cast<int>(K) - 1); t.thr_id = readlane_u(t.id, static_cast<int>(K) - 1); }
}
__device__ __forceinline__ bool wmin_accepts(const WaveTopKMin& t, uint32_t K, float d, uint32_t id) {
  return t.cnt < K || closer(d, id, t.thr_d, t.thr_id);
}

// fp16 rows: cuda_refine.cu:326-382
template <bool ALIGNED>
__device__ __forceinline__ float l2_f16_ref_order(const unsigned short* __restrict__ x, const float* __restrict__ q, uint32_t dim) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const uint32_t d2 = dim >> 1;
  uint32_t p = 0;
#pragma unroll 2
  for (; p + 3 < d2; p += 4) {          // 4 pairs = 8 dims per step
    float xv[8];
    load8<DT_F16, ALIGNED>(x, 2 * p, xv);
    float dx, dy;
    dx = q[2 * p + 0] - xv[0]; dy = q[2 * p + 1] - xv[1]; a0 = __builtin_fmaf(dx, dx, a0); a0 = __builtin_fmaf(dy, dy, a0);
    dx = q[2 * p + 2] - xv[2]; dy = q[2 * p + 3] - xv[3]; a1 = __builtin_fmaf(dx, dx, a1); a1 = __builtin_fmaf(dy, dy, a1);
    dx = q[2 * p + 4] - xv[4]; dy = q[2 * p + 5] - xv[5]; a2 = __builtin_fmaf(dx, dx, a2); a2 = __builtin_fmaf(dy, dy, a2);
    dx = q[2 * p + 6] - xv[6]; dy = q[2 * p + 7] - xv[7]; a3 = __builtin_fmaf(dx, dx, a3); a3 = __builtin_fmaf(dy, dy, a3);
  }
  for (p = d2 & ~3u; p < d2; ++p) {     // leftover pairs all go to accumulator 0 (:368-376)
    const float dx = q[2 * p] - half_bits_to_float(x[2 * p]);
    const float dy = q[2 * p + 1] - half_bits_to_float(x[2 * p + 1]);
    a0 = __builtin_fmaf(dx, dx, a0); a0 = __builtin_fmaf(dy, dy, a0);
  }
  return (a0 + a1) + (a2 + a3);
}

// fp32 rows: cuda_refine.cu:383-392 (`acc += diff*diff`, contracted to FMA by nvcc's default)
template <bool ALIGNED>
__device__ __forceinline__ float l2_f32_ref_order(const float* __restrict__ x, const float* __restrict__ q, uint32_t dim) {
  float acc = 0.f;
  uint32_t j = 0;
  if constexpr (ALIGNED) {
#pragma unroll 2
    for (; j + 8 <= dim; j += 8) {
      float xv[8];
      load8<DT_F32, true>(x, j, xv);
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float d = q[j + u] - xv[u]; acc = __builtin_fmaf(d, d, acc); }
    }
  }
  for (; j < dim; ++j) { const float d = q[j] - x[j]; acc = __builtin_fmaf(d, d, acc); }
  return acc;
}

// grid = Q, block = 256.  out_dist may be null.
template <int DT, bool ALIGNED>
__global__ __launch_bounds__(256) void refine_l2_kernel(const void* __restrict__ rows, uint64_t n, uint32_t dim,
                                                        const float* __restrict__ queries, const uint32_t* __restrict__ cand,
                                                        uint32_t R, uint32_t K, uint32_t* __restrict__ out_ids,
                                                        float* __restrict__ out_dist) {
  __shared__ float lds_d[4][64];
  __shared__ uint32_t lds_id[4][64];
  __shared__ uint32_t lds_cnt[4];
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* __restrict__ qv = queries + static_cast<uint64_t>(q) * dim;
  const uint32_t* __restrict__ cq = cand + static_cast<uint64_t>(q) * R;

  WaveTopKMin tk;
  tk.d = 1e30f; tk.id = 0xFFFFFFFFu; tk.cnt = 0; tk.thr_d = 1e30f; tk.thr_id = 0xFFFFFFFFu;

  for (uint32_t r0 = wave * 64u; r0 < R; r0 += 256u) {
    const uint32_t r = r0 + lane;
    const uint32_t id = (r < R) ? cq[r] : 0xFFFFFFFFu;
    const bool valid = (id != 0xFFFFFFFFu) && (static_cast<uint64_t>(id) < n);     // cuda_refine.cu:437
    const uint32_t rid = valid ? id : 0u;
    float d;
    if constexpr (DT == DT_F16) d = l2_f16_ref_order<ALIGNED>(static_cast<const unsigned short*>(rows) + static_cast<uint64_t>(rid) * dim, qv, dim);
    else d = l2_f32_ref_order<ALIGNED>(static_cast<const float*>(rows) + static_cast<uint64_t>(rid) * dim, qv, dim);
    unsigned long long m = __ballot(valid && wmin_accepts(tk, K, d, id));
    while (m) {
      const int L = __builtin_ctzll(m);
      m &= m - 1;
      const float cd = readlane_f(d, L);
      const uint32_t cid = readlane_u(id, L);
      if (wmin_accepts(tk, K, cd, cid)) wmin_insert(tk, K, cd, cid, lane);
    }
  }
  lds_d[wave][lane] = tk.d; lds_id[wave][lane] = tk.id;
  if (lane == 0) lds_cnt[wave] = tk.cnt;
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < 4; ++w) {
    const uint32_t c = lds_cnt[w];
    for (uint32_t j = 0; j < c; ++j) {
      const float cd = lds_d[w][j];
      const uint32_t cid = lds_id[w][j];
      if (wmin_accepts(tk, K, cd, cid)) wmin_insert(tk, K, cd, cid, lane);
    }
  }
  if (static_cast<uint32_t>(lane) < K) {
    const bool have = static_cast<uint32_t>(lane) < tk.cnt;
    out_ids[static_cast<uint64_t>(q) * K + lane] = have ? tk.id : 0xFFFFFFFFu;
    if (out_dist) out_dist[static_cast<uint64_t>(q) * K + lane] = have ? tk.d : 1e30f;
  }
}


// ------------------------------------------------------------------------------------------------
// refine, version 2: coalesced gather through LDS.
//
// The kernel above lets every lane walk its own 1.5 KB row with 16-byte loads: correct, but each
// wave-instruction touches 64 different cache lines and the gather ran at 1.5 TB/s (19 % of HBM).
// Here a wave stages 256-byte column chunks of its 64 candidate rows in LDS with direct-to-LDS loads
// whose per-lane SOURCE address does the gather (16 lanes x 16 B = one whole 256-byte piece of one row
// per quarter-wave, full 128-byte lines), double-buffered per wave with counted vmcnt (no workgroup
// barrier: a wave only reads what it staged itself).  Each lane then reads its own row chunk from LDS
// (16 x ds_read_b128; chunk c of row i is stored at position c ^ (i & 15), so the 16 rows of a
// ds_read_b128 lane group fall into 16 distinct bank slots) and continues the reference's fp32
// accumulation order exactly where the previous chunk stopped -- results are bit-identical to v1.
// LDS: 4 waves x 2 buffers x 16 KB = 128 KB, one workgroup per CU, 64 KB of gathers in flight per CU.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256, 1) void refine_l2_lds_kernel(const void* __restrict__ rows, uint64_t n, uint32_t dim,
                                                               const float* __restrict__ queries, const uint32_t* __restrict__ cand,
                                                               uint32_t R, uint32_t K, uint32_t* __restrict__ out_ids,
                                                               float* __restrict__ out_dist) {
  constexpr int BPE = (DT == DT_F16) ? 2 : 4;
  constexpr int CH_BYTES = 256, CH_ELEMS = CH_BYTES / BPE, BUF_BYTES = 64 * CH_BYTES;   // 16 KB per wave buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float lds_d[4][64];
  __shared__ uint32_t lds_id[4][64];
  __shared__ uint32_t lds_cnt[4];
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* __restrict__ qv = queries + static_cast<uint64_t>(q) * dim;
  const uint32_t* __restrict__ cq = cand + static_cast<uint64_t>(q) * R;
  const uint32_t row_bytes = dim * BPE;
  const uint32_t nchunks = (row_bytes + CH_BYTES - 1) / CH_BYTES;
  char* mybuf = smem + wave * 2 * BUF_BYTES;
  const uint32_t lds_mine = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(mybuf)));
  const char* gbase = static_cast<const char*>(rows);

  WaveTopKMin tk;
  tk.d = 1e30f; tk.id = 0xFFFFFFFFu; tk.cnt = 0; tk.thr_d = 1e30f; tk.thr_id = 0xFFFFFFFFu;

  const uint32_t sub = lane >> 4, pos = lane & 15;           // piece p stages rows 4p..4p+3; this lane: row 4p+sub, slot pos
  for (uint32_t r0 = wave * 64u; r0 < R; r0 += 256u) {
    const uint32_t r = r0 + lane;
    const uint32_t id = (r < R) ? cq[r] : 0xFFFFFFFFu;
    const bool valid = (id != 0xFFFFFFFFu) && (static_cast<uint64_t>(id) < n);     // cuda_refine.cu:437
    const uint32_t rid = valid ? id : 0u;                    // invalid lanes gather row 0 and are dropped below
    // row ids this lane gathers for: rows 4p+sub, p = 0..15
    uint32_t src_row_off_lo[16], src_row_off_hi[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const uint32_t rr = static_cast<uint32_t>(__shfl(static_cast<int>(rid), 4 * p + static_cast<int>(sub)));
      const uint64_t off = static_cast<uint64_t>(rr) * row_bytes;
      src_row_off_lo[p] = static_cast<uint32_t>(off); src_row_off_hi[p] = static_cast<uint32_t>(off >> 32);
    }
    auto issue_chunk = [&](uint32_t c, uint32_t buf) {
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const uint32_t rowi = 4u * p + sub;
        uint32_t coff = c * CH_BYTES + ((pos ^ (rowi & 15u)) << 4);          // source chunk for LDS slot `pos`
        if (coff + 16 > row_bytes) coff = row_bytes - 16;                     // ragged last chunk: any in-row bytes (unused)
        const uint64_t off = ((static_cast<uint64_t>(src_row_off_hi[p]) << 32) | src_row_off_lo[p]) + coff;
        glds16_v(gbase + off, lds_mine + buf * BUF_BYTES + p * 1024);
      }
    };
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    issue_chunk(0, 0);
    for (uint32_t c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) { issue_chunk(c + 1, (c + 1) & 1u); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const char* rowc = mybuf + (c & 1u) * BUF_BYTES + lane * CH_BYTES;
      const uint32_t e0 = c * CH_ELEMS;
      const uint32_t ne = (dim - e0 < static_cast<uint32_t>(CH_ELEMS)) ? dim - e0 : static_cast<uint32_t>(CH_ELEMS);
      if constexpr (DT == DT_F16) {
        for (uint32_t j = 0; j * 8 < ne; ++j) {                              // 8 dims = 4 pairs per 16-byte slot
          const uint4 v = *reinterpret_cast<const uint4*>(rowc + ((j ^ (static_cast<uint32_t>(lane) & 15u)) << 4));
          const float* qq = qv + e0 + 8 * j;
          float dx, dy;
          dx = qq[0] - half_bits_to_float(v.x & 0xFFFFu); dy = qq[1] - half_bits_to_float(v.x >> 16); a0 = __builtin_fmaf(dx, dx, a0); a0 = __builtin_fmaf(dy, dy, a0);
          dx = qq[2] - half_bits_to_float(v.y & 0xFFFFu); dy = qq[3] - half_bits_to_float(v.y >> 16); a1 = __builtin_fmaf(dx, dx, a1); a1 = __builtin_fmaf(dy, dy, a1);
          dx = qq[4] - half_bits_to_float(v.z & 0xFFFFu); dy = qq[5] - half_bits_to_float(v.z >> 16); a2 = __builtin_fmaf(dx, dx, a2); a2 = __builtin_fmaf(dy, dy, a2);
          dx = qq[6] - half_bits_to_float(v.w & 0xFFFFu); dy = qq[7] - half_bits_to_float(v.w >> 16); a3 = __builtin_fmaf(dx, dx, a3); a3 = __builtin_fmaf(dy, dy, a3);
        }
      } else {
        for (uint32_t j = 0; j * 4 < ne; ++j) {                              // single accumulator, one fma per element
          const float4 v = *reinterpret_cast<const float4*>(rowc + ((j ^ (static_cast<uint32_t>(lane) & 15u)) << 4));
          const float* qq = qv + e0 + 4 * j;
          float dd;
          dd = qq[0] - v.x; a0 = __builtin_fmaf(dd, dd, a0);
          dd = qq[1] - v.y; a0 = __builtin_fmaf(dd, dd, a0);
          dd = qq[2] - v.z; a0 = __builtin_fmaf(dd, dd, a0);
          dd = qq[3] - v.w; a0 = __builtin_fmaf(dd, dd, a0);
        }
      }
    }
    const float d = (DT == DT_F16) ? (a0 + a1) + (a2 + a3) : a0;
    unsigned long long m = __ballot(valid && wmin_accepts(tk, K, d, id));
    while (m) {
      const int L = __builtin_ctzll(m);
      m &= m - 1;
      const float cd = readlane_f(d, L);
      const uint32_t cid = readlane_u(id, L);
      if (wmin_accepts(tk, K, cd, cid)) wmin_insert(tk, K, cd, cid, lane);
    }
  }
  lds_d[wave][lane] = tk.d; lds_id[wave][lane] = tk.id;
  if (lane == 0) lds_cnt[wave] = tk.cnt;
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < 4; ++w) {
    const uint32_t c = lds_cnt[w];
    for (uint32_t j = 0; j < c; ++j) {
      const float cd = lds_d[w][j];
      const uint32_t cid = lds_id[w][j];
      if (wmin_accepts(tk, K, cd, cid)) wmin_insert(tk, K, cd, cid, lane);
    }
  }
  if (static_cast<uint32_t>(lane) < K) {
    const bool have = static_cast<uint32_t>(lane) < tk.cnt;
    out_ids[static_cast<uint64_t>(q) * K + lane] = have ? tk.id : 0xFFFFFFFFu;
    if (out_dist) out_dist[static_cast<uint64_t>(q) * K + lane] = have ? tk.d : 1e30f;
  }
}

}  // namespace nvdbhip

namespace nvdbhip {

// ------------------------------------------------------------------------------------------------
// refine, version 3 (fp16 rows, dim in {256, 384, 512, 768}): WHOLE rows per request, four lanes per row.
//
// v2 above fetches a row as six 256-byte pieces that are microseconds apart: every piece re-opens the row's DRAM page,
// and the gather saturated at 4.3 TB/s with FETCH_SIZE == algorithmic bytes (profiles/r02_refine_v2_*.txt) -- the
// memory system was the limit, not the kernel's issue rate.  Here a wave asks for 16 whole rows back to back
// (per row one 1-KB direct-to-LDS load with a SCALAR row base + one 512-byte remainder shared with a second row),
// the access shape MI355X_MICROARCH.md measures at 5.5-5.8 TB/s for random 1-2 KB rows.
//
//   * workgroup = 3 waves = one query; two workgroups per CU (2 x 3 x 24 KB of row slots), no workgroup barrier in the
//     loop: a wave waits only for its own loads (s_waitcnt vmcnt) -- while it computes, the other five keep
//     ~120 KB per CU in flight.
//   * lane (r, c) = lane/4, lane%4 handles row r of the step and accumulator c of the reference kernel
//     (cuda_refine.cu:326-382: half2 pair p feeds accumulator p & 3, dx then dy, pairs in ascending order):
//     it reads pair 4i + c of every 16-byte piece i with ds_read2_b32 at STATIC offsets and keeps its own 2 x dim/8
//     query elements in registers for the whole query.  dx = q - float(x) is ONE v_fma_mix_f32 (x read as the low /
//     high half of the packed register, times -1.0, plus q: a single rounding, the same bits as the subtraction).
//   * LDS image: row r's first KB (or the whole shorter row) in a block padded by 16 bytes (the 8 rows of a half-wave hit
//     8 x 4 distinct banks), remainders of rows j and j + 8 in one KB at j * 1040 (read by different half-waves).
//   * d = (a0 + a1) + (a2 + a3) by two quad shuffles, then the same wavefront-resident top-K as v1/v2.
// Results are bit-identical to v1/v2 and to the oracle's restatement (tests + tools_dev/fuzz_refine.py).
// ------------------------------------------------------------------------------------------------
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// 16 bytes per lane HBM -> LDS, global address = sbase + voff + IMM, LDS address = lds_off (wave-uniform) + lane * 16
template <int IMM>
__device__ __forceinline__ void glds16_imm(uint32_t voff, const void* sbase, uint32_t lds_off) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:%4\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_off), "s"(sbase), "n"(IMM) : "memory");
}

// q - float(half) in ONE instruction: v_fma_mix_f32 reads the low / high half of the packed register as fp16, multiplies
// by -1.0 (exact) and adds q with a single rounding -- the same bits as cvt + v_sub_f32 (hipcc folds the C++ form of
// this fma back into cvt + sub, hence the asm).
__device__ __forceinline__ float q_minus_half_lo(uint32_t xpk, float q) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(xpk), "v"(q));
  return r;
}
__device__ __forceinline__ float q_minus_half_hi(uint32_t xpk, float q) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(xpk), "v"(q));
  return r;
}

constexpr int REFINE3_WAVES = 3, REFINE3_ROWS = 16;
// LDS image of a wave's 16 rows: piece A = the row's first min(row, 1 KB) bytes (LA lanes x 16 B) in a block of
// LA * 16 + 16 bytes (the 16 bytes of padding put the 8 rows of a half-wave on 8 x 4 distinct banks), then -- for 1536-byte
// rows -- the 512-byte remainders of rows j and j + 8 share one 1040-byte block.
template <int DIM> constexpr int refine3_la() { return DIM * 2 >= 1024 ? 64 : DIM * 2 / 16; }
template <int DIM> constexpr int refine3_slot_bytes() {
  constexpr int RB = DIM * 2, LA = refine3_la<DIM>(), REM = RB - LA * 16;
  return REFINE3_ROWS * (LA * 16 + 16) + (REM ? (REFINE3_ROWS / 2) * 1040 : 0);
}

template <int DIM>
__global__ __launch_bounds__(64 * REFINE3_WAVES) void refine_l2_rows_kernel(const void* __restrict__ rows, uint64_t n, const float* __restrict__ queries,
                                                                            const uint32_t* __restrict__ cand, uint32_t R, uint32_t K,
                                                                            uint32_t* __restrict__ out_ids, float* __restrict__ out_dist) {
  constexpr int RB = DIM * 2;                       // row bytes
  constexpr int LA = refine3_la<DIM>();             // lanes of piece A (16 bytes each): the whole row up to 1 KB
  constexpr int REM = RB - LA * 16;                 // 0, or 512 for 1536-byte rows: remainder, two rows per piece
  static_assert(DIM % 8 == 0 && (REM == 0 || REM == 512), "rows of up to 1 KB, or of 1536 bytes");
  constexpr int NPAIR = DIM / 8;                    // pairs per lane = 16-byte pieces per row
  constexpr int SLOT = refine3_slot_bytes<DIM>();
  constexpr int ABLOCK = LA * 16 + 16, BBLOCK = 1040;
  constexpr int BOFF = REFINE3_ROWS * ABLOCK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float lds_d[REFINE3_WAVES][64];
  __shared__ uint32_t lds_id[REFINE3_WAVES][64];
  __shared__ uint32_t lds_cnt[REFINE3_WAVES];
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane >> 2, c = lane & 3;
  const uint32_t* __restrict__ cq = cand + static_cast<uint64_t>(q) * R;
  char* myslot = smem + wave * SLOT;
  const uint32_t lds_mine = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(myslot)));
  const char* gbase = static_cast<const char*>(rows);

  // this lane's query elements: pairs c, c + 4, c + 8, ... (two floats each), resident for the whole query
  float2 qv[NPAIR];
  {
    const float2* qp = reinterpret_cast<const float2*>(queries + static_cast<uint64_t>(q) * DIM) + c;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) qv[i] = qp[4 * i];
  }

  WaveTopKMin tk;
  tk.d = 1e30f; tk.id = 0xFFFFFFFFu; tk.cnt = 0; tk.thr_d = 1e30f; tk.thr_id = 0xFFFFFFFFu;

  const uint32_t voffA = static_cast<uint32_t>(lane) * 16u;                                 // piece A: lane l <- bytes [16 l, 16 l + 16) of the row
  const uint32_t lane_lo = static_cast<uint32_t>(lane & 31) * 16u + LA * 16u;               // piece B: 32 lanes per row, after the A part
  const char* rdA = myslot + r * ABLOCK + c * 4;                                            // this lane's read bases
  const char* rdB = myslot + BOFF + (r & 7) * BBLOCK + (r >> 3) * 512 + c * 4;

  uint32_t idx = wave * REFINE3_ROWS + (lane & 15);
  uint32_t ids_next = (idx < R) ? cq[idx] : 0xFFFFFFFFu;
  for (uint32_t s0 = wave * REFINE3_ROWS; s0 < R; s0 += REFINE3_WAVES * REFINE3_ROWS) {
    const uint32_t ids = ids_next;                  // lanes 0..15 (and their copies in 16..63): candidate of row lane & 15
    idx += REFINE3_WAVES * REFINE3_ROWS;
    ids_next = (idx < R) ? cq[idx] : 0xFFFFFFFFu;   // next step's ids travel while this step's rows do
    // ---- issue: 16 whole rows (rows j and j + 8 together: they share the remainder piece) ----
#pragma unroll
    for (int j = 0; j < REFINE3_ROWS / 2; ++j) {
      const uint32_t sid0 = readlane_u(ids, j), sid1 = readlane_u(ids, j + 8);
      const bool ok0 = (sid0 != 0xFFFFFFFFu) && (static_cast<uint64_t>(sid0) < n);         // cuda_refine.cu:437
      const bool ok1 = (sid1 != 0xFFFFFFFFu) && (static_cast<uint64_t>(sid1) < n);
      const uint64_t off0 = static_cast<uint64_t>(ok0 ? sid0 : 0u) * RB;                   // skipped candidates: row 0, dropped below
      const uint64_t off1 = static_cast<uint64_t>(ok1 ? sid1 : 0u) * RB;
      if (lane < LA) {                               // (all 64 lanes for rows of 1 KB and more); ok0 / ok1: wave-uniform branches
        if (ok0) glds16_imm<0>(voffA, gbase + off0, lds_mine + j * ABLOCK);
        if (ok1) glds16_imm<0>(voffA, gbase + off1, lds_mine + (j + 8) * ABLOCK);
      }
      if constexpr (REM != 0) {
        const uint64_t o = (lane < 32 ? off0 : off1) + lane_lo;                            // the two rows are read by different half-waves
        if (ok0 || ok1) glds16_v(gbase + o, lds_mine + BOFF + j * BBLOCK);
      }
    }
    const uint32_t my_id = static_cast<uint32_t>(__shfl(static_cast<int>(ids), r));
    const bool valid = (my_id != 0xFFFFFFFFu) && (static_cast<uint64_t>(my_id) < n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- consume: accumulator c of row r ----
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const char* src = (i < LA) ? rdA + i * 16 : rdB + (i - LA) * 16;
      const uint32_t x = *reinterpret_cast<const uint32_t*>(src);                          // half2 pair (x[2p], x[2p+1])
      const float dx = q_minus_half_lo(x, qv[i].x);
      const float dy = q_minus_half_hi(x, qv[i].y);
      acc = __builtin_fmaf(dx, dx, acc);
      acc = __builtin_fmaf(dy, dy, acc);
    }
    const float s1 = acc + __shfl_xor(acc, 1);      // a0 + a1 | a2 + a3
    const float d = s1 + __shfl_xor(s1, 2);         // (a0 + a1) + (a2 + a3): the same bits in all four lanes
    unsigned long long m = __ballot(valid && c == 0 && wmin_accepts(tk, K, d, my_id));
    while (m) {
      const int L = __builtin_ctzll(m);
      m &= m - 1;
      const float cd = readlane_f(d, L);
      const uint32_t cid = readlane_u(my_id, L);
      if (wmin_accepts(tk, K, cd, cid)) wmin_insert(tk, K, cd, cid, lane);
    }
  }
  lds_d[wave][lane] = tk.d; lds_id[wave][lane] = tk.id;
  if (lane == 0) lds_cnt[wave] = tk.cnt;
  __syncthreads();
  if (wave != 0) return;
  for (int w = 1; w < REFINE3_WAVES; ++w) {
    const uint32_t cn = lds_cnt[w];
    for (uint32_t j = 0; j < cn; ++j) {
      const float cd = lds_d[w][j];
      const uint32_t cid = lds_id[w][j];
      if (wmin_accepts(tk, K, cd, cid)) wmin_insert(tk, K, cd, cid, lane);
    }
  }
  if (static_cast<uint32_t>(lane) < K) {
    const bool have = static_cast<uint32_t>(lane) < tk.cnt;
    out_ids[static_cast<uint64_t>(q) * K + lane] = have ? tk.id : 0xFFFFFFFFu;
    if (out_dist) out_dist[static_cast<uint64_t>(q) * K + lane] = have ? tk.d : 1e30f;
  }
}

// Long fp16 rows (d = 1024 / 1536: 2 KB / 3 KB) stay on refine_l2_lds_kernel (v2).  A whole-row build of this kernel for them was
// measured in round 3 (query in LDS, 8 or 16 rows per wave and step): 4.28 TB/s at d = 1024 and 3.77 TB/s at d = 1536 against v2's
// 4.69 / 4.86 TB/s (Q = 10000, R = 1024, N = 1.5M; gpurun_out/r03e_refine.log, profiles/r03_refine_long_rows.txt).  With four lanes
// per row -- the reference's four accumulator chains cannot be split further -- a 3 KB row costs one lane 192 dependent pair
// steps while half the wave idles (8 rows x 3 KB fill the same 24 KB slot as 16 x 1.5 KB), and a 16-row slot no longer leaves
// room for six waves per CU; v2's lane-per-row layout keeps all 64 lanes busy on 64 rows.
// Two re-shapings of v2 itself were measured too and dropped (profiles/r03_refine_v2_variants.txt): 512-byte chunks on two waves
// (half as many, longer requests: 5.9 / 8.7 ms against 4.4 / 6.5 at d = 1024 / 1536) and three buffers on three waves (two chunks
// ahead: 5.1 / 7.4 ms) -- what v2 needs is its four waves, not longer or deeper requests.
}  // namespace nvdbhip
