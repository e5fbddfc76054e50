// kernels_exact.h -- gfx950 kernels that reproduce the reference's CPU arithmetic bit for bit.
//
//  * exact_scores<>      : one lane = one corpus row; the 8-lane stride-8 FMA order of
//                          dot_avx2_fma / dot_f32_f16base_avx2 / dot_f32_i8_avx2
//                          (reference src/simd_dot.cpp:26-49, 102-124, 160-199) restated with
//                          explicit fmaf chains, queries broadcast from SGPRs.
//  * scan_exact_kernel   : streaming scan with a wavefront-resident top-k (entry j lives in lane j)
//                          -- the always-correct path (small batches, any dim, overflow fallback,
//                          threshold bootstrap of the MFMA filter).
//  * select_kernel       : per-query bitonic sort of the candidate list in LDS, threshold update
//                          and compaction / final top-k emission, order (score desc, id asc).
//  * rescore_kernel      : exact scores for the survivors of the MFMA filter.
//  * merge_topk_kernel   : k-way merge of per-shard lists (multi-GPU).
//
// This translation unit is compiled with -ffp-contract=off: the only fused operations are the
// explicit __builtin_fmaf calls.
#pragma once
#include <hip/hip_runtime.h>

#include "nvdb_common.h"

namespace nvdbhip {

struct Cand { float score; uint32_t row; };   // 8 bytes, one candidate (query implied by the list)

constexpr int DT_F32 = 1, DT_F16 = 2, DT_I8 = 3;
constexpr float NEG_INF = -__builtin_huge_valf();

__device__ __forceinline__ bool better(float s1, uint32_t i1, float s2, uint32_t i2) {
  return (s1 > s2) || (s1 == s2 && i1 < i2);
}
__device__ __forceinline__ float half_bits_to_float(uint32_t h16) {
  return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(h16)));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
}
__device__ __forceinline__ float hsum8(const float (&a)[8]) {
  // (lo+hi), hadd, hadd  (simd_dot.cpp:38-44)
  const float s0 = a[0] + a[4], s1 = a[1] + a[5], s2 = a[2] + a[6], s3 = a[3] + a[7];
  return (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------------------------------------
// exact scores of ONE corpus row against QG queries (q is wave-uniform -> scalar loads)
// ------------------------------------------------------------------------------------------------
template <int DT, bool ALIGNED>
__device__ __forceinline__ void load8(const void* __restrict__ rowp, uint32_t i, float (&x)[8]) {
  if constexpr (DT == DT_F16) {
    if constexpr (ALIGNED) {
      const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const char*>(rowp) + 2 * i);
      x[0] = half_bits_to_float(v.x & 0xFFFFu); x[1] = half_bits_to_float(v.x >> 16);
      x[2] = half_bits_to_float(v.y & 0xFFFFu); x[3] = half_bits_to_float(v.y >> 16);
      x[4] = half_bits_to_float(v.z & 0xFFFFu); x[5] = half_bits_to_float(v.z >> 16);
      x[6] = half_bits_to_float(v.w & 0xFFFFu); x[7] = half_bits_to_float(v.w >> 16);
    } else {
      const unsigned short* p = static_cast<const unsigned short*>(rowp) + i;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = half_bits_to_float(p[j]);
    }
  } else if constexpr (DT == DT_F32) {
    if constexpr (ALIGNED) {
      const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(rowp) + i);
      const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(rowp) + i + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
      const float* p = static_cast<const float*>(rowp) + i;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = p[j];
    }
  } else {
    if constexpr (ALIGNED) {   // 8-byte aligned groups of 8 int8
      const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const char*>(rowp) + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = static_cast<float>(static_cast<int>(v.x << (24 - 8 * j)) >> 24);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 + j] = static_cast<float>(static_cast<int>(v.y << (24 - 8 * j)) >> 24);
    } else {
      const signed char* p = static_cast<const signed char*>(rowp) + i;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = static_cast<float>(p[j]);
    }
  }
}

template <int DT>
__device__ __forceinline__ float load1(const void* __restrict__ rowp, uint32_t i) {
  if constexpr (DT == DT_F16) return half_bits_to_float(static_cast<const unsigned short*>(rowp)[i]);
  else if constexpr (DT == DT_F32) return static_cast<const float*>(rowp)[i];
  else return static_cast<float>(static_cast<const signed char*>(rowp)[i]);
}

// out[g] = reference score of (query g, this row).  q points at query 0 of the group, queries are
// `qstride` floats apart (wave-uniform addresses: global memory -> scalar loads, LDS -> broadcast
// ds_read_b128).  `scale` is the row's int8 scale (ignored otherwise).
template <int DT, int QG, bool ALIGNED, typename QPtr>
__device__ __forceinline__ void exact_scores(const void* __restrict__ rowp, QPtr q, uint32_t qstride,
                                             uint32_t dim, float scale, float (&out)[QG]) {
  float acc[QG][8];
#pragma unroll
  for (int g = 0; g < QG; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;

  // vector body: f32/f16 consume floor(dim/8)*8 elements (simd_dot.cpp:31, 106), int8 consumes
  // floor(dim/16)*16 (simd_dot.cpp:164) -- two groups of 8 into the same accumulators.
  const uint32_t body = (DT == DT_I8) ? (dim & ~15u) : (dim & ~7u);
  uint32_t i = 0;
#pragma unroll 2
  for (; i < body; i += 8) {
    float x[8];
    load8<DT, ALIGNED>(rowp, i, x);
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      // 8 consecutive query elements; 16-byte aligned whenever qstride % 4 == 0 (the staging pads it)
      const float4 qa = *reinterpret_cast<const float4*>(&q[g * qstride + i]);
      const float4 qb = *reinterpret_cast<const float4*>(&q[g * qstride + i + 4]);
      acc[g][0] = __builtin_fmaf(qa.x, x[0], acc[g][0]); acc[g][1] = __builtin_fmaf(qa.y, x[1], acc[g][1]);
      acc[g][2] = __builtin_fmaf(qa.z, x[2], acc[g][2]); acc[g][3] = __builtin_fmaf(qa.w, x[3], acc[g][3]);
      acc[g][4] = __builtin_fmaf(qb.x, x[4], acc[g][4]); acc[g][5] = __builtin_fmaf(qb.y, x[5], acc[g][5]);
      acc[g][6] = __builtin_fmaf(qb.z, x[6], acc[g][6]); acc[g][7] = __builtin_fmaf(qb.w, x[7], acc[g][7]);
    }
  }
#pragma unroll
  for (int g = 0; g < QG; ++g) out[g] = hsum8(acc[g]);

  // tails (see oracle/nvdb_oracle.c oracle_dot_f32 for the f32 rule: a group of four unfused
  // multiply+add, then fused; f16 and int8 tails are fused throughout)
  if (i < dim) {
    if constexpr (DT == DT_F32) {
      if (dim - i >= 4) {
        for (int j = 0; j < 4; ++j) {
          const float x = load1<DT>(rowp, i + j);
#pragma unroll
          for (int g = 0; g < QG; ++g) { const float p = q[g * qstride + i + j] * x; out[g] = out[g] + p; }
        }
        i += 4;
      }
    }
    for (; i < dim; ++i) {
      const float x = load1<DT>(rowp, i);
#pragma unroll
      for (int g = 0; g < QG; ++g) out[g] = __builtin_fmaf(q[g * qstride + i], x, out[g]);
    }
  }
  if constexpr (DT == DT_I8) {
#pragma unroll
    for (int g = 0; g < QG; ++g) out[g] = out[g] * scale;   // simd_dot.cpp:198
  }
}

template <int DT>
__device__ __forceinline__ const void* row_ptr(const void* rows, uint64_t row, uint32_t dim) {
  constexpr int BPE = (DT == DT_F32) ? 4 : (DT == DT_F16 ? 2 : 1);
  return static_cast<const char*>(rows) + row * static_cast<uint64_t>(dim) * BPE;
}

// ------------------------------------------------------------------------------------------------
// wavefront-resident top-k: entry j (best first) lives in lane j; k <= 64
// ------------------------------------------------------------------------------------------------
struct WaveTopK {
  float s;        // per lane
  uint32_t id;    // per lane
  uint32_t cnt;   // uniform
  float thr_s;    // uniform: worst kept entry once cnt == k
  uint32_t thr_id;
};

__device__ __forceinline__ void wtk_init(WaveTopK& t) {
  t.s = NEG_INF; t.id = 0xFFFFFFFFu; t.cnt = 0; t.thr_s = NEG_INF; t.thr_id = 0xFFFFFFFFu;
}
__device__ __forceinline__ bool wtk_accepts(const WaveTopK& t, uint32_t k, float s, uint32_t id) {
  return t.cnt < k || better(s, id, t.thr_s, t.thr_id);
}
// all lanes call with the same (s,id)
__device__ __forceinline__ void wtk_insert(WaveTopK& t, uint32_t k, float s, uint32_t id, int lane) {
  const bool b = (static_cast<uint32_t>(lane) < t.cnt) && better(t.s, t.id, s, id);
  const uint32_t pos = static_cast<uint32_t>(__builtin_popcountll(__ballot(b)));
  if (pos >= k) return;
  const float up_s = __shfl_up(t.s, 1);
  const uint32_t up_id = __shfl_up(t.id, 1);
  if (static_cast<uint32_t>(lane) > pos) { t.s = up_s; t.id = up_id; }
  else if (static_cast<uint32_t>(lane) == pos) { t.s = s; t.id = id; }
  if (t.cnt < k) ++t.cnt;
  if (t.cnt == k) { t.thr_s = readlane_f(t.s, static_cast<int>(k) - 1); t.thr_id = readlane_u(t.id, static_cast<int>(k) - 1); }
}
// offer each lane's (s,id) where `pass` is set
__device__ __forceinline__ void wtk_offer(WaveTopK& t, uint32_t k, bool pass, float s, uint32_t id, int lane) {
  unsigned long long m = __ballot(pass);
  while (m) {
    const int L = __builtin_ctzll(m);
    m &= m - 1;
    const float cs = readlane_f(s, L);
    const uint32_t cid = readlane_u(id, L);
    if (wtk_accepts(t, k, cs, cid)) wtk_insert(t, k, cs, cid, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// exact streaming scan.  grid = (row splits P, ceil(nq/QG)); block = 256 (4 waves).
// Each workgroup scans rows [lo,hi) of the chunk for QG queries and appends its <= k best
// entries per query (those that also clear the query's global threshold) to the candidate lists.
// The QG queries are staged once per workgroup in LDS (dynamic, QG * qstride floats, qstride = dim
// rounded up to 4) and read back with broadcast ds_read_b128 -- scalar loads of 64 query elements
// per step made the compiler spill SGPRs and spend ~150 SALU instructions per 64 FMAs.
// ------------------------------------------------------------------------------------------------
template <int DT, int QG, bool ALIGNED>
__global__ __launch_bounds__(256) void scan_exact_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t dim, uint32_t row_lo, uint32_t row_hi,
    const float* __restrict__ q32, uint32_t nq, uint32_t q_first, uint32_t k, const float* __restrict__ thr,
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t qstride = (dim + 3u) & ~3u;
  float* q_lds = reinterpret_cast<float*>(smem_raw);                                   // [QG][qstride]
  float (*lds_s)[QG][64] = reinterpret_cast<float (*)[QG][64]>(q_lds + QG * qstride);  // [4][QG][64]
  uint32_t (*lds_id)[QG][64] = reinterpret_cast<uint32_t (*)[QG][64]>(reinterpret_cast<float*>(lds_s) + 4 * QG * 64);
  uint32_t (*lds_cnt)[QG] = reinterpret_cast<uint32_t (*)[QG]>(reinterpret_cast<uint32_t*>(lds_id) + 4 * QG * 64);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t P = gridDim.x, p = blockIdx.x;
  const uint64_t span = static_cast<uint64_t>(row_hi - row_lo);
  const uint32_t lo = row_lo + static_cast<uint32_t>(span * p / P);
  const uint32_t hi = row_lo + static_cast<uint32_t>(span * (p + 1) / P);
  const uint32_t qg0 = q_first + blockIdx.y * QG;           // first query of this group (uniform)

  // queries beyond nq: point at the last valid query and drop the result
  const uint32_t qbase = (qg0 + QG <= nq) ? qg0 : (nq >= static_cast<uint32_t>(QG) ? nq - QG : 0u);
  for (uint32_t e = threadIdx.x; e < QG * qstride; e += 256) {
    const uint32_t g = e / qstride, j = e % qstride;
    q_lds[e] = (j < dim) ? q32[static_cast<uint64_t>(qbase + g) * dim + j] : 0.f;
  }
  __syncthreads();
  const float* qptr = q_lds;

  WaveTopK tk[QG];
  float gthr[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    wtk_init(tk[g]);
    const uint32_t qi = qbase + g;
    gthr[g] = (thr != nullptr && qi < nq) ? thr[qi] : NEG_INF;
  }

  for (uint32_t base = lo + wave * 64u; base < hi; base += 256u) {
    const uint32_t row = base + lane;
    const bool valid = row < hi;
    const uint32_t rrow = valid ? row : (hi - 1);
    float sc[QG];
    const float scale = (DT == DT_I8) ? scales[rrow] : 1.f;
    exact_scores<DT, QG, ALIGNED>(row_ptr<DT>(rows, rrow, dim), qptr, qstride, dim, scale, sc);
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      const bool pass = valid && sc[g] >= gthr[g] && wtk_accepts(tk[g], k, sc[g], row);
      wtk_offer(tk[g], k, pass, sc[g], row, lane);
    }
  }

#pragma unroll
  for (int g = 0; g < QG; ++g) {
    lds_s[wave][g][lane] = tk[g].s;
    lds_id[wave][g][lane] = tk[g].id;
    if (lane == 0) lds_cnt[wave][g] = tk[g].cnt;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    for (int w = 1; w < 4; ++w) {
      const uint32_t c = lds_cnt[w][g];
      for (uint32_t j = 0; j < c; ++j) {
        const float cs = lds_s[w][g][j];
        const uint32_t cid = lds_id[w][g][j];
        if (wtk_accepts(tk[g], k, cs, cid)) wtk_insert(tk[g], k, cs, cid, lane);
      }
    }
    const uint32_t qi = qbase + g;
    // groups that were shifted back to stay in range own only the queries >= qg0
    const bool own = qi >= qg0 && qi < nq;
    if (own && tk[g].cnt > 0) {
      uint32_t slot0 = 0;
      if (lane == 0) slot0 = atomicAdd(&cnt[qi], tk[g].cnt);
      slot0 = readlane_u(slot0, 0);
      if (static_cast<uint32_t>(lane) < tk[g].cnt) {
        const uint32_t slot = slot0 + lane;
        if (slot < cap) cand[static_cast<uint64_t>(qi) * cap + slot] = Cand{tk[g].s, tk[g].id};
        else overflow[qi] = 1u;
      }
    }
  }
}

// Everything this search's self-checks found goes into the STICKY words any_overflow[6..8] (= misc[12..14]), which survive later
// searches until nvdb_hip_search_check reads and clears them.  any_overflow[-6], [-5] = misc[0], misc[1]: bound violations and
// wave-log overflow -- final only when every kernel (and, in the fused rescore + select kernel, every workgroup) that can raise
// them has finished: one thread of the search's last workgroup calls this.
__device__ __forceinline__ void fold_sticky_words(uint32_t* any_overflow) {
  const uint32_t viol = __hip_atomic_load(any_overflow - 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t logovf = __hip_atomic_load(any_overflow - 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (viol) atomicAdd(any_overflow + 7, viol);
  if (logovf) atomicOr(any_overflow + 8, 1u);
}

// ------------------------------------------------------------------------------------------------
// select: sort a query's candidate list (score desc, id asc) in LDS, then either
//   mode 0: thr[q] = (k-th score) - slack[q]; keep every entry with score >= thr (all of the top-k
//           plus whatever lies inside the filter's error band); write the list back compacted;
//   mode 1: emit the final top-k (global ids, padded with UINT64_MAX / -inf).
//   mode 2: like mode 0 but the list is emptied afterwards (thresholds from bootstrap tile maxima).
// grid = nq, block = 64..256 (any multiple of 64), dynamic LDS = cap * 8 bytes.
// ------------------------------------------------------------------------------------------------
// The body works on ONE query with all `nth` threads of the calling workgroup and `e` = LDS for the list (cap rounded up to a
// power of two entries): select_kernel calls it for blockIdx.x, and rescore_lds_kernel, which folds the final select into its
// own tail, calls it too -- one implementation, one order.  Every thread of the workgroup must make the call.
__device__ __forceinline__ void select_body(
    const uint32_t q, const uint32_t tid, const uint32_t nth, Cand* e,
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t k, const float* __restrict__ slack,
    float* __restrict__ thr, uint32_t* __restrict__ overflow, int mode, uint64_t row_base,
    unsigned long long* __restrict__ out_ids, float* __restrict__ out_scores, uint32_t out_k, uint32_t* __restrict__ any_overflow,
    float* __restrict__ xcdw, bool fold_sticky) {
  __shared__ uint32_t s_keep;
  // XCD balance (kernels_filter.h, ScatterArgs::xcdw): the filter launch before this kernel filed tile-loop time and tile
  // counts per XCD label; turn them into the relative speeds the next launch partitions its tiles by.  Half-way steps,
  // a +-10 % cage and renormalisation to mean 1 keep one odd launch from skewing the shares.
  if (xcdw != nullptr && q == 0 && tid == 0) {
    float sp[8], mean = 0.f;
    bool all = true;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const float us = xcdw[8 + x], tiles = xcdw[16 + x];
      all = all && us > 0.f && tiles > 0.f;
      sp[x] = tiles / fmaxf(us, 1e-3f);
      mean += sp[x] * 0.125f;
    }
    if (all) {
      float w[8], wsum = 0.f;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        w[x] = fminf(fmaxf(0.5f * xcdw[x] + 0.5f * sp[x] / mean, 0.9f), 1.1f);     // tiles per microsecond is the XCD's own speed, whatever share it ran with
        wsum += w[x];
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) xcdw[x] = w[x] * (8.f / wsum);
    }
#pragma unroll
    for (int x = 8; x < 24; ++x) xcdw[x] = 0.f;
  }
  uint32_t m = (mode == 2) ? out_k : cnt[q];     // mode 2 (bootstrap): every list holds exactly out_k tile maxima
  if (m > cap) { if (tid == 0) overflow[q] = 1u; m = cap; }
  // final select: fold the per-query flags (list overflow, non-finite query) into one word the host can read alone,
  // and everything this search's self-checks found into the STICKY words any_overflow[6..8] (= misc[12..14]), which
  // survive later searches until nvdb_hip_search_check reads and clears them -- a caller that enqueues many searches
  // and checks once still learns about a failure in any of them.  (any_overflow[-6], [-5] = misc[0], misc[1]: bound
  // violations and wave-log overflow, final here because every earlier kernel of this search has completed.)
  // mode 3 = mode 1 without the sticky fold: the host API judges its own search by misc[0], [1], [6] alone and must
  // neither set nor clear what device-API searches left for the next nvdb_hip_search_check.
  if ((mode & 1) && tid == 0) {
    if (overflow[q]) { atomicOr(any_overflow, 1u); if (mode == 1) atomicOr(any_overflow + 6, 1u); }
    if (mode == 1 && fold_sticky) fold_sticky_words(any_overflow);
  }
  Cand* mine = cand + static_cast<uint64_t>(q) * cap;
  if (m <= 512) {
    // short list (the usual case): rank every entry against all others with broadcast LDS reads -- one pass,
    // two barriers, instead of the O(log^2 m) barrier stages of the bitonic network below.  Entries are
    // distinct (a row occurs once per list), so ranks are a permutation = the sorted positions.
    __shared__ float s_kth;
    if (tid == 0) { s_keep = 0; s_kth = NEG_INF; }
    for (uint32_t i = tid; i < m; i += nth) e[i] = mine[i];
    __syncthreads();
    Cand my[2];
    uint32_t rk[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t i = tid + u * nth;
      if (i < m) {
        my[u] = e[i];
        uint32_t r = 0;
        uint32_t j = 0;
        for (; j + 8 <= m; j += 8) {              // 8 independent broadcast reads in flight
          Cand o[8];
#pragma unroll
          for (int v = 0; v < 8; ++v) o[v] = e[j + v];
#pragma unroll
          for (int v = 0; v < 8; ++v) r += better(o[v].score, o[v].row, my[u].score, my[u].row) ? 1u : 0u;
        }
        for (; j < m; ++j) { const Cand o = e[j]; r += better(o.score, o.row, my[u].score, my[u].row) ? 1u : 0u; }
        rk[u] = r;
        if (r == k - 1) s_kth = my[u].score;
      }
    }
    __syncthreads();
    if (mode & 1) {
      const uint32_t c = m < k ? m : k;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (rk[u] < c) { out_ids[static_cast<uint64_t>(q) * out_k + rk[u]] = row_base + my[u].row; out_scores[static_cast<uint64_t>(q) * out_k + rk[u]] = my[u].score; }
      for (uint32_t j = c + tid; j < out_k; j += nth) { out_ids[static_cast<uint64_t>(q) * out_k + j] = ~0ull; out_scores[static_cast<uint64_t>(q) * out_k + j] = NEG_INF; }
      return;
    }
    const float sl = slack ? slack[q] : 0.f;
    // thresholds only ever rise: the previous one stays a valid lower bound of (k-th best - slack) when this list is
    // shorter than k (s_kth = -inf) or its k-th entry is lower (bootstrap rows that are not part of the first chunk)
    const float t = fmaxf(s_kth - sl, thr[q]);
    uint32_t local = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) if (rk[u] != 0xFFFFFFFFu) local += (sl > 0.f ? (my[u].score >= t) : (rk[u] < k)) ? 1u : 0u;
    if (local) atomicAdd(&s_keep, local);
    __syncthreads();
    const uint32_t keep = s_keep;                 // the kept entries are exactly ranks 0..keep-1
#pragma unroll
    for (int u = 0; u < 2; ++u) if (rk[u] < keep) mine[rk[u]] = my[u];
    if (tid == 0) { cnt[q] = (mode == 2) ? 0u : keep; thr[q] = t; }
    return;
  }
  uint32_t M2 = 1;
  while (M2 < m) M2 <<= 1;
  for (uint32_t i = tid; i < M2; i += nth) e[i] = (i < m) ? mine[i] : Cand{NEG_INF, 0xFFFFFFFFu};
  __syncthreads();
  for (uint32_t size = 2; size <= M2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = tid; i < (M2 >> 1); i += nth) {
        const uint32_t a = 2 * i - (i & (stride - 1));   // index with bit `stride` clear
        const uint32_t b = a + stride;
        const bool desc = ((a & size) == 0);             // first half of each 2*size block: best first
        const Cand ea = e[a], eb = e[b];
        const bool b_better = better(eb.score, eb.row, ea.score, ea.row);
        if (b_better == desc) { e[a] = eb; e[b] = ea; }
      }
      __syncthreads();
    }
  }
  if (mode & 1) {
    const uint32_t c = m < k ? m : k;
    for (uint32_t j = tid; j < out_k; j += nth) {
      const bool have = j < c;
      out_ids[static_cast<uint64_t>(q) * out_k + j] = have ? (row_base + e[j].row) : ~0ull;
      out_scores[static_cast<uint64_t>(q) * out_k + j] = have ? e[j].score : NEG_INF;
    }
    return;
  }
  const float kth = (m >= k) ? e[k - 1].score : NEG_INF;
  const float sl = slack ? slack[q] : 0.f;
  const float t = fmaxf(kth - sl, thr[q]);        // thresholds only ever rise (see the short-list path above)
  if (tid == 0) s_keep = 0;
  __syncthreads();
  uint32_t local = 0;
  if (sl > 0.f) { for (uint32_t i = tid; i < m; i += nth) local += (e[i].score >= t) ? 1u : 0u; }
  else { for (uint32_t i = tid; i < m; i += nth) local += (i < k) ? 1u : 0u; }
  if (local) atomicAdd(&s_keep, local);
  __syncthreads();
  const uint32_t keep = s_keep;                          // sorted list -> the kept ones are a prefix
  for (uint32_t i = tid; i < keep; i += nth) mine[i] = e[i];
  if (tid == 0) { cnt[q] = (mode == 2) ? 0u : keep; thr[q] = t; }
}

static __global__ __launch_bounds__(256) void select_kernel(
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t k, const float* __restrict__ slack,
    float* __restrict__ thr, uint32_t* __restrict__ overflow, int mode, uint64_t row_base,
    unsigned long long* __restrict__ out_ids, float* __restrict__ out_scores, uint32_t out_k, uint32_t* __restrict__ any_overflow,
    float* __restrict__ xcdw) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // (every earlier kernel of the search has completed when this one runs: block 0 may fold the self-check words)
  select_body(blockIdx.x, threadIdx.x, blockDim.x, reinterpret_cast<Cand*>(smem_raw), cand, cnt, cap, k, slack, thr, overflow, mode, row_base,
              out_ids, out_scores, out_k, any_overflow, xcdw, blockIdx.x == 0);
}

// one launch that resets all per-search words: list lengths, overflow flags, thresholds (-inf), self-check words
static __global__ __launch_bounds__(256) void init_search_kernel(uint32_t* __restrict__ cnt, uint32_t* __restrict__ overflow,
                                                          float* __restrict__ thr, uint32_t* __restrict__ misc, uint32_t nq_pad,
                                                          uint32_t* __restrict__ prog, uint32_t prog_words) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < nq_pad) { cnt[i] = 0; overflow[i] = 0; thr[i] = NEG_INF; }
  if (i < 8) misc[i] = 0;
  // sibling-rendezvous progress counters of this search's filter launches (one region per launch): "not started"
  for (uint32_t j = i; j < prog_words; j += gridDim.x * 256) prog[j] = 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------------
// rescore: replace the filter score of every surviving candidate by the exact reference score;
// count violations of the filter's error bound (must stay 0).  grid = nq, block = 256.
// ------------------------------------------------------------------------------------------------
template <int DT, bool ALIGNED>
__global__ __launch_bounds__(256) void rescore_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t dim, const float* __restrict__ q32,
    Cand* __restrict__ cand, const uint32_t* __restrict__ cnt, uint32_t cap, const float* __restrict__ ebound,
    uint32_t* __restrict__ violations, unsigned long long* __restrict__ total_cands) {
  const uint32_t q = blockIdx.x;
  uint32_t m = cnt[q];
  if (m > cap) m = cap;
  if (m == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];      // the query, padded to a multiple of 4 floats
  float* qptr = reinterpret_cast<float*>(smem_raw);
  const uint32_t qstride = (dim + 3u) & ~3u;
  for (uint32_t j = threadIdx.x; j < qstride; j += 256) qptr[j] = (j < dim) ? q32[static_cast<uint64_t>(q) * dim + j] : 0.f;
  __syncthreads();
  const float eb = ebound ? ebound[q] : 0.f;
  Cand* mine = cand + static_cast<uint64_t>(q) * cap;
  for (uint32_t i = threadIdx.x; i < m; i += 256) {
    const Cand c = mine[i];
    float sc[1];
    const float scale = (DT == DT_I8) ? scales[c.row] : 1.f;
    exact_scores<DT, 1, ALIGNED>(row_ptr<DT>(rows, c.row, dim), qptr, qstride, dim, scale, sc);
    if (ebound && eb >= 0.f && !(__builtin_fabsf(sc[0] - c.score) <= eb)) atomicAdd(violations, 1u);
    mine[i].score = sc[0];
  }
}

// Same result, eight lanes per candidate: lane j of a group IS accumulator lane j of the reference kernels
// (acc[j] += q[8i+j] * x[8i+j], i ascending -- simd_dot.cpp:31-36, 106-111; int8's two groups of 8 per iteration
// land in the same accumulators in the same order, :164-190), then the (a0+a4)+(a1+a5) .. reduction across the
// eight lanes (fp32 addition is commutative, so every lane ends with the same bits) and the scalar tails.
// 32 candidates per workgroup pass instead of 256 lanes each walking a whole row alone: 12x fewer dependent
// memory round trips for the ~12 candidates a query has left.
template <int DT>
__global__ __launch_bounds__(256) void rescore8_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t dim, const float* __restrict__ q32,
    Cand* __restrict__ cand, const uint32_t* __restrict__ cnt, uint32_t cap, const float* __restrict__ ebound,
    uint32_t* __restrict__ violations, unsigned long long* __restrict__ total_cands) {
  const uint32_t q = blockIdx.x;
  uint32_t m = cnt[q];
  if (m > cap) m = cap;
  if (m == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* qptr = reinterpret_cast<float*>(smem_raw);
  for (uint32_t j = threadIdx.x; j < dim; j += 256) qptr[j] = q32[static_cast<uint64_t>(q) * dim + j];
  __syncthreads();
  const float eb = ebound ? ebound[q] : 0.f;
  Cand* mine = cand + static_cast<uint64_t>(q) * cap;
  const uint32_t j8 = threadIdx.x & 7u, grp = threadIdx.x >> 3;
  const uint32_t body = (DT == DT_I8) ? (dim & ~15u) : (dim & ~7u);
  for (uint32_t base = 0; base < m; base += 32) {                 // uniform trip count: the shuffles need whole groups
    const uint32_t i = base + grp;
    const bool live = i < m;
    const Cand c = live ? mine[i] : Cand{0.f, 0u};
    const void* rp = row_ptr<DT>(rows, live ? c.row : 0u, dim);
    float acc = 0.f;
    uint32_t e = j8;
#pragma unroll 48
    for (; e < body; e += 8) acc = __builtin_fmaf(qptr[e], load1<DT>(rp, e), acc);
    float s = acc + __shfl_xor(acc, 4);                            // a_j + a_{j^4}
    s = s + __shfl_xor(s, 1);                                      // (s0+s1), (s2+s3)
    s = s + __shfl_xor(s, 2);                                      // (s0+s1)+(s2+s3)
    uint32_t t = body;
    if (t < dim) {
      if constexpr (DT == DT_F32) {
        if (dim - t >= 4) {
          for (int u = 0; u < 4; ++u) { const float p = qptr[t + u] * load1<DT>(rp, t + u); s = s + p; }
          t += 4;
        }
      }
      for (; t < dim; ++t) s = __builtin_fmaf(qptr[t], load1<DT>(rp, t), s);
    }
    if constexpr (DT == DT_I8) s = s * (live ? scales[c.row] : 1.f);
    if (live && j8 == 0) {
      if (ebound && eb >= 0.f && !(__builtin_fabsf(s - c.score) <= eb)) atomicAdd(violations, 1u);
      mine[i].score = s;
    }
  }
}

// The same eight-lanes-per-candidate arithmetic with the rows staged through LDS: the workgroup fetches the `cpp`
// candidate rows of a pass with coalesced 16-byte loads (all in flight at once), then every group walks its row in
// LDS.  Rows must be 16-byte multiples (the launcher falls back to rescore8_kernel otherwise); row slots are
// padded by 16 bytes so that the eight groups of a wave start in different banks.
// The final select folded into the rescore launch (same grid: one workgroup per query): after its candidates are re-scored the
// workgroup orders ITS list and emits the top-k -- one launch fewer per search.  mode 0: no select here (the caller launches
// select_kernel).  The self-check words are final only when EVERY workgroup has re-scored: the last one to take a ticket folds
// them into the sticky words (mode 1) and, for the host API's small calls, copies the 8 status words to `status_out` (pinned
// host memory the kernel writes directly, like out_ids / out_scores: no D2H copy is enqueued for them).
struct FinalSelect {
  int mode;                          // 0 none, 1 final select + sticky fold, 3 final select (host API: own words only)
  uint32_t k, out_k;
  uint64_t row_base;
  unsigned long long* out_ids;
  float* out_scores;
  float* thr;
  uint32_t* overflow;
  uint32_t* any_overflow;            // = misc + 6
  uint32_t* ticket;                  // zeroed by the search's init; null: no fold / no status copy
  uint32_t* status_out;              // null: none
};

template <int DT>
__global__ __launch_bounds__(256) void rescore_lds_kernel(
    const void* __restrict__ rows, const float* __restrict__ scales, uint32_t dim, const float* __restrict__ q32,
    Cand* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, const float* __restrict__ ebound,
    uint32_t* __restrict__ violations, unsigned long long* __restrict__ total_cands, uint32_t cpp, FinalSelect fs) {
  constexpr uint32_t BPE = (DT == DT_F32) ? 4 : (DT == DT_F16 ? 2 : 1);
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  uint32_t m = cnt[q];
  if (m > cap) m = cap;
  if (m == 0 && fs.mode == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t qstride = (dim + 3u) & ~3u, row_bytes = dim * BPE, slot_bytes = row_bytes + 16, cpr = row_bytes >> 4;
  float* qptr = reinterpret_cast<float*>(smem_raw);
  char* slots = smem_raw + qstride * 4;
  __shared__ Cand s_cand[32];
  for (uint32_t j = tid; j < dim; j += 256) qptr[j] = q32[static_cast<uint64_t>(q) * dim + j];
  const float eb = ebound ? ebound[q] : 0.f;
  Cand* mine = cand + static_cast<uint64_t>(q) * cap;
  const uint32_t j8 = tid & 7u, grp = tid >> 3;
  const uint32_t body = (DT == DT_I8) ? (dim & ~15u) : (dim & ~7u);
  for (uint32_t base = 0; base < m; base += cpp) {
    const uint32_t nrows = (m - base < cpp) ? m - base : cpp;
    __syncthreads();                                               // previous pass has finished with the slots
    if (tid < nrows) s_cand[tid] = mine[base + tid];
    __syncthreads();
    for (uint32_t c = tid; c < nrows * cpr; c += 256) {
      const uint32_t r = c / cpr, ch = c - r * cpr;
      const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const char*>(rows) + static_cast<uint64_t>(s_cand[r].row) * row_bytes + ch * 16);
      *reinterpret_cast<uint4*>(slots + r * slot_bytes + ch * 16) = v;
    }
    __syncthreads();
    if (grp < nrows) {                                             // whole groups of 8 lanes: the shuffles below stay inside a group
      const Cand c = s_cand[grp];
      const void* rp = slots + grp * slot_bytes;
      float acc = 0.f;
      uint32_t e = j8;
#pragma unroll 8
      for (; e < body; e += 8) acc = __builtin_fmaf(qptr[e], load1<DT>(rp, e), acc);
      float s = acc + __shfl_xor(acc, 4);
      s = s + __shfl_xor(s, 1);
      s = s + __shfl_xor(s, 2);
      uint32_t t = body;
      if (t < dim) {
        if constexpr (DT == DT_F32) {
          if (dim - t >= 4) {
            for (int u = 0; u < 4; ++u) { const float p = qptr[t + u] * load1<DT>(rp, t + u); s = s + p; }
            t += 4;
          }
        }
        for (; t < dim; ++t) s = __builtin_fmaf(qptr[t], load1<DT>(rp, t), s);
      }
      if constexpr (DT == DT_I8) s = s * scales[c.row];
      if (j8 == 0) {
        if (ebound && eb >= 0.f && !(__builtin_fabsf(s - c.score) <= eb)) atomicAdd(violations, 1u);
        mine[base + grp].score = s;
      }
    }
  }
  if (fs.mode == 0) return;
  __syncthreads();                                                 // my list's new scores are written (workgroup scope); the LDS is free
  select_body(q, tid, 256u, reinterpret_cast<Cand*>(smem_raw), cand, cnt, cap, fs.k, nullptr, fs.thr, fs.overflow, fs.mode, fs.row_base,
              fs.out_ids, fs.out_scores, fs.out_k, fs.any_overflow, nullptr, false);
  if (fs.ticket == nullptr) return;
  // what the last workgroup reads of the others are agent-scope atomics (violations, flags): performed once vmcnt is 0 -- no
  // release fence (an L2 write-back per workgroup costs more than the select launch this fusion saves)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const uint32_t t = __hip_atomic_fetch_add(fs.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {                                      // every workgroup of the search has finished
      if (fs.mode == 1) fold_sticky_words(fs.any_overflow);
      if (fs.status_out != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fs.status_out[i] = __hip_atomic_load(fs.any_overflow - 6 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard top-k lists: in[s][nq][k] -> out[nq][k]; rank-based (entries are unique).
// grid = nq, block = 256; nshards*k <= 4096.
// ------------------------------------------------------------------------------------------------
// Shard s's ids start at ids + s*stride_ids_bytes, its scores at scores + s*stride_scores_bytes (so that one
// all-gather of a packed [ids | scores] buffer per rank can be merged in place).
static __global__ __launch_bounds__(256) void merge_topk_kernel(
    const unsigned long long* __restrict__ ids, const float* __restrict__ scores, uint32_t nshards, uint32_t nq,
    uint32_t k, unsigned long long* __restrict__ out_ids, float* __restrict__ out_scores, size_t stride_ids_bytes,
    size_t stride_scores_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t m = nshards * k, q = blockIdx.x;
  float* s = reinterpret_cast<float*>(smem_raw);
  unsigned long long* id = reinterpret_cast<unsigned long long*>(smem_raw + ((m * 4 + 15) & ~15u));
  for (uint32_t i = threadIdx.x; i < m; i += 256) {
    const uint32_t sh = i / k, j = i % k;
    s[i] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(scores) + sh * stride_scores_bytes)[static_cast<uint64_t>(q) * k + j];
    id[i] = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(ids) + sh * stride_ids_bytes)[static_cast<uint64_t>(q) * k + j];
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < m; i += 256) {
    const float si = s[i];
    const unsigned long long ii = id[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < m; ++j) {
      const float sj = s[j];
      const unsigned long long ij = id[j];
      // padding entries (id == ~0) sort last; equal pads are ordered by position
      const bool jb = (sj > si) || (sj == si && (ij < ii || (ij == ii && j < i)));
      rank += jb ? 1u : 0u;
    }
    if (rank < k) { out_ids[static_cast<uint64_t>(q) * k + rank] = ii; out_scores[static_cast<uint64_t>(q) * k + rank] = si; }
  }
}

// The same merge for nshards * k beyond what fits the LDS of merge_topk_kernel (4096 entries): every shard's list arrives sorted
// best-first by (score desc, id asc) with its padding (id ~0, -inf) last, so an entry's position in the merged list is its own
// position plus, for every other shard, the number of that shard's entries that come before it -- one binary search per shard
// straight out of the gathered buffer (L2).  Any k the flat path accepts (the reference bounds k by N only, flat_index.cpp:24).
// Total order = merge_topk_kernel's: (score desc, id asc, flat position asc); ids are global, hence distinct except for padding.
static __global__ __launch_bounds__(256) void merge_topk_sorted_kernel(
    const unsigned long long* __restrict__ ids, const float* __restrict__ scores, uint32_t nshards, uint32_t nq,
    uint32_t k, unsigned long long* __restrict__ out_ids, float* __restrict__ out_scores, size_t stride_ids_bytes,
    size_t stride_scores_bytes) {
  const uint32_t q = blockIdx.y;
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nshards * k) return;
  const uint32_t sh = i / k, j = i % k;
  const uint64_t qoff = static_cast<uint64_t>(q) * k;
  const float si = reinterpret_cast<const float*>(reinterpret_cast<const char*>(scores) + sh * stride_scores_bytes)[qoff + j];
  const unsigned long long ii = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(ids) + sh * stride_ids_bytes)[qoff + j];
  uint32_t rank = j;
  for (uint32_t t = 0; t < nshards && rank < k; ++t) {
    if (t == sh) continue;
    const float* ts = reinterpret_cast<const float*>(reinterpret_cast<const char*>(scores) + t * stride_scores_bytes) + qoff;
    const unsigned long long* ti = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(ids) + t * stride_ids_bytes) + qoff;
    uint32_t lo = 0, hi = k;                         // first position of shard t that does NOT come before my entry
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      const float sj = ts[mid];
      const unsigned long long ij = ti[mid];
      const bool before = (sj > si) || (sj == si && (ij < ii || (ij == ii && t < sh)));
      if (before) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  if (rank < k) { out_ids[qoff + rank] = ii; out_scores[qoff + rank] = si; }
}

}  // namespace nvdbhip
