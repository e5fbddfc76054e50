// kernels_largek.h -- exact top-k for ANY k (the reference clamps k only to N, src/flat_index.cpp:24).
//
// The wavefront-resident lists of kernels_exact.h hold k <= 64 and the MFMA filter's candidate lists k <= 1024.
// Beyond that (and as the always-correct fallback of the filter path for k > 64) the search runs here, on a
// sub-batch of queries at a time:
//
//   scores_exact_kernel   every (query,row) score in the reference's fp32 order (exact_scores<>), written to a
//                         [queries][rows] matrix in HBM;
//   radix_hist / _pick    8 passes of an 8-bit radix select over the 64-bit key  (order-preserving score bits << 32 |
//                         ~row): its descending order IS the canonical (score desc, id asc) order, keys are distinct,
//                         so the k-th largest key is unique;
//   collect_kernel        the k entries with key >= that key, in any order;
//   bitonic_*             sort them descending (in LDS up to 8192 entries, otherwise in global memory);
//   emit_kernel           ids (global) and the scores' original bits.
//
// Nothing here is on the headline path; it is HBM-bound bookkeeping behind a VALU-bound scoring kernel.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_exact.h"

namespace nvdbhip {

// order-preserving map of a score; -0.0 and +0.0 compare equal in better(), so they share a key
__device__ __forceinline__ uint32_t score_key(float s) {
  s = s + 0.0f;
  const uint32_t u = __builtin_bit_cast(uint32_t, s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long key64_of(float s, uint32_t row) {
  return (static_cast<unsigned long long>(score_key(s)) << 32) | static_cast<uint32_t>(~row);
}

// grid = (row blocks, ceil(nq / QG)), block = 256.  out[g][row] for the group's queries; ld = row stride of `out`.
template <int DT, int QG, bool ALIGNED>
__global__ __launch_bounds__(256) void scores_exact_kernel(const void* __restrict__ rows, const float* __restrict__ scales, uint32_t dim,
                                                           uint32_t n, const float* __restrict__ q32, uint32_t nq,
                                                           float* __restrict__ out, uint64_t ld) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t qstride = (dim + 3u) & ~3u;
  float* q_lds = reinterpret_cast<float*>(smem_raw);
  const uint32_t qg0 = blockIdx.y * QG;
  const uint32_t qbase = (qg0 + QG <= nq) ? qg0 : (nq >= static_cast<uint32_t>(QG) ? nq - QG : 0u);
  for (uint32_t e = threadIdx.x; e < QG * qstride; e += 256) {
    const uint32_t g = e / qstride, j = e % qstride;
    q_lds[e] = (j < dim && qbase + g < nq) ? q32[static_cast<uint64_t>(qbase + g) * dim + j] : 0.f;
  }
  __syncthreads();
  for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * 256u; base < n; base += static_cast<uint64_t>(gridDim.x) * 256u) {
    const uint64_t row = base + threadIdx.x;
    const bool valid = row < n;
    const uint32_t rrow = valid ? static_cast<uint32_t>(row) : n - 1;
    float sc[QG];
    const float scale = (DT == DT_I8) ? scales[rrow] : 1.f;
    exact_scores<DT, QG, ALIGNED>(row_ptr<DT>(rows, rrow, dim), q_lds, qstride, dim, scale, sc);
    if (valid) {
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        const uint32_t qi = qbase + g;
        if (qi >= qg0 && qi < nq) out[static_cast<uint64_t>(qi) * ld + row] = sc[g];     // a group shifted back owns only queries >= qg0
      }
    }
  }
}

// state per query: prefix (the key's top 8*pass bits found so far), krem (rank still to find inside that prefix)
struct RadixState { unsigned long long prefix; uint32_t krem; uint32_t taken; };

// pass p (0..7): histogram of byte (7-p) of the keys whose top p bytes equal the prefix.  grid = (G, nq), block 256.
static __global__ __launch_bounds__(256) void radix_hist_kernel(const float* __restrict__ scores, uint64_t ld, uint32_t n, int pass,
                                                         const RadixState* __restrict__ st, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  const uint32_t q = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long prefix = st[q].prefix;
  const int shift = 56 - 8 * pass;
  const float* s = scores + static_cast<uint64_t>(q) * ld;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256u) {
    const unsigned long long key = key64_of(s[i], static_cast<uint32_t>(i));
    if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  const uint32_t v = h[threadIdx.x];
  if (v) atomicAdd(&hist[q * 256u + threadIdx.x], v);
}

// grid = nq, block = 256: the digit that holds the krem-th largest key of this pass; clears the histogram for the next.
static __global__ __launch_bounds__(256) void radix_pick_kernel(RadixState* __restrict__ st, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  __shared__ uint32_t above[256];
  const uint32_t q = blockIdx.x, d = threadIdx.x;
  const uint32_t krem = st[q].krem;                   // read by every thread before the one winner below rewrites it
  const unsigned long long prefix = st[q].prefix;
  h[d] = hist[q * 256u + d];
  hist[q * 256u + d] = 0;
  __syncthreads();
  uint32_t a = 0;                                     // keys of this pass with a larger digit
  for (uint32_t j = d + 1; j < 256; ++j) a += h[j];
  above[d] = a;
  __syncthreads();
  if (above[d] < krem && krem <= above[d] + h[d]) {   // exactly one digit satisfies this (krem >= 1, total >= krem)
    st[q].prefix = (prefix << 8) | d;
    st[q].krem = krem - above[d];
  }
}

static __global__ void radix_init_kernel(RadixState* __restrict__ st, uint32_t* __restrict__ hist, uint32_t nq, uint32_t k) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < nq) { st[i].prefix = 0; st[i].krem = k; st[i].taken = 0; }
  if (i < nq * 256u) hist[i] = 0;
}

// sel[q][0..k) = every key >= the k-th largest (after pass 7 the prefix IS that key); slots >= k up to K2 are padding 0
static __global__ __launch_bounds__(256) void collect_kernel(const float* __restrict__ scores, uint64_t ld, uint32_t n, RadixState* __restrict__ st,
                                                      unsigned long long* __restrict__ sel, uint32_t K2, uint32_t k) {
  const uint32_t q = blockIdx.y;
  const unsigned long long kth = st[q].prefix;
  const float* s = scores + static_cast<uint64_t>(q) * ld;
  unsigned long long* mine = sel + static_cast<uint64_t>(q) * K2;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256u) {
    const unsigned long long key = key64_of(s[i], static_cast<uint32_t>(i));
    if (key >= kth) {
      const uint32_t slot = atomicAdd(&st[q].taken, 1u);
      if (slot < k) mine[slot] = key;
    }
  }
  if (blockIdx.x == 0) for (uint32_t j = k + threadIdx.x; j < K2; j += 256) mine[j] = 0ull;
}

// descending bitonic sort of K2 (power of two, <= 8192) keys per query in LDS.  grid = nq, block = 256, LDS = K2 * 8.
static __global__ __launch_bounds__(256) void bitonic_lds_kernel(unsigned long long* __restrict__ sel, uint32_t K2) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned long long* e = reinterpret_cast<unsigned long long*>(smem_raw);
  unsigned long long* mine = sel + static_cast<uint64_t>(blockIdx.x) * K2;
  for (uint32_t i = threadIdx.x; i < K2; i += 256) e[i] = mine[i];
  __syncthreads();
  for (uint32_t size = 2; size <= K2; size <<= 1)
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t i = threadIdx.x; i < (K2 >> 1); i += 256) {
        const uint32_t a = 2 * i - (i & (stride - 1)), b = a + stride;
        const bool desc = ((a & size) == 0);
        const unsigned long long ea = e[a], eb = e[b];
        if ((eb > ea) == desc) { e[a] = eb; e[b] = ea; }
      }
      __syncthreads();
    }
  for (uint32_t i = threadIdx.x; i < K2; i += 256) mine[i] = e[i];
}

// one compare-exchange step of the same network in global memory (K2 > 8192).  grid = (K2 / 512, nq), block = 256.
static __global__ __launch_bounds__(256) void bitonic_global_step_kernel(unsigned long long* __restrict__ sel, uint32_t K2, uint32_t size, uint32_t stride) {
  unsigned long long* mine = sel + static_cast<uint64_t>(blockIdx.y) * K2;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (K2 >> 1)) return;
  const uint32_t a = 2 * i - (i & (stride - 1)), b = a + stride;
  const bool desc = ((a & size) == 0);
  const unsigned long long ea = mine[a], eb = mine[b];
  if ((eb > ea) == desc) { mine[a] = eb; mine[b] = ea; }
}

// out[q][j] for j < out_k: the j-th key's row (global id) and the score's original bits; j >= k_eff padded
static __global__ __launch_bounds__(256) void emit_kernel(const unsigned long long* __restrict__ sel, uint32_t K2, const float* __restrict__ scores, uint64_t ld,
                                                   uint32_t k_eff, uint32_t out_k, uint64_t row_base, unsigned long long* __restrict__ out_ids,
                                                   float* __restrict__ out_scores) {
  const uint32_t q = blockIdx.y;
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= out_k) return;
  const uint64_t o = static_cast<uint64_t>(q) * out_k + j;
  if (j < k_eff) {
    const uint32_t row = ~static_cast<uint32_t>(sel[static_cast<uint64_t>(q) * K2 + j]);
    out_ids[o] = row_base + row;
    out_scores[o] = scores[static_cast<uint64_t>(q) * ld + row];
  } else {
    out_ids[o] = ~0ull;
    out_scores[o] = NEG_INF;
  }
}

// the same keys into the filter path's candidate lists instead (exact bootstrap of a wide-k search, search_core): entry j of
// query q = (exact score, row inside the shard); cnt[q] = k
static __global__ __launch_bounds__(256) void seed_lists_kernel(const unsigned long long* __restrict__ sel, uint32_t K2, const float* __restrict__ scores, uint64_t ld,
                                                         uint32_t k, Cand* __restrict__ cand, uint32_t cap, uint32_t* __restrict__ cnt) {
  const uint32_t q = blockIdx.y;
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= k) return;
  const uint32_t row = ~static_cast<uint32_t>(sel[static_cast<uint64_t>(q) * K2 + j]);
  cand[static_cast<uint64_t>(q) * cap + j] = Cand{scores[static_cast<uint64_t>(q) * ld + row], row};
  if (j == 0) cnt[q] = k;
}

}  // namespace nvdbhip
