/*
 * nvdb_oracle.c -- CPU restatement of nano-vectordb's flat-scan + exact-refine arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under nano-vectordb_amd/ (the product) links, loads or calls
 * this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * only as the checker.
 *
 * Pinning: every function below is checked bit-for-bit against the real reference, compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/ (see oracle/make_golden.py); the resulting
 * vectors are committed under tests/golden/.  The exception is the refine path
 * (oracle_refine_*): the reference's CUDA kernel cannot be built or run here and the reference
 * holds no fixtures for it -> "parity unpinned" for those two functions.
 *
 * All citations are file:line into /root/reference.
 *
 * Build: gcc -O2 -mfma -mf16c -ffp-contract=off -fopenmp -shared -fPIC  (see oracle/Makefile).
 * -ffp-contract=off so that the ONLY fused operations are the explicit fmaf() calls below.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* half <-> float                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* include/nvdb/f16_scalar.h:9-36 and src/simd_dot.cpp:67-99 (identical): IEEE half bits -> float,
 * exact for every input (subnormals normalised, Inf/NaN payload kept). _mm256_cvtph_ps
 * (simd_dot.cpp:110) computes the same function. */
float oracle_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)(h & 0x8000u)) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, out;
  if (e == 0) {
    if (m == 0) out = sign;
    else {
      int ex = 1;
      while ((m & 0x400u) == 0) { m <<= 1; ex--; }
      m &= 0x3FFu;
      out = sign | ((uint32_t)(ex - 15 + 127) << 23) | (m << 13);
    }
  } else if (e == 0x1F) out = sign | 0x7F800000u | (m << 13);
  else out = sign | ((e - 15 + 127) << 23) | (m << 13);
  float f; memcpy(&f, &out, 4); return f;
}

/* tools/nvdb_convert_f16.cpp:20-94 (scalar RNE) == _mm256_cvtps_ph(v,0) at :99-107 for all
 * finite inputs that are not f32 subnormals; f32 subnormals -> signed zero in both (:40-43; the
 * F16C instruction rounds them to zero as well because they are < 2^-25). */
static uint16_t f32_to_f16_impl(float f, int scalar_path) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 31) & 1u, be = (x >> 23) & 0xFFu, mant = x & 0x7FFFFFu;
  if (be == 0xFF) {
    if (mant == 0) return (uint16_t)((sign << 15) | 0x7C00u);
    /* NaN: scalar path keeps payload>>13 and forces it non-zero (:32-37). NaN corpora are outside
     * the tested domain. */
    uint16_t pm = (uint16_t)((mant >> 13) & 0x3FFu); if (!pm) pm = 1;
    return (uint16_t)((sign << 15) | 0x7C00u | pm);
  }
  if (be == 0) return (uint16_t)(sign << 15);            /* f32 subnormal -> signed zero (:40-43) */
  int e = (int)be - 127;
  mant |= 0x800000u;
  if (e > 15) return (uint16_t)((sign << 15) | 0x7C00u); /* overflow -> Inf (:49-51) */
  if (e < -14) {                                         /* half subnormal (:54-69) */
    int shift = -14 - e;
    if (shift > 24) return (uint16_t)(sign << 15);
    uint32_t ms = mant >> (shift + 13);
    uint32_t rem = mant & ((1u << (shift + 13)) - 1u);
    uint32_t half = 1u << (shift + 12);
    if (rem > half || (rem == half && (ms & 1u))) ms++;
    /* A carry into bit 10 (value rounds up to the smallest normal) is DROPPED by the reference's
     * scalar path, which masks with 0x3FF (:68); the F16C instruction yields 0x0400. */
    return (uint16_t)((sign << 15) | (ms & (scalar_path ? 0x3FFu : 0x7FFu)));
  }
  uint32_t he = (uint32_t)(e + 15);                      /* normal half (:72-93) */
  if (scalar_path) {
    /* Faithful to two defects of the reference's scalar routine, which only ever sees the dim%8
     * tail elements on F16C hosts (:106): (i) the "round to even" fix-up clears bit 0 of the
     * biased mantissa (:77-79), which has no effect after >>13, so exact halfway cases round UP;
     * (ii) the mantissa is masked with 0x3FF BEFORE the overflow test (:81-89), so a carry out of
     * the mantissa never bumps the exponent (e.g. -0.0624997 -> 0xA800 = -0.03125).  The product's
     * converter does NOT copy these (DESIGN.md "reference defects not copied"). */
    uint32_t mr = mant + 0x1000u;
    if ((mant & 0x1FFFu) == 0x1000u) mr &= ~1u;
    return (uint16_t)((sign << 15) | (he << 10) | ((mr >> 13) & 0x3FFu));
  }
  uint32_t hm = mant >> 13, rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1u))) hm++;
  if (hm & 0x800u) { hm >>= 1; he++; }
  if (he >= 0x1F) return (uint16_t)((sign << 15) | 0x7C00u);
  return (uint16_t)((sign << 15) | (he << 10) | (hm & 0x3FFu));
}

uint16_t oracle_f32_to_f16(float f) { return f32_to_f16_impl(f, 0); }

/* One row, as tools/nvdb_convert_f16.cpp:96-119 converts it on an F16C host: groups of 8 by the
 * instruction (:101-105), the dim%8 tail by the scalar routine (:106). */
void oracle_convert_f32_to_f16(const float* src, uint16_t* dst, uint64_t n) {
  uint64_t i = 0, n8 = n & ~(uint64_t)7;
  for (; i < n8; ++i) dst[i] = f32_to_f16_impl(src[i], 0);
  for (; i < n; ++i) dst[i] = f32_to_f16_impl(src[i], 1);
}

/* apps/nvdb_quantize_i8.cpp:12-16, 71-80: per-row scale = max_abs/127 (1 if the row is all zero),
 * inv = 1.0f/scale, q = lrint(x*inv) (float product, current rounding mode = nearest-even),
 * clamp to [-127,127]. */
float oracle_quantize_i8_row(const float* row, uint32_t dim, int8_t* out) {
  float max_abs = 0.f;
  for (uint32_t j = 0; j < dim; ++j) { float a = fabsf(row[j]); if (a > max_abs) max_abs = a; }
  float scale = (max_abs > 0.f) ? (max_abs / 127.f) : 1.f;
  float inv = 1.0f / scale;
  for (uint32_t j = 0; j < dim; ++j) {
    float p = row[j] * inv;
    long q = lrint((double)p);
    if (q > 127) q = 127; if (q < -127) q = -127;
    out[j] = (int8_t)q;
  }
  return scale;
}

/* ------------------------------------------------------------------------------------------ */
/* dot kernels: the AVX2+FMA paths restated as scalar code with explicit fmaf()                */
/* ------------------------------------------------------------------------------------------ */

/* horizontal sum of the 8-lane accumulator: src/simd_dot.cpp:38-44 (identical at :114-119 and
 * :186-191): (lo+hi) -> hadd -> hadd  ==  ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)). */
static inline float hsum8(const float a[8]) {
  float s0 = a[0] + a[4], s1 = a[1] + a[5], s2 = a[2] + a[6], s3 = a[3] + a[7];
  return (s0 + s1) + (s2 + s3);
}

/* src/simd_dot.cpp:26-49 (dot_avx2_fma), reached from dot_f32 (:52-64) on AVX2+FMA hosts.
 * Lane j accumulates elements j, j+8, j+16, ... with one fused multiply-add each.
 * Tail (:47, `for (; i < dim; ++i) out += a[i]*b[i];`, at most 7 elements): what the C++ source
 * leaves to the compiler (contraction is optional) is restated as GCC 11.4 -O3 compiles it in
 * oracle/_ref (checked in the disassembly and pinned by the dim = 100/37/15/13/300/... goldens):
 * the loop is vectorised in groups of FOUR with a separate multiply (vmulps) and in-order adds
 * (vaddss), i.e. unfused; the remaining 0..3 elements use vfmadd231ss, i.e. fused.  The f16 and
 * int8 kernels' tails are not vectorised by that compiler and are fused throughout. */
float oracle_dot_f32(const float* a, const float* b, uint32_t dim) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t i = 0;
  for (; i + 8 <= dim; i += 8)
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(a[i + j], b[i + j], acc[j]);
  float out = hsum8(acc);
  if (dim - i >= 4) {
    for (int j = 0; j < 4; ++j) { float p = a[i + j] * b[i + j]; out = out + p; }
    i += 4;
  }
  for (; i < dim; ++i) out = fmaf(a[i], b[i], out);
  return out;
}

/* src/simd_dot.cpp:102-124 (dot_f32_f16base_avx2) via :127-136. */
float oracle_dot_f32_f16base(const float* q, const uint16_t* x, uint32_t dim) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t i = 0;
  for (; i + 8 <= dim; i += 8)
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(q[i + j], oracle_f16_to_f32(x[i + j]), acc[j]);
  float out = hsum8(acc);
  for (; i < dim; ++i) out = fmaf(q[i], oracle_f16_to_f32(x[i]), out);
  return out;
}

/* src/simd_dot.cpp:160-199 (dot_f32_i8_avx2) via :202-213.  16 int8 per iteration, two FMAs into
 * the SAME 8-lane accumulator (elements i..i+7 then i+8..i+15), i.e. the same stride-8 lane order
 * as the other two kernels but only over floor(dim/16)*16 elements; the tail (:196) is scalar;
 * one fp32 multiply by the row scale at the end (:198). */
float oracle_dot_f32_i8base(const float* q, const int8_t* x, uint32_t dim, float scale) {
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t i = 0;
  for (; i + 16 <= dim; i += 16) {
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(q[i + j], (float)x[i + j], acc[j]);
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(q[i + 8 + j], (float)x[i + 8 + j], acc[j]);
  }
  float out = hsum8(acc);
  for (; i < dim; ++i) out = fmaf(q[i], (float)x[i], out);
  return out * scale;
}

/* scalar fall-backs (NVDB_FORCE_SCALAR=1): src/simd_dot.cpp:18-22, :133-135, :151-157. */
float oracle_dot_f32_scalar(const float* a, const float* b, uint32_t dim) {
  double s = 0.0; for (uint32_t i = 0; i < dim; ++i) s += (double)a[i] * (double)b[i]; return (float)s;
}
float oracle_dot_f32_i8base_scalar(const float* q, const int8_t* x, uint32_t dim, float scale) {
  double s = 0.0; for (uint32_t i = 0; i < dim; ++i) s += (double)q[i] * (double)x[i];
  return (float)(s * (double)scale);
}

/* include/nvdb/score_dispatch.h:25-48: dtype 1=f32, 2=f16, 3=i8(+scale). */
static inline float score_at(const void* base, const float* scales, uint32_t dtype, uint32_t dim,
                             uint64_t row, const float* q) {
  if (dtype == 1) return oracle_dot_f32(q, (const float*)base + row * dim, dim);
  if (dtype == 2) return oracle_dot_f32_f16base(q, (const uint16_t*)base + row * dim, dim);
  return oracle_dot_f32_i8base(q, (const int8_t*)base + row * dim, dim, scales[row]);
}

/* All N scores of one query (what FlatIndex computes inside its loop, src/flat_index.cpp:31-35). */
void oracle_scores(const void* base, const float* scales, uint32_t dtype, uint64_t n, uint32_t dim,
                   const float* q, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = score_at(base, scales, dtype, dim, (uint64_t)i, q);
}

/* ------------------------------------------------------------------------------------------ */
/* top-k                                                                                       */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float s; uint64_t id; } ent_t;

/* canonical order: score descending, id ascending.  The reference's variants disagree inside
 * exact-score ties; this order is what src/flat_index.cpp:36-37 produces at
 * the k-boundary (strict '>' keeps the earlier = lower id) and is used for every tie. */
static inline int better(float s1, uint64_t i1, float s2, uint64_t i2) {
  return (s1 > s2) || (s1 == s2 && i1 < i2);
}

/* Bounded insertion list kept sorted best-first; O(k) per accepted element like topK.h:15-69. */
static void topk_consider(ent_t* buf, uint32_t* cnt, uint32_t k, float s, uint64_t id) {
  if (*cnt == k && !better(s, id, buf[k - 1].s, buf[k - 1].id)) return;
  uint32_t pos = (*cnt < k) ? (*cnt)++ : k - 1;
  while (pos > 0 && better(s, id, buf[pos - 1].s, buf[pos - 1].id)) { buf[pos] = buf[pos - 1]; --pos; }
  buf[pos].s = s; buf[pos].id = id;
}

/* src/flat_index.cpp:16-48 (FlatIndex::search_topk_dot) for nq queries:
 *   k==0 -> nothing; k>n -> clamp (:24); result best-first (:40-47).
 * out_ids/out_scores are [nq][k_eff], k_eff = min(k,n) is returned.  Returns 0 on "Empty base"
 * (:17 throws there).  Ties: canonical order above. */
uint32_t oracle_flat_topk(const void* base, const float* scales, uint32_t dtype, uint64_t n,
                          uint32_t dim, const float* queries, uint32_t nq, uint32_t k,
                          uint64_t* out_ids, float* out_scores) {
  if (n == 0 || k == 0) return 0;
  if (k > n) k = (uint32_t)n;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t qi = 0; qi < (int64_t)nq; ++qi) {
    ent_t* buf = (ent_t*)malloc(sizeof(ent_t) * k);
    uint32_t cnt = 0;
    const float* q = queries + (uint64_t)qi * dim;
    for (uint64_t i = 0; i < n; ++i) topk_consider(buf, &cnt, k, score_at(base, scales, dtype, dim, i, q), i);
    for (uint32_t j = 0; j < k; ++j) { out_ids[(uint64_t)qi * k + j] = buf[j].id; out_scores[(uint64_t)qi * k + j] = buf[j].s; }
    free(buf);
  }
  return k;
}

/* Same selection on a caller-provided score vector (used by size-independent property tests). */
uint32_t oracle_topk_of_scores(const float* scores, uint64_t n, uint32_t k, uint64_t* out_ids, float* out_scores) {
  if (n == 0 || k == 0) return 0;
  if (k > n) k = (uint32_t)n;
  ent_t* buf = (ent_t*)malloc(sizeof(ent_t) * k);
  uint32_t cnt = 0;
  for (uint64_t i = 0; i < n; ++i) topk_consider(buf, &cnt, k, scores[i], i);
  for (uint32_t j = 0; j < k; ++j) { out_ids[j] = buf[j].id; out_scores[j] = buf[j].s; }
  free(buf);
  return k;
}

/* ------------------------------------------------------------------------------------------ */
/* exact-L2 refine (PARITY UNPINNED: no reference fixtures, CUDA path not runnable here)       */
/* ------------------------------------------------------------------------------------------ */

/* src/cuda_refine.cu:326-382 (l2_fp16_base_half2): dims are consumed as half2 pairs p=0..D/2-1;
 * groups of 4 pairs (8 dims) feed accumulators 0..3 (pair p -> acc[p&3]), each pair doing
 * acc=fmaf(dx,dx,acc); acc=fmaf(dy,dy,acc); leftover pairs (D/2 % 4) all go to acc0 (:368-376);
 * result (acc0+acc1)+(acc2+acc3) (:378).  D is assumed even (:339). */
float oracle_l2_f16_gpu_order(const float* q, const uint16_t* x, uint32_t dim) {
  float acc[4] = {0, 0, 0, 0};
  uint32_t d2 = dim / 2, p = 0;
  for (; p + 3 < d2; p += 4)
    for (uint32_t c = 0; c < 4; ++c) {
      float dx = q[2 * (p + c)] - oracle_f16_to_f32(x[2 * (p + c)]);
      float dy = q[2 * (p + c) + 1] - oracle_f16_to_f32(x[2 * (p + c) + 1]);
      acc[c] = fmaf(dx, dx, acc[c]); acc[c] = fmaf(dy, dy, acc[c]);
    }
  for (p = d2 & ~3u; p < d2; ++p) {
    float dx = q[2 * p] - oracle_f16_to_f32(x[2 * p]);
    float dy = q[2 * p + 1] - oracle_f16_to_f32(x[2 * p + 1]);
    acc[0] = fmaf(dx, dx, acc[0]); acc[0] = fmaf(dy, dy, acc[0]);
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

/* src/cuda_refine.cu:383-392 (l2_fp32_base): single accumulator, `acc += diff*diff`, which nvcc's
 * default -fmad=true contracts to one FMA per element. (The reference never launches this kernel,
 * cuda_refine.cu:1055-1085 -- SURVEY.md section 0.8; the build implements it properly.) */
float oracle_l2_f32_gpu_order(const float* q, const float* x, uint32_t dim) {
  float acc = 0.f;
  for (uint32_t j = 0; j < dim; ++j) { float d = q[j] - x[j]; acc = fmaf(d, d, acc); }
  return acc;
}

/* include/nvdb/to_f32_row.h:10-34 (base_row_to_f32): fp32 copied, fp16 through f16_to_f32_scalar
 * (include/nvdb/f16_scalar.h:8-38), int8 as float(v) * scale.  PINNED: tests/golden/refine_conv_golden.npz holds the
 * real reference's output for all 65 536 half patterns and for rows of each dtype. */
void oracle_base_row_to_f32(const void* x, float scale, uint32_t dtype, uint32_t dim, float* out) {
  for (uint32_t j = 0; j < dim; ++j)
    out[j] = (dtype == 1) ? ((const float*)x)[j]
           : (dtype == 2) ? oracle_f16_to_f32(((const uint16_t*)x)[j])
                          : (float)((const int8_t*)x)[j] * scale;
}

/* apps/nvdb_ivf_eval.cpp:232-240 (l2_sqr_f32) on a row widened by base_row_to_f32: double
 * accumulation of (double(a)-double(b))^2, cast to float at the end. */
float oracle_l2_cpu_double(const float* q, const void* x, uint32_t dtype, uint32_t dim) {
  float row[dim];
  oracle_base_row_to_f32(x, 0.f, dtype, dim, row);
  double s = 0.0;
  for (uint32_t j = 0; j < dim; ++j) { double d = (double)q[j] - (double)row[j]; s += d * d; }
  return (float)s;
}

static inline int closer(float d1, uint32_t i1, float d2, uint32_t i2) {
  return (d1 < d2) || (d1 == d2 && i1 < i2);
}

/* Batched rerank with the interface of cuda_l2_topk_batch (include/nvdb/cuda_refine.h:25-38,
 * src/cuda_refine.cu:839-1173): candidates equal to 0xFFFFFFFF or >= N are skipped (:437);
 * outputs ascending by distance, padded with id 0xFFFFFFFF / dist 1e30 (:892-894, :248-252).
 * mode 0: distances in the GPU kernel's fp32 order; mode 1: the CPU refine's double accumulation
 * (apps/nvdb_ivf_eval.cpp:278-307).  Ties: (dist asc, id asc); duplicates of one id in a
 * candidate list are kept as separate entries, as in both reference paths. */
void oracle_refine_l2_topk(const void* base, uint32_t dtype, uint64_t n, uint32_t dim,
                           const float* queries, const uint32_t* cand, uint32_t Q, uint32_t R,
                           uint32_t K, int mode, uint32_t* out_ids, float* out_dist) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t qi = 0; qi < (int64_t)Q; ++qi) {
    const float* q = queries + (uint64_t)qi * dim;
    uint32_t* oi = out_ids + (uint64_t)qi * K; float* od = out_dist + (uint64_t)qi * K;
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < K; ++j) { oi[j] = 0xFFFFFFFFu; od[j] = 1e30f; }
    for (uint32_t r = 0; r < R; ++r) {
      uint32_t id = cand[(uint64_t)qi * R + r];
      if (id == 0xFFFFFFFFu || (uint64_t)id >= n) continue;
      const void* row = (dtype == 1) ? (const void*)((const float*)base + (uint64_t)id * dim)
                                     : (const void*)((const uint16_t*)base + (uint64_t)id * dim);
      float d = (mode == 1) ? oracle_l2_cpu_double(q, row, dtype, dim)
              : (dtype == 1) ? oracle_l2_f32_gpu_order(q, (const float*)row, dim)
                             : oracle_l2_f16_gpu_order(q, (const uint16_t*)row, dim);
      if (cnt == K && !closer(d, id, od[K - 1], oi[K - 1])) continue;
      uint32_t pos = (cnt < K) ? cnt++ : K - 1;
      while (pos > 0 && closer(d, id, od[pos - 1], oi[pos - 1])) { od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1]; --pos; }
      od[pos] = d; oi[pos] = id;
    }
  }
}

/* CPU-baseline helper for bench.py when oracle/_ref is absent (kind "port"): OpenMP static
 * partition + per-thread lists + serial merge, the structure of src/flat_index_omp.cpp:16-85. */
uint32_t oracle_flat_topk_omp(const void* base, const float* scales, uint32_t dtype, uint64_t n,
                              uint32_t dim, const float* q, uint32_t k, int nthreads,
                              uint64_t* out_ids, float* out_scores) {
  if (n == 0 || k == 0) return 0;
  if (k > n) k = (uint32_t)n;
  if (nthreads < 1) nthreads = 1;
  ent_t* lists = (ent_t*)malloc(sizeof(ent_t) * (size_t)k * nthreads);
  uint32_t* cnts = (uint32_t*)calloc(nthreads, sizeof(uint32_t));
#pragma omp parallel num_threads(nthreads)
  {
    int t = 0, T = 1;
#ifdef _OPENMP
    extern int omp_get_thread_num(void); extern int omp_get_num_threads(void);
    t = omp_get_thread_num(); T = omp_get_num_threads();
#endif
    uint64_t lo = n * (uint64_t)t / T, hi = n * (uint64_t)(t + 1) / T;
    for (uint64_t i = lo; i < hi; ++i)
      topk_consider(lists + (size_t)t * k, &cnts[t], k, score_at(base, scales, dtype, dim, i, q), i);
  }
  ent_t* g = (ent_t*)malloc(sizeof(ent_t) * k); uint32_t gc = 0;
  for (int t = 0; t < nthreads; ++t)
    for (uint32_t j = 0; j < cnts[t]; ++j) topk_consider(g, &gc, k, lists[(size_t)t * k + j].s, lists[(size_t)t * k + j].id);
  for (uint32_t j = 0; j < gc; ++j) { out_ids[j] = g[j].id; out_scores[j] = g[j].s; }
  free(g); free(lists); free(cnts);
  return gc;
}
