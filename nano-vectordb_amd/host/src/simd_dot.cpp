// CPU dot kernels of the host layer.
//
// Dispatch follows the reference (src/simd_dot.cpp:52-64, 127-136, 202-213): the AVX2 bodies run only where
// __builtin_cpu_supports says the CPU has avx2 + fma (+ f16c for the fp16 kernel); anywhere else -- and, for
// fp32 / int8, whenever set_force_scalar(true) -- the double-accumulating scalar path runs.  Like the
// reference, the fp16 kernel ignores the force-scalar switch.  Nothing in this file needs -mavx2 on the
// command line: the SIMD code lives in functions carrying their own target attribute.
//
// SIMD path: one templated AVX2 loop serves the three base dtypes: a "widen" functor turns 8 base elements
// into 8 floats, one FMA per 8 elements into a single 8-lane accumulator, then the (lo+hi)/hadd/hadd
// reduction -- the arithmetic order of reference src/simd_dot.cpp:26-49 / 102-124 / 160-199, so results are
// bit-identical to it (pinned against the goldens, tests/test_host_capi.py).  Tails follow oracle/nvdb_oracle.c.
#include "nvdb/simd_dot.h"

#include <immintrin.h>

#include <atomic>
#include <cmath>
#include <cstring>

#include "nvdb/f16_scalar.h"

namespace nvdb {

namespace {
std::atomic<bool> g_scalar{false};

#define NVDB_SIMD __attribute__((target("avx2,fma,f16c")))

NVDB_SIMD inline float reduce8(__m256 acc) {
  __m128 s = _mm_add_ps(_mm256_castps256_ps128(acc), _mm256_extractf128_ps(acc, 1));
  s = _mm_hadd_ps(s, s);
  s = _mm_hadd_ps(s, s);
  return _mm_cvtss_f32(s);
}

struct WidenF32 { NVDB_SIMD static __m256 at(const float* x, uint32_t i) { return _mm256_loadu_ps(x + i); } };
struct WidenF16 { NVDB_SIMD static __m256 at(const uint16_t* x, uint32_t i) { return _mm256_cvtph_ps(_mm_loadu_si128(reinterpret_cast<const __m128i*>(x + i))); } };
struct WidenI8 { NVDB_SIMD static __m256 at(const int8_t* x, uint32_t i) { return _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_loadl_epi64(reinterpret_cast<const __m128i*>(x + i)))); } };

template <class W, class T>
NVDB_SIMD float body(const float* q, const T* x, uint32_t upto) {
  __m256 acc = _mm256_setzero_ps();
  // ONE accumulator chain (the reference's order); unrolled only to spend fewer loop uops per row, so that the
  // out-of-order window reaches further into the next row's chain (the chain is FMA-latency-bound at d=768)
#pragma GCC unroll 4
  for (uint32_t i = 0; i < upto; i += 8) acc = _mm256_fmadd_ps(_mm256_loadu_ps(q + i), W::at(x, i), acc);
  return reduce8(acc);
}

// one fused multiply-add in fp32 (vfmadd231ss; std::fmaf would be a libm call without -mfma)
NVDB_SIMD inline float fma1(float a, float b, float c) { return _mm_cvtss_f32(_mm_fmadd_ss(_mm_set_ss(a), _mm_set_ss(b), _mm_set_ss(c))); }

NVDB_SIMD float dot_f32_simd(const float* a, const float* b, uint32_t dim) {
  uint32_t i = dim & ~7u;
  float out = body<WidenF32>(a, b, i);
  if (dim - i >= 4) {   // the reference's tail as its compiler builds it: four unfused, then fused
    for (int j = 0; j < 4; ++j) { volatile float p = a[i + j] * b[i + j]; out = out + p; }
    i += 4;
  }
  for (; i < dim; ++i) out = fma1(a[i], b[i], out);
  return out;
}

NVDB_SIMD float dot_f16_simd(const float* q, const uint16_t* x, uint32_t dim) {
  uint32_t i = dim & ~7u;
  float out = body<WidenF16>(q, x, i);
  for (; i < dim; ++i) out = fma1(q[i], f16_to_f32_scalar(x[i]), out);
  return out;
}

NVDB_SIMD float dot_i8_simd(const float* q, const int8_t* x, uint32_t dim, float scale) {
  uint32_t i = dim & ~15u;
  float out = body<WidenI8>(q, x, i);
  for (; i < dim; ++i) out = fma1(q[i], static_cast<float>(x[i]), out);
  return out * scale;
}

// The scalar fallbacks accumulate double(a) * double(b) in double (src/simd_dot.cpp:18-22, 133-135, 143-149).  A product
// of two floats is exact in double, so whether the compiler fuses the multiply-add or not cannot change a bit.
template <class T, class F>
float dot_scalar(const float* q, const T* x, uint32_t dim, F widen) {
  double s = 0.0;
  for (uint32_t i = 0; i < dim; ++i) s += static_cast<double>(q[i]) * static_cast<double>(widen(x[i]));
  return static_cast<float>(s);
}

bool cpu_has_avx2_fma() {
  static const bool v = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  return v;
}
bool cpu_has_f16c() {
  static const bool v = __builtin_cpu_supports("f16c");
  return v;
}
}  // namespace

void set_force_scalar(bool v) { g_scalar.store(v, std::memory_order_relaxed); }

bool simd_dot_available() { return cpu_has_avx2_fma() && cpu_has_f16c(); }

float dot_f32(const float* a, const float* b, uint32_t dim) {
  if (!g_scalar.load(std::memory_order_relaxed) && cpu_has_avx2_fma()) return dot_f32_simd(a, b, dim);
  return dot_scalar(a, b, dim, [](float v) { return v; });
}

float dot_f32_f16base(const float* q, const uint16_t* x, uint32_t dim) {
  if (cpu_has_avx2_fma() && cpu_has_f16c()) return dot_f16_simd(q, x, dim);      // no force-scalar switch here (reference :127-136)
  return dot_scalar(q, x, dim, [](uint16_t h) { return f16_to_f32_scalar(h); });
}

float dot_f32_i8base(const float* q, const int8_t* x, uint32_t dim, float scale) {
  if (!g_scalar.load(std::memory_order_relaxed) && cpu_has_avx2_fma()) return dot_i8_simd(q, x, dim, scale);
  double s = 0.0;
  for (uint32_t i = 0; i < dim; ++i) s += static_cast<double>(q[i]) * static_cast<double>(x[i]);
  return static_cast<float>(s * static_cast<double>(scale));
}

// fp16 kernel's scalar fallback on its own (what a host without AVX2/F16C computes); test hook
float dot_f32_f16base_scalar(const float* q, const uint16_t* x, uint32_t dim) {
  return dot_scalar(q, x, dim, [](uint16_t h) { return f16_to_f32_scalar(h); });
}

}  // namespace nvdb
