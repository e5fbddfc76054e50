// CPU dot kernels of the host layer.  One templated AVX2 loop serves the three base dtypes: a
// "widen" functor turns 8 base elements into 8 floats, one FMA per 8 elements into a single 8-lane
// accumulator, then the (lo+hi)/hadd/hadd reduction -- the arithmetic order of reference
// src/simd_dot.cpp:26-49 / 102-124 / 160-199, so results are bit-identical to it (pinned by
// tests/test_host_cpp.py against the goldens).  Tails follow oracle/nvdb_oracle.c.
#include "nvdb/simd_dot.h"

#include <immintrin.h>

#include <atomic>
#include <cmath>
#include <cstring>

namespace nvdb {

namespace {
std::atomic<bool> g_scalar{false};

inline float half_to_float(uint16_t h) { return _cvtsh_ss(h); }

__attribute__((target("avx2,fma"))) inline float reduce8(__m256 acc) {
  __m128 s = _mm_add_ps(_mm256_castps256_ps128(acc), _mm256_extractf128_ps(acc, 1));
  s = _mm_hadd_ps(s, s);
  s = _mm_hadd_ps(s, s);
  return _mm_cvtss_f32(s);
}

struct WidenF32 { __attribute__((target("avx2"))) static __m256 at(const float* x, uint32_t i) { return _mm256_loadu_ps(x + i); } };
struct WidenF16 { __attribute__((target("avx2,f16c"))) static __m256 at(const uint16_t* x, uint32_t i) { return _mm256_cvtph_ps(_mm_loadu_si128(reinterpret_cast<const __m128i*>(x + i))); } };
struct WidenI8 { __attribute__((target("avx2"))) static __m256 at(const int8_t* x, uint32_t i) { return _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_loadl_epi64(reinterpret_cast<const __m128i*>(x + i)))); } };

template <class W, class T>
__attribute__((target("avx2,fma,f16c"))) float body(const float* q, const T* x, uint32_t upto) {
  __m256 acc = _mm256_setzero_ps();
  for (uint32_t i = 0; i < upto; i += 8) acc = _mm256_fmadd_ps(_mm256_loadu_ps(q + i), W::at(x, i), acc);
  return reduce8(acc);
}
}  // namespace

void set_force_scalar(bool v) { g_scalar.store(v, std::memory_order_relaxed); }

float dot_f32(const float* a, const float* b, uint32_t dim) {
  if (g_scalar.load(std::memory_order_relaxed)) {
    double s = 0.0;
    for (uint32_t i = 0; i < dim; ++i) s = std::fma(static_cast<double>(a[i]), static_cast<double>(b[i]), s);
    return static_cast<float>(s);
  }
  uint32_t i = dim & ~7u;
  float out = body<WidenF32>(a, b, i);
  if (dim - i >= 4) {   // the reference's tail as its compiler builds it: four unfused, then fused
    for (int j = 0; j < 4; ++j) { volatile float p = a[i + j] * b[i + j]; out = out + p; }
    i += 4;
  }
  for (; i < dim; ++i) out = std::fmaf(a[i], b[i], out);
  return out;
}

float dot_f32_f16base(const float* q, const uint16_t* x, uint32_t dim) {
  uint32_t i = dim & ~7u;
  float out = body<WidenF16>(q, x, i);
  for (; i < dim; ++i) out = std::fmaf(q[i], half_to_float(x[i]), out);
  return out;
}

float dot_f32_i8base(const float* q, const int8_t* x, uint32_t dim, float scale) {
  if (g_scalar.load(std::memory_order_relaxed)) {
    double s = 0.0;
    for (uint32_t i = 0; i < dim; ++i) s = std::fma(static_cast<double>(q[i]), static_cast<double>(x[i]), s);
    return static_cast<float>(s * static_cast<double>(scale));
  }
  uint32_t i = dim & ~15u;
  float out = body<WidenI8>(q, x, i);
  for (; i < dim; ++i) out = std::fmaf(q[i], static_cast<float>(x[i]), out);
  return out * scale;
}

}  // namespace nvdb
