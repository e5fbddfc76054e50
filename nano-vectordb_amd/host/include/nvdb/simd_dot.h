// CPU dot kernels (host side of the product: the st/omp modes of nvdb_bench, ground-truth tools).
// Same results, bit for bit, as reference src/simd_dot.cpp: AVX2+FMA(+F16C) bodies where the CPU has them
// (runtime dispatch), the reference's double-accumulating scalar path elsewhere.
#pragma once
#include <cstdint>

namespace nvdb {
float dot_f32(const float* a, const float* b, uint32_t dim);
float dot_f32_f16base(const float* q_f32, const uint16_t* x_f16, uint32_t dim);
float dot_f32_i8base(const float* q_f32, const int8_t* x_i8, uint32_t dim, float scale);
void set_force_scalar(bool v);

// additions (not in the reference header)
bool simd_dot_available();                                                                  // avx2 + fma + f16c present
float dot_f32_f16base_scalar(const float* q_f32, const uint16_t* x_f16, uint32_t dim);      // the fp16 kernel's fallback, callable on any host
}  // namespace nvdb
