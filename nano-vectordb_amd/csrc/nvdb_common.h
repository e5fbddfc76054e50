// nvdb_common.h -- definitions shared by host code and gfx950 device code.
//
// * the synthetic corpus generator: integer-only hash -> sum of four
//   16-bit uniforms (Irwin-Hall, approx. Gaussian) -> L2 normalisation in double.  No libm
//   transcendental is involved, so the CPU and the GPU produce identical bits.
// * IEEE half <-> float conversions for the host side (the device uses v_cvt_* directly).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define NVDB_HD __host__ __device__ inline
#else
#define NVDB_HD inline
#endif

namespace nvdbhip {

NVDB_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

// key of a row: depends on (seed,row) only, computed once per row
NVDB_HD uint32_t synth_row_key(uint64_t seed, uint64_t row) {
  uint32_t k = mix32(static_cast<uint32_t>(seed) ^ 0x9E3779B9u) ^ mix32(static_cast<uint32_t>(seed >> 32) + 0x7F4A7C15u);
  k = mix32(k ^ static_cast<uint32_t>(row));
  k = mix32(k + static_cast<uint32_t>(row >> 32) * 0xC2B2AE35u + 0x27D4EB2Fu);
  return k;
}

// un-normalised element: integer in [-131070, 131070], mean 0, approx. normal
NVDB_HD int32_t synth_raw(uint32_t row_key, uint32_t col) {
  const uint32_t h1 = mix32(row_key ^ (col * 0x9E3779B1u + 0x165667B1u));
  const uint32_t h2 = mix32((h1 + 0x6C8E9CF5u) ^ (col * 0x85EBCA77u));
  return static_cast<int32_t>(h1 & 0xFFFFu) + static_cast<int32_t>(h1 >> 16) +
         static_cast<int32_t>(h2 & 0xFFFFu) + static_cast<int32_t>(h2 >> 16) - 131070;
}

// normalised fp32 element given the row's exact integer sum of squares
NVDB_HD double synth_inv_norm(uint64_t sumsq) { return sumsq ? 1.0 / sqrt(static_cast<double>(sumsq)) : 0.0; }
NVDB_HD float synth_elem(int32_t raw, double inv_norm) { return static_cast<float>(static_cast<double>(raw) * inv_norm); }

// ---- host-side IEEE half conversion (round-to-nearest-even; f32 subnormals -> signed zero) -------
inline uint16_t f32_to_f16_rne(float f) {
  uint32_t x; std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u, be = (x >> 23) & 0xFFu;
  uint32_t mant = x & 0x7FFFFFu;
  if (be == 0xFF) return static_cast<uint16_t>(sign | 0x7C00u | (mant ? (0x200u | (mant >> 13)) : 0u));
  if (be == 0) return static_cast<uint16_t>(sign);
  const int e = static_cast<int>(be) - 127;
  mant |= 0x800000u;
  if (e > 15) return static_cast<uint16_t>(sign | 0x7C00u);
  if (e < -14) {
    const int shift = -14 - e;                    // 1..: result is a half subnormal (or zero)
    if (shift > 24) return static_cast<uint16_t>(sign);
    uint32_t m = mant >> (shift + 13);
    const uint32_t rem = mant & ((1u << (shift + 13)) - 1u), half = 1u << (shift + 12);
    if (rem > half || (rem == half && (m & 1u))) ++m;
    return static_cast<uint16_t>(sign | m);       // a carry into bit 10 is the smallest normal
  }
  uint32_t he = static_cast<uint32_t>(e + 15), m = mant >> 13;
  const uint32_t rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (m & 1u))) ++m;
  if (m & 0x800u) { m >>= 1; ++he; }
  if (he >= 0x1F) return static_cast<uint16_t>(sign | 0x7C00u);
  return static_cast<uint16_t>(sign | (he << 10) | (m & 0x3FFu));
}

inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, out;
  if (e == 0) {
    if (!m) out = sign;
    else { int ex = -14; while (!(m & 0x400u)) { m <<= 1; --ex; } out = sign | (static_cast<uint32_t>(ex + 127) << 23) | ((m & 0x3FFu) << 13); }
  } else if (e == 31) out = sign | 0x7F800000u | (m << 13);
  else out = sign | ((e + 112u) << 23) | (m << 13);
  float f; std::memcpy(&f, &out, 4); return f;
}

// per-row int8 quantisation rule (apps/nvdb_quantize_i8.cpp:71-80): scale = max|x|/127 (1 when the
// row is all zero), q = rint(x * (1/scale)) clamped to [-127,127]
inline float quantize_i8_row(const float* row, uint32_t dim, int8_t* out) {
  float mx = 0.f;
  for (uint32_t j = 0; j < dim; ++j) mx = std::fmax(mx, std::fabs(row[j]));
  const float scale = mx > 0.f ? mx / 127.f : 1.f;
  const float inv = 1.0f / scale;
  for (uint32_t j = 0; j < dim; ++j) {
    long q = std::lrint(row[j] * inv);
    q = q > 127 ? 127 : (q < -127 ? -127 : q);
    out[j] = static_cast<int8_t>(q);
  }
  return scale;
}

}  // namespace nvdbhip
