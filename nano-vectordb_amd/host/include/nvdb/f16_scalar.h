// IEEE 754 half (bit pattern) -> float, portable scalar code.  Surface of reference include/nvdb/f16_scalar.h:8-38
// (used by to_f32_row.h, the int8 quantiser and the scalar dot fallback); every one of the 65 536 patterns is
// pinned against the reference's function through tests/golden/refine_conv_golden.npz.
#pragma once
#include <cstdint>
#include <cstring>

namespace nvdb {

inline float f16_to_f32_scalar(uint16_t h) {
  const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t bits;
  if (e == 31u) bits = 0x7F800000u | (m << 13);                 // infinity / NaN: payload kept in the top mantissa bits
  else if (e != 0u) bits = ((e + 112u) << 23) | (m << 13);      // normal: exponent bias 15 -> 127
  else if (m == 0u) bits = 0u;                                  // zero
  else {                                                        // subnormal half = m * 2^-24: normalise
    const uint32_t lz = static_cast<uint32_t>(__builtin_clz(m));   // top set bit at position 31 - lz (0..9)
    bits = ((134u - lz) << 23) | ((m << (lz - 8u)) & 0x007FFFFFu);
  }
  bits |= static_cast<uint32_t>(h & 0x8000u) << 16;
  float f;
  std::memcpy(&f, &bits, sizeof(f));
  return f;
}

}  // namespace nvdb
