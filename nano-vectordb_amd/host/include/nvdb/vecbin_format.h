// vecbin64 container: 64-byte little-endian header + row-major payload (+ float scales[count] for int8).
// Layout contract: reference include/nvdb/vecbin_format.h:7-59 and scripts/build_vecbin_chunked.py:119-125.
#pragma once
#include <cstddef>
#include <cstdint>

namespace nvdb {

static constexpr uint64_t kMagic = 0x4E56444256454331ULL;  // "NVDBVEC1"
static constexpr uint32_t kVersion = 1;

enum class DType : uint32_t { Float32 = 1, Float16 = 2, Int8 = 3 };

#pragma pack(push, 1)
struct VecbinHeader {
  uint64_t magic;
  uint32_t version;
  uint32_t dtype;
  uint32_t dim;
  uint32_t reserved0;
  uint64_t count;
  uint8_t reserved[32];
};
#pragma pack(pop)
static_assert(sizeof(VecbinHeader) == 64, "vecbin header is 64 bytes");

inline size_t bytes_per_elem(uint32_t dtype) {
  switch (dtype) {
    case 1: return 4;
    case 2: return 2;
    case 3: return 1;
    default: return 0;
  }
}
inline size_t bytes_for_vectors_typed(uint64_t count, uint32_t dim, uint32_t dtype) {
  return static_cast<size_t>(count) * dim * bytes_per_elem(dtype);
}
inline size_t bytes_for_scales(uint64_t count, uint32_t dtype) { return dtype == 3 ? static_cast<size_t>(count) * sizeof(float) : 0; }
inline size_t bytes_for_payload_and_aux(uint64_t count, uint32_t dim, uint32_t dtype) {
  return bytes_for_vectors_typed(count, dim, dtype) + bytes_for_scales(count, dtype);
}

}  // namespace nvdb
