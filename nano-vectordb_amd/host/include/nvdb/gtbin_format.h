// gtbin: ground-truth ids, 64-byte header + uint32 ids[Q*k]  (reference include/nvdb/gtbin_format.h:7-35).
#pragma once
#include <cstddef>
#include <cstdint>

namespace nvdb {

static constexpr uint64_t kGtMagic = 0x4E56444247543031ULL;  // "NVDBGT01"
static constexpr uint32_t kGtVersion = 1;
enum class GtMetric : uint32_t { DotEquivalentL2 = 1 };

#pragma pack(push, 1)
struct GtBinHeader {
  uint64_t magic;
  uint32_t version, metric, k, dim;
  uint64_t Q, N;
  uint8_t reserved[24];
};
#pragma pack(pop)
static_assert(sizeof(GtBinHeader) == 64, "gtbin header is 64 bytes");

inline size_t gt_payload_bytes(uint64_t Q, uint32_t k) { return static_cast<size_t>(Q) * k * sizeof(uint32_t); }

}  // namespace nvdb
