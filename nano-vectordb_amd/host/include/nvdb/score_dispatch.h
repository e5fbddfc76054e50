// Score of one (query, row) pair for any base dtype, and the once-per-search dtype check.
// Surface of reference include/nvdb/score_dispatch.h:13-20 (ensure_supported_base_dtype) and :25-48
// (score_query_base_at): header-only there and here, so that the bench-side batched loops
// (reference apps/nvdb_bench.cpp:47-251) inline the dispatch into their row loop.
#pragma once
#include <cstdint>
#include <stdexcept>

#include "nvdb/simd_dot.h"
#include "nvdb/vecbin_format.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

inline void ensure_supported_base_dtype(const VectorDataset& base) {
  if (bytes_per_elem(base.dtype()) == 0) throw std::runtime_error("Unsupported base dtype (Float32/Float16/Int8 only)");
}

inline float score_query_base_at(const VectorDataset& base, const float* q_f32, uint64_t row_id, uint32_t dim, uint32_t base_dtype) {
  switch (static_cast<DType>(base_dtype)) {
    case DType::Float32: return dot_f32(q_f32, base.vector_ptr_f32(row_id), dim);
    case DType::Float16: return dot_f32_f16base(q_f32, base.vector_ptr_f16(row_id), dim);
    case DType::Int8: return dot_f32_i8base(q_f32, base.vector_ptr_i8(row_id), dim, *base.scale_ptr_i8(row_id));
  }
  throw std::runtime_error("Unsupported base dtype in score_query_base_at");
}

}  // namespace nvdb
