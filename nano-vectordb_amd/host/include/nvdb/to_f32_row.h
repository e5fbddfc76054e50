// One corpus row widened to float32, whatever the base dtype.  Surface of reference include/nvdb/to_f32_row.h:10-34:
// fp32 copied, fp16 through f16_to_f32_scalar, int8 as float(v) * scale (one fp32 multiply per element).
#pragma once
#include <cstdint>
#include <stdexcept>

#include "nvdb/f16_scalar.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

inline void base_row_to_f32(const VectorDataset& base, uint64_t row, float* out) {
  const uint32_t dim = base.dim();
  switch (static_cast<DType>(base.dtype())) {
    case DType::Float32: {
      const float* src = base.vector_ptr_f32(row);
      for (uint32_t j = 0; j < dim; ++j) out[j] = src[j];
      return;
    }
    case DType::Float16: {
      const uint16_t* src = base.vector_ptr_f16(row);
      for (uint32_t j = 0; j < dim; ++j) out[j] = f16_to_f32_scalar(src[j]);
      return;
    }
    case DType::Int8: {
      const int8_t* src = base.vector_ptr_i8(row);
      const float scale = *base.scale_ptr_i8(row);
      for (uint32_t j = 0; j < dim; ++j) out[j] = static_cast<float>(src[j]) * scale;
      return;
    }
  }
  throw std::runtime_error("Unsupported dtype in base_row_to_f32");
}

}  // namespace nvdb
