// Zero-copy typed view of a vecbin64 / raw12 file.  Surface of reference include/nvdb/vector_dataset.h:10-52.
#pragma once
#include <cstdint>
#include <string>

#include "nvdb/mmap_file.h"
#include "nvdb/vecbin_format.h"

namespace nvdb {

class VectorDataset {
 public:
  void load(const std::string& path);            // throws std::runtime_error on any inconsistency

  uint64_t count() const { return count_; }
  uint32_t dim() const { return dim_; }
  uint32_t dtype() const { return dtype_; }      // 1/2/3 as in the vecbin header; raw12 is always 1

  const float* vector_ptr_f32(uint64_t i) const;
  const uint16_t* vector_ptr_f16(uint64_t i) const;
  const int8_t* vector_ptr_i8(uint64_t i) const;
  const float* scale_ptr_i8(uint64_t i) const;
  const float* vector_ptr(uint64_t i) const { return vector_ptr_f32(i); }

  // start of the row-major payload.  NOTE: the reference returns nullptr for int8 here
  // (src/vector_dataset.cpp:152-157); kept, use data_ptr_i8()/scales_ptr() for int8.
  const void* data_ptr() const { return dtype_ == 1 ? static_cast<const void*>(f32_) : (dtype_ == 2 ? static_cast<const void*>(f16_) : nullptr); }
  const float* data_ptr_f32() const { return f32_; }
  const uint16_t* data_ptr_f16() const { return f16_; }
  const int8_t* data_ptr_i8() const { return i8_; }      // addition
  const float* scales_ptr() const { return scales_; }    // addition
  const void* payload_ptr() const { return payload_; }   // addition: any dtype

 private:
  void check(uint64_t i, uint32_t want, const char* what) const;
  MmapFile mm_;
  uint64_t count_ = 0;
  uint32_t dim_ = 0, dtype_ = 1;
  const void* payload_ = nullptr;
  const float* f32_ = nullptr;
  const uint16_t* f16_ = nullptr;
  const int8_t* i8_ = nullptr;
  const float* scales_ = nullptr;
};

}  // namespace nvdb
