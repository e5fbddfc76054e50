#include "nvdb/vector_dataset.h"

#include <cstring>
#include <stdexcept>

namespace nvdb {

namespace {
struct Raw12 { uint32_t count, reserved, dim; };   // legacy header: always float32 (reference src/vector_dataset.cpp:11-22)
}

void VectorDataset::load(const std::string& path) {
  mm_.open_readonly(path);
  count_ = 0; dim_ = 0; dtype_ = 1;
  payload_ = nullptr; f32_ = nullptr; f16_ = nullptr; i8_ = nullptr; scales_ = nullptr;
  const uint8_t* p = mm_.data();
  const size_t sz = mm_.size();

  // vecbin64 first: accepted only if magic, version, dtype, dim, count are all sane; then the file size must
  // match exactly (reference src/vector_dataset.cpp:39-70)
  if (sz >= sizeof(VecbinHeader)) {
    VecbinHeader h;
    std::memcpy(&h, p, sizeof(h));
    if (h.magic == kMagic && h.version == kVersion && h.dim > 0 && h.count > 0 && bytes_per_elem(h.dtype) != 0) {
      if (sz != sizeof(VecbinHeader) + bytes_for_payload_and_aux(h.count, h.dim, h.dtype))
        throw std::runtime_error("VecbinHeader ok but file size mismatch");
      count_ = h.count; dim_ = h.dim; dtype_ = h.dtype;
      payload_ = p + sizeof(VecbinHeader);
      if (dtype_ == 1) f32_ = static_cast<const float*>(payload_);
      else if (dtype_ == 2) f16_ = static_cast<const uint16_t*>(payload_);
      else {
        i8_ = static_cast<const int8_t*>(payload_);
        scales_ = reinterpret_cast<const float*>(p + sizeof(VecbinHeader) + bytes_for_vectors_typed(count_, dim_, dtype_));
      }
      return;
    }
  }
  // raw12 fallback: {u32 count, u32 0, u32 dim} + float32 payload, exact size (reference :97-118)
  if (sz < sizeof(Raw12)) throw std::runtime_error("File too small (neither vecbin64 nor raw12)");
  Raw12 r;
  std::memcpy(&r, p, sizeof(r));
  if (r.count == 0 || r.dim == 0) throw std::runtime_error("raw12 header invalid (count/dim == 0)");
  if (sz != sizeof(Raw12) + static_cast<size_t>(r.count) * r.dim * sizeof(float))
    throw std::runtime_error("raw12 header parsed but file size mismatch");
  count_ = r.count; dim_ = r.dim; dtype_ = 1;
  payload_ = p + sizeof(Raw12);
  f32_ = static_cast<const float*>(payload_);
}

void VectorDataset::check(uint64_t i, uint32_t want, const char* what) const {
  if (dtype_ != want) throw std::runtime_error(what);
  if (i >= count_) throw std::runtime_error("Index out of range");
}

const float* VectorDataset::vector_ptr_f32(uint64_t i) const { check(i, 1, "Dataset is not float32"); return f32_ + i * dim_; }
const uint16_t* VectorDataset::vector_ptr_f16(uint64_t i) const { check(i, 2, "Dataset is not float16"); return f16_ + i * dim_; }
const int8_t* VectorDataset::vector_ptr_i8(uint64_t i) const { check(i, 3, "Dataset is not int8"); return i8_ + i * dim_; }
const float* VectorDataset::scale_ptr_i8(uint64_t i) const { check(i, 3, "Dataset is not int8 (scales missing)"); return scales_ + i; }

}  // namespace nvdb
