// GPU flat index: the MI355X drop-in for FlatIndex / FlatIndexOMP (same constructor argument, same
// search_topk_dot signature and return type) plus the batched entry the reference does in its bench
// app (apps/nvdb_bench.cpp:47-159).  Thin C++ over the C ABI in include/nvdb_hip.h; every non-zero
// status becomes std::runtime_error (reference convention, src/flat_index.cpp:17).
#pragma once
#include <cstdint>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

struct nvdb_hip_ctx;

namespace nvdb {

class FlatIndexHIP {
 public:
  // Uploads the dataset's rows to HBM once (device `device`); ids returned are row_base + row.
  explicit FlatIndexHIP(const VectorDataset* base, int device = 0, uint64_t row_base = 0);
  ~FlatIndexHIP();
  FlatIndexHIP(const FlatIndexHIP&) = delete;
  FlatIndexHIP& operator=(const FlatIndexHIP&) = delete;

  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const;
  // nq queries [nq][dim]; result [nq][min(k,N)] row-major, best first
  std::vector<SearchResult> search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const;
  double last_kernel_ms() const { return last_kernel_ms_; }
  nvdb_hip_ctx* context() const { return ctx_; }

 private:
  nvdb_hip_ctx* ctx_ = nullptr;
  uint64_t n_ = 0;
  uint32_t dim_ = 0;
  mutable double last_kernel_ms_ = 0.0;
};

}  // namespace nvdb
