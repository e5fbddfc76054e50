// Exact flat scan on the CPU, single thread and OpenMP.  Surface of reference
// include/nvdb/flat_index.h:9-18 and include/nvdb/flat_index_omp.h:11-19.
#pragma once
#include <cstdint>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

class FlatIndex {
 public:
  explicit FlatIndex(const VectorDataset* base) : base_(base) {}
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const;
 private:
  const VectorDataset* base_;
};

class FlatIndexOMP {
 public:
  explicit FlatIndexOMP(const VectorDataset* base) : base_(base) {}
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const;
 private:
  const VectorDataset* base_;
};

// score of one row for any base dtype (reference include/nvdb/score_dispatch.h:25-48)
float score_query_base_at(const VectorDataset& base, const float* q_f32, uint64_t row_id, uint32_t dim, uint32_t base_dtype);
void ensure_supported_base_dtype(const VectorDataset& base);

}  // namespace nvdb
