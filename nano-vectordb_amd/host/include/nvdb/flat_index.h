// Exact flat scan on the CPU, one thread.  Surface of reference include/nvdb/flat_index.h:9-18
// (constructor from a borrowed dataset, const re-entrant search_topk_dot returning best-first results).
#pragma once
#include <cstdint>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

class FlatIndex {
 public:
  explicit FlatIndex(const VectorDataset* base) : base_(base) {}
  // throws "Empty base"; k == 0 -> empty; k > count is clamped (reference src/flat_index.cpp:17-24)
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const;
 private:
  const VectorDataset* base_;
};

}  // namespace nvdb
