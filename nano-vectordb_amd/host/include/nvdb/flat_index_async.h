// Exact flat scan on per-call worker threads.  Surface of reference include/nvdb/flat_index_async.h:9-21;
// same arithmetic as FlatIndex, contiguous ceil(n/threads) row blocks per worker (reference
// src/flat_index_async.cpp:33), the caller scans block 0 itself.
#pragma once
#include <cstdint>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

class FlatIndexAsync {
 public:
  explicit FlatIndexAsync(const VectorDataset* base) : base_(base) {}
  // re-entrant; threads < 1 is treated as 1
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k, int threads) const;
 private:
  const VectorDataset* base_;
};

}  // namespace nvdb
