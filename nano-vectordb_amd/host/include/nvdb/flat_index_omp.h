// Exact flat scan on the CPU, OpenMP: static row partition, one best-k list per thread, serial merge.
// Surface of reference include/nvdb/flat_index_omp.h:11-19; behaviour of src/flat_index_omp.cpp:16-85.
// This is the "AVX2+OMP CPU path" the GPU numbers are quoted beside.
#pragma once
#include <cstdint>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

class FlatIndexOMP {
 public:
  explicit FlatIndexOMP(const VectorDataset* base) : base_(base) {}
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const;   // same contract as FlatIndex
 private:
  const VectorDataset* base_;
};

}  // namespace nvdb
