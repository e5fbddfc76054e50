// Exact flat scan on a persistent, pinned worker team.  Surface of reference include/nvdb/flat_index_pool.h:13-45
// (constructor (base, threads), non-copyable, NON-const search: one query in flight at a time,
// src/flat_index_pool.cpp:194-215); contiguous ceil(n/threads) row blocks per worker (:155).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

namespace nvdb {

class FlatIndexPool {
 public:
  FlatIndexPool(const VectorDataset* base, int threads);   // throws "Empty base"
  ~FlatIndexPool();
  FlatIndexPool(const FlatIndexPool&) = delete;
  FlatIndexPool& operator=(const FlatIndexPool&) = delete;
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k);   // throws "Null query"
 private:
  void work(int tid);
  const VectorDataset* base_;
  int threads_;
  std::vector<std::thread> team_;
  std::vector<TopKBuffer> part_;
  std::mutex mu_;
  std::condition_variable go_, done_;
  const float* q_ = nullptr;
  uint32_t k_ = 0;
  uint64_t round_ = 0;        // bumped per query; workers run one scan per value they have not seen
  int pending_ = 0;
  bool quit_ = false;
};

}  // namespace nvdb
