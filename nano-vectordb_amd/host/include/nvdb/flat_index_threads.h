// The two remaining CPU threadings of the exact flat scan: per-call worker threads (FlatIndexAsync) and a
// persistent pinned worker team (FlatIndexPool).  Surface of reference include/nvdb/flat_index_async.h:9-21
// and include/nvdb/flat_index_pool.h:13-45; same arithmetic as FlatIndex (src/flat_index.cpp here), same
// contiguous ceil(n/threads) row blocks per worker (reference src/flat_index_async.cpp:33, flat_index_pool.cpp:155).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>
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

class FlatIndexPool {
 public:
  FlatIndexPool(const VectorDataset* base, int threads);   // throws "Empty base"
  ~FlatIndexPool();
  FlatIndexPool(const FlatIndexPool&) = delete;
  FlatIndexPool& operator=(const FlatIndexPool&) = delete;
  // one query in flight at a time (as the reference, flat_index_pool.cpp:194-215); throws "Null query"
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k);
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
