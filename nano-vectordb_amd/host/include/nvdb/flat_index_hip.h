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

// Row-sharded flat index over several GPUs of one node (the reference has no multi-GPU path): shard g holds the
// contiguous rows [g*N/G, (g+1)*N/G) on devices[g] with global ids; a batch is searched on all shards concurrently
// (one host thread per GPU) and the per-shard top-k lists are merged on the host in the canonical order, which makes
// the result identical to the single-GPU result.  (bench.py does the same across processes with an RCCL all-gather.)
class FlatIndexHIPSharded {
 public:
  FlatIndexHIPSharded(const VectorDataset* base, const std::vector<int>& devices);
  ~FlatIndexHIPSharded();
  FlatIndexHIPSharded(const FlatIndexHIPSharded&) = delete;
  FlatIndexHIPSharded& operator=(const FlatIndexHIPSharded&) = delete;

  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const { return search_topk_dot_batch(q, 1, k); }
  std::vector<SearchResult> search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const;
  size_t shards() const { return ctx_.size(); }

 private:
  std::vector<nvdb_hip_ctx*> ctx_;
  uint64_t n_ = 0;
  uint32_t dim_ = 0;
};

}  // namespace nvdb
