// GPU flat index: the MI355X drop-in for FlatIndex / FlatIndexOMP (same constructor argument, same
// search_topk_dot signature and return type) plus the batched entry the reference does in its bench
// app (apps/nvdb_bench.cpp:47-159).  Thin C++ over the C ABI in include/nvdb_hip.h; every non-zero
// status becomes std::runtime_error (reference convention, src/flat_index.cpp:17).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

struct nvdb_hip_ctx;
struct nvdb_hip_group;

namespace nvdb {

namespace detail { class CallCoalescer; }        // flat_index_hip.cpp: concurrent callers of one index share GPU batches

class FlatIndexHIP {
 public:
  // Uploads the dataset's rows to HBM once (device `device`); ids returned are row_base + row.
  explicit FlatIndexHIP(const VectorDataset* base, int device = 0, uint64_t row_base = 0);
  ~FlatIndexHIP();
  FlatIndexHIP(const FlatIndexHIP&) = delete;
  FlatIndexHIP& operator=(const FlatIndexHIP&) = delete;

  // Like FlatIndex / FlatIndexOMP::search_topk_dot (const, callable from several threads at once: reference
  // include/nvdb/flat_index.h:11-16).  The device context behind it is single-owner and a batch of 8 costs the GPU what one
  // query costs, so callers that overlap are COALESCED: whoever finds the context free becomes the leader, gathers the
  // queries that queued meanwhile (same k, up to 1024 in all; after a shared batch it lingers <= 50 us for the same callers
  // to come back), runs ONE nvdb_hip_search_batch and hands every caller its rows.  A lone caller never waits.  Results are
  // per query and do not depend on the batch they rode in.
  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const;
  // nq queries [nq][dim]; result [nq][min(k,N)] row-major, best first
  std::vector<SearchResult> search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const;
  double last_kernel_ms() const { return last_kernel_ms_; }       // of the calling thread's last search only when callers do not overlap
  nvdb_hip_ctx* context() const { return ctx_; }

 private:
  nvdb_hip_ctx* ctx_ = nullptr;
  uint64_t n_ = 0;
  uint32_t dim_ = 0;
  mutable double last_kernel_ms_ = 0.0;
  std::unique_ptr<detail::CallCoalescer> calls_;   // one GPU batch at a time on the context, shared by the callers that overlap
};

// Row-sharded flat index over several GPUs of one node, driven from ONE process (the reference has no multi-GPU path):
// shard g holds the contiguous rows [g*N/G, (g+1)*N/G) on devices[g] with global ids.  Thin C++ over the device group of
// the C ABI (nvdb_hip_group_*, include/nvdb_hip.h): per batch every shard is searched on its own stream, the per-shard
// top-k blocks are exchanged by an RCCL all-gather over xGMI (peer copies when a device is listed twice), and the k-way
// merge runs on devices[0] in the canonical order -- the result equals the single-GPU result bit for bit.
// (bench.py does the same across one process per GPU with torch.distributed's all-gather.)
class FlatIndexHIPSharded {
 public:
  FlatIndexHIPSharded(const VectorDataset* base, const std::vector<int>& devices);
  ~FlatIndexHIPSharded();
  FlatIndexHIPSharded(const FlatIndexHIPSharded&) = delete;
  FlatIndexHIPSharded& operator=(const FlatIndexHIPSharded&) = delete;

  std::vector<SearchResult> search_topk_dot(const float* q, uint32_t k) const { return search_topk_dot_batch(q, 1, k); }
  std::vector<SearchResult> search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const;
  size_t shards() const;
  bool exchange_is_rccl() const;                       // false: peer copies (why: exchange_note())
  const char* exchange_note() const;
  unsigned host_merge_fallbacks() const { return fallbacks_; }   // sub-batches that went through the host merge so far

 private:
  nvdb_hip_group* grp_ = nullptr;
  uint64_t n_ = 0;
  uint32_t dim_ = 0;
  mutable unsigned fallbacks_ = 0;
  std::unique_ptr<detail::CallCoalescer> calls_;   // one batch at a time on the group, shared by the callers that overlap
};

}  // namespace nvdb
