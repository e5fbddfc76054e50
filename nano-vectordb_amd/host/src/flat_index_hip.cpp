#include "nvdb/flat_index_hip.h"

#include <algorithm>
#include <stdexcept>
#include <string>

#include "nvdb/detail/call_coalescer.h"
#include "nvdb_hip.h"

namespace nvdb {


FlatIndexHIP::FlatIndexHIP(const VectorDataset* base, int device, uint64_t row_base) {
  if (!base || base->count() == 0) throw std::runtime_error("Empty base");
  if (bytes_per_elem(base->dtype()) == 0) throw std::runtime_error("Unsupported base dtype (Float32/Float16/Int8 only)");
  if (nvdb_hip_create(device, &ctx_) != NVDB_OK) throw std::runtime_error(std::string(nvdb_hip_last_error(nullptr)));
  n_ = base->count();
  dim_ = base->dim();
  const nvdb_status s = nvdb_hip_upload_corpus(ctx_, base->payload_ptr(), base->scales_ptr(), n_, dim_, base->dtype(), row_base);
  if (s != NVDB_OK) {
    const std::string msg = nvdb_hip_last_error(ctx_);
    nvdb_hip_destroy(ctx_);
    ctx_ = nullptr;
    throw std::runtime_error(msg);
  }
  // one batch at a time on the context (the error text lives in the context: read before the next batch may start)
  calls_.reset(new detail::CallCoalescer(dim_, [this](const float* q, uint32_t nq, uint32_t k, uint64_t* ids, float* sc, uint32_t* keff) {
    nvdb_hip_timing t;
    const nvdb_status st = nvdb_hip_search_batch(ctx_, q, nq, k, ids, sc, keff, &t);
    if (st != NVDB_OK) return std::string(nvdb_hip_last_error(ctx_));
    last_kernel_ms_ = t.kernel_ms;
    return std::string();
  }));
}

FlatIndexHIP::~FlatIndexHIP() { nvdb_hip_destroy(ctx_); }

std::vector<SearchResult> FlatIndexHIP::search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const {
  if (!queries) throw std::runtime_error("Null query");
  if (k == 0 || nq == 0) return {};
  std::vector<uint64_t> ids(static_cast<size_t>(nq) * k);
  std::vector<float> sc(static_cast<size_t>(nq) * k);
  const uint32_t keff = calls_->search(queries, nq, k, ids.data(), sc.data());
  std::vector<SearchResult> out(static_cast<size_t>(nq) * keff);
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t j = 0; j < keff; ++j) out[static_cast<size_t>(q) * keff + j] = SearchResult{ids[static_cast<size_t>(q) * k + j], sc[static_cast<size_t>(q) * k + j]};
  return out;
}

std::vector<SearchResult> FlatIndexHIP::search_topk_dot(const float* q, uint32_t k) const { return search_topk_dot_batch(q, 1, k); }

// ---- row-sharded over several GPUs: the C ABI's device group ---------------------------------------------------
FlatIndexHIPSharded::FlatIndexHIPSharded(const VectorDataset* base, const std::vector<int>& devices) {
  if (!base || base->count() == 0) throw std::runtime_error("Empty base");
  if (devices.empty()) throw std::runtime_error("FlatIndexHIPSharded: no devices");
  if (bytes_per_elem(base->dtype()) == 0) throw std::runtime_error("Unsupported base dtype (Float32/Float16/Int8 only)");
  n_ = base->count();
  dim_ = base->dim();
  const size_t G = std::min<size_t>(devices.size(), n_);          // never more shards than rows
  if (nvdb_hip_group_create(devices.data(), static_cast<uint32_t>(G), &grp_) != NVDB_OK)
    throw std::runtime_error(std::string(nvdb_hip_group_last_error(nullptr)));
  if (nvdb_hip_group_upload_corpus(grp_, base->payload_ptr(), base->scales_ptr(), n_, dim_, base->dtype()) != NVDB_OK) {
    const std::string msg = nvdb_hip_group_last_error(grp_);
    nvdb_hip_group_destroy(grp_);
    grp_ = nullptr;
    throw std::runtime_error(msg);
  }
  calls_.reset(new detail::CallCoalescer(dim_, [this](const float* q, uint32_t nq, uint32_t k, uint64_t* ids, float* sc, uint32_t* keff) {
    nvdb_hip_group_stats gs{};
    if (nvdb_hip_group_search_batch(grp_, q, nq, k, ids, sc, keff, &gs) != NVDB_OK) return std::string(nvdb_hip_group_last_error(grp_));
    fallbacks_ += gs.host_merge_fallbacks;
    return std::string();
  }));
}

FlatIndexHIPSharded::~FlatIndexHIPSharded() { nvdb_hip_group_destroy(grp_); }

size_t FlatIndexHIPSharded::shards() const { return nvdb_hip_group_size(grp_); }
bool FlatIndexHIPSharded::exchange_is_rccl() const { return nvdb_hip_group_exchange(grp_, nullptr) == 1; }
const char* FlatIndexHIPSharded::exchange_note() const {
  const char* why = "";
  (void)nvdb_hip_group_exchange(grp_, &why);
  return why;
}

std::vector<SearchResult> FlatIndexHIPSharded::search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const {
  if (!queries) throw std::runtime_error("Null query");
  if (k == 0 || nq == 0) return {};
  const size_t per = static_cast<size_t>(nq) * k;
  std::vector<uint64_t> ids(per);
  std::vector<float> sc(per);
  const uint32_t keff = calls_->search(queries, nq, k, ids.data(), sc.data());
  std::vector<SearchResult> out(static_cast<size_t>(nq) * keff);
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t j = 0; j < keff; ++j) out[static_cast<size_t>(q) * keff + j] = SearchResult{ids[static_cast<size_t>(q) * k + j], sc[static_cast<size_t>(q) * k + j]};
  return out;
}

}  // namespace nvdb
