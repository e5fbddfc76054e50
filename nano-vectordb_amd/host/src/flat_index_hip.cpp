#include "nvdb/flat_index_hip.h"

#include <stdexcept>
#include <string>

#include "nvdb_hip.h"

namespace nvdb {

namespace {
void check(nvdb_hip_ctx* c, nvdb_status s) {
  if (s != NVDB_OK) throw std::runtime_error(std::string(nvdb_hip_last_error(c)));
}
}  // namespace

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
}

FlatIndexHIP::~FlatIndexHIP() { nvdb_hip_destroy(ctx_); }

std::vector<SearchResult> FlatIndexHIP::search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const {
  if (!queries) throw std::runtime_error("Null query");
  if (k == 0 || nq == 0) return {};
  std::vector<uint64_t> ids(static_cast<size_t>(nq) * k);
  std::vector<float> sc(static_cast<size_t>(nq) * k);
  uint32_t keff = 0;
  nvdb_hip_timing t;
  check(ctx_, nvdb_hip_search_batch(ctx_, queries, nq, k, ids.data(), sc.data(), &keff, &t));
  last_kernel_ms_ = t.kernel_ms;
  std::vector<SearchResult> out(static_cast<size_t>(nq) * keff);
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t j = 0; j < keff; ++j) out[static_cast<size_t>(q) * keff + j] = SearchResult{ids[static_cast<size_t>(q) * k + j], sc[static_cast<size_t>(q) * k + j]};
  return out;
}

std::vector<SearchResult> FlatIndexHIP::search_topk_dot(const float* q, uint32_t k) const { return search_topk_dot_batch(q, 1, k); }

}  // namespace nvdb
