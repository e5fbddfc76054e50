#include "nvdb/flat_index_hip.h"

#include <algorithm>

#include <stdexcept>
#include <string>
#include <thread>

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

// ---- row-sharded over several GPUs -------------------------------------------------------------------------
FlatIndexHIPSharded::FlatIndexHIPSharded(const VectorDataset* base, const std::vector<int>& devices) {
  if (!base || base->count() == 0) throw std::runtime_error("Empty base");
  if (devices.empty()) throw std::runtime_error("FlatIndexHIPSharded: no devices");
  if (bytes_per_elem(base->dtype()) == 0) throw std::runtime_error("Unsupported base dtype (Float32/Float16/Int8 only)");
  n_ = base->count();
  dim_ = base->dim();
  const size_t G = std::min<size_t>(devices.size(), n_);
  const size_t row_bytes = static_cast<size_t>(dim_) * bytes_per_elem(base->dtype());
  try {
    for (size_t g = 0; g < G; ++g) {
      const uint64_t lo = n_ * g / G, hi = n_ * (g + 1) / G;
      nvdb_hip_ctx* c = nullptr;
      if (nvdb_hip_create(devices[g], &c) != NVDB_OK) throw std::runtime_error(std::string(nvdb_hip_last_error(nullptr)));
      ctx_.push_back(c);
      const char* rows = static_cast<const char*>(base->payload_ptr()) + lo * row_bytes;
      const float* scales = base->scales_ptr() ? base->scales_ptr() + lo : nullptr;
      check(c, nvdb_hip_upload_corpus(c, rows, scales, hi - lo, dim_, base->dtype(), lo));
    }
  } catch (...) {
    for (auto* c : ctx_) nvdb_hip_destroy(c);
    ctx_.clear();
    throw;
  }
}

FlatIndexHIPSharded::~FlatIndexHIPSharded() { for (auto* c : ctx_) nvdb_hip_destroy(c); }

std::vector<SearchResult> FlatIndexHIPSharded::search_topk_dot_batch(const float* queries, uint32_t nq, uint32_t k) const {
  if (!queries) throw std::runtime_error("Null query");
  if (k == 0 || nq == 0) return {};
  const size_t G = ctx_.size();
  const size_t per = static_cast<size_t>(nq) * k;
  std::vector<uint64_t> ids(G * per, ~0ull);
  std::vector<float> sc(G * per, -__builtin_huge_valf());
  std::vector<std::string> errs(G);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; ++g)
    th.emplace_back([&, g] {
      uint32_t keff = 0;
      if (nvdb_hip_search_batch(ctx_[g], queries, nq, k, ids.data() + g * per, sc.data() + g * per, &keff, nullptr) != NVDB_OK)
        errs[g] = nvdb_hip_last_error(ctx_[g]);
    });
  for (auto& t : th) t.join();
  for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
  std::vector<uint64_t> mi(per);
  std::vector<float> ms(per);
  if (nvdb_merge_topk_host(ids.data(), sc.data(), static_cast<uint32_t>(G), nq, k, mi.data(), ms.data()) != NVDB_OK)
    throw std::runtime_error("merge_topk_host failed");
  const uint32_t keff = static_cast<uint32_t>(std::min<uint64_t>(k, n_));
  std::vector<SearchResult> out(static_cast<size_t>(nq) * keff);
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t j = 0; j < keff; ++j) out[static_cast<size_t>(q) * keff + j] = SearchResult{mi[static_cast<size_t>(q) * k + j], ms[static_cast<size_t>(q) * k + j]};
  return out;
}

}  // namespace nvdb
