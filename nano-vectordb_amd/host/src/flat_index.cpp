// CPU exact scan: single thread and OpenMP (static row partition, per-thread lists, serial merge --
// the structure of reference src/flat_index.cpp:16-48 and src/flat_index_omp.cpp:16-85).
#include "nvdb/flat_index.h"

#include <stdexcept>

#include "nvdb/flat_index_omp.h"
#include "nvdb/score_dispatch.h"

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace nvdb {

namespace {
void scan_range(const VectorDataset& base, const float* q, uint64_t lo, uint64_t hi, TopKBuffer& out) {
  const uint32_t dim = base.dim(), dt = base.dtype();
  for (uint64_t i = lo; i < hi; ++i) out.consider(i, score_query_base_at(base, q, i, dim, dt));
}
uint32_t prepare(const VectorDataset* base, uint32_t k) {
  if (!base || base->count() == 0) throw std::runtime_error("Empty base");
  ensure_supported_base_dtype(*base);
  return k > base->count() ? static_cast<uint32_t>(base->count()) : k;
}
}  // namespace

std::vector<SearchResult> FlatIndex::search_topk_dot(const float* q, uint32_t k) const {
  k = prepare(base_, k);
  if (k == 0) return {};
  TopKBuffer best(k);
  scan_range(*base_, q, 0, base_->count(), best);
  return best.finalize_sorted_desc();
}

std::vector<SearchResult> FlatIndexOMP::search_topk_dot(const float* q, uint32_t k) const {
  k = prepare(base_, k);
  if (k == 0) return {};
  const uint64_t n = base_->count();
  TopKBuffer best(k);
#if defined(_OPENMP)
  std::vector<TopKBuffer> part(static_cast<size_t>(omp_get_max_threads()), TopKBuffer(k));
#pragma omp parallel
  {
    const uint64_t t = static_cast<uint64_t>(omp_get_thread_num()), T = static_cast<uint64_t>(omp_get_num_threads());
    scan_range(*base_, q, n * t / T, n * (t + 1) / T, part[t]);
  }
  for (auto& p : part) best.merge_from(p.raw());
#else
  scan_range(*base_, q, 0, n, best);
#endif
  return best.finalize_sorted_desc();
}

}  // namespace nvdb
