// CPU exact-L2 rerank of a candidate list: the semantics of the reference's stage-B CPU path
// (apps/nvdb_ivf_eval.cpp:232-240 l2_sqr_f32, :278-307 refine_topk_l2_ids).  There it is a pair of
// static functions inside an app that needs FAISS; here it is a header so that nvdb_cuda_refine_eval and the tests use
// one definition.  Distances accumulate in DOUBLE over (double(a) - double(b))^2 and are cast to float at the end;
// the k best are kept in a max-heap that is replaced only by a strictly smaller distance; negative ids are skipped;
// the result is ordered best -> worst.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "nvdb/to_f32_row.h"

namespace nvdb {

inline float l2_sqr_f32(const float* a, const float* b, uint32_t d) {
  double s = 0.0;
  for (uint32_t i = 0; i < d; ++i) {
    const double t = static_cast<double>(a[i]) - static_cast<double>(b[i]);
    s += t * t;
  }
  return static_cast<float>(s);
}

struct RefineNode {
  float dist;
  uint64_t id;
};

// cand_ids: int64 like faiss::idx_t (negative = no candidate).  With want_dist the distances come back too.
inline std::vector<uint64_t> refine_topk_l2_ids(const VectorDataset& base, const float* q, const int64_t* cand_ids, int cand_k,
                                                uint32_t k, std::vector<float>* want_dist = nullptr) {
  if (k == 0 || cand_k <= 0) return {};
  if (k > static_cast<uint32_t>(cand_k)) k = static_cast<uint32_t>(cand_k);
  const auto worse_first = [](const RefineNode& x, const RefineNode& y) { return x.dist < y.dist; };   // max-heap on dist
  std::vector<RefineNode> heap;
  heap.reserve(k);
  std::vector<float> row(base.dim());
  for (int i = 0; i < cand_k; ++i) {
    if (cand_ids[i] < 0) continue;
    const uint64_t id = static_cast<uint64_t>(cand_ids[i]);
    base_row_to_f32(base, id, row.data());
    const float dist = l2_sqr_f32(q, row.data(), base.dim());
    if (heap.size() < k) {
      heap.push_back({dist, id});
      std::push_heap(heap.begin(), heap.end(), worse_first);
    } else if (dist < heap.front().dist) {
      std::pop_heap(heap.begin(), heap.end(), worse_first);
      heap.back() = {dist, id};
      std::push_heap(heap.begin(), heap.end(), worse_first);
    }
  }
  std::sort_heap(heap.begin(), heap.end(), worse_first);   // ascending distance = best -> worst
  std::vector<uint64_t> ids(heap.size());
  if (want_dist) want_dist->resize(heap.size());
  for (size_t j = 0; j < heap.size(); ++j) {
    ids[j] = heap[j].id;
    if (want_dist) (*want_dist)[j] = heap[j].dist;
  }
  return ids;
}

}  // namespace nvdb
