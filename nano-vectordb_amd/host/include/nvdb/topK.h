// SearchResult and a small bounded best-k buffer.  Surface of reference include/nvdb/topK.h:9-69;
// ordering here is the build's canonical (score desc, id asc) -- see DESIGN.md "ties".
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace nvdb {

struct SearchResult {
  uint64_t id;
  float score;
};

inline bool result_before(const SearchResult& a, const SearchResult& b) {
  return a.score > b.score || (a.score == b.score && a.id < b.id);
}

class TopKBuffer {
 public:
  explicit TopKBuffer(uint32_t k) : k_(k) { items_.reserve(k); }

  void consider(uint64_t id, float score) {
    if (k_ == 0) return;
    const SearchResult cand{id, score};
    if (items_.size() == k_) {
      if (!result_before(cand, items_.back())) return;
      items_.pop_back();
    }
    items_.insert(std::upper_bound(items_.begin(), items_.end(), cand, result_before), cand);
  }
  void merge_from(const std::vector<SearchResult>& other) { for (const auto& r : other) consider(r.id, r.score); }
  std::vector<SearchResult> finalize_sorted_desc() { return items_; }
  const std::vector<SearchResult>& raw() const { return items_; }

 private:
  uint32_t k_;
  std::vector<SearchResult> items_;   // always sorted best-first
};

}  // namespace nvdb
