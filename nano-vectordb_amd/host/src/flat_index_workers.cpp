// FlatIndexAsync / FlatIndexPool: contiguous row blocks per worker, per-worker best-k lists, serial merge in
// worker order.  Behaviour of reference src/flat_index_async.cpp:23-55 and src/flat_index_pool.cpp:98-215
// (errors: "Empty base", "Null query", k == 0 -> empty; unsupported dtype checked once on the caller's thread).
#include <algorithm>
#include <stdexcept>

#include "nvdb/flat_index_async.h"
#include "nvdb/flat_index_pool.h"
#include "nvdb/score_dispatch.h"

#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

namespace nvdb {
namespace {

struct Block { uint64_t lo, hi; };
Block block_of(uint64_t n, int threads, int t) {
  const uint64_t per = (n + static_cast<uint64_t>(threads) - 1) / static_cast<uint64_t>(threads);
  const uint64_t lo = std::min(n, per * static_cast<uint64_t>(t));
  return {lo, std::min(n, lo + per)};
}

void scan_block(const VectorDataset& base, const float* q, Block b, TopKBuffer& out) {
  const uint32_t dim = base.dim(), dt = base.dtype();
  for (uint64_t i = b.lo; i < b.hi; ++i) out.consider(i, score_query_base_at(base, q, i, dim, dt));
}

// the CPUs this process may run on, in ascending order (workers are pinned round-robin over them)
std::vector<int> allowed_cpus() {
  std::vector<int> cpus;
#if defined(__linux__)
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0)
    for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set)) cpus.push_back(c);
#endif
  return cpus;
}

void pin_self(int cpu) {
#if defined(__linux__)
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
#else
  (void)cpu;
#endif
}

}  // namespace

std::vector<SearchResult> FlatIndexAsync::search_topk_dot(const float* q, uint32_t k, int threads) const {
  if (!base_ || base_->count() == 0) throw std::runtime_error("Empty base");
  if (k == 0) return {};
  ensure_supported_base_dtype(*base_);
  threads = std::max(threads, 1);
  const uint64_t n = base_->count();
  std::vector<TopKBuffer> part(static_cast<size_t>(threads), TopKBuffer(k));
  std::vector<std::thread> workers;
  for (int t = 1; t < threads; ++t) {
    const Block b = block_of(n, threads, t);
    if (b.lo < b.hi) workers.emplace_back([this, q, b, &part, t] { scan_block(*base_, q, b, part[static_cast<size_t>(t)]); });
  }
  scan_block(*base_, q, block_of(n, threads, 0), part[0]);       // the caller scans block 0 itself
  for (auto& w : workers) w.join();
  TopKBuffer best(k);
  for (const auto& p : part) best.merge_from(p.raw());
  return best.finalize_sorted_desc();
}

FlatIndexPool::FlatIndexPool(const VectorDataset* base, int threads) : base_(base), threads_(std::max(threads, 1)) {
  if (!base_ || base_->count() == 0) throw std::runtime_error("Empty base");
  part_.assign(static_cast<size_t>(threads_), TopKBuffer(0));
  for (int t = 0; t < threads_; ++t) team_.emplace_back([this, t] { work(t); });
}

FlatIndexPool::~FlatIndexPool() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    quit_ = true;
  }
  go_.notify_all();
  for (auto& w : team_) w.join();
}

void FlatIndexPool::work(int tid) {
  static const std::vector<int> cpus = allowed_cpus();
  if (!cpus.empty()) pin_self(cpus[static_cast<size_t>(tid) % cpus.size()]);
  uint64_t seen = 0;
  for (;;) {
    const float* q;
    uint32_t k;
    {
      std::unique_lock<std::mutex> lk(mu_);
      go_.wait(lk, [&] { return quit_ || round_ != seen; });
      if (quit_) return;
      seen = round_; q = q_; k = k_;
    }
    TopKBuffer mine(k);
    scan_block(*base_, q, block_of(base_->count(), threads_, tid), mine);
    {
      std::lock_guard<std::mutex> lk(mu_);
      part_[static_cast<size_t>(tid)] = std::move(mine);
      if (--pending_ == 0) done_.notify_one();
    }
  }
}

std::vector<SearchResult> FlatIndexPool::search_topk_dot(const float* q, uint32_t k) {
  if (k == 0) return {};
  if (!q) throw std::runtime_error("Null query");
  ensure_supported_base_dtype(*base_);
  std::unique_lock<std::mutex> lk(mu_);
  q_ = q; k_ = k; pending_ = threads_; ++round_;
  go_.notify_all();
  done_.wait(lk, [&] { return pending_ == 0; });
  TopKBuffer best(k);
  for (const auto& p : part_) best.merge_from(p.raw());
  return best.finalize_sorted_desc();
}

}  // namespace nvdb
