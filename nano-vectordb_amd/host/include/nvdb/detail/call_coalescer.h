// call_coalescer.h -- group-commit for searches on one GPU context (used by nvdb::FlatIndexHIP / FlatIndexHIPSharded; header-only so
// that tests/ can drive it without a GPU: tests/test_host_cpp.py builds tests/coalescer_check.cpp under ThreadSanitizer).
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace nvdb {
namespace detail {

// Group-commit for searches.  The reference's FlatIndex::search_topk_dot is const and re-entrant (flat_index.h:11-16): its
// callers overlap freely and each uses its own core.  One GPU context runs one batch at a time -- but a batch of 64 queries
// costs it what one query costs -- so overlapping callers are served TOGETHER instead of one after the other:
//   * a caller enqueues its request; if no batch is running it becomes the leader, otherwise it sleeps;
//   * the leader takes every queued request with its own k (<= 1024 queries in all), copies the queries into one block,
//     runs ONE batch, scatters the rows, wakes everybody; the oldest request left over leads the next batch;
//   * linger: only when the previous batch was shared does a leader that is still alone wait -- at most 50 us, or until as
//     many requests as last time have arrived -- because those callers are on their way back.  A lone caller never waits.
class CallCoalescer {
 public:
  // runs one batch: (queries, nq, k, ids[nq*k], scores[nq*k], &keff) -> "" or the error text
  using Run = std::function<std::string(const float*, uint32_t, uint32_t, uint64_t*, float*, uint32_t*)>;
  CallCoalescer(uint32_t dim, Run run) : dim_(dim), run_(std::move(run)) {}

  // blocks until the caller's rows are in ids / scores ([nq][k]); throws what the batch threw
  uint32_t search(const float* queries, uint32_t nq, uint32_t k, uint64_t* ids, float* scores) {
    Request me{queries, nq, k, ids, scores};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&me);
    arrived_.notify_all();                                     // a lingering leader counts arrivals
    wake_.wait(lk, [&] { return me.done || (!busy_ && queue_.front() == &me); });
    if (!me.done) lead(lk, me);
    if (!me.err.empty()) throw std::runtime_error(me.err);
    return me.keff;
  }

 private:
  struct Request {
    const float* q; uint32_t nq, k; uint64_t* ids; float* sc;
    uint32_t keff = 0; bool done = false; std::string err;
  };
  static constexpr uint32_t MAX_BATCH = 1024;                   // one sub-batch of nvdb_hip_search_batch (its one-copy result path)

  void lead(std::unique_lock<std::mutex>& lk, Request& me) {
    busy_ = true;
    if (last_shared_ > 1 && queue_.size() < last_shared_)
      // (wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer understands; wait_for's pthread_cond_clockwait
      //  is invisible to GCC 11's libtsan and makes every access under this mutex look like a race)
      arrived_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(50), [&] { return queue_.size() >= last_shared_; });
    // my batch: me + every queued request with my k while the total stays within one sub-batch
    std::vector<Request*> mine{&me};
    uint32_t total = me.nq;
    for (auto it = queue_.begin(); it != queue_.end();) {
      Request* r = *it;
      if (r == &me) { it = queue_.erase(it); continue; }
      if (r->k == me.k && total + r->nq <= MAX_BATCH && me.nq <= MAX_BATCH) { mine.push_back(r); total += r->nq; it = queue_.erase(it); }
      else ++it;
    }
    last_shared_ = static_cast<uint32_t>(mine.size());
    lk.unlock();
    std::string err;
    uint32_t keff = 0;
    if (mine.size() == 1) err = run_(me.q, me.nq, me.k, me.ids, me.sc, &keff);          // alone: straight through, no copies
    else {
      qbuf_.resize(static_cast<size_t>(total) * dim_);
      ibuf_.resize(static_cast<size_t>(total) * me.k);
      sbuf_.resize(static_cast<size_t>(total) * me.k);
      size_t off = 0;
      for (Request* r : mine) { std::memcpy(qbuf_.data() + off * dim_, r->q, static_cast<size_t>(r->nq) * dim_ * sizeof(float)); off += r->nq; }
      err = run_(qbuf_.data(), total, me.k, ibuf_.data(), sbuf_.data(), &keff);
      off = 0;
      if (err.empty())
        for (Request* r : mine) {
          std::memcpy(r->ids, ibuf_.data() + off * me.k, static_cast<size_t>(r->nq) * me.k * sizeof(uint64_t));
          std::memcpy(r->sc, sbuf_.data() + off * me.k, static_cast<size_t>(r->nq) * me.k * sizeof(float));
          off += r->nq;
        }
    }
    lk.lock();
    for (Request* r : mine) { r->keff = keff; r->err = err; r->done = true; }
    busy_ = false;
    wake_.notify_all();
  }

  const uint32_t dim_;
  Run run_;
  std::mutex mu_;
  std::condition_variable wake_, arrived_;
  std::deque<Request*> queue_;
  bool busy_ = false;
  uint32_t last_shared_ = 1;                                    // requests in the previous batch
  std::vector<float> qbuf_, sbuf_;                              // the leader's staging (one leader at a time)
  std::vector<uint64_t> ibuf_;
};

}  // namespace detail

}  // namespace nvdb
