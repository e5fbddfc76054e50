// coalescer_check.cpp -- TEST ONLY: drives nvdb::detail::CallCoalescer (host/include/nvdb/detail/call_coalescer.h) without a GPU.
// The "GPU batch" is a function that sleeps 200 us whatever the batch size (a batch of 64 costs the GPU what one query costs)
// and answers every query from its first element; T threads x R single-query calls (+ a few multi-query and odd-k calls) must
//   * all get exactly their own rows back,
//   * never run two batches at once,
//   * share batches (far fewer batches than calls when T > 1) and finish in well under T x the solo time,
//   * propagate an error to every caller of the failing batch.
// Built and run by tests/test_host_cpp.py, also under -fsanitize=thread.   usage: coalescer_check <threads> <reps>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "nvdb/detail/call_coalescer.h"

int main(int argc, char** argv) {
  const int T = argc > 1 ? std::atoi(argv[1]) : 6, R = argc > 2 ? std::atoi(argv[2]) : 50;
  const uint32_t dim = 8;
  std::atomic<int> running{0}, overlaps{0}, batches{0}, max_batch{0};
  std::atomic<bool> fail_next{false};
  nvdb::detail::CallCoalescer co(dim, [&](const float* q, uint32_t nq, uint32_t k, uint64_t* ids, float* sc, uint32_t* keff) -> std::string {
    if (running.fetch_add(1) != 0) ++overlaps;
    ++batches;
    int mb = max_batch.load();
    while (static_cast<int>(nq) > mb && !max_batch.compare_exchange_weak(mb, static_cast<int>(nq))) {}
    std::this_thread::sleep_for(std::chrono::microseconds(200));
    for (uint32_t i = 0; i < nq; ++i)
      for (uint32_t j = 0; j < k; ++j) { ids[static_cast<size_t>(i) * k + j] = static_cast<uint64_t>(q[static_cast<size_t>(i) * dim]) * 1000 + j; sc[static_cast<size_t>(i) * k + j] = q[static_cast<size_t>(i) * dim] + 0.5f * j; }
    *keff = k;
    --running;
    return fail_next.exchange(false) ? std::string("injected failure") : std::string();
  });
  auto one_call = [&](int tag, uint32_t nq, uint32_t k) -> int {       // 0 ok, 1 wrong rows, 2 exception
    std::vector<float> q(static_cast<size_t>(nq) * dim, 0.f);
    for (uint32_t i = 0; i < nq; ++i) q[static_cast<size_t>(i) * dim] = static_cast<float>(tag * 16 + static_cast<int>(i));
    std::vector<uint64_t> ids(static_cast<size_t>(nq) * k, 0);
    std::vector<float> sc(static_cast<size_t>(nq) * k, 0.f);
    try {
      if (co.search(q.data(), nq, k, ids.data(), sc.data()) != k) return 1;
    } catch (const std::exception&) { return 2; }
    for (uint32_t i = 0; i < nq; ++i)
      for (uint32_t j = 0; j < k; ++j)
        if (ids[static_cast<size_t>(i) * k + j] != static_cast<uint64_t>(tag * 16 + static_cast<int>(i)) * 1000 + j || sc[static_cast<size_t>(i) * k + j] != q[static_cast<size_t>(i) * dim] + 0.5f * j) return 1;
    return 0;
  };
  // solo
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < R; ++r) if (one_call(r, 1, 10)) { std::printf("FAIL solo\n"); return 1; }
  const double solo_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const int solo_batches = batches.exchange(0);
  // T threads at once: single queries, every 7th call a 3-query request, every 11th a request with another k (must not be mixed in)
  std::atomic<int> wrong{0};
  t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      for (int r = 0; r < R; ++r) {
        const int tag = t * R + r;
        const uint32_t nq = (tag % 7 == 0) ? 3u : 1u, k = (tag % 11 == 0) ? 4u : 10u;
        if (one_call(tag, nq, k)) ++wrong;
      }
    });
  for (auto& x : th) x.join();
  const double par_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const int par_batches = batches.exchange(0);
  // an injected failure reaches its callers as an exception and the next call works again
  fail_next = true;
  const int failed = one_call(1, 1, 10);
  const int after = one_call(2, 1, 10);
  std::printf("threads=%d reps=%d solo_ms=%.2f solo_batches=%d par_ms=%.2f par_batches=%d max_batch=%d overlaps=%d wrong=%d failed=%d after=%d\n",
              T, R, solo_ms, solo_batches, par_ms, par_batches, max_batch.load(), overlaps.load(), wrong.load(), failed, after);
  const bool ok = overlaps == 0 && wrong == 0 && solo_batches == R && failed == 2 && after == 0 && (T == 1 || (par_batches < T * R && max_batch > 1));
  std::printf(ok ? "OK\n" : "FAIL\n");
  return ok ? 0 : 1;
}
