// host_capi.cpp -- extern "C" test hooks onto the C++ host layer (lib/libnvdb_host_capi.so).
//
// NOT the drop-in boundary (that is include/nvdb_hip.h) and not linked into the tools: it exists so that the CPU
// tests can call the nvdb:: host functions directly through ctypes and compare them with the reference goldens --
// the dot kernels on both dispatch branches, f16_to_f32_scalar / base_row_to_f32 (reference f16_scalar.h,
// to_f32_row.h) and the CPU refine (apps/nvdb_ivf_eval.cpp:232-240, 278-307).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "nvdb/cpu_refine.h"
#include "nvdb/f16_scalar.h"
#include "nvdb/flat_index_hip.h"
#include "nvdb/simd_dot.h"
#include "nvdb/to_f32_row.h"
#include "nvdb/vector_dataset.h"

static std::string g_err;

extern "C" {

const char* nvdb_host_last_error() { return g_err.c_str(); }
int nvdb_host_simd_available() { return nvdb::simd_dot_available() ? 1 : 0; }
void nvdb_host_set_force_scalar(int v) { nvdb::set_force_scalar(v != 0); }
float nvdb_host_dot_f32(const float* a, const float* b, uint32_t dim) { return nvdb::dot_f32(a, b, dim); }
float nvdb_host_dot_f32_f16base(const float* q, const uint16_t* x, uint32_t dim) { return nvdb::dot_f32_f16base(q, x, dim); }
float nvdb_host_dot_f32_f16base_scalar(const float* q, const uint16_t* x, uint32_t dim) { return nvdb::dot_f32_f16base_scalar(q, x, dim); }
float nvdb_host_dot_f32_i8base(const float* q, const int8_t* x, uint32_t dim, float scale) { return nvdb::dot_f32_i8base(q, x, dim, scale); }

void nvdb_host_f16_to_f32(const uint16_t* h, uint64_t n, float* out) {
  for (uint64_t i = 0; i < n; ++i) out[i] = nvdb::f16_to_f32_scalar(h[i]);
}

void* nvdb_host_dataset_open(const char* path) {
  try {
    auto* ds = new nvdb::VectorDataset();
    ds->load(path);
    return ds;
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void nvdb_host_dataset_close(void* h) { delete static_cast<nvdb::VectorDataset*>(h); }

int nvdb_host_base_row_to_f32(void* h, uint64_t row, float* out) {
  try { nvdb::base_row_to_f32(*static_cast<nvdb::VectorDataset*>(h), row, out); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// out_ids[k] padded with UINT64_MAX, out_dist[k] (optional) padded with +inf; returns the number of results
int nvdb_host_refine_topk_l2(void* h, const float* q, const int64_t* cand, int cand_k, uint32_t k, uint64_t* out_ids, float* out_dist) {
  try {
    std::vector<float> dist;
    const std::vector<uint64_t> ids = nvdb::refine_topk_l2_ids(*static_cast<nvdb::VectorDataset*>(h), q, cand, cand_k, k, &dist);
    for (uint32_t j = 0; j < k; ++j) {
      out_ids[j] = j < ids.size() ? ids[j] : ~0ull;
      if (out_dist) out_dist[j] = j < dist.size() ? dist[j] : __builtin_huge_valf();
    }
    return static_cast<int>(ids.size());
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// FlatIndexHIP / FlatIndexHIPSharded::search_topk_dot from `threads` host threads at once (const entry points, like the
// reference's FlatIndex): thread t takes the queries t, t + threads, ...; out_ids / out_scores [nq][k] (padded with
// UINT64_MAX / -inf).  devices == nullptr: one GPU (device 0); else a row-sharded index over n_devices devices.
int nvdb_host_hip_concurrent_search(void* h, const float* queries, uint32_t nq, uint32_t k, int threads, const int* devices,
                                    uint32_t n_devices, uint64_t* out_ids, float* out_scores) {
  try {
    const auto* ds = static_cast<nvdb::VectorDataset*>(h);
    std::unique_ptr<nvdb::FlatIndexHIP> one;
    std::unique_ptr<nvdb::FlatIndexHIPSharded> many;
    if (devices) many.reset(new nvdb::FlatIndexHIPSharded(ds, std::vector<int>(devices, devices + n_devices)));
    else one.reset(new nvdb::FlatIndexHIP(ds));
    const uint32_t dim = ds->dim();
    std::vector<std::string> errs(static_cast<size_t>(threads));
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
      th.emplace_back([&, t] {
        try {
          for (uint32_t q = static_cast<uint32_t>(t); q < nq; q += static_cast<uint32_t>(threads)) {
            const float* qp = queries + static_cast<size_t>(q) * dim;
            const std::vector<nvdb::SearchResult> r = one ? one->search_topk_dot(qp, k) : many->search_topk_dot(qp, k);
            for (uint32_t j = 0; j < k; ++j) {
              out_ids[static_cast<size_t>(q) * k + j] = j < r.size() ? r[j].id : ~0ull;
              out_scores[static_cast<size_t>(q) * k + j] = j < r.size() ? r[j].score : -__builtin_huge_valf();
            }
          }
        } catch (const std::exception& e) { errs[static_cast<size_t>(t)] = e.what(); }
      });
    for (auto& x : th) x.join();
    for (const auto& e : errs) if (!e.empty()) { g_err = e; return -1; }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Throughput of overlapping single-query callers (the coalescing of FlatIndexHIP::search_topk_dot): out_ms[0] = one thread running
// `reps` single-query searches, out_ms[1] = `threads` threads running `reps` each at the same time (thread t cycles through the
// queries t, t + threads, ...).  Every result is compared with out-of-band batched answers by the caller through out_ids / out_scores
// of the LAST search of each query index.
int nvdb_host_hip_concurrent_timing(void* h, const float* queries, uint32_t nq, uint32_t k, int threads, uint32_t reps, uint64_t* out_ids,
                                    float* out_scores, double* out_ms) {
  try {
    const auto* ds = static_cast<nvdb::VectorDataset*>(h);
    nvdb::FlatIndexHIP idx(ds);
    const uint32_t dim = ds->dim();
    auto one = [&](uint32_t q) {
      const std::vector<nvdb::SearchResult> r = idx.search_topk_dot(queries + static_cast<size_t>(q) * dim, k);
      for (uint32_t j = 0; j < k; ++j) {
        out_ids[static_cast<size_t>(q) * k + j] = j < r.size() ? r[j].id : ~0ull;
        out_scores[static_cast<size_t>(q) * k + j] = j < r.size() ? r[j].score : -__builtin_huge_valf();
      }
    };
    for (uint32_t i = 0; i < 5; ++i) one(i % nq);                                        // warm-up: buffers, kernels
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < reps; ++i) one(i % nq);
    out_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<std::string> errs(static_cast<size_t>(threads));
    std::vector<std::thread> th;
    t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t)
      th.emplace_back([&, t] {
        try { for (uint32_t i = 0; i < reps; ++i) one((static_cast<uint32_t>(t) + i * static_cast<uint32_t>(threads)) % nq); }
        catch (const std::exception& e) { errs[static_cast<size_t>(t)] = e.what(); }
      });
    for (auto& x : th) x.join();
    out_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (const auto& e : errs) if (!e.empty()) { g_err = e; return -1; }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
