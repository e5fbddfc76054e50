// ref_shim.cpp -- thin extern "C" window onto the REAL reference library.
//
// TEST INFRASTRUCTURE ONLY (see oracle/nvdb_oracle.c header).  This file is ours; it is compiled
// together with the reference's own sources, taken where they lie under /root/reference, by
// oracle/Makefile, into oracle/_ref/libnvdb_ref.so (git-ignored, never committed).  It lets
// Python (ctypes) call the reference's dot kernels and its FlatIndex / FlatIndexOMP classes so
// that (a) the C restatement in nvdb_oracle.c can be pinned bit-for-bit, (b) golden vectors can
// be generated (oracle/make_golden.py) and (c) bench.py can time the reference's AVX2+OpenMP path
// on the GPU box's host cores (cpu_baseline.kind == "reference").
#include "nvdb/flat_index.h"
#include "nvdb/flat_index_omp.h"
#include "nvdb/simd_dot.h"
#include "nvdb/to_f32_row.h"
#include "nvdb/topK.h"
#include "nvdb/vector_dataset.h"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#if NVDB_HAS_OPENMP
#include <omp.h>
#endif

static std::string g_err;

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

float ref_dot_f32(const float* a, const float* b, uint32_t dim) { return nvdb::dot_f32(a, b, dim); }
float ref_dot_f32_f16base(const float* q, const uint16_t* x, uint32_t dim) { return nvdb::dot_f32_f16base(q, x, dim); }
float ref_dot_f32_i8base(const float* q, const int8_t* x, uint32_t dim, float scale) { return nvdb::dot_f32_i8base(q, x, dim, scale); }
void ref_set_force_scalar(int v) { nvdb::set_force_scalar(v != 0); }

void* ref_dataset_open(const char* path) {
  try {
    auto* ds = new nvdb::VectorDataset();
    ds->load(path);
    return ds;
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void ref_dataset_close(void* h) { delete static_cast<nvdb::VectorDataset*>(h); }
uint64_t ref_dataset_count(void* h) { return static_cast<nvdb::VectorDataset*>(h)->count(); }
uint32_t ref_dataset_dim(void* h) { return static_cast<nvdb::VectorDataset*>(h)->dim(); }
uint32_t ref_dataset_dtype(void* h) { return static_cast<nvdb::VectorDataset*>(h)->dtype(); }

// mode 0: FlatIndex (single thread)   mode 1: FlatIndexOMP (threads>0 -> omp_set_num_threads)
// ids/scores are [nq][k]; returns the number of results per query (min(k,N)), or -1 on throw.
// elapsed_ms (optional) receives the wall time of the nq searches only.
int ref_flat_search(void* h, const float* queries, uint32_t nq, uint32_t k, int mode, int threads,
                    uint64_t* ids, float* scores, double* elapsed_ms) {
  try {
    auto* ds = static_cast<nvdb::VectorDataset*>(h);
#if NVDB_HAS_OPENMP
    if (mode == 1 && threads > 0) omp_set_num_threads(threads);
#endif
    nvdb::FlatIndex st(ds);
    nvdb::FlatIndexOMP omp(ds);
    int got = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t qi = 0; qi < nq; ++qi) {
      const float* q = queries + static_cast<size_t>(qi) * ds->dim();
      std::vector<nvdb::SearchResult> r = (mode == 1) ? omp.search_topk_dot(q, k) : st.search_topk_dot(q, k);
      got = static_cast<int>(r.size());
      for (size_t j = 0; j < r.size(); ++j) {
        if (ids) ids[static_cast<size_t>(qi) * k + j] = r[j].id;
        if (scores) scores[static_cast<size_t>(qi) * k + j] = r[j].score;
      }
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return got;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// include/nvdb/f16_scalar.h:8-38 and include/nvdb/to_f32_row.h:10-34 -- the row conversion under the reference's CPU
// refine (apps/nvdb_ivf_eval.cpp:256, 295).  Header-only in the reference; instantiated here so that goldens exist.
// results go to memory, not through a float return value: a signalling-NaN pattern would be quieted on its way through
// the caller's float -> double conversion, and the golden must hold the function's own bits
void ref_f16_to_f32_array(const uint16_t* h, uint64_t n, float* out) { for (uint64_t i = 0; i < n; ++i) out[i] = nvdb::f16_to_f32_scalar(h[i]); }
int ref_base_row_to_f32(void* h, uint64_t row, float* out) {
  try { nvdb::base_row_to_f32(*static_cast<nvdb::VectorDataset*>(h), row, out); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}

int ref_omp_max_threads() {
#if NVDB_HAS_OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
