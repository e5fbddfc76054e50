// nvdb_gt_build -- exact top-k ids of every query -> .gtbin (reference apps/nvdb_gt_build.cpp:22-129).
// GT_MODE = omp (default, as in the reference) | gpu | st | async | pool ; WARMUP as in the reference.
// Loader errors escape like the reference's (uncaught std::runtime_error); a failure of the GPU path (no device, out of
// memory ...) is reported and ends the program with exit code 7.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "nvdb/flat_index.h"
#include "nvdb/flat_index_hip.h"
#include "nvdb/flat_index_async.h"
#include "nvdb/flat_index_omp.h"
#include "nvdb/flat_index_pool.h"
#include "nvdb/score_dispatch.h"
#include "nvdb/gtbin_format.h"
#include "nvdb_hip.h"

int main(int argc, char** argv) {
  if (argc < 5) {
    std::cerr << "Usage: nvdb_gt_build <base.vecbin> <query.vecbin> <k> <out.gtbin>\nEnv:\n  GT_MODE=gpu|omp|st|async|pool\n  WARMUP=2\n";
    return 1;
  }
  const uint32_t k = static_cast<uint32_t>(std::stoul(argv[3]));
  const char* m = std::getenv("GT_MODE");
  const std::string mode = m ? m : "omp";                     // reference default (apps/nvdb_gt_build.cpp:33-35); GT_MODE=gpu selects the MI355X path
  nvdb::VectorDataset base, query;
  base.load(argv[1]);
  query.load(argv[2]);
  if (query.dtype() != 1) { std::cerr << "Query must be float32 vecbin (got dtype=" << query.dtype() << ")\n"; return 2; }
  if (base.dim() != query.dim()) { std::cerr << "Dim mismatch: base.dim=" << base.dim() << " query.dim=" << query.dim() << "\n"; return 3; }
  if (k == 0) { std::cerr << "k must be > 0\n"; return 4; }
  const uint64_t N = base.count(), Q = query.count();
  std::cout << "GT build: N=" << N << " Q=" << Q << " d=" << base.dim() << " k=" << k << " mode=" << mode << "\n";
  std::vector<uint32_t> ids(static_cast<size_t>(Q) * k, 0);
  auto store = [&](uint64_t qi, const nvdb::SearchResult* r, size_t got) -> bool {
    if (got != k) { std::cerr << "GT size mismatch at qi=" << qi << " got=" << got << " expected=" << k << "\n"; return false; }
    for (uint32_t j = 0; j < k; ++j) ids[qi * k + j] = static_cast<uint32_t>(r[j].id);
    return true;
  };
  if (mode == "gpu") {
   try {
    nvdb::FlatIndexHIP idx(&base);
    const uint64_t B = 1024;
    for (uint64_t q0 = 0; q0 < Q; q0 += B) {
      const uint32_t b = static_cast<uint32_t>(std::min<uint64_t>(B, Q - q0));
      const auto res = idx.search_topk_dot_batch(query.vector_ptr_f32(q0), b, k);
      const size_t ke = res.size() / b;
      for (uint32_t i = 0; i < b; ++i) if (!store(q0 + i, res.data() + i * ke, ke)) return 5;
    }
  } catch (const std::exception& e) {
    std::cerr << "nvdb_gt_build: GT_MODE=gpu failed: " << e.what() << "\n";
    return 7;
   }
  } else {
    nvdb::FlatIndex st(&base);
    nvdb::FlatIndexOMP omp(&base);
    nvdb::FlatIndexAsync async(&base);
    const int nthreads = std::max(1u, std::thread::hardware_concurrency());
    std::unique_ptr<nvdb::FlatIndexPool> pool;
    if (mode == "pool") pool = std::make_unique<nvdb::FlatIndexPool>(&base, nthreads);
    for (uint64_t qi = 0; qi < Q; ++qi) {
      const float* qv = query.vector_ptr_f32(qi);
      const auto r = mode == "st" ? st.search_topk_dot(qv, k) : mode == "async" ? async.search_topk_dot(qv, k, nthreads)
                   : mode == "pool" ? pool->search_topk_dot(qv, k) : omp.search_topk_dot(qv, k);
      if (!store(qi, r.data(), r.size())) return 5;
      if ((qi + 1) % 200 == 0) std::cout << "GT " << (qi + 1) << "/" << Q << "\n";
    }
  }
  nvdb::GtBinHeader h;
  std::memset(&h, 0, sizeof(h));
  h.magic = nvdb::kGtMagic; h.version = nvdb::kGtVersion; h.metric = static_cast<uint32_t>(nvdb::GtMetric::DotEquivalentL2);
  h.k = k; h.dim = base.dim(); h.Q = Q; h.N = N;
  std::ofstream out(argv[4], std::ios::binary);
  if (!out) { std::cerr << "Failed to open output: " << argv[4] << "\n"; return 6; }
  out.write(reinterpret_cast<const char*>(&h), sizeof(h));
  out.write(reinterpret_cast<const char*>(ids.data()), static_cast<std::streamsize>(ids.size() * sizeof(uint32_t)));
  std::cout << "Wrote GT: " << argv[4] << " (header=64B, payload=" << ids.size() * sizeof(uint32_t) << " bytes)\n";
  return 0;
}

