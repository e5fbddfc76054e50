// nvdb_cuda_refine_eval -- FAISS-free harness for the exact-refine stage.
//
// The reference ships this file EMPTY (apps/nvdb_cuda_refine_eval.cpp, 0 bytes); its only caller of
// cuda_l2_topk_batch is apps/nvdb_ivf_eval.cpp:496-601 behind FAISS.  This program is that stage B on its
// own: it builds (or reads) candidate lists, runs the CPU refine (double accumulation, the arithmetic of
// apps/nvdb_ivf_eval.cpp:232-240, 278-307) and the GPU refine (nvdb::cuda_l2_topk_batch with the reference's
// signature), checks them against each other and prints the reference's refine lines (:572-576, :743-779).
//
//   nvdb_cuda_refine_eval <base.vecbin> <query.vecbin> <k>
//   env: REFINE_K (default 1024)   CAND_PATH (raw uint32[Q*REFINE_K]; default: synthetic, see below)
//        CUDA_REFINE_WARMUP (1)    CUDA_RETURN_DIST (1)   CUDA_PINNED (0)   GIT_SHA
// Synthetic candidates (FAISS absent): per query the exact top-min(64,REFINE_K) by dot (== L2 for
// normalised data) found with the GPU flat scan, filled up to REFINE_K with distinct pseudo-random ids,
// shuffled with a seeded generator; 1 % of the slots are set to 0xFFFFFFFF to exercise the skip path.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <queue>
#include <random>
#include <string>
#include <unordered_set>
#include <vector>

#include "nvdb/cpu_refine.h"
#include "nvdb/cuda_refine.h"
#include "nvdb/flat_index_hip.h"
#include "nvdb/vector_dataset.h"

static int env_int(const char* k, int d) { const char* v = std::getenv(k); return v ? std::atoi(v) : d; }
static std::string env_str(const char* k, const char* d) { const char* v = std::getenv(k); return v ? v : d; }

int main(int argc, char** argv) {
  if (argc < 4) { std::cerr << "Usage: nvdb_cuda_refine_eval <base.vecbin> <query.vecbin> <k>\n"; return 1; }
  const int k = std::atoi(argv[3]);
  const int refine_k = env_int("REFINE_K", 1024);
  nvdb::VectorDataset base, query;
  base.load(argv[1]);
  query.load(argv[2]);
  if (base.dim() != query.dim()) { std::cerr << "Dim mismatch\n"; return 2; }
  if (query.dtype() != 1) { std::cerr << "Query must be float32\n"; return 3; }
  if (base.dtype() != 1 && base.dtype() != 2) { std::cerr << "CUDA refine supports base dtype fp16/fp32 only (got dtype=" << base.dtype() << ")\n"; return 12; }
  if (k <= 0 || refine_k < k) { std::cerr << "need 0 < k <= REFINE_K\n"; return 4; }
  const uint64_t N = base.count(), Q = query.count();
  const uint32_t d = base.dim(), R = static_cast<uint32_t>(refine_k);

  // ---- stage A stand-in: candidate lists -------------------------------------------------------------
  std::vector<uint32_t> cand(static_cast<size_t>(Q) * R, 0xFFFFFFFFu);
  const std::string cand_path = env_str("CAND_PATH", "");
  if (!cand_path.empty()) {
    std::ifstream in(cand_path, std::ios::binary);
    if (!in.read(reinterpret_cast<char*>(cand.data()), static_cast<std::streamsize>(cand.size() * 4))) { std::cerr << "CAND_PATH too short\n"; return 5; }
  } else {
    const uint32_t seedk = std::min<uint32_t>(64, R);
    nvdb::FlatIndexHIP flat(&base);
    std::vector<nvdb::SearchResult> nn;
    for (uint64_t q0 = 0; q0 < Q; q0 += 1024) {
      const uint32_t b = static_cast<uint32_t>(std::min<uint64_t>(1024, Q - q0));
      const auto part = flat.search_topk_dot_batch(query.vector_ptr_f32(q0), b, seedk);
      nn.insert(nn.end(), part.begin(), part.end());
    }
    const size_t ke = nn.size() / Q;
    std::mt19937 rng(20240613u);
    for (uint64_t qi = 0; qi < Q; ++qi) {
      std::unordered_set<uint32_t> seen;
      uint32_t* c = cand.data() + qi * R;
      uint32_t fill = 0;
      for (size_t j = 0; j < ke && fill < R; ++j) { const uint32_t id = static_cast<uint32_t>(nn[qi * ke + j].id); if (seen.insert(id).second) c[fill++] = id; }
      while (fill < R && seen.size() < N) { const uint32_t id = static_cast<uint32_t>(rng() % N); if (seen.insert(id).second) c[fill++] = id; }
      std::shuffle(c, c + fill, rng);
      for (uint32_t r = 0; r < fill; ++r) if (rng() % 100 == 0) c[r] = 0xFFFFFFFFu;
    }
  }

  // ---- CPU refine: nvdb::refine_topk_l2_ids (double accumulation over base_row_to_f32 rows; reference
  //      apps/nvdb_ivf_eval.cpp:232-240, 278-307, include/nvdb/to_f32_row.h:10-34) -----------------------------
  std::vector<uint32_t> cpu_ids(static_cast<size_t>(Q) * k, 0xFFFFFFFFu);
  const auto tc0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t qi = 0; qi < static_cast<int64_t>(Q); ++qi) {
    std::vector<int64_t> c64(R);
    for (uint32_t r = 0; r < R; ++r) {
      const uint32_t id = cand[qi * R + r];
      c64[r] = (id == 0xFFFFFFFFu || id >= N) ? -1 : static_cast<int64_t>(id);      // the stage-A "no candidate" marker (:513-516)
    }
    const std::vector<uint64_t> best = nvdb::refine_topk_l2_ids(base, query.vector_ptr_f32(qi), c64.data(), static_cast<int>(R), static_cast<uint32_t>(k));
    for (size_t j = 0; j < best.size(); ++j) cpu_ids[qi * k + j] = static_cast<uint32_t>(best[j]);
  }
  const double cpu_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();

  // ---- GPU refine (the drop-in call, apps/nvdb_ivf_eval.cpp:542-549) ------------------------------------
  std::vector<float> queries_all(static_cast<size_t>(Q) * d);
  for (uint64_t qi = 0; qi < Q; ++qi) std::memcpy(queries_all.data() + qi * d, query.vector_ptr_f32(qi), d * sizeof(float));
  std::vector<uint32_t> topk_ids;
  std::vector<float> topk_dist;
  nvdb::CudaRefineTiming t{};
  if (env_int("CUDA_REFINE_WARMUP", 1))
    nvdb::cuda_l2_topk_batch(base.data_ptr(), base.dtype(), N, d, queries_all.data(), cand.data(), static_cast<uint32_t>(Q), R, k, topk_ids, topk_dist, &t);
  nvdb::cuda_l2_topk_batch(base.data_ptr(), base.dtype(), N, d, queries_all.data(), cand.data(), static_cast<uint32_t>(Q), R, k, topk_ids, topk_dist, &t);

  std::cout << "CUDA_REFINE=1 refine_ms_total=" << t.total_ms << " (h2d=" << t.h2d_ms << " kernel=" << t.kernel_ms << " d2h=" << t.d2h_ms << " ms)"
            << " avg=" << (t.total_ms / double(Q)) << " ms/query\n";

  // recall of the GPU ids against the CPU refine ids (the reference's own check is recall equality, Performance_CUDA.md:117)
  double recall_sum = 0.0;
  uint64_t exact_rows = 0;
  for (uint64_t qi = 0; qi < Q; ++qi) {
    int hit = 0, valid = 0;
    for (int j = 0; j < k; ++j) {
      const uint32_t g = cpu_ids[qi * k + j];
      if (g == 0xFFFFFFFFu) continue;
      ++valid;
      for (int i = 0; i < k; ++i) if (topk_ids[qi * k + i] == g) { ++hit; break; }
    }
    recall_sum += valid ? double(hit) / valid : 1.0;
    exact_rows += std::equal(cpu_ids.begin() + qi * k, cpu_ids.begin() + (qi + 1) * k, topk_ids.begin() + qi * k) ? 1 : 0;
  }
  std::cout << std::fixed << std::setprecision(6);
  std::cout << "refine_recall_gpu_vs_cpu=" << recall_sum / double(Q) << " identical_rows=" << exact_rows << "/" << Q << " cpu_refine_ms_total=" << cpu_ms << "\n";
  std::cout << "RESULT refine_k=" << refine_k << " Q=" << Q << " k=" << k << " cuda_refine=1 refine_enabled=1 refine_backend=cuda"
            << " refine_ms_total=" << t.total_ms << " refine_ms_per_q=" << t.total_ms / double(Q)
            << " kernel_mode=wave64 cuda_pinned=" << env_int("CUDA_PINNED", 0) << " cuda_return_dist=" << env_int("CUDA_RETURN_DIST", 1) << " git_rev=" << env_str("GIT_SHA", "NA")
            << " refine_h2d_ms=" << t.h2d_ms << " refine_kernel_ms=" << t.kernel_ms << " refine_d2h_ms=" << t.d2h_ms
            << " refine_kernel_ms_per_q=" << (Q ? t.kernel_ms / double(Q) : 0.0)
            << " cuda_threads=" << t.threads << " cuda_nwarps=" << t.nwarps << " cuda_shmem_bytes=" << t.shmem_bytes
            << " cuda_forced_threads=0 cuda_shmem_optin=0"
            << " gather_GBps=" << (t.kernel_ms > 0 ? double(Q) * R * d * nvdb::bytes_per_elem(base.dtype()) * 1e-6 / t.kernel_ms : 0.0)
            << " cpu_refine_ms_total=" << cpu_ms << " recall_vs_cpu=" << recall_sum / double(Q) << "\n";
  return 0;
}
