// nvdb::cuda_l2_topk_batch on MI355X: same signature and conventions as reference
// src/cuda_refine.cu:839-1173, implemented over the C ABI (nvdb_hip_refine_l2_topk).
#include <cstdio>
#include <cstdlib>

#include "nvdb/cuda_refine.h"
#include "nvdb_hip.h"

namespace nvdb {

namespace {
struct Cache {          // single-owner state, like the reference's file-static caches (:26-90)
  nvdb_hip_ctx* ctx = nullptr;
  const void* base = nullptr;
  uint64_t n = 0;
  uint32_t d = 0, dt = 0;
  ~Cache() { nvdb_hip_destroy(ctx); }
};
Cache g;

[[noreturn]] void die(const char* what, int code) {
  std::fprintf(stderr, "[HIP ERROR] %s: %s\n", what, nvdb_hip_last_error(g.ctx));
  std::exit(code);
}
}  // namespace

void cuda_l2_topk_batch(const void* base_ptr, uint32_t base_dtype, uint64_t N, uint32_t D, const float* queries_f32,
                        const uint32_t* cand_ids, uint32_t Q, uint32_t R, uint32_t K, std::vector<uint32_t>& out_ids,
                        std::vector<float>& out_dist, CudaRefineTiming* timing) {
  if (K == 0 || Q == 0 || R == 0) {   // :853-857
    out_ids.clear(); out_dist.clear();
    if (timing) *timing = {};
    return;
  }
  if (K > NVDB_HIP_REFINE_KMAX) { std::fprintf(stderr, "cuda_l2_topk_batch: K=%u not supported (K<=%d)\n", K, NVDB_HIP_REFINE_KMAX); std::exit(3); }
  if (base_dtype != 1 && base_dtype != 2) { std::fprintf(stderr, "ensure_base_on_gpu: unsupported base_dtype=%u\n", base_dtype); std::exit(2); }
  if (!g.ctx && nvdb_hip_create(0, &g.ctx) != NVDB_OK) die("create", 1);
  if (g.base != base_ptr || g.n != N || g.d != D || g.dt != base_dtype) {      // one-time upload, cached (:188-203)
    if (nvdb_hip_upload_corpus(g.ctx, base_ptr, nullptr, N, D, base_dtype, 0) != NVDB_OK) die("H2D base(cache)", 1);
    g.base = base_ptr; g.n = N; g.d = D; g.dt = base_dtype;
  }
  const char* pn = std::getenv("CUDA_PINNED");                                  // :875 pinned host staging of the call's buffers
  if (nvdb_hip_set_option(g.ctx, "refine_pinned", (pn && std::atoi(pn) != 0) ? 1 : 0) != NVDB_OK) die("set_option", 1);
  const char* rd = std::getenv("CUDA_RETURN_DIST");
  const bool want_dist = !(rd && std::atoi(rd) == 0);
  out_ids.assign(static_cast<size_t>(Q) * K, 0xFFFFFFFFu);
  if (want_dist) out_dist.assign(static_cast<size_t>(Q) * K, 1e30f); else out_dist.clear();
  nvdb_hip_timing t;
  if (nvdb_hip_refine_l2_topk(g.ctx, queries_f32, cand_ids, Q, R, K, out_ids.data(), want_dist ? out_dist.data() : nullptr, &t) != NVDB_OK)
    die("refine", 1);
  if (timing) {
    *timing = {};
    timing->h2d_ms = t.h2d_ms; timing->kernel_ms = t.kernel_ms; timing->d2h_ms = t.d2h_ms; timing->total_ms = t.total_ms;
    timing->threads = t.threads; timing->nwarps = t.nwarps; timing->K = t.K; timing->R = t.R; timing->shmem_bytes = t.shmem_bytes;
  }
}

}  // namespace nvdb
