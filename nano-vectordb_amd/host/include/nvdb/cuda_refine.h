// Exact-L2 rerank on the GPU with the reference's interface (include/nvdb/cuda_refine.h:7-38):
// same struct, same function name and argument list, so apps/nvdb_ivf_eval.cpp:532,542 compile
// unchanged.  Implemented with HIP kernels for gfx950 through the C ABI (include/nvdb_hip.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace nvdb {

struct CudaRefineTiming {
  float h2d_ms = 0, kernel_ms = 0, d2h_ms = 0, total_ms = 0;
  uint32_t threads = 0;
  uint32_t nwarps = 0;
  uint32_t K = 0;
  uint32_t R = 0;
  size_t shmem_bytes = 0;
  uint32_t dbg_q = 0;
  double dbg_dist_cycles_avg = 0.0, dbg_write_cycles_avg = 0.0, dbg_merge_cycles_avg = 0.0;
  double dbg_dist_pct = 0.0, dbg_write_pct = 0.0, dbg_merge_pct = 0.0;
};

// base_dtype: 1 = fp32, 2 = fp16.  The device copy of the base is cached across calls, keyed by
// (pointer, N, D, dtype), like the reference (src/cuda_refine.cu:188-203).  Errors: std::exit with the
// reference's codes (1 runtime error, 2 bad dtype, 3 K > 64).  Env: CUDA_RETURN_DIST=0 -> ids only; CUDA_PINNED=1 -> the
// call's host buffers are staged through pinned memory (reference src/cuda_refine.cu:875, 902-914).
void cuda_l2_topk_batch(const void* base_ptr, uint32_t base_dtype, uint64_t N, uint32_t D, const float* queries_f32,
                        const uint32_t* cand_ids, uint32_t Q, uint32_t R, uint32_t K, std::vector<uint32_t>& out_topk_ids,
                        std::vector<float>& out_topk_dist, CudaRefineTiming* timing = nullptr);

}  // namespace nvdb
