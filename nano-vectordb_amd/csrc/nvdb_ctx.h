// nvdb_ctx.h -- internal to libnvdb_hip.so: the context behind the C ABI (include/nvdb_hip.h), the small host helpers every
// translation unit uses, and the declarations of what one unit calls in another.  Not installed, not part of the boundary.
//
//   nvdb_corpus.cpp        create / destroy, corpus upload / adopt / generate (+ shadow copies, row-norm pass), options, statistics
//   nvdb_search.cpp        orchestration of one flat search (search_core), the search entry points and their self-checks
//   nvdb_launch_f16.cpp    launch helpers of the fp16 MFMA filter kernels (kernels_filter.h), query prep for them
//   nvdb_launch_i8.cpp     ... of the int8 kernels (kernels_filter.h, kernels_filter_i8s.h)
//   nvdb_launch_exact.cpp  ... of the exact fp32-order kernels, select / rescore / merge, the any-k path (kernels_exact*.h, kernels_largek.h)
//   nvdb_refine.cpp        exact-L2 refine (kernels_refine.h)
//   nvdb_debug.cpp         developer entry points (libnvdb_hip_dev.so only)
//   nvdb_group.cpp         device group, layered on the public ABI (does not include this header)
// There is NO CPU fallback anywhere in these files: without a working HIP device every entry point that computes returns
// NVDB_ERR_HIP.
#pragma once
// the library is built with -fvisibility=hidden: only what the public headers declare is exported
#pragma GCC visibility push(default)
#include "../../include/nvdb_hip.h"
#ifdef NVDB_HIP_DEV
#include "../../include/nvdb_hip_dev.h"
#endif
#pragma GCC visibility pop

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "kernels_exact.h"       // Cand, FinalSelect
#include "kernels_filter.h"      // Hit, ScatterArgs, PrepInit, tile constants
#include "nvdb_common.h"

using namespace nvdbhip;

namespace nvdbhip {

constexpr uint32_t SELECT_MAX_CAP = 8192;     // 64 KB of LDS in select_kernel
constexpr uint32_t WAVE_KMAX = 64;            // k the wavefront-resident top-k lists hold (entry j in lane j)
constexpr uint32_t FILTER_KMAX = 1024;        // k the filter path's 8192-entry lists (and its bootstrap over 8k tile maxima) hold
constexpr float FILTER_REL_F16 = 7.5e-4f;     // |filter - reference| <= REL * ||q|| * max||x||   (DESIGN.md "error bound")
constexpr uint32_t PROG_SLOTS = 16;           // filter launches per search whose rendezvous counters the init kernel pre-clears
constexpr uint32_t F16_FILTER_MAX_DIM = 3072;
constexpr uint32_t I8W_TILE_ROWS = 64;        // rows per tile of the int8 two-stage kernel (two 32-row blocks)
constexpr uint32_t PAD_ROWS = 64;             // zero rows every library-owned corpus / shadow is padded with: the largest tile
constexpr uint32_t I8_FILTER_MAX_DIM = 1536;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace nvdbhip

struct nvdb_hip_ctx {
  int device = 0;
  int num_cu = 256;
  hipStream_t stream = nullptr;
  std::string err;

  // resident corpus
  void* rows = nullptr;
  float* scales = nullptr;
  bool owned = false;
  uint64_t n = 0;
  uint32_t dim = 0, dtype = 0;
  uint64_t row_base = 0;
  float max_norm = 0.f;
  bool i8_scales_signed = false;                   // int8 corpus with a negative or NaN row scale: the in-loop second-stage build (no biased accumulators)
  signed char* shadow8 = nullptr;                  // int8 corpus with a dim the kernels are not instantiated for: rows zero-padded to fdim; or (q8shadow) the int8 FILTER shadow of an fp16 / fp32 corpus
  bool q8shadow = false;                           // fp16 / fp32 corpus filtered through an int8 shadow (option q8_shadow): the int8 MFMA kernels stream shadow8, every survivor is re-scored from the original rows
  float filter_max_norm = 0.f;                     // max row norm of what the int8 filter streams (== max_norm for an int8 corpus; the shadow's for q8shadow)
  float resid_max = 0.f;                           // q8shadow: largest ||x - scale * x_q|| over the rows -- the corpus side of the filter's error bound
  int64_t opt_q8_shadow = 0;                       // set before the corpus is loaded: build the int8 filter shadow for fp16 / fp32 corpora whose dim the int8 kernels take
  float* shadow8_scales = nullptr;                 // ... and its scales in a buffer padded to whole tiles
  _Float16* shadow16 = nullptr;                    // fp16 copy streamed by the MFMA filter (fp32 corpus and/or padded dim)
  uint32_t fdim = 0;                               // dim the filter kernels run at (>= dim; == dim without padding)
  DevBuf qdelta;                                   // int8: per query, what the lo plane can add to a filter value
  int64_t opt_i8_wide = 1;
  int64_t opt_i8_waves8 = 0;                       // ... on 8 waves of 32 queries (two per SIMD) instead of 4 of 64: 1 % slower (profiles/r02_i8_waves8_ab.txt), off
  int64_t opt_i8_defer = 0;                        // pipelined build: 1 = second stage inside the tile loop (deferred v_dot4 slots), 0 = log the first stage's survivors, finish them after the stream
  int64_t opt_i8_mfma16 = 1;                       // int8 batches > 128, d = 512 / 768: the pipelined build on v_mfma_i32_16x16x64_i8 (kernels_filter_i8s.h): +7.3 % (profiles/r03_i8_mfma16_ab.txt); 0 (developer build): the 32x32x32 build
  int64_t opt_i8_pipe = 1;                         // int8 batches > 128: software-pipelined build (stage-1 test in the shadow of the other row block's MFMAs)
  int64_t opt_waves8 = 1;                          // d=768: 8-wave workgroups (two waves per SIMD, 32 queries each) for the fp16 m16 kernel: +2.3 % (0: four waves x 64 queries)
  void* pinned = nullptr;                           // pinned host staging of small calls: status words, results, queries
  void* pinned_dev = nullptr;                       // ... as the device addresses it (hipHostGetDevicePointer)
  size_t pinned_bytes = 0;
  bool perm_on = false;                             // this search streams tiles in permuted order (set by search_core)
  int64_t opt_tile_permute = 1;
  uint32_t cap_hint = 0;                            // this corpus has needed the longest candidate lists before: start with them
  bool stats_lazy = false;                          // stats.candidates not read back yet (nvdb_hip_get_stats does it)
  hipEvent_t launch_e0 = nullptr, launch_e1 = nullptr;   // attached to the next filter launch (hipExtLaunchKernelGGL): its own start/stop timestamps, no extra packets
  std::vector<hipEvent_t> kl_pool;                  // recycled events of collected launches
  uint32_t prog_slot = 0;                          // next free region of the rendezvous counters (reset per search)
  int64_t opt_time_launches = 0;                   // host API with a timing struct: 1 = also attach start / stop events to every filter launch (stats.filter_kernel_ms); costs ~0.1 ms per launch-rich pass
  int64_t opt_exact_lds = 1;                       // exact MFMA kernels: full groups of 64 queries stage their row tiles through LDS once per workgroup: 1 = for fp32 rows (101 vs 75 TFLOP/s; fp16 / int8 rows are faster register-direct: 86 vs 77), 2 = always, 0 = never
  int64_t opt_exact_img = 1;                       // exact MFMA kernels, fp16 / int8 rows, full groups of 64 queries: tile converted once per workgroup into an fp32 LDS image (exact_mfma_img_kernel); 0: register-direct / raw-staged builds
  int64_t opt_exact_wgs = 1;                       // exact MFMA SCAN kernels: workgroups per CU in all (one is resident at a time).  1 = a single round: every workgroup pays the start-up of its
                                                   // top-k lists (the first ~150 tiles of a stream take the insertion path) once, and the grid (wgs * CUs / query groups, rounded DOWN) never leaves
                                                   // a partly filled last round -- 2, the value of round 3, cost 5-30 % (profiles/r04_exact_wgs_sweep.txt: 128 queries 67 -> 90 TFLOP/s)
  int64_t opt_exact_prescan = 1;                   // path 1, more than 8 queries, >= 1M rows: scan the first 1/64 of the rows on its own and start the rest with its exact k-th best scores as the bar
  int64_t opt_exact_mfma = 1;                      // exact fp32-order scores on the fp32 matrix cores where the shape allows (kernels_exact_mfma.h); 0: VALU kernels only
  int64_t opt_rescore8 = 2;                        // rescore kernel: 0 lane per candidate, 1 eight lanes per candidate, 2 = 1 + rows staged through LDS

  // grow-only workspace
  DevBuf q32, q16, qscale, qinv, ebound, slack, thr, cnt, overflow, cand, out_ids, out_scores, misc, hitlog, prog;
  DevBuf hostblock;                                // host API, <= 1024 queries: [status words (= misc, aliased) | ids | scores] in one allocation, one D2H copy
  DevBuf tickets;                                  // FUSE_TICKETS words, zeroed by the search's prep launch: "last workgroup" ticket of the fused rescore + final select
  int64_t opt_fuse = 1;                            // 1: init folded into the prep launch, the final select into the rescore launch; 0: separate launches
  int64_t opt_zero_copy = 1;                       // host API, small calls: queries read from / results written to pinned host memory by the kernels themselves (no H2D / D2H copy enqueued)
  bool status_by_kernel = false;                   // last search_core: its final kernel wrote the status words to the caller's pinned block
  size_t q32_dirty = 0;                            // bytes of q32 (from its start) that may hold old queries: beyond them the buffer is zero
  DevBuf rq, rcand, rout_ids, rout_dist;           // refine
  DevBuf xcdw;                                     // XCD balance: 8 speed weights + 16 accumulators (kernels_filter.h ScatterArgs::xcdw)
  int64_t opt_xcd_balance = 1;
  int64_t opt_i8_lo_bits = 7;                      // int8: bits of a quantised query's lo plane (ScatterArgs::lo_bits)
  int64_t opt_boot_tiles = 0;                      // threshold bootstrap over this many 32-row tile maxima (0: max(64, 8k))
  int64_t opt_i8_small8 = 1;                       // int8 d = 512 / 768, batches <= 128: 1 = the 8-wave 16x16x64 logged build, 0 = filter_i8w_kernel<768, 1> (developer library)
  int64_t dbg_rows = 0;                            // developer build: rows the stamped launches of nvdb_hip_debug_clock_i8 cover (0: the corpus)
  DevBuf lk_scores, lk_sel, lk_hist, lk_state;     // any-k path (kernels_largek.h): score matrix of a query sub-batch, selected keys, radix state
  int64_t opt_refine_pinned = 0;                   // refine host call: stage queries / candidates / results through pinned host buffers (reference CUDA_PINNED)
  void* rpinned = nullptr;                         // ... [queries | candidates | out ids | out dist]
  size_t rpinned_bytes = 0;
  int64_t opt_largek_budget_mb = 8192;             // HBM the any-k path may use for its score matrix

  // options
  int64_t opt_path = 0, opt_chunk0 = 512, opt_cap = 0, opt_min_filter_batch = 1, opt_growth = 0;   // opt_growth 0 = automatic

  // state of the last search
  nvdb_hip_scan_stats stats{};
  uint32_t last_nq = 0, last_cap = 0;
  bool last_filter = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, int>> ev_filter;      // (start,stop) event indices of filter launches (last search)
  // kernel-time accounting across searches ("time_kernels" option): one entry per dominant-kernel launch
  struct KLaunch { hipEvent_t e0, e1; double flops, bytes; };
  std::vector<KLaunch> klaunch;
  int64_t opt_time_kernels = 0;
  int64_t opt_sync_every = 4, opt_sync_lead = 4;   // rendezvous period (power of two, tiles) and allowed lead
  int64_t opt_sibling_sync = 1;                    // 1: co-streaming workgroups rendezvous every 8 tiles (L2 sharing)
  int64_t opt_f32_shadow = 1;                      // 1: fp32 corpora get an fp16 shadow copy for the MFMA filter
  int64_t opt_mfma_boot = 1;                       // 1: threshold bootstrap on the matrix cores (fp16 corpora)
  int64_t opt_refine_v2 = 2;                       // refine kernel: 0 lane per row, 1 column chunks through LDS, 2 whole rows through LDS (fp16 d = 256/384/512/768; else 1)
  int64_t opt_mfma16 = 1;                          // 1: use the 16x16x32 MFMA build for 256-query tiles
  std::set<const void*> lds_attr_set;              // kernels whose dynamic-LDS limit was raised on this device
};

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return NVDB_ERR_HIP;                                                                       \
    }                                                                                            \
  } while (0)

namespace nvdbhip {

inline nvdb_status fail(nvdb_hip_ctx* c, nvdb_status s, const std::string& msg) {
  c->err = msg;
  return s;
}

inline nvdb_status ensure(nvdb_hip_ctx* c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return NVDB_OK;
  if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
  size_t want = std::max<size_t>(bytes, 256);
  HIPCHK(c, hipMalloc(&b.p, want));
  b.bytes = want;
  return NVDB_OK;
}

inline size_t bpe_of(uint32_t dtype) { return dtype == NVDB_DTYPE_F32 ? 4 : (dtype == NVDB_DTYPE_F16 ? 2 : (dtype == NVDB_DTYPE_I8 ? 1 : 0)); }

inline bool aligned_rows(uint32_t dtype, uint32_t dim) {
  if (dtype == NVDB_DTYPE_F32) return dim % 4 == 0;
  return dim % 8 == 0;    // f16: 16-byte groups of 8; int8: 8-byte groups of 8
}

// dims the fp16 MFMA kernels are instantiated for: multiples of 128 up to 768 (64 queries per wave: their fragments fill
// 384 registers at 768), 1024 / 1536 on the 16-row-tile build (32 queries per wave), 2048 / 3072 on the K-split build
// (16 queries per wave, a tile streamed as two half-K stages)

inline bool f16_filter_dim(uint32_t dim) { return dim == 768 || dim == 640 || dim == 512 || dim == 384 || dim == 256 || dim == 128 || dim == 896 || dim == 1024 || dim == 1152 || dim == 1280 || dim == 1408 || dim == 1536 || dim == 2048 || dim == 2560 || dim == 3072; }
inline bool i8_filter_dim(uint32_t dim) { return dim % 128 == 0 && dim >= 256 && dim <= 1536; }   // int8 rows: every multiple of 128 bytes from 256 (swz_chunk's two families)
inline bool refine3_dim(uint32_t dim) { return dim == 768 || dim == 512 || dim == 384 || dim == 256; }   // fp16 dims of the whole-row refine kernel

// what the fp16 MFMA kernels stream: the corpus itself, or the fp16 shadow of an fp32 corpus
inline const signed char* filter_rows_i8(const nvdb_hip_ctx* c) { return c->shadow8 ? c->shadow8 : static_cast<const signed char*>(c->rows); }
inline const float* filter_scales_i8(const nvdb_hip_ctx* c) { return c->shadow8 ? c->shadow8_scales : c->scales; }
inline const _Float16* filter_rows_f16(const nvdb_hip_ctx* c) {
  return c->shadow16 ? c->shadow16 : static_cast<const _Float16*>(c->rows);
}

// the int8 MFMA kernels do the filtering: an int8 corpus, or an fp16 / fp32 corpus with an int8 filter shadow
inline bool filter_is_i8(const nvdb_hip_ctx* c) { return c->dtype == NVDB_DTYPE_I8 || c->q8shadow; }
inline bool filter_supported(const nvdb_hip_ctx* c) {
  if (c->q8shadow) return true;
  if (c->dtype == NVDB_DTYPE_F16) return f16_filter_dim(c->dim) || c->shadow16 != nullptr;
  if (c->dtype == NVDB_DTYPE_F32) return c->shadow16 != nullptr;
  if (c->dtype == NVDB_DTYPE_I8) return i8_filter_dim(c->dim) || c->shadow8 != nullptr;
  return false;
}

// int8: the two-stage kernel (hi plane resident, lo plane on demand); option i8_wide = 0 selects the two-plane kernel
inline bool i8_two_stage(const nvdb_hip_ctx* c) { return filter_is_i8(c) && c->opt_i8_wide; }

// NB = 32-query blocks per wave: 1 for nq <= 128 (HBM-bound regime) and for the two-plane int8 kernel, else 2
inline uint32_t filter_nb(const nvdb_hip_ctx* c, uint32_t nq) {
  if (filter_is_i8(c) && (!i8_two_stage(c) || c->fdim > 768)) return 1u;   // two-plane kernel; dims > 768: 32 queries per wave
  if (!filter_is_i8(c) && c->fdim > 768) return 1u;          // 16-row-tile build: 128 queries per workgroup
  return nq <= 128 ? 1u : 2u;
}

inline hipEvent_t get_event(nvdb_hip_ctx* c, size_t idx) {
  while (c->ev_pool.size() <= idx) { hipEvent_t e; (void)hipEventCreate(&e); c->ev_pool.push_back(e); }
  return c->ev_pool[idx];
}

// ---- defined in one translation unit, called from another ----------------------------------------------------------
// nvdb_launch_exact.cpp
nvdb_status launch_init_search(nvdb_hip_ctx* c, hipStream_t s, uint32_t nq_pad, uint32_t prog_words);
nvdb_status launch_scan_exact(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, const float* q32, uint32_t nq, uint32_t k,
                              const float* thr, uint32_t cap, uint32_t reserve);
nvdb_status launch_select(nvdb_hip_ctx* c, hipStream_t s, uint32_t nq, uint32_t cap, uint32_t k, const float* slack, int mode, uint64_t* out_ids,
                          float* out_scores, uint32_t out_k);
nvdb_status launch_rescore(nvdb_hip_ctx* c, hipStream_t s, const float* q32, uint32_t nq, uint32_t cap, FinalSelect fs = FinalSelect{}, bool* fused = nullptr);
nvdb_status search_largek(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t k, uint64_t* dev_out_ids, float* dev_out_scores,
                          uint32_t n_rows = 0, Cand* seed_cand = nullptr, uint32_t* seed_cnt = nullptr, uint32_t seed_cap = 0);
// nvdb_launch_f16.cpp / nvdb_launch_i8.cpp
nvdb_status launch_prep_q16(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t nq_pad, const PrepInit& pinit);
nvdb_status launch_prep_q8(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t nq_pad, const PrepInit& pinit);
nvdb_status launch_filter_f16(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap);
nvdb_status launch_filter_i8(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap);
nvdb_status launch_boot_f16(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap, uint32_t nb);
nvdb_status launch_boot_i8(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap);
// nvdb_search.cpp
nvdb_status next_prog_region(nvdb_hip_ctx* c, hipStream_t s, uint32_t nwg, uint32_t** out);
ScatterArgs scatter_args(nvdb_hip_ctx* c, uint32_t cap, uint32_t trows = 0);

}  // namespace nvdbhip
