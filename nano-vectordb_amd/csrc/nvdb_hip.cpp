// nvdb_hip.cpp -- implementation of the C ABI in include/nvdb_hip.h (compiled with hipcc for gfx950).
//
// Orchestration of one flat search (DESIGN.md "pipeline"):
//   path 1 (exact):   scan_exact_kernel over the whole corpus -> select(final)
//   path 2 (filter):  prep_q16 -> scan_exact on the first chunk0 rows (threshold bootstrap)
//                     -> select(threshold) -> { filter_f16_kernel on a chunk -> select(threshold) }*
//                     with doubling chunks -> rescore_kernel (exact fp32 order) -> select(final)
// There is NO CPU fallback anywhere in this file: without a working HIP device every entry point
// that computes returns NVDB_ERR_HIP.
#include "../../include/nvdb_hip.h"
#ifdef NVDB_HIP_DEV
#include "../../include/nvdb_hip_dev.h"
#endif

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "kernels_exact.h"
#include "kernels_exact_mfma.h"
#include "kernels_filter.h"
#include "kernels_filter_i8s.h"
#include "kernels_largek.h"
#include "kernels_refine.h"
#include "nvdb_common.h"

using namespace nvdbhip;

namespace {

std::string g_create_err;

constexpr uint32_t SELECT_MAX_CAP = 8192;     // 64 KB of LDS in select_kernel
constexpr uint32_t WAVE_KMAX = 64;            // k the wavefront-resident top-k lists hold (entry j in lane j)
constexpr uint32_t FILTER_KMAX = 1024;        // k the filter path's 8192-entry lists (and its bootstrap over 8k tile maxima) hold
constexpr float FILTER_REL_F16 = 7.5e-4f;     // |filter - reference| <= REL * ||q|| * max||x||   (DESIGN.md "error bound")

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

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
  signed char* shadow8 = nullptr;                  // int8 corpus with a dim the kernels are not instantiated for: rows zero-padded to fdim
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
  int64_t opt_exact_mfma = 1;                      // exact fp32-order scores on the fp32 matrix cores where the shape allows (kernels_exact_mfma.h); 0: VALU kernels only
  int64_t opt_rescore8 = 2;                        // rescore kernel: 0 lane per candidate, 1 eight lanes per candidate, 2 = 1 + rows staged through LDS

  // grow-only workspace
  DevBuf q32, q16, qscale, qinv, ebound, slack, thr, cnt, overflow, cand, out_ids, out_scores, misc, hitlog, prog;
  DevBuf hostblock;                                // host API, <= 1024 queries: [status words (= misc, aliased) | ids | scores] in one allocation, one D2H copy
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

namespace {

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return NVDB_ERR_HIP;                                                                       \
    }                                                                                            \
  } while (0)

nvdb_status fail(nvdb_hip_ctx* c, nvdb_status s, const std::string& msg) {
  c->err = msg;
  return s;
}

nvdb_status ensure(nvdb_hip_ctx* c, DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return NVDB_OK;
  if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
  size_t want = std::max<size_t>(bytes, 256);
  HIPCHK(c, hipMalloc(&b.p, want));
  b.bytes = want;
  return NVDB_OK;
}

// The host API's result block: the 64 bytes of status words (c->misc points INTO the block from then on), ids, scores -- one
// allocation, so that one copy brings down everything a search's caller waits for.  Growing it carries the status words
// (the sticky ones outlive a search) over to the new allocation.
nvdb_status ensure_hostblock(nvdb_hip_ctx* c, size_t out_bytes) {
  const size_t need = 64 + out_bytes;
  if (c->hostblock.p && c->hostblock.bytes >= need) return NVDB_OK;
  HIPCHK(c, hipDeviceSynchronize());
  void* np = nullptr;
  const size_t want = std::max<size_t>(need + need / 2, static_cast<size_t>(1) << 20);
  HIPCHK(c, hipMalloc(&np, want));
  if (c->misc.p) HIPCHK(c, hipMemcpy(np, c->misc.p, 64, hipMemcpyDeviceToDevice));
  else HIPCHK(c, hipMemset(np, 0, 64));
  if (c->hostblock.p) HIPCHK(c, hipFree(c->hostblock.p));
  else if (c->misc.p) HIPCHK(c, hipFree(c->misc.p));
  c->hostblock.p = np; c->hostblock.bytes = want;
  c->misc.p = np; c->misc.bytes = 64;
  return NVDB_OK;
}

size_t bpe_of(uint32_t dtype) { return dtype == NVDB_DTYPE_F32 ? 4 : (dtype == NVDB_DTYPE_F16 ? 2 : (dtype == NVDB_DTYPE_I8 ? 1 : 0)); }

bool aligned_rows(uint32_t dtype, uint32_t dim) {
  if (dtype == NVDB_DTYPE_F32) return dim % 4 == 0;
  return dim % 8 == 0;    // f16: 16-byte groups of 8; int8: 8-byte groups of 8
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

void free_corpus(nvdb_hip_ctx* c) {
  if (c->owned) {
    if (c->rows) (void)hipFree(c->rows);
    if (c->scales) (void)hipFree(c->scales);
  }
  if (c->shadow16) { (void)hipFree(c->shadow16); c->shadow16 = nullptr; }
  if (c->shadow8) { (void)hipFree(c->shadow8); c->shadow8 = nullptr; }
  if (c->shadow8_scales) { (void)hipFree(c->shadow8_scales); c->shadow8_scales = nullptr; }
  c->rows = nullptr; c->scales = nullptr; c->owned = false; c->n = 0; c->dim = 0; c->fdim = 0; c->dtype = 0; c->max_norm = 0.f;
  c->cap_hint = 0;
}

// dims the fp16 MFMA kernels are instantiated for: multiples of 128 up to 768 (64 queries per wave: their fragments fill
// 384 registers at 768), 1024 / 1536 on the 16-row-tile build (32 queries per wave), 2048 / 3072 on the K-split build
// (16 queries per wave, a tile streamed as two half-K stages)
constexpr uint32_t PROG_SLOTS = 16;   // filter launches per search whose rendezvous counters the init kernel pre-clears

bool f16_filter_dim(uint32_t dim) { return dim == 768 || dim == 640 || dim == 512 || dim == 384 || dim == 256 || dim == 128 || dim == 896 || dim == 1024 || dim == 1152 || dim == 1280 || dim == 1408 || dim == 1536 || dim == 2048 || dim == 2560 || dim == 3072; }
constexpr uint32_t F16_FILTER_MAX_DIM = 3072;
constexpr uint32_t I8W_TILE_ROWS = 64;                // rows per tile of the int8 two-stage kernel (two 32-row blocks)
constexpr uint32_t PAD_ROWS = 64;                     // zero rows every library-owned corpus / shadow is padded with: the largest tile
bool i8_filter_dim(uint32_t dim) { return dim % 128 == 0 && dim >= 256 && dim <= 1536; }   // int8 rows: every multiple of 128 bytes from 256 (swz_chunk's two families)
constexpr uint32_t I8_FILTER_MAX_DIM = 1536;
bool refine3_dim(uint32_t dim) { return dim == 768 || dim == 512 || dim == 384 || dim == 256; }   // fp16 dims of the whole-row refine kernel

nvdb_status compute_max_norm(nvdb_hip_ctx* c) {
  nvdb_status st = ensure(c, c->misc, 64);
  if (st) return st;
  HIPCHK(c, hipMemsetAsync(c->misc.p, 0, 64, c->stream));
  uint32_t* bits = static_cast<uint32_t*>(c->misc.p) + 8;
  const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((c->n + 3) / 4, 4096));
  if (c->dtype == NVDB_DTYPE_F32) row_norm_max_kernel<DT_F32><<<grid, 256, 0, c->stream>>>(c->rows, c->scales, c->n, c->dim, bits);
  else if (c->dtype == NVDB_DTYPE_F16) row_norm_max_kernel<DT_F16><<<grid, 256, 0, c->stream>>>(c->rows, c->scales, c->n, c->dim, bits);
  else row_norm_max_kernel<DT_I8><<<grid, 256, 0, c->stream>>>(c->rows, c->scales, c->n, c->dim, bits);
  HIPCHK(c, hipGetLastError());
  uint32_t h = 0, hb[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(hb, bits, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h = hb[0];
  std::memcpy(&c->max_norm, &h, 4);
  c->i8_scales_signed = hb[1] != 0;
  // Which dim do the MFMA kernels run at?  fp16 corpus with an instantiated dim: the corpus itself, no copy.
  // fp32 corpus, or fp16 with another dim <= 1536: an fp16 shadow copy, rows zero-padded to the next instantiated
  // dim (skipped when values would overflow a half).  int8: its own instantiations, no shadow.
  c->fdim = c->dim;
  if (c->dtype != NVDB_DTYPE_I8 && c->dim <= F16_FILTER_MAX_DIM && !(c->dtype == NVDB_DTYPE_F16 && f16_filter_dim(c->dim)) && c->opt_f32_shadow) {
    uint32_t sdim = 128;
    while (!f16_filter_dim(sdim) || sdim < c->dim) sdim += 128;
    const size_t count = static_cast<size_t>(c->n) * sdim;
    const size_t pad = static_cast<size_t>(PAD_ROWS) * sdim * 2 + 4096;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow16), count * 2 + pad));
    HIPCHK(c, hipMemsetAsync(reinterpret_cast<char*>(c->shadow16) + count * 2, 0, pad, c->stream));
    HIPCHK(c, hipMemsetAsync(bits, 0, 4, c->stream));
    if (c->dtype == NVDB_DTYPE_F32) shadow_f16_kernel<float><<<4096, 256, 0, c->stream>>>(static_cast<const float*>(c->rows), c->shadow16, c->n, c->dim, sdim, bits);
    else shadow_f16_kernel<_Float16><<<4096, 256, 0, c->stream>>>(static_cast<const _Float16*>(c->rows), c->shadow16, c->n, c->dim, sdim, bits);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&h, bits, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float maxabs; std::memcpy(&maxabs, &h, 4);
    if (!(maxabs < 60000.f)) { (void)hipFree(c->shadow16); c->shadow16 = nullptr; }
    else c->fdim = sdim;
  }
  if (c->dtype == NVDB_DTYPE_I8 && c->dim <= I8_FILTER_MAX_DIM && !i8_filter_dim(c->dim) && c->opt_f32_shadow) {
    uint32_t sdim = 256;
    while (!i8_filter_dim(sdim) || sdim < c->dim) sdim += 128;
    const size_t count = static_cast<size_t>(c->n) * sdim, pad = static_cast<size_t>(PAD_ROWS) * sdim + 4096;
    const size_t n_pad = (static_cast<size_t>(c->n) + PAD_ROWS - 1) / PAD_ROWS * PAD_ROWS + PAD_ROWS;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow8), count + pad));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow8_scales), n_pad * 4));
    HIPCHK(c, hipMemsetAsync(c->shadow8 + count, 0, pad, c->stream));
    HIPCHK(c, hipMemsetAsync(c->shadow8_scales, 0, n_pad * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->shadow8_scales, c->scales, static_cast<size_t>(c->n) * 4, hipMemcpyDeviceToDevice, c->stream));
    shadow_i8_kernel<<<4096, 256, 0, c->stream>>>(static_cast<const signed char*>(c->rows), c->shadow8, c->n, c->dim, sdim);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fdim = sdim;
  }
  return NVDB_OK;
}

nvdb_status check_corpus_args(nvdb_hip_ctx* c, uint64_t n, uint32_t dim, uint32_t dtype) {
  if (!c) return NVDB_ERR_INVALID;
  if (n == 0 || dim == 0) return fail(c, NVDB_ERR_INVALID, "corpus: count and dim must be > 0");
  if (bpe_of(dtype) == 0) return fail(c, NVDB_ERR_INVALID, "Unsupported base dtype (Float32/Float16/Int8 only)");
  if (n >= 0xFFFFFFF0ull) return fail(c, NVDB_ERR_UNSUPPORTED, "corpus shard must hold fewer than 2^32-16 rows (shard it)");
  return NVDB_OK;
}

// ---- kernel launch helpers ------------------------------------------------------------------------

template <int QG>
nvdb_status launch_scan_exact_qg(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, const float* q32,
                                 uint32_t nq, uint32_t k, const float* thr, uint32_t P, uint32_t cap) {
  const dim3 grid(P, (nq + QG - 1) / QG);
  Cand* cand = static_cast<Cand*>(c->cand.p);
  uint32_t* cnt = static_cast<uint32_t*>(c->cnt.p);
  uint32_t* ovf = static_cast<uint32_t*>(c->overflow.p);
  const bool al = aligned_rows(c->dtype, c->dim);
  const uint32_t qstride = (c->dim + 3u) & ~3u;
  const size_t lds = static_cast<size_t>(QG) * qstride * 4 + static_cast<size_t>(4) * QG * 64 * 8 + 4 * QG * 4;
#define NVDB_LAUNCH_SCAN(DT, AL) \
  scan_exact_kernel<DT, QG, AL><<<grid, 256, lds, s>>>(c->rows, c->scales, c->dim, row_lo, row_hi, q32, nq, 0u, k, thr, cand, cnt, cap, ovf)
  if (c->dtype == NVDB_DTYPE_F32) { if (al) NVDB_LAUNCH_SCAN(DT_F32, true); else NVDB_LAUNCH_SCAN(DT_F32, false); }
  else if (c->dtype == NVDB_DTYPE_F16) { if (al) NVDB_LAUNCH_SCAN(DT_F16, true); else NVDB_LAUNCH_SCAN(DT_F16, false); }
  else { if (al) NVDB_LAUNCH_SCAN(DT_I8, true); else NVDB_LAUNCH_SCAN(DT_I8, false); }
#undef NVDB_LAUNCH_SCAN
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// dims the fp32-MFMA exact kernels are instantiated for (dim % 32 == 0: no scalar tail; 16 queries x dim floats in registers)
bool exact_mfma_dim(uint32_t dim) { return dim == 768 || dim == 640 || dim == 512 || dim == 384 || dim == 256 || dim == 128; }

// The exact scan on the fp32 matrix cores (kernels_exact_mfma.h).  A query receives at most P * nslice * k list entries, nslice = 4 / (16-query
// blocks of its group of 64, rounded up to 1, 2 or 4).
template <int DT>
nvdb_status launch_scan_exact_mfma(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, const float* q32,
                                   uint32_t nq, uint32_t k, const float* thr, uint32_t cap, uint32_t reserve) {
  const uint32_t tiles = (row_hi - row_lo + EXACT_MFMA_ROWS - 1) / EXACT_MFMA_ROWS;
  Cand* cand = static_cast<Cand*>(c->cand.p);
  uint32_t* cnt = static_cast<uint32_t*>(c->cnt.p);
  uint32_t* ovf = static_cast<uint32_t*>(c->overflow.p);
  uint32_t q0 = 0;                                  // queries [0, q0) are served by the LDS-staged kernel (full groups of 64)
  bool lds_done = false;
#define NVDB_SCAN_LDS(D)                                                                                                           \
  if constexpr (exact_lds_shape<DT, D>()) {                                                                                        \
    if ((c->opt_exact_lds == 2 || (c->opt_exact_lds == 1 && DT == DT_F32)) && nq >= 64 && c->dim == D) {                          \
      const uint32_t gy = nq / 64;                                                                                                 \
      const uint32_t pmax = (cap > reserve + k) ? (cap - reserve) / k : 1;                                                         \
      uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu) + gy - 1) / gy);                                  \
      P = std::max<uint32_t>(1, std::min(std::min(P, std::max<uint32_t>(1, tiles / 8)), pmax));                                    \
      const void* fn = reinterpret_cast<const void*>(exact_mfma_lds_kernel<DT, D, false>);                                         \
      if (!c->lds_attr_set.count(fn)) {                                                                                            \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, exact_lds_bytes<DT, D>()));                  \
        c->lds_attr_set.insert(fn);                                                                                                \
      }                                                                                                                            \
      exact_mfma_lds_kernel<DT, D, false><<<dim3(P, gy), 256, exact_lds_bytes<DT, D>(), s>>>(c->rows, c->scales, row_lo, row_hi, q32, gy * 64, k, thr, \
                                                                                             cand, cnt, cap, ovf, nullptr, 0);     \
      HIPCHK(c, hipGetLastError());                                                                                                \
      q0 = gy * 64; lds_done = true;                                                                                               \
    }                                                                                                                              \
  }
  NVDB_SCAN_LDS(768) NVDB_SCAN_LDS(640) NVDB_SCAN_LDS(512) NVDB_SCAN_LDS(384) NVDB_SCAN_LDS(256) NVDB_SCAN_LDS(128)
#undef NVDB_SCAN_LDS
  (void)lds_done;
  if (q0 >= nq) return NVDB_OK;
  // the rest (fewer than 64 queries, or everything when the LDS-staged build does not take the shape): register-direct kernel
  const uint32_t nr = nq - q0;
  const uint32_t gy = (nr + 63) / 64;
  const uint32_t last_blocks = (nr - (gy - 1) * 64 + 15) / 16;                    // 16-query blocks of the last (partial) group
  const uint32_t nslice_max = last_blocks >= 3 ? 1u : (last_blocks == 2 ? 2u : 4u);
  const uint32_t pmax = (cap > reserve + k * nslice_max) ? (cap - reserve) / (k * nslice_max) : 1;
  uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu) + gy - 1) / gy);   // ~2 workgroups per CU in all (one resident at a time)
  P = std::min(P, std::max<uint32_t>(1, tiles / 8));                                           // >= 8 tiles each
  P = std::max<uint32_t>(1, std::min(P, pmax));
  const dim3 grid(P, gy);
  const float* qr = q32 + static_cast<size_t>(q0) * c->dim;
  const float* thr_r = thr ? thr + q0 : nullptr;
#define NVDB_SCAN_MFMA(D) scan_exact_mfma_kernel<DT, D><<<grid, 256, 0, s>>>(c->rows, c->scales, row_lo, row_hi, qr, nr, k, thr_r, cand + static_cast<size_t>(q0) * cap, cnt + q0, cap, ovf + q0)
  switch (c->dim) {
    case 768: NVDB_SCAN_MFMA(768); break;
    case 640: NVDB_SCAN_MFMA(640); break;
    case 512: NVDB_SCAN_MFMA(512); break;
    case 384: NVDB_SCAN_MFMA(384); break;
    case 256: NVDB_SCAN_MFMA(256); break;
    default: NVDB_SCAN_MFMA(128); break;
  }
#undef NVDB_SCAN_MFMA
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// rows [row_lo,row_hi) x all nq queries; appends at most P*k entries per query
nvdb_status launch_scan_exact(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, const float* q32,
                              uint32_t nq, uint32_t k, const float* thr, uint32_t cap, uint32_t reserve) {
  if (k > WAVE_KMAX) return fail(c, NVDB_ERR_INTERNAL, "scan_exact: k beyond the wavefront-resident lists (64)");
  // more than 8 queries, whole MFMA K-steps and enough rows for the tiles: the fp32 matrix cores (same bits, ~4x the rate)
  if (c->opt_exact_mfma && nq > 8 && exact_mfma_dim(c->dim) && row_hi - row_lo >= 64u * EXACT_MFMA_ROWS) {
    if (c->dtype == NVDB_DTYPE_F32) return launch_scan_exact_mfma<DT_F32>(c, s, row_lo, row_hi, q32, nq, k, thr, cap, reserve);
    if (c->dtype == NVDB_DTYPE_F16) return launch_scan_exact_mfma<DT_F16>(c, s, row_lo, row_hi, q32, nq, k, thr, cap, reserve);
    return launch_scan_exact_mfma<DT_I8>(c, s, row_lo, row_hi, q32, nq, k, thr, cap, reserve);
  }
  uint32_t QG = nq >= 8 ? 8 : (nq >= 4 ? 4 : (nq >= 2 ? 2 : 1));
  while (QG > 1 && static_cast<size_t>(QG) * (((c->dim + 3u) & ~3u) * 4 + 4 * 64 * 8 + 16) > 60 * 1024) QG >>= 1;   // LDS budget
  if (static_cast<size_t>((c->dim + 3u) & ~3u) * 4 + 4 * 64 * 8 + 16 > 60 * 1024) return fail(c, NVDB_ERR_UNSUPPORTED, "dim too large for the exact kernel's LDS query staging (max ~14800)");
  const uint32_t gy = (nq + QG - 1) / QG;
  const uint32_t rows = row_hi - row_lo;
  uint32_t pmax = (cap > reserve + k) ? (cap - reserve) / k : 1;       // list capacity
  uint32_t P = std::max<uint32_t>(1, (8u * static_cast<uint32_t>(c->num_cu) + gy - 1) / gy);   // ~8 workgroups per CU
  P = std::min(P, std::max<uint32_t>(1, (rows + 255) / 256));                                   // >= one 256-row sweep each
  P = std::max<uint32_t>(1, std::min(P, pmax));
  switch (QG) {
    case 8: return launch_scan_exact_qg<8>(c, s, row_lo, row_hi, q32, nq, k, thr, P, cap);
    case 4: return launch_scan_exact_qg<4>(c, s, row_lo, row_hi, q32, nq, k, thr, P, cap);
    case 2: return launch_scan_exact_qg<2>(c, s, row_lo, row_hi, q32, nq, k, thr, P, cap);
    default: return launch_scan_exact_qg<1>(c, s, row_lo, row_hi, q32, nq, k, thr, P, cap);
  }
}

nvdb_status launch_select(nvdb_hip_ctx* c, hipStream_t s, uint32_t nq, uint32_t cap, uint32_t k, const float* slack,
                          int mode, uint64_t* out_ids, float* out_scores, uint32_t out_k) {
  const void* fn = reinterpret_cast<const void*>(select_kernel);
  if (!c->lds_attr_set.count(fn)) {
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SELECT_MAX_CAP * sizeof(Cand)));
    c->lds_attr_set.insert(fn);
  }
  // the bitonic branch pads a list to the next power of two >= its length (<= cap): size the LDS for that
  uint32_t cap2 = 1;
  while (cap2 < cap) cap2 <<= 1;
  static_assert(SELECT_MAX_CAP * sizeof(Cand) <= 64 * 1024 && (SELECT_MAX_CAP & (SELECT_MAX_CAP - 1)) == 0, "select_kernel LDS");
  select_kernel<<<nq, 256, cap2 * sizeof(Cand), s>>>(static_cast<Cand*>(c->cand.p), static_cast<uint32_t*>(c->cnt.p), cap, k, slack,
                                                   static_cast<float*>(c->thr.p), static_cast<uint32_t*>(c->overflow.p), mode,
                                                   c->row_base, reinterpret_cast<unsigned long long*>(out_ids), out_scores, out_k,
                                                   static_cast<uint32_t*>(c->misc.p) + 6,
                                                   c->opt_xcd_balance ? static_cast<float*>(c->xcdw.p) : nullptr);
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

nvdb_status launch_rescore(nvdb_hip_ctx* c, hipStream_t s, const float* q32, uint32_t nq, uint32_t cap) {
  Cand* cand = static_cast<Cand*>(c->cand.p);
  const uint32_t* cnt = static_cast<const uint32_t*>(c->cnt.p);
  const float* eb = static_cast<const float*>(c->ebound.p);
  uint32_t* viol = static_cast<uint32_t*>(c->misc.p);
  unsigned long long* tot = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->misc.p) + 8);
  const size_t rs_lds = static_cast<size_t>((c->dim + 3u) & ~3u) * 4;
  const size_t row_bytes = static_cast<size_t>(c->dim) * bpe_of(c->dtype);
  uint32_t cpp = 32;                               // candidates per pass of rescore_lds_kernel: rows + query within 60 KB of LDS
  while (cpp > 1 && rs_lds + cpp * (row_bytes + 16) > 60 * 1024) cpp >>= 1;
  if (c->opt_rescore8 >= 2 && (row_bytes & 15) == 0 && cpp >= 4) {
    const size_t lds = rs_lds + cpp * (row_bytes + 16);
    if (c->dtype == NVDB_DTYPE_F32) rescore_lds_kernel<DT_F32><<<nq, 256, lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot, cpp);
    else if (c->dtype == NVDB_DTYPE_F16) rescore_lds_kernel<DT_F16><<<nq, 256, lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot, cpp);
    else rescore_lds_kernel<DT_I8><<<nq, 256, lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot, cpp);
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  if (c->opt_rescore8) {
    if (c->dtype == NVDB_DTYPE_F32) rescore8_kernel<DT_F32><<<nq, 256, rs_lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot);
    else if (c->dtype == NVDB_DTYPE_F16) rescore8_kernel<DT_F16><<<nq, 256, rs_lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot);
    else rescore8_kernel<DT_I8><<<nq, 256, rs_lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot);
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  const bool al = aligned_rows(c->dtype, c->dim);
#define NVDB_LAUNCH_RS(DT, AL) rescore_kernel<DT, AL><<<nq, 256, rs_lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot)
  if (c->dtype == NVDB_DTYPE_F32) { if (al) NVDB_LAUNCH_RS(DT_F32, true); else NVDB_LAUNCH_RS(DT_F32, false); }
  else if (c->dtype == NVDB_DTYPE_F16) { if (al) NVDB_LAUNCH_RS(DT_F16, true); else NVDB_LAUNCH_RS(DT_F16, false); }
  else { if (al) NVDB_LAUNCH_RS(DT_I8, true); else NVDB_LAUNCH_RS(DT_I8, false); }
#undef NVDB_LAUNCH_RS
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// rendezvous counters for the next filter launch: a region the init kernel already cleared, or (past PROG_SLOTS
// launches in one search) a region cleared here
nvdb_status next_prog_region(nvdb_hip_ctx* c, hipStream_t s, uint32_t nwg, uint32_t** out) {
  const size_t region_words = static_cast<size_t>(c->num_cu) * 8;
  if (nwg * 8u > region_words) return fail(c, NVDB_ERR_INTERNAL, "rendezvous region too small for this grid");
  uint32_t* base = static_cast<uint32_t*>(c->prog.p) + static_cast<size_t>(c->prog_slot % PROG_SLOTS) * region_words;
  if (c->prog_slot >= PROG_SLOTS) HIPCHK(c, hipMemsetAsync(base, 0xFF, region_words * 4, s));
  ++c->prog_slot;
  *out = base;
  return NVDB_OK;
}

// trows = rows per tile of the kernel being launched (0: identity tile order).  With the permutation on, logical tile g
// of the corpus' T = ceil-or-floor(n / trows) tiles is streamed from physical tile perm_tile(g) (kernels_filter.h).
ScatterArgs scatter_args(nvdb_hip_ctx* c, uint32_t cap, uint32_t trows = 0) {
  ScatterArgs a{static_cast<Cand*>(c->cand.p), static_cast<uint32_t*>(c->cnt.p), static_cast<uint32_t*>(c->overflow.p),
                static_cast<uint32_t*>(c->misc.p) + 1, cap, static_cast<uint32_t>(c->n), 1u, 0u, 0u,
                c->opt_xcd_balance ? static_cast<float*>(c->xcdw.p) : nullptr, static_cast<uint32_t>(c->opt_i8_lo_bits)};
  if (trows && c->perm_on) {
    const bool padded = c->owned || c->shadow16 != nullptr || c->shadow8 != nullptr;
    const uint32_t n = static_cast<uint32_t>(c->n);
    const uint32_t T = padded ? (n + trows - 1) / trows : n / trows;
    perm_params(T, a.perm_mul, a.perm_mask);
    a.perm_T = T;
  }
  return a;
}

// what the fp16 MFMA kernels stream: the corpus itself, or the fp16 shadow of an fp32 corpus
const signed char* filter_rows_i8(const nvdb_hip_ctx* c) { return c->shadow8 ? c->shadow8 : static_cast<const signed char*>(c->rows); }
const float* filter_scales_i8(const nvdb_hip_ctx* c) { return c->shadow8 ? c->shadow8_scales : c->scales; }
const _Float16* filter_rows_f16(const nvdb_hip_ctx* c) {
  return c->shadow16 ? c->shadow16 : static_cast<const _Float16*>(c->rows);
}

bool filter_supported(const nvdb_hip_ctx* c) {
  if (c->dtype == NVDB_DTYPE_F16) return f16_filter_dim(c->dim) || c->shadow16 != nullptr;
  if (c->dtype == NVDB_DTYPE_F32) return c->shadow16 != nullptr;
  if (c->dtype == NVDB_DTYPE_I8) return i8_filter_dim(c->dim) || c->shadow8 != nullptr;
  return false;
}

template <int DIM, int NB>
nvdb_status launch_filter_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT,
                              uint32_t cap) {
  constexpr int MBK = DIM <= 384 ? 4 : 2;                 // 16-row blocks per tile of the m16 build: 64-row tiles up to d=384
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES) * FILTER_ROWS * DIM * 2 * (DIM <= 384 ? 2 : 1);
#ifdef NVDB_HIP_DEV
  const bool m16 = (NB == 2) && c->opt_mfma16;           // developer build: option mfma16 = 0 selects the 32x32x16 build for batches > 128 (A/B only)
  constexpr bool HAS_WIDE32 = true;
#else
  const bool m16 = (NB == 2);
  constexpr bool HAS_WIDE32 = (NB == 1);                 // the product instantiates filter_f16_kernel for batches <= 128 (and as the bootstrap build) only
#endif
  const void* fn = m16 ? reinterpret_cast<const void*>(filter_f16_m16_kernel<DIM, 6, false, false, 0, MBK>) : reinterpret_cast<const void*>(filter_f16_kernel<DIM, HAS_WIDE32 ? NB : 1>);
  if (!c->lds_attr_set.count(fn)) {
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    c->lds_attr_set.insert(fn);
  }
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  nvdb_status st;
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 4 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  if (m16) {
    // only when the kernel's XCD-aware mapping is active (the QT workgroups of a row stream share an XCD label)
    const bool sync = c->opt_sibling_sync && QT > 1 && QT <= 8 && (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
    if (sync) {
      uint32_t* prog = nullptr;                    // unused / not-yet-started slots read 0xFFFFFFFF = "far ahead"
      if ((st = next_prog_region(c, s, nwg, &prog))) return st;
      if constexpr (DIM <= 768) {
        if (c->opt_waves8) {
          if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 8 * FILTER_LOGCAP * sizeof(Hit)))) return st;
          const void* f8 = reinterpret_cast<const void*>(filter_f16_m16_kernel<DIM, 4, true, false, 0, MBK, 2, 8>);
          if (!c->lds_attr_set.count(f8)) {
            HIPCHK(c, hipFuncSetAttribute(f8, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            c->lds_attr_set.insert(f8);
          }
          hipExtLaunchKernelGGL((filter_f16_m16_kernel<DIM, 4, true, false, 0, MBK, 2, 8>), dim3(nwg), dim3(512), lds, s, c->launch_e0, c->launch_e1, 0,
                                filter_rows_f16(c), row_lo, row_hi, static_cast<const _Float16*>(c->q16.p), nq, QT, static_cast<const float*>(c->thr.p),
                                static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p),
                                scatter_args(c, cap, 16 * MBK), prog, static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));
          HIPCHK(c, hipGetLastError());
          return NVDB_OK;
        }
      }
      const void* fs = reinterpret_cast<const void*>(filter_f16_m16_kernel<DIM, 6, true, false, 0, MBK>);
      if (!c->lds_attr_set.count(fs)) {
        HIPCHK(c, hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        c->lds_attr_set.insert(fs);
      }
      hipExtLaunchKernelGGL((filter_f16_m16_kernel<DIM, 6, true, false, 0, MBK>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, filter_rows_f16(c), row_lo, row_hi,
                                                                static_cast<const _Float16*>(c->q16.p), nq, QT, static_cast<const float*>(c->thr.p),
                                                                static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),
                                                                static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, 16 * MBK),
                                                                prog, static_cast<uint32_t>(c->opt_sync_every - 1),
                                                                static_cast<uint32_t>(c->opt_sync_lead));
    } else {
      hipExtLaunchKernelGGL((filter_f16_m16_kernel<DIM, 6, false, false, 0, MBK>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, filter_rows_f16(c), row_lo, row_hi,
                                                       static_cast<const _Float16*>(c->q16.p), nq, QT, static_cast<const float*>(c->thr.p),
                                                       static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),
                                                       static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, 16 * MBK), nullptr, 0u, 0u);
    }
  }
  else if constexpr (HAS_WIDE32)
    hipExtLaunchKernelGGL((filter_f16_kernel<DIM, NB>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, filter_rows_f16(c), row_lo, row_hi,
                                                     static_cast<const _Float16*>(c->q16.p), nq, QT, static_cast<const float*>(c->thr.p),
                                                     static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),
                                                     static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, FILTER_ROWS), 0u);
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// dims 2048 / 3072: K-split build, 16-row tiles in two half-K stages, 16 queries per wave, 64 per workgroup
template <int DIM>
nvdb_status launch_filter_k2_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES) * 16 * (DIM / 2) * 2;
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  nvdb_status st;
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 4 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  const bool sync = c->opt_sibling_sync && QT > 1 && QT <= 8 && (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t* prog = nullptr;
  if (sync && (st = next_prog_region(c, s, nwg, &prog))) return st;
#define NVDB_K2_LAUNCH(SYNCV)                                                                                                   \
  {                                                                                                                             \
    const void* fn = reinterpret_cast<const void*>(filter_f16_k2_kernel<DIM, SYNCV>);                                           \
    if (!c->lds_attr_set.count(fn)) {                                                                                           \
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                    \
      c->lds_attr_set.insert(fn);                                                                                               \
    }                                                                                                                           \
    hipExtLaunchKernelGGL((filter_f16_k2_kernel<DIM, SYNCV>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0,      \
                          filter_rows_f16(c), row_lo, row_hi, static_cast<const _Float16*>(c->q16.p), nq, QT,                   \
                          static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p),                          \
                          static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, 16), prog, \
                          static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));               \
  }
  if (sync) NVDB_K2_LAUNCH(true) else NVDB_K2_LAUNCH(false)
#undef NVDB_K2_LAUNCH
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// dims 1024 / 1536: 16-row tiles, 32 queries per wave (MB = 1, NQB = 2), 128 queries per workgroup
template <int DIM>
nvdb_status launch_filter_k_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES) * 16 * DIM * 2;
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  nvdb_status st;
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 4 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  const bool sync = c->opt_sibling_sync && QT > 1 && QT <= 8 && (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t* prog = nullptr;
  if (sync && (st = next_prog_region(c, s, nwg, &prog))) return st;
#define NVDB_K_LAUNCH(SYNCV)                                                                                                    \
  {                                                                                                                             \
    const void* fn = reinterpret_cast<const void*>(filter_f16_m16_kernel<DIM, 6, SYNCV, false, 0, 1, 2>);                       \
    if (!c->lds_attr_set.count(fn)) {                                                                                           \
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                    \
      c->lds_attr_set.insert(fn);                                                                                               \
    }                                                                                                                           \
    hipExtLaunchKernelGGL((filter_f16_m16_kernel<DIM, 6, SYNCV, false, 0, 1, 2>), dim3(nwg), dim3(256), lds, s, c->launch_e0,  \
                          c->launch_e1, 0, filter_rows_f16(c), row_lo, row_hi, static_cast<const _Float16*>(c->q16.p), nq, QT,  \
                          static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p),                          \
                          static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, 16), prog, \
                          static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));               \
  }
  if constexpr (DIM % 256 == 0) if (sync && c->opt_waves8) {     // (a tile's DIM / 32 pieces split over 8 waves)
    // two waves per SIMD: 8 waves x 16 queries (DIM/8 <= 192 registers of fragments per wave)
    if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 8 * FILTER_LOGCAP * sizeof(Hit)))) return st;
    const void* f8 = reinterpret_cast<const void*>(filter_f16_m16_kernel<DIM, 4, true, false, 0, 1, 1, 8>);
    if (!c->lds_attr_set.count(f8)) {
      HIPCHK(c, hipFuncSetAttribute(f8, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      c->lds_attr_set.insert(f8);
    }
    hipExtLaunchKernelGGL((filter_f16_m16_kernel<DIM, 4, true, false, 0, 1, 1, 8>), dim3(nwg), dim3(512), lds, s, c->launch_e0, c->launch_e1, 0,
                          filter_rows_f16(c), row_lo, row_hi, static_cast<const _Float16*>(c->q16.p), nq, QT, static_cast<const float*>(c->thr.p),
                          static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p),
                          scatter_args(c, cap, 16), prog, static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  if (sync) NVDB_K_LAUNCH(true) else NVDB_K_LAUNCH(false)
#undef NVDB_K_LAUNCH
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

#ifdef NVDB_HIP_DEV   // the two-plane int8 kernel (option i8_wide = 0): the reference build the two-stage kernels are compared with
template <int DIM>
nvdb_status launch_filter_i8_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT,
                                 uint32_t nq_pad, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES_I8) * (FILTER_ROWS * DIM + 4 * 1024);
  const void* fn = reinterpret_cast<const void*>(filter_i8_kernel<DIM>);
  if (!c->lds_attr_set.count(fn)) {
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    c->lds_attr_set.insert(fn);
  }
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  nvdb_status st;
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 4 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  const signed char* qhi = static_cast<const signed char*>(c->q16.p);
  const signed char* qlo = qhi + static_cast<size_t>(nq_pad) * DIM;
  const bool sync = c->opt_sibling_sync && QT > 1 && QT <= 8 && (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  if (sync) {
    uint32_t* prog = nullptr;
    if ((st = next_prog_region(c, s, nwg, &prog))) return st;
    const void* fs = reinterpret_cast<const void*>(filter_i8_kernel<DIM, false, 6, true>);
    if (!c->lds_attr_set.count(fs)) {
      HIPCHK(c, hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      c->lds_attr_set.insert(fs);
    }
    hipExtLaunchKernelGGL((filter_i8_kernel<DIM, false, 6, true>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, filter_rows_i8(c), filter_scales_i8(c), row_lo, row_hi, qhi, qlo, nq, QT,
                                                                static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p),
                                                                static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p),
                                                                scatter_args(c, cap, FILTER_ROWS), 0u, prog,
                                                                static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));
  } else {
    hipExtLaunchKernelGGL((filter_i8_kernel<DIM>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, filter_rows_i8(c), filter_scales_i8(c), row_lo, row_hi, qhi, qlo, nq, QT,
                                                static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p),
                                                static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p),
                                                scatter_args(c, cap, FILTER_ROWS), 0u, nullptr, 0u, 0u);
  }
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

#endif  // NVDB_HIP_DEV

// threshold bootstrap on the matrix cores (fp16): best (score,row) of every 32-row tile of rows [0,n0) per query
// -> cand[q][tile]; the caller then runs select(mode 2) to turn the k-th largest tile maximum into thr[q].
template <int DIM, int NB>
nvdb_status launch_boot_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES) * FILTER_ROWS * DIM * 2;
  const void* fn = reinterpret_cast<const void*>(filter_f16_kernel<DIM, NB, 7>);
  if (!c->lds_attr_set.count(fn)) {
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    c->lds_attr_set.insert(fn);
  }
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  filter_f16_kernel<DIM, NB, 7><<<nwg, 256, lds, s>>>(filter_rows_f16(c), 0, n0, static_cast<const _Float16*>(c->q16.p), nq, QT,
                                                      static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p),
                                                      static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->cand.p),
                                                      scatter_args(c, cap, FILTER_ROWS), cap);
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

template <int DIM, int NB>
nvdb_status launch_filter_i8w_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT,
                                  uint32_t nq_pad, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(3) * (I8W_TILE_ROWS * DIM + 4 * 1024) + 4096;   // 64-row tiles, three stages + 1 KB of lo-plane scratch per wave
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  nvdb_status st;
  if ((row_hi - row_lo) % I8W_TILE_ROWS) return fail(c, NVDB_ERR_INTERNAL, "int8 two-stage kernel: row range is not a multiple of its 64-row tile");
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 8 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  const signed char* qhi = static_cast<const signed char*>(c->q16.p);
  const signed char* qlo = qhi + static_cast<size_t>(nq_pad) * DIM;
  uint32_t* counts = static_cast<uint32_t*>(c->misc.p) + 4;            // [4], [5]: tiles past stage 0 / stage 1
  const bool sync = c->opt_sibling_sync && QT > 1 && QT <= 8 && (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
#define NVDB_I8W_LAUNCH(SYNCV, PROG, MASK, LEAD)                                                                                  \
  {                                                                                                                             \
    const void* fn = reinterpret_cast<const void*>(filter_i8w_kernel<DIM, NB, 6, SYNCV, 2>);                                           \
    if (!c->lds_attr_set.count(fn)) {                                                                                           \
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                    \
      c->lds_attr_set.insert(fn);                                                                                               \
    }                                                                                                                           \
    hipExtLaunchKernelGGL((filter_i8w_kernel<DIM, NB, 6, SYNCV, 2>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, filter_rows_i8(c), filter_scales_i8(c), row_lo, row_hi, qhi, qlo, nq, QT, \
        static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),     \
        static_cast<const float*>(c->qdelta.p), static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, I8W_TILE_ROWS), PROG, MASK, LEAD, counts); \
  }
#define NVDB_I8P_LAUNCH(SYNCV, PROG, MASK, LEAD, WPBV, DEFERV)                                                                            \
  {                                                                                                                             \
   if constexpr (WPBV == 4 || DIM != 384) {           /* (the 8-wave developer build has no d = 384 schedule) */                \
    /* stages (tile + scale copies) + per wave the deferred lo-plane rows: 4 waves x 4, 8 waves x 2 */                           \
    constexpr size_t ldsp = static_cast<size_t>(3) * (I8W_TILE_ROWS * DIM + (WPBV == 4 ? 4 : 1) * 256) + (DEFERV ? 16 * DIM : 0); \
    const void* fn = reinterpret_cast<const void*>(filter_i8p_kernel<DIM, SYNCV, false, 6, 0, WPBV, DEFERV>);                           \
    if (!c->lds_attr_set.count(fn)) {                                                                                           \
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsp)));                   \
      c->lds_attr_set.insert(fn);                                                                                               \
    }                                                                                                                           \
    hipExtLaunchKernelGGL((filter_i8p_kernel<DIM, SYNCV, false, 6, 0, WPBV, DEFERV>), dim3(nwg), dim3(64 * WPBV), ldsp, s, c->launch_e0, c->launch_e1, 0, filter_rows_i8(c), filter_scales_i8(c), row_lo, row_hi, qhi, qlo, nq, QT, \
        static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),     \
        static_cast<const float*>(c->qdelta.p), static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, I8W_TILE_ROWS), PROG, MASK, LEAD, counts); \
   }                                                                                                                            \
  }
#define NVDB_I8S_LAUNCH(SYNCV, PROG, MASK, LEAD) NVDB_I8S_LAUNCH_W(SYNCV, PROG, MASK, LEAD, 4)
#define NVDB_I8S_LAUNCH_W(SYNCV, PROG, MASK, LEAD, WPBV)                                                                          \
  {                                                                                                                             \
    if constexpr (DIM >= 384 && (WPBV == 4 || DIM == 768 || DIM == 512)) {                                                                    \
      constexpr size_t ldss = static_cast<size_t>(3) * (I8W_TILE_ROWS * DIM + WPBV * 256);                                      \
      const void* fn = reinterpret_cast<const void*>(filter_i8s_kernel<DIM, SYNCV, false, 6, 0, WPBV>);                         \
      if (!c->lds_attr_set.count(fn)) {                                                                                         \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldss)));                 \
        c->lds_attr_set.insert(fn);                                                                                             \
      }                                                                                                                         \
      hipExtLaunchKernelGGL((filter_i8s_kernel<DIM, SYNCV, false, 6, 0, WPBV>), dim3(nwg), dim3(64 * WPBV), ldss, s, c->launch_e0, c->launch_e1, 0, filter_rows_i8(c), filter_scales_i8(c), row_lo, row_hi, qhi, qlo, nq, QT, \
          static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),   \
          static_cast<const float*>(c->qdelta.p), static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, I8W_TILE_ROWS), PROG, MASK, LEAD, counts); \
    }                                                                                                                           \
  }
#ifdef NVDB_HIP_DEV
  const bool pipe = (NB == 2) && c->opt_i8_pipe;         // developer build: i8_pipe = 0 runs filter_i8w_kernel at 64 queries per wave, i8_waves8 = 1 the 8-wave pipelined build
  const bool w8 = pipe && c->opt_i8_waves8 && DIM != 384;
  constexpr bool HAS_I8W = true, HAS_I8P32 = true;
#else
  const bool pipe = (NB == 2);
  constexpr bool w8 = false;
  constexpr bool HAS_I8W = (NB == 1);                    // the product runs filter_i8w_kernel for batches <= 128 only
  constexpr bool HAS_I8P32 = (DIM < 384);                // ... and the 32x32x32 logged build only where the 16x16x64 build does not exist
#endif
  const bool defer = c->opt_i8_defer != 0 || c->i8_scales_signed;
  // batches <= 128 at d = 512 / 768: the 16x16x64 logged build on 8 waves of 32 queries (waves without queries only load): its first-stage
  // test rides in the MFMA shadow, so four busy waves stay inside the tile time the HBM stream allows (+6.5 % at batch 128,
  // +2.4 % at 64 over filter_i8w_kernel<768, 1>, profiles/r03_i8_small_batch_ab.txt); signed / huge scales keep the in-loop build
  if constexpr (NB == 1 && DIM == 384) {             // d = 384 has no 8-wave schedule: 64 < batch <= 128 on the 4-wave build, two waves without queries (+4 % at 128; equal at 64)
    if (c->opt_i8_small8 && !defer && QT == 1 && nq > 64) {
      NVDB_I8S_LAUNCH(false, nullptr, 0u, 0u)
      HIPCHK(c, hipGetLastError());
      return NVDB_OK;
    }
  }
  if constexpr (NB == 1 && (DIM == 768 || DIM == 512)) {
    if (c->opt_i8_small8 && !defer && QT == 1 && nq > 8) {      // (a handful of queries: equal within noise, the old kernel stays)
      NVDB_I8S_LAUNCH_W(false, nullptr, 0u, 0u, 8)
      HIPCHK(c, hipGetLastError());
      return NVDB_OK;
    }
  }
  const bool s16 = pipe && !w8 && !defer && c->opt_i8_mfma16 && DIM >= 384;
  [[maybe_unused]] const bool s16w8 = pipe && w8 && !defer && c->opt_i8_mfma16 && DIM == 768;        // developer build: the 16x16x64 build on 8 waves (d = 768 only; measured equal to 4 waves)
  const uint32_t smask = static_cast<uint32_t>(c->opt_sync_every - 1), slead = static_cast<uint32_t>(c->opt_sync_lead);
  if (sync) {
    uint32_t* prog = nullptr;
    if ((st = next_prog_region(c, s, nwg, &prog))) return st;
#ifdef NVDB_HIP_DEV
    if (s16w8) NVDB_I8S_LAUNCH_W(true, prog, smask, slead, 8)
    else
#endif
    if (s16) NVDB_I8S_LAUNCH(true, prog, smask, slead)
#ifdef NVDB_HIP_DEV
    else if (w8) NVDB_I8P_LAUNCH(true, prog, smask, slead, 8, true)              // the 8-wave variant exists with the in-loop second stage only
#endif
    else if (pipe && defer) NVDB_I8P_LAUNCH(true, prog, smask, slead, 4, true)
    else if (pipe) { if constexpr (HAS_I8P32) NVDB_I8P_LAUNCH(true, prog, smask, slead, 4, false) }
    else if constexpr (HAS_I8W) NVDB_I8W_LAUNCH(true, prog, smask, slead)
  } else {
#ifdef NVDB_HIP_DEV
    if (s16w8) NVDB_I8S_LAUNCH_W(false, nullptr, 0u, 0u, 8)
    else
#endif
    if (s16) NVDB_I8S_LAUNCH(false, nullptr, 0u, 0u)
#ifdef NVDB_HIP_DEV
    else if (w8) NVDB_I8P_LAUNCH(false, nullptr, 0u, 0u, 8, true)
#endif
    else if (pipe && defer) NVDB_I8P_LAUNCH(false, nullptr, 0u, 0u, 4, true)
    else if (pipe) { if constexpr (HAS_I8P32) NVDB_I8P_LAUNCH(false, nullptr, 0u, 0u, 4, false) }
    else if constexpr (HAS_I8W) NVDB_I8W_LAUNCH(false, nullptr, 0u, 0u)
  }
#undef NVDB_I8W_LAUNCH
#undef NVDB_I8P_LAUNCH
#undef NVDB_I8S_LAUNCH
#undef NVDB_I8S_LAUNCH_W
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// int8, 768 < dim <= 1536: the two-stage kernel on 32-row tiles with one 32-query block per wave (128 queries per workgroup);
// the reference takes any dim (src/simd_dot.cpp:160-213)
template <int DIM>
nvdb_status launch_filter_i8w_big_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT,
                                      uint32_t nq_pad, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(3) * (FILTER_ROWS * DIM + 4 * 1024);
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  nvdb_status st;
  if ((row_hi - row_lo) % FILTER_ROWS) return fail(c, NVDB_ERR_INTERNAL, "int8 kernel: row range is not a multiple of its 32-row tile");
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 4 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  const signed char* qhi = static_cast<const signed char*>(c->q16.p);
  const signed char* qlo = qhi + static_cast<size_t>(nq_pad) * DIM;
  uint32_t* counts = static_cast<uint32_t*>(c->misc.p) + 4;
  const bool sync = c->opt_sibling_sync && QT > 1 && QT <= 8 && (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t* prog = nullptr;
  if (sync && (st = next_prog_region(c, s, nwg, &prog))) return st;
#define NVDB_I8BIG_LAUNCH(SYNCV)                                                                                                \
  {                                                                                                                             \
    const void* fn = reinterpret_cast<const void*>(filter_i8w_kernel<DIM, 1, 6, SYNCV, 1>);                                     \
    if (!c->lds_attr_set.count(fn)) {                                                                                           \
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                    \
      c->lds_attr_set.insert(fn);                                                                                               \
    }                                                                                                                           \
    hipExtLaunchKernelGGL((filter_i8w_kernel<DIM, 1, 6, SYNCV, 1>), dim3(nwg), dim3(256), lds, s, c->launch_e0, c->launch_e1, 0, \
                          filter_rows_i8(c), filter_scales_i8(c), row_lo, row_hi, qhi, qlo, nq, QT, static_cast<const float*>(c->thr.p), \
                          static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<const float*>(c->qdelta.p), \
                          static_cast<Hit*>(c->hitlog.p), scatter_args(c, cap, FILTER_ROWS), prog,                              \
                          static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead), counts);       \
  }
  if (sync) NVDB_I8BIG_LAUNCH(true) else NVDB_I8BIG_LAUNCH(false)
#undef NVDB_I8BIG_LAUNCH
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

template <int DIM>
nvdb_status launch_boot_i8_dim(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap) {
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES_I8) * (FILTER_ROWS * DIM + 4 * 1024);
  const void* fn = reinterpret_cast<const void*>(filter_i8_kernel<DIM, true>);
  if (!c->lds_attr_set.count(fn)) {
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    c->lds_attr_set.insert(fn);
  }
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0) nwg = QT;
  const signed char* qhi = static_cast<const signed char*>(c->q16.p);
  const signed char* qlo = qhi + static_cast<size_t>(QT) * 128u * DIM;
  filter_i8_kernel<DIM, true><<<nwg, 256, lds, s>>>(filter_rows_i8(c), filter_scales_i8(c), 0, n0, qhi, qlo, nq, QT,
                                                    static_cast<const float*>(c->thr.p), static_cast<const float*>(c->qscale.p),
                                                    static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->cand.p),
                                                    scatter_args(c, cap, FILTER_ROWS), cap, nullptr, 0u, 0u);
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

nvdb_status launch_boot(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap, uint32_t nb) {
  if (c->dtype == NVDB_DTYPE_I8) {
    QT *= nb;                                      // the boot build is the 128-queries-per-workgroup kernel; same padded batch
    if (c->fdim == 768) return launch_boot_i8_dim<768>(c, s, n0, nq, QT, cap);
    if (c->fdim == 512) return launch_boot_i8_dim<512>(c, s, n0, nq, QT, cap);
    if (c->fdim == 640) return launch_boot_i8_dim<640>(c, s, n0, nq, QT, cap);
    if (c->fdim == 384) return launch_boot_i8_dim<384>(c, s, n0, nq, QT, cap);
    if (c->fdim == 256) return launch_boot_i8_dim<256>(c, s, n0, nq, QT, cap);
    return fail(c, NVDB_ERR_UNSUPPORTED, "int8 boot kernel: unsupported dim");
  }
#define NVDB_BOOT_DIM(D) if (c->fdim == D) return nb == 1 ? launch_boot_dim<D, 1>(c, s, n0, nq, QT, cap) : launch_boot_dim<D, 2>(c, s, n0, nq, QT, cap)
  NVDB_BOOT_DIM(768); NVDB_BOOT_DIM(640); NVDB_BOOT_DIM(512); NVDB_BOOT_DIM(384); NVDB_BOOT_DIM(256); NVDB_BOOT_DIM(128);
#undef NVDB_BOOT_DIM
  return fail(c, NVDB_ERR_UNSUPPORTED, "boot kernel: unsupported dim");
}

// int8: the two-stage kernel (hi plane resident, lo plane on demand); option i8_wide = 0 selects the two-plane kernel
bool i8_two_stage(const nvdb_hip_ctx* c) { return c->dtype == NVDB_DTYPE_I8 && c->opt_i8_wide; }

// NB = 32-query blocks per wave: 1 for nq <= 128 (HBM-bound regime) and for the two-plane int8 kernel, else 2
uint32_t filter_nb(const nvdb_hip_ctx* c, uint32_t nq) {
  if (c->dtype == NVDB_DTYPE_I8 && (!i8_two_stage(c) || c->fdim > 768)) return 1u;   // two-plane kernel; dims > 768: 32 queries per wave
  if (c->dtype != NVDB_DTYPE_I8 && c->fdim > 768) return 1u;          // 16-row-tile build: 128 queries per workgroup
  return nq <= 128 ? 1u : 2u;
}

nvdb_status launch_filter(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap) {
  const uint32_t nb = filter_nb(c, nq);
  if (c->dtype == NVDB_DTYPE_I8) {
    const uint32_t nq_pad = QT * 128u * nb;
    if (c->fdim == 896) return launch_filter_i8w_big_dim<896>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 1024) return launch_filter_i8w_big_dim<1024>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 1152) return launch_filter_i8w_big_dim<1152>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 1408) return launch_filter_i8w_big_dim<1408>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 1280) return launch_filter_i8w_big_dim<1280>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 1536) return launch_filter_i8w_big_dim<1536>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (i8_two_stage(c)) {
#define NVDB_I8W_DIM(D) if (c->fdim == D) return nb == 2 ? launch_filter_i8w_dim<D, 2>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap) : launch_filter_i8w_dim<D, 1>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap)
      NVDB_I8W_DIM(768); NVDB_I8W_DIM(640); NVDB_I8W_DIM(512); NVDB_I8W_DIM(384); NVDB_I8W_DIM(256);
#undef NVDB_I8W_DIM
    }
#ifdef NVDB_HIP_DEV
    if (c->fdim == 768) return launch_filter_i8_dim<768>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 512) return launch_filter_i8_dim<512>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 640) return launch_filter_i8_dim<640>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 384) return launch_filter_i8_dim<384>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
    if (c->fdim == 256) return launch_filter_i8_dim<256>(c, s, row_lo, row_hi, nq, QT, nq_pad, cap);
#endif
    return fail(c, NVDB_ERR_UNSUPPORTED, "int8 filter kernel: unsupported dim");
  }
  if (c->fdim == 2048) return launch_filter_k2_dim<2048>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 2560) return launch_filter_k2_dim<2560>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 3072) return launch_filter_k2_dim<3072>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 896) return launch_filter_k_dim<896>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 1024) return launch_filter_k_dim<1024>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 1152) return launch_filter_k_dim<1152>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 1280) return launch_filter_k_dim<1280>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 1408) return launch_filter_k_dim<1408>(c, s, row_lo, row_hi, nq, QT, cap);
  if (c->fdim == 1536) return launch_filter_k_dim<1536>(c, s, row_lo, row_hi, nq, QT, cap);
#define NVDB_FILTER_DIM(D) if (c->fdim == D) return nb == 1 ? launch_filter_dim<D, 1>(c, s, row_lo, row_hi, nq, QT, cap) : launch_filter_dim<D, 2>(c, s, row_lo, row_hi, nq, QT, cap)
  NVDB_FILTER_DIM(768); NVDB_FILTER_DIM(640); NVDB_FILTER_DIM(512); NVDB_FILTER_DIM(384); NVDB_FILTER_DIM(256); NVDB_FILTER_DIM(128);
#undef NVDB_FILTER_DIM
  return fail(c, NVDB_ERR_UNSUPPORTED, "filter kernel: unsupported dim");
}

// ---- any k (kernels_largek.h): exact scores of a query sub-batch -> radix select of the k-th key -> sort -> emit ----------
template <int QG>
nvdb_status launch_scores_exact_qg(nvdb_hip_ctx* c, hipStream_t s, const float* q32, uint32_t nq, float* out, uint64_t ld, uint32_t n) {
  const dim3 grid(std::min<uint32_t>((n + 255u) / 256u, 8u * static_cast<uint32_t>(c->num_cu)), (nq + QG - 1) / QG);
  const bool al = aligned_rows(c->dtype, c->dim);
  const size_t lds = static_cast<size_t>(QG) * ((c->dim + 3u) & ~3u) * 4;
#define NVDB_LAUNCH_SC(DT, AL) scores_exact_kernel<DT, QG, AL><<<grid, 256, lds, s>>>(c->rows, c->scales, c->dim, n, q32, nq, out, ld)
  if (c->dtype == NVDB_DTYPE_F32) { if (al) NVDB_LAUNCH_SC(DT_F32, true); else NVDB_LAUNCH_SC(DT_F32, false); }
  else if (c->dtype == NVDB_DTYPE_F16) { if (al) NVDB_LAUNCH_SC(DT_F16, true); else NVDB_LAUNCH_SC(DT_F16, false); }
  else { if (al) NVDB_LAUNCH_SC(DT_I8, true); else NVDB_LAUNCH_SC(DT_I8, false); }
#undef NVDB_LAUNCH_SC
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// the same score matrix from the fp32 matrix cores (kernels_exact_mfma.h)
nvdb_status launch_scores_exact_mfma(nvdb_hip_ctx* c, hipStream_t s, const float* q32, uint32_t nq, float* out, uint64_t ld) {
  const uint32_t n = static_cast<uint32_t>(c->n);
  const uint32_t tiles = (n + EXACT_MFMA_ROWS - 1) / EXACT_MFMA_ROWS;
  uint32_t q0 = 0;
#define NVDB_SC_LDS(DT, D)                                                                                                          \
  if constexpr (exact_lds_shape<DT, D>()) {                                                                                        \
    if ((c->opt_exact_lds == 2 || (c->opt_exact_lds == 1 && DT == DT_F32)) && nq >= 64 && c->dim == D) {                          \
      const uint32_t gy = nq / 64;                                                                                                 \
      uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu) + gy - 1) / gy);                                  \
      P = std::max<uint32_t>(1, std::min(P, std::max<uint32_t>(1, tiles / 8)));                                                    \
      const void* fn = reinterpret_cast<const void*>(exact_mfma_lds_kernel<DT, D, true>);                                          \
      if (!c->lds_attr_set.count(fn)) {                                                                                            \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, exact_lds_bytes<DT, D>()));                  \
        c->lds_attr_set.insert(fn);                                                                                                \
      }                                                                                                                            \
      exact_mfma_lds_kernel<DT, D, true><<<dim3(P, gy), 256, exact_lds_bytes<DT, D>(), s>>>(c->rows, c->scales, 0u, n, q32, gy * 64, 0u, nullptr, \
                                                                                            nullptr, nullptr, 0u, nullptr, out, ld);   \
      HIPCHK(c, hipGetLastError());                                                                                                \
      q0 = gy * 64;                                                                                                                \
    }                                                                                                                              \
  }
#define NVDB_SC_LDS_DT(DT) NVDB_SC_LDS(DT, 768) NVDB_SC_LDS(DT, 640) NVDB_SC_LDS(DT, 512) NVDB_SC_LDS(DT, 384) NVDB_SC_LDS(DT, 256) NVDB_SC_LDS(DT, 128)
  if (c->dtype == NVDB_DTYPE_F32) { NVDB_SC_LDS_DT(DT_F32) } else if (c->dtype == NVDB_DTYPE_F16) { NVDB_SC_LDS_DT(DT_F16) } else { NVDB_SC_LDS_DT(DT_I8) }
#undef NVDB_SC_LDS_DT
#undef NVDB_SC_LDS
  if (q0 >= nq) return NVDB_OK;
  const uint32_t nr = nq - q0;
  const uint32_t gy = (nr + 63) / 64;
  uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu) + gy - 1) / gy);
  P = std::max<uint32_t>(1, std::min(P, tiles / 8));
  const dim3 grid(P, gy);
  const float* qr = q32 + static_cast<size_t>(q0) * c->dim;
  float* outr = out + static_cast<size_t>(q0) * ld;
#define NVDB_SC_MFMA(DT, D) scores_exact_mfma_kernel<DT, D><<<grid, 256, 0, s>>>(c->rows, c->scales, n, qr, nr, outr, ld)
#define NVDB_SC_MFMA_DT(DT)                                                                                       \
  switch (c->dim) {                                                                                              \
    case 768: NVDB_SC_MFMA(DT, 768); break;                                                                      \
    case 640: NVDB_SC_MFMA(DT, 640); break;                                                                      \
    case 512: NVDB_SC_MFMA(DT, 512); break;                                                                      \
    case 384: NVDB_SC_MFMA(DT, 384); break;                                                                      \
    case 256: NVDB_SC_MFMA(DT, 256); break;                                                                      \
    default: NVDB_SC_MFMA(DT, 128); break;                                                                       \
  }
  if (c->dtype == NVDB_DTYPE_F32) NVDB_SC_MFMA_DT(DT_F32) else if (c->dtype == NVDB_DTYPE_F16) NVDB_SC_MFMA_DT(DT_F16) else NVDB_SC_MFMA_DT(DT_I8)
#undef NVDB_SC_MFMA_DT
#undef NVDB_SC_MFMA
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

// seed_cand != nullptr: only the rows [0, n_rows) and the result goes into the filter path's candidate lists (exact bootstrap of a
// wide-k search on dims without an MFMA bootstrap build) instead of the output arrays
nvdb_status search_largek(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t k, uint64_t* dev_out_ids, float* dev_out_scores,
                          uint32_t n_rows = 0, Cand* seed_cand = nullptr, uint32_t* seed_cnt = nullptr, uint32_t seed_cap = 0) {
  const uint32_t n = seed_cand ? n_rows : static_cast<uint32_t>(c->n);
  const uint32_t k_eff = static_cast<uint32_t>(std::min<uint64_t>(k, n));
  const uint64_t ld = (static_cast<uint64_t>(n) + 63u) & ~63ull;
  uint32_t K2 = 2;
  while (K2 < k_eff) K2 <<= 1;
  const size_t qstride_bytes = static_cast<size_t>((c->dim + 3u) & ~3u) * 4;
  if (qstride_bytes > 60 * 1024) return fail(c, NVDB_ERR_UNSUPPORTED, "dim too large for the exact kernel's LDS query staging (max ~14800)");
  uint32_t QG = nq >= 8 ? 8 : (nq >= 4 ? 4 : (nq >= 2 ? 2 : 1));
  while (QG > 1 && QG * qstride_bytes > 60 * 1024) QG >>= 1;
  // queries per sub-batch: what the score matrix + key lists may take of HBM
  const size_t per_query = ld * 4 + static_cast<size_t>(K2) * 8;
  size_t free_b = 0, total_b = 0;
  HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
  const size_t have = c->lk_scores.bytes + c->lk_sel.bytes;                     // already ours: counts as available
  const size_t budget = std::min<size_t>(static_cast<size_t>(c->opt_largek_budget_mb) << 20, (free_b + have) / 2);
  uint32_t QB = static_cast<uint32_t>(std::min<size_t>(nq, std::max<size_t>(1, budget / per_query)));
  if (QB >= QG) QB = QB / QG * QG;
  if (per_query > free_b + have) return fail(c, NVDB_ERR_HIP, "any-k path: not enough free HBM for one query's score row");
  nvdb_status st;
  if ((st = ensure(c, c->lk_scores, static_cast<size_t>(QB) * ld * 4))) return st;
  if ((st = ensure(c, c->lk_sel, static_cast<size_t>(QB) * K2 * 8))) return st;
  if ((st = ensure(c, c->lk_hist, static_cast<size_t>(QB) * 256 * 4))) return st;
  if ((st = ensure(c, c->lk_state, static_cast<size_t>(QB) * sizeof(RadixState)))) return st;
  float* scores = static_cast<float*>(c->lk_scores.p);
  unsigned long long* sel = static_cast<unsigned long long*>(c->lk_sel.p);
  uint32_t* hist = static_cast<uint32_t*>(c->lk_hist.p);
  RadixState* rst = static_cast<RadixState*>(c->lk_state.p);
  if (K2 <= 8192) {
    const void* fn = reinterpret_cast<const void*>(bitonic_lds_kernel);
    if (!c->lds_attr_set.count(fn)) {
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
      c->lds_attr_set.insert(fn);
    }
  }
  for (uint32_t q0 = 0; q0 < nq; q0 += QB) {
    const uint32_t b = std::min(QB, nq - q0);
    const float* q = dev_q + static_cast<size_t>(q0) * c->dim;
    radix_init_kernel<<<b, 256, 0, s>>>(rst, hist, b, k_eff);
    if (!seed_cand && c->opt_exact_mfma && b > 8 && exact_mfma_dim(c->dim) && n >= 64u * EXACT_MFMA_ROWS) st = launch_scores_exact_mfma(c, s, q, b, scores, ld);
    else switch (QG) {
      case 8: st = launch_scores_exact_qg<8>(c, s, q, b, scores, ld, n); break;
      case 4: st = launch_scores_exact_qg<4>(c, s, q, b, scores, ld, n); break;
      case 2: st = launch_scores_exact_qg<2>(c, s, q, b, scores, ld, n); break;
      default: st = launch_scores_exact_qg<1>(c, s, q, b, scores, ld, n); break;
    }
    if (st) return st;
    const uint32_t G = std::max<uint32_t>(1, std::min<uint32_t>((n + 255u) / 256u, (8u * static_cast<uint32_t>(c->num_cu) + b - 1) / b));
    for (int pass = 0; pass < 8; ++pass) {
      radix_hist_kernel<<<dim3(G, b), 256, 0, s>>>(scores, ld, n, pass, rst, hist);
      radix_pick_kernel<<<b, 256, 0, s>>>(rst, hist);
    }
    collect_kernel<<<dim3(G, b), 256, 0, s>>>(scores, ld, n, rst, sel, K2, k_eff);
    if (K2 <= 8192) bitonic_lds_kernel<<<b, 256, static_cast<size_t>(K2) * 8, s>>>(sel, K2);
    else
      for (uint32_t size = 2; size <= K2; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1)
          bitonic_global_step_kernel<<<dim3((K2 / 2 + 255) / 256, b), 256, 0, s>>>(sel, K2, size, stride);
    if (seed_cand) seed_lists_kernel<<<dim3((k_eff + 255) / 256, b), 256, 0, s>>>(sel, K2, scores, ld, k_eff, seed_cand + static_cast<size_t>(q0) * seed_cap, seed_cap, seed_cnt + q0);
    else emit_kernel<<<dim3((k + 255) / 256, b), 256, 0, s>>>(sel, K2, scores, ld, k_eff, k, c->row_base,
                                                             reinterpret_cast<unsigned long long*>(dev_out_ids) + static_cast<size_t>(q0) * k,
                                                             dev_out_scores + static_cast<size_t>(q0) * k);
    HIPCHK(c, hipGetLastError());
  }
  if (seed_cand) return NVDB_OK;
  c->stats.chunks = (nq + QB - 1) / QB;
  c->stats.rows_scanned = c->n;
  return NVDB_OK;
}

hipEvent_t get_event(nvdb_hip_ctx* c, size_t idx) {
  while (c->ev_pool.size() <= idx) { hipEvent_t e; (void)hipEventCreate(&e); c->ev_pool.push_back(e); }
  return c->ev_pool[idx];
}

// Enqueue one whole search of nq (<= 2048) queries resident at dev_q.  No host synchronisation.
nvdb_status search_core(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t k, uint64_t* dev_out_ids,
                        float* dev_out_scores, int force_path, bool time_filter, uint32_t cap_override = 0, bool sticky = true) {
  const int final_mode = sticky ? 1 : 3;          // select_kernel: 3 = final select without folding into the sticky self-check words
  const uint32_t k_eff = static_cast<uint32_t>(std::min<uint64_t>(k, c->n));
  const uint32_t n = static_cast<uint32_t>(c->n);
  int path = force_path ? force_path : static_cast<int>(c->opt_path);
  if (path == 0) path = (filter_supported(c) && nq >= c->opt_min_filter_batch && c->n >= 4ull * c->opt_chunk0) ? 2 : 1;
  if (path == 2 && !filter_supported(c)) return fail(c, NVDB_ERR_UNSUPPORTED, "MFMA filter path needs an fp16/fp32 corpus with dim <= 3072 or an int8 corpus with dim <= 1536");

  uint32_t cap = cap_override ? cap_override : c->opt_cap > 0 ? static_cast<uint32_t>(c->opt_cap) : std::max<uint32_t>(c->cap_hint, nq <= 64 ? SELECT_MAX_CAP : 2048u);
  cap = std::min(cap, SELECT_MAX_CAP);
  if (cap < 4 * k_eff) cap = std::min<uint32_t>(SELECT_MAX_CAP, 4 * k_eff);
  // 64 < k <= 1024 on the filter path (its kernels do not depend on k; the lists do): the longest lists, a bootstrap over
  // 8k tile maxima and chunks small enough that k * (growth - 1) new survivors + the k kept ones + the error band fit.
  // Anything else beyond the wavefront lists' 64 entries takes the any-k path.
  const bool k_wide = k_eff > WAVE_KMAX;
  // dims whose kernels have no MFMA bootstrap build (768 < dim): an EXACT bootstrap over the first 8k tiles' rows on the any-k
  // machinery (score matrix of the sample -> radix select -> the k best seed the lists), then the filter streams the rest
  const bool wide_exact_boot = k_wide && path == 2 && force_path != 1 && k_eff <= FILTER_KMAX && c->fdim > 768 &&
                               c->n >= 4ull * FILTER_ROWS * 8 * k_eff;
  const bool wide_on_filter = wide_exact_boot || (k_wide && path == 2 && force_path != 1 && k_eff <= FILTER_KMAX && c->opt_mfma_boot &&
                              c->fdim <= 768 && c->n >= 2ull * FILTER_ROWS * 8 * k_eff);
  if (wide_on_filter) cap = SELECT_MAX_CAP;
  // queries per filter workgroup: 256 / 128, or 64 on the K-split build (dims > 1536)
  const uint32_t QPB = (c->dtype != NVDB_DTYPE_I8 && c->fdim > 1536) ? 64u : 128u * filter_nb(c, nq);
  const uint32_t QT = (nq + QPB - 1) / QPB;
  const uint32_t nq_pad = QT * QPB;

  nvdb_status st;
  if ((st = ensure(c, c->thr, nq_pad * 4))) return st;
  if ((st = ensure(c, c->cnt, nq_pad * 4))) return st;
  if ((st = ensure(c, c->overflow, nq_pad * 4))) return st;
  if ((st = ensure(c, c->cand, static_cast<size_t>(nq) * cap * sizeof(Cand)))) return st;
  if ((st = ensure(c, c->misc, 64))) return st;
  // one region of sibling-rendezvous counters per filter launch of this search, all reset by the init kernel
  const uint32_t prog_words = PROG_SLOTS * static_cast<uint32_t>(c->num_cu) * 8u;
  if ((st = ensure(c, c->prog, static_cast<size_t>(prog_words) * 4))) return st;
  c->prog_slot = 0;
  // (one query tile per stream has no siblings to keep in step: nothing to reset)
  init_search_kernel<<<(nq_pad + 255) / 256, 256, 0, s>>>(static_cast<uint32_t*>(c->cnt.p), static_cast<uint32_t*>(c->overflow.p),
                                                          static_cast<float*>(c->thr.p), static_cast<uint32_t*>(c->misc.p), nq_pad,
                                                          static_cast<uint32_t*>(c->prog.p), QT > 1 ? prog_words : 0u);
  HIPCHK(c, hipGetLastError());

  c->stats = nvdb_hip_scan_stats{};
  c->stats.path = static_cast<uint32_t>(path);
  c->last_nq = nq; c->last_cap = cap; c->last_filter = (path == 2);
  c->ev_filter.clear();

  if (k_wide && !wide_on_filter) {
    // beyond the wavefront-resident lists (k <= 64) and not on the filter path: the any-k path (scores -> radix select -> sort)
    c->stats.path = 3; c->last_filter = false;
    return search_largek(c, s, dev_q, nq, k, dev_out_ids, dev_out_scores);
  }
  if (path == 1) {
    if ((st = launch_scan_exact(c, s, 0, n, dev_q, nq, k_eff, nullptr, cap, 0))) return st;
    c->stats.chunks = 1; c->stats.rows_scanned = c->n;
    return launch_select(c, s, nq, cap, k_eff, nullptr, final_mode, dev_out_ids, dev_out_scores, k);
  }

  // ---- path 2: MFMA filter ----------------------------------------------------------------------
  if ((st = ensure(c, c->q16, static_cast<size_t>(nq_pad) * c->fdim * 2))) return st;
  if ((st = ensure(c, c->qscale, nq_pad * 4))) return st;
  if ((st = ensure(c, c->qinv, nq_pad * 4))) return st;
  if ((st = ensure(c, c->ebound, nq_pad * 4))) return st;
  if ((st = ensure(c, c->slack, nq_pad * 4))) return st;
  if ((st = ensure(c, c->qdelta, nq_pad * 4))) return st;
  if (c->dtype == NVDB_DTYPE_I8)
    prep_q8_kernel<<<nq_pad, 256, 0, s>>>(dev_q, nq, c->dim, c->fdim, c->max_norm, static_cast<signed char*>(c->q16.p),
                                          static_cast<signed char*>(c->q16.p) + static_cast<size_t>(nq_pad) * c->fdim,
                                          static_cast<float*>(c->qscale.p), static_cast<float*>(c->qinv.p),
                                          static_cast<float*>(c->ebound.p), static_cast<float*>(c->slack.p), static_cast<float*>(c->qdelta.p),
                                          static_cast<uint32_t*>(c->overflow.p), static_cast<uint32_t>(c->opt_i8_lo_bits));
  else
    // fp32 corpus: the shadow adds 2^-11 relative (normal halves) and <= 2^-25 absolute per element (subnormal halves)
    prep_q16_kernel<<<nq_pad, 256, 0, s>>>(dev_q, nq, c->dim, c->fdim, c->max_norm, c->dtype == NVDB_DTYPE_F32 ? FILTER_REL_F16 + 4.9e-4f : FILTER_REL_F16,
                                           c->dtype == NVDB_DTYPE_F32 ? 3.0e-8f * std::sqrt(static_cast<float>(c->dim)) : 0.f, static_cast<uint32_t*>(c->overflow.p),
                                           static_cast<_Float16*>(c->q16.p),
                                           static_cast<float*>(c->qscale.p), static_cast<float*>(c->qinv.p),
                                           static_cast<float*>(c->ebound.p), static_cast<float*>(c->slack.p));
  HIPCHK(c, hipGetLastError());
  const float* slack = static_cast<const float*>(c->slack.p);
  // Whole tiles: a corpus this library allocated is zero-padded to a multiple of 32 rows (the padded rows are
  // dropped when the wave files its survivors); for an adopted corpus the ragged tail goes to the exact kernel.
  const bool padded = c->owned || c->shadow16 != nullptr || c->shadow8 != nullptr;          // a shadow copy is always ours, hence padded
  // chunk boundaries are whole tiles of the streaming kernel: 64 rows for the int8 two-stage kernel and for the m16
  // fp16 build at d <= 384, 32 otherwise
  const bool f16_wide_tiles = c->dtype != NVDB_DTYPE_I8 && c->fdim <= 384 && filter_nb(c, nq) == 2 && c->opt_mfma16;
  const uint32_t tile_rows = ((i8_two_stage(c) && c->fdim <= 768) || f16_wide_tiles) ? I8W_TILE_ROWS : FILTER_ROWS;
  const uint32_t n_al = padded ? (n + tile_rows - 1) / tile_rows * tile_rows : n / tile_rows * tile_rows;
  uint32_t r = 0;
  uint64_t size;
  // chunk i covers (growth-1) x the rows seen before it.  fp16: 8 (flat between 4 and 8).  int8 batches > 128: 3 --
  // tighter thresholds earlier mean fewer tiles for which the two-stage kernel needs the lo plane, and a tile costs
  // what its slowest wave costs (profiles/r01d_i8_growth_sweep.txt)
  // (with the first-stage survivors finished after the stream a flagged value costs little: 6 and a 1024-tile bootstrap on big
  // corpora, profiles/r02_i8_boot_growth_sweep.txt; the in-loop second stage wants 3)
  const bool i8_big = i8_two_stage(c) && nq > 128 && c->fdim <= 768;
  const bool i8_log = i8_big && c->opt_i8_pipe && !c->opt_i8_defer && !c->i8_scales_signed && !c->opt_i8_waves8 && c->n >= 64ull * FILTER_ROWS * 1024;
  uint64_t growth = c->opt_growth > 0 ? static_cast<uint64_t>(c->opt_growth) : (i8_log ? 6u : i8_big ? 3u : 8u);
  if (k_wide) growth = std::max<uint64_t>(2, std::min<uint64_t>(growth, cap / (3ull * k_eff)));     // k * (growth - 1) + k + band <= cap
  // T tile maxima with T >= 8k: their k-th largest is then close to the k-th best of the 32*T rows (with T == k it
  // would be the smallest tile maximum, a uselessly weak threshold)
  uint32_t boot_tiles = std::max<uint32_t>(64u, 8u * k_eff);
  if (c->opt_boot_tiles > 0) boot_tiles = std::max<uint32_t>(boot_tiles, std::min<uint32_t>(static_cast<uint32_t>(c->opt_boot_tiles), cap));
  else if (i8_log && !k_wide) boot_tiles = std::max<uint32_t>(boot_tiles, std::min<uint32_t>(1024u, cap));
  const uint32_t boot_rows = FILTER_ROWS * boot_tiles;
  const bool mfma_boot = c->opt_mfma_boot && n >= boot_rows && boot_rows / FILTER_ROWS >= k_eff &&
                         boot_rows / FILTER_ROWS <= cap &&
                         c->fdim <= 768;    // no bootstrap build of the 16-row-tile fp16 kernel / the 32-query int8 kernel: exact bootstrap chunk
  if (k_wide && !mfma_boot && !wide_exact_boot) {
    // 64 < k on the filter path needs the MFMA bootstrap (the exact bootstrap chunk's wavefront lists hold 64 entries);
    // e.g. option boot_tiles larger than the corpus: the any-k path takes the search instead
    c->stats.path = 3; c->last_filter = false;
    return search_largek(c, s, dev_q, nq, k, dev_out_ids, dev_out_scores);
  }
  // permuted tile order needs the bootstrap whose entries are discarded (the exact bootstrap chunk keeps rows [0, r))
  c->perm_on = c->opt_tile_permute && mfma_boot;
  if (mfma_boot) {
    // thresholds from the k-th largest of the 64 tile maxima of rows [0,2048); those rows are then scanned
    // again by the normal build, so the bootstrap entries are discarded (select mode 2)
    if ((st = launch_boot(c, s, boot_rows, nq, QT, cap, filter_nb(c, nq)))) return st;
    if ((st = launch_select(c, s, nq, cap, k_eff, slack, 2, nullptr, nullptr, boot_rows / FILTER_ROWS))) return st;   // mode 2: out_k = list length
    size = static_cast<uint64_t>(boot_rows) * growth;
  } else {
    // bootstrap chunk [0,r) on the exact kernel; r is a multiple of the 32-row MFMA tile
    r = std::min<uint32_t>(n_al, (static_cast<uint32_t>(k_wide ? FILTER_ROWS * 8u * k_eff : c->opt_chunk0) + tile_rows - 1) / tile_rows * tile_rows);
    if (r > n) r = n / tile_rows * tile_rows;
    if (k_wide) { if ((st = search_largek(c, s, dev_q, nq, k_eff, nullptr, nullptr, r, static_cast<Cand*>(c->cand.p), static_cast<uint32_t*>(c->cnt.p), cap))) return st; }
    else
    if ((st = launch_scan_exact(c, s, 0, r, dev_q, nq, k_eff, nullptr, cap, 0))) return st;
    if ((st = launch_select(c, s, nq, cap, k_eff, slack, 0, nullptr, nullptr, 0))) return st;
    size = static_cast<uint64_t>(r) * (growth - 1);
  }
  size_t ev = 0;
  while (r < n_al) {
    const uint32_t hi = static_cast<uint32_t>(std::min<uint64_t>(n_al, static_cast<uint64_t>(r) + size));
    nvdb_hip_ctx::KLaunch kl{nullptr, nullptr, 0.0, 0.0};
    const bool acct = c->opt_time_kernels && c->klaunch.size() < 8192;
    if (acct) {
      // the kernel's own start/stop timestamps (events attached to the launch itself: no barrier packets, no gaps)
      for (hipEvent_t* e : {&kl.e0, &kl.e1}) {
        if (!c->kl_pool.empty()) { *e = c->kl_pool.back(); c->kl_pool.pop_back(); }
        else HIPCHK(c, hipEventCreate(e));
      }
      kl.flops = 2.0 * nq * static_cast<double>(std::min(hi, n) - r) * c->dim;   // algorithmic: real queries, real rows
      kl.bytes = static_cast<double>(std::min(hi, n) - r) * (c->dtype == NVDB_DTYPE_I8 ? c->fdim + 4.0 : c->fdim * 2.0);   // rows streamed once
      c->launch_e0 = kl.e0; c->launch_e1 = kl.e1;
    } else if (time_filter) {
      c->launch_e0 = get_event(c, ev); c->launch_e1 = get_event(c, ev + 1);
    }
    if (time_filter && acct) { HIPCHK(c, hipEventRecord(get_event(c, ev), s)); }
    st = launch_filter(c, s, r, hi, nq, QT, cap);
    c->launch_e0 = nullptr; c->launch_e1 = nullptr;
    if (st) return st;
    if (time_filter && acct) { HIPCHK(c, hipEventRecord(get_event(c, ev + 1), s)); }
    if (time_filter) { c->ev_filter.emplace_back(ev, ev + 1); ev += 2; }
    if (acct) c->klaunch.push_back(kl);
    if ((st = launch_select(c, s, nq, cap, k_eff, slack, 0, nullptr, nullptr, 0))) return st;
    c->stats.chunks++;
    c->stats.rows_scanned += static_cast<uint64_t>(hi - r) * QT;
    r = hi;
    size = static_cast<uint64_t>(r) * (growth - 1);   // rows seen so far x (growth-1)
  }
  if (n_al < n) {   // ragged tail of an adopted corpus: exact scores, pruned by the current thresholds
    // (fewer than one tile of rows per workgroup: the wavefront lists' 64 entries keep every row that clears the threshold)
    if ((st = launch_scan_exact(c, s, n_al, n, dev_q, nq, std::min(k_eff, WAVE_KMAX), static_cast<const float*>(c->thr.p), cap, 0))) return st;
    c->stats.rows_scanned += static_cast<uint64_t>(n - n_al) * QT;
  }
  if ((st = launch_rescore(c, s, dev_q, nq, cap))) return st;
  return launch_select(c, s, nq, cap, k_eff, nullptr, final_mode, dev_out_ids, dev_out_scores, k);
}

}  // namespace

// =====================================================================================================
extern "C" {

int nvdb_hip_abi_version(void) { return NVDB_HIP_ABI_VERSION; }

int nvdb_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

nvdb_status nvdb_hip_create(int device_ordinal, nvdb_hip_ctx** out_ctx) {
  if (!out_ctx) return NVDB_ERR_INVALID;
  *out_ctx = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return NVDB_ERR_HIP;
  }
  if (device_ordinal < 0 || device_ordinal >= ndev) { g_create_err = "device ordinal out of range"; return NVDB_ERR_INVALID; }
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { g_create_err = hipGetErrorString(e); return NVDB_ERR_HIP; }
  auto* c = new nvdb_hip_ctx();
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_err = hipGetErrorString(e);
    delete c;
    return NVDB_ERR_HIP;
  }
  {
    float w0[24];
    for (int i = 0; i < 24; ++i) w0[i] = i < 8 ? 1.f : 0.f;
    if ((e = hipMalloc(&c->xcdw.p, sizeof(w0))) != hipSuccess || (e = hipMemcpy(c->xcdw.p, w0, sizeof(w0), hipMemcpyHostToDevice)) != hipSuccess) {
      g_create_err = hipGetErrorString(e);
      if (c->xcdw.p) (void)hipFree(c->xcdw.p);
      (void)hipStreamDestroy(c->stream);
      delete c;
      return NVDB_ERR_HIP;
    }
    c->xcdw.bytes = sizeof(w0);
  }
  *out_ctx = c;
  return NVDB_OK;
}

void nvdb_hip_destroy(nvdb_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_corpus(c);
  if (c->hostblock.p) c->misc.p = nullptr;         // (misc lives inside the host API's result block)
  for (DevBuf* b : {&c->q32, &c->q16, &c->qscale, &c->qinv, &c->ebound, &c->slack, &c->thr, &c->cnt, &c->overflow, &c->cand,
                    &c->out_ids, &c->out_scores, &c->misc, &c->hostblock, &c->hitlog, &c->prog, &c->qdelta, &c->rq, &c->rcand, &c->rout_ids, &c->rout_dist, &c->lk_scores, &c->lk_sel, &c->lk_hist, &c->lk_state, &c->xcdw})
    if (b->p) (void)hipFree(b->p);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (auto& k : c->klaunch) { (void)hipEventDestroy(k.e0); (void)hipEventDestroy(k.e1); }
  for (hipEvent_t e : c->kl_pool) (void)hipEventDestroy(e);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->rpinned) (void)hipHostFree(c->rpinned);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* nvdb_hip_last_error(const nvdb_hip_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

nvdb_status nvdb_hip_upload_corpus(nvdb_hip_ctx* c, const void* rows, const float* scales, uint64_t n, uint32_t dim,
                                   uint32_t dtype, uint64_t global_row_base) {
  nvdb_status st = check_corpus_args(c, n, dim, dtype);
  if (st) return st;
  if (!rows) return fail(c, NVDB_ERR_INVALID, "upload_corpus: null rows");
  if (dtype == NVDB_DTYPE_I8 && !scales) return fail(c, NVDB_ERR_INVALID, "upload_corpus: int8 corpus needs per-row scales");
  HIPCHK(c, hipSetDevice(c->device));
  free_corpus(c);
  const size_t bytes = static_cast<size_t>(n) * dim * bpe_of(dtype);
  const size_t pad = static_cast<size_t>(PAD_ROWS) * dim * bpe_of(dtype) + 4096;   // zero rows up to a whole tile (+ slack for vector loads)
  HIPCHK(c, hipMalloc(&c->rows, bytes + pad));
  HIPCHK(c, hipMemset(static_cast<char*>(c->rows) + bytes, 0, pad));
  c->owned = true;
  const size_t chunk = size_t(256) << 20;
  for (size_t off = 0; off < bytes; off += chunk) {
    const size_t take = std::min(chunk, bytes - off);
    HIPCHK(c, hipMemcpy(static_cast<char*>(c->rows) + off, static_cast<const char*>(rows) + off, take, hipMemcpyHostToDevice));
  }
  if (dtype == NVDB_DTYPE_I8) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->scales), (n + PAD_ROWS) * sizeof(float)));
    HIPCHK(c, hipMemset(c->scales + n, 0, PAD_ROWS * sizeof(float)));
    HIPCHK(c, hipMemcpy(c->scales, scales, n * sizeof(float), hipMemcpyHostToDevice));
  }
  c->n = n; c->dim = dim; c->dtype = dtype; c->row_base = global_row_base;
  return compute_max_norm(c);
}

nvdb_status nvdb_hip_adopt_corpus(nvdb_hip_ctx* c, void* dev_rows, float* dev_scales, uint64_t n, uint32_t dim, uint32_t dtype,
                                  uint64_t global_row_base) {
  nvdb_status st = check_corpus_args(c, n, dim, dtype);
  if (st) return st;
  if (!dev_rows) return fail(c, NVDB_ERR_INVALID, "adopt_corpus: null rows");
  if (dtype == NVDB_DTYPE_I8 && !dev_scales) return fail(c, NVDB_ERR_INVALID, "adopt_corpus: int8 corpus needs per-row scales");
  HIPCHK(c, hipSetDevice(c->device));
  free_corpus(c);
  c->rows = dev_rows; c->scales = dev_scales; c->owned = false;
  c->n = n; c->dim = dim; c->dtype = dtype; c->row_base = global_row_base;
  return compute_max_norm(c);
}

nvdb_status nvdb_hip_generate_corpus(nvdb_hip_ctx* c, uint64_t seed, uint64_t n, uint32_t dim, uint32_t dtype,
                                     uint64_t global_row_base) {
  nvdb_status st = check_corpus_args(c, n, dim, dtype);
  if (st) return st;
  HIPCHK(c, hipSetDevice(c->device));
  free_corpus(c);
  const size_t bytes = static_cast<size_t>(n) * dim * bpe_of(dtype);
  const size_t pad = static_cast<size_t>(PAD_ROWS) * dim * bpe_of(dtype) + 4096;
  HIPCHK(c, hipMalloc(&c->rows, bytes + pad));
  HIPCHK(c, hipMemset(static_cast<char*>(c->rows) + bytes, 0, pad));
  c->owned = true;
  if (dtype == NVDB_DTYPE_I8) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->scales), (n + PAD_ROWS) * sizeof(float)));
    HIPCHK(c, hipMemset(c->scales + n, 0, PAD_ROWS * sizeof(float)));
  }
  c->n = n; c->dim = dim; c->dtype = dtype; c->row_base = global_row_base;
  // launch in slabs so that a single launch stays well inside the grid-size limit
  const uint64_t slab = 1ull << 24;
  for (uint64_t r0 = 0; r0 < n; r0 += slab) {
    const uint64_t cntr = std::min(slab, n - r0);
    const unsigned grid = static_cast<unsigned>((cntr + 3) / 4);
    char* rp = static_cast<char*>(c->rows) + r0 * dim * bpe_of(dtype);
    if (dtype == NVDB_DTYPE_F32) gen_rows_kernel<DT_F32><<<grid, 256, 0, c->stream>>>(seed, global_row_base + r0, cntr, dim, rp, nullptr);
    else if (dtype == NVDB_DTYPE_F16) gen_rows_kernel<DT_F16><<<grid, 256, 0, c->stream>>>(seed, global_row_base + r0, cntr, dim, rp, nullptr);
    else gen_rows_kernel<DT_I8><<<grid, 256, 0, c->stream>>>(seed, global_row_base + r0, cntr, dim, rp, c->scales + r0);
    HIPCHK(c, hipGetLastError());
  }
  return compute_max_norm(c);
}

nvdb_status nvdb_hip_corpus_info(const nvdb_hip_ctx* c, uint64_t* n, uint32_t* dim, uint32_t* dtype, uint64_t* base, float* mx) {
  if (!c) return NVDB_ERR_INVALID;
  if (n) *n = c->n;
  if (dim) *dim = c->dim;
  if (dtype) *dtype = c->dtype;
  if (base) *base = c->row_base;
  if (mx) *mx = c->max_norm;
  return c->rows ? NVDB_OK : NVDB_ERR_NO_CORPUS;
}

nvdb_status nvdb_hip_download_rows(nvdb_hip_ctx* c, uint64_t row0, uint64_t nrows, void* rows_out, float* scales_out) {
  if (!c) return NVDB_ERR_INVALID;
  if (!c->rows) return fail(c, NVDB_ERR_NO_CORPUS, "Empty base");
  if (row0 + nrows > c->n || !rows_out) return fail(c, NVDB_ERR_INVALID, "download_rows: range out of bounds");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t rb = static_cast<size_t>(c->dim) * bpe_of(c->dtype);
  HIPCHK(c, hipMemcpy(rows_out, static_cast<const char*>(c->rows) + row0 * rb, nrows * rb, hipMemcpyDeviceToHost));
  if (scales_out && c->scales) HIPCHK(c, hipMemcpy(scales_out, c->scales + row0, nrows * sizeof(float), hipMemcpyDeviceToHost));
  return NVDB_OK;
}

nvdb_status nvdb_hip_set_option(nvdb_hip_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return NVDB_ERR_INVALID;
  const std::string k(key);
  if (k == "path") { if (value < 0 || value > 2) return fail(c, NVDB_ERR_INVALID, "path must be 0,1,2"); c->opt_path = value; }
  else if (k == "chunk0_rows") { if (value < 256) return fail(c, NVDB_ERR_INVALID, "chunk0_rows must be >= 256"); c->opt_chunk0 = value; }
  else if (k == "cand_cap") { if (value < 0 || value > SELECT_MAX_CAP) return fail(c, NVDB_ERR_INVALID, "cand_cap out of range"); c->opt_cap = value; }
  else if (k == "time_kernels") { c->opt_time_kernels = value ? 1 : 0; }
  else if (k == "sync_every") { if (value < 1 || (value & (value - 1))) return fail(c, NVDB_ERR_INVALID, "sync_every must be a power of two"); c->opt_sync_every = value; }
  else if (k == "sync_lead") { if (value < 1) return fail(c, NVDB_ERR_INVALID, "sync_lead must be >= 1"); c->opt_sync_lead = value; }
  else if (k == "sibling_sync") { c->opt_sibling_sync = value ? 1 : 0; }
  else if (k == "f32_shadow") { c->opt_f32_shadow = value ? 1 : 0; }
#ifdef NVDB_HIP_DEV
  else if (k == "mfma16") { c->opt_mfma16 = value ? 1 : 0; }
  else if (k == "i8_wide") { c->opt_i8_wide = value ? 1 : 0; }
  else if (k == "i8_pipe") { c->opt_i8_pipe = value ? 1 : 0; }
  else if (k == "i8_waves8") { c->opt_i8_waves8 = value ? 1 : 0; }
  else if (k == "i8_mfma16") { c->opt_i8_mfma16 = value ? 1 : 0; }
  else if (k == "i8_small8") { c->opt_i8_small8 = value ? 1 : 0; }
#else
  // kernel variants that lost their A/B (32x32x16 fp16 build for batches > 128, two-plane int8 kernel, filter_i8w_kernel at 64 queries
  // per wave, 8-wave int8 build, 32x32x32 int8 build at d >= 384) live in libnvdb_hip_dev.so only; the product accepts their default values
  else if (k == "mfma16" || k == "i8_wide" || k == "i8_pipe" || k == "i8_mfma16" || k == "i8_small8") { if (!value) return fail(c, NVDB_ERR_UNSUPPORTED, k + " = 0 selects a developer-build kernel variant (libnvdb_hip_dev.so)"); }
  else if (k == "i8_waves8") { if (value) return fail(c, NVDB_ERR_UNSUPPORTED, "i8_waves8 = 1 selects a developer-build kernel variant (libnvdb_hip_dev.so)"); }
#endif

  else if (k == "i8_defer") { c->opt_i8_defer = value ? 1 : 0; }
  else if (k == "xcd_balance") { c->opt_xcd_balance = value ? 1 : 0; }
  else if (k == "i8_lo_bits") { if (value < 2 || value > 7) return fail(c, NVDB_ERR_INVALID, "i8_lo_bits must be in [2,7]"); c->opt_i8_lo_bits = value; }
  else if (k == "boot_tiles") { if (value < 0 || value > SELECT_MAX_CAP) return fail(c, NVDB_ERR_INVALID, "boot_tiles out of range"); c->opt_boot_tiles = value; }
#ifdef NVDB_HIP_DEV
  else if (k == "debug_rows") { c->dbg_rows = value < 0 ? 0 : value; }
#endif
  else if (k == "waves8") { c->opt_waves8 = value ? 1 : 0; }
  else if (k == "exact_mfma") { c->opt_exact_mfma = value ? 1 : 0; }
  else if (k == "exact_lds") { if (value < 0 || value > 2) return fail(c, NVDB_ERR_INVALID, "exact_lds must be 0, 1 or 2"); c->opt_exact_lds = value; }
  else if (k == "time_launches") { c->opt_time_launches = value ? 1 : 0; }
  else if (k == "tile_permute") { c->opt_tile_permute = value ? 1 : 0; }
  else if (k == "rescore8") { c->opt_rescore8 = value < 0 ? 0 : (value > 2 ? 2 : value); }
  else if (k == "mfma_boot") { c->opt_mfma_boot = value ? 1 : 0; }
  else if (k == "refine_v2") { c->opt_refine_v2 = value < 0 ? 0 : (value > 2 ? 2 : value); }
  else if (k == "refine_pinned") { c->opt_refine_pinned = value ? 1 : 0; }
  else if (k == "largek_budget_mb") { if (value < 1) return fail(c, NVDB_ERR_INVALID, "largek_budget_mb must be >= 1"); c->opt_largek_budget_mb = value; }
  else if (k == "chunk_growth") { if (value != 0 && (value < 2 || value > 64)) return fail(c, NVDB_ERR_INVALID, "chunk_growth must be 0 (automatic) or in [2,64]"); c->opt_growth = value; }
  else if (k == "min_filter_batch") { if (value < 1) return fail(c, NVDB_ERR_INVALID, "min_filter_batch must be >= 1"); c->opt_min_filter_batch = value; }
  else return fail(c, NVDB_ERR_INVALID, "unknown option: " + k);
  return NVDB_OK;
}

static nvdb_status search_args(nvdb_hip_ctx* c, const void* q, uint32_t nq, uint32_t k, const void* oi, const void* os) {
  if (!c) return NVDB_ERR_INVALID;
  if (!c->rows || c->n == 0) return fail(c, NVDB_ERR_NO_CORPUS, "Empty base");
  if (nq > 0 && k > 0 && (!q || !oi || !os)) return fail(c, NVDB_ERR_INVALID, q ? "null output" : "Null query");
  return NVDB_OK;
}

nvdb_status nvdb_hip_search_batch_dev(nvdb_hip_ctx* c, const float* dev_q, uint32_t nq, uint32_t k, uint64_t* dev_out_ids,
                                      float* dev_out_scores, void* hip_stream) {
  nvdb_status st = search_args(c, dev_q, nq, k, dev_out_ids, dev_out_scores);
  if (st) return st;
  if (nq == 0 || k == 0) return NVDB_OK;
  if (nq > 2048) return fail(c, NVDB_ERR_UNSUPPORTED, "search_batch_dev: at most 2048 queries per call");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
  return search_core(c, s, dev_q, nq, k, dev_out_ids, dev_out_scores, 0, false);
}

// own_only: the host API's view of ITS OWN last search (misc[0], [1], per-query flags); the sticky words that device-API
// searches left for the caller's next nvdb_hip_search_check are neither read into the verdict nor cleared
static nvdb_status search_check_impl(nvdb_hip_ctx* c, nvdb_hip_scan_stats* stats, bool own_only) {
  if (!c) return NVDB_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  // caller has synchronised its stream; read the self-check words
  std::vector<uint32_t> ovf(c->last_nq);
  uint32_t misc[16] = {0};
  if (c->last_nq) HIPCHK(c, hipMemcpy(ovf.data(), c->overflow.p, c->last_nq * 4, hipMemcpyDeviceToHost));
  if (c->misc.p) {
    HIPCHK(c, hipMemcpy(misc, c->misc.p, 64, hipMemcpyDeviceToHost));
    if (own_only) misc[12] = misc[13] = misc[14] = 0;
    const uint32_t zero[3] = {0, 0, 0};             // sticky words (select_kernel): what ANY search since the last check found
    if (misc[12] | misc[13] | misc[14]) HIPCHK(c, hipMemcpy(static_cast<uint32_t*>(c->misc.p) + 12, zero, 12, hipMemcpyHostToDevice));
  }
  c->stats.sticky_overflow = (misc[12] | misc[14]) ? 1u : 0u;
  c->stats.sticky_violations = misc[13];
  c->stats.i8_stage1_tiles = misc[4]; c->stats.i8_stage2_blocks = misc[5];
  uint32_t nov = 0;
  for (uint32_t v : ovf) nov += v ? 1u : 0u;
  if (std::getenv("NVDB_DEBUG_OVERFLOW") && (nov || misc[1])) {
    std::vector<uint32_t> cn(c->last_nq);
    (void)hipMemcpy(cn.data(), c->cnt.p, c->last_nq * 4, hipMemcpyDeviceToHost);
    std::fprintf(stderr, "[nvdb debug] log_overflow=%u list flags:", misc[1]);
    for (uint32_t q = 0; q < c->last_nq; ++q) if (ovf[q]) std::fprintf(stderr, " q%u(cnt=%u)", q, cn[q]);
    std::fprintf(stderr, "\n");
  }
  if (misc[1]) nov = c->last_nq;                 // a wave's survivor log overflowed: which queries lost entries is unknown
  c->stats.overflow_queries = nov;
  c->stats.bound_violations = misc[0];
  // candidates that reached the rescore = the list lengths left by the last thresholding select (the final select
  // does not touch them); summed here rather than by 1024 same-address atomics in the rescore kernel
  unsigned long long tot = 0;
  if (c->last_filter && c->last_nq) {
    std::vector<uint32_t> cn(c->last_nq);
    HIPCHK(c, hipMemcpy(cn.data(), c->cnt.p, c->last_nq * 4, hipMemcpyDeviceToHost));
    for (uint32_t v : cn) tot += std::min(v, c->last_cap);
  }
  c->stats.candidates = tot;
  float fms = 0.f;
  for (auto& pr : c->ev_filter) { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev_pool[pr.first], c->ev_pool[pr.second]) == hipSuccess) fms += ms; }
  c->stats.filter_kernel_ms = fms;
  if (stats) *stats = c->stats;
  if (misc[0]) return fail(c, NVDB_ERR_INTERNAL, "filter error bound violated (bound_violations > 0)");
  if (nov) return fail(c, NVDB_ERR_INTERNAL, "candidate list overflow: re-run these queries with option path=1");
  if (misc[13]) return fail(c, NVDB_ERR_INTERNAL, "filter error bound violated in an earlier search since the last check");
  if (misc[12] | misc[14]) return fail(c, NVDB_ERR_INTERNAL, "candidate list overflow in an earlier search since the last check");
  return NVDB_OK;
}

nvdb_status nvdb_hip_search_check(nvdb_hip_ctx* c, nvdb_hip_scan_stats* stats) { return search_check_impl(c, stats, false); }

nvdb_status nvdb_hip_search_batch(nvdb_hip_ctx* c, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_ids,
                                  float* out_scores, uint32_t* out_k_eff, nvdb_hip_timing* timing) {
  nvdb_status st = search_args(c, queries, nq, k, out_ids, out_scores);
  if (st) return st;
  if (timing) std::memset(timing, 0, sizeof(*timing));
  if (out_k_eff) *out_k_eff = static_cast<uint32_t>(std::min<uint64_t>(k, c->n));
  if (nq == 0 || k == 0) return NVDB_OK;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t qbytes = static_cast<size_t>(nq) * c->dim * 4;
  if ((st = ensure(c, c->q32, qbytes + 8 * static_cast<size_t>(c->dim) * 4))) return st;
  if (nq <= 1024) { if ((st = ensure_hostblock(c, static_cast<size_t>(nq) * k * 12))) return st; }
  else {
    if ((st = ensure(c, c->out_ids, static_cast<size_t>(nq) * k * 8))) return st;
    if ((st = ensure(c, c->out_scores, static_cast<size_t>(nq) * k * 4))) return st;
  }
  hipEvent_t e0 = get_event(c, 60), e1 = get_event(c, 61), e2 = get_event(c, 62), e3 = get_event(c, 63);
  c->stats_lazy = false;
  if (nq <= 1024) {
    // One sub-batch: everything the host needs comes back in ONE synchronisation through pinned staging -- the
    // self-check words (32 B), ids and scores -- instead of five small pageable copies of ~20 us each (a third of
    // a single-query search at N = 1M).  Small query blocks go up through the same staging buffer.
    const size_t ob_ids = static_cast<size_t>(nq) * k * 8, ob_sc = static_cast<size_t>(nq) * k * 4;
    const size_t q_stage = qbytes <= 64 * 1024 ? qbytes : 0;
    const bool stage_out = ob_ids + ob_sc <= (static_cast<size_t>(16) << 20);     // very large k: results go straight to the caller's buffers
    const size_t need = 64 + (stage_out ? ob_ids + ob_sc : 0) + q_stage;
    if (c->pinned_bytes < need) {
      if (c->pinned) (void)hipHostFree(c->pinned);
      c->pinned = nullptr; c->pinned_bytes = 0;
      HIPCHK(c, hipHostMalloc(&c->pinned, need + need / 2, hipHostMallocDefault));
      c->pinned_bytes = need + need / 2;
    }
    char* pin = static_cast<char*>(c->pinned);
    uint32_t* pin_status = reinterpret_cast<uint32_t*>(pin);
    char* pin_ids = pin + 64; char* pin_sc = pin_ids + (stage_out ? ob_ids : 0); char* pin_q = pin_sc + (stage_out ? ob_sc : 0);
    HIPCHK(c, hipEventRecord(e0, s));
    HIPCHK(c, hipMemsetAsync(static_cast<char*>(c->q32.p) + qbytes, 0, 8 * static_cast<size_t>(c->dim) * 4, s));
    if (q_stage) { std::memcpy(pin_q, queries, qbytes); HIPCHK(c, hipMemcpyAsync(c->q32.p, pin_q, qbytes, hipMemcpyHostToDevice, s)); }
    else HIPCHK(c, hipMemcpyAsync(c->q32.p, queries, qbytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(e1, s));
    const float* dq = static_cast<const float*>(c->q32.p);
    uint64_t* oi = reinterpret_cast<uint64_t*>(static_cast<char*>(c->hostblock.p) + 64);
    float* os = reinterpret_cast<float*>(static_cast<char*>(c->hostblock.p) + 64 + ob_ids);
    if ((st = search_core(c, s, dq, nq, k, oi, os, 0, timing != nullptr && c->opt_time_launches, 0, false))) return st;
    HIPCHK(c, hipEventRecord(e2, s));
    auto fetch = [&]() -> nvdb_status {
      if (stage_out) {
        // status words, ids and scores are adjacent on the device (ensure_hostblock) and in the staging buffer: ONE copy
        HIPCHK(c, hipMemcpyAsync(pin, c->hostblock.p, 64 + ob_ids + ob_sc, hipMemcpyDeviceToHost, s));
      } else {
        HIPCHK(c, hipMemcpyAsync(pin_status, c->misc.p, 32, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(out_ids, oi, ob_ids, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(out_scores, os, ob_sc, hipMemcpyDeviceToHost, s));
      }
      HIPCHK(c, hipEventRecord(e3, s));
      HIPCHK(c, hipStreamSynchronize(s));
      return NVDB_OK;
    };
    if ((st = fetch())) return st;
    nvdb_hip_scan_stats part = c->stats;
    if (pin_status[0] | pin_status[1] | pin_status[6]) {
      // self-check tripped (rare): exact counts for the statistics, then redo the batch -- first on the filter path with
      // the longest candidate lists the select kernel can sort (near-duplicate-heavy corpora: thousands of rows inside
      // the filter's error band of the k-th score; re-scoring them is cheap, only the list was too short), and if that
      // is still not enough, or the bound itself was violated, on the always-correct exact path
      nvdb_status chk = search_check_impl(c, &part, true);
      if (chk == NVDB_ERR_HIP) return chk;
      bool done = false;
      if (part.path == 2 && !pin_status[0] && c->last_cap < SELECT_MAX_CAP) {
        if ((st = search_core(c, s, dq, nq, k, oi, os, 2, false, SELECT_MAX_CAP, false))) return st;
        if ((st = fetch())) return st;
        done = !(pin_status[0] | pin_status[1] | pin_status[6]);
        if (done) c->cap_hint = SELECT_MAX_CAP;
      }
      if (!done) {
        if ((st = search_core(c, s, dq, nq, k, oi, os, 1, false, 0, false))) return st;
        if ((st = fetch())) return st;
      }
      c->stats = part;
    } else {
      part.i8_stage1_tiles = pin_status[4]; part.i8_stage2_blocks = pin_status[5];
      float fms = 0.f;
      for (auto& pr : c->ev_filter) { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev_pool[pr.first], c->ev_pool[pr.second]) == hipSuccess) fms += ms; }
      part.filter_kernel_ms = fms;
      c->stats = part;
      c->stats_lazy = c->last_filter;                 // candidates: read back on demand
    }
    if (stage_out) {
      std::memcpy(out_ids, pin_ids, ob_ids);
      std::memcpy(out_scores, pin_sc, ob_sc);
    }
    if (timing) {
      (void)hipEventElapsedTime(&timing->h2d_ms, e0, e1);
      (void)hipEventElapsedTime(&timing->kernel_ms, e1, e2);
      (void)hipEventElapsedTime(&timing->d2h_ms, e2, e3);
      timing->total_ms = timing->h2d_ms + timing->kernel_ms + timing->d2h_ms;
      timing->threads = 256; timing->nwarps = 4; timing->K = k;
      timing->shmem_bytes = part.path != 2 ? 0 : c->dtype == NVDB_DTYPE_I8 ? static_cast<size_t>(FILTER_STAGES_I8) * (FILTER_ROWS * c->fdim + 4096) : static_cast<size_t>(FILTER_STAGES) * FILTER_ROWS * c->fdim * 2;
    }
    if (part.bound_violations) return fail(c, NVDB_ERR_INTERNAL, "filter error bound violated; results were recomputed on the exact path");
    return NVDB_OK;
  }
  HIPCHK(c, hipEventRecord(e0, s));
  HIPCHK(c, hipMemsetAsync(static_cast<char*>(c->q32.p) + qbytes, 0, 8 * static_cast<size_t>(c->dim) * 4, s));
  HIPCHK(c, hipMemcpyAsync(c->q32.p, queries, qbytes, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(e1, s));
  nvdb_hip_scan_stats total{};
  float filter_ms = 0.f;
  for (uint32_t q0 = 0; q0 < nq; q0 += 1024) {
    const uint32_t b = std::min<uint32_t>(1024, nq - q0);
    const float* dq = static_cast<const float*>(c->q32.p) + static_cast<size_t>(q0) * c->dim;
    uint64_t* oi = static_cast<uint64_t*>(c->out_ids.p) + static_cast<size_t>(q0) * k;
    float* os = static_cast<float*>(c->out_scores.p) + static_cast<size_t>(q0) * k;
    if ((st = search_core(c, s, dq, b, k, oi, os, 0, timing != nullptr && c->opt_time_launches, 0, false))) return st;
    HIPCHK(c, hipStreamSynchronize(s));
    nvdb_hip_scan_stats part{};
    nvdb_status chk = search_check_impl(c, &part, true);
    if (chk == NVDB_ERR_HIP) return chk;
    if (chk == NVDB_ERR_INTERNAL) {
      // self-check tripped: longest lists first, then the always-correct exact path (see the small-call path above)
      bool done = false;
      if (part.path == 2 && !part.bound_violations && c->last_cap < SELECT_MAX_CAP) {
        if ((st = search_core(c, s, dq, b, k, oi, os, 2, false, SELECT_MAX_CAP, false))) return st;
        HIPCHK(c, hipStreamSynchronize(s));
        nvdb_hip_scan_stats again{};
        const nvdb_status chk2 = search_check_impl(c, &again, true);
        if (chk2 == NVDB_ERR_HIP) return chk2;
        done = (chk2 == NVDB_OK);
        if (done) c->cap_hint = SELECT_MAX_CAP;
      }
      if (!done) {
        if ((st = search_core(c, s, dq, b, k, oi, os, 1, false, 0, false))) return st;
        HIPCHK(c, hipStreamSynchronize(s));
      }
    }
    total.path = std::max(total.path, part.path);
    total.chunks += part.chunks; total.rows_scanned += part.rows_scanned; total.candidates += part.candidates;
    total.overflow_queries += part.overflow_queries; total.bound_violations += part.bound_violations;
    total.i8_stage1_tiles += part.i8_stage1_tiles; total.i8_stage2_blocks += part.i8_stage2_blocks;
    filter_ms += part.filter_kernel_ms;
  }
  HIPCHK(c, hipEventRecord(e2, s));
  HIPCHK(c, hipMemcpyAsync(out_ids, c->out_ids.p, static_cast<size_t>(nq) * k * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(out_scores, c->out_scores.p, static_cast<size_t>(nq) * k * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipEventRecord(e3, s));
  HIPCHK(c, hipStreamSynchronize(s));
  total.filter_kernel_ms = filter_ms;
  c->stats = total;
  if (timing) {
    (void)hipEventElapsedTime(&timing->h2d_ms, e0, e1);
    (void)hipEventElapsedTime(&timing->kernel_ms, e1, e2);
    (void)hipEventElapsedTime(&timing->d2h_ms, e2, e3);
    timing->total_ms = timing->h2d_ms + timing->kernel_ms + timing->d2h_ms;
    timing->threads = 256; timing->nwarps = 4; timing->K = k;
    timing->shmem_bytes = total.path != 2 ? 0 : c->dtype == NVDB_DTYPE_I8 ? static_cast<size_t>(FILTER_STAGES_I8) * (FILTER_ROWS * c->fdim + 4096) : static_cast<size_t>(FILTER_STAGES) * FILTER_ROWS * c->fdim * 2;
  }
  if (total.bound_violations) return fail(c, NVDB_ERR_INTERNAL, "filter error bound violated; results were recomputed on the exact path");
  return NVDB_OK;
}

nvdb_status nvdb_hip_collect_kernel_times(nvdb_hip_ctx* c, uint32_t* launches, double* total_ms, double* total_flops,
                                          double* total_bytes) {
  if (!c) return NVDB_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  double ms = 0, fl = 0, by = 0;
  uint32_t cnt = 0;
  for (auto& k : c->klaunch) {
    float t = 0.f;
    HIPCHK(c, hipEventSynchronize(k.e1));
    HIPCHK(c, hipEventElapsedTime(&t, k.e0, k.e1));
    ms += t; fl += k.flops; by += k.bytes; ++cnt;
    c->kl_pool.push_back(k.e0); c->kl_pool.push_back(k.e1);
  }
  c->klaunch.clear();
  if (launches) *launches = cnt;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (total_bytes) *total_bytes = by;
  return NVDB_OK;
}

#ifdef NVDB_HIP_DEV
// Developer aid (not part of the drop-in surface): time ablation builds of the filter kernel on the
// resident fp16 d=768 corpus with the query workspace left by the previous path-2 search.  Thresholds
// are +inf (no survivors), so only the streaming/MFMA machinery is timed.
nvdb_status nvdb_hip_debug_filter_variant(nvdb_hip_ctx* c, int variant, uint32_t nq, uint32_t reps, float* ms_per_launch) {
  if (!c || !ms_per_launch) return NVDB_ERR_INVALID;
  if (!c->rows || c->dtype != NVDB_DTYPE_F16 || c->dim != 768 || !c->q16.p) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: run a path-2 search on an fp16 d=768 corpus first");
  HIPCHK(c, hipSetDevice(c->device));
  if (nq <= 128) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: variants are built for nq > 128 (NB = 2)");
  const uint32_t QT = (nq + 255) / 256, nq_pad = QT * 256;
  DevBuf inf;
  nvdb_status st = ensure(c, inf, nq_pad * 4);
  if (st) return st;
  fill_u32_kernel<<<(nq_pad + 255) / 256, 256, 0, c->stream>>>(static_cast<uint32_t*>(inf.p), 0x7F800000u, nq_pad);
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES) * FILTER_ROWS * 768 * 2;
  const uint32_t n_al = static_cast<uint32_t>(c->n / FILTER_ROWS * FILTER_ROWS);
  uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
#define NVDB_DBG_LAUNCH(V, RG)                                                                                                 \
  {                                                                                                                            \
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(filter_f16_kernel<768, 2, V, RG>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
    for (uint32_t r = 0; r <= reps; ++r) {                                                                                     \
      if (r == 1) HIPCHK(c, hipEventRecord(e0, c->stream));                                                                    \
      filter_f16_kernel<768, 2, V, RG><<<nwg, 256, lds, c->stream>>>(static_cast<const _Float16*>(c->rows), 0, n_al, static_cast<const _Float16*>(c->q16.p), nq, QT, \
          static_cast<const float*>(inf.p), static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<Hit*>(c->hitlog.p),     \
          scatter_args(c, c->last_cap), 0u);                                                                                   \
    }                                                                                                                          \
  }
  switch (variant) {
    case 0: NVDB_DBG_LAUNCH(0, 4) break;
    case 1: NVDB_DBG_LAUNCH(1, 4) break;
    case 2: NVDB_DBG_LAUNCH(2, 4) break;
    case 3: NVDB_DBG_LAUNCH(3, 4) break;
    case 5: NVDB_DBG_LAUNCH(5, 4) break;
    case 6: NVDB_DBG_LAUNCH(0, 6) break;
    case 10: NVDB_DBG_LAUNCH(6, 6) break;
    case 7: NVDB_DBG_LAUNCH(0, 8) break;
    case 8: NVDB_DBG_LAUNCH(0, 3) break;
    case 9: NVDB_DBG_LAUNCH(0, 12) break;
    default: return fail(c, NVDB_ERR_INVALID, "debug: unknown variant");
  }
#undef NVDB_DBG_LAUNCH
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  *ms_per_launch = ms / static_cast<float>(reps ? reps : 1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(inf.p);
  return NVDB_OK;
}

nvdb_status nvdb_hip_debug_clock(nvdb_hip_ctx* c, int variant, uint32_t nq, float seconds, float* out4) {
  if (!c || !out4) return NVDB_ERR_INVALID;
  if (!c->rows || c->dtype != NVDB_DTYPE_F16 || c->dim != 768 || !c->q16.p) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: run a path-2 search on an fp16 d=768 corpus first");
  if (nq <= 128 || nq > (c->last_nq + 255u) / 256u * 256u) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: 128 < nq <= the last search's padded batch");
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t QT = (nq + 255) / 256, nq_pad = QT * 256;
  const uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0 || (nwg & 7u) || ((nwg >> 3) % QT)) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: batch does not map onto the XCD-aware grid");
  DevBuf inf;
  nvdb_status st = ensure(c, inf, nq_pad * 4);
  if (st) return st;
  fill_u32_kernel<<<(nq_pad + 255) / 256, 256, 0, c->stream>>>(static_cast<uint32_t*>(inf.p), 0x7F800000u, nq_pad);
  const size_t prog_bytes = static_cast<size_t>(nwg) * 8 * 4, stamp_bytes = static_cast<size_t>(nwg) * 16;
  if ((st = ensure(c, c->prog, prog_bytes + stamp_bytes))) return st;
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 4 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  constexpr size_t lds = static_cast<size_t>(FILTER_STAGES) * FILTER_ROWS * 768 * 2;
  const uint32_t n_al = static_cast<uint32_t>(c->n / FILTER_ROWS * FILTER_ROWS);
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  const auto t_start = std::chrono::steady_clock::now();
  float ms = 0.f;
  const uint32_t burst = 8;
#define NVDB_CLK_LAUNCH(V)                                                                                                       \
  {                                                                                                                              \
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(filter_f16_m16_kernel<768, 6, true, true, V>),                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                           \
    for (uint32_t r = 0; r < burst; ++r) {                                                                                       \
      HIPCHK(c, hipMemsetAsync(c->prog.p, 0xFF, prog_bytes, c->stream));                                                         \
      filter_f16_m16_kernel<768, 6, true, true, V><<<nwg, 256, lds, c->stream>>>(                                                \
          static_cast<const _Float16*>(c->rows), 0, n_al, static_cast<const _Float16*>(c->q16.p), nq, QT,                        \
          static_cast<const float*>(inf.p), static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),        \
          static_cast<Hit*>(c->hitlog.p), scatter_args(c, c->last_cap), static_cast<uint32_t*>(c->prog.p),                       \
          static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));                                \
    }                                                                                                                            \
  }
  for (;;) {                                       // back-to-back launches until `seconds` have passed, the last 8 timed
    const bool last = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_start).count() >= seconds;
    if (last) HIPCHK(c, hipEventRecord(e0, c->stream));
    switch (variant) {
      case 0: NVDB_CLK_LAUNCH(0) break;
      case 1: NVDB_CLK_LAUNCH(1) break;
      case 5: NVDB_CLK_LAUNCH(5) break;
      case 15: NVDB_CLK_LAUNCH(15) break;
      case 16: NVDB_CLK_LAUNCH(16) break;
      case 17: NVDB_CLK_LAUNCH(17) break;
      case 20: {                                  // the 8-wave production build (two waves per SIMD, 32 queries each), stamped
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(filter_f16_m16_kernel<768, 4, true, true, 0, 2, 2, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 8 * FILTER_LOGCAP * sizeof(Hit)))) return st;
        for (uint32_t r = 0; r < burst; ++r) {
          HIPCHK(c, hipMemsetAsync(c->prog.p, 0xFF, prog_bytes, c->stream));
          filter_f16_m16_kernel<768, 4, true, true, 0, 2, 2, 8><<<nwg, 512, lds, c->stream>>>(
              static_cast<const _Float16*>(c->rows), 0, n_al, static_cast<const _Float16*>(c->q16.p), nq, QT,
              static_cast<const float*>(inf.p), static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p),
              static_cast<Hit*>(c->hitlog.p), scatter_args(c, c->last_cap), static_cast<uint32_t*>(c->prog.p),
              static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead));
        }
      } break;
      default: return fail(c, NVDB_ERR_INVALID, "debug: unknown variant");
    }
    HIPCHK(c, hipGetLastError());
    if (last) {
      HIPCHK(c, hipEventRecord(e1, c->stream));
      HIPCHK(c, hipEventSynchronize(e1));
      HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
      ms /= static_cast<float>(burst);
      break;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
#undef NVDB_CLK_LAUNCH
  std::vector<uint64_t> stamps(static_cast<size_t>(nwg) * 2);
  HIPCHK(c, hipMemcpy(stamps.data(), static_cast<const char*>(c->prog.p) + prog_bytes, stamp_bytes, hipMemcpyDeviceToHost));
  std::vector<float> ghz;
  for (uint32_t w = 0; w < nwg; ++w)
    if (stamps[2 * w + 1]) ghz.push_back(static_cast<float>(static_cast<double>(stamps[2 * w]) / static_cast<double>(stamps[2 * w + 1]) * 0.1));   // realtime ticks at 100 MHz
  std::sort(ghz.begin(), ghz.end());
  out4[0] = ms;
  out4[1] = ghz.empty() ? 0.f : ghz[ghz.size() / 2];
  out4[2] = ghz.empty() ? 0.f : ghz.front();
  out4[3] = ghz.empty() ? 0.f : ghz.back();
  // how long the workgroups' tile loops ran (100 MHz ticks -> us): mean and max -- the launch ends with the slowest
  double sum_us = 0.0, max_us = 0.0;
  for (uint32_t w = 0; w < nwg; ++w) { const double us = static_cast<double>(stamps[2 * w + 1]) * 0.01; sum_us += us; max_us = std::max(max_us, us); }
  out4[4] = static_cast<float>(sum_us / nwg);
  out4[5] = static_cast<float>(max_us);
  // per XCD label (blockIdx % 8): mean duration, and the spread inside the label (max - min)
  for (uint32_t x = 0; x < 8; ++x) {
    double sx = 0.0, mn = 1e30, mxv = 0.0; uint32_t cnt = 0;
    for (uint32_t w = x; w < nwg; w += 8) { const double us = static_cast<double>(stamps[2 * w + 1]) * 0.01; sx += us; mn = std::min(mn, us); mxv = std::max(mxv, us); ++cnt; }
    out4[6 + 2 * x] = cnt ? static_cast<float>(sx / cnt) : 0.f;
    out4[7 + 2 * x] = cnt ? static_cast<float>(mxv - mn) : 0.f;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(inf.p);
  return NVDB_OK;
}


nvdb_status nvdb_hip_debug_clock_i8(nvdb_hip_ctx* c, int variant, uint32_t nq, float seconds, float* out4) {
  if (!c || !out4) return NVDB_ERR_INVALID;
  if (!c->rows || c->dtype != NVDB_DTYPE_I8 || c->dim != 768 || !c->q16.p || !c->opt_i8_wide) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: run a path-2 search on an int8 d=768 corpus first");
  if (nq <= 128 || nq > (c->last_nq + 255u) / 256u * 256u) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: 128 < nq <= the last search's padded batch");
  HIPCHK(c, hipSetDevice(c->device));
  const uint32_t QT = (nq + 255) / 256, nq_pad = QT * 256;
  const uint32_t nwg = (static_cast<uint32_t>(c->num_cu) / QT) * QT;
  if (nwg == 0 || (nwg & 7u) || ((nwg >> 3) % QT)) return fail(c, NVDB_ERR_UNSUPPORTED, "debug: batch does not map onto the XCD-aware grid");
  // thresholds: the ones the last search ended with (realistic stage-1 / stage-2 rates for the production variant)
  const size_t prog_bytes = static_cast<size_t>(nwg) * 8 * 4, stamp_bytes = static_cast<size_t>(nwg) * 16;
  nvdb_status st;
  if ((st = ensure(c, c->prog, std::max(prog_bytes + stamp_bytes, static_cast<size_t>(PROG_SLOTS) * c->num_cu * 8 * 4)))) return st;
  if ((st = ensure(c, c->hitlog, static_cast<size_t>(nwg) * 8 * FILTER_LOGCAP * sizeof(Hit)))) return st;
  constexpr size_t lds = static_cast<size_t>(3) * (I8W_TILE_ROWS * 768 + 4 * 1024) + 4096;
  const uint64_t n_dbg = c->dbg_rows > 0 ? std::min<uint64_t>(c->n, static_cast<uint64_t>(c->dbg_rows)) : c->n;
  const uint32_t n_al = static_cast<uint32_t>(n_dbg / I8W_TILE_ROWS * I8W_TILE_ROWS);
  const signed char* qhi = static_cast<const signed char*>(c->q16.p);
  const signed char* qlo = qhi + static_cast<size_t>(nq_pad) * 768;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  const auto t_start = std::chrono::steady_clock::now();
  float ms = 0.f;
  const uint32_t burst = 8;
#define NVDB_CLK_I8(V)                                                                                                           \
  {                                                                                                                              \
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(filter_i8w_kernel<768, 2, 6, true, 2, true, V>),                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                           \
    for (uint32_t r = 0; r < burst; ++r) {                                                                                       \
      HIPCHK(c, hipMemsetAsync(c->prog.p, 0xFF, prog_bytes, c->stream));                                                         \
      filter_i8w_kernel<768, 2, 6, true, 2, true, V><<<nwg, 256, lds, c->stream>>>(                                              \
          filter_rows_i8(c), filter_scales_i8(c), 0, n_al, qhi, qlo, nq, QT, static_cast<const float*>(c->thr.p),                \
          static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<const float*>(c->qdelta.p),  \
          static_cast<Hit*>(c->hitlog.p), scatter_args(c, c->last_cap, I8W_TILE_ROWS), static_cast<uint32_t*>(c->prog.p),        \
          static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead), static_cast<uint32_t*>(c->misc.p) + 4);  \
    }                                                                                                                            \
  }
  for (;;) {
    const bool last = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_start).count() >= seconds;
    if (last) { HIPCHK(c, hipMemsetAsync(static_cast<uint32_t*>(c->misc.p) + 4, 0, 8, c->stream)); HIPCHK(c, hipEventRecord(e0, c->stream)); }
    switch (variant) {
      case 0: NVDB_CLK_I8(0) break;
      case 1: NVDB_CLK_I8(1) break;
      case 2: NVDB_CLK_I8(2) break;
      case 3: NVDB_CLK_I8(3) break;
#define NVDB_CLK_I8P(V, DF)                                                                                                      \
      {                                                                                                                          \
        constexpr size_t ldsp = static_cast<size_t>(3) * (I8W_TILE_ROWS * 768 + 4 * 256) + (DF ? 16 * 768 : 0);                  \
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(filter_i8p_kernel<768, true, true, 6, V, 4, DF>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsp))); \
        for (uint32_t r = 0; r < burst; ++r) {                                                                                   \
          HIPCHK(c, hipMemsetAsync(c->prog.p, 0xFF, prog_bytes, c->stream));                                                     \
          filter_i8p_kernel<768, true, true, 6, V, 4, DF><<<nwg, 256, ldsp, c->stream>>>(                                               \
              filter_rows_i8(c), filter_scales_i8(c), 0, n_al, qhi, qlo, nq, QT, static_cast<const float*>(c->thr.p),            \
              static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<const float*>(c->qdelta.p), \
              static_cast<Hit*>(c->hitlog.p), scatter_args(c, c->last_cap, I8W_TILE_ROWS), static_cast<uint32_t*>(c->prog.p),    \
              static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead), static_cast<uint32_t*>(c->misc.p) + 4); \
        }                                                                                                                        \
      }
      case 10: NVDB_CLK_I8P(0, true) break;             // the software-pipelined production build, stamped
      case 11: NVDB_CLK_I8P(1, true) break;             // ... its structure alone: no test, no rare path
      case 12: NVDB_CLK_I8P(2, true) break;             // ... test in the MFMA shadow, rare path never taken
      case 13: NVDB_CLK_I8P(3, true) break;             // ... rare path, deferred values never consumed
      case 14: NVDB_CLK_I8P(4, true) break;             // ... rare path entered and left at once
      case 15: NVDB_CLK_I8P(5, true) break;             // ... production loop, cycles inside rare_path / consume_slots (wave 0 of every workgroup)
#define NVDB_CLK_I8S(V) NVDB_CLK_I8SW(V, 4)
#define NVDB_CLK_I8SW(V, W)                                                                                                      \
      {                                                                                                                          \
        constexpr size_t ldss = static_cast<size_t>(3) * (I8W_TILE_ROWS * 768 + W * 256);                                        \
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(filter_i8s_kernel<768, true, true, 6, V, W>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldss))); \
        for (uint32_t r = 0; r < burst; ++r) {                                                                                   \
          HIPCHK(c, hipMemsetAsync(c->prog.p, 0xFF, prog_bytes, c->stream));                                                     \
          filter_i8s_kernel<768, true, true, 6, V, W><<<nwg, 64 * W, ldss, c->stream>>>(                                         \
              filter_rows_i8(c), filter_scales_i8(c), 0, n_al, qhi, qlo, nq, QT, static_cast<const float*>(c->thr.p),            \
              static_cast<const float*>(c->qscale.p), static_cast<const float*>(c->qinv.p), static_cast<const float*>(c->qdelta.p), \
              static_cast<Hit*>(c->hitlog.p), scatter_args(c, c->last_cap, I8W_TILE_ROWS), static_cast<uint32_t*>(c->prog.p),    \
              static_cast<uint32_t>(c->opt_sync_every - 1), static_cast<uint32_t>(c->opt_sync_lead), static_cast<uint32_t*>(c->misc.p) + 4); \
        }                                                                                                                        \
      }
      case 30: NVDB_CLK_I8S(0) break;                // the 16x16x64 build (kernels_filter_i8s.h), stamped
      case 31: NVDB_CLK_I8S(1) break;                // ... its structure alone: no test, nothing logged
      case 32: NVDB_CLK_I8S(2) break;                // ... test, nothing logged
      case 33: NVDB_CLK_I8S(3) break;                // ... structure alone without the in-loop LDS-DMA issue
      case 34: NVDB_CLK_I8S(4) break;                // ... structure alone without the A-fragment LDS reads
      case 35: NVDB_CLK_I8SW(0, 8) break;            // the 8-wave 16x16x64 build, stamped
      case 36: NVDB_CLK_I8SW(1, 8) break;            // ... its structure alone
      case 20: NVDB_CLK_I8P(0, false) break;         // the default build (first-stage survivors logged, finished after the stream), stamped
      case 22: NVDB_CLK_I8P(2, false) break;         // ... test, nothing logged
      case 24: NVDB_CLK_I8P(4, false) break;         // ... logging entered and left at once
      default: return fail(c, NVDB_ERR_INVALID, "debug: unknown variant");
    }
    HIPCHK(c, hipGetLastError());
    if (last) {
      HIPCHK(c, hipEventRecord(e1, c->stream));
      HIPCHK(c, hipEventSynchronize(e1));
      HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
      ms /= static_cast<float>(burst);
      break;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
#undef NVDB_CLK_I8
#undef NVDB_CLK_I8P
#undef NVDB_CLK_I8S
#undef NVDB_CLK_I8SW
  std::vector<uint64_t> stamps(static_cast<size_t>(nwg) * 2);
  HIPCHK(c, hipMemcpy(stamps.data(), static_cast<const char*>(c->prog.p) + prog_bytes, stamp_bytes, hipMemcpyDeviceToHost));
  std::vector<float> ghz;
  double rare_cyc = 0.0, cons_cyc = 0.0;
  for (uint32_t w = 0; w < nwg; ++w) {
    if (!stamps[2 * w + 1]) continue;
    if (variant == 15) { rare_cyc += static_cast<double>(stamps[2 * w] >> 32); cons_cyc += static_cast<double>(stamps[2 * w] & 0xFFFFFFFFull); }
    else ghz.push_back(static_cast<float>(static_cast<double>(stamps[2 * w]) / static_cast<double>(stamps[2 * w + 1]) * 0.1));
  }
  if (variant == 15) { ghz.assign(3, static_cast<float>(rare_cyc / nwg)); ghz[2] = static_cast<float>(cons_cyc / nwg); }   // out[2] / out[3]: mean cycles of a workgroup's wave 0 inside rare_path / consume_slots
  else std::sort(ghz.begin(), ghz.end());
  out4[0] = ms;
  out4[1] = ghz.empty() ? 0.f : ghz[ghz.size() / 2];
  out4[2] = ghz.empty() ? 0.f : ghz.front();
  out4[3] = ghz.empty() ? 0.f : ghz.back();
  uint32_t counts[2] = {0, 0};                      // rare-path entries / lo-plane MFMA blocks of the timed burst
  HIPCHK(c, hipMemcpy(counts, static_cast<uint32_t*>(c->misc.p) + 4, 8, hipMemcpyDeviceToHost));
  out4[4] = static_cast<float>(counts[0]) / burst;
  out4[5] = static_cast<float>(counts[1]) / burst;
  double sum_us = 0.0, max_us = 0.0;                // the tile loop alone, per workgroup (100 MHz ticks)
  uint32_t nstamped = 0;
  for (uint32_t w = 0; w < nwg; ++w)
    if (stamps[2 * w + 1]) { const double us = static_cast<double>(stamps[2 * w + 1]) * 0.01; sum_us += us; max_us = std::max(max_us, us); ++nstamped; }
  out4[6] = nstamped ? static_cast<float>(sum_us / nstamped) : 0.f;
  out4[7] = static_cast<float>(max_us);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return NVDB_OK;
}

// the device's own stream_tile_range for every stream of a launch, with the weights given (or the context's current ones)
__global__ void tile_ranges_kernel(uint32_t T, uint32_t S, const float* w, uint32_t* lo, uint32_t* hi) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S) stream_tile_range(T, S, s, true, w, lo[s], hi[s]);
}

nvdb_status nvdb_hip_debug_tile_ranges(nvdb_hip_ctx* c, uint32_t n_tiles, uint32_t n_streams, const float* weights8, uint32_t* out_lo, uint32_t* out_hi,
                                       float* weights_out8) {
  if (!c || !n_streams || !out_lo || !out_hi) return c ? fail(c, NVDB_ERR_INVALID, "bad argument") : NVDB_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf buf;
  nvdb_status st;
  if ((st = ensure(c, buf, static_cast<size_t>(n_streams) * 8 + 32))) return st;
  uint32_t* lo = static_cast<uint32_t*>(buf.p);
  uint32_t* hi = lo + n_streams;
  float* w = reinterpret_cast<float*>(hi + n_streams);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (weights8) HIPCHK(c, hipMemcpy(w, weights8, 32, hipMemcpyHostToDevice));
  else HIPCHK(c, hipMemcpy(w, c->xcdw.p, 32, hipMemcpyDeviceToDevice));
  tile_ranges_kernel<<<(n_streams + 255) / 256, 256, 0, c->stream>>>(n_tiles, n_streams, w, lo, hi);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out_lo, lo, static_cast<size_t>(n_streams) * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(out_hi, hi, static_cast<size_t>(n_streams) * 4, hipMemcpyDeviceToHost));
  if (weights_out8) HIPCHK(c, hipMemcpy(weights_out8, w, 32, hipMemcpyDeviceToHost));
  HIPCHK(c, hipFree(buf.p));
  return NVDB_OK;
}
#endif  // NVDB_HIP_DEV

nvdb_status nvdb_hip_get_stats(nvdb_hip_ctx* c, nvdb_hip_scan_stats* stats) {
  if (!c || !stats) return NVDB_ERR_INVALID;
  if (c->stats_lazy) {                             // the small-call path skips this read-back; do it now
    c->stats_lazy = false;
    unsigned long long tot = 0;
    if (c->last_nq) {
      HIPCHK(c, hipSetDevice(c->device));
      std::vector<uint32_t> cn(c->last_nq);
      HIPCHK(c, hipMemcpy(cn.data(), c->cnt.p, c->last_nq * 4, hipMemcpyDeviceToHost));
      for (uint32_t v : cn) tot += std::min(v, c->last_cap);
    }
    c->stats.candidates = tot;
  }
  *stats = c->stats;
  return NVDB_OK;
}

nvdb_status nvdb_hip_merge_topk_strided_dev(nvdb_hip_ctx* c, const uint64_t* dev_ids, const float* dev_scores, size_t stride_ids_bytes,
                                            size_t stride_scores_bytes, uint32_t nshards, uint32_t nq, uint32_t k, uint64_t* dev_out_ids,
                                            float* dev_out_scores, void* hip_stream) {
  if (!c) return NVDB_ERR_INVALID;
  if (!dev_ids || !dev_scores || !dev_out_ids || !dev_out_scores) return fail(c, NVDB_ERR_INVALID, "merge_topk: null pointer");
  if (nshards == 0 || nq == 0 || k == 0) return NVDB_OK;
  const uint64_t m64 = static_cast<uint64_t>(nshards) * k;
  if (m64 >= (1ull << 31) || nq > 65535) return fail(c, NVDB_ERR_UNSUPPORTED, "merge_topk: nshards*k must be < 2^31 and nq <= 65535");
  const uint32_t m = static_cast<uint32_t>(m64);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
  if (m <= 4096) {
    // every entry ranks itself against all others out of LDS (the lists need not be sorted)
    const size_t lds = ((m * 4 + 15) & ~15u) + static_cast<size_t>(m) * 8;
    merge_topk_kernel<<<nq, 256, lds, s>>>(reinterpret_cast<const unsigned long long*>(dev_ids), dev_scores, nshards, nq, k,
                                           reinterpret_cast<unsigned long long*>(dev_out_ids), dev_out_scores, stride_ids_bytes,
                                           stride_scores_bytes);
  } else {
    // longer lists: one binary search per other shard (the per-shard lists are sorted best-first, as every search path emits them)
    merge_topk_sorted_kernel<<<dim3((m + 255) / 256, nq), 256, 0, s>>>(reinterpret_cast<const unsigned long long*>(dev_ids), dev_scores, nshards, nq, k,
                                                                      reinterpret_cast<unsigned long long*>(dev_out_ids), dev_out_scores,
                                                                      stride_ids_bytes, stride_scores_bytes);
  }
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

nvdb_status nvdb_hip_merge_topk_dev(nvdb_hip_ctx* c, const uint64_t* dev_ids, const float* dev_scores, uint32_t nshards,
                                    uint32_t nq, uint32_t k, uint64_t* dev_out_ids, float* dev_out_scores, void* hip_stream) {
  return nvdb_hip_merge_topk_strided_dev(c, dev_ids, dev_scores, static_cast<size_t>(nq) * k * 8, static_cast<size_t>(nq) * k * 4, nshards, nq,
                                         k, dev_out_ids, dev_out_scores, hip_stream);
}

nvdb_status nvdb_merge_topk_host(const uint64_t* ids, const float* scores, uint32_t nshards, uint32_t nq, uint32_t k,
                                 uint64_t* out_ids, float* out_scores) {
  if (!ids || !scores || !out_ids || !out_scores) return NVDB_ERR_INVALID;
  std::vector<std::pair<float, uint64_t>> v(static_cast<size_t>(nshards) * k);
  for (uint32_t q = 0; q < nq; ++q) {
    for (uint32_t s = 0; s < nshards; ++s)
      for (uint32_t j = 0; j < k; ++j) {
        const size_t src = (static_cast<size_t>(s) * nq + q) * k + j;
        v[static_cast<size_t>(s) * k + j] = {scores[src], ids[src]};
      }
    std::stable_sort(v.begin(), v.end(), [](const std::pair<float, uint64_t>& a, const std::pair<float, uint64_t>& b) {
      return a.first > b.first || (a.first == b.first && a.second < b.second);
    });
    for (uint32_t j = 0; j < k; ++j) { out_scores[static_cast<size_t>(q) * k + j] = v[j].first; out_ids[static_cast<size_t>(q) * k + j] = v[j].second; }
  }
  return NVDB_OK;
}

// ---- refine -----------------------------------------------------------------------------------------
static nvdb_status refine_args(nvdb_hip_ctx* c, const void* q, const void* cand, uint32_t K, const void* out_ids) {
  if (!c) return NVDB_ERR_INVALID;
  if (!c->rows || c->n == 0) return fail(c, NVDB_ERR_NO_CORPUS, "Empty base");
  if (c->dtype != NVDB_DTYPE_F16 && c->dtype != NVDB_DTYPE_F32)
    return fail(c, NVDB_ERR_UNSUPPORTED, "refine supports base dtype fp16/fp32 only");
  if (K > NVDB_HIP_REFINE_KMAX) return fail(c, NVDB_ERR_INVALID, "refine: K not supported (K<=64)");
  if (!q || !cand || !out_ids) return fail(c, NVDB_ERR_INVALID, "refine: null pointer");
  return NVDB_OK;
}

static nvdb_status launch_refine(nvdb_hip_ctx* c, hipStream_t s, const float* dq, const uint32_t* dc, uint32_t Q, uint32_t R,
                                 uint32_t K, uint32_t* doi, float* dod) {
  const bool al = aligned_rows(c->dtype, c->dim);
  // v3 (whole rows per request, four lanes per row): fp16 rows of 512 / 1024 / 1536 bytes
  if (c->opt_refine_v2 >= 2 && c->dtype == NVDB_DTYPE_F16 && refine3_dim(c->dim)) {
#define NVDB_REFINE3(D)                                                                                                        \
    {                                                                                                                          \
      constexpr size_t lds = static_cast<size_t>(REFINE3_WAVES) * refine3_slot_bytes<D>();                                      \
      const void* fn = reinterpret_cast<const void*>(refine_l2_rows_kernel<D>);                                                 \
      if (!c->lds_attr_set.count(fn)) {                                                                                        \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                 \
        c->lds_attr_set.insert(fn);                                                                                            \
      }                                                                                                                        \
      refine_l2_rows_kernel<D><<<Q, 64 * REFINE3_WAVES, lds, s>>>(c->rows, c->n, dq, dc, R, K, doi, dod);                       \
    }
    if (c->dim == 768) NVDB_REFINE3(768) else if (c->dim == 512) NVDB_REFINE3(512) else if (c->dim == 384) NVDB_REFINE3(384) else NVDB_REFINE3(256)
#undef NVDB_REFINE3
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  // v2 (coalesced gather through LDS): whole 16-byte steps only (f16: dim % 8 == 0, f32: dim % 4 == 0)
  if (al && c->opt_refine_v2 && static_cast<uint64_t>(c->dim) * bpe_of(c->dtype) >= 256) {
    constexpr size_t lds = 4 * 2 * 64 * 256;
    const void* fn = c->dtype == NVDB_DTYPE_F16 ? reinterpret_cast<const void*>(refine_l2_lds_kernel<DT_F16>)
                                                : reinterpret_cast<const void*>(refine_l2_lds_kernel<DT_F32>);
    if (!c->lds_attr_set.count(fn)) {
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      c->lds_attr_set.insert(fn);
    }
    if (c->dtype == NVDB_DTYPE_F16) refine_l2_lds_kernel<DT_F16><<<Q, 256, lds, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    else refine_l2_lds_kernel<DT_F32><<<Q, 256, lds, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  if (c->dtype == NVDB_DTYPE_F16) {
    if (al) refine_l2_kernel<DT_F16, true><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    else refine_l2_kernel<DT_F16, false><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
  } else {
    if (al) refine_l2_kernel<DT_F32, true><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    else refine_l2_kernel<DT_F32, false><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
  }
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

nvdb_status nvdb_hip_refine_l2_topk_dev(nvdb_hip_ctx* c, const float* dq, const uint32_t* dc, uint32_t Q, uint32_t R, uint32_t K,
                                        uint32_t* doi, float* dod, void* hip_stream) {
  if (c && (K == 0 || Q == 0 || R == 0)) return NVDB_OK;
  nvdb_status st = refine_args(c, dq, dc, K, doi);
  if (st) return st;
  HIPCHK(c, hipSetDevice(c->device));
  return launch_refine(c, hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream, dq, dc, Q, R, K, doi, dod);
}

nvdb_status nvdb_hip_refine_l2_topk(nvdb_hip_ctx* c, const float* queries, const uint32_t* cand_ids, uint32_t Q, uint32_t R,
                                    uint32_t K, uint32_t* out_ids, float* out_dist, nvdb_hip_timing* timing) {
  if (timing) std::memset(timing, 0, sizeof(*timing));
  if (c && (K == 0 || Q == 0 || R == 0)) return NVDB_OK;       // cuda_refine.cu:853-857
  nvdb_status st = refine_args(c, queries, cand_ids, K, out_ids);
  if (st) return st;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t qb = static_cast<size_t>(Q) * c->dim * 4, cb = static_cast<size_t>(Q) * R * 4, ob = static_cast<size_t>(Q) * K * 4;
  if ((st = ensure(c, c->rq, qb))) return st;
  if ((st = ensure(c, c->rcand, cb))) return st;
  if ((st = ensure(c, c->rout_ids, ob))) return st;
  if ((st = ensure(c, c->rout_dist, ob))) return st;
  hipEvent_t e0 = get_event(c, 56), e1 = get_event(c, 57), e2 = get_event(c, 58), e3 = get_event(c, 59);
  // optional pinned staging (reference: CUDA_PINNED, src/cuda_refine.cu:875, 902-914): inputs are packed into pinned host
  // buffers BEFORE the timed region, the asynchronous copies then run at the link's rate instead of through the runtime's
  // pageable bounce buffers; results come back into pinned memory and are copied out after the synchronisation.
  const void* h_q = queries;
  const void* h_c = cand_ids;
  void* h_oi = out_ids;
  void* h_od = out_dist;
  if (c->opt_refine_pinned) {
    const size_t need = qb + cb + 2 * ob;
    if (c->rpinned_bytes < need) {
      if (c->rpinned) (void)hipHostFree(c->rpinned);
      c->rpinned = nullptr; c->rpinned_bytes = 0;
      HIPCHK(c, hipHostMalloc(&c->rpinned, need, hipHostMallocDefault));
      c->rpinned_bytes = need;
    }
    char* pin = static_cast<char*>(c->rpinned);
    std::memcpy(pin, queries, qb);
    std::memcpy(pin + qb, cand_ids, cb);
    h_q = pin; h_c = pin + qb; h_oi = pin + qb + cb; h_od = pin + qb + cb + ob;
  }
  HIPCHK(c, hipEventRecord(e0, s));
  HIPCHK(c, hipMemcpyAsync(c->rq.p, h_q, qb, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->rcand.p, h_c, cb, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(e1, s));
  if ((st = launch_refine(c, s, static_cast<const float*>(c->rq.p), static_cast<const uint32_t*>(c->rcand.p), Q, R, K,
                          static_cast<uint32_t*>(c->rout_ids.p), out_dist ? static_cast<float*>(c->rout_dist.p) : nullptr)))
    return st;
  HIPCHK(c, hipEventRecord(e2, s));
  HIPCHK(c, hipMemcpyAsync(h_oi, c->rout_ids.p, ob, hipMemcpyDeviceToHost, s));
  if (out_dist) HIPCHK(c, hipMemcpyAsync(h_od, c->rout_dist.p, ob, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipEventRecord(e3, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (c->opt_refine_pinned) {
    std::memcpy(out_ids, h_oi, ob);
    if (out_dist) std::memcpy(out_dist, h_od, ob);
  }
  if (timing) {
    (void)hipEventElapsedTime(&timing->h2d_ms, e0, e1);
    (void)hipEventElapsedTime(&timing->kernel_ms, e1, e2);
    (void)hipEventElapsedTime(&timing->d2h_ms, e2, e3);
    timing->total_ms = timing->h2d_ms + timing->kernel_ms + timing->d2h_ms;
    const bool rows_kernel = c->opt_refine_v2 >= 2 && c->dtype == NVDB_DTYPE_F16 && refine3_dim(c->dim);
    timing->threads = rows_kernel ? 64 * REFINE3_WAVES : 256; timing->nwarps = rows_kernel ? REFINE3_WAVES : 4; timing->K = K; timing->R = R;
    timing->shmem_bytes = rows_kernel ? static_cast<size_t>(REFINE3_WAVES) * (c->dim == 768 ? refine3_slot_bytes<768>() : c->dim == 512 ? refine3_slot_bytes<512>() : c->dim == 384 ? refine3_slot_bytes<384>() : refine3_slot_bytes<256>())
                                      : (c->opt_refine_v2 ? 4 * 2 * 64 * 256 : 4 * 64 * 8 + 16);
  }
  return NVDB_OK;
}

// ---- host helpers -----------------------------------------------------------------------------------
#ifdef NVDB_HIP_DEV
uint32_t nvdb_permuted_tile(uint32_t g, uint32_t n_tiles) {
  uint32_t mul, mask;
  perm_params(n_tiles, mul, mask);
  return perm_tile_raw(g, mul, mask, n_tiles);
}
#endif

void nvdb_synth_rows_f32(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, float* out) {
  std::vector<int32_t> raw(dim);
  for (uint64_t r = 0; r < nrows; ++r) {
    const uint32_t key = synth_row_key(seed, row0 + r);
    uint64_t ss = 0;
    for (uint32_t c = 0; c < dim; ++c) { raw[c] = synth_raw(key, c); ss += static_cast<uint64_t>(static_cast<int64_t>(raw[c]) * raw[c]); }
    const double inv = synth_inv_norm(ss);
    for (uint32_t c = 0; c < dim; ++c) out[r * dim + c] = synth_elem(raw[c], inv);
  }
}

void nvdb_f32_to_f16(const float* src, uint16_t* dst, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) dst[i] = f32_to_f16_rne(src[i]);
}

void nvdb_quantize_i8_rows(const float* rows, uint64_t nrows, uint32_t dim, int8_t* out, float* scales) {
  for (uint64_t r = 0; r < nrows; ++r) scales[r] = quantize_i8_row(rows + r * dim, dim, out + r * dim);
}

}  // extern "C"
