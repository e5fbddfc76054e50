// nvdb_launch_exact.cpp -- launch helpers of the exact fp32-order kernels (kernels_exact.h: VALU; kernels_exact_mfma.h: fp32 matrix
// cores), of select / rescore / merge, and the any-k path (kernels_largek.h).
#include "nvdb_ctx.h"
#include "kernels_exact_mfma.h"
#include "kernels_largek.h"

namespace nvdbhip {

// one launch that resets all per-search words (the exact and any-k paths; the filter path's prep launch does it itself)
nvdb_status launch_init_search(nvdb_hip_ctx* c, hipStream_t s, uint32_t nq_pad, uint32_t prog_words) {
  init_search_kernel<<<(nq_pad + 255) / 256, 256, 0, s>>>(static_cast<uint32_t*>(c->cnt.p), static_cast<uint32_t*>(c->overflow.p),
                                                          static_cast<float*>(c->thr.p), static_cast<uint32_t*>(c->misc.p), nq_pad,
                                                          static_cast<uint32_t*>(c->prog.p), prog_words);
  HIPCHK(c, hipGetLastError());
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
      uint32_t P = std::max<uint32_t>(1, (static_cast<uint32_t>(c->opt_exact_wgs) * static_cast<uint32_t>(c->num_cu)) / gy);                                  \
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
  // fp16 / int8 rows, full groups of 64 queries: the tile converted once per workgroup into an fp32 LDS image (exact_mfma_img_kernel)
#define NVDB_SCAN_IMG(D)                                                                                                           \
  if constexpr (exact_img_shape<DT, D>()) {                                                                                        \
    if (c->opt_exact_img && c->opt_exact_lds != 2 && nq >= 64 && c->dim == D) {                                                    \
      const uint32_t gy = nq / 64;                                                                                                 \
      const uint32_t pmax = (cap > reserve + k) ? (cap - reserve) / k : 1;                                                         \
      uint32_t P = std::max<uint32_t>(1, (static_cast<uint32_t>(c->opt_exact_wgs) * static_cast<uint32_t>(c->num_cu)) / gy);                                  \
      P = std::max<uint32_t>(1, std::min(std::min(P, std::max<uint32_t>(1, tiles / 8)), pmax));                                    \
      const void* fn = reinterpret_cast<const void*>(exact_mfma_img_kernel<DT, D, false>);                                         \
      if (!c->lds_attr_set.count(fn)) {                                                                                            \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, exact_img_bytes<DT, D>()));                  \
        c->lds_attr_set.insert(fn);                                                                                                \
      }                                                                                                                            \
      exact_mfma_img_kernel<DT, D, false><<<dim3(P, gy), 256, exact_img_bytes<DT, D>(), s>>>(c->rows, c->scales, row_lo, row_hi, q32, gy * 64, k, thr, \
                                                                                             cand, cnt, cap, ovf, nullptr, 0);     \
      HIPCHK(c, hipGetLastError());                                                                                                \
      q0 = gy * 64; lds_done = true;                                                                                               \
    }                                                                                                                              \
  }
  NVDB_SCAN_IMG(768) NVDB_SCAN_IMG(640) NVDB_SCAN_IMG(512) NVDB_SCAN_IMG(384) NVDB_SCAN_IMG(256) NVDB_SCAN_IMG(128)
#undef NVDB_SCAN_IMG
  if (!lds_done) { NVDB_SCAN_LDS(768) NVDB_SCAN_LDS(640) NVDB_SCAN_LDS(512) NVDB_SCAN_LDS(384) NVDB_SCAN_LDS(256) NVDB_SCAN_LDS(128) }
#undef NVDB_SCAN_LDS
  (void)lds_done;
  if (q0 >= nq) return NVDB_OK;
  // the rest (fewer than 64 queries, or everything when the LDS-staged build does not take the shape): register-direct kernel
  const uint32_t nr = nq - q0;
  const uint32_t gy = (nr + 63) / 64;
  const uint32_t last_blocks = (nr - (gy - 1) * 64 + 15) / 16;                    // 16-query blocks of the last (partial) group
  const uint32_t nslice_max = last_blocks >= 3 ? 1u : (last_blocks == 2 ? 2u : 4u);
  const uint32_t pmax = (cap > reserve + k * nslice_max) ? (cap - reserve) / (k * nslice_max) : 1;
  uint32_t P = std::max<uint32_t>(1, (static_cast<uint32_t>(c->opt_exact_wgs) * static_cast<uint32_t>(c->num_cu)) / gy);   // one workgroup per CU in all (see opt_exact_wgs)
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

// fs.mode != 0: the caller wants the final select folded into the rescore launch; *fused tells whether this launch did it
// (rescore_lds_kernel only: the other rescore kernels leave the select to the caller)
nvdb_status launch_rescore(nvdb_hip_ctx* c, hipStream_t s, const float* q32, uint32_t nq, uint32_t cap, FinalSelect fs, bool* fused) {
  Cand* cand = static_cast<Cand*>(c->cand.p);
  uint32_t* cnt = static_cast<uint32_t*>(c->cnt.p);
  if (fused) *fused = false;
  const float* eb = static_cast<const float*>(c->ebound.p);
  uint32_t* viol = static_cast<uint32_t*>(c->misc.p);
  unsigned long long* tot = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->misc.p) + 8);
  const size_t rs_lds = static_cast<size_t>((c->dim + 3u) & ~3u) * 4;
  const size_t row_bytes = static_cast<size_t>(c->dim) * bpe_of(c->dtype);
  uint32_t cpp = 32;                               // candidates per pass of rescore_lds_kernel: rows + query within 60 KB of LDS
  while (cpp > 1 && rs_lds + cpp * (row_bytes + 16) > 60 * 1024) cpp >>= 1;
  if (c->opt_rescore8 >= 2 && (row_bytes & 15) == 0 && cpp >= 4) {
    size_t lds = rs_lds + cpp * (row_bytes + 16);
    if (fs.mode != 0) {
      uint32_t cap2 = 1;
      while (cap2 < cap) cap2 <<= 1;
      lds = std::max(lds, cap2 * sizeof(Cand));                    // the folded select sorts the list in the same LDS
    }
    const void* fn = c->dtype == NVDB_DTYPE_F32 ? reinterpret_cast<const void*>(rescore_lds_kernel<DT_F32>)
                   : c->dtype == NVDB_DTYPE_F16 ? reinterpret_cast<const void*>(rescore_lds_kernel<DT_F16>) : reinterpret_cast<const void*>(rescore_lds_kernel<DT_I8>);
    if (lds > 48 * 1024 && !c->lds_attr_set.count(fn)) {
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(SELECT_MAX_CAP * sizeof(Cand))));
      c->lds_attr_set.insert(fn);
    }
    if (c->dtype == NVDB_DTYPE_F32) rescore_lds_kernel<DT_F32><<<nq, 256, lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot, cpp, fs);
    else if (c->dtype == NVDB_DTYPE_F16) rescore_lds_kernel<DT_F16><<<nq, 256, lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot, cpp, fs);
    else rescore_lds_kernel<DT_I8><<<nq, 256, lds, s>>>(c->rows, c->scales, c->dim, q32, cand, cnt, cap, eb, viol, tot, cpp, fs);
    HIPCHK(c, hipGetLastError());
    if (fused) *fused = fs.mode != 0;
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
      uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu)) / gy);                                  \
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
#define NVDB_SC_IMG(DT, D)                                                                                                          \
  if constexpr (exact_img_shape<DT, D>()) {                                                                                        \
    if (c->opt_exact_img && c->opt_exact_lds != 2 && nq >= 64 && c->dim == D && q0 == 0) {                                         \
      const uint32_t gy = nq / 64;                                                                                                 \
      uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu)) / gy);                                  \
      P = std::max<uint32_t>(1, std::min(P, std::max<uint32_t>(1, tiles / 8)));                                                    \
      const void* fn = reinterpret_cast<const void*>(exact_mfma_img_kernel<DT, D, true>);                                          \
      if (!c->lds_attr_set.count(fn)) {                                                                                            \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, exact_img_bytes<DT, D>()));                  \
        c->lds_attr_set.insert(fn);                                                                                                \
      }                                                                                                                            \
      exact_mfma_img_kernel<DT, D, true><<<dim3(P, gy), 256, exact_img_bytes<DT, D>(), s>>>(c->rows, c->scales, 0u, n, q32, gy * 64, 0u, nullptr, \
                                                                                            nullptr, nullptr, 0u, nullptr, out, ld);   \
      HIPCHK(c, hipGetLastError());                                                                                                \
      q0 = gy * 64;                                                                                                                \
    }                                                                                                                              \
  }
#define NVDB_SC_LDS_DT(DT) NVDB_SC_IMG(DT, 768) NVDB_SC_IMG(DT, 640) NVDB_SC_IMG(DT, 512) NVDB_SC_IMG(DT, 384) NVDB_SC_IMG(DT, 256) NVDB_SC_IMG(DT, 128) \
  if (q0 == 0) { NVDB_SC_LDS(DT, 768) NVDB_SC_LDS(DT, 640) NVDB_SC_LDS(DT, 512) NVDB_SC_LDS(DT, 384) NVDB_SC_LDS(DT, 256) NVDB_SC_LDS(DT, 128) }
  if (c->dtype == NVDB_DTYPE_F32) { NVDB_SC_LDS_DT(DT_F32) } else if (c->dtype == NVDB_DTYPE_F16) { NVDB_SC_LDS_DT(DT_F16) } else { NVDB_SC_LDS_DT(DT_I8) }
#undef NVDB_SC_LDS_DT
#undef NVDB_SC_LDS
#undef NVDB_SC_IMG
  if (q0 >= nq) return NVDB_OK;
  const uint32_t nr = nq - q0;
  const uint32_t gy = (nr + 63) / 64;
  uint32_t P = std::max<uint32_t>(1, (2u * static_cast<uint32_t>(c->num_cu)) / gy);
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
                          uint32_t n_rows, Cand* seed_cand, uint32_t* seed_cnt, uint32_t seed_cap) {
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

}  // namespace nvdbhip

extern "C" {

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

}  // extern "C"
