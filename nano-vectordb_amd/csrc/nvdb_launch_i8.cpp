// nvdb_launch_i8.cpp -- launch helpers of the int8 MFMA filter kernels (kernels_filter.h: two-stage / bootstrap builds;
// kernels_filter_i8s.h: the 16x16x64 logged build); query prep (two int8 planes).
#include "nvdb_ctx.h"
#include "kernels_filter_i8s.h"

namespace nvdbhip {

nvdb_status launch_prep_q8(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t nq_pad, const PrepInit& pinit) {
  // (filter_max_norm: row norms of what the int8 kernels stream -- the corpus itself, or the int8 filter shadow of an fp16 / fp32 corpus,
  //  whose quantisation residual resid_max joins the error bound)
  prep_q8_kernel<<<nq_pad, 256, 0, s>>>(dev_q, nq, c->dim, c->fdim, c->filter_max_norm, c->q8shadow ? c->resid_max : 0.f, static_cast<signed char*>(c->q16.p),
                                        static_cast<signed char*>(c->q16.p) + static_cast<size_t>(nq_pad) * c->fdim,
                                        static_cast<float*>(c->qscale.p), static_cast<float*>(c->qinv.p),
                                        static_cast<float*>(c->ebound.p), static_cast<float*>(c->slack.p), static_cast<float*>(c->qdelta.p),
                                        static_cast<uint32_t*>(c->overflow.p), static_cast<uint32_t>(c->opt_i8_lo_bits), pinit);
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

// the boot build is the 128-queries-per-workgroup two-plane kernel; QT counts ITS query tiles (the caller multiplies by nb)
nvdb_status launch_boot_i8(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap) {
  if (c->fdim == 768) return launch_boot_i8_dim<768>(c, s, n0, nq, QT, cap);
  if (c->fdim == 512) return launch_boot_i8_dim<512>(c, s, n0, nq, QT, cap);
  if (c->fdim == 640) return launch_boot_i8_dim<640>(c, s, n0, nq, QT, cap);
  if (c->fdim == 384) return launch_boot_i8_dim<384>(c, s, n0, nq, QT, cap);
  if (c->fdim == 256) return launch_boot_i8_dim<256>(c, s, n0, nq, QT, cap);
  return fail(c, NVDB_ERR_UNSUPPORTED, "int8 boot kernel: unsupported dim");
}

nvdb_status launch_filter_i8(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap) {
  const uint32_t nb = filter_nb(c, nq);
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

}  // namespace nvdbhip
