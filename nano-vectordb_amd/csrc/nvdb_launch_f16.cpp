// nvdb_launch_f16.cpp -- launch helpers of the fp16 MFMA filter kernels (kernels_filter.h): which build streams which shape, its LDS
// attribute, its rendezvous counters, its survivor logs; the bootstrap build; query prep.
#include "nvdb_ctx.h"

namespace nvdbhip {

nvdb_status launch_prep_q16(nvdb_hip_ctx* c, hipStream_t s, const float* dev_q, uint32_t nq, uint32_t nq_pad, const PrepInit& pinit) {
  // fp32 corpus: the shadow adds 2^-11 relative (normal halves) and <= 2^-25 absolute per element (subnormal halves)
  prep_q16_kernel<<<nq_pad, 256, 0, s>>>(dev_q, nq, c->dim, c->fdim, c->max_norm, c->dtype == NVDB_DTYPE_F32 ? FILTER_REL_F16 + 4.9e-4f : FILTER_REL_F16,
                                         c->dtype == NVDB_DTYPE_F32 ? 3.0e-8f * std::sqrt(static_cast<float>(c->dim)) : 0.f, static_cast<uint32_t*>(c->overflow.p),
                                         static_cast<_Float16*>(c->q16.p),
                                         static_cast<float*>(c->qscale.p), static_cast<float*>(c->qinv.p),
                                         static_cast<float*>(c->ebound.p), static_cast<float*>(c->slack.p), pinit);
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
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

nvdb_status launch_boot_f16(nvdb_hip_ctx* c, hipStream_t s, uint32_t n0, uint32_t nq, uint32_t QT, uint32_t cap, uint32_t nb) {
#define NVDB_BOOT_DIM(D) if (c->fdim == D) return nb == 1 ? launch_boot_dim<D, 1>(c, s, n0, nq, QT, cap) : launch_boot_dim<D, 2>(c, s, n0, nq, QT, cap)
  NVDB_BOOT_DIM(768); NVDB_BOOT_DIM(640); NVDB_BOOT_DIM(512); NVDB_BOOT_DIM(384); NVDB_BOOT_DIM(256); NVDB_BOOT_DIM(128);
#undef NVDB_BOOT_DIM
  return fail(c, NVDB_ERR_UNSUPPORTED, "boot kernel: unsupported dim");
}

nvdb_status launch_filter_f16(nvdb_hip_ctx* c, hipStream_t s, uint32_t row_lo, uint32_t row_hi, uint32_t nq, uint32_t QT, uint32_t cap) {
  const uint32_t nb = filter_nb(c, nq);
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

}  // namespace nvdbhip
