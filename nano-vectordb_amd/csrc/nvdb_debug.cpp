// nvdb_debug.cpp -- developer entry points (include/nvdb_hip_dev.h): stamped / ablation builds of the filter kernels, the device's
// tile partition function.  Compiled into libnvdb_hip_dev.so only; the product library does not contain this file's code.
#ifdef NVDB_HIP_DEV
#include "nvdb_ctx.h"
#include "kernels_filter_i8s.h"

namespace {
__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

extern "C" {

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
}  // extern "C"
#endif  // NVDB_HIP_DEV
