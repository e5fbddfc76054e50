/*
 * nvdb_hip_dev.h -- developer entry points of libnvdb_hip_dev.so (csrc/nvdb_debug.cpp; the library built with -DNVDB_HIP_DEV).
 *
 * NOT part of the drop-in surface: the product library libnvdb_hip.so exports none of these and contains none of the
 * timing-only kernel builds behind them (they return wrong results by design).  The dev library is a superset of the
 * product library (same C ABI, include/nvdb_hip.h); tools_dev/ and one CPU test load it explicitly.
 */
#ifndef NVDB_HIP_DEV_H
#define NVDB_HIP_DEV_H

#include "nvdb_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Developer aid, not part of the drop-in surface: time ablation builds of the fp16 d=768 filter kernel (thresholds
 * +inf, results discarded) on the resident corpus with the query workspace of the previous search (nq > 128).
 * Variants: 0 normal, 1 no direct-to-LDS loads, 2 = 1 + no barrier, 3 no MFMA, 5 no LDS reads, 6-9 fragment ring
 * 6/8/3/12, 10 L2-resident corpus. */
nvdb_status nvdb_hip_debug_filter_variant(nvdb_hip_ctx* ctx, int variant, uint32_t nq, uint32_t reps, float* ms_per_launch);

/* Developer aid: the clock the chip holds inside the production fp16 d=768 filter kernel.  Launches a diagnostic
 * build (identical code + one s_memtime / s_memrealtime stamp pair around the tile loop of every workgroup) back to
 * back for `seconds`, then reports out4 = { ms per launch (last 8), median, min, max over workgroups of
 * delta(s_memtime) / delta(s_memrealtime) x 100 MHz in GHz, mean and max over workgroups of the tile loop's duration in us, then per XCD label (blockIdx % 8) the mean duration and its spread } (22 floats).  variant: 0 = the production loop; timing-only
 * ablations 1 = no direct-to-LDS loads, 5 = no LDS reads, 15 = neither.  Same preconditions as the call above. */
nvdb_status nvdb_hip_debug_clock(nvdb_hip_ctx* ctx, int variant, uint32_t nq, float seconds, float* out4);

/* The tile range [lo, hi) the device's XCD-balanced partition (kernels_filter.h stream_tile_range) gives each of n_streams
 * row streams over n_tiles tiles, evaluated ON the device with weights8 (8 floats; NULL = the context's current, adapted
 * weights, which are also returned in weights_out8 if that is non-NULL).  tests/test_gpu_parity.py checks that the ranges tile
 * [0, n_tiles) exactly for any weights. */
nvdb_status nvdb_hip_debug_tile_ranges(nvdb_hip_ctx* ctx, uint32_t n_tiles, uint32_t n_streams, const float* weights8, uint32_t* out_lo,
                                       uint32_t* out_hi, float* weights_out8);

/* Developer aid (host only, no GPU): the physical tile the filter kernels stream for logical tile g of a corpus of n_tiles
 * tiles -- a bijection of [0, n_tiles) (identity below 64 tiles).  tests/test_cabi_cpu.py checks that property. */
uint32_t nvdb_permuted_tile(uint32_t g, uint32_t n_tiles);

/* The same for the int8 two-stage kernel (filter_i8w_kernel<768,2>, int8 d=768 corpus, nq > 128): stamped builds of the
 * production loop (variant 0) and of its timing-only ablations 1 = no stage 2 (the lo-plane pass never runs),
 * 2 = no stage-1 test either (stream + hi-plane MFMAs only), 3 = 2 + no per-tile barrier; 20 / 22 / 24 = the default build (filter_i8p_kernel, first-stage
 * survivors logged), its test without logging, logging entered and left at once; 10 = the software-pipelined build with the in-loop second stage
 * (filter_i8p_kernel), 11 = its structure alone, 12 = test without rare path.  out[0..3] as above, out[4], out[5] = rare-path
 * entries and lo-plane MFMA blocks per launch, out[6], out[7] = mean and longest tile-loop duration of a workgroup in us (out must hold 8 floats).
 * Developer option "debug_rows" (nvdb_hip_set_option, this build only): the launches cover rows [0, debug_rows) instead of the corpus. */
nvdb_status nvdb_hip_debug_clock_i8(nvdb_hip_ctx* ctx, int variant, uint32_t nq, float seconds, float* out4);

#ifdef __cplusplus
}
#endif
#endif /* NVDB_HIP_DEV_H */
