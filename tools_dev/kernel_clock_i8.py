#!/usr/bin/env python3
"""In-kernel clock and MFMA-busy fraction of the int8 two-stage kernel (filter_i8w_kernel<768,2>, batch 1024) and of its
timing-only ablations: stamped builds (s_memtime / s_memrealtime around the tile loop) launched back to back for a few
seconds on the synthetic int8 corpus, with the thresholds the preceding search ended with.  Developer tool
(libnvdb_hip_dev.so); run on the GPU box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import nvdb_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ctx = nvdb_amd.HipContext(0, dev=True)
ctx.generate_corpus(20240613, n, 768, nvdb_amd.DT_I8)
lib = ctx.lib
nq = 1024
q = nvdb_amd.synth_rows_f32(20240614, 0, nq, 768)
ctx.set_option("path", 2); ctx.search_batch(q, 10)
st = ctx.stats()
print(f"search: {st}", flush=True)
names = {33: "16x16x64 build, structure alone WITHOUT the in-loop LDS-DMA issue", 34: "16x16x64 build, structure alone WITHOUT the A-fragment LDS reads", 35: "16x16x64 build on 8 waves", 36: "16x16x64 build on 8 waves, structure alone", 30: "16x16x64 build (kernels_filter_i8s.h)", 31: "16x16x64 build, structure alone (no test, nothing logged)", 32: "16x16x64 build, test only", 20: "pipelined build, first-stage survivors logged and finished after the stream (the default)", 22: "default build, test only (nothing logged)", 24: "default build, logging entered and left at once", 10: "pipelined build with the in-loop second stage (i8_defer=1)", 11: "pipelined structure alone (no test, no rare path)", 12: "pipelined, test in the MFMA shadow, rare path never taken", 13: "pipelined, rare path without consuming the deferred values", 14: "pipelined, rare path entered and left at once", 0: "filter_i8w_kernel loop", 1: "no stage 2 (lo plane never multiplied)", 2: "no stage-1 test either (stream + hi-plane MFMAs)", 3: "the same without the per-tile barrier"}
for var in ([int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (20, 24, 22, 11, 10, 20, 24, 22)):
    out = (C.c_float * 8)()
    rc = lib.nvdb_hip_debug_clock_i8(ctx.h, var, nq, secs, out)
    assert rc == 0, lib.nvdb_hip_last_error(ctx.h)
    ms, med, lo, hi = out[0], out[1], out[2], out[3]
    tops = 2.0 * nq * n * 768 / ms / 1e9
    mfma_cycles = (n / 64) / 64.0 * 96 * 32            # per SIMD: tiles per workgroup x 96 v_mfma_i32_32x32x32_i8 x 32 cycles
    busy = mfma_cycles / (ms * 1e-3 * med * 1e9)
    print(f"[{names[var]}] whole-corpus launch {ms:.3f} ms = {tops:.0f} TOP/s algorithmic (hi plane only: half the int8 work of the two-plane kernel); "
          f"in-kernel clock median {med:.3f} GHz (min {lo:.3f}, max {hi:.3f}); MFMA pipe busy {busy:.3f}; cycles per tile {3072 / busy:.0f}; "
          f"rare-path entries {out[4]:.0f}, lo-plane MFMA blocks {out[5]:.0f} per launch ({n // 64 * 16 * 2} wave-halves)", flush=True)
# where a rare-path entry's time goes: cycles wave 0 of every workgroup spent inside rare_path / consume_slots (variant 15)
if len(sys.argv) > 3: sys.exit(0)
out = (C.c_float * 8)()
rc = lib.nvdb_hip_debug_clock_i8(ctx.h, 15, nq, secs, out)
assert rc == 0, lib.nvdb_hip_last_error(ctx.h)
per_wave = out[4] / 1024.0
print(f"[production loop, stamped inside the rare path] launch {out[0]:.3f} ms; rare-path entries {out[4]:.0f} per launch = {per_wave:.1f} per wave; "
      f"wave 0 of a workgroup spends {out[2]:.0f} cycles inside rare_path (= {out[2] / max(per_wave, 1e-9):.0f} per entry) and {out[3]:.0f} inside consume_slots "
      f"(= {out[3] / max(per_wave, 1e-9):.0f} per entry) of a {out[6]:.0f} us tile loop", flush=True)
