#!/usr/bin/env python3
"""In-kernel clock of the production fp16 filter kernel (MI355X_MICROARCH.md 'DVFS give-back' item 6):
diagnostic build stamped with s_memtime / s_memrealtime, launched back to back for a few seconds on the
synthetic (random) corpus.  Developer tool; run on the GPU box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import nvdb_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ctx = nvdb_amd.HipContext(0, dev=True)            # libnvdb_hip_dev.so
ctx.generate_corpus(20240613, n, 768, nvdb_amd.DT_F16)
lib = ctx.lib
for nq in [int(x) for x in os.environ.get("CLOCK_NQ", "1024").split(",")]:          # CLOCK_NQ=1024,512,256: the query tiles per stream 4 / 2 / 1
    q = nvdb_amd.synth_rows_f32(1, 0, nq, 768)
    ctx.set_option("path", 2)
    for bal, var, name in ((0, 20, "8-wave build, equal tile shares"), (1, 20, "8-wave build (default at d=768), XCD-balanced shares"), (1, 0, "4-wave build, XCD-balanced"),
                           (0, 20, "8-wave build, equal tile shares"), (1, 20, "8-wave build (default at d=768), XCD-balanced shares")):
        ctx.set_option("xcd_balance", bal)
        for _ in range(3):
            ctx.search_batch(q, 10)                 # the weights adapt after every big launch of a search
        out = (C.c_float * 22)()
        st = lib.nvdb_hip_debug_clock(ctx.h, var, nq, secs, out)
        assert st == 0, lib.nvdb_hip_last_error(ctx.h)
        ms, med, lo, hi = out[0], out[1], out[2], out[3]
        tf = 2.0 * nq * n * 768 / ms / 1e9
        mfma_cycles = (n / 32) * ((nq + 255) // 256) / 256.0 * 3072        # per SIMD: tiles per workgroup x 192 MFMAs x 16 cycles (4-wave: one wave, 8-wave: two waves x 96)
        busy = mfma_cycles / (ms * 1e-3 * med * 1e9)
        print(f"nq={nq} [{name}] whole-corpus launch {ms:.3f} ms = {tf:.0f} TFLOP/s; in-kernel clock median {med:.3f} GHz (min {lo:.3f}, max {hi:.3f}); "
              f"MFMA pipe busy {busy:.3f} of the cycles at that clock; peak at that clock {2.5e3 * med / 2.4:.0f} TFLOP/s -> {tf / (2.5e3 * med / 2.4):.3f}; "
              f"tile loop per workgroup: mean {out[4]:.0f} us, slowest {out[5]:.0f} us (+{(out[5] / max(out[4], 1e-9) - 1) * 100:.1f} %); per XCD label mean (spread inside): "
              + ", ".join(f"{out[6 + 2 * x]:.0f} ({out[7 + 2 * x]:.0f})" for x in range(8)), flush=True)
