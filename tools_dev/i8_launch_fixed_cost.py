#!/usr/bin/env python3
"""What a launch of the int8 pipelined kernel costs beyond its tiles: stamped production build (variant 10) and its bare
structure (11) over the first R rows of the corpus, R = 1 .. 256 tiles per row stream, at the thresholds the preceding
search ended with.  launch = event time per launch (includes a 3-us memset of the rendezvous words), loop = the tile loop
alone inside the kernel (mean / longest workgroup).  Developer tool (libnvdb_hip_dev.so); run on the GPU box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import nvdb_amd
n = 10_000_000
ctx = nvdb_amd.HipContext(0, dev=True)
ctx.generate_corpus(20240613, n, 768, nvdb_amd.DT_I8)
lib = ctx.lib
nq = 1024
q = nvdb_amd.synth_rows_f32(20240614, 0, nq, 768)
ctx.set_option("path", 2); ctx.search_batch(q, 10)
for tiles in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    rows = 64 * 64 * tiles                               # 64 row streams x 64-row tiles
    ctx.set_option("debug_rows", rows)
    for var in (10, 11):
        out = (C.c_float * 8)()
        rc = lib.nvdb_hip_debug_clock_i8(ctx.h, var, nq, 0.3, out)
        assert rc == 0, lib.nvdb_hip_last_error(ctx.h)
        print(f"tiles/stream {tiles:4d} rows {rows:8d} variant {var}: launch {out[0] * 1e3:8.1f} us, loop mean {out[6]:8.1f} us, longest {out[7]:8.1f} us, "
              f"per tile (loop mean) {out[6] / tiles:6.2f} us, outside the loop {out[0] * 1e3 - out[7]:6.1f} us; clock {out[1]:.3f} GHz; rare entries {out[4]:.0f}", flush=True)
