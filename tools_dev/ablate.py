#!/usr/bin/env python3
"""Time ablation builds of the filter kernel (developer tool; run on the GPU box)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = nvdb_amd.HipContext(0, dev=True)            # libnvdb_hip_dev.so: the product library has no debug entry points
ctx.generate_corpus(20240613, n, 768, nvdb_amd.DT_F16)
lib = ctx.lib
names = {0: "normal", 1: "no glds", 2: "no glds, no barrier", 3: "no MFMA", 4: "no epilogue", 5: "no LDS reads", 6: "ring 6", 7: "ring 8", 8: "ring 3", 9: "ring 12", 10: "L2-resident corpus (8 tiles/stream), ring 6"}
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(names)
for nq in (1024, 256):
    q = nvdb_amd.synth_rows_f32(1, 0, nq, 768)
    ctx.set_option("path", 2); ctx.search_batch(q, 10)
    for rnd in range(2):
        for v in variants:
            ms = C.c_float()
            st = lib.nvdb_hip_debug_filter_variant(ctx.h, v, nq, 3, C.byref(ms))
            assert st == 0, lib.nvdb_hip_last_error(ctx.h)
            tiles = (n // 32) * ((nq + 255) // 256) / 256.0     # tiles per workgroup
            print(f"nq={nq} round={rnd} var={v} {names[v]:22s} {ms.value:8.3f} ms  {ms.value*1e3/tiles:6.3f} us/tile  "
                  f"{2.0*nq*n*768/ms.value/1e9:8.1f} TFLOP/s  {n*1536/ms.value/1e6:7.1f} GB/s(alg)", flush=True)
