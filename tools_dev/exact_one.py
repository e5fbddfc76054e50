#!/usr/bin/env python3
"""One configuration of the exact path for a rocprofv3 pass:  exact_one.py <f16|i8|f32> <nq> <rows> <mode: img|lds|reg|valu> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
tag, nq, n, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ctx = nvdb_amd.HipContext(0)
ctx.generate_corpus(20240613, n, 768, {"f16": nvdb_amd.DT_F16, "i8": nvdb_amd.DT_I8, "f32": nvdb_amd.DT_F32}[tag])
ctx.set_option("path", 1)
ctx.set_option("exact_mfma", 0 if mode == "valu" else 1)
ctx.set_option("exact_img", 1 if mode == "img" else 0)
ctx.set_option("exact_lds", 2 if mode == "lds" else 1 if mode == "img" else 0)
q = nvdb_amd.synth_rows_f32(20240614, 0, nq, 768)
for _ in range(reps):
    ids, sc, t = ctx.search_batch(q, 10, want_timing=True)
print(f"{tag} nq={nq} n={n} mode={mode}: kernel_ms {t.kernel_ms:.3f} = {2.0 * nq * n * 768 / t.kernel_ms / 1e9:.1f} TFLOP/s")
