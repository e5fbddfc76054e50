import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dt = {"f16": nvdb_amd.DT_F16, "f32": nvdb_amd.DT_F32, "i8": nvdb_amd.DT_I8}[sys.argv[2] if len(sys.argv) > 2 else "f16"]
ctx = nvdb_amd.HipContext(0); ctx.generate_corpus(7, N, 768, dt)
q = nvdb_amd.synth_rows_f32(8, 0, 64, 768)
for i in range(5): ctx.search_batch(q[i], 10)
t0 = time.perf_counter()
for i in range(40): ctx.search_batch(q[i], 10)
print(f"N={N} host-API single query: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms", ctx.stats())
