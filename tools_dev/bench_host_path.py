#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-buffer entry point nvdb_hip_search_batch (queries and results in host memory)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
N, D, B, K = 10_000_000, 768, 1024, 10
ctx = nvdb_amd.HipContext(0)
ctx.generate_corpus(20240613, N, D, nvdb_amd.DT_F16)
q = nvdb_amd.synth_rows_f32(20240614, 0, 4 * B, D)
for b in (B, 64, 1):
    best = None
    for it in range(5):
        ids, sc, t = ctx.search_batch(q[it % 4 * B:it % 4 * B + b], K, want_timing=True)
        if best is None or t.total_ms < best[3]:
            best = (t.h2d_ms, t.kernel_ms, t.d2h_ms, t.total_ms)
    print(json.dumps({"batch": b, "h2d_ms": best[0], "kernel_ms": best[1], "d2h_ms": best[2], "total_ms": best[3],
                      "qps_pcie_inclusive": b / (best[3] * 1e-3), "qps_resident": b / (best[1] * 1e-3)}))
