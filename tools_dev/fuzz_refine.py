#!/usr/bin/env python3
"""Randomised check of the refine kernels on the GPU box: random (n, dim, dtype, Q, R, K), candidate lists with invalid and
out-of-range ids and duplicates; ids and distance bits must equal the oracle's restatement of the reference kernel order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nvdb_amd, pyoracle as po
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rs = np.random.RandomState(seed)
orc = po.Oracle(); ctx = nvdb_amd.HipContext(0)
t0 = time.time(); fails = 0
for ci in range(cases):
    tag = rs.choice(["f16", "f16", "f32"])
    dim = int(rs.choice([8, 24, 100, 128, 200, 384, 512, 768, 768, 1000, 1024, 1536, 1536]))
    n = int(rs.choice([1, 50, 1000, 20000, 60000]))
    Q = int(rs.choice([1, 3, 17, 40])); R = int(rs.choice([1, 7, 64, 65, 300, 700, 1024])); K = int(rs.choice([1, 10, 10, 33, 64]))
    dt = nvdb_amd.DT_F16 if tag == "f16" else nvdb_amd.DT_F32
    ctx.generate_corpus(3000 + ci, n, dim, dt)
    base, _ = nvdb_amd.synth_corpus(3000 + ci, 0, n, dim, dt)
    q = nvdb_amd.synth_rows_f32(7000 + ci, 0, Q, dim)
    cand = rs.randint(0, n, size=(Q, R)).astype(np.uint32)
    cand[rs.rand(Q, R) < 0.02] = 0xFFFFFFFF
    cand[rs.rand(Q, R) < 0.02] = n + 5                        # out of range: skipped (cuda_refine.cu:437)
    for v2 in (2, 1, 0):
        ctx.set_option("refine_v2", v2)
        ids, dist = ctx.refine_l2_topk(q, cand, K)
        oi, od = orc.refine(base, po.DT_F16 if tag == "f16" else po.DT_F32, q, cand, K, mode=0)
        ok = np.array_equal(ids, oi) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))
        if not ok:
            fails += 1
            print(f"FAIL case {ci}: {tag} n={n} dim={dim} Q={Q} R={R} K={K} refine_v2={v2}", flush=True)
    ctx.set_option("refine_v2", 2)
print(f"refine fuzz seed {seed}: {cases} cases x 2 kernels, {fails} failures, {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
