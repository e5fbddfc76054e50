#!/usr/bin/env python3
"""64 < k <= 1024 on the filter path against the any-k path (path = 1), whole searches through the host API.
usage: wide_k_bench.py [rows:dim:dtype:k:batch,...]   Developer tool; GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
DT = {"f16": nvdb_amd.DT_F16, "f32": nvdb_amd.DT_F32, "i8": nvdb_amd.DT_I8}
CASES = sys.argv[1] if len(sys.argv) > 1 else "5000000:1536:f16:100:1024,5000000:1024:i8:100:1024,10000000:768:f16:100:1024,5000000:1536:f16:1000:256"
for case in CASES.split(","):
    n, d, tag, k, B = case.split(":"); n, d, k, B = int(n), int(d), int(k), int(B)
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(20240613, n, d, DT[tag])
    q = nvdb_amd.synth_rows_f32(20240614, 0, B, d)
    res = {}
    for path, reps in ((0, 3), (1, 1)):
        ctx.set_option("path", path)
        ctx.search_batch(q, k)
        t0 = time.perf_counter()
        for _ in range(reps): r = ctx.search_batch(q, k)
        el = (time.perf_counter() - t0) / reps
        st = ctx.stats()
        res[path] = r
        print(f"N={n} d={d} {tag} k={k} batch={B} path option {path}: ran path {st['path']}, {st['chunks']} chunks, {el * 1e3:.1f} ms per batch = {B / el:.0f} queries/s", flush=True)
    print("   same ids and score bits:", np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32)), flush=True)
    ctx.close()
