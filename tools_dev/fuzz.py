#!/usr/bin/env python3
"""Randomised cross-check on the GPU box: for random (n, dim, dtype, nq, k, options) the automatic path must return exactly
what the exact path returns (ids and score bits); a few queries per case are also checked against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, nvdb_amd, pyoracle as po
from parity import assert_topk_equal
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rs = np.random.RandomState(seed)
orc = po.Oracle()
ctx = nvdb_amd.HipContext(0)
DT = {"f32": nvdb_amd.DT_F32, "f16": nvdb_amd.DT_F16, "i8": nvdb_amd.DT_I8}
t0 = time.time(); fails = 0
for ci in range(cases):
    tag = rs.choice(["f16", "f16", "i8", "i8", "f32"])
    dim = int(rs.choice([8, 24, 64, 100, 128, 200, 256, 384, 500, 512, 640, 768, 768, 768, 896, 1000, 1024, 1152, 1280, 1408, 1536, 1600, 2048, 2560, 3072, 3100]))
    if tag == "i8" and dim > 768 and rs.rand() < 0.5: dim = 768
    n = int(rs.choice([1, 7, 33, 500, 2047, 2048, 2049, 4096, 10000, 33333, 70001, 150000, 400000]))      # (>= 131K rows: the adaptive bootstrap size comes into play)
    nq = int(rs.choice([1, 2, 31, 32, 33, 64, 65, 128, 129, 200, 256, 257, 700, 1024, 1100]))
    k = int(rs.choice([1, 3, 10, 10, 10, 33, 64, 64, 65, 100, 300, 1024, 1500]))      # > 64: wide k on the filter / the any-k path
    if n * dim > 400000 * 384: n = 400000 * 384 // dim
    q8 = tag != "i8" and dim % 128 == 0 and 256 <= dim <= 1536 and rs.rand() < 0.35     # the int8 filter shadow (set before the corpus is loaded)
    ctx.set_option("q8_shadow", 1 if q8 else 0)
    ctx.generate_corpus(1000 + ci, n, dim, DT[tag])
    q = nvdb_amd.synth_rows_f32(5000 + ci, 0, nq, dim)
    if rs.rand() < 0.3: q[0] *= np.float32(10.0 ** rs.uniform(-6, 6))
    opts = {}
    if rs.rand() < 0.2: opts["tile_permute"] = 0
    if rs.rand() < 0.2: opts["chunk_growth"] = int(rs.choice([2, 3, 16]))
    if rs.rand() < 0.15: opts["mfma_boot"] = 0
    if rs.rand() < 0.15: opts["sibling_sync"] = 0
    if rs.rand() < 0.2: opts["xcd_balance"] = 0
    if rs.rand() < 0.2: opts["exact_mfma"] = 0
    if rs.rand() < 0.3: opts["exact_lds"] = int(rs.choice([0, 2]))
    if rs.rand() < 0.25: opts["exact_img"] = 0
    if rs.rand() < 0.2: opts["exact_wgs"] = int(rs.choice([2, 3]))
    if rs.rand() < 0.25: opts["fuse"] = 0
    if rs.rand() < 0.25: opts["zero_copy"] = 0
    if rs.rand() < 0.2 and tag != "i8": opts["boot_tiles"] = int(rs.choice([96, 256, 300]))
    if tag == "i8":
        if rs.rand() < 0.25: opts["i8_defer"] = 1
        if rs.rand() < 0.2: opts["i8_lo_bits"] = int(rs.choice([3, 5, 6]))
        if rs.rand() < 0.2: opts["boot_tiles"] = int(rs.choice([128, 512, 1024]))
    for k_, v in opts.items(): ctx.set_option(k_, v)
    try:
        ctx.set_option("path", 0); ai, asc = ctx.search_batch(q, k); st = ctx.stats()
        ctx.set_option("path", 1); ei, es = ctx.search_batch(q, k)
        ok = ai.shape == ei.shape and np.array_equal(ai, ei) and np.array_equal(asc.view(np.uint32), es.view(np.uint32))
        base, scales = nvdb_amd.synth_corpus(1000 + ci, 0, n, dim, DT[tag])
        for qi in sorted(set([0, nq // 2, nq - 1])):
            oi, os_ = orc.flat_topk(base, {"f32": po.DT_F32, "f16": po.DT_F16, "i8": po.DT_I8}[tag], q[qi:qi + 1], k, scales)
            ok = ok and np.array_equal(ei[qi], oi[0]) and np.array_equal(es[qi].view(np.uint32), os_[0].view(np.uint32))
    except Exception as e:
        ok = False; st = {"error": str(e)}
    for k_ in opts: ctx.set_option(k_, {"tile_permute": 1, "chunk_growth": 0, "mfma_boot": 1, "sibling_sync": 1, "xcd_balance": 1, "i8_defer": 0, "i8_lo_bits": 7, "boot_tiles": 0, "exact_mfma": 1, "exact_lds": 1, "exact_img": 1, "exact_wgs": 1, "fuse": 1, "zero_copy": 1}[k_])
    ctx.set_option("path", 0)
    if (ci + 1) % 50 == 0: print(f"  .. {ci + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)      # (a long run keeps talking)
    if not ok:
        fails += 1
        print(f"FAIL case {ci}: {tag} n={n} dim={dim} nq={nq} k={k} q8_shadow={q8} opts={opts} stats={st}", flush=True)
print(f"fuzz seed {seed}: {cases} cases, {fails} failures, {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if fails else 0)
