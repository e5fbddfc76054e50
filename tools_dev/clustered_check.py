#!/usr/bin/env python3
"""Robustness check on clustered data (real embeddings cluster; the synthetic benchmark corpus does not): path-2 search on a
corpus of C cluster centres + noise, queries near centres.  Prints stats, time, and equality with the exact path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
N, D, C, B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 768, 500, 1024, 10
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ordered = len(sys.argv) > 3 and sys.argv[3] == "ordered"      # rows grouped by cluster (insertion order = topic order)
rs = np.random.RandomState(3)
cent = rs.randn(C, D).astype(np.float32); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
base = np.empty((N, D), dtype=np.float16)
for lo in range(0, N, 200_000):
    hi = min(N, lo + 200_000)
    cid = (np.arange(lo, hi) * C // N) if ordered else rs.randint(0, C, size=hi - lo)
    x = cent[cid] + noise * rs.randn(hi - lo, D).astype(np.float32) / np.sqrt(D)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    base[lo:hi] = x.astype(np.float16)
q = cent[rs.randint(0, C, size=B)] + noise * rs.randn(B, D).astype(np.float32) / np.sqrt(D)
q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
ctx = nvdb_amd.HipContext(0); ctx.upload_corpus(base.view(np.uint16), nvdb_amd.DT_F16)
for tag, dt in (("f16", None),):
    ctx.set_option("path", 2)
    ids, sc = ctx.search_batch(q, K)
    t0 = time.perf_counter()
    for _ in range(5): ids, sc = ctx.search_batch(q, K)
    t = (time.perf_counter() - t0) / 5
    st = ctx.stats()
    ctx.set_option("path", 1)
    ei, es = ctx.search_batch(q[:64], K)
    same = np.array_equal(ids[:64], ei) and np.array_equal(sc[:64].view(np.uint32), es.view(np.uint32))
    tagc = " ORDERED" if ordered else ""
    print(f"clustered{tagc} N={N} C={C} noise={noise}: {t*1e3:.2f} ms per {B}-query batch ({B/t:.0f} q/s); candidates/query {st['candidates']/B:.1f}, "
          f"overflow_queries {st['overflow_queries']}, chunks {st['chunks']}, path {st['path']}; first 64 == exact path: {same}", flush=True)
