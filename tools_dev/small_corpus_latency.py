#!/usr/bin/env python3
"""Latency of one host-API search on corpora of the reference's own sizes (Performance.md: 500K .. 2.9M rows of 384-d
all-MiniLM-L6-v2 embeddings, k = 10, one query at a time or batches of 8): the automatic path against the exact path and the
MFMA filter path, so that the fixed per-search cost (init, query prep, bootstrap, chunk boundaries, selects, the host
round trip) shows next to the streaming time.  Developer tool; GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
DT = {"f16": nvdb_amd.DT_F16, "f32": nvdb_amd.DT_F32, "i8": nvdb_amd.DT_I8}
BPE = {"f16": 2, "f32": 4, "i8": 1}
CASES = [(500_000, 384, "f32"), (500_000, 384, "f16"), (1_000_000, 384, "f16"), (2_900_000, 384, "f16"), (2_900_000, 384, "i8"), (500_000, 768, "f32"), (2_900_000, 768, "f16")]
if len(sys.argv) > 1: CASES = [(int(a.split(":")[0]), int(a.split(":")[1]), a.split(":")[2]) for a in sys.argv[1].split(",")]
for n, d, tag in CASES:
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(7, n, d, DT[tag])
    for kv in os.environ.get("SMALL_OPTS", "").split(","):          # e.g. SMALL_OPTS=chunk_growth=32,boot_tiles=256
        if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    q = nvdb_amd.synth_rows_f32(8, 0, 256, d)
    for B in (1, 8, 64):
        line = []
        ref = None
        for path in ((0,) if os.environ.get("SMALL_OPTS") else (0, 1, 2)):
            ctx.set_option("path", path)
            for i in range(4): ctx.search_batch(q[i * B:(i + 1) * B] if B > 1 else q[i], 10)
            reps = 40
            t0 = time.perf_counter()
            for i in range(reps):
                j = i % (256 // B)
                res = ctx.search_batch(q[j * B:(j + 1) * B] if B > 1 else q[j], 10)
            el = (time.perf_counter() - t0) / reps
            st = ctx.stats()
            if ref is None: ref = res
            same = np.array_equal(ref[0], res[0]) and np.array_equal(ref[1].view(np.uint32), res[1].view(np.uint32))
            line.append(f"path {path}: {el * 1e3:.3f} ms (ran path {st['path']}, {st['chunks']} chunks{'' if same else ', DIFFERENT RESULTS'})")
        stream_ms = n * d * BPE[tag] / 6.3e12 * 1e3
        print(f"N={n} d={d} {tag} batch={B}: " + "; ".join(line) + f"; streaming the rows once at 6.3 TB/s = {stream_ms:.3f} ms", flush=True)
    ctx.close()
