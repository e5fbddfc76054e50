#!/usr/bin/env python3
"""A/B of the fused launch chain (option fuse) and the zero-copy small calls (option zero_copy) on corpora of the reference's own
sizes: host-API latency of one search, identical ids and score bits required.  Developer tool; GPU box.
usage: small_fuse_ab.py [rows:dim:dtype,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
DT = {"f16": nvdb_amd.DT_F16, "f32": nvdb_amd.DT_F32, "i8": nvdb_amd.DT_I8}
BPE = {"f16": 2, "f32": 4, "i8": 1}
CASES = [(500_000, 384, "f16"), (1_000_000, 384, "f16"), (2_900_000, 384, "f16"), (2_900_000, 384, "i8"), (500_000, 384, "f32"), (2_900_000, 768, "f16"), (1_250_000, 768, "f16")]
if len(sys.argv) > 1: CASES = [(int(a.split(":")[0]), int(a.split(":")[1]), a.split(":")[2]) for a in sys.argv[1].split(",")]
MODES = [("separate launches", 0, 0), ("fused", 1, 0), ("fused + zero-copy", 1, 1)]
for n, d, tag in CASES:
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(7, n, d, DT[tag])
    q = nvdb_amd.synth_rows_f32(8, 0, 256, d)
    for B in (1, 4, 8, 64, 1024) if n * d <= 1_250_000 * 768 else (1, 4, 8, 64):
        qq = nvdb_amd.synth_rows_f32(8, 0, max(256, 4 * B), d) if B > 64 else q
        line, ref = [], None
        best = {}
        for rnd in range(2):                                      # interleaved rounds, best of two
            for name, fuse, zc in MODES:
                ctx.set_option("fuse", fuse); ctx.set_option("zero_copy", zc)
                for i in range(4): ctx.search_batch(qq[i * B:(i + 1) * B] if B > 1 else qq[i], 10)
                reps = 60 if B <= 64 else 12
                t0 = time.perf_counter()
                for i in range(reps):
                    j = i % (len(qq) // B)
                    res = ctx.search_batch(qq[j * B:(j + 1) * B] if B > 1 else qq[j], 10)
                el = (time.perf_counter() - t0) / reps
                if ref is None: ref = res if reps % (len(qq) // B) == 0 or True else res
                st = ctx.stats()
                best[name] = min(best.get(name, 1e9), el)
                # same final query batch in every mode (reps is the same): bit-identical results required
                if not (np.array_equal(ref[0], res[0]) and np.array_equal(ref[1].view(np.uint32), res[1].view(np.uint32))):
                    print(f"DIFFERENT RESULTS: N={n} d={d} {tag} batch={B} mode {name}", flush=True); sys.exit(1)
        stream_ms = n * d * BPE[tag] / 6.3e12 * 1e3
        print(f"N={n} d={d} {tag} batch={B}: " + "; ".join(f"{k}: {v * 1e3:.3f} ms" for k, v in best.items()) + f"  (path {st['path']}, {st['chunks']} chunks; streaming the rows once at 6.3 TB/s = {stream_ms:.3f} ms)", flush=True)
    ctx.close()
