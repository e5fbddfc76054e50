#!/usr/bin/env python3
"""Corpus upload rate of nvdb_hip_upload_corpus from pageable host memory (an np.memmap of a file in /dev/shm, like the tools'
mmap of a vecbin): the threaded pinned-staging path against the plain chunked hipMemcpy (NVDB_UPLOAD_THREADS=1).
usage: upload_bench.py [rows=6000000] [dim=768]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
path = f"/dev/shm/nvdb_upload_bench_{os.getpid()}.bin"
try:
    block = np.random.RandomState(1).randint(0, 2 ** 15, size=(1 << 20, d), dtype=np.uint16)      # finite halves
    with open(path, "wb") as f:
        for i in range(0, n, len(block)):
            f.write(block[:min(len(block), n - i)].tobytes())
    gb = n * d * 2 / 1e9
    for threads in ("1", "2", "4", "6", "8", "12", ""):
        if threads: os.environ["NVDB_UPLOAD_THREADS"] = threads
        else: os.environ.pop("NVDB_UPLOAD_THREADS", None)
        best = 1e9
        for rep in range(2):
            rows = np.memmap(path, dtype=np.uint16, mode="r", shape=(n, d))              # a fresh mapping: its pages are faulted in by the upload
            ctx = nvdb_amd.HipContext(0)
            t0 = time.perf_counter()
            ctx.upload_corpus(rows, nvdb_amd.DT_F16)
            el = time.perf_counter() - t0
            got, _ = ctx.download_rows(n - 3, 3)
            assert np.array_equal(got, np.asarray(rows[n - 3:])), "upload corrupted"
            ctx.close(); del rows
            best = min(best, el)
        print(f"upload {gb:.1f} GB, NVDB_UPLOAD_THREADS={threads or 'default'}: {best:.3f} s = {gb / best:.1f} GB/s (incl. max-norm pass over the rows)", flush=True)
finally:
    if os.path.exists(path): os.remove(path)
