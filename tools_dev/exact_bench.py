#!/usr/bin/env python3
"""Exact fp32-order path (path 1 / the any-k score matrix): the fp32-MFMA kernels (kernels_exact_mfma.h, option exact_mfma = 1)
against the VALU kernels (exact_mfma = 0), interleaved rounds in one process; ids and score bits compared.  Developer tool; GPU box.
    exact_bench.py [rows=2000000] [dim=768]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
K = 10
MODES = ["valu", "mfma raw LDS stages", "mfma register-direct", "mfma fp32 LDS image"]
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
strm = torch.cuda.current_stream().cuda_stream
for tag, dt in (("f16", nvdb_amd.DT_F16), ("i8", nvdb_amd.DT_I8), ("f32", nvdb_amd.DT_F32)):
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(20240613, n, d, dt)
    ctx.set_option("path", 1)
    for nq in (16, 64, 256):
        q = torch.from_numpy(nvdb_amd.synth_rows_f32(20240614, 0, nq, d)).to(dev)
        oi = torch.empty((nq, K), dtype=torch.int64, device=dev); os_ = torch.empty((nq, K), dtype=torch.float32, device=dev)
        ref = None
        for rnd in range(2):
            for mf in (3, 1, 2, 0):
                if mf == 3 and tag == "f32": continue
                ctx.set_option("exact_mfma", 1 if mf else 0); ctx.set_option("exact_lds", 2 if mf == 1 else 0); ctx.set_option("exact_img", 1 if mf == 3 else 0)
                ctx.search_batch_dev(q.data_ptr(), nq, K, oi.data_ptr(), os_.data_ptr(), strm)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 3
                for _ in range(reps): ctx.search_batch_dev(q.data_ptr(), nq, K, oi.data_ptr(), os_.data_ptr(), strm)
                torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
                ctx.search_check()
                got = (oi.cpu().numpy().copy(), os_.cpu().numpy().copy())
                if ref is None: ref = got
                same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
                bpe = {"f16": 2, "i8": 1, "f32": 4}[tag]
                print(f"{tag} n={n} d={d} nq={nq} round {rnd} mode={MODES[mf]}: {el * 1e3:.3f} ms per pass = {2.0 * nq * n * d / el / 1e12:.1f} TFLOP/s, "
                      f"rows {n * d * bpe / el / 1e9:.0f} GB/s per query group pass; same results: {same}", flush=True)
    ctx.close()
