#!/usr/bin/env python3
"""Exact path (fp32-image build) against the number of 64-query groups and the corpus size: where does a single group lose its 20 %?
usage: exact_gy.py  (developer tool; GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
d, K = 768, 10
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
strm = torch.cuda.current_stream().cuda_stream
for n in (4_000_000, 10_000_000):
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(20240613, n, d, nvdb_amd.DT_F16)
    ctx.set_option("path", 1)
    for nq, wgs, pre in [(a, b, p_) for a in (16, 64, 128, 256) for b in (1,) for p_ in (0, 1, 0, 1)]:
        ctx.set_option("exact_wgs", wgs); ctx.set_option("exact_prescan", pre)
        q = torch.from_numpy(nvdb_amd.synth_rows_f32(20240614, 0, nq, d)).to(dev)
        oi = torch.empty((nq, K), dtype=torch.int64, device=dev); os_ = torch.empty((nq, K), dtype=torch.float32, device=dev)
        for _ in range(2): ctx.search_batch_dev(q.data_ptr(), nq, K, oi.data_ptr(), os_.data_ptr(), strm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 3
        for _ in range(reps): ctx.search_batch_dev(q.data_ptr(), nq, K, oi.data_ptr(), os_.data_ptr(), strm)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
        ctx.search_check()
        print(f"f16 n={n} nq={nq} (groups of 64: {nq // 64}) exact_wgs={wgs} prescan={pre}: {el * 1e3:.3f} ms per pass = {2.0 * nq * n * d / el / 1e12:.1f} TFLOP/s", flush=True)
    ctx.close()
