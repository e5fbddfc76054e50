#!/usr/bin/env python3
"""A/B of the int8 batch-1024 kernels in ONE process (interleaved rounds): filter_i8p_kernel on 8 waves of 32 queries
(i8_waves8 = 1, the default) and on 4 waves of 64 (i8_waves8 = 0), and filter_i8w_kernel (i8_pipe = 0); whole passes through
nvdb_hip_search_batch_dev.  Developer tool; run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
n, d, B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(os.environ.get("I8_AB_DIM", "768")), int(os.environ.get("I8_AB_BATCH", "1024")), 10
VARIANTS = ((1, 0, 0, 1), (1, 1, 0, 1), (1, 0, 0, 0), (1, 1, 0, 1), (1, 0, 0, 1)) if d == 768 else ((1, 0, 0, 1), (1, 0, 0, 0), (1, 0, 0, 1), (1, 0, 0, 0))
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = nvdb_amd.HipContext(0, dev=True)     # the developer library holds the variants
ctx.generate_corpus(20240613, n, d, nvdb_amd.DT_I8)
q = torch.from_numpy(nvdb_amd.synth_rows_f32(20240614, 0, 4 * B, d)).to(dev)
oi = torch.empty((B, K), dtype=torch.int64, device=dev); os_ = torch.empty((B, K), dtype=torch.float32, device=dev)
ref = None
for rnd in range(3):
    for pipe, w8, defer, m16 in VARIANTS:
        ctx.set_option("i8_pipe", pipe)
        ctx.set_option("i8_mfma16", m16)
        ctx.set_option("i8_waves8", w8)
        ctx.set_option("i8_defer", defer)
        strm = torch.cuda.current_stream().cuda_stream
        for i in range(2): ctx.search_batch_dev(q[i * B:(i + 1) * B].data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8): ctx.search_batch_dev(q[(i % 4) * B:(i % 4 + 1) * B].data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
        st = ctx.search_check()
        got = (oi.cpu().numpy().copy(), os_.cpu().numpy().copy())
        if ref is None: ref = got
        same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
        print(f"round {rnd} i8_pipe={pipe} i8_waves8={w8} i8_defer={defer} i8_mfma16={m16}: {el * 1e3:.3f} ms per pass = {B / el:.0f} queries/s = {2.0 * B * n * d / el / 1e12:.0f} TOP/s algorithmic; "
              f"stage1 {st['i8_stage1_tiles']} stage2 blocks {st['i8_stage2_blocks']} candidates {st['candidates']}; same results as the first run: {same}", flush=True)
