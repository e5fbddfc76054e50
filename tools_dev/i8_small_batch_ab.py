#!/usr/bin/env python3
"""int8 d = 768, batches <= 128: the 8-wave 16x16x64 logged build (i8_small8 = 1, the product's default since round 3) against
filter_i8w_kernel<768, 1> (i8_small8 = 0, developer library), interleaved, whole passes; ids and score bits compared.  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
strm = torch.cuda.current_stream().cuda_stream
n, d, K = 10_000_000, int(os.environ.get("AB_DIM", "768")), 10
ctx = nvdb_amd.HipContext(0, dev=True)
ctx.generate_corpus(20240613, n, d, nvdb_amd.DT_I8)
for B in (128, 100, 64, 32, 8, 1):
    q = torch.from_numpy(nvdb_amd.synth_rows_f32(20240614, 0, B, d)).to(dev)
    oi = torch.empty((B, K), dtype=torch.int64, device=dev); os_ = torch.empty((B, K), dtype=torch.float32, device=dev)
    ref = None
    for rnd in range(2):
        for w8 in (0, 1, 0, 1):
            ctx.set_option("i8_small8", w8)
            for _ in range(2): ctx.search_batch_dev(q.data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): ctx.search_batch_dev(q.data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
            st = ctx.search_check()
            got = (oi.cpu().numpy().copy(), os_.cpu().numpy().copy())
            if ref is None: ref = got
            same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
            print(f"B={B} i8_small8={w8}: {el*1e3:.3f} ms per pass = {n*(d+4)/el/1e12:.2f} TB/s; cand {st['candidates']} same {same}", flush=True)
