#!/usr/bin/env python3
"""int8 corpora with 768 < dim <= 1536 (round 3): the MFMA filter path (filter_i8w_kernel<DIM, 1, .., MB = 1>) against the exact
fp32-order kernel these dims took before; whole passes, ids and score bits compared.  Developer tool; GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
strm = torch.cuda.current_stream().cuda_stream
K = 10
DIMS = ((1536, 5_000_000), (1024, 5_000_000), (1280, 3_000_000), (1000, 3_000_000))
if len(sys.argv) > 1:                              # e.g. "384:10000000,512:10000000,256:10000000"
    DIMS = tuple((int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1].split(","))
BATCHES = tuple(int(x) for x in os.environ.get("I8_BATCHES", "1024,64").split(","))
for d, n in DIMS:
    ctx = nvdb_amd.HipContext(0)
    ctx.generate_corpus(20240613, n, d, nvdb_amd.DT_I8)
    for B in BATCHES:
        q = torch.from_numpy(nvdb_amd.synth_rows_f32(20240614, 0, B, d)).to(dev)
        oi = torch.empty((B, K), dtype=torch.int64, device=dev); os_ = torch.empty((B, K), dtype=torch.float32, device=dev)
        res = {}
        for path, reps in ((2, 5), (1, 1)):
            if path == 1 and B > 64: continue
            ctx.set_option("path", path)
            ctx.search_batch_dev(q.data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): ctx.search_batch_dev(q.data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
            st = ctx.search_check()
            res[path] = (oi.cpu().numpy().copy(), os_.cpu().numpy().copy())
            print(f"int8 d={d} n={n} batch={B} path={st['path']}: {el * 1e3:.3f} ms per pass = {B / el:.0f} queries/s = {2.0 * B * n * d / el / 1e12:.0f} TOP/s algorithmic, "
                  f"{n * (d + 4) / el / 1e9:.0f} GB/s; chunks {st['chunks']} candidates {st['candidates']}", flush=True)
        if 1 in res and 2 in res:
            print("   filter path == exact path:", np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1].view(np.uint32), res[2][1].view(np.uint32)), flush=True)
    ctx.close()
