#!/usr/bin/env python3
"""int8 batch 1024, N = 10M: pass time against the size of the threshold bootstrap (boot_tiles = 32-row tile maxima per
query) and the chunk growth.  Whole passes through nvdb_hip_search_batch_dev, interleaved rounds.  Developer tool; GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
dt = nvdb_amd.DT_I8 if (len(sys.argv) < 2 or sys.argv[1] == "i8") else nvdb_amd.DT_F16
n, d, B, K = (int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000), 768, (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 10
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ctx = nvdb_amd.HipContext(0)
ctx.generate_corpus(20240613, n, d, dt)
q = torch.from_numpy(nvdb_amd.synth_rows_f32(20240614, 0, 4 * B, d)).to(dev)
oi = torch.empty((B, K), dtype=torch.int64, device=dev); os_ = torch.empty((B, K), dtype=torch.float32, device=dev)
ref = None
growths = (5, 6, 8) if dt == nvdb_amd.DT_I8 else ((0, 4, 6, 12, 16) if B > 128 else (0, 16, 32, 64))
if len(sys.argv) > 4: growths = tuple(int(x) for x in sys.argv[4].split(','))
tile_list = [int(x) for x in sys.argv[5].split(',')] if len(sys.argv) > 5 else None
for rnd in range(2):
    for lb, tiles in ([(7, t_) for t_ in tile_list] if tile_list else ((7, 0), (7, 256), (7, 384), (7, 512), (7, 768), (7, 1024), (7, 1536)) if dt == nvdb_amd.DT_I8 else ((7, 0), (7, 256), (7, 1024), (7, 2048))):
        for growth in growths:
            if dt == nvdb_amd.DT_I8: ctx.set_option("i8_lo_bits", lb)
            ctx.set_option("boot_tiles", tiles)
            ctx.set_option("chunk_growth", growth)
            strm = torch.cuda.current_stream().cuda_stream
            for i in range(2): ctx.search_batch_dev(q[i * B:(i + 1) * B].data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(8): ctx.search_batch_dev(q[(i % 4) * B:(i % 4 + 1) * B].data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), strm)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 8
            st = ctx.search_check()
            got = (oi.cpu().numpy().copy(), os_.cpu().numpy().copy())
            if ref is None: ref = got
            same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
            print(f"round {rnd} lo_bits {lb} boot_tiles {tiles:5d} growth {growth}: {el * 1e3:.3f} ms per pass = {B / el:.0f} queries/s; chunks {st['chunks']} "
                  f"stage1 {st['i8_stage1_tiles']} stage2 blocks {st['i8_stage2_blocks']} candidates {st['candidates']}; same results: {same}", flush=True)
