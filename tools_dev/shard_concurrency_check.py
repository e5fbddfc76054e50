import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, torch, nvdb_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B, K, D, SEED, W = 1024, 10, 768, 20240613, 2
dev = torch.device("cuda", 0)
q = nvdb_amd.synth_rows_f32(SEED + 1, 0, B, D)
qd = torch.from_numpy(q).to(dev)
full = nvdb_amd.HipContext(0); full.generate_corpus(SEED, N, D, nvdb_amd.DT_F16, row_base=0)
fi, fs = full.search_batch(q, K); full.close()
ctxs = []
for r in range(W):
    lo, hi = N * r // W, N * (r + 1) // W
    c = nvdb_amd.HipContext(0); c.generate_corpus(SEED, hi - lo, D, nvdb_amd.DT_F16, row_base=lo); ctxs.append(c)
PACK = B * K * 12
def run(mode, sync):
    for c in ctxs: c.set_option("sibling_sync", sync)
    packed = [torch.empty(PACK, dtype=torch.uint8, device=dev) for _ in range(W)]
    streams = [torch.cuda.Stream() for _ in range(W)]
    def one(r):
        oi = packed[r][:B * K * 8].view(torch.int64).view(B, K); os_ = packed[r][B * K * 8:].view(torch.float32).view(B, K)
        for _ in range(3):
            ctxs[r].search_batch_dev(qd.data_ptr(), B, K, oi.data_ptr(), os_.data_ptr(), streams[r].cuda_stream)
        streams[r].synchronize()
    if mode == "seq":
        for r in range(W): one(r)
    else:
        th = [threading.Thread(target=one, args=(r,)) for r in range(W)]
        [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    for r in range(W): ctxs[r].search_check()
    g = torch.cat(packed)
    mi = torch.empty((B, K), dtype=torch.int64, device=dev); ms = torch.empty((B, K), dtype=torch.float32, device=dev)
    ctxs[0].merge_topk_strided_dev(g.data_ptr(), g.data_ptr() + B * K * 8, PACK, PACK, W, B, K, mi.data_ptr(), ms.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    bad = int((mi.cpu().numpy().astype(np.uint64) != fi).sum())
    print(f"mode={mode} sibling_sync={sync}: mismatching ids {bad} of {fi.size}", flush=True)
for mode, sync in (("seq", 1), ("par", 1), ("par", 1), ("par", 0), ("par", 0), ("seq", 1)):
    run(mode, sync)
