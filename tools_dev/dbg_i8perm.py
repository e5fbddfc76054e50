import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, nvdb_amd
SEED = 20240613
n, d, nq, k = 120007, 768, 300, 10
ctx = nvdb_amd.HipContext(0); ctx.generate_corpus(SEED + 40, n, d, nvdb_amd.DT_I8)
queries = nvdb_amd.synth_rows_f32(SEED + 41, 0, nq, d)
ctx.set_option("path", 1); ei, es = ctx.search_batch(queries, k)
dev = torch.device("cuda", 0); st = torch.cuda.Stream()
with torch.cuda.stream(st):
    tq = torch.from_numpy(queries).to(dev); ti = torch.empty((nq, k), dtype=torch.int64, device=dev); ts = torch.empty((nq, k), dtype=torch.float32, device=dev)
    ctx.set_option("path", 2)
    ctx.search_batch_dev(tq.data_ptr(), nq, k, ti.data_ptr(), ts.data_ptr(), st.cuda_stream)
st.synchronize()
try:
    print(ctx.search_check())
except Exception as e:
    print("check:", e)
fi, fs = ti.cpu().numpy().astype(np.uint64), ts.cpu().numpy()
bad = np.argwhere((fi != ei).any(axis=1)).ravel()
print("first attempt mismatching queries:", bad.tolist())
for qx in bad[:3]:
    print(qx, "filter path:", fi[qx].tolist(), fs[qx].tolist()); print("   exact:", ei[qx].tolist(), es[qx].tolist())
