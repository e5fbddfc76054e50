import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import numpy as np, nvdb_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B, K, D, SEED = 1024, 10, 768, 20240613
q = nvdb_amd.synth_rows_f32(SEED + 1, 0, B, D)
full = nvdb_amd.HipContext(0); full.generate_corpus(SEED, N, D, nvdb_amd.DT_F16, row_base=0)
fi, fs = full.search_batch(q, K); print("full", full.stats()); full.close()
parts = []
W = 2
for r in range(W):
    lo, hi = N * r // W, N * (r + 1) // W
    c = nvdb_amd.HipContext(0); c.generate_corpus(SEED, hi - lo, D, nvdb_amd.DT_F16, row_base=lo)
    parts.append(c.search_batch(q, K)); print("shard", r, c.stats()); c.close()
ids = np.stack([p[0] for p in parts]); sc = np.stack([p[1] for p in parts])
mi, ms = nvdb_amd.merge_topk_host(ids, sc)
bad = np.argwhere(mi != fi)
print("mismatching entries:", len(bad), "score mismatches:", int((ms.view(np.uint32) != fs.view(np.uint32)).sum()))
for qi, j in bad[:10]:
    print(qi, j, "full", fi[qi, j], fs[qi, j], "merged", mi[qi, j], ms[qi, j])
