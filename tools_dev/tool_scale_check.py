#!/usr/bin/env python3
"""Run the C++ tools on a 1M x 768 fp16 vecbin (written to /tmp): nvdb_bench gpu (batched + per query) vs omp sink,
nvdb_gt_build gpu vs omp.  Developer check for the GPU box."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, nvdb_amd, pyoracle as po
N, D, Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768, 2048
BIN = os.path.join(ROOT, "nano-vectordb_amd", "bin")
base, _ = nvdb_amd.synth_corpus(7, 0, N, D, nvdb_amd.DT_F16)
q = nvdb_amd.synth_rows_f32(8, 0, Q, D)
po.write_vecbin("/tmp/b16.vecbin", base, po.DT_F16); po.write_raw12("/tmp/q.raw12", q)
def run(*a, env=None):
    t0 = time.time(); e = dict(os.environ); e.update(env or {})
    out = subprocess.run([os.path.join(BIN, a[0]), *map(str, a[1:])], check=True, capture_output=True, text=True, env=e).stdout
    return out, time.time() - t0
g, tg = run("nvdb_bench", "/tmp/b16.vecbin", "/tmp/q.raw12", 10, "gpu", 0, 2, 1024)
print("gpu batched:", re.search(r"sink=(\S+)", g).group(1), [l for l in g.splitlines() if "Total" in l or "gpu_" in l], f"{tg:.1f}s")
po.write_raw12("/tmp/q64.raw12", q[:64])
g1, _ = run("nvdb_bench", "/tmp/b16.vecbin", "/tmp/q64.raw12", 10, "gpu", 0, 2, 1)
c1, tc = run("nvdb_bench", "/tmp/b16.vecbin", "/tmp/q64.raw12", 10, "omp", 16, 1, 1)
s_g, s_c = re.search(r"sink=(\S+)", g1).group(1), re.search(r"sink=(\S+)", c1).group(1)
print("64 queries one at a time: gpu sink", s_g, "omp sink", s_c, "EQUAL" if s_g == s_c else "DIFFERENT", f"(omp {tc:.1f}s)")
print([l for l in g1.splitlines() if l.startswith(("Avg_query", "p50", "p99"))])
run("nvdb_gt_build", "/tmp/b16.vecbin", "/tmp/q64.raw12", 10, "/tmp/gt_gpu.gtbin", env={"GT_MODE": "gpu"})
run("nvdb_gt_build", "/tmp/b16.vecbin", "/tmp/q64.raw12", 10, "/tmp/gt_omp.gtbin", env={"GT_MODE": "omp"})
print("gtbin gpu == omp:", open("/tmp/gt_gpu.gtbin", "rb").read() == open("/tmp/gt_omp.gtbin", "rb").read())
