#!/usr/bin/env python3
"""A few single-query host-API searches on a small corpus, for a rocprofv3 --kernel-trace timeline (tools_dev/trace_timeline.py).
usage: small_one.py <rows> <dim> <f16|f32|i8> [batch] [fuse=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vectordb_amd"))
import nvdb_amd
n, d, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
FUSE = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # 0: separate launches + copies (options fuse = 0, zero_copy = 0)
ctx = nvdb_amd.HipContext(0)
ctx.set_option("fuse", FUSE); ctx.set_option("zero_copy", FUSE)
ctx.generate_corpus(7, n, d, {"f16": nvdb_amd.DT_F16, "f32": nvdb_amd.DT_F32, "i8": nvdb_amd.DT_I8}[tag])
q = nvdb_amd.synth_rows_f32(8, 0, 64 * B, d)
for i in range(12): ctx.search_batch(q[i * B:(i + 1) * B] if B > 1 else q[i], 10)
print(ctx.stats())
