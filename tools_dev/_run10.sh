R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "nano-vectordb_amd"); sys.path.insert(0, "oracle")
import pyoracle as po
n, d = 10_000_000, 768
hdr_done = False
block = np.random.RandomState(1).randint(0, 2 ** 15, size=(1 << 20, d), dtype=np.uint16)
import struct
with open("/dev/shm/up_test.vecbin", "wb") as f:
    f.write(struct.pack("<QIIIIQ", po.VEC_MAGIC, 1, po.DT_F16, d, 0, n) + b"\0" * 32)
    for i in range(0, n, len(block)): f.write(block[:min(len(block), n - i)].tobytes())
q = np.random.RandomState(2).randn(64, d).astype(np.float32)
po.write_raw12("/dev/shm/up_test_q.raw12", q)
PY
for t in default 4 6 8 12; do
  echo "== NVDB_UPLOAD_THREADS=$t"
  if [ $t = default ]; then NVDB_UPLOAD_DEBUG=1 nano-vectordb_amd/bin/nvdb_bench /dev/shm/up_test.vecbin /dev/shm/up_test_q.raw12 10 gpu 0 1 64 2>&1 | grep -E "nvdb upload|gpu_upload"; 
  else NVDB_UPLOAD_THREADS=$t NVDB_UPLOAD_DEBUG=1 nano-vectordb_amd/bin/nvdb_bench /dev/shm/up_test.vecbin /dev/shm/up_test_q.raw12 10 gpu 0 1 64 2>&1 | grep -E "nvdb upload|gpu_upload"; fi
done > $O/r04_upload_cli.txt 2>&1
rm -f /dev/shm/up_test.vecbin /dev/shm/up_test_q.raw12
cat $O/r04_upload_cli.txt | cut -c1-400
