set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not 100M and not config5 and not exact" > $O/r04_t7.log 2>&1; tail -4 $O/r04_t7.log
timeout -k 10 300 python tools_dev/small_fuse_ab.py 500000:384:f16,1000000:384:f16,2900000:384:f16,2900000:384:i8,500000:384:f32,1250000:768:f16 > $O/r04_small_fuse_final.txt 2>&1; grep "^N=" $O/r04_small_fuse_final.txt
for sp in 0 300; do echo "== sync_spin_us=$sp"; SMALL_OPTS=sync_spin_us=$sp timeout -k 10 200 python tools_dev/small_corpus_latency.py 500000:384:f16,2900000:384:f16 2>&1 | grep "^N="; done > $O/r04_sync_spin.txt; cat $O/r04_sync_spin.txt
