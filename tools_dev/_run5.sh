set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_gpu_full.log 2>&1; tail -6 $O/r04_gpu_full.log
timeout -k 10 300 python tools_dev/small_fuse_ab.py 500000:384:f16,1000000:384:f16,2900000:384:f16,2900000:384:i8,1250000:768:f16 > $O/r04_small_fuse_final.txt 2>&1; grep "^N=" $O/r04_small_fuse_final.txt
for bt in 128 256 512; do echo "== boot_tiles=$bt"; SMALL_OPTS=boot_tiles=$bt timeout -k 10 200 python tools_dev/small_corpus_latency.py 500000:384:f16,2900000:384:f16 2>&1 | grep "^N="; done > $O/r04_boot_tiles_sweep.txt; cat $O/r04_boot_tiles_sweep.txt
