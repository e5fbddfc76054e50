set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout -k 10 400 python $R/tools_dev/small_fuse_ab.py 500000:384:f16,2900000:384:f16 > $O/r04_small_fuse_ab3.txt 2>&1; tail -12 $O/r04_small_fuse_ab3.txt
for mode in 1 0; do
  rocprofv3 --kernel-trace --output-format csv -d $O/r04_tl_$mode -- python3 $R/tools_dev/small_one.py 500000 384 f16 1 $mode > $O/r04_tl_$mode.log 2>&1
  python3 $R/tools_dev/trace_timeline.py $O/r04_tl_$mode > $O/r04_timeline_fuse$mode.txt 2>&1; cat $O/r04_timeline_fuse$mode.txt; rm -rf $O/r04_tl_$mode
done
timeout -k 10 300 python3 $R/tools_dev/upload_bench.py > $O/r04_upload_bench.txt 2>&1; cat $O/r04_upload_bench.txt
cd $R && timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact" > $O/r04_t4.log 2>&1; tail -8 $O/r04_t4.log
timeout -k 10 300 python tools_dev/exact_bench.py 4000000 768 > $O/r04_exact_bench2.txt 2>&1; grep "round 1" $O/r04_exact_bench2.txt | grep -v "nq=16"
