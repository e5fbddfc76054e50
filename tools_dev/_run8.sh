set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not full_size and not 100M and not config5 and not exact" > $O/r04_t8.log 2>&1; tail -3 $O/r04_t8.log
timeout -k 10 200 python tools_dev/small_fuse_ab.py 500000:384:f16,2900000:384:f16,1250000:768:f16 > $O/r04_small_fuse_final2.txt 2>&1; grep "^N=" $O/r04_small_fuse_final2.txt
cd /tmp && export TMPDIR=/tmp
for mode in 1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/r04_tl_$mode -- python3 $R/tools_dev/small_one.py 500000 384 f16 1 $mode > $O/r04_tl_$mode.log 2>&1
  python3 $R/tools_dev/trace_timeline.py $O/r04_tl_$mode > $O/r04_timeline_final.txt 2>&1; cat $O/r04_timeline_final.txt; rm -rf $O/r04_tl_$mode
done
cd $R && bash tools_dev/profile_r04.sh r04f
