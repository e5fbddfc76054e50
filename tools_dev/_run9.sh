set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_gpu_full2.log 2>&1; tail -4 $O/r04_gpu_full2.log
timeout -k 10 300 python tools_dev/small_fuse_ab.py 500000:384:f16,1000000:384:f16,2900000:384:f16,2900000:384:i8,500000:384:f32,1250000:768:f16 > $O/r04_small_fuse_final.txt 2>&1; grep "^N=" $O/r04_small_fuse_final.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r04_tl_1 -- python3 $R/tools_dev/small_one.py 500000 384 f16 1 1 > $O/r04_tl_1.log 2>&1
python3 $R/tools_dev/trace_timeline.py $O/r04_tl_1 > $O/r04_timeline_final.txt 2>&1; cat $O/r04_timeline_final.txt; rm -rf $O/r04_tl_1
cd $R && bash tools_dev/profile_r04.sh r04f
