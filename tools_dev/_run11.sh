R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 200 python tools_dev/exact_gy.py > $O/r04_exact_gy.txt 2>&1; cat $O/r04_exact_gy.txt
NVDB_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --rows 4000000 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/r04_share_gpu_rehearsal.json 2> $O/r04_share_gpu_rehearsal.err; echo "rc=$?"; tail -c 1500 $O/r04_share_gpu_rehearsal.json; tail -5 $O/r04_share_gpu_rehearsal.err
