R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact or any_k or fuzz" > $O/r04_t13.log 2>&1; tail -4 $O/r04_t13.log
timeout -k 10 300 python tools_dev/exact_bench.py 4000000 768 > $O/r04_exact_bench3.txt 2>&1; grep "round 1" $O/r04_exact_bench3.txt | cut -c1-150
