R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 300 python tools_dev/exact_gy.py > $O/r04_exact_prescan.txt 2>&1; grep "^f16" $O/r04_exact_prescan.txt
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact or any_k or fuzz or non_finite or overflow or full_size_10M" > $O/r04_t14.log 2>&1; tail -4 $O/r04_t14.log
