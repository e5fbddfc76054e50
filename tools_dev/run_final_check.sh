set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04_gpu_full3.log 2>&1 && tail -4 $O/r04_gpu_full3.log &&
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1 && tail -2 $O/r04_smoke.log &&
timeout -k 10 400 python tools_dev/fuzz.py 77 250 > $O/r04_fuzz_seed77.txt 2>&1 && tail -2 $O/r04_fuzz_seed77.txt &&
bash tools_dev/profile_r04.sh r04f
