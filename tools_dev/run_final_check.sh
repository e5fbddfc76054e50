# end-of-round check on the GPU box: the whole GPU suite, smoke(), the default bench line.  usage: run_final_check.sh [tag]
set -o pipefail
TAG=${1:-r04h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_suite.log 2>&1 && tail -4 $O/${TAG}_gpu_suite.log &&
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1 && tail -2 $O/${TAG}_smoke.log &&
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err && echo "bench default done"
