#!/bin/bash
# Profile the refine kernel at BASELINE config 5 (N=2.9M fp16, Q=10000, R=1024, K=10) on the GPU box: kernel trace, then
# FETCH_SIZE and the SQ counters each in a pass of their own.   usage: tools_dev/profile_refine.sh <tag> [ENV=VAL ...]
set -e -o pipefail
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="$R/tools_dev/bench_refine.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $P > $O/${TAG}_kt.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_kt $O/${TAG}_kernel_trace_stats.txt > /dev/null
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $P > $O/${TAG}_fetch.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_fetch $O/${TAG}_pmc_fetch_size.txt refine > /dev/null
echo "FETCH_SIZE done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_sq -- python3 $P > $O/${TAG}_sq.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_sq $O/${TAG}_pmc_sq.txt refine > /dev/null
echo "SQ done"
rm -rf $O/${TAG}_kt $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq
