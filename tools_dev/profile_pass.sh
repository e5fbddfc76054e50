#!/bin/bash
# One profiling pass on the GPU box (run through gpurun): kernel trace + separate PMC passes of the default bench
# command, summaries under gpurun_out/<tag>_*.txt.   usage: tools_dev/profile_pass.sh <tag> [extra bench args]
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $B --steps 4 --warmup 1 > $O/${TAG}_kt.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_kt $O/${TAG}_kernel_trace_stats.txt > /dev/null
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $B --steps 2 --warmup 1 > $O/${TAG}_fetch.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_fetch $O/${TAG}_pmc_fetch_size.txt filter > /dev/null
echo "FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $B --steps 2 --warmup 1 > $O/${TAG}_write.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_write $O/${TAG}_pmc_write_size.txt filter > /dev/null
echo "WRITE_SIZE done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_sq -- python3 $B --steps 2 --warmup 1 > $O/${TAG}_sq.log 2>&1
python3 $R/tools_dev/summarize_prof.py $O/${TAG}_sq $O/${TAG}_pmc_sq.txt filter > /dev/null
echo "SQ done"
rm -rf $O/${TAG}_kt $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq
