#!/bin/bash
# Round-4 evidence, ONE lease: the un-profiled bench line and the rocprofv3 passes of the SAME command on the
# SAME box, so that "dominant-kernel average x launches" can be set against that box's own ms_per_step.
#   usage (through gpurun): bash tools_dev/profile_r04.sh [tag]      outputs: gpurun_out/<tag>_*
# Every rocprofv3 command has the program directly after `--` (no env / bash -c hop) and --pmc passes carry no trace domains
# other than the kernel trace.
set -e -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$R/tools_dev/summarize_prof.py
PMC_SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"

# (0) the driver's command, un-profiled (extras + CPU baseline), then its --no-extras twin = the command the profiler runs
python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
echo "bench default done"
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${TAG}_bench_noextras.json 2> $O/${TAG}_bench_noextras.err
echo "bench --no-extras done"

pass() {   # pass <name> <filter-substring> <bench args...>: kernel trace (steps 20 / warmup 5) + FETCH / WRITE / SQ passes (steps 2 / warmup 1)
  local NAME=$1 FILT=$2; shift 2
  local B="$R/bench.py --no-cpu-baseline --no-extras $*"
  python3 $B --steps 20 --warmup 5 > $O/${TAG}_${NAME}_unprofiled.json 2> $O/${TAG}_${NAME}_unprofiled.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${NAME}_kt -- python3 $B --steps 20 --warmup 5 > $O/${TAG}_${NAME}_kt.json 2> $O/${TAG}_${NAME}_kt.err
  python3 $S $O/${TAG}_${NAME}_kt $O/${TAG}_${NAME}_kernel_trace_stats.txt > /dev/null
  echo "$NAME kernel trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${NAME}_fetch -- python3 $B --steps 2 --warmup 1 > $O/${TAG}_${NAME}_fetch.log 2>&1
  python3 $S $O/${TAG}_${NAME}_fetch $O/${TAG}_${NAME}_pmc_fetch_size.txt $FILT > /dev/null
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_${NAME}_write -- python3 $B --steps 2 --warmup 1 > $O/${TAG}_${NAME}_write.log 2>&1
  python3 $S $O/${TAG}_${NAME}_write $O/${TAG}_${NAME}_pmc_write_size.txt $FILT > /dev/null
  rocprofv3 --pmc $PMC_SQ --output-format csv -d $O/${TAG}_${NAME}_sq -- python3 $B --steps 2 --warmup 1 > $O/${TAG}_${NAME}_sq.log 2>&1
  python3 $S $O/${TAG}_${NAME}_sq $O/${TAG}_${NAME}_pmc_sq.txt $FILT > /dev/null
  echo "$NAME counters done"
  rm -rf $O/${TAG}_${NAME}_kt $O/${TAG}_${NAME}_fetch $O/${TAG}_${NAME}_write $O/${TAG}_${NAME}_sq
}

pass b1024 filter                      # (i)  fp16 N=10M batch 1024: the headline
pass b64 filter --batch 64             # (ii) fp16 N=10M batch 64: the HBM-bound point
pass i8 filter --dtype i8              # int8 N=10M batch 1024 (configs[2])

# (iii) N=100M batch 64 (the north star's own HBM point): kernel trace only
B100="$R/bench.py --no-cpu-baseline --no-extras --rows 100000000 --batch 64"
python3 $B100 --steps 6 --warmup 2 > $O/${TAG}_100M_b64_unprofiled.json 2> $O/${TAG}_100M_b64_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_100M_b64_kt -- python3 $B100 --steps 6 --warmup 2 > $O/${TAG}_100M_b64_kt.json 2> $O/${TAG}_100M_b64_kt.err
python3 $S $O/${TAG}_100M_b64_kt $O/${TAG}_100M_b64_kernel_trace_stats.txt > /dev/null
rm -rf $O/${TAG}_100M_b64_kt
echo "100M batch 64 kernel trace done"
