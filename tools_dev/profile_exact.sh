#!/bin/bash
# SQ counters + kernel trace of the exact fp32-MFMA kernels (one configuration each).  usage: tools_dev/profile_exact.sh <tag> ["<dtype> <queries> <rows> <mode>" ...]
set -e -o pipefail
TAG=$1; shift
if [ $# -eq 0 ]; then set -- "f16 256 4000000 img" "f16 64 10000000 img" "i8 256 4000000 img" "f32 256 4000000 lds"; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  name=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${name}_kt -- python3 $R/tools_dev/exact_one.py $cfg > $O/${TAG}_${name}_kt.log 2>&1
  python3 $R/tools_dev/summarize_prof.py $O/${TAG}_${name}_kt $O/${TAG}_${name}_kernel_trace.txt > /dev/null
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${TAG}_${name}_sq -- python3 $R/tools_dev/exact_one.py $cfg > $O/${TAG}_${name}_sq.log 2>&1
  python3 $R/tools_dev/summarize_prof.py $O/${TAG}_${name}_sq $O/${TAG}_${name}_pmc_sq.txt exact > /dev/null
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_${name}_grbm -- python3 $R/tools_dev/exact_one.py $cfg > $O/${TAG}_${name}_grbm.log 2>&1
  python3 $R/tools_dev/summarize_prof.py $O/${TAG}_${name}_grbm $O/${TAG}_${name}_pmc_grbm.txt exact > /dev/null
  rm -rf $O/${TAG}_${name}_kt $O/${TAG}_${name}_sq $O/${TAG}_${name}_grbm
  echo "$cfg done"
done
