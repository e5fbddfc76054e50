#!/bin/bash
# End-of-round evidence on the GPU box: rocprofv3 passes of the default bench command (fp16), of the int8 config and of the refine
# config, each counter set in a pass of its own; summaries under gpurun_out/<round>_*.txt.   usage: tools_dev/profile_round.sh r02
set -e -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools_dev/profile_pass.sh ${TAG}f
echo "fp16 passes done"
bash $R/tools_dev/profile_pass.sh ${TAG}f_i8 --dtype i8
echo "int8 passes done"
bash $R/tools_dev/profile_refine.sh ${TAG}f_refine REFINE_V2=2
echo "refine passes done"
