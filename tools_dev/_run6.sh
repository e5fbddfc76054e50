set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
(for cfg in "f16 1 1000000" "f16 1 2900000" "f16 1 10000000" "f16 64 1250000" "f16 64 10000000" "f16 1024 1250000" "f16 1024 10000000" "i8 1 2900000" "i8 64 2900000" "i8 64 10000000" "i8 1024 1250000"; do
  echo "== $cfg"; timeout -k 10 120 python tools_dev/i8_boot_sweep.py $cfg 0 0,128,256,384 2>&1 | grep "round 1"; done) > $O/r04_boot_tiles_sweep2.txt
cat $O/r04_boot_tiles_sweep2.txt | cut -c1-120
