#!/bin/bash
# Dev (GPU box): timing arms of the DCN kernel, each a rebuild with one -D flag (results of the arms are garbage by design).
#   bash tools/ab_dcn_arms.sh ["-Dflag ..." ...]   -> gpurun_out/dcn_arms.txt
set -e
# whatever happens under set -e: the in-tree library is the DEFAULT build again on exit
trap 'python -m centerfusiondetect3d_amd.build --force > /dev/null' EXIT
OUT=gpurun_out/dcn_arms.txt
: > $OUT
ARMS=("" "-DCF_DCN_NOBARRIER" "-DCF_DCN_NOBLEND" "-DCF_DCN_NOBLEND -DCF_DCN_NOBARRIER" "-DCF_DCN_NOLOAD -DCF_DCN_NOBLEND -DCF_DCN_NOBARRIER")
if [ $# -gt 0 ]; then ARMS=("$@"); fi
for arm in "${ARMS[@]}"; do
  echo "== arm: [$arm]" >> $OUT
  CF_EXTRA_FLAGS="${arm:+-DCF_DEV_ARMS }$arm" python -m centerfusiondetect3d_amd.build --force > /dev/null 2>&1
  python tools/bench_dcn.py 8,64,64,112,200 16,64,64,112,200 8,128,64,56,100 >> $OUT 2>&1
done
python -m centerfusiondetect3d_amd.build --force > /dev/null 2>&1
cat $OUT
