#!/bin/bash
# Dev: whole-tree A/B of the working tree against an older commit on ONE GPU box (library AND host code: tools/ab_lib_prev.sh
# swaps only the .so).  Build the old tree here (no GPU needed), it travels with the snapshot (git-ignored directory):
#   bash tools/ab_tree_prev.sh prepare <commit>       # git archive -> _ab_prev/, build its libcfhip.so
#   gpurun -- bash tools/ab_tree_prev.sh run [pairs]   # alternating bench runs -> gpurun_out/ab_tree_prev.txt
set -e
if [ "$1" = "prepare" ]; then
  rm -rf _ab_prev && mkdir -p _ab_prev
  git archive "$2" | tar -x -C _ab_prev
  (cd _ab_prev && python -m centerfusiondetect3d_amd.build > /dev/null)
  echo "$2" > _ab_prev/.commit
  ls -la _ab_prev/centerfusiondetect3d_amd/libcfhip.so
else
  N=${2:-4}
  OUT=gpurun_out/ab_tree_prev.txt
  echo "working tree vs $(cat _ab_prev/.commit): bench.py --no-cpu-baseline --steps 40, alternating, one box" > $OUT
  for i in $(seq $N); do
    python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['step_ms_p50'])" >> $OUT
    (cd _ab_prev && python bench.py --no-cpu-baseline --steps 40 2>/dev/null) | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', d['ms_per_step'], d['step_ms_p50'])" >> $OUT
  done
  cat $OUT
fi
