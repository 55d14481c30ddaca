#!/bin/bash
# Dev (GPU box): several 16-channel slices per round (SPR) in the 3x3 patch kernel, forced through CF_CONV3_CFG
# ("wc,wp,wk,t2,ct,rt,spr").  Correctness first (the patch tests are bit-exact against the slot kernel), then per-layer
# times and the whole forward.   bash tools/ab_conv_spr.sh -> gpurun_out/conv_spr_ab.txt
set -e
OUT=gpurun_out/conv_spr_ab.txt
: > $OUT
for cfg in "4,1,1,0,2,2,2" "4,1,1,1,2,2,2" "4,2,1,0,2,2,2" "4,2,1,1,2,2,2" "2,2,1,1,2,2,2" "4,1,1,0,2,2,4" "1,4,1,1,2,2,2" "1,4,1,0,2,2,2" "2,2,1,0,2,2,2"; do
  echo "== correctness under CF_CONV3_CFG=$cfg" >> $OUT
  CF_CONV3_CFG=$cfg python -m pytest tests/test_gpu_ops.py -q -x -k "test_conv3x3_f16x3_patch" 2>&1 | tail -2 >> $OUT
done
L3="8,128,128,56,100,1 16,128,128,56,100,1"
L4="8,256,256,28,50,1 16,256,256,28,50,1"
L5="8,512,512,14,25,1"
run() { echo "-- $1 [$2]" >> $OUT; CF_CONV3_CFG=$2 python tools/bench_conv.py $3 2>&1 | grep -o "^.*GF)\|f16 patch.*" | paste - - >> $OUT; }
for rep in 1 2; do
  run "level3 default" "" "$L3"
  run "level3 2,2,1 T2 SPR2" "2,2,1,1,2,2,2" "$L3"
  run "level4 default" "" "$L4"
  run "level4 4,1,1 flat SPR2" "4,1,1,0,2,2,2" "$L4"
  run "level4 4,1,1 T2 SPR2" "4,1,1,1,2,2,2" "$L4"
  run "level4 4,2,1 flat SPR2" "4,2,1,0,2,2,2" "$L4"
  run "level4 4,1,1 flat SPR4" "4,1,1,0,2,2,4" "$L4"
  run "level5 default" "" "$L5"
  run "level5 4,1,1 flat SPR2" "4,1,1,0,2,2,2" "$L5"
  run "level5 4,1,1 flat SPR4" "4,1,1,0,2,2,4" "$L5"
done
echo "== whole forward (layer_times, bs=16 one stream)" >> $OUT
for cfg in "" "4,1,1,0,2,2,2" ""; do
  echo "-- CF_CONV3_ONLY_N=256 CF_CONV3_CFG=[$cfg]" >> $OUT
  CF_CONV3_ONLY_N=256 CF_CONV3_CFG=$cfg python tools/layer_times.py --iters 5 2>&1 | grep -E "sum of launches|backbone|neck.offset|level4.tree2.tree1.conv1|level3.tree2.tree1.conv1|level5.tree2.conv1|level2.tree2.conv1" >> $OUT
done
cat $OUT
