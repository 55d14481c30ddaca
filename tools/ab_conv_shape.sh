#!/bin/bash
# Dev (GPU box): what the 16x16x32 MFMA shape would buy the 3x3 patch kernel BEFORE rewriting it for that shape - the
# timing arm -DCF_CONV3_SHAPE16T issues the same FLOPs, LDS reads and weight loads as two 16x16x32 MFMAs per 32x32x16 one
# (garbage results).   bash tools/ab_conv_shape.sh  -> gpurun_out/conv_shape_ab.txt
set -e
# whatever happens under set -e: the in-tree library is the DEFAULT build again on exit
trap 'python -m centerfusiondetect3d_amd.build --force > /dev/null' EXIT
OUT=gpurun_out/conv_shape_ab.txt
: > $OUT
SHAPES="8,64,64,112,200,1 8,128,128,56,100,1 8,256,256,28,50,1 8,512,512,14,25,1 8,64,27,112,200,1 8,128,27,56,100,1"
for arm in "" "-DCF_CONV3_SHAPE16T" "" "-DCF_CONV3_SHAPE16T"; do
  echo "== arm: [$arm]" >> $OUT
  CF_EXTRA_FLAGS="${arm:+-DCF_DEV_ARMS }$arm" python -m centerfusiondetect3d_amd.build --force > /dev/null 2>&1
  python tools/bench_conv.py $SHAPES 2>&1 | sed 's/fp32 [0-9.]* us | //' >> $OUT
  python tools/layer_times.py --iters 5 2>&1 | grep -E "sum of launches|backbone|neck.offset" >> $OUT
done
python -m centerfusiondetect3d_amd.build --force > /dev/null 2>&1
cat $OUT
