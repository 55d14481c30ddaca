#!/bin/bash
# Dev (GPU box): timing arms of head_patch16_kernel<MX> - rebuilds cf_heads.o with each -D flag and times the two head
# launches alone (tools/bench_heads.py).   bash tools/ab_heads_arms.sh "" "-DCF_MX_ARM_NOCVT" ...  -> gpurun_out/heads_arms.txt
set -e
OUT=gpurun_out/heads_arms.txt
: > $OUT
PKG=centerfusiondetect3d_amd
relink() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc ${1:+-DCF_DEV_ARMS} $1 -c $PKG/csrc/cf_heads.hip -o $PKG/_build/cf_heads.o
  hipcc --offload-arch=gfx950 -shared -fPIC $PKG/_build/*.o -o $PKG/libcfhip.so
}
# whatever happens (a failed arm build, a failed bench run under set -e): the in-tree library is the DEFAULT build again on exit
trap 'relink ""' EXIT
for arm in "$@" ; do
  echo "== arm: [$arm]" >> $OUT
  relink "$arm"
  python tools/bench_heads.py ${BENCH_ARGS} 2>&1 | grep -E "tails|pack" >> $OUT
done
relink ""
cat $OUT
