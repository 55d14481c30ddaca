#!/bin/bash
# Dev (GPU box): time cf_dcn_v2_f16x3 on layer shapes under rebuilds of cf_gemm_f16.o with -D flags.
#   SHAPES="8,64,64,112,200 ..." bash tools/ab_dcn_flags.sh "" "-DCF_DCN_NODEEP" ...   -> gpurun_out/dcn_flags.txt
set -e
OUT=gpurun_out/dcn_flags.txt
: > $OUT
PKG=centerfusiondetect3d_amd
SHAPES=${SHAPES:-"16,64,64,112,200 8,64,64,112,200 16,128,64,56,100 8,128,64,56,100 8,256,64,28,50"}
relink() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc ${1:+-DCF_DEV_ARMS} $1 -c $PKG/csrc/cf_gemm_f16.hip -o $PKG/_build/cf_gemm_f16.o
  hipcc --offload-arch=gfx950 -shared -fPIC $PKG/_build/*.o -o $PKG/libcfhip.so
}
# whatever happens (a failed arm build, a failed bench run under set -e): the in-tree library is the DEFAULT build again on exit
trap 'relink ""' EXIT
for arm in "$@" ; do
  echo "== flags: [$arm]" >> $OUT
  relink "$arm"
  python tools/bench_dcn.py $SHAPES 2>&1 | grep -v amdgpu >> $OUT
  if [ -n "$BENCH" ]; then python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  step', d['ms_per_step'], d['step_ms_p50'])" >> $OUT; fi
done
relink ""
cat $OUT
