#!/bin/bash
# Dev (GPU box): time + issue-slot counters of dcn_f16x3_kernel for a list of timing arms (each a rebuild of cf_gemm_f16.o with
# -D flags; results of the arms are garbage by design).   bash tools/profile_dcn_arms.sh "" "-DCF_DCN_NOLOAD" ...
#   -> gpurun_out/dcn_arm_profile.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/dcn_arm_profile.txt
: > $OUT
PKG=centerfusiondetect3d_amd
SHAPE=${SHAPE:-16,64,64,112,200}
relink() {
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$PKG/csrc ${1:+-DCF_DEV_ARMS} $1 -c $PKG/csrc/cf_gemm_f16.hip -o $PKG/_build/cf_gemm_f16.o
  hipcc --offload-arch=gfx950 -shared -fPIC $PKG/_build/*.o -o $PKG/libcfhip.so
}
# whatever happens (a failed arm build, a failed bench run under set -e): the in-tree library is the DEFAULT build again on exit
trap 'relink ""' EXIT
for arm in "$@" ; do
  echo "== arm: [$arm]" >> $OUT
  relink "$arm"
  python tools/bench_dcn.py $SHAPE 2>&1 | grep -v Warn >> $OUT
  D=gpurun_out/prof_dcn_arm; rm -rf $D; mkdir -p $D
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d $D/a -- python3 tools/run_dcn_once.py $SHAPE > $D/a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $D/b -- python3 tools/run_dcn_once.py $SHAPE > $D/b.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $D/c -- python3 tools/run_dcn_once.py $SHAPE > $D/c.log 2>&1
  python3 tools/pmc_kernels.py $D/a $D/b $D/c --top 3 --raw | grep -A1 "dcn_f16x3" >> $OUT
  rm -rf $D
done
relink ""
cat $OUT
