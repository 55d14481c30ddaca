#!/bin/bash
# Dev (GPU box): what bounds the wave-specialised DCN kernel - its gather waves and its MFMA waves are decoupled, so
# removing one side's work shows the other's pace.  Each arm is a rebuild with -D flags (results are garbage by design).
#   bash tools/ab_dcn_ws_arms.sh -> gpurun_out/dcn_ws_arms.txt
# whatever happens: the in-tree library is the DEFAULT build again on exit
trap 'python -m centerfusiondetect3d_amd.build --force > /dev/null' EXIT
OUT=gpurun_out/dcn_ws_arms.txt
: > $OUT
for arm in "" "-DCF_DCNWS_NOMFMA" "-DCF_DCNWS_NOLOAD" "-DCF_DCNWS_NOBLEND" "-DCF_DCNWS_NOLOAD -DCF_DCNWS_NOBLEND" "-DCF_DCNWS_NOLOAD -DCF_DCNWS_NOBLEND -DCF_DCNWS_NOMFMA"; do
  echo "== arm: [$arm]" >> $OUT
  CF_EXTRA_FLAGS="${arm:+-DCF_DEV_ARMS }$arm" python -m centerfusiondetect3d_amd.build --force > /dev/null 2>&1
  CF_DCN_WS=1 CF_DCN_WS_MIN=150 timeout -k 10 200 python tools/bench_dcn.py 16,64,64,112,200 8,64,64,112,200 16,128,64,56,100 2>&1 | grep -v amdgpu.ids >> $OUT
done
python -m centerfusiondetect3d_amd.build --force > /dev/null 2>&1
cat $OUT
