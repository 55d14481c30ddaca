#!/bin/bash
# Round profiles on the GPU box (one gpurun call): kernel trace + the three PMC passes of bench.py for the metric's
# configuration (C2) and for BASELINE config 5 (3x896x1600, bs=8).  Outputs under gpurun_out/prof_<tag>/.
#   bash tools/profile_round.sh r4
# The program sits directly behind `--` (no env / bash -c hop under rocprofv3); PMC passes carry only --kernel-trace.
set -e
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
C2="bench.py --steps 5 --warmup 2 --no-cpu-baseline"
C5="bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch 8 --height 896 --width 1600"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for cfg in c2 c5; do
  if [ $cfg = c2 ]; then ARGS=$C2; else ARGS=$C5; fi
  rocprofv3 --kernel-trace --stats -d $OUT/${cfg}_trace -- python3 $ARGS > $OUT/${cfg}_trace.log 2>&1
  echo "$cfg trace done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${cfg}_fetch -- python3 $ARGS > $OUT/${cfg}_fetch.log 2>&1
  echo "$cfg fetch done"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${cfg}_write -- python3 $ARGS > $OUT/${cfg}_write.log 2>&1
  echo "$cfg write done"
  rocprofv3 --pmc $SQ --kernel-trace -d $OUT/${cfg}_sq -- python3 $ARGS > $OUT/${cfg}_sq.log 2>&1
  echo "$cfg sq done"
done
python3 tools/pmc_traffic.py $OUT/c2_fetch $OUT/c2_write $OUT/c2_pmc_hbm_traffic.json 11 > $OUT/c2_traffic.txt
python3 tools/pmc_traffic.py $OUT/c5_fetch $OUT/c5_write $OUT/c5_pmc_hbm_traffic.json 11 > $OUT/c5_traffic.txt
python3 tools/pmc_kernels.py $OUT/c2_sq --top 30 > $OUT/c2_pmc_mfma_util.txt
python3 tools/pmc_kernels.py $OUT/c5_sq --top 30 > $OUT/c5_pmc_mfma_util.txt
for cfg in c2 c5; do
  python3 tools/prof_db.py $OUT/${cfg}_trace > $OUT/${cfg}_kernel_summary.txt || true
done
# per-launch tables (HIP events around every launch, one stream) and the test counts DESIGN.md quotes
python3 tools/layer_times.py --iters 5 > $OUT/layer_times_bs16.txt 2>/dev/null || true
python3 tools/layer_times.py --iters 5 --batch 1 > $OUT/layer_times_bs1.txt 2>/dev/null || true
python3 tools/check_design.py --counts > $OUT/test_counts.txt 2>&1 || true
# keep the merge small: the raw traces / databases stay on the box
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -type d -empty -delete
ls -la $OUT
