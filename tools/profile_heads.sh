# Counters of the two head launches alone (tools/bench_heads.py): three PMC passes, joined by tools/pmc_kernels.py.
#   gpurun -- bash tools/profile_heads.sh [extra bench_heads.py flags]
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_heads
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_heads.py --iters 5 $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $OUT/c -- python3 $ARGS > $OUT/c.log 2>&1
python3 tools/pmc_kernels.py $OUT/a $OUT/b $OUT/c --top 6 > $OUT/pmc_heads.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -type d -empty -delete
cut -c1-400 $OUT/pmc_heads.txt
