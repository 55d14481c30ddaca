set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_${1:-r5}_issue
mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
echo pass a done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
echo pass b done
python3 tools/pmc_kernels.py $OUT/a $OUT/b --top 30 > $OUT/pmc_issue_breakdown.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -type d -empty -delete
head -20 $OUT/pmc_issue_breakdown.txt | cut -c1-260
