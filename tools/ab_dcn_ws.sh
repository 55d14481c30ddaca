#!/bin/bash
# Dev (GPU box): the wave-specialised DCN kernel (dcn_ws_kernel) against the tile kernels it replaces, same box.
#   bash tools/ab_dcn_ws.sh  -> gpurun_out/dcn_ws_ab.txt
OUT=gpurun_out/dcn_ws_ab.txt
: > $OUT
SH="16,64,64,112,200 8,64,64,112,200 16,128,64,56,100 8,128,64,56,100"
for rep in 1 2; do
  for ws in 0 1; do
    echo "== CF_DCN_WS=$ws CF_DCN_WS_MIN=150" >> $OUT
    CF_DCN_WS=$ws CF_DCN_WS_MIN=150 timeout -k 10 300 python tools/bench_dcn.py $SH 2>&1 | grep -v amdgpu.ids >> $OUT || { echo "bench_dcn FAILED rc=$?" >> $OUT; cat $OUT; exit 1; }
  done
done
for ws in 0 1 0 1; do
  echo "== layer_times CF_DCN_WS=$ws" >> $OUT
  CF_DCN_WS=$ws timeout -k 10 300 python tools/layer_times.py --iters 5 2>&1 | grep -E "sum of launches|neck.dcn|ida_2.node_1 |ida_2.proj_1 |ida_up.node_2 " >> $OUT
done
for ws in 0 1 0 1; do
  echo "== bench.py CF_DCN_WS=$ws" >> $OUT
  CF_DCN_WS=$ws timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" >> $OUT
done
cat $OUT
