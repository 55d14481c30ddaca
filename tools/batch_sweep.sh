# Dev (GPU box): the default bench line at other batch sizes (one line per batch: ms per step, frames/s, p50 / p95) - how the
# step scales below and above the metric's bs=16.   gpurun -- bash tools/batch_sweep.sh
OUT=gpurun_out/batch_sweep.txt
: > $OUT
for B in 1 2 4 6 8 12 16 24 32 48; do
  python bench.py --batch $B --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bs=%3d  %7.3f ms/step  %8.1f frames/s  p50 %7.3f  p95 %7.3f  ms/frame %6.3f' % ($B, d['ms_per_step'], d['value'], d['step_ms_p50'], d['step_ms_p95'], d['ms_per_step']/$B))" | tee -a $OUT
done
