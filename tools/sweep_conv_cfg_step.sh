# Dev (GPU box): whole-step A/B of forced 3x3 tilings, one output width at a time (CF_CONV3_ONLY_N + CF_CONV3_CFG; the forced
# form applies to the plain stride-1 launches of that width only) - do the dispatch thresholds still hold at a trunk's sub-batch?
#   gpurun -- bash tools/sweep_conv_cfg_step.sh
OUT=gpurun_out/sweep_conv_cfg_step.txt
: > $OUT
run() { # label, env...
  local label=$1; shift
  local v=$(env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_p50'])")
  echo "$label: $v" | tee -a $OUT
}
run "default"
for cfg in "64 1,4,1" "64 1,4,1,1" "64 2,2,1" "64 4,1,1" "128 2,2,1" "128 2,2,1,1" "128 1,4,1,1" "128 4,1,1" "128 4,2,1" "256 4,1,1" "256 4,2,1" "256 2,2,1" "256 2,1,2" "512 4,1,1" "512 4,2,1" "512 2,1,2"; do
  set -- $cfg
  run "N=$1 cfg=$2" CF_CONV3_ONLY_N=$1 CF_CONV3_CFG=$2
done
run "default"
