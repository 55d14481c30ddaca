run() { python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | sed "s/^/$1 /"; }
for i in 1 2; do
run base
CF_CONV3_ONLY_N=256 CF_CONV3_CFG=4,2,1 run n256_421
CF_CONV3_ONLY_N=512 CF_CONV3_CFG=4,1,1 run n512_411
CF_CONV3_ONLY_N=512 CF_CONV3_CFG=4,2,1 run n512_421
CF_CONV3_ONLY_N=64 CF_CONV3_CFG=1,4,1,0 run n64_flat
CF_CONV3_ONLY_N=128 CF_CONV3_CFG=2,2,1,1,4 run n128_ct4_t2
done
