# Dev (GPU box): A/B of two builds of libcfhip.so on ONE box - the library under centerfusiondetect3d_amd/_ab/libcfhip_prev.so (built by
# hand from an older source, git-ignored, travels with the snapshot) against the in-tree one, alternating bench runs + layer times.
#   gpurun -- bash tools/ab_lib_prev.sh
set -e
P=centerfusiondetect3d_amd
cp $P/libcfhip.so /tmp/new.so
# whatever happens: the in-tree library is the new build again on exit
trap 'cp /tmp/new.so $P/libcfhip.so' EXIT
for i in 1 2 3; do
  cp /tmp/new.so $P/libcfhip.so; python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/ab_lib_new.txt
  cp $P/_ab/libcfhip_prev.so $P/libcfhip.so; python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/ab_lib_prev.txt
done
cp /tmp/new.so $P/libcfhip.so
python tools/layer_times.py > gpurun_out/ab_lib_lt_new.txt 2>&1
cp $P/_ab/libcfhip_prev.so $P/libcfhip.so
python tools/layer_times.py > gpurun_out/ab_lib_lt_prev.txt 2>&1
cp /tmp/new.so $P/libcfhip.so
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 or proj" > gpurun_out/ab_lib_t.log 2>&1; tail -2 gpurun_out/ab_lib_t.log
