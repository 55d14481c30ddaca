"""Dev tool (GPU box): the two head launches of one bs=16 448x800 forward, alone, back to back - HIP events around N
repetitions of the plan's own cf_head_fused steps (buffers as a real forward left them).
    gpurun -- python tools/bench_heads.py [--batch 16] [--iters 20] [--heads-bf16x3]
Under rocprofv3 --pmc this is the cheap way to counters of head_patch16_kernel alone (tools/pmc_kernels.py)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--heads-bf16x3", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
H, W = 448, 800
m = getModel(centerfusion_middle_config((H, W)))
m.heads_mx = not a.heads_bf16x3
m.streams = 1
m = bench.synthetic_weights(m).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(a.batch, H, W, dev, 1000)
with torch.no_grad():
    for _ in range(2):
        out = m(images, pc_dep=pc_dep, calib=calib)
torch.cuda.synchronize()
plan = list(m._plans.values())[-1]
st = _lib.stream_ptr()
for name in ("tails.primary", "tails.secondary") + (("feat.pack_mx",) if "feat.pack_mx" in plan.step_index else ()):
    step = plan.steps[plan.step_index[name]]
    for _ in range(3):
        step[0](*step[1:], st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        rc = step[0](*step[1:], st)
        assert rc == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = plan.step_flops.get(name, 0.0)
    print(f"{name:18s} {ms * 1e3:8.1f} us  {fl / ms / 1e9 if ms else 0:7.1f} TFLOP/s", flush=True)
