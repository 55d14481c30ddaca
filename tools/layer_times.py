"""Dev tool (GPU box): per-launch time / TFLOP/s table of one bs=16 forward, via HIP events around
every launch of the plan.   gpurun -- python tools/layer_times.py [--batch 16] [--fp32-heads]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_packed

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--fp32-heads", action="store_true")
ap.add_argument("--no-precise", action="store_true")
ap.add_argument("--fp32-convs", action="store_true")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--heads-bf16x3", action="store_true", help="heads' first layers on bf16x3 instead of fp16 + FP6 (A/B)")
ap.add_argument("--no-proj-fuse", action="store_true", help="the `project` convolutions as their own launches (A/B)")
a = ap.parse_args()
dev = torch.device("cuda:0")
H, W = 448, 800
m = getModel(centerfusion_middle_config((H, W)))
m.heads_bf16 = not a.fp32_heads
m.precise = not a.no_precise
m.conv_f16 = not a.fp32_convs
m.heads_mx = not a.heads_bf16x3
m.proj_fuse = not a.no_proj_fuse
m.streams = 1          # per-launch event timing needs one stream
m.lanes = False
m.heads_lanes = False
m = bench.synthetic_weights(m).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(a.batch, H, W, dev, 1000)
with torch.no_grad():
    for _ in range(2):
        decode_packed(m(images, pc_dep=pc_dep, calib=calib), (112, 200), 100)
    m.time_all(True)
    for _ in range(a.iters):
        decode_packed(m(images, pc_dep=pc_dep, calib=calib), (112, 200), 100)
rows = m.all_launch_times()
tot = sum(r[1] for r in rows)
groups = {}
for nm, ms, fl in rows:
    g = ("heads" if nm.startswith("heads") else "neck.dcn" if (".proj_" in nm or ".node_" in nm) and "offset" not in nm
         else "neck.offset" if "conv_offset_mask" in nm else "backbone" if nm.startswith("base") else "other")
    groups.setdefault(g, [0.0, 0.0]); groups[g][0] += ms; groups[g][1] += fl
    print(f"{nm:44s} {ms * 1e3:9.1f} us {fl / ms / 1e9 if ms else 0:8.1f} TF")
print(f"sum of launches {tot:.3f} ms")
for g, (ms, fl) in groups.items():
    print(f"  {g:12s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF")
