"""Dev tool: forward latency at small batch - one stream, two-lane neck (model.lanes), HIP-graph replay."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
H, W = 448, 800
dev = torch.device("cuda")
model = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W))), seed=0).to(dev).eval()
def run(B, **flags):
    for k, v in flags.items():
        setattr(model, k, v)
    model.invalidate()
    images, pc_dep, calib = bench.make_inputs(B, H, W, dev, seed=1)
    with torch.no_grad():
        for _ in range(5): model(images, pc_dep=pc_dep, calib=calib)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 50
        for _ in range(n): model(images, pc_dep=pc_dep, calib=calib)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for B in (1, 2, 4, 16):
    a = run(B, lanes=False, use_graph=False, streams=1)
    b = run(B, lanes=True, use_graph=False, streams=1)
    c = run(B, lanes=False, use_graph=True, streams=1)
    d = run(B, lanes=True, use_graph=False, streams=2)
    print(f"bs={B}: one stream {a:.3f} ms | two-lane neck {b:.3f} ms | graph replay {c:.3f} ms | default (lanes, streams=2) {d:.3f} ms "
          f"= {B / d * 1e3:.0f} frames/s", flush=True)
