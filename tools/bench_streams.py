"""Dev tool: forward + decode throughput with the batch on 1 / 2 / 4 concurrent HIP streams (model.streams)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_packed
H, W = 448, 800
dev = torch.device("cuda")
model = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W))), seed=0).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(16, H, W, dev, seed=1)
def step():
    out = model(images, pc_dep=pc_dep, calib=calib)
    return decode_packed(out, (H // 4, W // 4), 100)[0]
with torch.no_grad():
    ref = step()
    for n in (1, 2, 4, 1, 2):
        model.streams = n
        for _ in range(3): d = step()
        assert torch.equal(d, ref)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 30
        print(f"streams={n}: {dt * 1e3:.3f} ms per 16 frames = {16 / dt:.0f} frames/s")
