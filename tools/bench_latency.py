"""Dev tool: forward latency at small batch, eager launches vs HIP-graph replay (model.use_graph)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
H, W = 448, 800
dev = torch.device("cuda")
model = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W))), seed=0).to(dev).eval()
for B in (1, 2, 4, 16):
    images, pc_dep, calib = bench.make_inputs(B, H, W, dev, seed=1)
    res = []
    for use_graph in (False, True):
        model.use_graph = use_graph
        with torch.no_grad():
            for _ in range(5): model(images, pc_dep=pc_dep, calib=calib)
            torch.cuda.synchronize(); t = time.perf_counter()
            n = 50
            for _ in range(n): model(images, pc_dep=pc_dep, calib=calib)
            torch.cuda.synchronize(); res.append((time.perf_counter() - t) / n * 1e3)
    print(f"bs={B}: forward eager {res[0]:.3f} ms, graph replay {res[1]:.3f} ms")
