"""Dev tool: forward + decode throughput with the trunk on 1 / 2 / 4 concurrent HIP streams (model.streams), launched
eagerly from Python or replayed as one captured HIP graph (model.use_graph)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_packed
H, W, B = 448, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
model = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W))), seed=0).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(B, H, W, dev, seed=1)
for graph in (False, True):
    for n in (1, 2, 4, 8):
        if B // n < 4 and n > 1:
            continue
        model.streams, model.use_graph = n, graph
        with torch.no_grad():
            for _ in range(4): decode_packed(model(images, pc_dep=pc_dep, calib=calib), (H // 4, W // 4), 100)
            torch.cuda.synchronize(); t = time.perf_counter()
            k = 30
            for _ in range(k): decode_packed(model(images, pc_dep=pc_dep, calib=calib), (H // 4, W // 4), 100)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / k
        print(f"graph={int(graph)} streams={n}: {dt * 1e3:.3f} ms per {B} frames = {B / dt:.0f} frames/s", flush=True)
