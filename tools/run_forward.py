"""Dev tool: N forwards at a given batch size (for rocprofv3 timelines).  python tools/run_forward.py B [lanes] [streams]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
B = int(sys.argv[1]); lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1; streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda")
m = bench.synthetic_weights(getModel(centerfusion_middle_config((448, 800))), seed=0).to(dev).eval()
m.lanes, m.streams = bool(lanes), streams
images, pc_dep, calib = bench.make_inputs(B, 448, 800, dev, seed=1)
with torch.no_grad():
    for _ in range(8):
        m(images, pc_dep=pc_dep, calib=calib)
torch.cuda.synchronize()
