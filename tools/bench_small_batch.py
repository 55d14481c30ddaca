import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_post_packed
from centerfusiondetect3d_amd.postprocess import inverse_affine_device
from centerfusiondetect3d_amd import model as M
dev = torch.device("cuda:0")
H, W = 448, 800
for B in (8, 12, 16):
    images, pc_dep, calib = bench.make_inputs(B, H, W, dev, seed=2000)
    tinv = inverse_affine_device(np.array([800.0, 450.0], np.float32), 1600.0, (W // 4, H // 4), dev)
    for streams, lmf in ((1, 4), (2, 4)):        # (lanes_max_frames 16 instead of 4: B=8 5.16 vs 5.23, B=16 two streams 8.96 vs 8.42)
        m = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W)))).to(dev).eval()
        m.streams = streams
        m.lanes_max_frames = lmf
        with torch.no_grad():
            for _ in range(6): decode_post_packed(m(images, pc_dep=pc_dep, calib=calib), calib, tinv, (H // 4, W // 4), 100)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): decode_post_packed(m(images, pc_dep=pc_dep, calib=calib), calib, tinv, (H // 4, W // 4), 100)
            torch.cuda.synchronize()
        print(f"B={B} streams={streams} lanes_max_frames={lmf}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", "trunk plans" if any("trunk" in k for k in m._plans) else "single plan", flush=True)
        del m
