"""Dev tool: throughput of the on-device image pre-processing (cf_preprocess_images) at the nuScenes
geometry, frames resident in HBM as uint8, against its HBM roofline and the CPU oracle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import preProcessImages
from centerfusiondetect3d_amd.preprocess import NUSCENES_MEAN, NUSCENES_STD
from centerfusiondetect3d_amd.pointcloud import getAffineTransform
from oracle import preprocess_ref
B, Hs, Ws, inH, inW = 16, 900, 1600, 448, 800
frames = torch.randint(0, 256, (B, Hs, Ws, 3), dtype=torch.uint8, device="cuda")
out = torch.empty(B, 3, inH, inW, device="cuda")
for _ in range(3): preProcessImages(frames, (inH, inW), out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): preProcessImages(frames, (inH, inW), out=out)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 50 * 1e3
touched = B * (inH * inW * 3 + inH * inW * 3 * 4)      # bytes that must move: the sampled source bytes + fp32 out
print(f"cf_preprocess_images bs={B} {Hs}x{Ws} -> {inH}x{inW}: {us:.1f} us/batch = {B / us * 1e6:.0f} frames/s, "
      f"{touched / us / 1e3:.2f} GB/s of compulsory bytes ({touched / 1e6:.1f} MB)")
M = getAffineTransform(np.array([Ws / 2.0, Hs / 2.0], np.float32), 1600.0, 0, [inW, inH])
f = frames[:2].cpu().numpy()
t = time.perf_counter(); preprocess_ref.pre_process_images(list(f), M, (inH, inW), NUSCENES_MEAN, NUSCENES_STD); dt = time.perf_counter() - t
print(f"CPU oracle (numpy, 1 thread): {2 / dt:.1f} frames/s")
