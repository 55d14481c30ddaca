"""Dev tool (GPU box): what the range guard costs, off the hot path: check_ranges / calibrate / activation_ranges on the bench
configuration (bs=16, 3x448x800), wall clock incl. the shadow model's construction and fp32 weight packing.
    python tools/time_range_guard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config

dev = torch.device("cuda:0")
H, W, B = 448, 800, 16
m = bench.synthetic_weights(getModel(centerfusion_middle_config((H, W)))).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(B, H, W, dev, 1000)
with torch.no_grad():
    for _ in range(3):
        m(images, pc_dep=pc_dep, calib=calib)
    torch.cuda.synchronize()
    for name, fn in (("check_ranges", lambda: m.check_ranges(images, pc_dep=pc_dep, calib=calib)),
                     ("check_ranges (2nd)", lambda: m.check_ranges(images, pc_dep=pc_dep, calib=calib)),
                     ("activation_ranges (resident buffers)", lambda: m.activation_ranges()),
                     ("check_resident_ranges", lambda: m.check_resident_ranges()),
                     ("calibrate", lambda: m.calibrate(images, pc_dep=pc_dep, calib=calib))):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        print(f"{name:40s} {1e3 * (time.perf_counter() - t):9.1f} ms   ({len(r)} layers, max |input| {max(r.values()):.1f} at {max(r, key=r.get)})")
    t = time.perf_counter(); m(images, pc_dep=pc_dep, calib=calib); torch.cuda.synchronize()
    print(f"{'first forward behind calibrate (re-pack)':40s} {1e3 * (time.perf_counter() - t):9.1f} ms; pre-scales != 16: "
          f"{ {k: v for k, v in m.activation_scales().items() if v != 16} }")
    print("peak memory", torch.cuda.max_memory_allocated() / 2**30, "GiB")
