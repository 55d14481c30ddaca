"""Dev (GPU box): the whole forward at random input geometries - every frame alone must reproduce its slice of the batch
bit for bit (all output maps) and every map and decoded row must be finite.   python tools/sweep_model_shapes.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.golden import cases
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_packed

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
sd = cases.tuned_state_dict(radar=True, seed=0)
bad = 0
for case in range(n_cases):
    H = 32 * int(rs.randint(3, 17))          # 96 .. 512
    W = 32 * int(rs.randint(4, 27))          # 128 .. 832
    B = int(rs.choice([1, 2, 3, 5, 8]))
    if B * H * W > 6 * 448 * 800:
        B = max(1, (6 * 448 * 800) // (H * W))
    m = getModel(centerfusion_middle_config((H, W)))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    x, pc_dep, calib = cases.model_inputs(B, H, W, seed=case, radar=True, n_points=(20, 120))
    xd, pd, cd = x.to(dev), pc_dep.to(dev), calib.to(dev)
    ok = True
    with torch.no_grad():
        full = m(xd, pc_dep=pd, calib=cd)
        for k, v in full[0].items():
            ok &= bool(torch.isfinite(v).all())
        for f in range(B):
            one = m(xd[f:f + 1].contiguous(), pc_dep=pd[f:f + 1].contiguous(), calib=cd[f:f + 1].contiguous())
            for k, v in one[0].items():
                if not torch.equal(v, full[0][k][f:f + 1]):
                    ok = False
                    print("   differs:", k, "frame", f)
        det, _ = decode_packed(full, (H // 4, W // 4), 100)
        ok &= bool(torch.isfinite(det).all())
    print(f"case {case:3d}: B={B} {H}x{W}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
    del m
print("mismatches:", bad)
sys.exit(1 if bad else 0)
