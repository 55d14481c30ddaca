"""Dev tool (GPU box): A/B of the PRECISE (two-level) accumulation - accuracy vs fp64 and step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import model_ref
from tests.golden import cases
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, decode_packed
H, W = 448, 800
sd = cases.tuned_state_dict(radar=True, seed=0)
x, pc_dep, calib = cases.model_inputs(1, H, W, seed=2, radar=True, n_points=(80, 200))
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
with torch.no_grad():
    r32 = model_ref.forward(sd, x, pc_dep=pc_dep, calib=calib)[0]
    r64 = model_ref.forward(sd64, x.double(), pc_dep=pc_dep.double(), calib=calib.double())[0]
def nerr(a, b): return float((a.double() - b).abs().max() / b.abs().max())
for precise, f16 in ((True, False), (True, True)):
    m = getModel(centerfusion_middle_config((H, W))); m.load_state_dict(sd); m.precise = precise; m.conv_f16 = f16; m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda(), pc_dep=pc_dep.cuda(), calib=calib.cuda())[0]
    worst = max((nerr(y[k].cpu(), r64[k]), k) for k in r64 if k not in ("calib",) and r64[k].abs().max() > 0)
    cpu = max((nerr(r32[k], r64[k]), k) for k in r64 if k not in ("calib",) and r64[k].abs().max() > 0)
    vs32 = max((nerr(y[k].cpu(), r32[k].double()), k) for k in r64 if k not in ("calib",) and r64[k].abs().max() > 0)
    plan = list(m._plans.values())[0]
    f = nerr(plan.feat.permute(0, 3, 1, 2).cpu(), model_ref.img2feats(sd64, x.double()))
    print(f"precise={precise} conv_f16={f16}: worst head gpu-vs-fp64 {worst[0]:.2e} ({worst[1]}), cpu32-vs-fp64 {cpu[0]:.2e}, gpu-vs-cpu32 {vs32[0]:.2e}, feat gpu-vs-fp64 {f:.2e}")
    B = 16
    xb = torch.randn(B, 3, H, W, device="cuda"); pb = pc_dep.cuda().repeat(B, 1, 1, 1); cb = calib.cuda().repeat(B, 1, 1)
    with torch.no_grad():
        for _ in range(3): decode_packed(m(xb, pc_dep=pb, calib=cb), (112, 200), 100)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): decode_packed(m(xb, pc_dep=pb, calib=cb), (112, 200), 100)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"   bs=16 step {dt*1e3:.2f} ms = {B/dt:.1f} frames/s")
    del m
