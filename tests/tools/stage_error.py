"""Dev tool (GPU box): per-stage normwise error of the HIP path against the fp64 oracle, next to the
CPU fp32 oracle's own error against fp64 - shows where fp32 rounding is amplified (the DCN neck).
    gpurun -- python tests/tools/stage_error.py
"""
import torch, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import model_ref
from tests.golden import cases
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
H,W,B = 448,800,1
sd = cases.tuned_state_dict(radar=True, seed=0)
x, pc_dep, calib = cases.model_inputs(B, H, W, seed=2, radar=True, n_points=(80, 200))
sd64 = {k:(v.double() if v.is_floating_point() else v) for k,v in sd.items()}
def stages(sd_, x_):
    layers = model_ref.dla34_base(sd_, x_)
    d = {f"y{i}": t for i,t in enumerate(layers)}
    layers = list(layers)
    out=[layers[-1]]
    for i in range(3):
        model_ref._ida(sd_, f"dla_up.ida_{i}", layers, len(layers)-i-2, len(layers))
        out.insert(0, layers[-1])
    for i,t in enumerate(out): d[f"up{i}"]=t
    y=[out[i].clone() for i in range(3)]
    model_ref._ida(sd_, "ida_up", y, 0, 3)
    d["feat"]=y[-1]
    return d
with torch.no_grad():
    r32 = stages(sd, x); r64 = stages(sd64, x.double())
import os as _os
m = getModel(centerfusion_middle_config((H,W))); m.load_state_dict(sd)
m.conv_f16 = _os.environ.get('CF_F16','1')=='1'; m=m.cuda()
with torch.no_grad():
    out = m(x.cuda(), pc_dep=pc_dep.cuda(), calib=calib.cuda())
plan = list(m._plans.values())[0]
dbg = dict(plan.debug); dbg["feat"]=plan.feat
def nerr(a,b): return float((a-b).abs().max()/b.abs().max())
def rerr(a,b): return float((a-b).pow(2).mean().sqrt()/b.pow(2).mean().sqrt())
for k in r64:
    if k not in dbg: continue   # (y0 stays in LDS when the stem is fused)
    g = dbg[k].permute(0,3,1,2).double().cpu()
    print(f"{k:>5s}: max-norm gpu-vs-fp64 {nerr(g,r64[k]):.2e} cpu32-vs-fp64 {nerr(r32[k].double(),r64[k]):.2e} | rms gpu-vs-fp64 {rerr(g,r64[k]):.2e} cpu32-vs-fp64 {rerr(r32[k].double(),r64[k]):.2e}")
