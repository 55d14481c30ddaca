"""Dev tool (CPU): the oracle's fp32 and float64 forwards of Centerfusion_Middle (tuned weights seed 0, inputs seed 5) at B H W ->
/tmp/e2e_B_H_W.pt (feature map, fp32 outputs, float64 outputs, frustum map): the inputs of tools/eval_head_mx.py.
    python tests/tools/eval_head_mx_inputs.py 2 448 800"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import model_ref, frustum_ref, mx_emul
from tests.golden import cases
torch.set_num_threads(8)
radar = True
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sd = cases.tuned_state_dict(radar=radar, seed=0)
x, pc_dep, calib = cases.model_inputs(B, H, W, seed=5, radar=radar, n_points=(80, 200))
t0 = time.time()
with torch.no_grad():
    feat32 = model_ref.img2feats(sd, x)
    print("feat32", time.time() - t0, flush=True)
    r32 = model_ref.forward(sd, x, pc_dep=pc_dep, calib=calib, radar=radar)[0]
    print("r32", time.time() - t0, flush=True)
    hm = frustum_ref.pc_frustum_heatmap(r32, pc_dep, calib, 100, 60.0)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    r64 = model_ref.forward(sd64, x.double(), pc_dep=pc_dep.double(), calib=calib, radar=radar, pc_hm_override=hm)[0]
    print("r64", time.time() - t0, flush=True)
torch.save({"feat32": feat32, "r32": r32, "r64": r64, "hm": hm}, f"/tmp/e2e_{B}_{H}_{W}.pt")
