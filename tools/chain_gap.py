"""Dev tool (GPU box): the time between the END of the primary head launch and the START of the secondary one in the default
two-stream bs=16 step, without the profiler: HIP events recorded on the launch stream around the two head launches
(model.time_launch), gap = end(primary) -> start(secondary).  Also the tail: end(secondary) -> end of the decode.
    python tools/chain_gap.py [--old]     (--old: cf_topk_peaks + cf_frustum_assoc, lane behind the primary launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
from centerfusiondetect3d_amd.decode import decode_post_packed
from centerfusiondetect3d_amd.postprocess import inverse_affine

dev = torch.device("cuda:0")
H, W, B = 448, 800, 16
m = getModel(centerfusion_middle_config((H, W)))
if "--old" in sys.argv:
    m.frustum_fused, m.peaks_behind_frustum = False, False
m = bench.synthetic_weights(m).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(B, H, W, dev, 1000)
tinv = torch.from_numpy(inverse_affine((W / 2.0, H / 2.0), float(max(H, W)), (W // 4, H // 4))).to(dev)


def step():
    out = m(images, pc_dep=pc_dep, calib=calib)
    return decode_post_packed(out, calib, tinv, outputSize=(H // 4, W // 4), K=100)


with torch.no_grad():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for n in ("tails.primary", "tails.secondary"):
        m.time_launch(n, True)
    ends = []
    for _ in range(30):
        step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ends.append(e)
    torch.cuda.synchronize()
plans = [p for p in m._all_plans() if "tails.primary" in p.step_index]
assert len(plans) == 1
p = plans[0]
prim = p.timed[p.step_index["tails.primary"]]
sec = p.timed[p.step_index["tails.secondary"]]
gaps = sorted(a[1].elapsed_time(b[0]) * 1e3 for a, b in zip(prim, sec))
tails = sorted(b[1].elapsed_time(e) * 1e3 for b, e in zip(sec, ends))
print(f"{'old' if '--old' in sys.argv else 'new'}: primary end -> secondary start: median {gaps[len(gaps) // 2]:.1f} us (min {gaps[0]:.1f}); "
      f"secondary end -> decode end: median {tails[len(tails) // 2]:.1f} us (min {tails[0]:.1f})")
