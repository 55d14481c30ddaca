"""Dev tool (GPU box): do the two trunk sub-batches of the bs=16 step run side by side?  HIP events around each trunk
(model.record_spans), no profiler: per trunk its span, when it starts relative to the first, and the overlap.
    python tools/trunk_balance.py [--no-trunk-on-caller] [--split a,b]"""
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
if "--no-trunk-on-caller" in sys.argv:
    m.trunk_on_caller = False
m.record_spans = True
m = bench.synthetic_weights(m).to(dev).eval()
images, pc_dep, calib = bench.make_inputs(B, H, W, dev, 1000)
tinv = torch.from_numpy(inverse_affine((W / 2.0, H / 2.0), float(max(H, W)), (W // 4, H // 4))).to(dev)
with torch.no_grad():
    for _ in range(5):
        decode_post_packed(m(images, pc_dep=pc_dep, calib=calib), calib, tinv, outputSize=(H // 4, W // 4), K=100)
    rows = []
    for _ in range(20):
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
        decode_post_packed(m(images, pc_dep=pc_dep, calib=calib), calib, tinv, outputSize=(H // 4, W // 4), K=100)
        t1 = torch.cuda.Event(enable_timing=True); t1.record()
        spans = m.trunk_spans
        torch.cuda.synchronize()
        (a0, a1), (b0, b1) = spans[:2]
        rows.append((t0.elapsed_time(a0), a0.elapsed_time(a1), t0.elapsed_time(b0), b0.elapsed_time(b1), t0.elapsed_time(t1)))
rows.sort(key=lambda r: r[4])
r = rows[len(rows) // 2]
print(f"median step {r[4]:.3f} ms: trunk A starts +{r[0]:.3f} ms, runs {r[1]:.3f} ms; trunk B starts +{r[2]:.3f} ms, runs {r[3]:.3f} ms; "
      f"A ends +{r[0] + r[1]:.3f}, B ends +{r[2] + r[3]:.3f}  (synchronised per step: the host is NOT ahead here)")
