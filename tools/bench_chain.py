"""Dev tool (GPU box): the index chain between the two head launches, each launch alone (HIP events, back to back on one stream,
a cold-ish map: a 64 MB fill between iterations evicts nothing from the 256 MB Infinity Cache but drains the L2s' dirty lines).
    python tools/bench_chain.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops, _lib
from tests.golden import cases

B = int(next((a for a in sys.argv[1:] if not a.startswith("--")), "16"))
dev = torch.device("cuda:0")
y, pc_dep, calib = cases.frustum_case(0, B=2)
rep = lambda t: t.repeat((B + 1) // 2, *([1] * (t.dim() - 1)))[:B].contiguous().to(dev)
d = {k: rep(v) for k, v in y.items()}
d["heatmap"] = torch.rand_like(d["heatmap"]) * 0.01 + d["heatmap"]
pc_dep, calib = rep(pc_dep), rep(calib)
if "--small-boxes" in sys.argv:                 # the bench configuration's boxes (widthHeight bias 9 x 7): ROIs of ~100 pixels
    d["widthHeight"] = torch.stack([torch.full_like(d["widthHeight"][:, 0], 9.0), torch.full_like(d["widthHeight"][:, 1], 7.0)], 1) \
        + torch.randn_like(d["widthHeight"])
K = 100
lib = _lib.load()
scratch = torch.empty(16 << 20, device=dev)


def timed(fn, iters=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        scratch.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


s, inds, cls = ops.topk_peaks(d["heatmap"], K, nms=False)
pk = ops.topk_peaks(d["heatmap"], K, nms=True)
sums = torch.empty(2 * ops.CHECKSUM_PARTS, device=dev, dtype=torch.int64)
ops.checksum64(d["heatmap"], out=sums[:ops.CHECKSUM_PARTS])


def guard():
    ops.checksum64(d["heatmap"], out=sums[ops.CHECKSUM_PARTS:])
    ops.topk_peaks(d["heatmap"], K, nms=True, out=pk, only_if_changed=sums)


rows = [("topk_peaks (slice + merge)", lambda: ops.topk_peaks(d["heatmap"], K, nms=False)),
        ("topk_peaks nms=2 (nms + slice + merge)", lambda: ops.topk_peaks(d["heatmap"], K, nms=True)),
        ("frustum_assoc (peaks given)", lambda: ops.frustum_assoc(inds, d["depth"], d["widthHeight"], d["dimension"], d["rotation"], calib, pc_dep, 60.0)),
        ("topk_frustum (slice + merge-in-prologue frustum)", lambda: ops.topk_frustum(d["heatmap"], d["depth"], d["widthHeight"], d["dimension"], d["rotation"], calib, pc_dep, K, 60.0, want_split8=True)),
        ("checksum64(heat)", lambda: ops.checksum64(d["heatmap"])),
        ("peaks guard, unchanged map (checksum + conditional launch)", lambda: guard()),
        ("absmax(heat)", lambda: ops.absmax(d["heatmap"])),
        ("empty launch pair (spin 1 us x2)", lambda: (lib.cf_spin_us(1, _lib.stream_ptr()), lib.cf_spin_us(1, _lib.stream_ptr())))]
for name, fn in rows:
    med, mn = timed(fn)
    print(f"{name:52s} median {med:7.1f} us   min {mn:7.1f} us   (B={B})", flush=True)
