"""Dev tool (GPU box): phase cycle counts of conv3x3_f16x3_kernel (build with CF_EXTRA_FLAGS='-DCF_DEV_ARMS -DCF_CONV3_PROF' first;
outputs are then overwritten by the counters).  Phases of thread 0 per workgroup: 0 = prologue (first patch),
1 = tap loops, 2 = end of round (second half of the next patch + barrier), 3 = epilogue.
   python tools/prof_conv3.py B,C,N,H,W [...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(16, 64, 64, 112, 200), (16, 128, 128, 56, 100), (16, 256, 256, 28, 50)]
for (B, C, N, H, W) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5
    pc = packing.pack_conv_f16(w, torch.randn(N, generator=g), [packing.Source(C, C)], stride=1).to(dev)
    out = torch.empty(B, H, W, N, device=dev)
    for _ in range(3):
        ops.conv2d_f16x3(pc, [x], B, H, W, act=0, out=out, patch=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_f16x3(pc, [x], B, H, W, act=0, out=out, patch=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    t = out[..., :4].reshape(-1, 4).double().cpu()
    t = t[(t[:, 1] > 1000) & (t == t.round()).all(1) & (t < 1e9).all(1) & (t >= 0).all(1)]
    tot = t.sum(1).mean()
    print(f"{B}x{C}->{N} {H}x{W}: {us:.1f} us per launch, {len(t)} workgroups, {tot:.0f} cycles per workgroup")
    for i, n in enumerate(["prologue", "tap loops", "round ends", "epilogue"]):
        print(f"    {n:12s} {t[:, i].mean():9.0f} cycles {100 * t[:, i].mean() / tot:5.1f} %")
