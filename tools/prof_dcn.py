"""Dev tool (GPU box): phase cycle counts of dcn_f16x3_kernel (build with CF_EXTRA_FLAGS='-DCF_DEV_ARMS -DCF_DCN_PROF' first; the first
output values of every tile are then overwritten by the counters).  Phases of thread 0 per workgroup: 0 = sampling
descriptors, 1 = K loop, 2 = epilogue.     python tools/prof_dcn.py [B,C,N,H,W ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(16, 64, 64, 112, 200), (16, 128, 64, 56, 100), (16, 256, 128, 28, 50)]
for (B, C, N, H, W) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g)
    om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
    om = om.to(dev)
    pd = packing.pack_dcn_f16(torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5, torch.randn(N, generator=g)).to(dev)
    for _ in range(3):
        out = ops.dcn_v2_fused(pd, x, om, k_split=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = ops.dcn_v2_fused(pd, x, om, k_split=False)
    e1.record(); torch.cuda.synchronize()
    t = out.reshape(-1, N)[:, :3].double().cpu()
    t = t[(t[:, 1] > 1000) & (t == t.round()).all(1) & (t < 1e9).all(1) & (t >= 0).all(1)]
    tot = t.sum(1).mean()
    print(f"{B}x{C}->{N} {H}x{W}: {e0.elapsed_time(e1) * 100:.1f} us per launch, {len(t)} workgroups, {tot:.0f} cycles each")
    for i, n in enumerate(["descriptors", "K loop", "epilogue"]):
        print(f"    {n:12s} {t[:, i].mean():9.0f} cycles {100 * t[:, i].mean() / tot:5.1f} %")
