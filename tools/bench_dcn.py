"""Dev tool (GPU box): time cf_dcn_v2_f16x3 on layer shapes.  python tools/bench_dcn.py [B,C,N,H,W ...] [--mag 2.0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
mag = float(sys.argv[sys.argv.index("--mag") + 1]) if "--mag" in sys.argv else 2.0
shapes = [tuple(int(v) for v in a.split(",")) for a in args] or [(16, 64, 64, 112, 200), (8, 64, 64, 112, 200), (16, 128, 64, 56, 100), (16, 256, 128, 28, 50)]
w_ = torch.randn(4096, 4096, device=dev)
for _ in range(30): w_ = (w_ @ w_) * 1e-4            # warm the clocks
for (B, C, N, H, W) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    om = torch.zeros(B, H, W, 32)
    om[..., :18] = torch.randn(B, H, W, 18, generator=g) * mag
    om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
    om = om.to(dev)
    pd = packing.pack_dcn_f16(torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5, torch.randn(N, generator=g)).to(dev)
    best = 1e9
    for rep in range(3):
        for _ in range(3): out = ops.dcn_v2_fused(pd, x, om)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out = ops.dcn_v2_fused(pd, x, om)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    print(f"{B}x{C}->{N} {H}x{W} offsets ~{mag} px: {best:.1f} us  checksum {float(out.double().sum()):.6e}", flush=True)
