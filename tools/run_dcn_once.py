"""Dev tool (GPU box): a few launches of cf_dcn_v2_f16x3 on ONE layer shape, for rocprofv3 --pmc passes.
    python tools/run_dcn_once.py [B,C,N,H,W] [n_launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda:0")
B, C, N, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,64,64,112,200").split(","))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, C, generator=g).to(dev)
om = torch.zeros(B, H, W, 32)
om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 2.0
om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
om = om.to(dev)
pd = packing.pack_dcn_f16(torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5, torch.randn(N, generator=g)).to(dev)
for _ in range(n):
    out = ops.dcn_v2_fused(pd, x, om)
torch.cuda.synchronize()
