"""Dev: where does a forced conv3x3 tiling differ from the default?  python tools/dbg_conv_ct4.py "2,2,1,0,4" B,C,N,H,W"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda")
cfg = sys.argv[1]
B, C, N, H, W = (int(v) for v in sys.argv[2].split(","))
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device=dev)
w = torch.randn(N, C, 3, 3) * (C * 9) ** -0.5
pc = packing.pack_conv_f16(w, torch.randn(N), [packing.Source(C, C)], stride=1).to(dev)
os.environ.pop("CF_CONV3_CFG", None)
ref = ops.conv2d_f16x3(pc, [x], B, H, W, act=0, patch=True).clone()
os.environ["CF_CONV3_CFG"] = cfg
got = ops.conv2d_f16x3(pc, [x], B, H, W, act=0, patch=True).clone()
d = (got - ref).abs()
bad = (d > 1e-4 * ref.abs().max()) | ~torch.isfinite(got)
print("bad fraction", float(bad.float().mean()), "max", float(d.max()))
flat = bad.view(-1, N)
pix = flat.any(1).nonzero().flatten()
print("bad pixels:", pix[:40].tolist(), "... count", pix.numel(), "of", flat.shape[0])
ch = flat.any(0).nonzero().flatten()
print("bad channels:", ch.tolist()[:70])
if pix.numel():
    import collections
    print("pixel % 128 // 32 histogram:", collections.Counter(((pix % 128) // 32).tolist()))
    print("pixel % 256 // 32 histogram:", collections.Counter(((pix % 256) // 32).tolist()))
