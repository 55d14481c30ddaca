import torch, sys, os
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import ops, packing
dev = torch.device('cuda')
B, H, W = 16, 448, 800
x = torch.randn(B, 3, H, W, device=dev)
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)
ps = packing.pack_stem(r(16, 3, 7, 7) * 0.08, r(16), r(16, 16, 3, 3) * 0.08, r(16), r(32, 16, 3, 3) * 0.08, r(32)).to(dev)
out = torch.empty(B, H // 2, W // 2, 32, device=dev)
for _ in range(3): ops.stem_fused(ps, x, out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.stem_fused(ps, x, out)
e.record(); torch.cuda.synchronize()
print(f'stem fused: {s.elapsed_time(e) / 20 * 1e3:.1f} us')
