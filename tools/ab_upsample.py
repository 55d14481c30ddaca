"""Dev (GPU box): cf_upsample_dw - the 2 x 2-output-block kernel (default) against the per-pixel kernel
(CF_UPSAMPLE_BLOCK=0): sha256 of the output bytes per shape (the two must print the same hashes) and the time.
    CF_UPSAMPLE_BLOCK=0 python tools/ab_upsample.py ; CF_UPSAMPLE_BLOCK=1 python tools/ab_upsample.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda:0")
for (B, C, H, W, with_skip) in [(16, 64, 56, 100, True), (8, 64, 56, 100, True), (16, 128, 28, 50, True), (16, 256, 14, 25, True),
                                (2, 64, 7, 9, True), (3, 128, 5, 1, False), (1, 64, 1, 6, True), (2, 32, 9, 11, True)]:
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = packing.pack_upsample(torch.randn(C, 1, 4, 4, generator=g)).to(dev)
    skip = torch.randn(B, 2 * H, 2 * W, C, generator=g).to(dev) if with_skip else None
    out = ops.upsample_dw(x, w, 2, skip=skip)
    torch.cuda.synchronize()
    h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): ops.upsample_dw(x, w, 2, skip=skip, out=out)
    e0.record()
    for _ in range(50): ops.upsample_dw(x, w, 2, skip=skip, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{B}x{C} {H}x{W} skip={with_skip}: {h}  {e0.elapsed_time(e1) * 20:.1f} us", flush=True)
