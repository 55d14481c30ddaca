"""Dev tool: time one conv shape on the three conv paths (fp32 MFMA, f16x3 slots, f16x3 patch)."""
import torch, sys, os, time
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import ops, packing, _lib
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(16, 16, 16, 448, 800, 1), (16, 16, 32, 448, 800, 2)]
for (B, C, N, H, W, stride) in shapes:
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(N, C, 3, 3) * (C * 9) ** -0.5
    bias = torch.randn(N)
    src = [packing.Source(C, C)]
    pr = packing.pack_conv(w, bias, src, stride=stride).to(dev)
    t32 = timeit(lambda: ops.conv2d_fused(pr, [x], B, H, W, act=1))
    res = [f'fp32 {t32:.1f} us']
    if C % 8 == 0:
        pc = packing.pack_conv_f16(w, bias, src, stride=stride).to(dev)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        out = torch.empty(B, Ho, Wo, N if N != 27 else 32, device=dev)
        res.append(f'f16 slots {timeit(lambda: ops.conv2d_f16x3(pc, [x], B, H, W, act=1, out=out, patch=False)):.1f} us')
        if pc.patch:
            res.append(f'f16 patch {timeit(lambda: ops.conv2d_f16x3(pc, [x], B, H, W, act=1, out=out, patch=True)):.1f} us')
    gf = 2.0 * B * (H // stride) * (W // stride) * N * C * 9 / 1e9
    print(f'{B}x{C}->{N} {H}x{W} s{stride} ({gf:.1f} GF):', ' | '.join(res))
