"""Dev tool: glitch stress of the f16x3 conv kernels (generic + patch) under cache/timing perturbation;
reference = fp32 kernel."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import ops, packing
torch.manual_seed(0)
dev = torch.device('cuda')
R = int(os.environ.get('REPS', '200'))
cases = [(16, 64, 64, 112, 200, 1, False), (16, 64, 128, 112, 200, 2, False), (16, 128, 128, 56, 100, 1, False),
         (16, 128, 256, 56, 100, 2, False), (16, 256, 256, 28, 50, 1, False), (16, 512, 512, 14, 25, 1, False),
         (16, 64, 64, 112, 200, 1, True), (16, 128, 128, 56, 100, 1, True), (16, 256, 256, 28, 50, 1, True),
         (16, 64, 27, 112, 200, 1, True), (16, 256, 27, 28, 50, 1, True), (16, 128, 27, 56, 100, 1, True)]
for (B, C, N, H, W, stride, patch) in cases:
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(N, C, 3, 3) * (C * 9) ** -0.5
    bias = torch.randn(N)
    pc = packing.pack_conv_f16(w, bias, [packing.Source(C, C)], stride=stride).to(dev)
    pr = packing.pack_conv(w, bias, [packing.Source(C, C)], stride=stride).to(dev)
    ref = ops.conv2d_fused(pr, [x], B, H, W)
    so = 32 if N == 27 else N
    Ho, Wo = ref.shape[1], ref.shape[2]
    outs = [torch.full((B, Ho, Wo, so), float('nan'), device=dev) for _ in range(3)]
    noise = torch.randn(4096, 4096, device=dev)
    bad = 0
    for i in range(R):
        k = i % 3
        if i % 2: noise = (noise @ noise) * 1e-4
        else: outs[k].fill_(float('nan'))
        ops.conv2d_f16x3(pc, [x], B, H, W, out=outs[k], patch=patch)
        e = float((outs[k][..., :N] - ref).abs().nan_to_num(9.0).max())
        if e > 1e-3:
            bad += 1
            if bad <= 2: print('     glitch', i, e)
    print(f'conv patch={patch} {B}x{C}->{N} {H}x{W} s{stride}: {bad}/{R} glitched launches')
