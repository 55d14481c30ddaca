"""Dev tool: time the LDS-patch 3x3 conv on one layer shape under each forced tiling (CF_CONV3_CFG).
   python tools/bench_conv_cfg.py B,C,N,H,W [...]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda")
def timeit(fn, n=60, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
w_ = torch.randn(4096, 4096, device=dev)
for _ in range(50): w_ = (w_ @ w_) * 1e-4
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
cfgs = (os.environ.get("CFGS") or ",2,2,1;2,1,2;4,1,1;4,2,1;1,4,1;1,4,1,1;2,2,1,0,1;1,4,1,0,1;1,4,1,1,1;4,1,1,0,1").split(";")
cfgs = ["" if c == "," else c.lstrip(",") for c in cfgs]
for (B, C, N, H, W) in shapes:
    x = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(N, C, 3, 3) * (C * 9) ** -0.5
    pc = packing.pack_conv_f16(w, torch.randn(N), [packing.Source(C, C)], stride=1).to(dev)
    out = torch.empty(B, H, W, (N + 31) // 32 * 32 if N % 4 else N, device=dev)
    ref = None
    res = []
    for c in cfgs:
        if c: os.environ["CF_CONV3_CFG"] = c
        else: os.environ.pop("CF_CONV3_CFG", None)
        o = ops.conv2d_f16x3(pc, [x], B, H, W, act=1, out=out, patch=True).clone()
        if ref is None: ref = o
        err = float((o - ref).abs().max())
        t = timeit(lambda: ops.conv2d_f16x3(pc, [x], B, H, W, act=1, out=out, patch=True))
        res.append(f"{c or 'default':9s} {t:7.1f} us (max diff vs default {err:.1e})")
    gf = 2.0 * B * H * W * N * C * 9 / 1e9
    print(f"{B}x{C}->{N} {H}x{W} ({gf:.2f} GF):\n   " + "\n   ".join(res), flush=True)
os.environ.pop("CF_CONV3_CFG", None)
