"""Dev tool (GPU box): run ONE layer kind in a loop for a few seconds (for tools/sample_clocks.sh: clock / power per kernel).
    python tools/loop_layer.py conv3|offset|dcn|stem|up B,C,N,H,W [seconds]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from centerfusiondetect3d_amd import ops, packing
dev = torch.device("cuda")
kind = sys.argv[1]
B, C, N, H, W = (int(v) for v in sys.argv[2].split(","))
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
g = torch.Generator().manual_seed(0)
if kind in ("conv3", "offset"):
    n = 27 if kind == "offset" else N
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    pc = packing.pack_conv_f16(torch.randn(n, C, 3, 3, generator=g) * (C * 9) ** -0.5, torch.randn(n, generator=g), [packing.Source(C, C)]).to(dev)
    out = torch.empty(B, H, W, 32 if n == 27 else n, device=dev)
    fn = lambda: ops.conv2d_f16x3(pc, [x], B, H, W, act=1, out=out, patch=True)
elif kind == "dcn":
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    om = torch.zeros(B, H, W, 32); om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 2; om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
    om = om.to(dev)
    pd = packing.pack_dcn_f16(torch.randn(N, C, 3, 3, generator=g) * (C * 9) ** -0.5, torch.randn(N, generator=g)).to(dev)
    fn = lambda: ops.dcn_v2_fused(pd, x, om)
elif kind == "up":
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    skip = torch.randn(B, 2 * H, 2 * W, C, generator=g).to(dev)
    wk = torch.randn(4, 4, C, generator=g).to(dev)
    out = torch.empty(B, 2 * H, 2 * W, C, device=dev)
    lib = ops._lib.load()
    fn = lambda: ops._lib.check(lib.cf_upsample_dw(x.data_ptr(), wk.data_ptr(), skip.data_ptr(), out.data_ptr(), B, H, W, C, 2, ops._lib.stream_ptr()), "up")
else:
    raise SystemExit("kind?")
for _ in range(5): fn()
torch.cuda.synchronize()
t0 = time.perf_counter(); n_it = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(50): fn()
    n_it += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
print(f"{kind} {B}x{C}->{N} {H}x{W}: {e0.elapsed_time(e1) / n_it * 1e3:.1f} us per launch over {n_it} launches")
