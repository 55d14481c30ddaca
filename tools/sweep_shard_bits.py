"""Dev (GPU box): randomised check of the rule behind SURVEY 8(e) - a shard of a batch reproduces the full batch BIT FOR BIT -
on the kernel families whose tile shape follows the launch size (conv3x3 LDS-patch kernel: half-height tiles while the
grid fits one round - stride 1, stride 2, the fused conv2 + Root launch, which also switches between one and two launches,
and conv2 + project;
DCN: 64-pixel tiles for 64-output layers).  Random (B, H, W, Cin, Cout) around the dispatch thresholds;
every frame alone and every pair must equal its slice of the full-batch result, and the full batch must agree with float64.
    python tools/sweep_shard_bits.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from centerfusiondetect3d_amd import ops, packing

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
bad = 0
for case in range(n_cases):
    B = int(rs.choice([1, 2, 3, 4, 6, 8, 16]))
    H = int(rs.choice([7, 14, 28, 56, 112, int(rs.randint(5, 120))]))
    W = int(rs.choice([13, 25, 50, 100, 200, int(rs.randint(5, 210))]))
    Ci = int(rs.choice([32, 64, 128, 256]))
    Co = int(rs.choice([27, 64, 128, 256]))
    if B * H * W * max(Ci, Co) > 3.0e8:
        continue
    g = torch.Generator().manual_seed(case)
    x = F.relu(torch.randn(B, Ci, H, W, generator=g)) * 3
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5
    b = torch.randn(Co, generator=g)
    # ---- conv3x3 (patch kernel where the map fits, the slot kernel otherwise: the dispatcher's choice either way)
    pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)]).to(dev)
    so = 32 if Co == 27 else Co
    xd = nhwc(x).to(dev)
    full = torch.zeros((B, H, W, so), device=dev)
    ops.conv2d_f16x3(pc, [xd], B, H, W, act=1, out=full, patch=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
    err = float((full[..., :Co].permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
    ok = err < 1.5e-6
    for lo, hi in [(i, i + 1) for i in range(B)] + [(i, i + 2) for i in range(0, B - 1, 2)]:
        part = torch.zeros((hi - lo, H, W, so), device=dev)
        ops.conv2d_f16x3(pc, [xd[lo:hi].contiguous()], hi - lo, H, W, act=1, out=part, patch=True)
        ok &= bool(torch.equal(part[..., :Co], full[lo:hi, ..., :Co]))
    pairs = [(i, i + 1) for i in range(B)] + [(i, i + 2) for i in range(0, B - 1, 2)]
    # ---- stride-2 form (round 4)
    ok_s = True
    if Co != 27:
        pc2 = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)], stride=2).to(dev)
        fulls = ops.conv2d_f16x3(pc2, [xd], B, H, W, act=1, patch=True)
        refs = F.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1))
        ok_s = float((fulls.permute(0, 3, 1, 2).cpu().double() - refs).abs().max() / refs.abs().max()) < 1.5e-6
        for lo, hi in pairs:
            ok_s &= bool(torch.equal(ops.conv2d_f16x3(pc2, [xd[lo:hi].contiguous()], hi - lo, H, W, act=1, patch=True), fulls[lo:hi]))
    # ---- conv2 + Root in one launch (round 4): Cin == Cout layers, with 0-2 children of 64 / 128 channels
    ok_r = True
    if Co != 27 and Ci == Co and Co <= 256:
        kids = [int(c) for c in rs.choice([64, 128], size=int(rs.randint(0, 3)))]
        K = 2 * Co + sum(kids)
        wr, br = torch.randn(Co, K, 1, 1, generator=g) * K ** -0.5, torch.randn(Co, generator=g)
        x1 = nhwc(F.relu(torch.randn(B, Co, H, W, generator=g))).to(dev)
        chd = [nhwc(F.relu(torch.randn(B, c, H, W, generator=g))).to(dev) for c in kids]
        pcr = packing.pack_conv_f16(wr, br, [packing.Source(Co, Co), packing.Source(Co, Co)] + [packing.Source(c, c) for c in kids]).to(dev)
        fullr, _ = ops.conv3x3_root_f16x3(pc, pcr, xd, x1, chd)
        x2 = ops.conv2d_f16x3(pc, [xd], B, H, W, act=1, residual=x1)
        ok_r = bool(torch.equal(fullr, ops.conv2d_f16x3(pcr, [x2, x1, *chd], B, H, W, act=1)))
        for lo, hi in pairs:
            part, _ = ops.conv3x3_root_f16x3(pc, pcr, xd[lo:hi].contiguous(), x1[lo:hi].contiguous(), [c[lo:hi].contiguous() for c in chd])
            ok_r &= bool(torch.equal(part, fullr[lo:hi]))
    # ---- conv2 + the Tree's project of the pooled input in one launch (round 5): 64+ output channels, Cin of the 3x3 = Cout
    ok_p = True
    if Co != 27 and Ci == Co:
        Cp = int(rs.choice([32, 64, 96, 128, 256]))
        wpj, bpj = torch.randn(Co, Cp, 1, 1, generator=g) * Cp ** -0.5, torch.randn(Co, generator=g)
        pooled = F.relu(torch.randn(B, Cp, H, W, generator=g)) * 2
        pcj = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)], proj=(wpj, bpj, packing.Source(Cp, Cp))).to(dev)
        pld = nhwc(pooled).to(dev)
        fullp = ops.conv3x3_proj_f16x3(pcj, xd, pld)
        refp = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1) + F.conv2d(pooled.double(), wpj.double(), bpj.double()))
        ok_p = float((fullp.permute(0, 3, 1, 2).cpu().double() - refp).abs().max() / refp.abs().max()) < 1.5e-6
        for lo, hi in pairs:
            ok_p &= bool(torch.equal(ops.conv3x3_proj_f16x3(pcj, xd[lo:hi].contiguous(), pld[lo:hi].contiguous()), fullp[lo:hi]))
    # ---- DCN (Cout padded to 32s by the packer)
    ok_d = True
    if Co != 27:
        om = torch.zeros(B, H, W, 32)
        om[..., :18] = torch.randn(B, H, W, 18, generator=g) * 2.0
        om[..., 18:27] = torch.randn(B, H, W, 9, generator=g)
        om = om.to(dev)
        pd = packing.pack_dcn_f16(w, b).to(dev)
        fulld = ops.dcn_v2_fused(pd, xd, om)
        for lo, hi in [(i, i + 1) for i in range(B)] + [(i, i + 2) for i in range(0, B - 1, 2)]:
            partd = ops.dcn_v2_fused(pd, xd[lo:hi].contiguous(), om[lo:hi].contiguous())
            ok_d &= bool(torch.equal(partd, fulld[lo:hi]))
    print(f"case {case:3d}: B={B} {Ci}->{Co} {H}x{W}: conv err {err:.1e} {'ok' if ok else 'MISMATCH'}; stride 2 {'ok' if ok_s else 'MISMATCH'}; "
          f"conv2+root {'ok' if ok_r else 'MISMATCH'}; conv2+project {'ok' if ok_p else 'MISMATCH'}; dcn {'ok' if ok_d else 'MISMATCH'}", flush=True)
    bad += (not ok) + (not ok_d) + (not ok_s) + (not ok_r) + (not ok_p)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
