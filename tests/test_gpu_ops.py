"""GPU parity, operator level: every C-ABI entry point against the CPU oracle on seeded inputs.
fp32 GEMM-family kernels: rtol 1e-4 / atol 1e-4 (the north-star budget is 1e-3 end to end);
index-path kernels (top-k, frustum, pillar, decode): bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dcn_ref, frustum_ref, decode_ref, pillar_ref
from tests.golden import cases


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X box"
    from centerfusiondetect3d_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(a, b, rtol=1e-4, atol=1e-4):
    torch.testing.assert_close(a.cpu(), b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("B,Ci,Co,H,W,k,stride,act,res", [
    (2, 16, 16, 40, 56, 3, 1, 1, False),     # level0-like, N_pad 32
    (2, 16, 32, 40, 56, 3, 2, 1, False),     # level1 stride 2
    (1, 64, 64, 28, 50, 3, 1, 1, True),      # BasicBlock conv2 + residual
    (2, 128, 256, 14, 25, 3, 2, 1, False),   # 128-wide N tile, stride 2
    (1, 512, 512, 7, 13, 3, 1, 1, True),     # small M, long K
    (3, 64, 27, 23, 31, 3, 1, 0, False),     # conv_offset_mask (N=27), ragged M
    (1, 256, 10, 16, 24, 1, 1, 0, False),    # head output 1x1
])
def test_conv2d_fused(dev, B, Ci, Co, H, W, k, stride, act, res):
    from centerfusiondetect3d_amd import ops, packing
    x, w, b = rnd(B, Ci, H, W, seed=1), rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5), rnd(Co, seed=3)
    bn = (torch.rand(Co) + 0.5, rnd(Co, seed=4, scale=0.1), rnd(Co, seed=5, scale=0.1), torch.rand(Co) + 0.5)
    ref = F.conv2d(x, w, b, stride, k // 2)
    ref = F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    r = rnd(*ref.shape, seed=6) if res else None
    if res:
        ref = ref + r
    if act:
        ref = F.relu(ref)
    wf, bf = packing.fold_bn(w, b, bn)
    pc = packing.pack_conv(wf, bf, [packing.Source(Ci, Ci)], stride=stride).to(dev)
    out = ops.conv2d_fused(pc, [nhwc(x).to(dev)], B, H, W, act=act,
                           residual=nhwc(r).to(dev) if res else None)
    close(nchw(out), ref)


def test_conv2d_multi_source_root_and_small_source(dev):
    """Root: 1x1 over a never-materialised concat of 4 sources; secondary head: 3x3 over
    feat(64) || pc_hm(3 channels stored with stride 4)."""
    from centerfusiondetect3d_amd import ops, packing
    B, H, W = 2, 14, 25
    chans = [128, 128, 64, 128]
    xs = [rnd(B, c, H, W, seed=10 + i) for i, c in enumerate(chans)]
    w = rnd(128, sum(chans), 1, 1, seed=20, scale=sum(chans) ** -0.5)
    ref = F.relu(F.conv2d(torch.cat(xs, 1), w))
    pc = packing.pack_conv(w, torch.zeros(128), [packing.Source(c, c) for c in chans]).to(dev)
    out = ops.conv2d_fused(pc, [nhwc(x).to(dev) for x in xs], B, H, W, act=1)
    close(nchw(out), ref)

    feat, pch = rnd(B, 64, H, W, seed=30), rnd(B, 3, H, W, seed=31)
    w = rnd(256, 67, 3, 3, seed=32, scale=(67 * 9) ** -0.5)
    b = rnd(256, seed=33)
    ref = F.relu(F.conv2d(torch.cat([feat, pch], 1), w, b, 1, 1))
    pc = packing.pack_conv(w, b, [packing.Source(64, 64), packing.Source(3, 4)]).to(dev)
    pch4 = torch.cat([nhwc(pch), torch.full((B, H, W, 1), float("nan"))], dim=3)  # pad lane must be ignored... by zero weights
    pch4 = torch.nan_to_num(pch4, nan=0.0)
    out = ops.conv2d_fused(pc, [nhwc(feat).to(dev), pch4.to(dev)], B, H, W, act=1)
    close(nchw(out), ref)


def test_conv2d_stem_from_nchw_image_and_nchw_outputs(dev):
    from centerfusiondetect3d_amd import ops, packing
    B, H, W = 2, 32, 48
    x = rnd(B, 3, H, W, seed=1)
    w = rnd(16, 3, 7, 7, seed=2, scale=147 ** -0.5)
    ref = F.relu(F.conv2d(x, w, None, 1, 3))
    pc = packing.pack_conv(w, torch.zeros(16), [packing.Source(3, 4)]).to(dev)
    x4 = ops.nchw_to_nhwc4(x.to(dev))
    close(x4[..., :3], nhwc(x), 0, 0)
    assert float(x4[..., 3].abs().max()) == 0.0
    out = ops.conv2d_fused(pc, [x4], B, H, W, act=1)
    close(nchw(out), ref)
    close(ops.nhwc_to_nchw(out), ref)
    # NCHW epilogues: plain, sigmoid-clamp, raw + sigmoid-depth; HoWo not a multiple of 4
    for (hh, ww) in ((28, 50), (9, 7)):
        f = rnd(B, 256, hh, ww, seed=3)
        w1, b1 = rnd(10, 256, 1, 1, seed=4, scale=1 / 16), rnd(10, seed=5)
        raw = F.conv2d(f, w1, b1)
        pc1 = packing.pack_conv(w1, b1, [packing.Source(256, 256)]).to(dev)
        fd = nhwc(f).to(dev)
        close(ops.conv2d_fused(pc1, [fd], B, hh, ww, layout=1), raw)
        close(ops.conv2d_fused(pc1, [fd], B, hh, ww, layout=1, act=2),
              torch.clamp(torch.sigmoid(raw), 1e-4, 1 - 1e-4), 1e-5, 1e-6)
        o1, o2 = ops.conv2d_fused(pc1, [fd], B, hh, ww, layout=1, act=3)
        close(o1, raw)
        close(o2, 1.0 / (torch.sigmoid(raw) + 1e-6) - 1.0, 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------- dcn
@pytest.mark.parametrize("B,Ci,Co,H,W,mag", [(2, 64, 64, 28, 50, 2.0), (1, 128, 64, 14, 25, 8.0),
                                             (1, 512, 256, 7, 13, 1.0), (2, 256, 128, 9, 11, 30.0)])
def test_dcn_v2_fused(dev, B, Ci, Co, H, W, mag):
    from centerfusiondetect3d_amd import ops, packing
    x = rnd(B, Ci, H, W, seed=1)
    om = rnd(B, 27, H, W, seed=2)
    om[:, :18] *= mag
    w, b = rnd(Co, Ci, 3, 3, seed=3, scale=(Ci * 9) ** -0.5), rnd(Co, seed=4)
    bn = (torch.rand(Co) + 0.5, rnd(Co, seed=5, scale=0.1), rnd(Co, seed=6, scale=0.1), torch.rand(Co) + 0.5)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    ref = dcn_ref.deform_conv2d(x, torch.cat((o1, o2), 1), w, b, (1, 1), (1, 1), (1, 1), torch.sigmoid(m))
    ref = F.relu(F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5))
    wf, bf = packing.fold_bn(w, b, bn)
    pd = packing.pack_dcn(wf, bf).to(dev)
    om32 = torch.zeros(B, H, W, 32)
    om32[..., :27] = nhwc(om)
    out = ops.dcn_v2_fused(pd, nhwc(x).to(dev), om32.to(dev))
    close(nchw(out), ref, 2e-4, 2e-4)


def test_dcn_known_answers_on_device(dev):
    """KATs run through the HIP kernel itself: zero offset/unit mask == conv2d; out of range == bias."""
    from centerfusiondetect3d_amd import ops, packing
    B, Ci, Co, H, W = 1, 64, 32, 12, 17
    x, w, b = rnd(B, Ci, H, W, seed=1), rnd(Co, Ci, 3, 3, seed=2, scale=1 / 24), rnd(Co, seed=3)
    pd = packing.pack_dcn(w, b).to(dev)
    om = torch.zeros(B, H, W, 32)
    om[..., 18:27] = 30.0                      # sigmoid(30) == 1.0f
    out = ops.dcn_v2_fused(pd, nhwc(x).to(dev), om.to(dev), act=0)
    close(nchw(out), F.conv2d(x, w, b, 1, 1))
    om[..., :18] = 1000.0
    out = ops.dcn_v2_fused(pd, nhwc(x).to(dev), om.to(dev), act=0)
    close(nchw(out), b.view(1, Co, 1, 1).expand(B, Co, H, W), 0, 1e-6)


# ----------------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("f,C,H,W", [(2, 64, 14, 25), (4, 64, 7, 13), (2, 256, 5, 9), (2, 128, 5, 1), (2, 64, 1, 6), (2, 24, 3, 4)])
def test_upsample_dw(dev, f, C, H, W):
    from centerfusiondetect3d_amd import ops, packing
    x, w = rnd(2, C, H, W, seed=1), rnd(C, 1, 2 * f, 2 * f, seed=2)
    skip = rnd(2, C, H * f, W * f, seed=3)
    ref = F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=C)
    wk = packing.pack_upsample(w).to(dev)
    close(nchw(ops.upsample_dw(nhwc(x).to(dev), wk, f)), ref, 1e-5, 1e-5)
    close(nchw(ops.upsample_dw(nhwc(x).to(dev), wk, f, skip=nhwc(skip).to(dev))), ref + skip, 1e-5, 1e-5)


def test_maxpool2x2(dev):
    from centerfusiondetect3d_amd import ops
    x = rnd(2, 32, 14, 26, seed=1)
    close(nchw(ops.maxpool2x2(nhwc(x).to(dev))), F.max_pool2d(x, 2, 2), 0, 0)


# ------------------------------------------------------------------------------------------ topk
def _check_topk(dev, heat, K, nms):
    from centerfusiondetect3d_amd import ops
    ref_heat = decode_ref.nms(heat) if nms else heat
    s, inds, cls, _, _ = frustum_ref.topk(ref_heat, K)
    if nms:      # negative scores: suppression RAISES them to -0 (heat * keep), both device forms must agree
        neg = (heat - 0.5).to(dev)
        a, b = ops.topk_peaks(neg, K, nms=True), ops.topk_peaks(neg, K, nms=1)
        r = frustum_ref.topk(decode_ref.nms(heat - 0.5), K)
        for x, y, z in zip(a, b, r[:3]):
            assert torch.equal(x, y) and np.array_equal(x.cpu().numpy(), z.numpy())
    gs, gi, gc = ops.topk_peaks(heat.to(dev), K, nms=nms)
    assert np.array_equal(gs.cpu().numpy(), s.numpy())
    assert np.array_equal(gi.cpu().numpy(), inds.numpy())
    assert np.array_equal(gc.cpu().numpy(), cls.numpy())


@pytest.mark.parametrize("nms", [False, True, 1])     # True: suppressed map first (two passes); 1: on the fly
def test_topk_random_and_ties(dev, nms):
    _check_topk(dev, cases.decode_case(0)["heatmap"], 100, nms)              # tie-free
    _check_topk(dev, cases.decode_case(3, tie_heavy=True)["heatmap"], 100, nms)  # plateau + ties
    flat = torch.full((2, 10, 112, 200), 1e-4)                               # everything ties
    _check_topk(dev, flat, 100, nms)
    _check_topk(dev, cases.frustum_case(1)[0]["heatmap"], 100, nms)


def test_topk_overflow_path_and_odd_shapes(dev):
    # > 4096 elements above the local-max bound: one thread's share holds all the large values
    heat = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0)) * 0.1
    flat = heat.view(-1)
    flat[0::1024][:12] = 0.95            # stride-1024 lane 0 owns 12 huge values
    big = torch.rand(6000, generator=torch.Generator().manual_seed(1)) * 0.3 + 0.5
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(2))[:6000]
    idx = idx[idx % 1024 < 90]           # only 90 distinct lanes hold the large values
    flat[idx] = big[: idx.numel()]
    _check_topk(dev, heat, 100, False)
    _check_topk(dev, heat, 100, True)
    for shape, K in (((1, 1, 5, 40), 100), ((3, 2, 17, 23), 7), ((1, 10, 224, 400), 100)):
        h = torch.rand(*shape, generator=torch.Generator().manual_seed(3))
        _check_topk(dev, h, K, False)
        _check_topk(dev, h, K, True)
    neg = rnd(2, 4, 20, 30, seed=9)      # negative scores order correctly too
    _check_topk(dev, neg, 50, False)


@pytest.mark.parametrize("stride,n_lanes", [(256, 90), (1024, 360)], ids=["kernel_256_threads", "register_kernel_1024_threads"])
@pytest.mark.parametrize("n_big", [300, 3000, 6000])
def test_topk_candidate_counts_across_the_three_ordering_paths(dev, n_big, stride, n_lanes):
    """A slice (14,000 elements at 10 x 112 x 200, scanned by 256 threads with stride 256 - behind the NMS - or by 1,024
    threads with stride 1,024 and group maxima over 4 neighbouring threads - the register-cached kernel of the plain top-K)
    whose large values all sit in fewer than K of the lanes / groups: the K-th largest LOCAL maximum is then small and every
    large value is a candidate - 300 of them are ordered by rank counting, 3,000 by the bitonic sort (256-thread kernel:
    > 1,024) or still by rank counting, 6,000 (> 4,096, the LDS capacity; 4,900 with stride 1,024) go through the radix
    select first.  All must give the oracle's top-K, also behind the NMS."""
    g = torch.Generator().manual_seed(n_big)
    heat = torch.rand(2, 10, 112, 200, generator=g) * 0.1
    flat = heat[1].view(-1)                                       # image 1, slice 0 = its first 14,000 elements
    lanes = torch.arange(14000)
    pool = lanes[lanes % stride < n_lanes]
    pick = pool[torch.randperm(pool.numel(), generator=g)[:n_big]] if n_big <= pool.numel() else pool
    flat[pick] = torch.rand(pick.numel(), generator=g) * 0.4 + 0.5
    _check_topk(dev, heat, 100, False)
    _check_topk(dev, heat, 100, True)


# --------------------------------------------------------------------------------------- frustum
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frustum_bit_exact_vs_reference_golden(dev, golden_dir, seed):
    from centerfusiondetect3d_amd import ops
    g = np.load(os.path.join(golden_dir, f"frustum_{seed}.npz"))
    y, pc_dep, calib = cases.frustum_case(seed)
    d = {k: v.to(dev) for k, v in y.items()}
    s, inds, cls = ops.topk_peaks(d["heatmap"], 100, nms=False)
    assert np.array_equal(inds.cpu().numpy(), g["topk_inds"])
    assert np.array_equal(cls.cpu().numpy(), g["topk_cls"])
    assert np.array_equal(s.cpu().numpy(), g["topk_scores"])
    pc_hm, hm4 = ops.frustum_assoc(inds, d["depth"], d["widthHeight"], d["dimension"], d["rotation"],
                                   calib.to(dev), pc_dep.to(dev), 60.0, want_nhwc4=True)
    flat = pc_hm.cpu().reshape(-1).numpy()
    nz = np.nonzero(flat)[0]
    assert np.array_equal(nz, g["nz_idx"])
    assert np.array_equal(flat[nz], g["nz_val"])
    assert np.array_equal(hm4[..., :3].cpu().numpy(), pc_hm.permute(0, 2, 3, 1).cpu().numpy())
    assert float(hm4[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("seed,B,H,W", [(0, 2, 112, 200), (1, 2, 112, 200), (2, 2, 112, 200), (7, 3, 28, 50), (8, 1, 224, 400)])
def test_topk_frustum_two_launches_equal_the_three(dev, seed, B, H, W):
    """cf_topk_frustum (slice top-K + association with the merge in its prologue; model.frustum_fused) against
    cf_topk_peaks + cf_frustum_assoc: peaks, painted map, its NHWC copy and its split-bf16 rows, bit for bit - on the
    reference-golden cases (borders, overlaps, gate misses), a small map and a 224 x 400 one (the 256-thread slice kernel)."""
    from centerfusiondetect3d_amd import ops
    y, pc_dep, calib = cases.frustum_case(seed, B=B, H=H, W=W)
    d = {k: v.to(dev) for k, v in y.items()}
    s, inds, cls = ops.topk_peaks(d["heatmap"], 100, nms=False)
    hm8_ref = torch.zeros((B, H, W, 2, 8), device=dev, dtype=torch.bfloat16)
    pc_ref, hm4_ref = ops.frustum_assoc(inds, d["depth"], d["widthHeight"], d["dimension"], d["rotation"], calib.to(dev),
                                        pc_dep.to(dev), 60.0, want_nhwc4=True, pc_hm_split8=hm8_ref)
    pc_hm, hm4, hm8, (s2, i2, c2) = ops.topk_frustum(d["heatmap"], d["depth"], d["widthHeight"], d["dimension"], d["rotation"],
                                                     calib.to(dev), pc_dep.to(dev), 100, 60.0, want_nhwc4=True,
                                                     want_split8=True, want_peaks=True)
    assert torch.equal(s2, s) and torch.equal(i2, inds) and torch.equal(c2, cls)
    assert torch.equal(pc_hm, pc_ref) and int((pc_ref != 0).sum()) > 0
    assert torch.equal(hm4, hm4_ref)
    assert torch.equal(hm8.view(torch.int16), hm8_ref.view(torch.int16))
    # a large ROI (a box covering most of the map: several rounds of the four-loads-per-lane search) with ties in depth
    big = {k: v.clone() for k, v in d.items()}
    big["widthHeight"][:] = max(H, W) * 0.8
    pcd = pc_dep.clone()
    pcd[:, 0][pcd[:, 0] != 0] = 20.0                                # every radar return at the same depth: the FIRST one in row-major order wins
    big["depth"][:] = 20.0
    _, inds_b, _ = ops.topk_peaks(big["heatmap"], 100, nms=False)
    ref_b = frustum_ref.pc_frustum_heatmap({k: v.cpu() for k, v in big.items()}, pcd, calib, 100, 60.0)
    got_b = ops.topk_frustum(big["heatmap"], big["depth"], big["widthHeight"], big["dimension"], big["rotation"],
                             calib.to(dev), pcd.to(dev), 100, 60.0)
    assert np.array_equal(got_b.cpu().numpy(), ref_b.numpy())
    assert torch.equal(got_b, ops.frustum_assoc(inds_b, big["depth"], big["widthHeight"], big["dimension"], big["rotation"],
                                                calib.to(dev), pcd.to(dev), 60.0))


def test_topk_and_guard_at_large_k(dev):
    """K above 128 takes the other branches of the round-6 kernels: 256 group maxima in the register-cached slice kernel, 38-49 KB
    of merge scratch (tree merge in `topk_merge_kernel`, in the frustum prologue and in the guard's one-workgroup fallback,
    which needs its dynamic-LDS limit raised beside 33 KB of static LDS)."""
    from centerfusiondetect3d_amd import ops
    g = torch.Generator().manual_seed(21)
    heat = torch.rand(2, 10, 112, 200, generator=g)
    for K in (129, 200, 256):
        _check_topk(dev, heat, K, False)
        _check_topk(dev, heat, K, True)
    # the guard: unchanged map -> the carried peaks stand (even if they are garbage); changed map -> recomputed, = cf_topk_peaks
    K = 256
    hd = heat.to(dev)
    ref = ops.topk_peaks(hd, K, nms=True)
    sums = torch.empty(2 * ops.CHECKSUM_PARTS, device=dev, dtype=torch.int64)
    ops.checksum64(hd, out=sums[:ops.CHECKSUM_PARTS])
    ops.checksum64(hd, out=sums[ops.CHECKSUM_PARTS:])
    assert torch.equal(sums[:ops.CHECKSUM_PARTS], sums[ops.CHECKSUM_PARTS:])
    carried = tuple(torch.full_like(t, 7) for t in ref)
    out = ops.topk_peaks(hd, K, nms=True, out=carried, only_if_changed=sums)
    assert all(bool((t == 7).all()) for t in out)                      # equal sums: nothing ran
    hd[1, 4, 50, 60] = 2.0                                             # one element changes
    ops.checksum64(hd, out=sums[ops.CHECKSUM_PARTS:])
    assert int((sums[:ops.CHECKSUM_PARTS] != sums[ops.CHECKSUM_PARTS:]).sum()) == 1     # exactly one part sees it
    out = ops.topk_peaks(hd, K, nms=True, out=carried, only_if_changed=sums)
    ref2 = ops.topk_peaks(hd, K, nms=True)
    assert all(torch.equal(a, b) for a, b in zip(out, ref2)) and not torch.equal(ref2[0], ref[0])
    # frustum chain at K = 160: two launches = three launches
    y, pc_dep, calib = cases.frustum_case(3, B=2, K=160)
    d = {k: v.to(dev) for k, v in y.items()}
    _, inds, _ = ops.topk_peaks(d["heatmap"], 160, nms=False)
    a = ops.frustum_assoc(inds, d["depth"], d["widthHeight"], d["dimension"], d["rotation"], calib.to(dev), pc_dep.to(dev), 60.0)
    b = ops.topk_frustum(d["heatmap"], d["depth"], d["widthHeight"], d["dimension"], d["rotation"], calib.to(dev), pc_dep.to(dev), 160, 60.0)
    assert torch.equal(a, b) and int((a != 0).sum()) > 0
    assert np.array_equal(a.cpu().numpy(), frustum_ref.pc_frustum_heatmap(y, pc_dep, calib, 160, 60.0).numpy())


def test_frustum_no_radar_and_single_box(dev):
    from centerfusiondetect3d_amd import ops
    y, pc_dep, calib = cases.frustum_case(5, B=1)
    d = {k: v.to(dev) for k, v in y.items()}
    _, inds, _ = ops.topk_peaks(d["heatmap"], 100)
    out = ops.frustum_assoc(inds, d["depth"], d["widthHeight"], d["dimension"], d["rotation"],
                            calib.to(dev), torch.zeros_like(pc_dep).to(dev))
    assert float(out.abs().max()) == 0.0
    ref = frustum_ref.pc_frustum_heatmap(y, pc_dep, calib, 1, 60.0)
    _, inds1, _ = ops.topk_peaks(d["heatmap"], 1)
    out = ops.frustum_assoc(inds1, d["depth"], d["widthHeight"], d["dimension"], d["rotation"],
                            calib.to(dev), pc_dep.to(dev))
    assert np.array_equal(out.cpu().numpy(), ref.numpy())


# ---------------------------------------------------------------------------------------- decode
@pytest.mark.parametrize("name,seed,radar,norm2d", [("decode_0.npz", 0, True, False),
                                                    ("decode_1.npz", 1, False, False),
                                                    ("decode_2_norm2d.npz", 2, True, True)])
def test_decode_bit_exact_vs_reference_golden(dev, golden_dir, name, seed, radar, norm2d):
    from centerfusiondetect3d_amd.decode import fusionDecode
    g = np.load(os.path.join(golden_dir, name))
    out = {k: v.to(dev) for k, v in cases.decode_case(seed, radar=radar).items()}
    keys_before = set(out.keys())
    det = fusionDecode([out], outputSize=(112, 200), K=100, norm2d=norm2d)
    assert set(det.keys()) == set(g.files)
    for k in g.files:
        assert det[k].shape == g[k].shape, k
        assert np.array_equal(det[k].cpu().numpy(), g[k]), k
    # the reference renames rotation2 -> rotation in the caller's dict (decode.py:120-121)
    if radar:
        assert "rotation2" not in out and "rotation" in out
    else:
        assert set(out.keys()) == keys_before


# ---------------------------------------------------------------------------------------- pillar
def _pillar_batch(frames, max_n, dev):
    B = len(frames)
    p2 = np.zeros((B, 3, max_n)); p3 = np.zeros((B, 18, max_n)); cnt = np.zeros(B, np.int32)
    cal = np.zeros((B, 3, 4)); tr = np.zeros((B, 2, 3))
    for b, (a, c, k, t) in enumerate(frames):
        n = a.shape[1]
        p2[b, :, :n], p3[b, :, :n], cnt[b], cal[b], tr[b] = a, c, n, k, t
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(p2), t(p3), t(cnt), t(cal), t(tr)


@pytest.mark.parametrize("out_hw,img_wh", [((112, 200), (1600, 900)), ((224, 400), (1600, 900))])
def test_pillar_expand_bit_exact_vs_oracle(dev, out_hw, img_wh):
    from centerfusiondetect3d_amd import ops
    rng = np.random.default_rng(0)
    trans = pillar_ref.affine_transform_matrix((img_wh[0] / 2, img_wh[1] / 2), float(max(img_wh)),
                                               (out_hw[1], out_hw[0]))
    frames = []
    for n in (0, 1, 37, 200, 600):
        a, c, k = pillar_ref.synth_radar(rng, n)
        frames.append((a, c, k, trans))
    args = _pillar_batch(frames, 640, dev)
    pc_dep, keep, xy = ops.pillar_expand(*args, out_hw, want_aux=True)
    for b, (a, c, k, t) in enumerate(frames):
        tp, p3, dm = pillar_ref.process_point_cloud(a, c, k, t, out_hw)
        assert np.array_equal(pc_dep[b].cpu().numpy(), dm), f"frame {b}"
        n = a.shape[1]
        kb = keep[b, :n].cpu().numpy().astype(bool)
        assert kb.sum() == tp.shape[1]
        assert np.array_equal(xy[b, :, :n].cpu().numpy()[:, kb], tp[:2])
        assert not keep[b, n:].any()


# ------------------------------------------------------------------------------- bf16x3 (heads)
def _split(x_nchw, dev, cs=None):
    from centerfusiondetect3d_amd import ops
    return ops.split_bf16(nhwc(x_nchw).to(dev), cs=cs)


def _unsplit(t):
    """(B,H,W,2,C) bf16 -> fp32 NCHW  (hi + lo)."""
    f = t.float()
    return (f[..., 0, :] + f[..., 1, :]).permute(0, 3, 1, 2).contiguous()


def test_split_bf16_roundtrip(dev):
    x = rnd(2, 67, 9, 11, seed=1) * 7
    s = _split(x, dev, cs=72)
    assert s.shape == (2, 9, 11, 2, 72) and s.dtype == torch.bfloat16
    back = _unsplit(s).cpu()
    assert float(back[:, 67:].abs().max()) == 0.0
    rel = ((back[:, :67] - x).abs() / x.abs().clamp_min(1e-20)).max()
    assert float(rel) < 2 ** -16                     # hi + lo carries 16 significant bits
    hi = s[..., 0, :67].float().permute(0, 3, 1, 2).cpu()
    assert torch.equal(hi, x.to(torch.bfloat16).float())


def test_legacy_head_kernels_are_not_in_the_default_build(dev):
    """The kernels of rounds 1-3 that nothing dispatches any more - cf_conv2d_bf16x3 (unfused heads), cf_head_tail, cf_head_fused
    on 32x32x16 fragments or on the slot table - are compiled only with -DCF_LEGACY_HEADS: the default library still exports the
    entry points (include/cf_hip.h) and answers them with an error that says so, never with a wrong result."""
    from centerfusiondetect3d_amd import ops, packing, _lib
    B, H, W = 1, 8, 16
    x = rnd(B, 64, H, W, seed=1)
    w, b = rnd(256, 64, 3, 3, seed=2, scale=1 / 24), rnd(256, seed=3)
    pc = packing.pack_conv_bf16(w, b, [packing.Source(64, 64)]).to(dev)
    with pytest.raises(_lib.CfHipError, match="legacy kernel path"):
        ops.conv2d_bf16x3(pc, [_split(x, dev)], B, H, W, act=1)
    hid = _split(F.relu(rnd(B, 256, H, W, seed=4)), dev)
    wo = rnd(8, 256, seed=5, scale=1 / 16)
    head = dict(c_base=0, w_hidden=[], b_hidden=[], w_out=packing.pack_fragments(wo).to(dev), b_out=torch.zeros(32, device=dev),
                n_out=8, act=0, out=torch.empty(B, 8, H, W, device=dev), out2=None)
    with pytest.raises(_lib.CfHipError, match="legacy kernel path"):
        ops.run_head_tail(ops.head_tail_args(hid, 256, B, H, W, [head]))
    pc32 = packing.pack_conv_bf16(w, b, [packing.Source(64, 64)], fragments=True).to(dev)           # 32x32x16 fragments
    h32 = dict(head, w_first=pc32.weight, b_first=pc32.bias[:256].contiguous(), w_out_perm=packing.pack_fragments(wo, acc_order=True).to(dev),
               mfma16=False)
    f = ops.head_fused_args([_split(x, dev)], [64], pc32.slots, pc32.k_pad, B, H, W, [h32])
    assert f.layout3x3 == 1 and f.mfma16 == 0
    with pytest.raises(_lib.CfHipError, match="legacy kernel path"):
        ops.run_head_fused(f)
    f0 = ops.head_fused_args([_split(x, dev)], [64], pc32.slots, pc32.k_pad, B, H, W, [dict(h32, w_out_perm=None)])
    assert f0.layout3x3 == 0
    with pytest.raises(_lib.CfHipError, match="legacy kernel path"):
        ops.run_head_fused(f0)


@pytest.mark.parametrize("n_hidden,radar,B,H,W,patch", [
    (0, False, 2, 9, 14, 16),         # 2-D patch kernel on the 16x16x32 MFMA shape: one partial 8x16 tile per image
    (0, True, 1, 21, 37, 16),         # ... with the pc_hm taps, ragged tiles in both directions
    (0, False, 3, 16, 32, 16),        # ... exact tiling
    (2, True, 2, 13, 19, 16),         # hidden layers behind the patch first layer
    (1, False, 1, 8, 40, 16),
])
def test_head_fused_whole_head(dev, n_hidden, radar, B, H, W, patch):
    """3x3 conv (feat [|| pc_hm]) + ReLU -> hidden chain -> output, one launch, vs fp32 torch: the 2-D LDS-patch kernel on
    v_mfma_f32_16x16x32_bf16 fragments (patch == 16; the 32x32x16 and slot-table forms of rounds 1-3 are legacy builds only)."""
    m16 = patch == 16
    patch = bool(patch)
    from centerfusiondetect3d_amd import ops, packing
    n_outs, acts = [10, 1, 3, 8], [2, 3, 0, 0]
    feat, pch = rnd(B, 64, H, W, seed=1), rnd(B, 3, H, W, seed=2)
    srcs = [_split(feat, dev)] + ([_split(pch, dev, cs=8)] if radar else [])
    sources = [packing.Source(64, 64)] + ([packing.Source(3, 8)] if radar else [])
    xin = torch.cat([feat, pch], 1) if radar else feat
    ci = xin.shape[1]
    heads, refs, slots, k_pad = [], [], None, None
    for i, (no, act) in enumerate(zip(n_outs, acts)):
        w1, b1 = rnd(256, ci, 3, 3, seed=300 + i, scale=(ci * 9) ** -0.5), rnd(256, seed=310 + i, scale=0.1)
        x = F.relu(F.conv2d(xin, w1, b1, 1, 1))
        pc = packing.pack_conv_bf16(w1, b1, sources, fragments=16 if m16 else True).to(dev)
        slots, k_pad = pc.slots, pc.k_pad
        wh, bh = [], []
        for l in range(n_hidden):
            w, b = rnd(256, 256, 1, 1, seed=10 * i + l, scale=1 / 16), rnd(256, seed=50 + 10 * i + l, scale=0.1)
            x = F.relu(F.conv2d(x, w, b))
            wh.append((packing.pack_fragments16 if m16 else packing.pack_fragments)(w.view(256, 256)).to(dev)); bh.append(b.to(dev))
        w, b = rnd(no, 256, 1, 1, seed=100 + i, scale=1 / 16), rnd(no, seed=200 + i)
        raw = F.conv2d(x, w, b)
        b32 = torch.zeros(32); b32[:no] = b
        out = torch.full((B, no, H, W), float("nan"), device=dev)
        out2 = torch.full((B, no, H, W), float("nan"), device=dev) if act == 3 else None
        heads.append(dict(w_first=pc.weight, b_first=pc.bias[:256].contiguous(), w_hidden=wh, b_hidden=bh,
                          w_out=(packing.pack_fragments16 if m16 else packing.pack_fragments)(w.view(no, 256)).to(dev), b_out=b32.to(dev),
                          w_out_perm=((packing.pack_fragments16 if m16 else packing.pack_fragments)(
                              w.view(no, 256), acc_order=True).to(dev) if patch else None),
                          mfma16=m16, n_out=no, act=act, out=out, out2=out2))
        refs.append(raw)
    f = ops.head_fused_args(srcs, [s.shape[-1] for s in srcs], slots, k_pad, B, H, W, heads)
    assert f.layout3x3 == int(patch) and f.mfma16 == int(m16)
    if m16:
        # the 128-pixel tile lies flat (8 x 16) or stands upright (16 x 8), whichever needs fewer tiles: force both,
        # the results must not differ by a bit
        firsts = None
        for tile in ("0", "1"):
            os.environ["CF_HEAD_TILE"] = tile
            try:
                for hd in heads:
                    hd["out"].fill_(float("nan"))
                ops.run_head_fused(f)
            finally:
                del os.environ["CF_HEAD_TILE"]
            outs = [hd["out"].clone() for hd in heads] + [hd["out2"].clone() for hd in heads if hd["out2"] is not None]
            if firsts is None:
                firsts = outs
            else:
                assert all(torch.equal(a, b) for a, b in zip(firsts, outs))
    ops.run_head_fused(f)
    for hd, raw in zip(heads, refs):
        scale = float(raw.abs().max())
        got = hd["out"].cpu()
        assert bool(torch.isfinite(got).all())
        if hd["act"] == 2:
            close(got, torch.clamp(torch.sigmoid(raw), 1e-4, 1 - 1e-4), 1e-4, 1e-5)
        else:
            assert float((got - raw).abs().max()) < 6e-5 * scale * (2 + n_hidden), float((got - raw).abs().max()) / scale
        if hd["act"] == 3:
            close(hd["out2"], 1.0 / (torch.sigmoid(raw) + 1e-6) - 1.0, 1e-3, 1e-3)


def test_head_fused_refuses_mfma16_fragments_it_cannot_read(dev):
    """16x16x32-packed weights are only readable by the 3x3 patch kernels: (a) a launch that would fall to the
    slot-table kernel (no w_out_perm -> layout3x3 = 0) with mfma16 set, and (b) a 16x16x32 head with more than 16
    outputs - with or without hidden layers - are refused, not run."""
    from centerfusiondetect3d_amd import ops, packing, _lib
    B, H, W = 1, 8, 16
    feat = rnd(B, 64, H, W, seed=1)
    srcs = [_split(feat, dev)]
    w1, b1 = rnd(256, 64, 3, 3, seed=2, scale=1 / 24), rnd(256, seed=3)
    pc = packing.pack_conv_bf16(w1, b1, [packing.Source(64, 64)], fragments=16).to(dev)

    def head(n_out, n_hidden, perm=True):
        w = rnd(n_out, 256, 1, 1, seed=4, scale=1 / 16)
        wp = torch.zeros(max(n_out, 16), 256)
        wp[:n_out] = w.view(n_out, 256)
        wh = [packing.pack_fragments16(rnd(256, 256, seed=5 + l, scale=1 / 16)).to(dev) for l in range(n_hidden)]
        return dict(w_first=pc.weight, b_first=pc.bias[:256].contiguous(), w_hidden=wh,
                    b_hidden=[torch.zeros(256, device=dev)] * n_hidden, w_out=packing.pack_fragments16(wp[:16]).to(dev),
                    b_out=torch.zeros(32, device=dev), w_out_perm=packing.pack_fragments16(wp[:16], acc_order=True).to(dev) if perm else None,
                    mfma16=True, n_out=n_out, act=0, out=torch.empty(B, n_out, H, W, device=dev), out2=None)

    ok = ops.head_fused_args(srcs, [64], pc.slots, pc.k_pad, B, H, W, [head(8, 0)])
    ops.run_head_fused(ok)                                           # the accepted form runs
    f = ops.head_fused_args(srcs, [64], pc.slots, pc.k_pad, B, H, W, [head(8, 0, perm=False)])
    assert f.layout3x3 == 0 and f.mfma16 == 1
    with pytest.raises(_lib.CfHipError, match="mfma16"):
        ops.run_head_fused(f)
    for n_hidden in (0, 2):
        f = ops.head_fused_args(srcs, [64], pc.slots, pc.k_pad, B, H, W, [head(24, n_hidden)])
        with pytest.raises(_lib.CfHipError, match="n_out=24"):
            ops.run_head_fused(f)


# ----------------------------------------------------------------------------------- postProcess
@pytest.mark.parametrize("seed", [0, 1])
def test_post_process_vs_reference_golden(dev, golden_dir, seed):
    from centerfusiondetect3d_amd import fusionDecode, postProcess
    from tests.test_oracle_golden import postprocess_inputs
    g = np.load(os.path.join(golden_dir, f"postprocess_{seed}.npz"))
    out, calibs = postprocess_inputs(seed)
    det = fusionDecode([{k: v.to(dev) for k, v in out.items()}], outputSize=(112, 200), K=100)
    pp = postProcess(det, np.array([800.0, 450.0], np.float32), 1600.0, 112, 200, calibs.to(dev))
    assert set(pp.keys()) == set(g.files)
    for k in g.files:
        got, ref = pp[k].cpu().numpy(), g[k]
        assert got.shape == ref.shape, k
        if k in ("scores", "classIds", "dimension", "amodal_offset", "nuscenes_att", "depth"):
            assert np.array_equal(got, ref), k                         # pass-through fields: exact
        else:                                                          # trig / affine arithmetic
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-4, err_msg=k)
    if seed == 1:
        bad = (g["dimension"] <= 0).any(-1)
        assert bad.any() and not pp["bboxes3d"].cpu().numpy()[bad].any()


# ----------------------------------------------------------------------- f16x3 (fp32 storage)
@pytest.mark.parametrize("B,Ci,Co,H,W,k,stride,act,res", [
    (2, 16, 32, 40, 56, 3, 2, 1, False),     # level1: N_pad 32 (RT=1)
    (1, 64, 64, 28, 50, 3, 1, 1, True),      # level2 block conv2 + residual, (1,4) waves
    (2, 64, 128, 30, 26, 3, 2, 1, False),    # (2,2) waves, stride 2, ragged M
    (2, 128, 256, 14, 25, 3, 1, 1, True),    # (4,1) waves
    (1, 256, 512, 14, 25, 3, 2, 1, False),   # two channel blocks
    (3, 64, 27, 23, 31, 3, 1, 0, False),     # conv_offset_mask (N=27), ragged M
])
def test_conv2d_f16x3_fp32_level_accuracy(dev, B, Ci, Co, H, W, k, stride, act, res):
    from centerfusiondetect3d_amd import ops, packing
    x, w, b = F.relu(rnd(B, Ci, H, W, seed=1)) * 3, rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5), rnd(Co, seed=3)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, k // 2)
    r = rnd(*ref.shape, seed=6) if res else None
    if res:
        ref = ref + r.double()
    if act:
        ref = F.relu(ref)
    pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)], stride=stride).to(dev)
    stride_out = 32 if Co == 27 else Co
    out = torch.zeros(B, ref.shape[2], ref.shape[3], stride_out, device=dev)
    ops.conv2d_f16x3(pc, [nhwc(x).to(dev)], B, H, W, act=act, residual=nhwc(r).to(dev) if res else None, out=out)
    got = nchw(out[..., :Co]).cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    ref32 = F.conv2d(x, w, b, stride, k // 2)
    if res:
        ref32 = ref32 + r
    if act:
        ref32 = F.relu(ref32)
    err32 = float((ref32.double() - ref).abs().max() / ref.abs().max())
    print(f"[f16x3] K={Ci * k * k}: max|err|/max|ref| = {err:.2e} (torch fp32 conv: {err32:.2e})")
    assert err < 1.5e-6, err


@pytest.mark.parametrize("B,Ci,Co,H,W,act,res,exact", [
    (3, 64, 27, 23, 31, 0, False, False),     # conv_offset_mask, small map: K split over the 4 waves
    (16, 64, 27, 112, 200, 0, False, True),   # conv_offset_mask at the bench size: 256-pixel runs
    (2, 512, 27, 14, 25, 0, False, False),    # ida_0.proj_1 offsets: 32 slices, WK = 4
    (2, 96, 27, 20, 30, 0, False, False),     # 6 slices: WK = 2
    (2, 48, 27, 9, 11, 0, False, True),       # 3 slices (odd k-step count, padded K): WK = 1
    (1, 64, 64, 28, 50, 1, True, True),       # level2 block conv2 + residual
    (2, 128, 128, 30, 26, 1, False, True),    # level3
    (2, 256, 256, 14, 25, 1, True, False),    # small map: K split over wave pairs
    (1, 512, 512, 7, 13, 1, False, False),    # level5: 128-channel blocks, K split over wave pairs
    (1, 64, 64, 5, 300, 1, False, True),      # too wide for the LDS patch: forwarded to the slot kernel
    (2, 16, 27, 1, 7, 0, False, True),        # single-row image (every dy != 0 tap is outside)
    (1, 64, 64, 112, 200, 1, True, True),     # level2 at the bench size: 16x16 tiles of one image (2-D patch)
    (2, 32, 27, 127, 127, 0, False, True),    # 2-D patch, last tile row / column one pixel short
    (1, 64, 64, 127, 126, 1, False, True),
    (16, 256, 256, 28, 50, 1, True, True),    # level4 at the bench size: 8 waves, 128-pixel runs
    (16, 128, 128, 56, 100, 1, True, True),   # level3 at the bench size: 8 waves, 256-pixel runs
    (1, 128, 128, 112, 200, 1, True, True),   # level3 of a 3x896x1600 input: 8 x 16 tiles (the flat patch does not fit)
    (2, 128, 128, 37, 150, 1, False, True),   # ... ragged tiles, tiled form as the fallback of the flat one
    (1, 256, 256, 56, 100, 1, True, True),    # level4 of a 3x896x1600 input: 4 x 16 tiles, 4 waves
    (8, 256, 256, 56, 100, 1, False, True),   # ... 8 waves (>= 160 tiles of 128 pixels), 8 x 16 tiles
    (1, 512, 512, 30, 110, 1, False, True),   # wide 512-channel map
])
def test_conv3x3_f16x3_patch(dev, B, Ci, Co, H, W, act, res, exact):
    """LDS-patch 3x3 kernel: fp32-level accuracy against float64, and - where the K loop is not split
    over waves - the very same bits as the generic slot kernel run on the same packed weights."""
    from centerfusiondetect3d_amd import ops, packing
    x, w, b = F.relu(rnd(B, Ci, H, W, seed=1)) * 3, rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5), rnd(Co, seed=3)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    r = rnd(*ref.shape, seed=6) if res else None
    if res:
        ref = ref + r.double()
    if act:
        ref = F.relu(ref)
    pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)]).to(dev)
    assert pc.patch
    stride_out = 32 if Co == 27 else Co
    xd, rd = nhwc(x).to(dev), (nhwc(r).to(dev) if res else None)
    out = torch.full((B, H, W, stride_out), float("nan"), device=dev)
    ops.conv2d_f16x3(pc, [xd], B, H, W, act=act, residual=rd, out=out, patch=True)
    got = nchw(out[..., :Co]).cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"[conv3x3 patch] {Ci}->{Co} {H}x{W}: max|err|/max|ref| = {err:.2e}")
    assert err < 1.5e-6, err
    if Co == 27:
        assert torch.isnan(out[..., 27:]).all()          # the padding channels are never written
    if exact:
        out2 = torch.zeros_like(out)
        ops.conv2d_f16x3(pc, [xd], B, H, W, act=act, residual=rd, out=out2, patch=False)
        assert torch.equal(out[..., :Co], out2[..., :Co])


@pytest.mark.parametrize("B,Ci,Co,H,W", [
    (2, 32, 64, 224, 400),     # level2.tree1.conv1 at the bench size: 8 x 16 output tiles, single-buffered patch
    (2, 64, 128, 112, 200),    # level3: 4 x 16 tiles
    (2, 128, 256, 56, 100),    # level4: output 28 x 50 (last tile column 2 of 16 wide)
    (3, 256, 512, 28, 50),     # level5: output 14 x 25, two channel blocks
    (1, 32, 64, 45, 67),       # odd input sizes: output 23 x 34, the last input row / column is a tap of the last output
    (3, 64, 128, 17, 31),      # output 9 x 16
    (1, 128, 256, 9, 130),     # output 5 x 65
    (2, 16, 64, 8, 6),         # one slice, map smaller than a tile
])
def test_conv3x3_f16x3_stride2_patch(dev, B, Ci, Co, H, W):
    """Stride-2 form of the LDS-patch kernel (odd / even column planes): fp32-level accuracy against float64 and the very
    same bits as the generic slot kernel on the same slice-major weights."""
    from centerfusiondetect3d_amd import ops, packing
    x, w, b = F.relu(rnd(B, Ci, H, W, seed=1)) * 3, rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5), rnd(Co, seed=3)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1))
    pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)], stride=2).to(dev)
    assert pc.patch and pc.stride == 2
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd = nhwc(x).to(dev)
    out = torch.full((B, Ho, Wo, Co), float("nan"), device=dev)
    ops.conv2d_f16x3(pc, [xd], B, H, W, act=1, out=out, patch=True)
    got = nchw(out).cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"[conv3x3 stride 2 patch] {Ci}->{Co} {H}x{W}: max|err|/max|ref| = {err:.2e}")
    assert err < 1.5e-6, err
    out2 = torch.zeros_like(out)
    ops.conv2d_f16x3(pc, [xd], B, H, W, act=1, out=out2, patch=False)
    assert torch.equal(out, out2)


def test_conv3x3_f16x3_stride2_with_residual_runs_the_slot_kernel(dev):
    """The stride-2 tile has no residual epilogue: such a call is forwarded to the slot kernel (same weights), not refused."""
    from centerfusiondetect3d_amd import ops, packing
    B, Ci, Co, H, W = 2, 32, 64, 30, 44
    x, w, b = rnd(B, Ci, H, W, seed=1), rnd(Co, Ci, 3, 3, seed=2, scale=(Ci * 9) ** -0.5), rnd(Co, seed=3)
    r = rnd(B, Co, 15, 22, seed=4)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1) + r.double())
    pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)], stride=2).to(dev)
    out = ops.conv2d_f16x3(pc, [nhwc(x).to(dev)], B, H, W, act=1, residual=nhwc(r).to(dev), patch=True)
    err = float((nchw(out).cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 1.5e-6, err


@pytest.mark.parametrize("C_,B,H,W,kids,fused", [
    (64, 2, 112, 200, (), True),        # level 2 at the bench size: 16 x 16 tiles of one image
    (64, 24, 37, 41, (), True),         # flat 256-pixel runs, ragged last run, image borders inside a run
    (64, 2, 128, 160, (), True),        # tiled, last tile column / row partly outside
    (64, 20, 5, 300, (), False),        # too wide for the flat LDS patch, too small a map for the tiled one: two launches
    (64, 1, 5, 300, (), True),          # ... the same map in a small launch: half tiles (8 x 16), fused
    (64, 1, 28, 50, (), True),          # a launch small enough for the one-round half tiles: those, fused
    (64, 1, 112, 200, (), True),        # ... level 2 of ONE frame (175 half tiles)
    (256, 1, 28, 50, (128, 256), True), # ... level4.tree2 of one frame, with children
    (128, 1, 56, 100, (), False),       # ... 128 channels: the half tiles hold 64 channels per workgroup - two launches
    (128, 8, 56, 100, (), True),        # level3.tree1: two waves per pixel group exchange their 64-channel pieces
    (128, 2, 112, 200, (), True),       # ... tiled form (level 3 of a high-resolution input)
    (128, 12, 37, 41, (), True),        # ... ragged
    (256, 8, 28, 50, (), True),         # level4.tree1: four waves per pixel group
    (256, 16, 31, 23, (), True),
    (256, 1, 14, 25, (), False),        # small map: K-split tiles, two launches
    (128, 8, 56, 100, (64, 128), True),     # level3.tree2: children = [pooled level-2 map, tree1's output], K = 448
    (256, 8, 28, 50, (128, 256), True),     # level4.tree2: K = 896, six 64-channel pieces in two rounds
    (64, 2, 112, 200, (64,), True),         # one child, one wave per pixel group
    (128, 12, 37, 41, (192,), True),        # three pieces for two waves: a round with one piece
    (128, 8, 56, 100, (32,), False),        # a child that is not a multiple of 64 channels: two launches
])
def test_conv3x3_root_fused_equals_two_launches(dev, C_, B, H, W, kids, fused):
    """cf_conv3x3_root_f16x3: tree2.conv2 (+ x1, ReLU) and the Tree's Root over (x2, x1, *children) in one launch - the same
    bits as cf_conv3x3_f16x3 followed by cf_conv2d_f16x3 on the same packed weights, x2 not written when fused, and
    fp32-level accuracy against float64."""
    from centerfusiondetect3d_amd import ops, packing
    t, x1 = F.relu(rnd(B, C_, H, W, seed=1)) * 2, F.relu(rnd(B, C_, H, W, seed=2)) * 2
    ch = [F.relu(rnd(B, c, H, W, seed=10 + i)) * 2 for i, c in enumerate(kids)]
    K = 2 * C_ + sum(kids)
    w2, b2 = rnd(C_, C_, 3, 3, seed=3, scale=(C_ * 9) ** -0.5), rnd(C_, seed=4)
    wr, br = rnd(C_, K, 1, 1, seed=5, scale=K ** -0.5), rnd(C_, seed=6)
    x2_ref = F.relu(F.conv2d(t.double(), w2.double(), b2.double(), 1, 1) + x1.double())
    ref = F.relu(F.conv2d(torch.cat([x2_ref, x1.double()] + [c.double() for c in ch], 1), wr.double(), br.double()))
    pc2 = packing.pack_conv_f16(w2, b2, [packing.Source(C_, C_)]).to(dev)
    pcr = packing.pack_conv_f16(wr, br, [packing.Source(C_, C_), packing.Source(C_, C_)] + [packing.Source(c, c) for c in kids]).to(dev)
    assert pc2.patch and pcr.k_pad == K
    td, x1d, chd = nhwc(t).to(dev), nhwc(x1).to(dev), [nhwc(c).to(dev) for c in ch]
    x2_buf = torch.full((B, H, W, C_), float("nan"), device=dev)
    out, _ = ops.conv3x3_root_f16x3(pc2, pcr, td, x1d, chd, x2_out=x2_buf)
    x2_two = ops.conv2d_f16x3(pc2, [td], B, H, W, act=1, residual=x1d)
    out_two = ops.conv2d_f16x3(pcr, [x2_two, x1d, *chd], B, H, W, act=1)
    assert torch.equal(out, out_two)
    if fused:
        assert torch.isnan(x2_buf).all()                 # x2 never left the chip
    else:
        assert torch.equal(x2_buf, x2_two)
    err = float((nchw(out).cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err


@pytest.mark.parametrize("B,Cp,C_,H,W,exact", [
    (2, 32, 64, 112, 200, True),        # level2.tree1.conv2 + level2.project at the bench size: tiled, a 32-channel (half) piece
    (8, 64, 128, 56, 100, True),        # level3.tree1: flat 128-pixel runs, two waves per pixel group share one piece
    (8, 128, 256, 28, 50, True),        # level4.tree1: four waves per pixel group, two pieces
    (8, 256, 512, 14, 25, False),       # level5: K split over wave pairs, four pieces in one round
    (16, 128, 256, 28, 50, True),       # ... 8 waves (two pixel groups per channel group)
    (1, 32, 64, 112, 200, True),        # one frame: the one-round half tiles (32 pixels per wave)
    (1, 64, 128, 56, 100, True),
    (1, 128, 256, 28, 50, True),
    (1, 256, 512, 14, 25, False),
    (12, 64, 128, 37, 41, True),        # ragged last run, image borders inside a run
    (3, 96, 64, 45, 67, True),          # one and a half pieces for one wave
    (2, 320, 256, 31, 23, True),        # five pieces for four waves: a second round with one piece
    (2, 64, 128, 112, 200, True),       # level 3 of a high-resolution input: 8 x 16 tiles
    (20, 32, 64, 5, 300, True),         # too wide for the patch: the slot kernel runs the same table
])
def test_conv3x3_proj_one_launch(dev, B, Cp, C_, H, W, exact):
    """cf_conv3x3_proj_f16x3: BasicBlock conv2 and the Tree's project (1x1 convolution of the pooled level input, the block's
    residual in the reference: dla.py:96-107, 56-62) summed in the same accumulators - fp32-level accuracy against float64,
    as close to it as the two launches it replaces, and (where the K loop is not split over waves) the very bits of the
    slot kernel run on the same slot table."""
    from centerfusiondetect3d_amd import ops, packing
    t, pooled = F.relu(rnd(B, C_, H, W, seed=1)) * 2, F.relu(rnd(B, Cp, H, W, seed=2)) * 2
    w2, b2 = rnd(C_, C_, 3, 3, seed=3, scale=(C_ * 9) ** -0.5), rnd(C_, seed=4)
    wp, bp = rnd(C_, Cp, 1, 1, seed=5, scale=Cp ** -0.5), rnd(C_, seed=6)
    ref = F.relu(F.conv2d(t.double(), w2.double(), b2.double(), 1, 1) + F.conv2d(pooled.double(), wp.double(), bp.double()))
    pc = packing.pack_conv_f16(w2, b2, [packing.Source(C_, C_)], proj=(wp, bp, packing.Source(Cp, Cp))).to(dev)
    assert pc.patch and pc.proj_k == Cp and pc.k_pad == 9 * C_ + Cp and pc.real_cin == (C_, Cp)
    td, pd = nhwc(t).to(dev), nhwc(pooled).to(dev)
    out = ops.conv3x3_proj_f16x3(pc, td, pd)
    err = float((nchw(out).cpu().double() - ref).abs().max() / ref.abs().max())
    # the two launches of the unfused plan: project -> residual tensor -> conv2 + residual + ReLU
    pcp = packing.pack_conv_f16(wp, bp, [packing.Source(Cp, Cp)]).to(dev)
    pc2 = packing.pack_conv_f16(w2, b2, [packing.Source(C_, C_)]).to(dev)
    res = ops.conv2d_f16x3(pcp, [pd], B, H, W, act=0)
    two = ops.conv2d_f16x3(pc2, [td], B, H, W, act=1, residual=res)
    err2 = float((nchw(two).cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"[conv3x3+proj] {C_}+{Cp} {H}x{W}: max|err|/max|ref| = {err:.2e} (two launches {err2:.2e})")
    assert err < 1.5e-6 and err < 2 * err2 + 2e-7, (err, err2)
    if exact:
        slot = ops.conv2d_f16x3(pc, [td, pd], B, H, W, act=1, patch=False)
        assert torch.equal(out, slot)


def test_conv3x3_proj_refuses_what_it_cannot_run(dev):
    from centerfusiondetect3d_amd import ops, packing, _lib
    w2, b2, wp, bp = rnd(64, 64, 3, 3, seed=1), rnd(64, seed=2), rnd(64, 32, 1, 1, seed=3), rnd(64, seed=4)
    pc = packing.pack_conv_f16(w2, b2, [packing.Source(64, 64)], proj=(wp, bp, packing.Source(32, 32))).to(dev)
    plain = packing.pack_conv_f16(w2, b2, [packing.Source(64, 64)]).to(dev)
    t, pd = torch.zeros(1, 8, 8, 64, device=dev), torch.zeros(1, 8, 8, 32, device=dev)
    with pytest.raises(AssertionError):
        ops.conv3x3_proj_f16x3(plain, t, pd)                    # weights without a projection part
    a = ops.conv_args(pc, [t, pd], [64, 32], 1, 8, 8, torch.empty(1, 8, 8, 64, device=dev), 64, 1, None, 0, 0, None, 0, False)
    import ctypes as C
    bad = (C.c_int32 * 2)(64, 64)                               # channel counts that do not add up to K_pad
    assert _lib.load().cf_conv3x3_proj_f16x3(C.byref(a), bad, None) != 0
    with pytest.raises(AssertionError):                         # a projection needs the slice-major stride-1 packing
        packing.pack_conv_f16(w2, b2, [packing.Source(64, 64)], stride=2, proj=(wp, bp, packing.Source(32, 32)))


def test_conv2d_f16x3_root_concat(dev):
    from centerfusiondetect3d_amd import ops, packing
    B, H, W = 2, 14, 25
    chans = [128, 128, 64, 128]
    xs = [rnd(B, c, H, W, seed=10 + i) for i, c in enumerate(chans)]
    w = rnd(128, sum(chans), 1, 1, seed=20, scale=sum(chans) ** -0.5)
    ref = F.relu(F.conv2d(torch.cat(xs, 1).double(), w.double()))
    pc = packing.pack_conv_f16(w, torch.zeros(128), [packing.Source(c, c) for c in chans]).to(dev)
    out = ops.conv2d_f16x3(pc, [nhwc(x).to(dev) for x in xs], B, H, W, act=1)
    assert float((nchw(out).cpu().double() - ref).abs().max() / ref.abs().max()) < 1.5e-6


@pytest.mark.parametrize("B,Ci,Co,H,W,mag", [(2, 64, 64, 28, 50, 2.0), (1, 128, 64, 14, 25, 8.0),
                                             (1, 512, 256, 7, 13, 1.0), (2, 256, 128, 9, 11, 30.0),
                                             (1, 128, 128, 20, 23, 3.0),
                                             # 64 outputs on maps without K split: ragged pixel tiles, offsets of
                                             # 0.5 / 1.5 / 12 px (far outside any local window), 48 of 64 output rows used
                                             (1, 64, 64, 37, 70, 0.5), (2, 128, 64, 33, 52, 1.5), (1, 64, 64, 48, 64, 12.0),
                                             (1, 32, 48, 50, 45, 1.0)])
def test_dcn_v2_f16x3(dev, B, Ci, Co, H, W, mag):
    from centerfusiondetect3d_amd import ops, packing
    x = rnd(B, Ci, H, W, seed=1)
    om = rnd(B, 27, H, W, seed=2)
    om[:, :18] *= mag
    w, b = rnd(Co, Ci, 3, 3, seed=3, scale=(Ci * 9) ** -0.5), rnd(Co, seed=4)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    ref = F.relu(dcn_ref.deform_conv2d(x.double(), torch.cat((o1, o2), 1).double(), w.double(), b.double(),
                                       (1, 1), (1, 1), (1, 1), torch.sigmoid(m.double())))
    pd = packing.pack_dcn_f16(w, b).to(dev)
    om32 = torch.zeros(B, H, W, 32)
    om32[..., :27] = nhwc(om)
    out = ops.dcn_v2_fused(pd, nhwc(x).to(dev), om32.to(dev))
    err = float((nchw(out).cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"[dcn f16x3] C={Ci}->{Co}: max|err|/max|ref| = {err:.2e}")
    assert err < 5e-6, err
    # optional second output: the same result as split-bf16 planes == cf_split_bf16 of the fp32 output
    split = torch.full((B, H, W, 2, Co), float("nan"), device=dev, dtype=torch.bfloat16)
    out2 = torch.empty_like(out)
    a = ops.dcn_args(pd, nhwc(x).to(dev), om32.to(dev), 32, B, H, W, out2, Co, out_split=split)
    ops.run_dcn(a)
    assert torch.equal(split, ops.split_bf16(out2))
    # (out used the K-split form on small maps, out2 - no workspace - did not: same value, other summation order)
    assert float((out2 - out).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert torch.equal(out2, ops.dcn_v2_fused(pd, nhwc(x).to(dev), om32.to(dev), k_split=False))


def test_dcn_v2_f16x3_rejects_odd_tile_counts_above_128(dev):
    """N_pad = 160 would give the two-row-tiles-per-wave kernel a tile past the packed weights: refused, not run."""
    from centerfusiondetect3d_amd import ops, packing, _lib
    B, Ci, Co, H, W = 1, 32, 160, 8, 8
    pd = packing.pack_dcn_f16(rnd(Co, Ci, 3, 3, seed=3), rnd(Co, seed=4)).to(dev)
    with pytest.raises(_lib.CfHipError):
        ops.dcn_v2_fused(pd, torch.zeros(B, H, W, Ci, device=dev), torch.zeros(B, H, W, 32, device=dev))


def test_dcn_v2_f16x3_checks_the_workspace_size(dev):
    """K is split on small maps only into a workspace of at least cf_dcn_v2_workspace_bytes(...)."""
    from centerfusiondetect3d_amd import ops, packing, _lib
    B, Ci, Co, H, W = 1, 128, 64, 14, 25
    pd = packing.pack_dcn_f16(rnd(Co, Ci, 3, 3, seed=3), rnd(Co, seed=4)).to(dev)
    x, om = torch.zeros(B, H, W, Ci, device=dev), torch.zeros(B, H, W, 32, device=dev)
    need = _lib.load().cf_dcn_v2_workspace_bytes(B, H, W, Ci, 64)
    assert need > 0
    out = torch.empty(B, H, W, Co, device=dev)
    small = torch.empty(need - 4, device=dev, dtype=torch.uint8)
    with pytest.raises(_lib.CfHipError):
        ops.run_dcn(ops.dcn_args(pd, x, om, 32, B, H, W, out, Co, workspace=small))
    ops.run_dcn(ops.dcn_args(pd, x, om, 32, B, H, W, out, Co, workspace=torch.empty(need, device=dev, dtype=torch.uint8)))


# ----------------------------------------------------------------------------------- fused stem
@pytest.mark.parametrize("B,C,H,W", [(2, 3, 64, 96), (1, 3, 16, 16), (1, 3, 34, 50), (3, 1, 18, 130), (1, 3, 160, 224), (1, 3, 448, 800)])
def test_stem_fused(dev, B, C, H, W):
    """base_layer 7x7 + level0 3x3 + level1 3x3/2 (+ bias + ReLU each) in one launch against float64
    torch: per-layer zero padding at the image border, ragged level1 tiles, fp32-level accuracy."""
    from centerfusiondetect3d_amd import ops, packing
    x = rnd(B, C, H, W, seed=1) * 2
    wb, bb = rnd(16, C, 7, 7, seed=2, scale=(C * 49) ** -0.5), rnd(16, seed=3, scale=0.3)
    w0, b0 = rnd(16, 16, 3, 3, seed=4, scale=144 ** -0.5), rnd(16, seed=5, scale=0.3)
    w1, b1 = rnd(32, 16, 3, 3, seed=6, scale=144 ** -0.5), rnd(32, seed=7, scale=0.3)
    t = F.relu(F.conv2d(x.double(), wb.double(), bb.double(), 1, 3))
    t = F.relu(F.conv2d(t, w0.double(), b0.double(), 1, 1))
    ref = F.relu(F.conv2d(t, w1.double(), b1.double(), 2, 1))
    ps = packing.pack_stem(wb, bb, w0, b0, w1, b1).to(dev)
    out = ops.stem_fused(ps, x.to(dev))
    assert out.shape == (B, H // 2, W // 2, 32)
    got = nchw(out).cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    t32 = F.relu(F.conv2d(F.relu(F.conv2d(F.relu(F.conv2d(x, wb, bb, 1, 3)), w0, b0, 1, 1)), w1, b1, 2, 1))
    err32 = float((t32.double() - ref).abs().max() / ref.abs().max())
    print(f"[stem] {B}x{C}x{H}x{W}: max|err|/max|ref| = {err:.2e} (torch fp32 chain: {err32:.2e})")
    assert err < 2e-6, err
    # the pooled second output (the level-2 Tree's MaxPool2d(2, 2) of this map, dla.py:96): the very values cf_maxpool2x2 /
    # torch give on the first output, floor semantics at odd sizes, nothing written outside; the first output unchanged
    pool = torch.full((B, H // 4, W // 4, 32), float("nan"), device=dev)
    out2 = ops.stem_fused(ps, x.to(dev), out_pool=pool)
    assert torch.equal(out2, out)
    assert torch.equal(nchw(pool), F.max_pool2d(nchw(out), 2, 2))


# ------------------------------------------------------------------------- image pre-processing
@pytest.mark.parametrize("B,Hs,Ws,inH,inW,M", [
    (2, 900, 1600, 448, 800, None),                                   # nuScenes: the detector's own matrix
    (1, 90, 160, 64, 96, [[0.61, 0.0, -2.3], [0.0, 0.61, 4.75]]),     # real bilinear blending + borders
    (3, 37, 53, 32, 64, [[1.13, 0.21, -3.5], [-0.17, 0.94, 6.25]]),   # rotation / shear, much of it outside
    (1, 16, 16, 16, 16, [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]),          # identity
])
def test_preprocess_images_bit_exact_vs_oracle(dev, B, Hs, Ws, inH, inW, M):
    from centerfusiondetect3d_amd import preProcessImages, getAffineTransform
    from centerfusiondetect3d_amd.preprocess import NUSCENES_MEAN, NUSCENES_STD
    from oracle import preprocess_ref
    frames = [np.random.RandomState(7 + i).randint(0, 256, size=(Hs, Ws, 3)).astype(np.uint8) for i in range(B)]
    Mh = np.array(M) if M is not None else getAffineTransform(np.array([Ws / 2.0, Hs / 2.0], np.float32),
                                                              max(Hs, Ws) * 1.0, 0, [inW, inH])
    ref = preprocess_ref.pre_process_images(frames, Mh, (inH, inW), NUSCENES_MEAN, NUSCENES_STD)
    got = preProcessImages(frames, (inH, inW), transMat=M, device=dev)
    assert got.shape == (B, 3, inH, inW) and got.dtype == torch.float32 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), ref)
    if M is None:   # 1600x900 -> 800x448 is an exact 2x decimation: rows 2y + 2, columns 2x
        raw = np.stack(frames)[:, 2:898:2, 0:1600:2].astype(np.float64)
        exp = ((raw / 255.0 - NUSCENES_MEAN) / NUSCENES_STD).astype(np.float32).transpose(0, 3, 1, 2)
        assert np.array_equal(ref, exp)


# ------------------------------------------------------------------------------- radar ingest
def _raw_sweep(rng, n, ties=False):
    pc = np.zeros((18, n))
    pc[2] = rng.uniform(-5.0, 80.0, n)
    if ties and n:
        pc[2] = np.round(pc[2] / 4.0) * 4.0                # many equal depths: order = original index
    pc[0] = rng.uniform(-0.8, 0.8, n) * np.abs(pc[2])
    pc[1] = rng.uniform(-2.0, 2.0, n)
    pc[3:] = rng.normal(0, 3, (15, n))
    return pc


@pytest.mark.parametrize("z_offset,descending,max_dist", [(0.0, False, 60.0), (0.4, True, 60.0), (0.0, False, 0.0)])
def test_radar_ingest_bit_exact_vs_oracle(dev, z_offset, descending, max_dist):
    from centerfusiondetect3d_amd import ops
    from oracle import radar_ref
    rng = np.random.RandomState(0)
    K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    sweeps = [_raw_sweep(rng, n, ties=(i % 2 == 1)) for i, n in enumerate((0, 1, 37, 250, 1024, 600))]
    B, max_n = len(sweeps), 1024
    pc = np.zeros((B, 18, max_n)); cnt = np.zeros(B, np.int32)
    for b, a in enumerate(sweeps):
        pc[b, :, :a.shape[1]], cnt[b] = a, a.shape[1]
    Ks = np.ascontiguousarray(np.broadcast_to(K, (B, 3, 3)))
    t = lambda a: torch.from_numpy(a).to(dev)
    p2, p3, c = ops.radar_ingest(t(pc), t(cnt), t(Ks), (1600, 900), max_dist, z_offset, descending)
    for b, a in enumerate(sweeps):
        r2, r3 = radar_ref.ingest_radar(a, K, (1600, 900), max_dist, z_offset, descending)
        m = r2.shape[1]
        assert int(c[b]) == m, (b, int(c[b]), m)
        assert np.array_equal(p2[b, :, :m].cpu().numpy(), r2) and np.array_equal(p3[b, :, :m].cpu().numpy(), r3)
        assert not p2[b, :, m:].any() and not p3[b, :, m:].any()


def test_radar_to_pc_dep_matches_the_oracle_chain(dev):
    """raw sweeps -> pc_dep entirely on the device == oracle ingest + oracle pillar expansion, bit for bit."""
    from centerfusiondetect3d_amd import radar_to_pc_dep
    from oracle import radar_ref
    rng = np.random.RandomState(1)
    K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    calib = np.concatenate([K, np.zeros((3, 1))], axis=1)
    out_hw, img_wh = (112, 200), (1600, 900)
    trans = pillar_ref.affine_transform_matrix((img_wh[0] / 2, img_wh[1] / 2), float(max(img_wh)), (out_hw[1], out_hw[0]))
    sweeps = [_raw_sweep(rng, n) for n in (180, 0, 420, 75)]
    got = radar_to_pc_dep(sweeps, K, img_wh, [calib] * 4, trans, out_hw, device=dev)
    assert got.shape == (4, 3, 112, 200)
    for b, a in enumerate(sweeps):
        r2, r3 = radar_ref.ingest_radar(a, K, img_wh, 60.0)
        _, _, dm = pillar_ref.process_point_cloud(r2, r3, calib, trans, out_hw)
        assert np.array_equal(got[b].cpu().numpy(), dm), f"frame {b}"
    assert int((got != 0).sum()) > 0


def test_shards_reproduce_the_batch_bit_for_bit_at_random_shapes(dev):
    """The tile shape of the 3x3 patch kernel and of the DCN kernel follows the launch size (half-height / 64-pixel tiles while
    the grid fits one round); the K order never does.  So every frame alone and every pair of frames must equal its slice of
    the full-batch result bit for bit - checked on 24 seeded random (B, H, W, Cin, Cout) around the dispatch
    thresholds (tools/sweep_shard_bits.py ran 480 such cases: none differed), with the full batch against float64."""
    from centerfusiondetect3d_amd import ops, packing
    rs = np.random.RandomState(11)
    done = 0
    while done < 24:
        B = int(rs.choice([2, 3, 4, 6]))
        H = int(rs.choice([7, 14, 28, 56, int(rs.randint(5, 90))]))
        W = int(rs.choice([13, 25, 50, 100, 200, int(rs.randint(5, 210))]))
        Ci, Co = int(rs.choice([32, 64, 128, 256])), int(rs.choice([27, 64, 128, 256]))
        if B * H * W * max(Ci, Co) > 6.0e7:
            continue
        done += 1
        x, w, b = F.relu(rnd(B, Ci, H, W, seed=done)) * 3, rnd(Co, Ci, 3, 3, seed=100 + done, scale=(Ci * 9) ** -0.5), rnd(Co, seed=200 + done)
        pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)]).to(dev)
        so = 32 if Co == 27 else Co
        xd = nhwc(x).to(dev)
        full = torch.zeros((B, H, W, so), device=dev)
        ops.conv2d_f16x3(pc, [xd], B, H, W, act=1, out=full, patch=True)
        ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
        tag = f"B={B} {Ci}->{Co} {H}x{W}"
        assert float((nchw(full[..., :Co]).cpu().double() - ref).abs().max() / ref.abs().max()) < 1.5e-6, tag
        shards = [(i, i + 1) for i in range(B)] + [(i, i + 2) for i in range(0, B - 1, 2)]
        for lo, hi in shards:
            part = torch.zeros((hi - lo, H, W, so), device=dev)
            ops.conv2d_f16x3(pc, [xd[lo:hi].contiguous()], hi - lo, H, W, act=1, out=part, patch=True)
            assert torch.equal(part[..., :Co], full[lo:hi, ..., :Co]), (tag, lo, hi)
        if Co != 27:
            om = torch.zeros(B, H, W, 32)
            om[..., :18] = rnd(B, H, W, 18, seed=300 + done) * 2.0
            om[..., 18:27] = rnd(B, H, W, 9, seed=400 + done)
            om = om.to(dev)
            pd = packing.pack_dcn_f16(w, b).to(dev)
            fulld = ops.dcn_v2_fused(pd, xd, om)
            for lo, hi in shards:
                partd = ops.dcn_v2_fused(pd, xd[lo:hi].contiguous(), om[lo:hi].contiguous())
                assert torch.equal(partd, fulld[lo:hi]), ("dcn", tag, lo, hi)
