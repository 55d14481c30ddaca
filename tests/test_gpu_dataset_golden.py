"""GPU parity (through the C ABI) of the dataset-side kernels against golden vectors produced by the
REFERENCE's own functions (tests/golden/make_golden_dataset.py): cf_pillar_expand and the
cf_radar_ingest -> cf_pillar_expand chain, bit for bit on the painted pc_dep maps."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PILLAR = sorted(glob.glob(os.path.join(GOLDEN, "pillar_*.npz")))
RADAR = sorted(glob.glob(os.path.join(GOLDEN, "radar_*.npz")))


def _name(p):
    return os.path.basename(p)[:-4]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X box"
    from centerfusiondetect3d_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("path", PILLAR, ids=_name)
def test_pillar_expand_bit_exact_vs_reference_golden(dev, path):
    from centerfusiondetect3d_amd import ops
    g = np.load(path)
    H, W = (int(v) for v in g["in_out_hw"])
    a, c = g["in_pc_2d"], g["in_pc_3d"]
    n = a.shape[1]
    max_n = max(1, n) + 3                                          # padding slots must be ignored
    p2 = np.zeros((1, 3, max_n)); p3 = np.zeros((1, 18, max_n))
    p2[0, :, :n], p3[0, :, :n] = a, c
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    pc_dep, keep, xy = ops.pillar_expand(t(p2), t(p3), t(np.array([n], np.int32)), t(g["in_calib"][None]),
                                         t(g["in_trans_out"][None]), (H, W), want_aux=True)
    assert np.array_equal(pc_dep[0].cpu().numpy(), g["out_depth_map"])
    kb = keep[0, :n].cpu().numpy().astype(bool)
    assert kb.sum() == g["out_pc_2d"].shape[1]
    assert np.array_equal(xy[0, :, :n].cpu().numpy()[:, kb], g["out_pc_2d"][:2])
    assert not keep[0, n:].any()


def test_pillar_expand_batched_fixtures_in_one_launch(dev):
    """All 112x200 fixtures as ONE batch (ragged counts, per-frame calib / transform)."""
    from centerfusiondetect3d_amd import ops
    gs = [np.load(p) for p in PILLAR]
    gs = [g for g in gs if tuple(g["in_out_hw"]) == (112, 200)]
    max_n = max(g["in_pc_2d"].shape[1] for g in gs)
    B = len(gs)
    p2 = np.zeros((B, 3, max_n)); p3 = np.zeros((B, 18, max_n)); cnt = np.zeros(B, np.int32)
    cal = np.zeros((B, 3, 4)); tr = np.zeros((B, 2, 3))
    for b, g in enumerate(gs):
        n = g["in_pc_2d"].shape[1]
        p2[b, :, :n], p3[b, :, :n], cnt[b], cal[b], tr[b] = g["in_pc_2d"], g["in_pc_3d"], n, g["in_calib"], g["in_trans_out"]
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    pc_dep = ops.pillar_expand(t(p2), t(p3), t(cnt), t(cal), t(tr), (112, 200))
    for b, g in enumerate(gs):
        assert np.array_equal(pc_dep[b].cpu().numpy(), g["out_depth_map"]), b


@pytest.mark.parametrize("path", RADAR, ids=_name)
def test_radar_chain_bit_exact_vs_reference_golden(dev, path):
    """Raw sweep -> cf_radar_ingest -> cf_pillar_expand == nuScenes.loadRadarPointCloud of the reference."""
    from centerfusiondetect3d_amd import ops, radar_to_pc_dep
    g = np.load(path)
    H, W = (int(v) for v in g["in_out_hw"])
    img_wh = tuple(int(v) for v in g["in_img_wh"])
    raw = g["in_radar_pc"]
    pc_dep = radar_to_pc_dep([raw], g["in_calib"][:, :3], img_wh, g["in_calib"][None], g["in_trans_out"], (H, W),
                             max_dist=60.0, z_offset=float(g["in_z_offset"]), descending=not bool(g["in_reverse"]),
                             device=dev)
    assert np.array_equal(pc_dep[0].cpu().numpy(), g["out_pc_dep"])
    # the ingest kernel on its own: kept set, order, rows
    n = raw.shape[1]
    max_n = max(1, n)
    pc = np.zeros((1, 18, max_n)); pc[0, :, :n] = raw
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    p2, p3, cnt = ops.radar_ingest(t(pc), t(np.array([n], np.int32)), t(g["in_calib"][None, :, :3].copy()), img_wh,
                                   60.0, float(g["in_z_offset"]), not bool(g["in_reverse"]))
    m = int(cnt[0])
    assert m >= int(g["out_pc_n"])              # (ingest counts before the output-map keep mask)
    # (the reference's list holds the points that also survive the output-map keep mask: compare as a subset in order)
    got3 = p3[0, :, :m].cpu().numpy()
    ref3 = g["out_pc_3d"]
    j = 0
    for i in range(m):
        if j < ref3.shape[1] and np.array_equal(got3[:, i], ref3[:, j]):
            j += 1
    assert j == ref3.shape[1]
