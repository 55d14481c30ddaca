"""CPU: hand-computable cases + properties holding the pillar-expansion restatement
(oracle/pillar_ref.py; the reference module is not importable here - parity unpinned)."""
import numpy as np

from oracle import pillar_ref as P

F, CX, CY = 1266.4, 816.3, 491.5
CALIB = np.array([[F, 0, CX, 0], [0, F, CY, 0], [0, 0, 1.0, 0]])


def _trans():
    return P.affine_transform_matrix((800.0, 450.0), 1600.0, (200, 112))


def test_affine_matrix_closed_form():
    m = _trans()
    np.testing.assert_allclose(m, [[0.125, 0, 0.0], [0, 0.125, 56 - 0.125 * 450]], atol=1e-12)


def _one_point(x, y, z, vx=1.0, vz=2.0):
    pc3 = np.zeros((18, 1)); pc3[:3, 0] = (x, y, z); pc3[8, 0] = vx; pc3[9, 0] = vz
    pc2, mask = P.map_pointcloud_to_image(pc3, CALIB[:, :3])
    return pc2, pc3[:, mask]


def test_single_point_on_axis_pillar_size():
    d = 20.0
    pc2, pc3 = _one_point(0.0, 0.5, d)
    wh = P.pillar_wh(pc3, CALIB, _trans())
    # nearest pillar face is at z = d - 0.1: height 1.5*f/(d-0.1)/8 px, width 0.2*f/(d-0.1)/8 px
    np.testing.assert_allclose(wh[1, 0], 1.5 * F / (d - 0.1) * 0.125, rtol=1e-6)
    np.testing.assert_allclose(wh[0, 0], 0.2 * F / (d - 0.1) * 0.125, rtol=1e-6)
    tp, _, dm = P.process_point_cloud(pc2, pc3, CALIB, _trans())
    cx, cy = tp[0, 0], tp[1, 0]
    box = np.round([max(cy - wh[1, 0], 0), cy, max(cx - wh[0, 0] / 2, 0), min(cx + wh[0, 0] / 2, 200)]).astype(int)
    exp = np.zeros((112, 200), bool)
    exp[box[0]:box[1], box[2]:box[3]] = True
    assert exp.sum() > 0
    assert np.array_equal(dm[0] != 0, exp)
    assert np.all(dm[0][exp] == np.float32(d)) and np.all(dm[1][exp] == 1.0) and np.all(dm[2][exp] == 2.0)


def test_far_point_overwrites_near():
    pc3 = np.zeros((18, 2))
    pc3[:3, 0] = (0.0, 0.5, 10.0); pc3[:3, 1] = (0.0, 0.5, 30.0)
    pc3[8] = (1.0, 2.0)
    pc2, mask = P.map_pointcloud_to_image(pc3, CALIB[:, :3])
    _, _, dm = P.process_point_cloud(pc2, pc3[:, mask], CALIB, _trans())
    vals = set(np.unique(dm[0]).tolist())
    assert vals == {0.0, 10.0, 30.0}
    # the far pillar is smaller and fully inside the near one: it must win where it covers
    far = dm[0] == 30.0
    assert far.sum() > 0 and np.all(dm[1][far] == 2.0)


def test_round_half_to_even_and_strict_mask():
    trans = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    calib = np.array([[8.0, 0, 0, 0], [0, 8.0, 0, 0], [0, 0, 1.0, 0]])
    # point exactly on x == 0 or y == H is dropped (strict inequalities)
    pc2 = np.array([[0.0, 50.0, 50.0], [30.0, 112.0, 30.5], [10.0, 10.0, 10.0]])
    pc3 = np.zeros((18, 3)); pc3[0] = pc2[0] * 10 / 8; pc3[1] = pc2[1] * 10 / 8; pc3[2] = 10.0
    tp, p3, dm = P.process_point_cloud(pc2, pc3, calib, trans)
    assert tp.shape[1] == 1 and tp[1, 0] == 30.5
    rows = np.nonzero((dm[0] != 0).any(1))[0]
    # h = 1.5*8/9.9 = 1.2121 -> y1 = 29.2879 -> 29 ; y2 = round(30.5) = 30 (half to even)
    assert rows.tolist() == [29]


def test_empty_and_ragged():
    pc2, pc3 = np.zeros((3, 0)), np.zeros((18, 0))
    _, _, dm = P.process_point_cloud(pc2, pc3, CALIB, _trans())
    assert dm.shape == (3, 112, 200) and not dm.any()
    rng = np.random.default_rng(0)
    for n in (1, 7, 200):
        a, b, c = P.synth_radar(rng, n)
        tp, p3, dm = P.process_point_cloud(a, b, c, _trans())
        assert tp.shape[1] == p3.shape[1] <= n
        assert np.all(np.diff(tp[2]) >= 0)
        if tp.shape[1]:
            assert set(np.unique(dm[0]).tolist()) <= set(np.float32(tp[2]).tolist()) | {0.0}
