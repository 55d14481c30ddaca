"""CPU: the dataset-side oracles (pillar expansion, radar ingest, nuScenes serialisation) against
golden vectors produced by the REFERENCE's own functions (tests/golden/make_golden_dataset.py:
`nuScenes.processPointCloud` / `getPcPillarsSize` / `loadRadarPointCloud` / `convert_eval_format`
run on real `nuScenes` instances of the imported reference).  Integer / index / painted-map results
are compared bit for bit; see each oracle's header for the third-party arithmetic outside the pin."""
import glob
import os

import numpy as np
import pytest

from oracle import pillar_ref, radar_ref, serialize_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PILLAR = sorted(glob.glob(os.path.join(GOLDEN, "pillar_*.npz")))
RADAR = sorted(glob.glob(os.path.join(GOLDEN, "radar_*.npz")))


def _name(p):
    return os.path.basename(p)[:-4]


def test_fixture_inventory():
    names = {_name(p) for p in PILLAR}
    assert {"pillar_empty", "pillar_one", "pillar_n37", "pillar_n200", "pillar_n600", "pillar_border",
            "pillar_halfeven", "pillar_overlap", "pillar_hires", "pillar_kitti"} <= names
    assert len(RADAR) >= 5


@pytest.mark.parametrize("path", PILLAR, ids=_name)
def test_pillar_expansion_matches_reference(path):
    g = np.load(path)
    H, W = (int(v) for v in g["in_out_hw"])
    tp, p3, dm = pillar_ref.process_point_cloud(g["in_pc_2d"], g["in_pc_3d"], g["in_calib"],
                                                g["in_trans_out"], (H, W))
    assert tp.shape == g["out_pc_2d"].shape and np.array_equal(tp, g["out_pc_2d"])      # kept set + coordinates
    assert p3.shape == g["out_pc_3d"].shape and np.array_equal(p3, g["out_pc_3d"])
    assert dm.dtype == np.float32 and np.array_equal(dm, g["out_depth_map"])            # painted map, bit for bit
    if "out_pillar_wh" in g:
        assert np.array_equal(pillar_ref.pillar_wh(p3, g["in_calib"], g["in_trans_out"]), g["out_pillar_wh"])
    # the 3-point solve that stands for cv2.getAffineTransform: same matrix from the oracle's helper
    m = pillar_ref.affine_transform_matrix(g["in_center"], float(g["in_scale"]), (W, H))
    assert np.array_equal(m, g["in_trans_out"])


def test_pillar_fixtures_exercise_the_edge_cases():
    g = np.load(os.path.join(GOLDEN, "pillar_border.npz"))
    assert g["out_pc_2d"].shape[1] < g["in_pc_2d"].shape[1]          # strict keep-mask dropped points
    g = np.load(os.path.join(GOLDEN, "pillar_halfeven.npz"))
    cy = g["out_pc_2d"][1]
    assert np.all(cy - np.floor(cy) == 0.5)                          # every pillar foot on a .5: rint decides
    ks = np.floor(cy).astype(int)
    assert (ks % 2 == 0).any() and (ks % 2 == 1).any()
    g = np.load(os.path.join(GOLDEN, "pillar_overlap.npz"))
    d = g["out_depth_map"][0]
    assert len(np.unique(d[d != 0])) >= 3                            # several depths survive in one column
    g = np.load(os.path.join(GOLDEN, "pillar_hires.npz"))
    assert g["out_depth_map"].shape == (3, 224, 400)
    g = np.load(os.path.join(GOLDEN, "pillar_kitti.npz"))
    assert np.all(g["in_calib"][:, 3] != 0)                          # 4th calib column takes part


@pytest.mark.parametrize("path", RADAR, ids=_name)
def test_radar_ingest_chain_matches_reference(path):
    g = np.load(path)
    H, W = (int(v) for v in g["in_out_hw"])
    p2, p3 = radar_ref.ingest_radar(g["in_radar_pc"], g["in_calib"][:, :3], tuple(int(v) for v in g["in_img_wh"]),
                                    60.0, float(g["in_z_offset"]), descending=not bool(g["in_reverse"]))
    tp, p3k, dm = pillar_ref.process_point_cloud(p2, p3, g["in_calib"], g["in_trans_out"], (H, W))
    assert tp.shape[1] == int(g["out_pc_n"])
    assert np.array_equal(p3k, g["out_pc_3d"])                       # kept set, order, every radar row
    assert np.array_equal(tp[2], g["out_pc_2d"][2])                  # depths
    if tp.shape[1]:
        # u, v: the reference hands the projection to BLAS (last bit undefined, oracle/radar_ref.py)
        np.testing.assert_allclose(tp[:2], g["out_pc_2d"][:2], rtol=1e-12, atol=0)
    assert np.array_equal(dm, g["out_pc_dep"])                       # pc_dep map, bit for bit


def test_serialisation_matches_reference():
    from tests.golden import cases_dataset as cd
    for name, case in cd.serialize_cases():
        g = np.load(os.path.join(GOLDEN, f"serialize_{name}.npz"))
        ret = serialize_ref.convert_eval_format(list(case["images"].keys()), case["images"], case["results"])
        assert sorted(ret.keys()) == [str(t) for t in g["tokens"]]
        n_cut = 0
        for t in ret:
            rows = ret[t]
            assert len(rows) == len(g[f"{t}_score"]) <= 500
            n_cut += len(rows) == 500
            if not rows:
                continue
            for key in ("translation", "size", "velocity", "rotation"):
                assert np.array_equal(np.array([r[key] for r in rows]), g[f"{t}_{key}"]), (t, key)
            assert np.array_equal(np.array([r["score"] for r in rows]), g[f"{t}_score"])
            assert [serialize_ref.CLASS_NAME[r["class_index"]] for r in rows] == list(g[f"{t}_name"])
            assert [serialize_ref.ID_TO_ATTRIBUTE[r["attribute"]] for r in rows] == list(g[f"{t}_attribute"])
            assert np.array_equal(np.array([r["sensor_id"] for r in rows]), g[f"{t}_sensor_id"])
        assert n_cut >= 1                                            # the top-500 cut was exercised


def test_dot4_order_is_numpys():
    rs = np.random.RandomState(5)
    for _ in range(300):
        M = (rs.standard_normal((4, 4)) * rs.choice([1, 100, 800])).astype(np.float32)
        v = (rs.standard_normal(4) * 30).astype(np.float32)
        assert np.array_equal(serialize_ref.dot4_f32(M, v), np.dot(M, v))


def test_box_rotation_known_answers():
    ident = [1.0, 0.0, 0.0, 0.0]
    np.testing.assert_allclose(serialize_ref.box_rotation(0.0, ident, ident), ident, atol=1e-15)
    q = serialize_ref.box_rotation(np.pi / 2, ident, ident)
    np.testing.assert_allclose(q, [np.sqrt(0.5), 0, np.sqrt(0.5), 0], atol=1e-15)
    # 90 deg about z after 90 deg about y:  (cos45 + k sin45)(cos45 + j sin45)
    qz = [np.sqrt(0.5), 0, 0, np.sqrt(0.5)]
    np.testing.assert_allclose(serialize_ref.box_rotation(np.pi / 2, qz, ident), [0.5, -0.5, 0.5, 0.5], atol=1e-15)
    np.testing.assert_allclose(serialize_ref.box_rotation(np.pi / 2, ident, qz), [0.5, -0.5, 0.5, 0.5], atol=1e-15)
