"""Known-answer tests of oracle/radar_ref.py (radar ingest; parity unpinned: nuscenes-devkit's
view_points is not in the image), plus consistency with the projection oracle/pillar_ref.py already uses."""
import numpy as np

from oracle import radar_ref, pillar_ref

K = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])


def test_hand_case():
    #            on axis   too far   behind   left of image   u just inside   ties with point 0
    pc = np.zeros((18, 6))
    pc[0] = [0.0, 0.0, 0.0, -40.0, -6.4, 1.0]
    pc[1] = [0.0, 0.0, 0.0, 0.0, 0.0, 0.5]
    pc[2] = [10.0, 61.0, -5.0, 50.0, 10.0, 10.0]
    pc[8] = [1, 2, 3, 4, 5, 6]
    p2, p3 = radar_ref.ingest_radar(pc, K, (1600, 900), 60.0)
    # kept: 0 (u = 816.3), 4 (u = 816.3 - 810.496 = 5.804 > 1), 5; dropped: 1 (far), 2 (behind), 3 (u < 1)
    assert p3[8].tolist() == [1, 5, 6]                      # depth ties keep their original order
    assert np.allclose(p2[0], [816.3, 816.3 - 1266.4 * 0.64, 816.3 + 126.64])
    assert p2[2].tolist() == [10.0, 10.0, 10.0] and np.isclose(p2[1, 2], 491.5 + 1266.4 * 0.05, rtol=1e-14)
    d2, d3 = radar_ref.ingest_radar(pc, K, (1600, 900), 60.0, descending=True)
    assert d3[8].tolist() == [6, 5, 1]
    z2, z3 = radar_ref.ingest_radar(pc, K, (1600, 900), 60.0, z_offset=0.5)
    assert z3[1].tolist() == [-0.5, -0.5, 0.0] and np.isclose(z2[1, 0], 491.5 + 1266.4 * -0.05, rtol=1e-14)


def test_agrees_with_the_pillar_oracles_projection_up_to_rounding():
    rng = np.random.RandomState(0)
    pc = np.zeros((18, 300))
    pc[2] = rng.uniform(-5, 80, 300)
    pc[0] = rng.uniform(-0.8, 0.8, 300) * np.abs(pc[2])
    pc[1] = rng.uniform(-2, 2, 300)
    p2, p3 = radar_ref.ingest_radar(pc, K)
    keep = pc[:, pc[2] <= 60.0]
    q2, mask = pillar_ref.map_pointcloud_to_image(keep, K)
    order = np.argsort(q2[2], kind="stable")
    assert p2.shape == q2.shape and np.array_equal(p3, keep[:, mask][:, order])
    np.testing.assert_allclose(p2, q2[:, order], rtol=1e-14)
    assert (np.diff(p2[2]) >= 0).all() and (p2[0] > 1).all() and (p2[0] < 1599).all()


def test_empty_and_all_rejected():
    p2, p3 = radar_ref.ingest_radar(np.zeros((18, 0)), K)
    assert p2.shape == (3, 0) and p3.shape == (18, 0)
    pc = np.zeros((18, 3)); pc[2] = [70.0, -1.0, 0.0]
    p2, p3 = radar_ref.ingest_radar(pc, K)
    assert p2.shape == (3, 0)
