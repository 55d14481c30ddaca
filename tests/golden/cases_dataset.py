"""Deterministic inputs for the dataset-side golden vectors (make_golden_dataset.py feeds them to the
reference; the parity tests read them back from the committed .npz files, so this module is only
needed where the fixtures are generated).  numpy RandomState only; no reference code, no oracle code.
"""
import numpy as np

NUSC_K = np.array([[1266.417203046554, 0.0, 816.2670197447984],
                   [0.0, 1266.417203046554, 491.50706579294757],
                   [0.0, 0.0, 1.0]])


def _calib(K=NUSC_K):
    return np.concatenate([K, np.zeros((3, 1))], axis=1)


def _project_sorted(pc, K=NUSC_K, img_wh=(1600, 900)):
    """(R,N) camera-frame points -> (pc_2d (3,M), pc_3d (R,M)) inside the image, depth ascending."""
    z = pc[2]
    u = (K[0, 0] * pc[0] + K[0, 2] * z) / z
    v = (K[1, 1] * pc[1] + K[1, 2] * z) / z
    m = (z > 0) & (u > 1) & (u < img_wh[0] - 1) & (v > 1) & (v < img_wh[1] - 1)
    pc_2d = np.stack([u[m], v[m], z[m]])
    pc_3d = pc[:, m]
    o = np.argsort(pc_2d[2], kind="stable")
    return pc_2d[:, o], pc_3d[:, o]


def _sweep(rs, n, max_z=60.0, lateral=0.6, height=1.0, rows=18):
    z = rs.uniform(1.0, max_z, n)
    pc = np.zeros((rows, n))
    pc[0] = rs.uniform(-lateral, lateral, n) * z
    pc[1] = rs.uniform(-height, height, n)
    pc[2] = z
    pc[8] = rs.normal(0, 5, n)
    pc[9] = rs.normal(0, 5, n)
    for r in (3, 4, 5, 6, 7):
        pc[r] = rs.uniform(0, 4, n)
    return pc


def pillar_cases():
    """(name, dict(out_hw, calib (3,4), center (2,) f32, scale, pc_2d (3,N) f64, pc_3d (18,N) f64))."""
    base = dict(calib=_calib(), center=np.array([800.0, 450.0], np.float32), scale=1600.0, out_hw=(112, 200))
    out = []

    def add(name, pc_2d, pc_3d, **over):
        c = dict(base, **over)
        c["pc_2d"], c["pc_3d"] = np.ascontiguousarray(pc_2d, np.float64), np.ascontiguousarray(pc_3d, np.float64)
        out.append((name, c))

    add("empty", np.zeros((3, 0)), np.zeros((18, 0)))
    one = np.zeros((18, 1)); one[2] = 20.0; one[8], one[9] = 1.25, -3.5         # on the optical axis, 20 m
    add("one", *_project_sorted(one))
    for name, n, seed in (("n37", 37, 11), ("n200", 200, 12), ("n600", 600, 13)):
        add(name, *_project_sorted(_sweep(np.random.RandomState(seed), n)))
    # points hugging the image border: many transform to x<=0 / y<=0 / x>=W / y>=H and are dropped by the
    # strict keep-mask; very near points give pillars taller / wider than the map (clipping)
    rs = np.random.RandomState(14)
    pc = _sweep(rs, 160, lateral=0.66, height=2.2)
    pc[2, :24] = rs.uniform(0.6, 2.5, 24)
    pc[0, :24] = rs.uniform(-0.6, 0.6, 24) * pc[2, :24]
    p2, p3 = _project_sorted(pc)
    # force exact border coordinates in original-image pixels: u = 0 / 1600 -> x = 0 / 200 (excluded)
    p2[0, 0::9] = np.array([1600.0, 0.0, 8.0, 1592.0, 1599.999])[np.arange(len(p2[0, 0::9])) % 5]
    p2[1, 1::9] = np.array([2.0, 898.0, 1.999, 897.9999, 450.0])[np.arange(len(p2[1, 1::9])) % 5]
    add("border", p2, p3)
    # half-to-even: with out = in/8 (+ -0.25 in y) choose v = 8k + 6 -> cy = k + .5 exactly, u so that
    # cx +- w/2 is as close to .5 as the projection allows; mixed parities of k
    rs = np.random.RandomState(15)
    pc = _sweep(rs, 96)
    p2, p3 = _project_sorted(pc)
    k = rs.randint(4, 108, p2.shape[1])
    p2[1] = 8.0 * k + 6.0
    p2[0] = 8.0 * rs.randint(4, 196, p2.shape[1]) + 4.0 * rs.randint(0, 2, p2.shape[1])
    add("halfeven", p2, p3)
    # stacked pillars: same image column, increasing depth -> the farthest covering point wins
    pc = np.zeros((18, 40))
    pc[2] = np.linspace(4.0, 58.0, 40)
    pc[0] = 0.05 * pc[2] + np.tile([0.0, 0.02, -0.02, 0.01], 10)
    pc[1] = np.tile([0.3, -0.2, 0.6, 0.0, 0.9], 8)
    pc[8] = np.arange(40) * 0.5 - 7
    pc[9] = 9 - np.arange(40) * 0.25
    add("overlap", *_project_sorted(pc))
    # BASELINE config C5: 896x1600 input, 224x400 map
    add("hires", *_project_sorted(_sweep(np.random.RandomState(16), 150)), out_hw=(224, 400))
    # a non-nuScenes camera / image size (centre and scale follow detector.py:206-208)
    K2 = np.array([[721.5377, 0.0, 609.5593], [0.0, 721.5377, 172.854], [0.0, 0.0, 1.0]])
    pc = _sweep(np.random.RandomState(17), 80, lateral=0.7, height=0.8)
    p2, p3 = _project_sorted(pc, K2, (1242, 375))
    cal2 = _calib(K2); cal2[0, 3], cal2[1, 3], cal2[2, 3] = 44.85728, 0.2163791, 0.002745884
    add("kitti", p2, p3, calib=cal2, center=np.array([621.0, 187.5], np.float32), scale=1242.0,
        out_hw=(96, 320))
    return out


def radar_cases():
    """(name, dict(radar_pc (18,N) f64 raw sweep, calib, img_wh, out_hw, z_offset, reverse, center, scale))."""
    base = dict(calib=_calib(), img_wh=(1600, 900), out_hw=(112, 200), z_offset=0.0, reverse=True,
                center=np.array([800.0, 450.0], np.float32), scale=1600.0)
    out = []

    def raw(seed, n):
        rs = np.random.RandomState(seed)
        pc = _sweep(rs, n, max_z=75.0, lateral=0.8, height=1.5)       # some beyond 60 m, some outside the image
        pc[2, ::17] = -rs.uniform(0.5, 30.0, len(pc[2, ::17]))        # behind the camera
        pc[2, 5::23] = 0.0
        return pc

    out.append(("empty", dict(base, radar_pc=np.zeros((18, 0)))))
    out.append(("n150", dict(base, radar_pc=raw(21, 150))))
    out.append(("n300_zoff", dict(base, radar_pc=raw(22, 300), z_offset=0.4)))
    out.append(("n120_nearfirst", dict(base, radar_pc=raw(23, 120), reverse=False)))
    out.append(("allout", dict(base, radar_pc=np.concatenate([raw(24, 30)[:, :0], np.array(
        [[0.0] * 5, [0.0] * 5, [61.0, 70.0, -3.0, 0.0, 100.0]] + [[0.0] * 5] * 15)], axis=1))))
    return out


def serialize_cases():
    """(name, dict(images {image_id: info}, results {image_id: [det dict of numpy values]}))."""
    out = []

    def rot_trans(rs):
        q = rs.standard_normal(4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        M = np.eye(4); M[:3, :3] = R; M[:3, 3] = rs.uniform(-800, 800, 3)
        V = np.eye(4); V[:3, :3] = R
        return M, V

    def dets(rs, n, tie_every=0):
        d = []
        for i in range(n):
            score = np.float32(rs.uniform(0.01, 0.99))
            if tie_every and i % tie_every == 0:
                score = np.float32(0.5)
            d.append(dict(**{"class": np.float32(rs.randint(1, 11))}, score=score,
                          dimension=rs.uniform(0.3, 6.0, 3).astype(np.float32),
                          location=(rs.standard_normal(3) * [15, 1, 20] + [0, 1, 30]).astype(np.float32),
                          yaw=np.float32(rs.uniform(-np.pi, np.pi)),
                          nuscenes_att=rs.standard_normal(8).astype(np.float32),
                          velocity=(rs.standard_normal(3) * 4).astype(np.float32),
                          rotation=rs.standard_normal(4)))
        return d

    rs = np.random.RandomState(31)
    images, results = {}, {}
    # sample "a": 6 cameras x 100 boxes -> 600 merged, top-500 kept; sample "b": 2 cameras, few boxes,
    # score ties; image 99 has no results (skipped); sample "c": one camera, zero boxes
    iid = 0
    for tok, cams, n, tie in (("a", 6, 100, 0), ("b", 2, 7, 3), ("c", 1, 0, 0)):
        for c in range(cams):
            iid += 1
            M, V = rot_trans(rs)
            images[iid] = dict(sample_token=f"tok_{tok}", trans_matrix=M.tolist(),
                               velocity_trans_matrix=V.tolist(), sensor_id=c + 1)
            results[iid] = dets(rs, n, tie)
    M, V = rot_trans(rs)
    images[99] = dict(sample_token="tok_d", trans_matrix=M.tolist(), velocity_trans_matrix=V.tolist(), sensor_id=1)
    out.append(("mixed", dict(images=images, results=results)))
    return out
