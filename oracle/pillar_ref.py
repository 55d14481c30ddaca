"""Oracle: radar pillar expansion (radar points -> pc_dep map), numpy fp64.

TEST INFRASTRUCTURE - see oracle/__init__.py.  PINNED by golden vectors generated from the
reference's own Python (tests/golden/make_golden_dataset.py runs `processPointCloud` /
`getPcPillarsSize` / `drawPcHeat` of the imported reference on real `nuScenes` instances;
tests/test_oracle_dataset_golden.py compares this restatement with tests/golden/pillar_*.npz
bit for bit: kept points, pillar sizes, painted maps).  It follows

  * /root/reference/src/lib/dataset/generic_dataset.py:738-828   processPointCloud
  * generic_dataset.py:830-867                                   transformPointCloud
  * generic_dataset.py:869-942                                   getPcPillarsSize
  * /root/reference/src/lib/dataset/datasets/nuscenes.py:221-263 getDepthMap / drawPcHeat
  * /root/reference/src/lib/utils/pointcloud.py:17-49            map_pointcloud_to_image
  * pointcloud.py:239-296 (numpy branch), utils/ddd.py:8-55      get3DCorners / get3dBox / project3DPoints
  * /root/reference/src/lib/utils/image.py:43-83                 getAffineTransform (rotation 0)

Outside the pin (third-party arithmetic absent from /root/reference and the image, served by
stand-ins while the fixtures were generated): cv2.transform on float64 (restated as
m0*x + m1*y + m2 in fp64, OpenCV's `transform_<double>` order) and cv2.getAffineTransform (3-point
solve) - the fixtures carry the resulting 2x3 matrix as an INPUT.  Hand-computable cases and
property tests: tests/test_oracle_pillar.py.
"""
import numpy as np


def affine_transform_matrix(center, scale, out_wh):
    """getAffineTransform(center, scale, 0, [w, h]) for rotation 0, shift 0 (image.py:43-83).

    The reference builds three float32 point pairs and calls cv2.getAffineTransform (fp64
    solve of the 3-point system); restated with numpy's solver.
    """
    src_w = np.float32(scale)
    dst_w, dst_h = out_wh
    center = np.asarray(center, np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src_dir = np.array([0, src_w * -0.5], np.float32)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src[0] = center
    src[1] = center + src_dir
    dst[0] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32)
    dst[1] = dst_dir + dst[0]

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], np.float32)

    src[2] = third(src[0], src[1])
    dst[2] = third(dst[0], dst[1])
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst.astype(np.float64)).T.copy()      # (2, 3)


def map_pointcloud_to_image(pc, cam_intrinsic, img_shape=(1600, 900)):
    """pointcloud.py:17-49 with nuscenes view_points(normalize=True) restated (K @ p, / z)."""
    width, height = img_shape
    depths = pc[2, :]
    proj = np.asarray(cam_intrinsic, np.float64) @ pc[:3, :]
    pts = proj / proj[2:3, :]
    mask = (depths > 0) & (pts[0] > 1) & (pts[0] < width - 1) & (pts[1] > 1) & (pts[1] < height - 1)
    pts = pts[:, mask]
    pts[2, :] = depths[mask]
    return pts, mask


def transform_points(xy, m):
    """cv2.transform of (2,N) fp64 points by a 2x3 matrix."""
    x, y = xy[0], xy[1]
    return np.stack([m[0, 0] * x + m[0, 1] * y + m[0, 2], m[1, 0] * x + m[1, 1] * y + m[1, 2]])


def pillar_wh(pc_3d, calib, trans_out, pillar_dims=(1.5, 0.2, 0.2)):
    """getPcPillarsSize: (2,N) [w; h] of each pillar's projected bbox in output pixels."""
    N = pc_3d.shape[1]
    h, w, l = pillar_dims
    # get3DCorners numpy branch with yaw = 0: float32 corner offsets (pointcloud.py:268-288)
    xc = np.full(8, 0.5, np.float32); xc[2:4] *= -1; xc[6:8] *= -1; xc *= l
    yc = np.zeros(8, np.float32); yc[4:] = h * -1
    zc = np.full(8, 0.5, np.float32); zc[1:3] *= -1; zc[5:7] *= -1; zc *= w
    corners = np.stack([xc, yc, zc], axis=1).astype(np.float64)           # (8,3)
    pts = corners[None, :, :] + pc_3d[:3, :].T.astype(np.float64)[:, None, :]   # (N,8,3)
    c = np.asarray(calib, np.float64).reshape(3, 4)
    X, Y, Z = pts[..., 0], pts[..., 1], pts[..., 2]
    # project3DPoints: the reference's einsum sums the four products of a row left to right,
    # each rounded (pinned: pillar_kitti.npz has a non-zero fourth calib column)
    p2 = np.stack([((c[i, 0] * X + c[i, 1] * Y) + c[i, 2] * Z) + c[i, 3] * 1.0 for i in range(3)], axis=-1)
    uv = p2[..., :2] / p2[..., 2:]
    out = transform_points(uv.reshape(-1, 2).T, trans_out).T.reshape(N, 8, 2)
    return np.stack([out[..., 0].max(1) - out[..., 0].min(1),
                     out[..., 1].max(1) - out[..., 1].min(1)])


def process_point_cloud(pc_2d, pc_3d, calib, trans_out, out_hw=(112, 200),
                        pillar_dims=(1.5, 0.2, 0.2)):
    """processPointCloud for PC_ROI_METHOD == 'pillars', ONE_HOT_PC False.

    pc_2d (3,N) [u, v, depth] in original-image pixels, already depth-sorted ascending;
    pc_3d (>=10,N) camera-frame radar rows (8 = vx, 9 = vz).  Returns
    (transformed (3,M), pc_3d masked (.,M), depth map (3,H,W) float32).
    """
    H, W = out_hw
    depth_map = np.zeros((3, H, W), np.float32)
    if pc_2d.shape[1] == 0:
        return pc_2d, pc_3d, depth_map
    t = transform_points(pc_2d[:2], trans_out)
    mask = (t[0] < W) & (t[1] < H) & (0 < t[0]) & (0 < t[1])
    tp = np.concatenate([t[:, mask], pc_2d[2:, mask]], axis=0)
    pc_3d = pc_3d[:, mask]
    wh = pillar_wh(pc_3d, calib, trans_out, pillar_dims)
    for i in range(tp.shape[1]):
        cx, cy, depth = tp[0, i], tp[1, i], tp[2, i]
        box = [max(cy - wh[1, i], 0), cy, max(cx - wh[0, i] / 2, 0), min(cx + wh[0, i] / 2, W)]
        box = np.round(box).astype(np.int32)
        depth_map[0, box[0]:box[1], box[2]:box[3]] = depth
        depth_map[1, box[0]:box[1], box[2]:box[3]] = pc_3d[8, i]
        depth_map[2, box[0]:box[1], box[2]:box[3]] = pc_3d[9, i]
    return tp, pc_3d, depth_map


def synth_radar(rng, n, intr=(1266.4, 816.3, 491.5), img_wh=(1600, 900), max_dist=60.0):
    """Synthetic radar sweep per SURVEY.md §8(d): returns (pc_2d, pc_3d, calib) ready for
    process_point_cloud (<=max_dist filter, image projection/border filter, ascending depth)."""
    f, cx, cy = intr
    z = rng.uniform(1.0, max_dist, n)
    x = rng.uniform(-0.6, 0.6, n) * z
    y = rng.uniform(-1.0, 1.0, n)
    pc = np.zeros((18, n))
    pc[0], pc[1], pc[2] = x, y, z
    pc[8] = rng.normal(0, 5, n)
    pc[9] = rng.normal(0, 5, n)
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    pc = pc[:, pc[2] <= max_dist]
    pc_2d, mask = map_pointcloud_to_image(pc, K, img_wh)
    pc_3d = pc[:, mask]
    order = np.argsort(pc_2d[2, :])
    calib = np.concatenate([K, np.zeros((3, 1))], axis=1)
    return pc_2d[:, order], pc_3d[:, order], calib
