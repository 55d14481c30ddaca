"""Oracle: top-k peak extraction and radar frustum association, CPU fp32.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Follows

  * /root/reference/src/lib/model/utils.py:6-72        topk / getFeature / transposeAndGetFeature
  * /root/reference/src/lib/utils/pointcloud.py:195-211 get_alpha
  * pointcloud.py:214-236                               cvtAlphaToYaw
  * pointcloud.py:239-296                               get3DCorners
  * pointcloud.py:299-328                               getDistanceThresh  (max - min/2, sic)
  * pointcloud.py:331-394                               getPcFrustumHeatmap
  * pointcloud.py:397-481                               cvtPcDepthToHeatmap

Tie-break (SURVEY.md §8(a) note): torch.topk leaves the order of equal scores undefined;
this oracle - and the HIP kernels - order by (score desc, class asc, flat pixel index asc).
Pinned against the reference's own functions on tie-free inputs by tests/golden/frustum_*.npz.

All float arithmetic below is done in numpy float32 scalar ops in the same order as the
reference's torch expressions so integer slice bounds come out bit-identical.
"""
import math

import numpy as np
import torch

f32 = np.float32
PI32 = f32(math.pi)
TWO_PI32 = f32(2 * math.pi)


def topk(heatmap: torch.Tensor, K: int = 100):
    """(scores, inds, classes, ys, xs) as model/utils.py:6-38, deterministic on ties."""
    B, C, H, W = heatmap.shape
    flat = heatmap.reshape(B, C * H * W)
    scores, order = torch.sort(flat, dim=1, descending=True, stable=True)
    scores, order = scores[:, :K], order[:, :K]
    classes = (order // (H * W)).int()
    inds = order % (H * W)
    ys = inds // W
    xs = inds % W
    return scores, inds, classes, ys, xs


def gather_feat(fmap: torch.Tensor, inds: torch.Tensor):
    """transposeAndGetFeature: (B,C,H,W),(B,K) -> (B,K,C)."""
    B, C, H, W = fmap.shape
    f = fmap.reshape(B, C, H * W)
    return torch.gather(f, 2, inds.view(B, 1, -1).expand(B, C, inds.shape[1])).permute(0, 2, 1)


def get_alpha(rot):
    """rot: (8,) float32 -> alpha float32 (pointcloud.py:207-210)."""
    idx = f32(1.0) if rot[1] > rot[5] else f32(0.0)
    a1 = f32(np.arctan2(rot[2], rot[3])) + f32(-0.5 * math.pi)
    a2 = f32(np.arctan2(rot[6], rot[7])) + f32(0.5 * math.pi)
    return f32(f32(a1 * idx) + f32(a2 * f32(f32(1.0) - idx)))


def distance_thresh(calib, cx, dim, alpha):
    """getDistanceThresh for one box; calib (3,4) f32, dim = (h,w,l)."""
    yaw = f32(alpha + f32(np.arctan2(f32(cx - calib[0, 2]), calib[0, 0])))
    if yaw > PI32:
        yaw = f32(yaw - TWO_PI32)
    if yaw < -PI32:
        yaw = f32(yaw + TWO_PI32)
    c, s = f32(np.cos(yaw)), f32(np.sin(yaw))
    h, w, l = dim[0], dim[1], dim[2]
    xc = [f32(f32(0.5) * l) * sg for sg in (1, 1, -1, -1, 1, 1, -1, -1)]
    yc = [f32(0)] * 4 + [f32(-h)] * 4
    zc = [f32(f32(0.5) * w) * sg for sg in (1, -1, -1, 1, 1, -1, -1, 1)]
    zs = [f32(f32(f32(-s) * f32(x)) + f32(f32(0) * y) + f32(c * f32(z))) for x, y, z in zip(xc, yc, zc)]
    return f32(max(zs) - f32(min(zs) / f32(2.0)))


def _slice(start, stop, n):
    """Python slice normalisation for a dimension of size n (negatives wrap)."""
    return slice(start, stop).indices(n)[:2]


def paint_box(pc_hm, pc_dep, depth, bbox, thr, max_pc_dist):
    """cvtPcDepthToHeatmap on numpy float32 arrays (pc_hm, pc_dep: (3,H,W)); in-place."""
    _, H, W = pc_dep.shape
    b0, b1, b2, b3 = (f32(v) for v in bbox)
    cx = f32(f32(b0 + b2) / f32(2.0))
    cy = f32(f32(b1 + b3) / f32(2.0))
    x0, y0 = int(math.floor(b0)), int(math.floor(b1))
    x1, y1 = int(math.ceil(b2)), int(math.ceil(b3))
    ys, ye = _slice(y0, y1 + 1, H)
    xs, xe = _slice(x0, x1 + 1, W)
    if ye <= ys or xe <= xs:
        return
    roi = pc_dep[:, ys:ye, xs:xe]
    d = roi[0]
    nz = np.nonzero(d)
    if len(nz[0]) == 0:
        return
    dv = d[nz]
    hi = f32(depth + thr)
    t = f32(depth - thr)
    lo = t if t > 0 else f32(0)
    ok = (dv < hi) & (dv > lo)
    if not ok.any():
        return
    cand = np.nonzero(ok)[0]
    j = cand[np.argmin(dv[cand])]          # first occurrence of the minimum, row-major
    dist = f32(dv[j] / f32(max_pc_dist))
    vx = roi[1][nz][j]
    vz = roi[2][nz][j]
    w = f32(b2 - b0)
    wi = f32(f32(0.3) * w)
    w_min = int(f32(cx - f32(wi / f32(2.0))))
    w_max = int(f32(cx + f32(wi / f32(2.0))))
    h = f32(b3 - b1)
    hi_ = f32(f32(0.3) * h)
    h_min = int(f32(cy - f32(hi_ / f32(2.0))))
    h_max = int(f32(cy + f32(hi_ / f32(2.0))))
    pc_hm[0, h_min:h_max + 1, w_min:w_max + 2] = dist
    pc_hm[1, h_min:h_max + 1, w_min:w_max + 2] = vx
    pc_hm[2, h_min:h_max + 1, w_min:w_max + 2] = vz


def box_params(y, calib, K):
    """Per-box quantities feeding the paint loop: (inds, bboxes, depth, thresh) as numpy."""
    heat = y["heatmap"]
    B = heat.shape[0]
    _, inds, _, ys, xs = topk(heat, K)
    xs = xs.to(torch.float32) + 0.5
    ys = ys.to(torch.float32) + 0.5
    depth = gather_feat(y["depth"], inds)[..., 0].numpy().astype(f32)
    wh = gather_feat(y["widthHeight"], inds).clone()
    wh[wh < 0] = 0
    bboxes = torch.stack([xs - wh[..., 0] / 2, ys - wh[..., 1] / 2,
                          xs + wh[..., 0] / 2, ys + wh[..., 1] / 2], dim=2).numpy().astype(f32)
    dim = gather_feat(y["dimension"], inds).numpy().astype(f32)
    rot = gather_feat(y["rotation"], inds).numpy().astype(f32)
    cal = calib.reshape(B, 3, 4).numpy().astype(f32)
    thr = np.zeros((B, K), f32)
    for b in range(B):
        for i in range(K):
            alpha = get_alpha(rot[b, i])
            cx = f32(f32(bboxes[b, i, 0] + bboxes[b, i, 2]) / f32(2.0))
            thr[b, i] = distance_thresh(cal[b], cx, dim[b, i], alpha)
    return inds.numpy(), bboxes, depth, thr


def pc_frustum_heatmap(y, pc_dep: torch.Tensor, calib: torch.Tensor, K=100, max_pc_dist=60.0):
    """getPcFrustumHeatmap (pointcloud.py:331-394): boxes painted in top-k order."""
    B = pc_dep.shape[0]
    _, bboxes, depth, thr = box_params(y, calib, K)
    dep_np = pc_dep.numpy().astype(f32)
    pc_hm = np.zeros_like(dep_np)
    for b in range(B):
        for i in range(K):
            paint_box(pc_hm[b], dep_np[b], depth[b, i], bboxes[b, i], thr[b, i], max_pc_dist)
    return torch.from_numpy(pc_hm)
