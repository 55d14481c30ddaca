"""Oracle: 2D -> 3D post-processing of decoded detections, CPU fp32.  TEST INFRASTRUCTURE.

Follows /root/reference/src/lib/utils/postProcess.py:13-85 with its helpers
  * utils/image.py:43-83, 86-110      getAffineTransform(inverse=True), affineTransform
  * utils/ddd.py:122-199              alpha2rot_y, project2DTo3D, cvtImgToCamCoord
  * utils/ddd.py:8-23 + utils/pointcloud.py:239-296   get3dBox / get3DCorners
  * utils/pointcloud.py:195-211       get_alpha
for the inference case (isGt=False, amodal_offset present).  Pinned by tests/golden/postprocess_*.npz
(the reference's own postProcess run here; its cv2.getAffineTransform import is served by a numpy
3-point solve, so only that solve is outside the pin).
"""
import math

import numpy as np
import torch

from .pillar_ref import affine_transform_matrix


def inverse_affine(center, scale, out_wh):
    """getAffineTransform(center, scale, 0, (w, h), inverse=True) as float32 (2,3)."""
    fwd = affine_transform_matrix(center, scale, out_wh)          # src image -> output map
    A = np.vstack([fwd, [0.0, 0.0, 1.0]])
    return np.linalg.inv(A)[:2].astype(np.float32)


def _affine(points, t):
    """affineTransform: (N,2) points, (2,3) float32 matrix."""
    ones = torch.ones(points.shape[0], 1)
    return (torch.from_numpy(t) @ torch.cat([points, ones], 1).T).T


def post_process(det, center, scale, height, width, calibs):
    """det: dict as returned by fusion_decode (not mutated) -> new dict with the reference's keys."""
    y = {k: v.clone() for k, v in det.items()}
    B, K = y["scores"].shape
    t = inverse_affine(center, scale, (width, height))
    y["classIds"] = y["classIds"] + 1
    y["centers"] = y["centers"] * torch.tensor([width, height], dtype=torch.float32)
    y["bboxes"] = _affine(y["bboxes"].reshape(-1, 2), t).reshape(B, K, 4)
    y["depth"] = y["depth"].reshape(B, K)
    rot = y.pop("rotation").reshape(-1, 8)
    idx = (rot[:, 1] > rot[:, 5]).float()
    a1 = torch.atan2(rot[:, 2], rot[:, 3]) + (-0.5 * math.pi)
    a2 = torch.atan2(rot[:, 6], rot[:, 7]) + (0.5 * math.pi)
    y["alpha"] = (a1 * idx + a2 * (1 - idx)).reshape(B, K)
    amodal = y["centers"] + y["amodal_offset"]
    y["centers"] = _affine(amodal.reshape(-1, 2), t).reshape(B, K, 2)
    cal = calibs.reshape(B, 1, 3, 4).expand(B, K, 3, 4)
    d = y["depth"]
    z = d - cal[:, :, 2, 3]
    x = (y["centers"][..., 0] * d - cal[:, :, 0, 3] - cal[:, :, 0, 2] * z) / cal[:, :, 0, 0]
    yy = (y["centers"][..., 1] * d - cal[:, :, 1, 3] - cal[:, :, 1, 2] * z) / cal[:, :, 1, 1]
    loc = torch.stack([x, yy, z], dim=-1)
    loc[:, :, 1] += y["dimension"][:, :, 0] / 2
    yaw = y["alpha"] + torch.atan2(y["centers"][..., 0] - cal[:, :, 0, 2], cal[:, :, 0, 0])
    yaw[yaw > math.pi] -= 2 * math.pi
    yaw[yaw < -math.pi] += 2 * math.pi
    y["locations"], y["yaws"] = loc, yaw
    V = torch.sqrt(y["velocity"][:, :, 0] ** 2 + y["velocity"][:, :, 2] ** 2)
    y["velocity"][:, :, 0] = torch.cos(yaw) * V
    y["velocity"][:, :, 2] = -torch.sin(yaw) * V
    dim = y["dimension"]
    c, s = torch.cos(yaw), torch.sin(yaw)
    l, w, h = dim[..., 2], dim[..., 1], dim[..., 0]
    sx = torch.tensor([1, 1, -1, -1, 1, 1, -1, -1], dtype=torch.float32) * 0.5
    sz = torch.tensor([1, -1, -1, 1, 1, -1, -1, 1], dtype=torch.float32) * 0.5
    xc = sx * l.unsqueeze(-1)
    yc = torch.cat([torch.zeros(B, K, 4), -h.unsqueeze(-1).expand(B, K, 4)], dim=-1)
    zc = sz * w.unsqueeze(-1)
    cx3 = c.unsqueeze(-1) * xc + s.unsqueeze(-1) * zc
    cz3 = -s.unsqueeze(-1) * xc + c.unsqueeze(-1) * zc
    box = torch.stack([cx3, yc, cz3], dim=-1) + loc.unsqueeze(2)
    box[torch.any(dim <= 0, dim=2)] = 0
    y["bboxes3d"] = box
    return y
