"""Drop-in for the reference's utils/postProcess.py:13-85 (`postProcess`) on the HIP path: the
step right after decode.  One kernel (cf_post_process) maps the
(B,K,33) detections to image-space boxes, 3D locations, yaw, heading-aligned velocity and the 8 box
corners; with `post_process_packed` the multi-GPU all-gather can ship these final (B,K,54) rows.
"""
import numpy as np
import torch

from . import _lib
from .decode import DET_FIELDS, DET_WIDTH
from .pointcloud import getAffineTransform

POST_FIELDS = [("scores", 1), ("classIds", 1), ("centers", 2), ("bboxes", 4), ("depth", 1), ("alpha", 1),
               ("dimension", 3), ("amodal_offset", 2), ("nuscenes_att", 8), ("velocity", 3),
               ("locations", 3), ("yaws", 1), ("bboxes3d", 24)]
POST_WIDTH = sum(n for _, n in POST_FIELDS)


def inverse_affine(center, scale, out_wh):
    """getAffineTransform(center, scale, 0, (w, h), inverse=True).astype(float32)."""
    fwd = getAffineTransform(center, scale, 0, out_wh)
    return np.linalg.inv(np.vstack([fwd, [0.0, 0.0, 1.0]]))[:2].astype(np.float32)


def inverse_affine_device(center, scale, out_wh, device):
    return torch.from_numpy(inverse_affine(center, scale, out_wh)).to(device)


def post_process_packed(det, calibs, center, scale, height, width):
    """det (B,K,33) device tensor -> (B,K,54) device tensor."""
    if not det.is_cuda:
        raise _lib.CfHipError("cf_post_process needs device tensors (no CPU path)")
    B, K, wdt = det.shape
    assert wdt == DET_WIDTH
    tinv = torch.from_numpy(inverse_affine(center, scale, (width, height))).to(det.device)
    calibs = calibs.reshape(B, 3, 4).float().contiguous()
    out = torch.empty((B, K, POST_WIDTH), device=det.device, dtype=torch.float32)
    _lib.check(_lib.load().cf_post_process(det.contiguous().data_ptr(), calibs.data_ptr(), tinv.data_ptr(), B, K,
                                           int(height), int(width), out.data_ptr(), _lib.stream_ptr()),
               "cf_post_process")
    return out


def unpack_post(out):
    ret, col = {}, 0
    for name, n in POST_FIELDS:
        v = out[..., col:col + n]
        if name in ("scores", "classIds", "depth", "alpha", "yaws"):
            v = v[..., 0]
        elif name == "bboxes3d":
            v = v.reshape(*out.shape[:2], 8, 3)
        ret[name] = v
        col += n
    return ret


def postProcess(y, center, scale, height, width, calibs, isGt=False):
    """Same signature and returned keys as the reference.  `y` is the dict returned by fusionDecode
    (all ten fields present, i.e. a 3D model); it is updated in place like the reference does
    (`rotation` is replaced by `alpha`; `locations`, `yaws`, `bboxes3d` are added)."""
    if isGt:
        raise NotImplementedError("ground-truth post-processing is a training/eval utility outside the hot path")
    need = [n for n, _ in DET_FIELDS]
    if any(k not in y for k in need):
        raise NotImplementedError("postProcess on the HIP path needs the full 3D detection dict " + str(need))
    cols = [y[n].reshape(*y["scores"].shape, -1) for n in need]
    det = torch.cat(cols, dim=2).contiguous()
    out = unpack_post(post_process_packed(det, calibs, center, scale, height, width))
    y.pop("rotation")
    y.update(out)
    return y
