"""Oracle: NMS + top-k decode of the head maps, CPU fp32.  TEST INFRASTRUCTURE.

Follows /root/reference/src/lib/model/utils.py:112-128 (nms) and
/root/reference/src/lib/model/decode.py:10-174 (fusionDecode) for the single-output-layer
case the DLA-34 model produces.  Ordering on ties: (score desc, class asc, pixel asc) - see
oracle/frustum_ref.py.  Pinned by tests/golden/decode_*.npz (reference fusionDecode on tie-free
heatmaps).
"""
import torch
import torch.nn.functional as F

from .frustum_ref import topk, gather_feat


def nms(heat, kernel=3):
    hmax = F.max_pool2d(heat, kernel, stride=1, padding=(kernel - 1) // 2)
    return heat * (hmax == heat).float()


def fusion_decode(outputs, output_size=(112, 200), K=100, norm2d=False):
    """outputs: [dict] as returned by the model.  Does not mutate its argument."""
    out = outputs[0]
    heat = out["heatmap"]
    B, _, H, W = heat.shape
    scores, inds, classes, ys, xs = topk(nms(heat), K)
    ys_n = ys.to(torch.float32) / H            # decode.py:40-41 (fp32 divide ...)
    xs_n = xs.to(torch.float32) / W
    ret = {"scores": scores, "classIds": classes.float(),
           "centers": torch.stack([xs_n, ys_n], dim=2)}
    xs_f = xs_n * output_size[1]               # ... and multiply back, decode.py:132-133
    ys_f = ys_n * output_size[0]
    if "reg" in out:
        reg = gather_feat(out["reg"], inds)
        xs_c = xs_f.unsqueeze(2) + reg[..., 0:1]
        ys_c = ys_f.unsqueeze(2) + reg[..., 1:2]
    else:
        xs_c = xs_f.unsqueeze(2) + 0.5
        ys_c = ys_f.unsqueeze(2) + 0.5
    scale = torch.tensor(output_size[::-1], dtype=torch.float32) if norm2d else 1
    if "widthHeight" in out:
        wh = gather_feat(out["widthHeight"], inds).clone()
        wh[wh < 0] = 0
        wh = wh * scale
        ret["bboxes"] = torch.cat([xs_c - wh[..., 0:1] / 2, ys_c - wh[..., 1:2] / 2,
                                   xs_c + wh[..., 0:1] / 2, ys_c + wh[..., 1:2] / 2], dim=2)
    src = {"rotation": "rotation2" if "rotation2" in out else "rotation",
           "dimension": "dimension", "amodal_offset": "amodal_offset",
           "nuscenes_att": "nuscenes_att", "velocity": "velocity",
           "depth": "depth2" if "depth2" in out else "depth"}
    for name, key in src.items():
        if key in out:
            v = gather_feat(out[key], inds)
            if name == "amodal_offset":
                v = v * scale
            ret[name] = v
    return ret


DET_FIELDS = [("scores", 1), ("classIds", 1), ("centers", 2), ("bboxes", 4), ("rotation", 8),
              ("dimension", 3), ("amodal_offset", 2), ("nuscenes_att", 8), ("velocity", 3),
              ("depth", 1)]


def pack_detections(ret):
    """(B,K,33) packing used by the multi-GPU all-gather."""
    cols = []
    for name, n in DET_FIELDS:
        v = ret[name]
        cols.append(v.reshape(v.shape[0], v.shape[1], n))
    return torch.cat(cols, dim=2)
