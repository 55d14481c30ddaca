"""Deterministic synthetic inputs shared by make_golden.py (which feeds them to the reference)
and by the parity tests (which feed them to the oracle and to the HIP path).

Everything here is numpy-RandomState driven so the same bytes come out in the build container
and on the GPU box.  No reference code is involved.
"""
import math

import numpy as np
import torch


def tuned_state_dict(radar=True, seed=0):
    """Seeded weights whose depth / dimension / widthHeight heads are biased so that the
    frustum association actually fires (random-weight heads never pass the depth gate)."""
    from oracle import model_ref
    sd = model_ref.make_state_dict(radar=radar, seed=seed)
    hp = "detectHead_0"
    sd[f"{hp}.depth.2.bias"].fill_(-3.0)                 # 1/(sigmoid(-3)+1e-6)-1 ~ 20 m
    sd[f"{hp}.dimension.2.bias"].copy_(torch.tensor([1.6, 1.9, 4.4]))
    sd[f"{hp}.widthHeight.2.bias"].copy_(torch.tensor([9.0, 7.0]))
    return sd


def model_inputs(B, H, W, seed=0, radar=True, n_points=(20, 60)):
    """images (B,3,H,W) f32, pc_dep (B,3,H/4,W/4) f32 built from random pillar-like rectangles,
    calib (B,3,4) f32."""
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, 3, H, W)).astype(np.float32)
    h4, w4 = H // 4, W // 4
    calib = np.zeros((B, 3, 4), np.float32)
    calib[:, 0, 0] = calib[:, 1, 1] = 1266.4 * (W / 1600.0)
    calib[:, 0, 2] = 816.3 * (W / 1600.0)
    calib[:, 1, 2] = 491.5 * (W / 1600.0)
    calib[:, 2, 2] = 1.0
    if not radar:
        return torch.from_numpy(x), None, torch.from_numpy(calib)
    pc = np.zeros((B, 3, h4, w4), np.float32)
    for b in range(B):
        n = rs.randint(*n_points)
        depth = np.sort(rs.uniform(8.0, 40.0, n))
        for d in depth:                                    # ascending: far overwrites near
            cx, cy = rs.randint(0, w4), rs.randint(2, h4)
            ph, pw = max(1, int(round(40.0 / d * h4 / 112.0 * 6))), rs.randint(1, 3)
            y0, x0 = max(cy - ph, 0), max(cx - pw // 2, 0)
            pc[b, 0, y0:cy, x0:x0 + pw] = d
            pc[b, 1, y0:cy, x0:x0 + pw] = rs.normal(0, 5)
            pc[b, 2, y0:cy, x0:x0 + pw] = rs.normal(0, 5)
    return torch.from_numpy(x), torch.from_numpy(pc), torch.from_numpy(calib)


def frustum_case(seed, B=2, H=112, W=200, K=100, border=True):
    """Hand-built head dict with K well-separated, strictly ordered peaks per image, about half
    of which sit on a radar pillar of matching depth.  Returns (y, pc_dep, calib) torch tensors.
    Exercises: boxes crossing the left/top border (negative slice starts), overlapping boxes,
    depth-gate misses, zero-width boxes (negative widthHeight)."""
    rs = np.random.RandomState(seed)
    C = 10
    heat = rs.uniform(1e-4, 2e-3, (B, C, H, W)).astype(np.float32)
    wh = rs.uniform(-2.0, 6.0, (B, 2, H, W)).astype(np.float32)
    dep = rs.uniform(3.0, 55.0, (B, 1, H, W)).astype(np.float32)
    dim = rs.uniform(0.3, 5.0, (B, 3, H, W)).astype(np.float32)
    rot = rs.standard_normal((B, 8, H, W)).astype(np.float32)
    pc = np.zeros((B, 3, H, W), np.float32)
    calib = np.zeros((B, 3, 4), np.float32)
    calib[:, 0, 0] = calib[:, 1, 1] = 1266.4 / 8
    calib[:, 0, 2], calib[:, 1, 2], calib[:, 2, 2] = 816.3 / 8, 491.5 / 8, 1.0
    for b in range(B):
        # background pillars
        for _ in range(rs.randint(30, 80)):
            d = rs.uniform(2.0, 58.0)
            cx, cy = rs.randint(0, W), rs.randint(2, H)
            ph, pw = rs.randint(2, 25), rs.randint(1, 4)
            y0, x0 = max(cy - ph, 0), max(cx - pw // 2, 0)
            pc[b, 0, y0:cy, x0:x0 + pw] = d
            pc[b, 1, y0:cy, x0:x0 + pw] = rs.normal(0, 5)
            pc[b, 2, y0:cy, x0:x0 + pw] = rs.normal(0, 5)
        scores = np.sort(rs.uniform(0.05, 0.99, K).astype(np.float32))[::-1]
        scores = np.unique(scores)[::-1]
        taken = set()
        i = 0
        while i < len(scores):
            c, yy, xx = rs.randint(C), rs.randint(H), rs.randint(W)
            if border and i % 9 == 0:
                xx = rs.randint(0, 3)
            if border and i % 11 == 0:
                yy = rs.randint(0, 3)
            if (yy, xx) in taken:
                continue
            taken.add((yy, xx))
            heat[b, c, yy, xx] = scores[i]
            bw, bh = rs.uniform(2.0, 40.0), rs.uniform(2.0, 30.0)
            if i % 13 == 5:
                bw = -1.0                                   # clamps to 0 -> degenerate box
            wh[b, 0, yy, xx], wh[b, 1, yy, xx] = bw, bh
            d = rs.uniform(5.0, 50.0)
            dep[b, 0, yy, xx] = d
            dim[b, :, yy, xx] = (rs.uniform(1.2, 2.0), rs.uniform(1.5, 2.2), rs.uniform(3.5, 5.0))
            if i % 2 == 0:                                  # plant a matching pillar in the box
                rd = d + rs.uniform(-1.5, 1.5)
                px = int(np.clip(xx + rs.randint(-2, 3), 0, W - 1))
                py = int(np.clip(yy + rs.randint(-2, 3), 1, H - 1))
                pc[b, 0, max(py - 4, 0):py + 1, px:px + 2] = rd
                pc[b, 1, max(py - 4, 0):py + 1, px:px + 2] = rs.normal(0, 5)
                pc[b, 2, max(py - 4, 0):py + 1, px:px + 2] = rs.normal(0, 5)
            i += 1
    t = torch.from_numpy
    y = {"heatmap": t(heat), "widthHeight": t(wh), "depth": t(dep), "dimension": t(dim),
         "rotation": t(rot)}
    return y, t(pc), t(calib)


def decode_case(seed, B=2, H=112, W=200, radar=True, tie_heavy=False):
    """Random head dict for fusionDecode.  tie_free: continuous random heatmap (ties have
    probability ~0 among the top-K).  tie_heavy: clamped plateau so everything ties."""
    rs = np.random.RandomState(seed)
    t = torch.from_numpy
    if tie_heavy:
        heat = np.full((B, 10, H, W), 1e-4, np.float32)
        for b in range(B):
            for k in range(40):
                heat[b, rs.randint(10), rs.randint(H), rs.randint(W)] = np.float32(0.5)
    else:
        heat = (1 / (1 + np.exp(-(rs.standard_normal((B, 10, H, W)) * 1.2 - 3.0)))).astype(np.float32)
        heat = np.clip(heat, 1e-4, 1 - 1e-4)
    out = {"heatmap": t(heat)}
    spec = {"reg": 2, "widthHeight": 2, "depth": 1, "rotation": 8, "dimension": 3,
            "amodal_offset": 2, "nuscenes_att": 8, "velocity": 3}
    if radar:
        spec.update({"depth2": 1, "rotation2": 8})
    for k, c in spec.items():
        out[k] = t(rs.standard_normal((B, c, H, W)).astype(np.float32))
    return out
