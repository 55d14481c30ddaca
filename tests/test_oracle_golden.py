"""CPU: the oracle restatement against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref, frustum_ref, decode_ref, postprocess_ref
from tests.golden import cases


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,radar,B,H,W", [("centerfusion_small", True, 2, 128, 160),
                                             ("centernet_small", False, 1, 96, 128)])
def test_model_forward_matches_reference(golden_dir, tag, radar, B, H, W):
    g = _load(golden_dir, f"model_{tag}.npz")
    sd = cases.tuned_state_dict(radar=radar, seed=0)
    x, pc_dep, calib = cases.model_inputs(B, H, W, seed=1, radar=radar)
    with torch.no_grad():
        y = model_ref.forward(sd, x, pc_dep=pc_dep, calib=calib, radar=radar, frustum=radar)[0]
    assert list(y.keys()) == [str(k) for k in g["key_order"]]
    for k in y:
        if k == "calib":
            continue
        _close(y[k], g[f"out_{k}"])
    if radar:
        assert int((y["pc_hm"] != 0).sum()) == int(g["n_painted"]) > 0
        # integer paint geometry must be bit-exact
        assert np.array_equal((y["pc_hm"] != 0).numpy(), g["out_pc_hm"] != 0)


def test_model_forward_fullres_samples(golden_dir):
    g = _load(golden_dir, "model_centerfusion_fullres.npz")
    sd = cases.tuned_state_dict(radar=True, seed=0)
    x, pc_dep, calib = cases.model_inputs(1, 448, 800, seed=2, radar=True, n_points=(80, 200))
    with torch.no_grad():
        y = model_ref.forward(sd, x, pc_dep=pc_dep, calib=calib)
        det = decode_ref.fusion_decode(y)
    y = y[0]
    for k, v in y.items():
        if k == "calib":
            continue
        flat = v.reshape(-1)
        _close(flat[g[f"idx_{k}"]], g[f"val_{k}"])
        np.testing.assert_allclose(float(flat.double().sum()), float(g[f"sum_{k}"]), rtol=1e-4,
                                   atol=1e-2)
    assert int((y["pc_hm"] != 0).sum()) == int(g["n_painted"])
    for k, v in det.items():
        _close(v, g[f"det_{k}"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(det["classIds"].numpy(), g["det_classIds"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frustum_matches_reference(golden_dir, seed):
    g = _load(golden_dir, f"frustum_{seed}.npz")
    y, pc_dep, calib = cases.frustum_case(seed)
    s, inds, cls, ys, xs = frustum_ref.topk(y["heatmap"], 100)
    assert np.array_equal(inds.numpy(), g["topk_inds"])          # index path: bit-exact
    assert np.array_equal(cls.numpy(), g["topk_cls"])
    assert np.array_equal(ys.numpy(), g["topk_ys"]) and np.array_equal(xs.numpy(), g["topk_xs"])
    assert np.array_equal(s.numpy(), g["topk_scores"])
    pc_hm = frustum_ref.pc_frustum_heatmap(y, pc_dep, calib, 100, 60.0)
    assert tuple(pc_hm.shape) == tuple(g["shape"])
    flat = pc_hm.reshape(-1).numpy()
    nz = np.nonzero(flat)[0]
    assert np.array_equal(nz, g["nz_idx"])                       # painted pixel set: bit-exact
    assert np.array_equal(flat[nz], g["nz_val"])                 # painted values: bit-exact


def test_frustum_handcases(golden_dir):
    g = _load(golden_dir, "frustum_handcases.npz")
    pc_dep = np.zeros((3, 112, 200), np.float32)
    pc_dep[0, 40:60, 0:10], pc_dep[1, 40:60, 0:10], pc_dep[2, 40:60, 0:10] = 10.0, 1.5, -2.5
    for name, box in (("neg", (-1.5, 42.0, 7.5, 58.0)), ("pos", (0.5, 42.0, 7.5, 58.0))):
        pc_hm = np.zeros_like(pc_dep)
        frustum_ref.paint_box(pc_hm, pc_dep, np.float32(10.5), box, np.float32(2.0), 60.0)
        nz = np.argwhere(pc_hm[0] != 0)
        assert np.array_equal(nz, g[f"{name}_nz"])
        if len(nz):
            assert np.array_equal(pc_hm[:, nz[:, 0], nz[:, 1]], g[f"{name}_val"])
    assert len(g["neg_nz"]) == 0 and len(g["pos_nz"]) == 6 * 5   # Appendix B.5


@pytest.mark.parametrize("name,seed,radar,norm2d", [("decode_0.npz", 0, True, False),
                                                    ("decode_1.npz", 1, False, False),
                                                    ("decode_2_norm2d.npz", 2, True, True)])
def test_decode_matches_reference(golden_dir, name, seed, radar, norm2d):
    g = _load(golden_dir, name)
    out = cases.decode_case(seed, radar=radar)
    det = decode_ref.fusion_decode([out], (112, 200), 100, norm2d)
    assert set(det.keys()) == set(g.files)
    for k in g.files:
        assert np.array_equal(det[k].numpy(), g[k]), k           # tie-free: bit-exact


def test_decode_ties_as_sets(golden_dir):
    g = _load(golden_dir, "decode_3_ties.npz")
    out = cases.decode_case(3, radar=True, tie_heavy=True)
    s, inds, cls, _, _ = frustum_ref.topk(decode_ref.nms(out["heatmap"]), 100)
    s, inds, cls = s.numpy(), inds.numpy(), cls.numpy()
    assert np.array_equal(s, g["scores"])                        # score sequence is well defined
    for b in range(s.shape[0]):
        distinct = s[b] > s[b].min()
        ours = set(zip(cls[b][distinct].tolist(), inds[b][distinct].tolist()))
        ref = set(zip(g["cls"][b][distinct].tolist(), g["inds"][b][distinct].tolist()))
        assert ours == ref
        # inside the tie plateau our order is (class asc, pixel asc)
        tie = ~distinct
        key = cls[b][tie].astype(np.int64) * (112 * 200) + inds[b][tie]
        assert np.all(np.diff(key) > 0)


def postprocess_inputs(seed):
    out = cases.decode_case(seed, radar=True)
    out["depth2"] = out["depth2"].abs() * 20 + 2
    out["dimension"] = out["dimension"].abs() + 0.1
    if seed == 1:
        out["dimension"][:, 1] -= 0.6
    return out, cases.model_inputs(2, 448, 800, seed=0)[2]


@pytest.mark.parametrize("seed", [0, 1])
def test_postprocess_matches_reference(golden_dir, seed):
    g = _load(golden_dir, f"postprocess_{seed}.npz")
    out, calibs = postprocess_inputs(seed)
    det = decode_ref.fusion_decode([out], (112, 200), 100)
    pp = postprocess_ref.post_process(det, (800.0, 450.0), 1600.0, 112, 200, calibs)
    assert set(pp.keys()) == set(g.files)
    for k in g.files:
        assert np.array_equal(pp[k].numpy(), g[k]), k
    if seed == 1:                       # boxes with a non-positive dimension are zeroed
        bad = (g["dimension"] <= 0).any(-1)
        assert bad.any() and not g["bboxes3d"][bad].any()


def _oracle_stages(sd, x):
    """backbone levels, DLA-up outputs and the feature map of the oracle, named like the goldens"""
    layers = list(model_ref.dla34_base(sd, x))
    d = {f"y{i}": t for i, t in enumerate(layers)}
    out = [layers[-1]]
    for i in range(len(layers) - 2 - 1):
        model_ref._ida(sd, f"dla_up.ida_{i}", layers, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    for i, t in enumerate(out):
        d[f"up{i}"] = t
    y = [out[i].clone() for i in range(3)]
    model_ref._ida(sd, "ida_up", y, 0, 3)
    d["feat"] = y[-1]
    return d


@pytest.mark.parametrize("tag,radar,B,H,W", [("centerfusion_small", True, 2, 128, 160),
                                             ("centernet_small", False, 1, 96, 128)])
def test_stage_outputs_match_reference_submodules(golden_dir, tag, radar, B, H, W):
    """Per-stage pin: the reference's own `base`, `dla_up`, `ida_up` outputs (sampled values, float64 sums)
    against the oracle's stages - levels 0-5, the four DLA-up maps, the 64-channel feature map."""
    g = _load(golden_dir, f"model_{tag}.npz")
    sd = cases.tuned_state_dict(radar=radar, seed=0)
    x, _, _ = cases.model_inputs(B, H, W, seed=1, radar=radar)
    with torch.no_grad():
        st = _oracle_stages(sd, x)
    names = sorted(k[len("stage_val_"):] for k in g.files if k.startswith("stage_val_"))
    assert names == sorted(st.keys()) and len(names) == 11
    for n in names:
        t = st[n]
        assert list(t.shape) == g[f"stage_shape_{n}"].tolist(), n
        flat = t.reshape(-1)
        _close(flat[g[f"stage_idx_{n}"]], g[f"stage_val_{n}"], rtol=1e-4, atol=1e-5 * float(flat.abs().max()))
        np.testing.assert_allclose(float(flat.double().sum()), float(g[f"stage_sum_{n}"]),
                                   rtol=1e-5, atol=1e-4 * float(flat.abs().max()) * flat.numel() ** 0.5)
