"""CPU: host-side logic of the product package - weight packing (validated by emulating the
implicit GEMM with the packed weights + slot table in numpy), BN folding, the parameter tree /
state_dict contract, config-derived heads, and the no-fallback rule."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from centerfusiondetect3d_amd import (getModel, centerfusion_middle_config, centernet_config, packing,
                                      getAffineTransform, _lib)
from oracle import model_ref, pillar_ref


def emulate_igemm(pc, srcs, stride):
    """out[m][n] = sum_slots sum_e  src[b, y*stride+dy, x*stride+dx, c_off+e] * W[n][4*slot+e]."""
    B, H, W, _ = srcs[0].shape
    Ho = (H + 2 * pc.pad - pc.kh) // stride + 1
    Wo = (W + 2 * pc.pad - pc.kh) // stride + 1
    out = np.zeros((B, Ho, Wo, pc.n_pad), np.float64)
    w = pc.weight.numpy().astype(np.float64)
    ys, xs = np.arange(Ho) * stride, np.arange(Wo) * stride
    for j, (src, dy, dx, c_off) in enumerate(pc.slots.numpy().tolist()):
        if c_off < 0:
            assert not w[:, 4 * j:4 * j + 4].any()
            continue
        t = srcs[src]
        yy, xx = ys + dy, xs + dx
        oky, okx = (yy >= 0) & (yy < H), (xx >= 0) & (xx < W)
        patch = np.zeros((B, Ho, Wo, 4))
        sub = t[:, yy[oky]][:, :, xx[okx]][..., c_off:c_off + 4]
        patch[np.ix_(range(B), np.nonzero(oky)[0], np.nonzero(okx)[0], range(sub.shape[-1]))] = sub
        out += patch @ w[:, 4 * j:4 * j + 4].T
    return out[..., :pc.n] + pc.bias.numpy()[:pc.n]


@pytest.mark.parametrize("ci_list,strides,k,stride", [([16], [16], 3, 2), ([3], [4], 7, 1),
                                                      ([64, 3], [64, 4], 3, 1), ([32, 64, 32], [32, 64, 32], 1, 1)])
def test_pack_conv_slot_table_is_a_convolution(ci_list, strides, k, stride):
    g = torch.Generator().manual_seed(0)
    B, H, W, co = 2, 9, 11, 10
    xs = [torch.randn(B, c, H, W, generator=g) for c in ci_list]
    w = torch.randn(co, sum(ci_list), k, k, generator=g)
    b = torch.randn(co, generator=g)
    ref = F.conv2d(torch.cat(xs, 1), w, b, stride, k // 2).permute(0, 2, 3, 1).numpy()
    pc = packing.pack_conv(w, b, [packing.Source(c, s) for c, s in zip(ci_list, strides)], stride=stride)
    assert pc.k_pad % 32 == 0 and pc.n_pad % 32 == 0 and pc.slots.shape == (pc.k_pad // 4, 4)
    sl = pc.slots.view(-1, 8, 4)
    assert bool((sl[:, :, 0] == sl[:, :1, 0]).all()), "a 32-wide K chunk must stay inside one source"
    srcs = []
    for x, s in zip(xs, strides):
        t = np.full((B, H, W, s), 7.0)          # padding lanes hold junk: weights there must be 0
        t[..., :x.shape[1]] = x.permute(0, 2, 3, 1).numpy()
        srcs.append(t)
    np.testing.assert_allclose(emulate_igemm(pc, srcs, stride), ref, rtol=1e-5, atol=1e-5)


def test_fold_bn_and_pack_dcn():
    g = torch.Generator().manual_seed(1)
    w, b = torch.randn(8, 32, 3, 3, generator=g), torch.randn(8, generator=g)
    bn = (torch.rand(8) + 0.5, torch.randn(8), torch.randn(8), torch.rand(8) + 0.5)
    x = torch.randn(1, 32, 6, 6, generator=g)
    ref = F.batch_norm(F.conv2d(x, w, b, 1, 1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    wf, bf = packing.fold_bn(w, b, bn)
    torch.testing.assert_close(F.conv2d(x, wf, bf, 1, 1), ref, rtol=1e-4, atol=1e-5)
    pd = packing.pack_dcn(wf, bf)
    assert pd.weight.shape == (32, 9 * 32) and pd.n == 8
    assert torch.equal(pd.weight[:8].view(8, 9, 32)[:, 4], wf[:, :, 1, 1])     # k = tap*C + c
    assert not pd.weight[8:].any()


@pytest.mark.parametrize("radar", [True, False])
def test_state_dict_contract(radar):
    m = getModel((centerfusion_middle_config if radar else centernet_config)())
    ref = model_ref.make_state_dict(radar=radar)
    sd = m.state_dict()
    assert set(sd) == set(ref)
    assert all(sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype for k in sd)
    n_params = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert abs(n_params / 1e6 - (21.36 if radar else 20.51)) < 0.01
    # reference init facts the detector relies on
    assert float(sd["detectHead_0.heatmap.2.bias"][0]) == pytest.approx(-4.6)
    assert not sd["ida_up.proj_1.conv_offset_mask.weight"].any()
    up = sd["ida_up.up_2.weight"]
    assert up.shape == (64, 1, 8, 8) and torch.equal(up[0], up[63])
    m._packed, m._plans = object(), {"x": 1}
    m.load_state_dict(ref, strict=True)
    assert m._packed is None and not m._plans          # weights changed -> re-pack lazily


def test_config_heads_match_reference_derivation():
    c = centerfusion_middle_config()
    assert list(c.heads.keys()) == ["heatmap", "reg", "widthHeight", "depth", "rotation", "dimension",
                                    "amodal_offset", "nuscenes_att", "velocity", "depth2", "rotation2"]
    assert c.head_conv["velocity"] == [256, 256, 256] and c.head_conv["reg"] == [256]
    n = centernet_config()
    assert "depth2" not in n.heads and all(v == [256] for v in n.head_conv.values())
    assert c.MODEL.OUTPUT_SIZE == (112, 200)


def test_no_cpu_fallback_and_scope_guards():
    m = getModel(centerfusion_middle_config((64, 64)))
    with pytest.raises(_lib.CfHipError):
        m(torch.zeros(1, 3, 64, 64), pc_dep=torch.zeros(1, 3, 16, 16), calib=torch.zeros(1, 3, 4))
    cfg = centerfusion_middle_config()
    cfg.MODEL.DLA.NODE = "Conv"
    with pytest.raises(NotImplementedError):
        getModel(cfg)
    cfg = centerfusion_middle_config()
    cfg.MODEL.FUSION_STRATEGY = "early"
    with pytest.raises(NotImplementedError):
        getModel(cfg)


def test_affine_transform_matches_oracle():
    for (w, h), out in (((1600, 900), (200, 112)), ((1600, 900), (800, 448)), ((1280, 720), (400, 224))):
        a = getAffineTransform((w / 2, h / 2), float(max(w, h)), 0, out)
        b = pillar_ref.affine_transform_matrix((w / 2, h / 2), float(max(w, h)), out)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
