"""GPU tests of the dynamic-range guard of the split-fp16 kernels.

The f16x3 / mx kernels multiply every activation by a power of two (16 by default) and split it into fp16 hi + lo: beyond
65504 / pre-scale the value is CLAMPED, where the reference's fp32 convolutions (model/networks/dla.py:124-159) accept any
magnitude.  These tests build weights whose BatchNorm gains push chosen layer inputs far beyond 4094 - compensated behind
the layer, so the network's outputs stay those of an ordinary model - and require:
  * the default path RAISES (`check_ranges`, `Detector`'s first batch) and names the layer - never a silently clamped map;
  * the calibrated path (`calibrate`: per-layer power-of-two pre-scales) matches the oracle within the suite's tolerance,
    discrete path included;
  * calibration changes nothing, bit for bit, on a model that does not need it.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model_ref
from tests.golden import cases
from tests.test_gpu_model import _assert_maps_close, _fp32_noise


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------ cf_absmax_f32
def test_absmax_kernel(dev):
    from centerfusiondetect3d_amd import ops
    g = torch.Generator().manual_seed(3)
    for shape in ((2, 28, 50, 256), (1, 7, 5, 27), (3, 1, 1, 4), (5,), (1, 13, 17, 64)):
        x = torch.randn(shape, generator=g) * 37.0
        assert float(ops.absmax(x.to(dev)).item()) == float(x.abs().max())
    x = torch.randn(4, 56, 100, 128, generator=g)
    x[3, 55, 99, 127] = -1e30                                        # the last element of a large tensor
    assert float(ops.absmax(x.to(dev)).item()) == 1e30 or float(ops.absmax(x.to(dev)).item()) == float(np.float32(1e30))
    x[0, 0, 0, 1] = float("inf")
    assert float(ops.absmax(x.to(dev)).item()) == float("inf")
    x[2, 3, 4, 5] = float("nan")
    assert np.isnan(float(ops.absmax(x.to(dev)).item()))
    # a row-strided view (27 channels of a 32-wide buffer: the unused tail must not be read) and an unaligned start
    buf = torch.full((1000, 32), 1e9)
    buf[:, :27] = torch.randn(1000, 27, generator=g)
    d = buf.to(dev)
    assert float(ops.absmax(d[:, :27]).item()) == float(buf[:, :27].abs().max())
    flat = torch.randn(4099, generator=g).to(dev)
    assert float(ops.absmax(flat[1:]).item()) == float(flat[1:].abs().max())
    assert float(ops.absmax(torch.empty(0, device=dev)).item()) == 0.0
    out = torch.full((1,), 7.0, device=dev)                          # the call zeroes `out` itself
    ops.absmax(torch.full((8, 4), 0.5, device=dev), out=out)
    assert float(out.item()) == 0.5


# --------------------------------------------------------------------- operator level: in_scale of the kernels
@pytest.mark.parametrize("amp", [1.0, 3000.0, 60000.0, 3.0e7])
def test_conv3x3_in_scale_matches_fp32_oracle(dev, amp):
    """cf_conv3x3_f16x3 with the pre-scale `in_scale_for(max |x|)` against F.conv2d in float64: same relative accuracy for
    inputs of magnitude 1, 3e3 (default pre-scale still fits), 6e4 and 3e7 (it does not: without in_scale the result is
    wrong, shown below)."""
    import torch.nn.functional as F
    from centerfusiondetect3d_amd import ops, packing
    g = torch.Generator().manual_seed(11)
    B, H, W, Ci, Co = 2, 28, 50, 64, 128
    x = torch.randn(B, Ci, H, W, generator=g) * amp
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    pc = packing.pack_conv_f16(w, b, [packing.Source(Ci, Ci)]).to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    s = ops.in_scale_for(float(x.abs().max()))
    assert (s == 16.0) == (amp <= 200.0)
    out = torch.empty((B, H, W, Co), device=dev)
    a = ops.conv_args(pc, [xd], [Ci], B, H, W, out, Co, ops.ACT_NONE, None, 0, ops.LAYOUT_NHWC, None, 0, False, in_scale=s)
    ops.run_conv_f16(a, patch=True)
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, err
    # ... and through the slot kernel (cf_conv2d_f16x3), same bits
    out2 = torch.empty_like(out)
    a2 = ops.conv_args(pc, [xd], [Ci], B, H, W, out2, Co, ops.ACT_NONE, None, 0, ops.LAYOUT_NHWC, None, 0, False, in_scale=s)
    ops.run_conv_f16(a2, patch=False)
    assert float((out2.double().cpu() - ref).abs().max() / ref.abs().max()) < 1e-6
    if amp * 16 > 65504 * 4:
        # the hazard itself: the default pre-scale clamps these inputs - the result is far off, and nothing says so
        a3 = ops.conv_args(pc, [xd], [Ci], B, H, W, out2, Co, ops.ACT_NONE, None, 0, ops.LAYOUT_NHWC, None, 0, False)
        ops.run_conv_f16(a3, patch=True)
        assert float((out2.double().cpu() - ref).abs().max() / ref.abs().max()) > 1e-2
    # an in_scale that is not a power of two is refused
    a.in_scale = 3.0
    from centerfusiondetect3d_amd import _lib
    with pytest.raises(_lib.CfHipError, match="power of two"):
        ops.run_conv_f16(a, patch=True)


@pytest.mark.parametrize("amp", [1.0, 5.0e4])
def test_deform_conv2d_operator_accepts_any_range(dev, amp):
    """ops.deform_conv2d (= torchvision's signature) measures its input and picks the pre-scale: inputs of 5e4 come out as
    accurate as inputs of 1; with the check switched off they are clamped (the documented contract of that switch)."""
    from centerfusiondetect3d_amd import ops
    from oracle import dcn_ref
    g = torch.Generator().manual_seed(5)
    B, C, H, W, N = 1, 64, 20, 24, 64
    x = torch.randn(B, C, H, W, generator=g) * amp
    off = torch.randn(B, 18, H, W, generator=g) * 1.5
    mask = torch.sigmoid(torch.randn(B, 9, H, W, generator=g))
    w = torch.randn(N, C, 3, 3, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    ref = dcn_ref.deform_conv2d(x.double(), off.double(), w.double(), b.double(), padding=(1, 1), mask=mask.double())
    got = ops.deform_conv2d(x.to(dev), off.to(dev), w.to(dev), b.to(dev), padding=(1, 1), mask=mask.to(dev))
    err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    if amp > 4094:
        prev = ops.set_dcn_range_check(False)
        try:
            bad = ops.deform_conv2d(x.to(dev), off.to(dev), w.to(dev), b.to(dev), padding=(1, 1), mask=mask.to(dev))
        finally:
            ops.set_dcn_range_check(prev)
        assert float((bad.double().cpu() - ref).abs().max() / ref.abs().max()) > 1e-2
    x[0, 0, 0, 0] = float("nan")
    from centerfusiondetect3d_amd import _lib
    with pytest.raises(_lib.CfHipError, match="not finite"):
        ops.deform_conv2d(x.to(dev), off.to(dev), w.to(dev), b.to(dev), padding=(1, 1), mask=mask.to(dev))


@pytest.mark.parametrize("scale", [16.0, 1.0, 0.03125])
def test_pack_feat_mx_scaled_rows_bit_exact(dev, scale):
    from centerfusiondetect3d_amd import ops
    from oracle import mx_emul
    g = torch.Generator().manual_seed(9)
    x = torch.randn(257, 64, generator=g) * (40.0 / scale)
    x[5] = 0.0
    x[6, :32] = 1e-39                                                # fp32 denormals: a zero block, never E8M0 255
    x[7, 32:] = 2.0 ** -130
    x[8] = 65504.0 * 2.0 / scale                                     # beyond the limit: clamped (pinned, the guard is upstream)
    rows = ops.pack_feat_mx(x.to(dev), scale=scale).cpu().numpy()
    ref = mx_emul.feat_rows_ref(x.numpy(), scale=scale)
    assert np.array_equal(rows, ref)
    assert rows[:, 256:260].max() < 255


# ------------------------------------------------------------------------------------- module level: the guard
def _boost(sd, kind, K):
    """BN gains that push ONE layer's input K x beyond its usual magnitude, compensated right behind that layer so the rest
    of the network sees ordinary values.  kind: which layer."""
    sd = {k: v.clone() for k, v in sd.items()}

    def gain(bn, f):
        sd[bn + ".weight"] *= f
        sd[bn + ".bias"] *= f

    if kind == "level5":            # input of base.level5.tree1.conv2 (its BN divides again)
        gain("base.level5.tree1.bn1", K)
        sd["base.level5.tree1.bn2.running_mean"] *= K
        sd["base.level5.tree1.bn2.running_var"] *= K * K
    elif kind == "stem":            # base_layer's output - an operand that never leaves LDS in the fused stem
        gain("base.base_layer.1", K)
        sd["base.level0.1.running_mean"] *= K
        sd["base.level0.1.running_var"] *= K * K
    elif kind == "x2":              # x2 of level 3's first Tree: on the chip in the conv2 + Root launch
        gain("base.level3.tree1.tree2.bn2", K)                       # conv2's BN: x2 = ReLU(K * (...) + x1)
        sd["base.level3.tree1.root.conv.weight"][:, :128] /= K       # Root reads (x2, x1): the x2 columns
    elif kind == "feat":            # the feature map the heads read (mx rows)
        gain("ida_up.node_2.activation.0", K)
        for k in sd:
            if k.startswith("detectHead_0.") and k.endswith(".0.weight"):
                sd[k][:, :64] /= K
    else:
        raise ValueError(kind)
    return sd


LAYER_OF = {"level5": "base.level5.tree1.conv2", "stem": "base.level0", "x2": "base.level3.tree1.root", "feat": "heads.primary.0"}


@pytest.mark.parametrize("kind", ["level5", "stem", "x2", "feat"])
def test_range_guard_raises_and_calibration_matches_the_oracle(dev, kind):
    from centerfusiondetect3d_amd import getModel, centerfusion_middle_config, _lib
    B, H, W = 2, 128, 160
    K = 4096.0
    sd = _boost(cases.tuned_state_dict(radar=True, seed=0), kind, K)
    x, pc_dep, calib = cases.model_inputs(B, H, W, seed=1, radar=True)
    noise, r32, _ = _fp32_noise(sd, x, pc_dep, calib, True)
    m = getModel(centerfusion_middle_config((H, W)))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    xd, pd, cd_ = x.to(dev), pc_dep.to(dev), calib.to(dev)
    layer = LAYER_OF[kind]

    # 1. the default path raises and names the layer
    with pytest.raises(_lib.CfHipError, match="activation range guard") as ei:
        m.check_ranges(xd, pc_dep=pd, calib=cd_)
    assert layer in str(ei.value), str(ei.value)
    ranges = m.measure_ranges(xd, pc_dep=pd, calib=cd_)
    assert ranges[layer] > 4094.0, (layer, ranges[layer])            # the boost really crosses the clamp

    # 2. what the guard protects from: the unguarded forward is silently far off on at least one output
    with torch.no_grad():
        y_bad = m(xd, pc_dep=pd, calib=cd_)[0]
    worst = max(float((y_bad[k].cpu() - r32[k]).abs().max() / (r32[k].abs().max() + 1e-12)) for k in r32 if k != "calib")
    assert worst > 1e-2, worst
    # ... and the cheap form of the guard sees it in the buffers that forward left behind, where the operand reaches HBM
    if layer not in m.hidden_layers():
        with pytest.raises(_lib.CfHipError, match=layer.replace(".", r"\.")):
            m.check_resident_ranges()

    # 3. calibrated: per-layer power-of-two pre-scales, the oracle's result within the suite's tolerance
    m.calibrate(xd, pc_dep=pd, calib=cd_)
    with torch.no_grad():
        y = m(xd, pc_dep=pd, calib=cd_)[0]
    scales = m.activation_scales()
    assert scales[layer] < 16.0 and scales[layer] * ranges[layer] <= 65504.0 / 8.0 and scales[layer] * ranges[layer] > 65504.0 / 16.5
    assert sum(1 for s in scales.values() if s != 16.0) <= 4, scales    # everything else keeps the default
    assert torch.equal(y["pc_hm"].cpu(), r32["pc_hm"]) and int((r32["pc_hm"] != 0).sum()) > 0
    for k in r32:
        if k != "calib":
            _assert_maps_close(y[k], r32[k], k, e32=noise[k])
    assert m.range_violations(ranges) == []
    m.check_ranges(xd, pc_dep=pd, calib=cd_)                         # the guard is quiet now

    # 4. new weights void the calibration
    m.load_state_dict(cases.tuned_state_dict(radar=True, seed=0))
    assert m.calibration() is None and m._range_checked is False


def test_calibration_is_a_no_op_where_no_layer_needs_it(dev):
    """calibrate() on an ordinary model keeps every pre-scale at 16: the forward is bit-identical to the uncalibrated one
    (the bench configuration's numbers do not depend on whether a deployment calibrates)."""
    from centerfusiondetect3d_amd import getModel, centerfusion_middle_config
    B, H, W = 2, 128, 160
    sd = cases.tuned_state_dict(radar=True, seed=0)
    x, pc_dep, calib = cases.model_inputs(B, H, W, seed=1, radar=True)
    m = getModel(centerfusion_middle_config((H, W)))
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    xd, pd, cd_ = x.to(dev), pc_dep.to(dev), calib.to(dev)
    with torch.no_grad():
        y0 = m(xd, pc_dep=pd, calib=cd_)[0]
    seed_state = torch.random.get_rng_state()
    r = m.calibrate(xd, pc_dep=pd, calib=cd_)
    assert torch.equal(torch.random.get_rng_state(), seed_state)      # the shadow's initialisation leaves the caller's RNG alone
    assert max(r.values()) < 1023.0
    with torch.no_grad():
        y1 = m(xd, pc_dep=pd, calib=cd_)[0]
    assert set(m.activation_scales().values()) == {16.0}
    for k in y0:
        assert torch.equal(y0[k], y1[k]), k
    # the resident-buffer ranges agree with the shadow's wherever both see the operand (two arithmetics: close, not equal)
    res = m.activation_ranges()
    hidden = set(m.hidden_layers())
    assert {"base.base_layer", "base.level0", "base.level1"} <= hidden
    common = [n for n in res if n in r and n not in hidden]
    assert len(common) > 60
    for n in common:
        assert abs(res[n] - r[n]) <= 1e-3 * r[n] + 1e-6, (n, res[n], r[n])


def test_detector_first_batch_goes_through_the_guard(dev):
    """Detector(range_policy=...): 'raise' (default) refuses the first batch of a model whose activations would be clamped,
    'calibrate' calibrates on it and runs, 'off' runs unguarded; the guard runs once per load."""
    from centerfusiondetect3d_amd import Detector, getModel, centerfusion_middle_config, _lib
    from tests.golden import cases_dataset as cd
    H, W, B = 128, 160, 2
    cfg = centerfusion_middle_config((H, W))
    sd = _boost(cases.tuned_state_dict(radar=True, seed=0), "feat", 4096.0)      # (the feature map: every head reads it)
    rs = np.random.RandomState(3)
    frames = [rs.randint(0, 256, (450, 800, 3)).astype(np.uint8) for _ in range(B)]
    K3 = cd.NUSC_K * 0.5
    K3[2, 2] = 1.0
    calib = np.concatenate([K3, np.zeros((3, 1))], axis=1)
    infos = [dict(calib=calib.tolist(), camera_intrinsic=K3.tolist(), width=800, height=450) for _ in range(B)]
    sweeps = [cd._sweep(np.random.RandomState(60 + b), 120, max_z=70.0, lateral=0.7) for b in range(B)]

    def detector(policy):
        model = getModel(cfg)
        model.load_state_dict(sd)
        return Detector(cfg, model=model, device=dev, range_policy=policy)

    with pytest.raises(_lib.CfHipError, match=r"heads\.primary\.0"):
        detector("raise").run(frames, infos, sweeps)
    det = detector("calibrate")
    ret = det.run(frames, infos, sweeps)
    assert det.model._range_checked and det.model.activation_scales()["heads.primary.0"] < 16.0
    plans = dict(det.model._plans)
    ret2 = det.run(frames, infos, sweeps)                              # second batch: no second calibration (plans survive)
    assert all(det.model._plans.get(k) is v for k, v in plans.items())
    assert torch.equal(ret["post"], ret2["post"])
    off = detector("off").run(frames, infos, sweeps)
    assert not torch.equal(off["post"], ret["post"])                   # (the unguarded result is the clamped one)
    with pytest.raises(ValueError):
        Detector(cfg, model=getModel(cfg), device=dev, range_policy="maybe")
    # a running service: inputs drift after the first batch was checked - the periodic re-test of the resident buffers sees it
    model = getModel(cfg)
    model.load_state_dict(cases.tuned_state_dict(radar=True, seed=0))
    det = Detector(cfg, model=model, device=dev, range_policy="raise", range_check_every=2)
    det.run(frames, infos, sweeps)                                     # batch 1: first-batch guard, fine
    assert det.model._range_checked
    with torch.no_grad():
        for k, v in det.model.state_dict().items():                   # (stands for drifting inputs: the feature map grows 4096 x
            if k.startswith("ida_up.node_2.activation.0.") and k.endswith((".weight", ".bias")):   #  without a reload of the weights)
                v.mul_(4096.0)
        det.model.invalidate()
    with pytest.raises(_lib.CfHipError, match=r"check_resident_ranges.*heads\.primary\.0"):
        det.run(frames, infos, sweeps)                                 # batch 2: every second batch is re-tested
