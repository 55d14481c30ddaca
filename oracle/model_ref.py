"""Oracle: CenterFusion / CenterNet (DLA-34 + DCNv2 neck + heads) forward, CPU fp32.

TEST INFRASTRUCTURE - see oracle/__init__.py.  A functional restatement that walks
a reference-format ``state_dict`` (key names of SURVEY.md Appendix C); it follows

  * /root/reference/src/lib/model/networks/dla.py:18-41     Root
  * dla.py:105-118                                          Tree.forward
  * dla.py:147-161                                          BasicBlock.forward
  * dla.py:271-278                                          DLA.forward
  * dla.py:456-472                                          DeformConv.forward
  * dla.py:518-524, 553-559, 627-635                        IDAUp / DLAUp / img2feats
  * networks/detectHeads.py:59-132, 165-191                 heads
  * networks/base_model.py:67-106                           forward plumbing
  * networks/fusionModules.py:18-35                         ConcateCombiner

Pinned by tests/golden/model_*.npz (generated from the reference's own Python by
tests/golden/make_golden.py; the reference's deform_conv2d import is served by
oracle/dcn_ref.py there, so the DCN arithmetic itself stays "parity unpinned").
"""
import torch
import torch.nn.functional as F

from .dcn_ref import deform_conv2d
from . import frustum_ref

BN_EPS = 1e-5

PRIMARY_HEADS = ["heatmap", "reg", "widthHeight", "depth", "rotation", "dimension",
                 "amodal_offset"]
SECONDARY_HEADS = ["velocity", "nuscenes_att", "depth2", "rotation2"]


def head_spec(radar: bool, num_classes: int = 10):
    """heads / head_conv exactly as config/utils.py:69-166 derives them for nuScenes."""
    heads = {"heatmap": num_classes, "reg": 2, "widthHeight": 2, "depth": 1, "rotation": 8,
             "dimension": 3, "amodal_offset": 2, "nuscenes_att": 8, "velocity": 3}
    if radar:
        heads.update({"depth2": 1, "rotation2": 8})
    head_conv = {h: [256] for h in heads}
    if radar:
        for h in ("depth2", "rotation2", "velocity", "nuscenes_att"):
            head_conv[h] = [256, 256, 256]
    return heads, head_conv


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _block(sd, p, x, stride, residual=None):
    if residual is None:
        residual = x
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1))
    return F.relu(out + residual)


def _root(sd, p, xs):
    x = F.conv2d(torch.cat(xs, 1), sd[p + ".conv.weight"])
    return F.relu(_bn(sd, p + ".bn", x))  # root_residual is False everywhere (dla.py:172)


def _tree(sd, p, levels, x, stride, level_root, children=None):
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if (p + ".project.0.weight") in sd:
        residual = _bn(sd, p + ".project.1", F.conv2d(bottom, sd[p + ".project.0.weight"]))
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _block(sd, p + ".tree1", x, stride, residual)
        x2 = _block(sd, p + ".tree2", x1, 1)
        return _root(sd, p + ".root", [x2, x1, *children])
    # nested Tree ignores the residual handed to it (Tree.forward recomputes its own)
    x1 = _tree(sd, p + ".tree1", levels - 1, x, stride, False)
    children.append(x1)
    return _tree(sd, p + ".tree2", levels - 1, x1, 1, False, children)


def dla34_base(sd, x, p="base"):
    x = F.relu(_bn(sd, p + ".base_layer.1", F.conv2d(x, sd[p + ".base_layer.0.weight"], None, 1, 3)))
    y = []
    x = F.relu(_bn(sd, p + ".level0.1", F.conv2d(x, sd[p + ".level0.0.weight"], None, 1, 1)))
    y.append(x)
    x = F.relu(_bn(sd, p + ".level1.1", F.conv2d(x, sd[p + ".level1.0.weight"], None, 2, 1)))
    y.append(x)
    for lvl, levels, root in ((2, 1, False), (3, 2, True), (4, 2, True), (5, 1, True)):
        x = _tree(sd, f"{p}.level{lvl}", levels, x, 2, root)
        y.append(x)
    return y


def deform_node(sd, p, x):
    """DeformConv(activation=True): offset/mask conv -> DCNv2 -> BN -> ReLU."""
    om = F.conv2d(x, sd[p + ".conv_offset_mask.weight"], sd[p + ".conv_offset_mask.bias"], 1, 1)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    m = torch.sigmoid(m)
    x = deform_conv2d(x, offset, sd[p + ".weight"], sd[p + ".bias"], (1, 1), (1, 1), (1, 1), m)
    return F.relu(_bn(sd, p + ".activation.0", x))


def _up(sd, p, x):
    w = sd[p + ".weight"]
    f = w.shape[-1] // 2
    return F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=w.shape[0])


def _ida(sd, p, layers, startp, endp):
    for i in range(startp + 1, endp):
        j = i - startp
        layers[i] = _up(sd, f"{p}.up_{j}", deform_node(sd, f"{p}.proj_{j}", layers[i]))
        layers[i] = deform_node(sd, f"{p}.node_{j}", layers[i] + layers[i - 1])


def img2feats(sd, x):
    layers = dla34_base(sd, x)
    out = [layers[-1]]
    for i in range(len(layers) - 2 - 1):
        _ida(sd, f"dla_up.ida_{i}", layers, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    y = [out[i].clone() for i in range(3)]
    _ida(sd, "ida_up", y, 0, 3)
    return y[-1]


def _head(sd, p, x, n_hidden):
    x = F.relu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1, 1))
    idx = 2
    for _ in range(n_hidden - 1):
        x = F.relu(F.conv2d(x, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"]))
        idx += 2
    return F.conv2d(x, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"])


def sigmoid_depth(x):
    return 1.0 / (torch.sigmoid(x) + 1e-6) - 1.0


def forward(sd, x, pc_dep=None, calib=None, radar=True, frustum=True, K=100,
            max_pc_dist=60.0, num_classes=10, hp="detectHead_0", pc_hm_override=None):
    """model(x, pc_dep=, calib=) in eval mode -> [dict] (base_model.py:67-106).

    pc_hm_override (tests only): use this (B,3,H,W) frustum map instead of computing it - lets a
    float64 evaluation share the DISCRETE decisions (top-k order, depth gate) of the fp32 run, so
    that the comparison measures arithmetic error and not a flipped box."""
    heads, head_conv = head_spec(radar, num_classes)
    feat = img2feats(sd, x)
    y = {}
    primary = [h for h in heads if not (radar and h in SECONDARY_HEADS)]
    for h in primary:
        y[h] = _head(sd, f"{hp}.{h}", feat, len(head_conv[h]))
    y["heatmap"] = torch.clamp(torch.sigmoid(y["heatmap"]), min=1e-4, max=1 - 1e-4)
    y["depthMap"] = y["depth"]
    y["depth"] = sigmoid_depth(y["depth"])
    y["calib"] = calib
    if not radar:
        return [y]
    y["pc_hm_in"] = pc_dep[:, :1]
    assert frustum, "non-frustum middle fusion is outside the hot path"
    if pc_hm_override is not None:
        pc_hm = pc_hm_override.to(feat.dtype)
    else:
        pc_hm = frustum_ref.pc_frustum_heatmap(y, pc_dep, calib, K, max_pc_dist)
    y["pc_hm"] = pc_hm[:, 0:1]
    sec = torch.cat([feat, pc_hm], dim=1)
    for h in SECONDARY_HEADS:
        y[h] = _head(sd, f"{hp}.{h}", sec, len(head_conv[h]))
    y["pc_hm_out"] = pc_hm[:, :1]
    y["depthMap"] = y["depth2"]
    y["depth2"] = sigmoid_depth(y["depth2"])
    return [y]


def make_state_dict(radar=True, seed=0, num_classes=10, offset_std=0.01, offset_bias_std=1.0):
    """Seeded random reference-format state_dict (SURVEY.md §8(d) synthetic weights).

    Conv weights: kaiming-uniform-like U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BN stats and affine
    randomised; conv_offset_mask non-zero so the bilinear path is exercised; heatmap bias -4.6.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k, bias=False, scale=1.0):
        bound = scale / (ci * k * k) ** 0.5
        sd[name + ".weight"] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * bound
        if bias:
            sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    # He-style gain keeps activations O(1) through ~45 layers
    G = 2.0 ** 0.5 * 1.7
    conv("base.base_layer.0", 16, 3, 7, scale=G); bn("base.base_layer.1", 16)
    conv("base.level0.0", 16, 16, 3, scale=G); bn("base.level0.1", 16)
    conv("base.level1.0", 32, 16, 3, scale=G); bn("base.level1.1", 32)

    def block(p, ci, co):
        conv(p + ".conv1", co, ci, 3, scale=G); bn(p + ".bn1", co)
        conv(p + ".conv2", co, co, 3, scale=G); bn(p + ".bn2", co)

    def tree1(p, ci, co, root_dim, project=True):
        block(p + ".tree1", ci, co)
        block(p + ".tree2", co, co)
        conv(p + ".root.conv", co, root_dim, 1, scale=G); bn(p + ".root.bn", co)
        if project and ci != co:
            conv(p + ".project.0", co, ci, 1, scale=G); bn(p + ".project.1", co)

    tree1("base.level2", 32, 64, 128)
    for lvl, ci, co in ((3, 64, 128), (4, 128, 256)):
        p = f"base.level{lvl}"
        tree1(p + ".tree1", ci, co, 2 * co)
        tree1(p + ".tree2", co, co, 3 * co + ci)
    tree1("base.level5", 256, 512, 2 * 512 + 256)

    def dcn(p, ci, co):
        conv(p, co, ci, 3, bias=True)
        sd[p + ".conv_offset_mask.weight"] = torch.randn(27, ci, 3, 3, generator=g) * offset_std
        sd[p + ".conv_offset_mask.bias"] = torch.randn(27, generator=g) * offset_bias_std
        bn(p + ".activation.0", co)

    def up(p, c, f):
        k = 2 * f
        base = torch.zeros(k, k)
        fl = (k + 1) // 2
        cc = (2 * fl - 1 - fl % 2) / (2.0 * fl)
        for i in range(k):
            for j in range(k):
                base[i, j] = (1 - abs(i / fl - cc)) * (1 - abs(j / fl - cc))
        w = base.view(1, 1, k, k).repeat(c, 1, 1, 1)
        # weights are parameters (Appendix B.13): perturb so "hard-coded bilinear" fails
        sd[p + ".weight"] = w * (1 + 0.1 * torch.randn(c, 1, k, k, generator=g))

    chans = [64, 128, 256, 512]
    in_ch = list(chans)
    for i in range(3):
        j = -i - 2
        o = chans[j]
        srcs = in_ch[j:]
        for n in range(1, len(srcs)):
            dcn(f"dla_up.ida_{i}.proj_{n}", srcs[n], o)
            dcn(f"dla_up.ida_{i}.node_{n}", o, o)
            up(f"dla_up.ida_{i}.up_{n}", o, 2)
        in_ch[j + 1:] = [o for _ in in_ch[j + 1:]]
    for n, (ci, f) in enumerate(((128, 2), (256, 4)), start=1):
        dcn(f"ida_up.proj_{n}", ci, 64)
        dcn(f"ida_up.node_{n}", 64, 64)
        up(f"ida_up.up_{n}", 64, f)

    heads, head_conv = head_spec(radar, num_classes)
    for h, n_out in heads.items():
        cin = 67 if (radar and h in SECONDARY_HEADS) else 64
        p = f"detectHead_0.{h}"
        hc = head_conv[h]
        conv(p + ".0", hc[0], cin, 3, bias=True, scale=G)
        idx = 2
        for i in range(1, len(hc)):
            conv(f"{p}.{idx}", hc[i], hc[i - 1], 1, bias=True, scale=G)
            idx += 2
        conv(f"{p}.{idx}", n_out, hc[-1], 1, bias=True)
        if h == "heatmap":
            sd[f"{p}.{idx}.bias"].fill_(-4.6)
    return sd
