"""`DLASeg`: the CenterFusion / CenterNet inference model on the MI355X HIP path.

Drop-in boundary: same factory (`getModel(config)`, model/model.py:18-44), same
`state_dict` key names and shapes as the reference's DLASeg (model/networks/dla.py:571-635 +
base_model.py:30-53 + detectHeads.py:32-163), same call
`model(images, pc_hm=None, pc_dep=None, calib=None) -> [dict]` (base_model.py:67-106) with the
same keys, order, shapes and the `pc_hm_in` view of the caller's `pc_dep`
(detectHeads.py:172).  Eval mode only - training is outside the hot path.

Nothing else is shared with the reference's design.  The module is a parameter tree plus an
*execution plan*: at first call for a given (B,H,W) it lays out every intermediate NHWC buffer in
HBM once, folds BN into the conv weights, packs them into the implicit-GEMM layout, pre-builds the
argument block of every kernel launch, and from then on a forward pass is a flat list of
asynchronous launches on the current stream (no allocation except the returned head maps, no host
sync, hipGraph-capturable).  There is no CPU fallback: without libcfhip.so / a GPU it raises.
"""
import ctypes as C
import math
import threading
from typing import Dict, List

import torch
from torch import nn

from . import _lib, ops, packing
from ._lib import (ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP, ACT_RAW_AND_SIGDEPTH, LAYOUT_NHWC,
                   LAYOUT_NCHW, LAYOUT_NHWC_SPLIT_BF16)
from .packing import Source

SECONDARY_HEADS = ["velocity", "nuscenes_att", "depth2", "rotation2"]   # detectHeads.py:146-153
CHANNELS = [16, 32, 64, 128, 256, 512]                                    # dla.py:303-307


class _Scope(nn.Module):
    """Empty container: only exists so parameters get the reference's dotted names."""


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer=False):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Scope())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


# ----------------------------------------------------------------------------- parameter spec
def _conv_init(co, ci, k):
    w = torch.empty(co, ci, k, k)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    return w


def _param_spec(config) -> List[tuple]:
    """[(name, tensor, is_buffer)] in the reference's registration order."""
    spec = []

    def conv(name, co, ci, k, bias=False):
        spec.append((name + ".weight", _conv_init(co, ci, k), False))
        if bias:
            bound = 1 / math.sqrt(ci * k * k)
            spec.append((name + ".bias", torch.empty(co).uniform_(-bound, bound), False))

    def bn(name, c):
        spec.append((name + ".weight", torch.ones(c), False))
        spec.append((name + ".bias", torch.zeros(c), False))
        spec.append((name + ".running_mean", torch.zeros(c), True))
        spec.append((name + ".running_var", torch.ones(c), True))
        spec.append((name + ".num_batches_tracked", torch.tensor(0, dtype=torch.long), True))

    conv("base.base_layer.0", 16, 3, 7); bn("base.base_layer.1", 16)
    conv("base.level0.0", 16, 16, 3); bn("base.level0.1", 16)
    conv("base.level1.0", 32, 16, 3); bn("base.level1.1", 32)

    def block(p, ci, co):
        conv(p + ".conv1", co, ci, 3); bn(p + ".bn1", co)
        conv(p + ".conv2", co, co, 3); bn(p + ".bn2", co)

    def tree1(p, ci, co, root_dim):
        block(p + ".tree1", ci, co)
        block(p + ".tree2", co, co)
        conv(p + ".root.conv", co, root_dim, 1); bn(p + ".root.bn", co)
        if ci != co:
            conv(p + ".project.0", co, ci, 1); bn(p + ".project.1", co)

    tree1("base.level2", 32, 64, 128)
    for lvl, ci, co in ((3, 64, 128), (4, 128, 256)):
        tree1(f"base.level{lvl}.tree1", ci, co, 2 * co)
        tree1(f"base.level{lvl}.tree2", co, co, 3 * co + ci)
    tree1("base.level5", 256, 512, 2 * 512 + 256)

    def dcn(p, ci, co):
        stdv = 1.0 / math.sqrt(ci * 9)
        spec.append((p + ".weight", torch.empty(co, ci, 3, 3).uniform_(-stdv, stdv), False))
        spec.append((p + ".bias", torch.zeros(co), False))
        bn(p + ".activation.0", co)
        spec.append((p + ".conv_offset_mask.weight", torch.zeros(27, ci, 3, 3), False))
        spec.append((p + ".conv_offset_mask.bias", torch.zeros(27), False))

    def up(p, c, f):
        k = 2 * f
        fl = math.ceil(k / 2)
        cc = (2 * fl - 1 - fl % 2) / (2.0 * fl)
        w = torch.zeros(c, 1, k, k)
        for i in range(k):
            for j in range(k):
                w[:, 0, i, j] = (1 - abs(i / fl - cc)) * (1 - abs(j / fl - cc))
        spec.append((p + ".weight", w, False))

    def ida(p, o, srcs, fs):
        for n in range(1, len(srcs)):
            dcn(f"{p}.proj_{n}", srcs[n], o)
            up(f"{p}.up_{n}", o, fs[n])
            dcn(f"{p}.node_{n}", o, o)

    chans = CHANNELS[2:]
    in_ch = list(chans)
    for i in range(3):
        j = -i - 2
        ida(f"dla_up.ida_{i}", chans[j], in_ch[j:], [1] + [2] * (len(in_ch[j:]) - 1))
        in_ch[j + 1:] = [chans[j]] * len(in_ch[j + 1:])
    ida("ida_up", 64, [64, 128, 256], [1, 2, 4])

    radar_middle = bool(config.DATASET.RADAR_PC) and config.MODEL.FUSION_STRATEGY == "middle"

    def head_conv_layer(name, co, ci, k, zero_bias, bias_fill=None):
        spec.append((name + ".weight", _conv_init(co, ci, k), False))
        bound = 1 / math.sqrt(ci * k * k)
        b = torch.zeros(co) if zero_bias else torch.empty(co).uniform_(-bound, bound)
        if bias_fill is not None:
            b.fill_(bias_fill)
        spec.append((name + ".bias", b, False))

    for h, n_out in config.heads.items():
        hc = list(config.head_conv[h])
        cin = 67 if (radar_middle and h in SECONDARY_HEADS) else 64
        p = f"detectHead_0.{h}"
        zero = h != "heatmap"                     # initConv2dWeights: bias 0 on non-heatmap heads
        head_conv_layer(p + ".0", hc[0], cin, 3, zero)
        idx = 2
        for i in range(1, len(hc)):
            head_conv_layer(f"{p}.{idx}", hc[i], hc[i - 1], 1, zero)
            idx += 2
        head_conv_layer(f"{p}.{idx}", n_out, hc[-1], 1, zero,
                        -4.6 if h == "heatmap" else None)      # detectHeads.py:93
    return spec


# ----------------------------------------------------------------------------- execution plan
_SIDE_STREAMS = {}    # (device, caller stream id) -> probed side streams, LRU of _SIDE_KEYS keys per PROCESS
_SIDE_KEYS = 16
_SIDE_LOCK = threading.Lock()
_CAPTURE_LOCK = threading.RLock()   # held while a stream captures AND while a captured graph is destroyed (see _forward_graph)
_PROBE_US = 300       # length of one probe spin; two of them take ~1x this when concurrent, ~2x when serialised
_PROBE_LOG = []       # [(device, sid, n, chosen indices, [(i, j, ms)])]: what the probes measured (tests / DESIGN)


def _concurrent(lib, a, b, us=_PROBE_US):
    """True if a `cf_spin_us` on stream `a` and one on stream `b`, issued back to back, overlap in time.  HIP maps
    streams onto a few hardware queues; two streams that share one run their kernels strictly one after the other
    (model.streams = 2 then measures 10.5 instead of 8.5 ms per bs=16 step).  -> (bool, ms from first start to last end)"""
    e0, e1a, e1b = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    gate = torch.cuda.Event()
    gate.record(a)
    b.wait_event(gate)                       # neither spin starts before both streams have drained to here
    e0.record(a)
    _lib.check(lib.cf_spin_us(us, a.cuda_stream), "cf_spin_us")
    e1a.record(a)
    _lib.check(lib.cf_spin_us(us, b.cuda_stream), "cf_spin_us")
    e1b.record(b)
    e1a.synchronize()
    e1b.synchronize()
    ms = max(e0.elapsed_time(e1a), e0.elapsed_time(e1b))
    return ms < 1.6e-3 * us, ms


def _pick_streams(device, cur, n, pool):
    """Extend `pool` to n streams that run concurrently with each other and with the caller's stream `cur` (the
    two-lane neck issues on `cur` and on pool[0]; the heads, the decode and `Detector.run_pipelined`'s consumer run on
    `cur` beside whatever the last pool stream feeds).  Candidates are taken from torch's stream pool
    one at a time and kept only if a pair of spin kernels says they overlap with everything chosen so far - whatever
    else of the process (RCCL's communicator stream, other models, user streams) already sits on the hardware queues.
    Falls back to plain creation order if no concurrent set turns up within 12 candidates (still correct, only slower)."""
    lib = _lib.load()
    tried, log = [], []
    with _CAPTURE_LOCK:                       # the probe synchronises: never beside another thread's stream capture
        torch.cuda.synchronize(device)
        while len(pool) < n and len(tried) < 12:
            c = torch.cuda.Stream(device)
            tried.append(c)
            ok = True
            for x in [cur] + pool:
                good, ms = _concurrent(lib, x, c)
                log.append((len(tried) - 1, "caller" if x is cur else pool.index(x), round(ms, 3)))
                if not good:
                    ok = False
                    break
            if ok:
                pool.append(c)
    fallback = len(pool) < n
    for c in tried:                           # not enough concurrent ones: take what was created, in order
        if len(pool) >= n:
            break
        if c not in pool:
            pool.append(c)
    _PROBE_LOG.append((str(device), int(cur.cuda_stream), n, fallback, log))
    del _PROBE_LOG[:-32]
    return pool


def _side_streams(device, sid, n):
    """The n side streams that work issued on caller stream `sid` of `device` forks onto.  One set per process and caller
    stream, not per model or plan, chosen ONCE by a probe (`_pick_streams`) instead of by creation order: HIP spreads
    streams over a few hardware queues, whether two side streams share a queue decides whether their kernels overlap
    at all (8.5 vs 10.5 ms per bs=16 step), and under torchrun RCCL has taken streams before the first model exists.
    During a graph capture nothing may synchronise: fresh streams are forked as they come (the replay's placement is
    the graph executor's, not these streams')."""
    key = (str(device), int(sid))
    with _SIDE_LOCK:                              # the common case: the set exists
        pool = _SIDE_STREAMS.get(key)
        if pool is not None and len(pool) >= n:
            _SIDE_STREAMS[key] = _SIDE_STREAMS.pop(key)          # re-inserted last: dict order is the LRU order
            return pool[:n]
    # streams are missing: the probe synchronises the device, which must not happen beside another thread's stream
    # capture - lock order is _CAPTURE_LOCK, then _SIDE_LOCK, everywhere (a capturing thread holds the first already)
    with _CAPTURE_LOCK, _SIDE_LOCK:
        pool = _SIDE_STREAMS.pop(key, [])
        if len(pool) < n:
            if torch.cuda.is_current_stream_capturing():
                while len(pool) < n:
                    pool.append(torch.cuda.Stream(device))
            else:
                pool = _pick_streams(device, torch.cuda.current_stream(device), n, pool)
        _SIDE_STREAMS[key] = pool
        while len(_SIDE_STREAMS) > _SIDE_KEYS:
            _SIDE_STREAMS.pop(next(iter(_SIDE_STREAMS)))
        return pool[:n]


class _Plan:
    """Buffers + pre-built launch list for one (B, H, W, device)."""

    def __init__(self, model: "DLASeg", B, H, W, device, part="all", feat=None, feat_in=None):
        """part: "all" (one plan per forward), or the two halves of the split forward (DLASeg.streams > 1):
        "trunk" = backbone + neck of a sub-batch, its last DCN writing the feature map (and its split-bf16 copy)
        into the caller's `feat` / `feat_in` slices; "heads" = everything behind the feature map for the WHOLE
        batch, reading the full `feat` / `feat_in` buffers the trunks filled."""
        self.B, self.H, self.W, self.device = B, H, W, device
        self.part = part
        self.lib = _lib.load()
        self.steps = []          # (fn, args...) tuples executed in order (stream appended at run)
        self.lanes = []          # per step: 0 = the caller's stream, 1 = the plan's side stream (see ida())
        self.lane = 0
        self.n_events = 0
        # two-lane issue of the neck for small batches (a launch cannot fill the chip there); never with the timed /
        # graph paths, which want one stream
        # (a trunk sub-batch of the two-stream forward keeps round 5's limit of 4 frames: its side lane would be a third / fourth
        #  stream beside the other trunk, and the event traffic of that costs more than the overlap gives)
        self.use_lanes = bool(model.lanes) and part != "heads" and \
            B * (H // 4) * (W // 4) <= (model.lanes_max_frames if part == "all" else min(4, model.lanes_max_frames)) * 112 * 200
        self.keep = []           # keeps arg blocks / buffers alive
        self.bytes = 0
        self.step_index = {}     # conv name -> index in self.steps
        self.step_flops = {}     # conv name -> algorithmic FLOPs of that launch (2*MACs)
        self.timed = {}          # step index -> [(start_event, end_event)] filled while timing is on
        self.inputs = {}         # layer name -> the resident NHWC tensors that layer's GEMM reads (DLASeg.activation_ranges)
        self.hidden = set()      # layer names with operands that never reach HBM in this plan (fused intermediates)
        pk = model._packed
        cfg = model.config
        heads, head_conv = dict(cfg.heads), {k: list(v) for k, v in cfg.head_conv.items()}
        radar = model.isRadarEnabled and model.fusionStrategy == "middle"
        K = int(cfg.MODEL.K)

        def buf(*shape, dtype=torch.float32):
            t = torch.empty(shape, device=device, dtype=dtype)
            self.bytes += t.numel() * t.element_size()
            self.keep.append(t)
            return t

        def conv(name, srcs, h, w, act=ACT_RELU, residual=None, out=None, out_stride=None,
                 out_offset=0, layout=LAYOUT_NHWC, out2=None, strides=None, precise=None):
            # everything that feeds the DCN neck sums in two levels (cf_gemm.hip: PRECISE); the
            # heads come after it, their rounding is not amplified
            precise = (model.precise and not name.startswith("heads.")) if precise is None else precise
            pc = pk[name]
            ho = (h + 2 * pc.pad - pc.kh) // pc.stride + 1
            wo = (w + 2 * pc.pad - pc.kh) // pc.stride + 1
            if out is None:
                out = buf(B, ho, wo, pc.n)
            a = ops.conv_args(pc, srcs, strides or [s.shape[-1] for s in srcs], B, h, w, out,
                              out_stride or pc.n, act, residual,
                              residual.shape[-1] if residual is not None else 0, layout, out2,
                              out_offset, precise, in_scale=model._scale(name) if pc.out_scale > 0 else None)
            self.keep.append(a)
            self.inputs[name] = list(srcs)
            self.step_index[name] = len(self.steps)
            self.step_flops[name] = 2.0 * B * ho * wo * pc.n * (pc.kh * pc.kh * sum(
                int(c) for c in pc.real_cin))
            fn = self.lib.cf_conv2d_fused
            if pc.out_scale > 0:
                fn = self.lib.cf_conv3x3_f16x3 if (pc.patch and model.conv_patch) else self.lib.cf_conv2d_f16x3
            self.add_step((fn, C.byref(a)))
            return out, a

        pooled = {}

        def pool(x):
            # (a two-level Tree pools its input for its own Root AND its first sub-tree pools the same tensor again,
            #  dla.py:96,107 at both nesting levels: one launch serves both)
            if x.data_ptr() in pooled:
                return pooled[x.data_ptr()]
            _, h, w, c = x.shape
            o = buf(B, h // 2, w // 2, c)
            self.add_step((self.lib.cf_maxpool2x2, x.data_ptr(), o.data_ptr(), B, h, w, c))
            pooled[x.data_ptr()] = o
            return o

        def block(p, x, residual, pooled=None):
            _, h, w, _ = x.shape
            t, _ = conv(p + ".conv1", [x], h, w)
            _, ho, wo, _ = t.shape
            if pooled is not None:
                # conv2 + the Tree's project of the pooled input in one step (weights packed together: _prepare.tree1)
                pc = pk[p + ".conv2"]
                o = buf(B, ho, wo, pc.n)
                a = ops.conv_args(pc, [t, pooled], [t.shape[-1], pooled.shape[-1]], B, ho, wo, o, pc.n, ACT_RELU, None, 0,
                                  LAYOUT_NHWC, None, 0, False, in_scale=model._scale(p + ".conv2"))   # (one pre-scale for both parts)
                self.inputs[p + ".conv2"], self.inputs[p[:-len(".tree1")] + ".project"] = [t], [pooled]
                ch = (C.c_int32 * 2)(*[int(c) for c in pc.real_cin])
                self.keep += [a, ch]
                name = p + ".conv2+project"
                self.step_index[name] = len(self.steps)
                self.step_flops[name] = 2.0 * B * ho * wo * pc.n * (9 * pc.real_cin[0] + pc.real_cin[1])
                if model.conv_patch:
                    self.add_step((self.lib.cf_conv3x3_proj_f16x3, C.byref(a), ch))
                else:
                    self.add_step((self.lib.cf_conv2d_f16x3, C.byref(a)))
                return o
            o, _ = conv(p + ".conv2", [t], ho, wo, residual=residual if residual is not None else x)
            return o

        def tree(p, levels, x, stride, level_root, children=None):
            children = [] if children is None else children
            bottom = pool(x) if stride > 1 else x
            proj_fused = levels == 1 and getattr(pk[p + ".tree1.conv2"], "proj_k", 0) > 0
            if proj_fused:
                residual = None
            elif (p + ".project") in pk:
                _, h, w, _ = bottom.shape
                residual, _ = conv(p + ".project", [bottom], h, w, act=ACT_NONE)
            else:
                residual = bottom
            if level_root:
                children.append(bottom)
            if levels == 1:
                x1 = block(p + ".tree1", x, residual, pooled=bottom if proj_fused else None)
                pc2, pcr = pk[p + ".tree2.conv2"], pk[p + ".root"]
                if (model.root_fuse and model.conv_patch and pc2.out_scale > 0 and pcr.out_scale > 0
                        and getattr(pc2, "patch", False) and pc2.stride == 1
                        and (not children or model.root_fuse_children)):
                    # tree2.conv2 and the Root as ONE step (cf_conv3x3_root_f16x3): x2 is never written where a workgroup
                    # holds every channel of its pixels (64 / 128 / 256 channels: levels 2-4; children are read from HBM
                    # inside the launch); the library runs the two launches for every other shape, bit-identical either way
                    _, h, w, _ = x1.shape
                    t, _ = conv(p + ".tree2.conv1", [x1], h, w)
                    x2, o = buf(B, h, w, pc2.n), buf(B, h, w, pcr.n)
                    a2 = ops.conv_args(pc2, [t], [t.shape[-1]], B, h, w, x2, pc2.n, ACT_RELU, x1, x1.shape[-1],
                                       LAYOUT_NHWC, None, 0, False, in_scale=model._scale(p + ".tree2.conv2"))
                    rsrcs = [x2, x1, *children]
                    ar = ops.conv_args(pcr, rsrcs, [s_.shape[-1] for s_ in rsrcs], B, h, w, o, pcr.n, ACT_RELU, None, 0,
                                       LAYOUT_NHWC, None, 0, False, in_scale=model._scale(p + ".root"))
                    self.inputs[p + ".tree2.conv2"], self.inputs[p + ".root"] = [t], [x1, *children]
                    self.hidden.add(p + ".root")             # (x2 stays on the chip)
                    rch = (C.c_int32 * len(rsrcs))(*[int(c) for c in pcr.real_cin])
                    self.keep += [a2, ar, rch]
                    name = p + ".tree2.conv2+root"
                    self.step_index[name] = len(self.steps)
                    self.step_flops[name] = 2.0 * B * h * w * (pc2.n * 9 * sum(int(c) for c in pc2.real_cin)
                                                               + pcr.n * sum(int(c) for c in pcr.real_cin))
                    self.add_step((self.lib.cf_conv3x3_root_f16x3, C.byref(a2), C.byref(ar), rch))
                    return o
                x2 = block(p + ".tree2", x1, None)
                _, h, w, _ = x2.shape
                o, _ = conv(p + ".root", [x2, x1, *children], h, w)
                return o
            x1 = tree(p + ".tree1", levels - 1, x, stride, False)
            children.append(x1)
            return tree(p + ".tree2", levels - 1, x1, 1, False, children)

        def dcn_node(p, x, out=None, feat_producer=False):
            _, h, w, c = x.shape
            om = buf(B, h, w, 32)
            conv(p + ".conv_offset_mask", [x], h, w, act=ACT_NONE, out=om, out_stride=32)
            pd = pk[p]
            o = buf(B, h, w, pd.n) if out is None else out
            ws = None
            if pd.out_scale > 0:
                nbytes = self.lib.cf_dcn_v2_workspace_bytes(B, h, w, pd.c, pd.n_pad)
                ws = buf(nbytes, dtype=torch.uint8) if nbytes else None
            a = ops.dcn_args(pd, x, om, 32, B, h, w, o, pd.n, ACT_RELU, precise=model.precise, workspace=ws,
                             in_scale=model._scale(p) if pd.out_scale > 0 else None)
            self.keep.append(a)
            if feat_producer:
                self.feat_producer = a                       # the DCN that writes the feature map (the last node of ida_up)
            self.inputs[p] = [x]
            self.step_index[p] = len(self.steps)
            self.step_flops[p] = 2.0 * B * h * w * pd.n * 9 * pd.c
            self.add_step((self.lib.cf_dcn_v2_f16x3 if pd.out_scale > 0 else self.lib.cf_dcn_v2_fused, C.byref(a)))
            return o

        def ida(p, layers, startp, endp, final_out=None, feat=False):
            """IDAUp.forward (dla.py:518-524).  The projections of one IDA level read maps that all exist when the level
            starts and do not depend on each other or on the nodes, so with `self.use_lanes` they are issued on a side
            stream (offset conv + DCN per projection) while the caller's stream runs the node chain
            upsample+skip -> offset conv -> DCN, waiting for projection j right before it consumes it.  Small
            batches only: there a single launch cannot fill the chip and the two chains overlap (bit-identical)."""
            projs = {}
            if self.use_lanes:
                self.ctl("rec", 0, ev0 := self.new_event())      # everything the projections read is complete here
                self.ctl("wait", 1, ev0)
                self.lane = 1
                for i in range(startp + 1, endp):
                    projs[i] = dcn_node(f"{p}.proj_{i - startp}", layers[i])
                    self.ctl("rec", 1, ev := self.new_event())
                    projs[i] = (projs[i], ev)
                self.lane = 0
            for i in range(startp + 1, endp):
                j = i - startp
                if self.use_lanes:
                    proj, ev = projs[i]
                    self.ctl("wait", 0, ev)
                else:
                    proj = dcn_node(f"{p}.proj_{j}", layers[i])
                wk, f = pk[f"{p}.up_{j}"]
                _, h, w, c = proj.shape
                summed = buf(B, h * f, w * f, c)          # up(proj(x)) + skip, fused
                self.add_step((self.lib.cf_upsample_dw, proj.data_ptr(), wk.data_ptr(),
                               layers[i - 1].data_ptr(), summed.data_ptr(), B, h, w, c, f))
                layers[i] = dcn_node(f"{p}.node_{j}", summed, out=final_out if i == endp - 1 else None,
                                     feat_producer=feat and i == endp - 1)

        bf = model._heads_bf()                             # fused split-bf16 head launches (False: the exact-fp32 layer-by-layer heads)
        h4, w4 = H // 4, W // 4
        self.h4, self.w4 = h4, w4
        self.in_step = None
        self.stem = None
        self.feat_producer = None    # argument block of the DCN whose output is the feature map (set by dcn_node)
        self.debug = {}
        if part != "heads":
            # ---- backbone
            self.in_step = len(self.steps)
            self.add_step(None)                            # first step reads the images: patched per call
            if "base.stem" in pk:
                # base_layer + level0 + level1 in one launch; the full-resolution maps stay in LDS
                y0, y1 = None, buf(B, H // 2, W // 2, 32)
                # ... and the level-2 Tree's 2x2 max-pool of that map (dla.py:96) comes out of the same launch
                y1p = buf(B, H // 4, W // 4, 32) if model.stem_pool else None
                if y1p is not None:
                    pooled[y1.data_ptr()] = y1p
                stem_layers = ("base.base_layer", "base.level0", "base.level1")
                self.stem = ops.stem_args(pk["base.stem"], None, y1, shape=(B, 3, H, W), out_pool=y1p,
                                          in_scales=[model._scale(n) for n in stem_layers])
                self.keep.append(self.stem)
                self.hidden.update(stem_layers)              # (the image is the caller's, the two maps stay in LDS)
                self.step_index["base.stem"] = self.in_step
                self.step_flops["base.stem"] = 2.0 * B * H * W * (16 * 147 + 16 * 144 + 32 * 144 / 4)
            else:
                self.x4 = buf(B, H, W, 4)
                t, _ = conv("base.base_layer", [self.x4], H, W)
                y0, _ = conv("base.level0", [t], H, W)
                y1, _ = conv("base.level1", [y0], H, W)
            layers = [y0, y1]
            x = y1
            for lvl, levels, root in ((2, 1, False), (3, 2, True), (4, 2, True), (5, 1, True)):
                x = tree(f"base.level{lvl}", levels, x, 2, root)
                layers.append(x)
            self.debug = {f"y{i}": t for i, t in enumerate(layers) if t is not None}   # NHWC stage outputs (tests only)
            # ---- DLA-up + IDA-up neck
            out = [layers[-1]]
            for i in range(len(layers) - 2 - 1):
                ida(f"dla_up.ida_{i}", layers, len(layers) - i - 2, len(layers))
                out.insert(0, layers[-1])
            for i, t in enumerate(out):
                self.debug[f"up{i}"] = t
            y = out[:3]
            ida("ida_up", y, 0, 3, final_out=feat, feat=True)
            feat = y[-1]
            if bf and model._mx_active:
                # heads' first layer on fp16 + FP6 (cf_head_fused mx = 1): the 272-byte rows it stages, one pass over the
                # fp32 feature map of this (sub-)batch
                if feat_in is None:
                    feat_in = buf(B, h4, w4, packing.MX_ROW, dtype=torch.uint8)
                # ... written by the epilogue of the DCN that produces the map (f16x3 kernel, no K split at this size); a
                # separate pass over the fp32 map only if that kernel is not in use
                # (the feature map's DCN runs WITHOUT a K split whenever it also writes the heads' rows - the two exclude each
                #  other - so on maps small enough for the split, <= 2048 pixels per image, the summation order of that one
                #  layer depends on pack_mx_fused / heads_mx: same arithmetic, rounding differs)
                pr = self.feat_producer
                if pr is not None and pr.out_scale > 0 and pr.N == 64 and pr.N_pad == 64 and bool(model.pack_mx_fused):
                    pr.out_mx = feat_in.data_ptr()
                    pr.mx_scale = model._feat_scale
                    pr.workspace = None
                else:
                    self.step_index["feat.pack_mx"] = len(self.steps)
                    self.add_step((self.lib.cf_pack_feat_mx_scaled, feat.data_ptr(), 64, feat_in.data_ptr(), C.c_long(B * h4 * w4),
                                   C.c_float(model._feat_scale)))
            elif bf:
                # the split-bf16 copy of the feature map the heads read is written by the epilogue of the DCN
                # that produces it (f16x3 kernel); a separate split pass only if that kernel is not in use
                if feat_in is None:
                    feat_in = buf(B, h4, w4, 2, 64, dtype=torch.bfloat16)
                pr = self.feat_producer
                if pr is not None and pr.out_scale > 0 and pr.N == 64:
                    pr.out_split_bf16, pr.split_stride = feat_in.data_ptr(), 64
                    pr.workspace = None                    # (the split output and a K-split reduction exclude each other)
                else:
                    self.add_step((self.lib.cf_split_bf16, feat.data_ptr(), feat_in.data_ptr(), B * h4 * w4, 64, 64, 64))
            else:
                feat_in = feat
        self.feat, self.feat_in = feat, feat_in
        self.primary, self.radar, self.K = [], False, K
        self.outs: Dict[str, List] = {}
        self.tails = {}
        if part == "trunk":
            return

        # ---- heads.  Per-call output tensors are patched into these arg blocks (self.outs).
        primary = [h for h in heads if not (radar and h in SECONDARY_HEADS)]
        self.primary = primary
        self.radar = radar
        M4 = B * h4 * w4
        if feat is not None:                                # the fp32 feature map both head groups read (as mx / split-bf16 / fp32)
            self.inputs["heads.primary.0"] = [feat]
            if radar:
                self.inputs["heads.secondary.0"] = [feat]

        def hconv(name, srcs, strides, out_c=None, out=None, out_offset=0, act=ACT_RELU):
            """One hidden head layer of the exact-fp32 heads (model.heads_bf16 = False): fp32 NHWC, cf_conv2d_fused."""
            pc = pk[name]
            if out is None:
                out = buf(B, h4, w4, out_c)
            stride = out.shape[-1]
            a = ops.conv_args(pc, srcs, strides, B, h4, w4, out, stride, act, None, 0, LAYOUT_NHWC, None, out_offset, False, 4)
            self.keep.append(a)
            self.step_index[name] = len(self.steps)
            self.step_flops[name] = 2.0 * M4 * pc.n * (pc.kh * pc.kh * sum(int(c) for c in pc.real_cin))
            self.add_step((self.lib.cf_conv2d_fused, C.byref(a)))
            return out

        def head_out(h, src, src_stride):
            act = ACT_SIGMOID_CLAMP if h == "heatmap" else (
                ACT_RAW_AND_SIGDEPTH if h in ("depth", "depth2") else ACT_NONE)
            pc = pk[f"heads.{h}.out"]
            a = ops.conv_args(pc, [src], [src_stride], B, h4, w4, src, 0, act, None, 0,
                              LAYOUT_NCHW, src if act == ACT_RAW_AND_SIGDEPTH else None, 0, False)
            self.keep.append(a)
            self.step_index[f"heads.{h}.out"] = len(self.steps)
            self.step_flops[f"heads.{h}.out"] = 2.0 * M4 * pc.n * 256
            self.add_step((self.lib.cf_conv2d_fused, C.byref(a)))
            self.outs[h] = a

        hs = 256 * len(primary)
        hid = None if bf else hconv("heads.primary.0", [feat_in], [64], out_c=hs)

        def act_of(h):
            return ACT_SIGMOID_CLAMP if h == "heatmap" else (
                ACT_RAW_AND_SIGDEPTH if h in ("depth", "depth2") else ACT_NONE)

        def fused_heads(name, names, srcs, strides, pkname=None):
            """One cf_head_fused launch: 3x3 + ReLU + tail for sibling heads, hidden never in HBM."""
            hd = [dict(pk[pkname or name][h], act=act_of(h)) for h in names]
            f = ops.head_fused_args(srcs, strides, hd[0].get("slots"), hd[0].get("k_pad", 0), B, h4, w4, hd)
            self.keep.append(f)
            for n, h in enumerate(names):
                self.tails[h] = (f.tail, n)
            self.step_index[name] = len(self.steps)
            self.step_flops[name] = sum(
                2.0 * M4 * 256 * (9 * sum(d["real_cin"]) + 256 * len(d["w_hidden"]) + heads[h])
                for h, d in zip(names, hd))
            self.add_step((self.lib.cf_head_fused, C.byref(f)))

        fuse_all = bf
        # Two lanes for the decoder's index kernels (model.heads_lanes, fused heads): behind the primary launch the side stream
        # runs the decoder's NMS + top-k (handed to decode.py through the heat map tensor, see run()) beside the frustum
        # path and the secondary launch, instead of alone on the chip behind the last head launch.  (Splitting the primary
        # launch in two so that the frustum path's top-k could go there as well costs the head launches more than both
        # top-k passes take: DESIGN.md section 9.)
        self.peaks_step = None
        split = fuse_all and bool(model.heads_lanes) and primary[0] == "heatmap"
        def peaks_lane():
            """the side lane: starts behind whatever the caller's stream has issued so far, runs the decoder's NMS + top-k"""
            ev_a = self.new_event()
            self.ctl("rec", 0, ev_a)
            self.lane = 1
            self.ctl("wait", 1, ev_a)
            self.pk_ws = buf(max(1, self.lib.cf_topk_workspace_bytes_nms(B, heads["heatmap"], h4, w4, K)), dtype=torch.uint8)
            self.peaks_step = len(self.steps); self.add_step(None)
            self.ev_peaks = self.new_event()
            self.ctl("rec", 1, self.ev_peaks)
            self.lane = 0

        if split:
            self.use_lanes = True
            fused_heads("tails.primary", primary, [feat_in], [64])
            # radar: the lane starts behind the frustum chain (below), not behind the primary launch - its chip-wide NMS pass
            # beside the chain's slice top-k tripled that kernel's time (41 vs 14 us) on the one path everything waits for
            if not (radar and model.peaks_behind_frustum):
                peaks_lane()
        elif fuse_all:
            fused_heads("tails.primary", primary, [feat_in], [64])
        else:
            for h in primary:
                head_out(h, hid, hs)
        if radar:
            self.pc_hm4 = None if bf else buf(B, h4, w4, 4)
            self.pc_hm8 = buf(B, h4, w4, 2, 8, dtype=torch.bfloat16) if bf else None
            self.tk_scores = buf(B, K)
            self.tk_inds = buf(B, K, dtype=torch.int32)
            self.tk_cls = buf(B, K, dtype=torch.int32)
            self.tk_ws = buf(max(1, self.lib.cf_topk_workspace_bytes(B, K)), dtype=torch.uint8)
            # top-k of the raw heat map -> association (pointcloud.py:347-392): cf_topk_frustum (two launches: the merge of the slice
            # lists runs in the association kernel's prologue) or, model.frustum_fused = False, cf_topk_peaks + cf_frustum_assoc (three)
            self.topk_step = None
            if not model.frustum_fused:
                self.topk_step = len(self.steps); self.add_step(None)
            self.frustum_step = len(self.steps); self.add_step(None)
            ss = 256 * len(SECONDARY_HEADS)
            if split and self.peaks_step is None:
                peaks_lane()
            if fuse_all:
                fused_heads("tails.secondary", SECONDARY_HEADS, [feat_in, self.pc_hm8], [64, 8])
                if split:
                    self.ctl("wait", 0, self.ev_peaks)
                return
            s1 = hconv("heads.secondary.0", [feat_in, self.pc_hm4], [64, 4], out_c=ss)
            s2 = buf(B, h4, w4, ss)
            for n, h in enumerate(SECONDARY_HEADS):
                hconv(f"heads.{h}.2", [s1], [ss], out=s2, out_offset=256 * n)
                hconv(f"heads.{h}.4", [s2], [ss], out=s1, out_offset=256 * n)
                head_out(h, s1, ss)
        elif split:
            self.ctl("wait", 0, self.ev_peaks)

    # ------------------------------------------------------------------------------------------
    def add_step(self, step):
        self.steps.append(step)
        self.lanes.append(self.lane)

    def ctl(self, op, lane, ev):
        """Cross-lane ordering: ("rec" | "wait", lane, event id)."""
        self.steps.append((op, ev))
        self.lanes.append(lane)

    def new_event(self):
        self.n_events += 1
        return self.n_events - 1

    def _launch(self, st):
        """Issue the plan's steps.  With lanes (and outside a stream capture) a step runs on the caller's stream (lane 0) or
        on the plan's side stream (lane 1), ordered by the ("rec" | "wait", event) control steps; otherwise everything runs
        in list order on the caller's stream (the list order is a valid sequential order).  Timed steps (model.time_launch)
        are bracketed by HIP events recorded on the stream the step runs on."""
        lanes_on = self.use_lanes and not torch.cuda.is_current_stream_capturing()
        timed = self.timed
        if not lanes_on:
            for i, step in enumerate(self.steps):
                if isinstance(step[0], str):
                    continue                                   # one stream: program order is the dependency order
                ev = timed.get(i) if timed else None
                if ev is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                rc = step[0](*step[1:], st)
                if ev is not None:
                    e1.record()
                    ev.append((e0, e1))
                if rc != 0:
                    _lib.check(rc, step[0].__name__)
            return
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = _side_streams(self.device, cur.cuda_stream, 1)[0]
            self._events = [torch.cuda.Event() for _ in range(self.n_events)]
        streams = (cur, self._side)
        ptrs = (st, self._side.cuda_stream)
        for i, (step, lane) in enumerate(zip(self.steps, self.lanes)):
            op = step[0]
            if op == "rec":
                self._events[step[1]].record(streams[lane])
            elif op == "wait":
                streams[lane].wait_event(self._events[step[1]])
            else:
                ev = timed.get(i) if timed else None
                if ev is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(streams[lane])
                rc = op(*step[1:], ptrs[lane])
                if ev is not None:
                    e1.record(streams[lane])
                    ev.append((e0, e1))
                if rc != 0:
                    _lib.check(rc, op.__name__)
        # (every side-lane launch is waited for by a main-lane step that consumes it: the caller's stream is again
        #  the only one with work in flight when this returns)

    def run_trunk(self, x):
        """part == "trunk": images of this sub-batch -> its slice of the shared feature buffers (current stream)."""
        lib = self.lib
        if self.stem is not None:
            self.stem.x = x.data_ptr()
            self.steps[self.in_step] = (lib.cf_stem_fused, C.byref(self.stem))
        else:
            self.steps[self.in_step] = (lib.cf_nchw_to_nhwc4, x.data_ptr(), self.x4.data_ptr(), self.B, 3, self.H, self.W)
        self._launch(_lib.stream_ptr())

    def run(self, model, x, pc_dep, calib, alloc=None):
        """alloc(c): where a (B, c, h4, w4) output goes (default: a fresh tensor).  part == "heads": `x` is unused
        (the feature buffers were filled by the trunk plans)."""
        B, H, W, dev = self.B, self.H, self.W, self.device
        lib = self.lib
        st = _lib.stream_ptr()
        h4, w4 = self.h4, self.w4
        heads = model.config.heads
        y = {}
        new = alloc or (lambda c: torch.empty((B, c, h4, w4), device=dev, dtype=torch.float32))
        def set_out(h, t, second=False):
            if h in self.tails:
                a, n = self.tails[h]
                (a.out2 if second else a.out)[n] = t.data_ptr()
            elif second:
                self.outs[h].out2 = t.data_ptr()
            else:
                self.outs[h].out = t.data_ptr()

        for h in self.primary:
            t = new(heads[h])
            y[h] = t
            set_out(h, t)
        depth_raw = y["depth"]                             # raw logits; "depth" gets the sigmoid form
        y["depthMap"] = depth_raw
        y["depth"] = new(1)
        set_out("depth", y["depth"], second=True)
        y["calib"] = calib
        if self.peaks_step is not None:
            # the decoder's peaks (3x3 NMS + top-K of the heat map, decode.py) are computed on the side lane beside the second
            # primary launch and travel with the heat map tensor: decode._peaks_and_maps picks them up when K and the
            # tensor's version still match, and computes them itself otherwise
            peaks_on = not torch.cuda.is_current_stream_capturing()   # (a captured forward hands out copies of its maps: nothing to carry)
            if peaks_on:
                pk_s = torch.empty((B, self.K), device=dev, dtype=torch.float32)
                pk_i = torch.empty((B, self.K), device=dev, dtype=torch.int32)
                pk_c = torch.empty((B, self.K), device=dev, dtype=torch.int32)
                pk_sum = torch.empty(2 * ops.CHECKSUM_PARTS, device=dev, dtype=torch.int64)   # checksum parts of the map the peaks belong to | decode's re-check
                n_words = B * heads["heatmap"] * h4 * w4
                self.steps[self.peaks_step] = (_peaks_and_checksum, lib, (y["heatmap"].data_ptr(), B, heads["heatmap"], h4, w4, self.K, 2,
                                               pk_s.data_ptr(), pk_i.data_ptr(), pk_c.data_ptr(), self.pk_ws.data_ptr()),
                                               (y["heatmap"].data_ptr(), C.c_long(n_words), pk_sum.data_ptr()))
            else:
                self.steps[self.peaks_step] = (_no_launch,)
        if self.in_step is not None:
            if self.stem is not None:
                self.stem.x = x.data_ptr()
                self.steps[self.in_step] = (lib.cf_stem_fused, C.byref(self.stem))
            else:
                self.steps[self.in_step] = (lib.cf_nchw_to_nhwc4, x.data_ptr(), self.x4.data_ptr(), B, 3, H, W)
        if self.radar:
            pc_hm = new(3)
            if self.topk_step is None:
                self.steps[self.frustum_step] = (
                    lib.cf_topk_frustum, y["heatmap"].data_ptr(), heads["heatmap"], self.K, y["depth"].data_ptr(),
                    y["widthHeight"].data_ptr(), y["dimension"].data_ptr(), y["rotation"].data_ptr(),
                    calib.data_ptr(), pc_dep.data_ptr(), B, h4, w4,
                    C.c_float(float(model.config.DATASET.MAX_PC_DIST)), pc_hm.data_ptr(),
                    _lib.ptr(self.pc_hm4), _lib.ptr(self.pc_hm8), self.tk_scores.data_ptr(), self.tk_inds.data_ptr(),
                    self.tk_cls.data_ptr(), self.tk_ws.data_ptr())
            else:
                self.steps[self.topk_step] = (lib.cf_topk_peaks, y["heatmap"].data_ptr(), B, heads["heatmap"],
                                              h4, w4, self.K, 0, self.tk_scores.data_ptr(),
                                              self.tk_inds.data_ptr(), self.tk_cls.data_ptr(),
                                              self.tk_ws.data_ptr())
                self.steps[self.frustum_step] = (
                    lib.cf_frustum_assoc, self.tk_inds.data_ptr(), self.K, y["depth"].data_ptr(),
                    y["widthHeight"].data_ptr(), y["dimension"].data_ptr(), y["rotation"].data_ptr(),
                    calib.data_ptr(), pc_dep.data_ptr(), B, h4, w4,
                    C.c_float(float(model.config.DATASET.MAX_PC_DIST)), pc_hm.data_ptr(),
                    _lib.ptr(self.pc_hm4), _lib.ptr(self.pc_hm8))
            y["pc_hm_in"] = pc_dep[:, :1]
            y["pc_hm"] = pc_hm[:, 0, :, :].unsqueeze(1)
            for h in SECONDARY_HEADS:
                t = new(heads[h])
                y[h] = t
                set_out(h, t)
            y["pc_hm_out"] = pc_hm[:, :1]
            y["depthMap"] = y["depth2"]                    # raw depth2 logits (detectHeads.py:188-190)
            y["depth2"] = new(1)
            set_out("depth2", y["depth2"], second=True)
        self._launch(st)
        if self.peaks_step is not None and peaks_on:
            hm = y["heatmap"]
            # (decode.py re-checks the map's contents against pk_sum[0] on the device before it trusts the peaks)
            hm._cf_peaks = (self.K, hm.data_ptr(), pk_s, pk_i, pk_c, pk_sum)
            side = getattr(self, "_side", None)
            if side is not None:                               # (allocator: these tensors were also used on the side stream)
                for t in (hm, pk_s, pk_i, pk_c, pk_sum):
                    t.record_stream(side)
        return [y]



def _no_launch(stream):
    """a plan step that issues nothing (status 0)"""
    return 0


def _peaks_and_checksum(lib, topk_args, sum_args, stream):
    """the decoder's NMS + top-k of the heat map and the checksum of the bits they were computed from (one plan step)"""
    rc = lib.cf_topk_peaks(*topk_args, stream)
    return rc if rc != 0 else lib.cf_checksum64(*sum_args, stream)


# ----------------------------------------------------------------------------------- the module
class DLASeg(nn.Module):
    def __init__(self, num_layers, in_channels, config):
        super().__init__()
        if str(num_layers) != "34":
            raise NotImplementedError("only DLA-34 is implemented (the reference ships nothing else)")
        if in_channels != 3:
            raise NotImplementedError("early fusion (radar channels in the image) is outside the hot path")
        if config.MODEL.DLA.NODE != "DeformConv":
            raise NotImplementedError("MODEL.DLA.NODE must be DeformConv (the only node type that works upstream)")
        if getattr(config.DATASET, "ONE_HOT_PC", False):
            raise NotImplementedError("ONE_HOT_PC is outside the hot path")
        self.config = config
        self.heads = config.heads
        self.isRadarEnabled = bool(config.DATASET.RADAR_PC)
        self.fusionStrategy = config.MODEL.FUSION_STRATEGY if self.isRadarEnabled else None
        if self.fusionStrategy not in (None, "middle"):
            raise NotImplementedError(f"fusion strategy {self.fusionStrategy!r} is outside the hot path")
        if self.isRadarEnabled and not config.MODEL.FRUSTUM:
            raise NotImplementedError("middle fusion without frustum association is outside the hot path")
        try:                                               # dla.py:578-580
            config.defrost()
            config.MODEL.PYRAMID_OUT_SIZE = [config.MODEL.OUTPUT_SIZE]
            config.freeze()
        except Exception:
            pass
        for name, tensor, is_buf in _param_spec(config):
            _register(self, name, tensor, is_buf)
        self._packed = None
        self._plans = {}         # (B, H, W, device, stream id [, role...]) -> _Plan; key[:5] names a plan SET (one forward shape)
        self._plan_sets = {}     # plan-set keys in least-recently-used order (dict order)
        self._graphs = {}        # (B, H, W, device, stream id, "graph", streams) -> captured forward, LRU as well
        self.max_plan_sets = 4   # plan sets / graphs kept per model: a set holds every intermediate of a forward (~6 GB at
                                 # bs=16, 448x800), so a service with varying batch sizes must not keep them all
        self._lock = threading.RLock()     # plans (buffers + argument blocks) are built / patched / launched under it
        self.precise = True      # two-level fp32 summation in backbone + neck (see cf_gemm.hip)
        self.conv_f16 = True     # backbone / offset convs: fp32 storage, split-fp16 products (cf_gemm_f16.hip)
        self.lanes = True        # small batches: the IDA projections on a side stream beside the node chain (_Plan.ida)
        self.lanes_max_frames = 10 # ... up to this many 448x800-frame equivalents per single-stream forward (round 6, ms per step with /
                                   # without: bs 5 3.35 / 3.39, 6 4.00 / 4.08, 7 4.23 / 4.31, 8 4.60 / 4.67, 10 5.26 / 5.35, 11 5.69 / 5.67)
        self.streams = 2         # > 1 (and batch >= min_sub_batch * streams): backbone + neck as that many sub-batches on
                                 # concurrent HIP streams with their own plans; heads on the caller's stream
        self.trunk_on_caller = True   # ... the last of those sub-batches on the caller's stream itself (one event wait less in front of the heads)
        self.min_sub_batch = 6   # ... and only when a sub-batch keeps at least this many 448x800-frame equivalents: measured (tools/
                                 # bench_small_batch.py, ms per forward + decode, one stream vs two): B=8 5.28 vs 5.91,
                                 # B=12 7.72 vs 6.86, B=16 9.16 vs 8.61 - four-frame trunks lose to one eight-frame forward
        self.use_graph = False   # replay the forward as ONE captured HIP graph (inputs / outputs staged through
                                 # static buffers) instead of ~100 launches from Python (_forward_graph)
        self.stem_fused = True   # with conv_f16: base_layer + level0 + level1 in one launch (cf_stem.hip)
        self.stem_pool = True    # ... which also writes the level-2 Tree's max-pool of its output (one launch less per trunk)
        self.conv_patch = True   # 3x3 stride-1 f16x3 convs: LDS patch reuse (cf_conv3x3_f16.hip)
        self.root_fuse = True    # one-level Trees: tree2.conv2 + Root as one step (cf_conv3x3_root_f16x3) ...
        self.root_fuse_children = False   # ... also where the Root has further sources (level3.tree2, level4.tree2): same bits, and
                                          # in the two-stream step the two launches are 0.024 ms faster (round 5, 6 of 6 A/B pairs)
        self.heads_lanes = True  # fused heads: the decoder's NMS + top-k on a side stream beside the frustum path and the secondary
                                 # launch (_Plan heads section), handed to decode.py with the heat map
        self.peaks_behind_frustum = True  # heads_lanes on a radar model: the decoder's lane starts behind the frustum chain instead of
                                          # behind the primary head launch (False: round 5's order; same results)
        self.frustum_fused = True  # radar: top-k of the raw heat map + frustum association as cf_topk_frustum (2 launches, the merge in the
                                   # association kernel's prologue) instead of cf_topk_peaks + cf_frustum_assoc (3); same bits
        self.proj_fuse = True    # the sub-tree that opens a level: `project` of the pooled input as k-steps of tree1.conv2
                                 # (cf_conv3x3_proj_f16x3) instead of a launch + a residual tensor; set before the first forward
        self.heads_bf16 = True   # head GEMMs on the bf16 / f16 MFMA pipe with split operands: ONE cf_head_fused launch per head group
                                 # on v_mfma_f32_16x16x32_* (hidden maps stay in LDS).  False - or a head with more than 16 outputs,
                                 # which the 16-row output tile of that kernel does not hold - : exact-fp32 layer-by-layer heads
        self.heads_mx = True     # ... and the FIRST layer of every fused head as fp16 main term + block-scaled FP6 cross terms
                                 # (v_mfma_scale_f32_16x16x128_f8f6f4): 1.5 MFMA passes per product instead of 3; hidden and
                                 # output layers stay bf16x3 (the float64-anchored gate rejects FP6 cross terms there)
        self._mx_active = False  # set by _prepare: heads_mx and everything it needs (bf16 fused heads on 16x16x32 fragments)
        self.pack_mx_fused = True  # with heads_mx: the feature map's DCN writes the heads' operand rows from its epilogue
                                   # (cf_dcn_args.out_mx); False: a separate cf_pack_feat_mx pass (A/B, byte-identical)
        self.record_spans = False  # dev / tests: keep HIP events around each trunk of _forward_concurrent (trunk_overlap)
        self.trunk_spans = []
        # Dynamic range of the split-fp16 operands: every activation is multiplied by a per-layer power of
        # two before it is split into fp16 hi + lo - 16 unless `calibrate` has measured that a layer's inputs need less.
        self._ranges = None        # {layer name: max |input| measured by calibrate()}; None = the default pre-scale everywhere
        self._feat_scale = ops.DEFAULT_IN_SCALE   # pre-scale of the heads' mx feature rows (set by _prepare from _ranges)
        self.range_headroom = 8.0  # a calibrated layer keeps max |input| * scale <= 65504 / this
        self._range_checked = False  # a check_ranges / calibrate has run on these weights (Detector's first-batch guard)
        self.register_load_state_dict_post_hook(lambda m, _k: m._weights_changed())
        self.eval()

    def _weights_changed(self):
        """load_state_dict: new weights - re-pack lazily, and whatever was known about the activation ranges is void."""
        self._ranges = None
        self._range_checked = False
        self.invalidate()

    # weights changed or moved (load_state_dict / .to()) -> re-pack lazily
    def invalidate(self):
        self._packed = None
        self._plans = {}
        self._plan_sets = {}
        with _CAPTURE_LOCK:                  # (graphs are never destroyed while another thread's stream is capturing)
            self._graphs = {}

    def _plan(self, key, build, store=None):
        """The plan under `key`, built on first use.  Plans of the eager path live in `self._plans` under an LRU of
        `max_plan_sets` plan sets (key[:5] = batch, height, width, device, stream): the least recently used set is
        dropped whole, its buffers go back to torch's allocator (every launch that used them was issued on - or joined
        into - the stream they were allocated on, so the allocator's stream-ordered reuse is safe).  Plans captured
        into a HIP graph live in the graph's own `store` instead: the graph addresses their buffers by raw pointer, so
        they must live exactly as long as the graph."""
        if store is not None:
            plan = store.get(key)
            if plan is None:
                plan = store[key] = build()
            return plan
        sk = key[:5]
        self._plan_sets.pop(sk, None)
        self._plan_sets[sk] = True                           # most recently used last
        while len(self._plan_sets) > max(1, int(self.max_plan_sets)):
            old = next(iter(self._plan_sets))
            del self._plan_sets[old]
            for k in [k for k in self._plans if k[:5] == old]:
                del self._plans[k]
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = build()
        return plan

    def _all_plans(self):
        """Every live plan: the eager ones and those owned by captured graphs."""
        out = list(self._plans.values())
        for g in self._graphs.values():
            out += list(g[-1].values())
        return out

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------------------ weight prep
    def _prepare(self, device):
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        pk = {}
        self._mx_active = False

        def bn(p):
            return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])

        f16 = bool(self.conv_f16)

        def pack_any(w, b, sources, stride=1):
            """split-fp16 products (fp32 storage) wherever the sources allow 8-channel slots."""
            # (<= 32 output channels stay on the fp32 kernels: a 32-row MFMA tile per wave leaves the
            #  f16 path bound by the on-the-fly operand split)
            ok8 = all(s.stride % 8 == 0 and s.channels % 8 == 0 for s in sources)
            patchable = (w.shape[2] == 3 and stride == 1 and len(sources) == 1 and sources[0].channels % 16 == 0)
            if f16 and ok8 and (w.shape[0] > 32 or patchable):
                return packing.pack_conv_f16(w, b, sources, stride=stride).to(device)
            return packing.pack_conv(w, b, sources, stride=stride).to(device)

        def conv_bn(name, wkey, bnkey, sources, stride=1, bias=None):
            w, b = packing.fold_bn(sd[wkey], bias, bn(bnkey) if bnkey else None)
            pk[name] = pack_any(w, b, sources, stride=stride)

        if f16 and self.stem_fused:
            folded = [packing.fold_bn(sd[f"base.{n}.0.weight"], None, bn(f"base.{n}.1"))
                      for n in ("base_layer", "level0", "level1")]
            pk["base.stem"] = packing.pack_stem(*[t for wb in folded for t in wb]).to(device)
        else:
            conv_bn("base.base_layer", "base.base_layer.0.weight", "base.base_layer.1", [Source(3, 4)])
            conv_bn("base.level0", "base.level0.0.weight", "base.level0.1", [Source(16, 16)])
            conv_bn("base.level1", "base.level1.0.weight", "base.level1.1", [Source(16, 16)], stride=2)

        def tree1(p, ci, co, root_srcs):
            conv_bn(p + ".tree1.conv1", p + ".tree1.conv1.weight", p + ".tree1.bn1", [Source(ci, ci)],
                    stride=2 if ci != co else 1)
            if f16 and self.proj_fuse and (p + ".project.0.weight") in sd and co >= 64 and ci % 32 == 0:
                # the Tree's project (dla.py:98-103) rides in tree1.conv2's accumulators: no ".project" entry, no launch
                w2, b2 = packing.fold_bn(sd[p + ".tree1.conv2.weight"], None, bn(p + ".tree1.bn2"))
                wp, bp = packing.fold_bn(sd[p + ".project.0.weight"], None, bn(p + ".project.1"))
                pk[p + ".tree1.conv2"] = packing.pack_conv_f16(w2, b2, [Source(co, co)],
                                                               proj=(wp, bp, Source(ci, ci))).to(device)
            else:
                conv_bn(p + ".tree1.conv2", p + ".tree1.conv2.weight", p + ".tree1.bn2", [Source(co, co)])
            conv_bn(p + ".tree2.conv1", p + ".tree2.conv1.weight", p + ".tree2.bn1", [Source(co, co)])
            conv_bn(p + ".tree2.conv2", p + ".tree2.conv2.weight", p + ".tree2.bn2", [Source(co, co)])
            conv_bn(p + ".root", p + ".root.conv.weight", p + ".root.bn", [Source(c, c) for c in root_srcs])
            if (p + ".project.0.weight") in sd and not pk[p + ".tree1.conv2"].proj_k:
                conv_bn(p + ".project", p + ".project.0.weight", p + ".project.1", [Source(ci, ci)])

        tree1("base.level2", 32, 64, [64, 64])
        for lvl, ci, co in ((3, 64, 128), (4, 128, 256)):
            tree1(f"base.level{lvl}.tree1", ci, co, [co, co])
            tree1(f"base.level{lvl}.tree2", co, co, [co, co, ci, co])
        tree1("base.level5", 256, 512, [512, 512, 256])

        def dcn(p, ci, co):
            w, b = packing.fold_bn(sd[p + ".weight"], sd[p + ".bias"], bn(p + ".activation.0"))
            pk[p] = (packing.pack_dcn_f16 if f16 else packing.pack_dcn)(w, b).to(device)
            pk[p + ".conv_offset_mask"] = pack_any(
                sd[p + ".conv_offset_mask.weight"].float().cpu(),
                sd[p + ".conv_offset_mask.bias"].float().cpu(), [Source(ci, ci)])

        def ida(p, o, srcs, fs):
            for n in range(1, len(srcs)):
                dcn(f"{p}.proj_{n}", srcs[n], o)
                dcn(f"{p}.node_{n}", o, o)
                pk[f"{p}.up_{n}"] = (packing.pack_upsample(sd[f"{p}.up_{n}.weight"]).to(device), fs[n])

        chans = CHANNELS[2:]
        in_ch = list(chans)
        for i in range(3):
            j = -i - 2
            ida(f"dla_up.ida_{i}", chans[j], in_ch[j:], [1] + [2] * (len(in_ch[j:]) - 1))
            in_ch[j + 1:] = [chans[j]] * len(in_ch[j + 1:])
        ida("ida_up", 64, [64, 128, 256], [1, 2, 4])

        # heads: sibling first layers share their input -> one GEMM with concatenated outputs
        heads = dict(self.config.heads)
        head_conv = {k: list(v) for k, v in self.config.head_conv.items()}
        radar = self.isRadarEnabled and self.fusionStrategy == "middle"
        hp = "detectHead_0"
        primary = [h for h in heads if not (radar and h in SECONDARY_HEADS)]
        for h in heads:
            if any(c != 256 for c in head_conv[h]):
                raise NotImplementedError("head_conv widths other than 256 are not on the path")
        bf = self._heads_bf()
        pack = packing.pack_conv_bf16 if bf else packing.pack_conv
        feat_src = Source(64, 64)
        pc_src = Source(3, 8 if bf else 4)
        hw = lambda h, i: sd[f"{hp}.{h}.{i}.weight"].float().cpu()
        hb = lambda h, i: sd[f"{hp}.{h}.{i}.bias"].float().cpu()
        pk["heads.primary.0"] = pack(torch.cat([hw(h, 0) for h in primary], 0),
                                     torch.cat([hb(h, 0) for h in primary], 0), [feat_src]).to(device)
        for n, h in enumerate(primary):
            assert len(head_conv[h]) == 1
            pk[f"heads.{h}.out"] = pack(hw(h, 2), hb(h, 2), [Source(256, 256 * len(primary), 256 * n)]).to(device)
        if radar:
            pk["heads.secondary.0"] = pack(torch.cat([hw(h, 0) for h in SECONDARY_HEADS], 0),
                                           torch.cat([hb(h, 0) for h in SECONDARY_HEADS], 0),
                                           [feat_src, pc_src]).to(device)
            ns = 256 * len(SECONDARY_HEADS)
            for n, h in enumerate(SECONDARY_HEADS):
                assert len(head_conv[h]) == 3
                for idx in (2, 4):
                    pk[f"heads.{h}.{idx}"] = pack(hw(h, idx), hb(h, idx), [Source(256, ns, 256 * n)]).to(device)
                pk[f"heads.{h}.out"] = pack(hw(h, 6), hb(h, 6), [Source(256, ns, 256 * n)]).to(device)
        if bf:
            m16 = True               # 16x16x32 fragments: what head_patch16_kernel reads (_heads_bf: every head has <= 16 outputs)
            def tail(h, hidden_idx, out_idx):
                n_out = heads[h]
                b32 = torch.zeros(32)
                b32[:n_out] = hb(h, out_idx)
                w2 = hw(h, out_idx).view(n_out, 256)
                perm = (packing.pack_fragments16(w2, acc_order=True) if m16 else packing.pack_fragments(w2, acc_order=True))
                pf = packing.pack_fragments16 if m16 else packing.pack_fragments
                return dict(w_hidden=[pf(hw(h, i).view(256, 256)).to(device) for i in hidden_idx],
                            b_hidden=[hb(h, i).to(device) for i in hidden_idx],
                            w_out=pf(w2).to(device), w_out_perm=perm.to(device),
                            b_out=b32.to(device), n_out=n_out, mfma16=m16)
            mx = m16 and bool(self.heads_mx)
            self._mx_active = mx
            # the mx rows' pre-scale: one for both head groups (they read the same rows)
            fr = [self._ranges[n] for n in ("heads.primary.0", "heads.secondary.0") if self._ranges and n in self._ranges]
            self._feat_scale = ops.in_scale_for(max(fr), self.range_headroom) if (mx and fr) else ops.DEFAULT_IN_SCALE
            def first(h, srcs):
                if mx:
                    d = packing.pack_head_first_mx(hw(h, 0), hb(h, 0), pc=len(srcs) == 2, feat_scale=self._feat_scale)
                    return dict(w_first=d["w_first"].to(device), b_first=d["b_first"].to(device), first_scale=d["first_scale"],
                                real_cin=d["real_cin"])
                pc = packing.pack_conv_bf16(hw(h, 0), hb(h, 0), srcs, fragments=16 if m16 else True).to(device)
                return dict(w_first=pc.weight, b_first=pc.bias[:256].contiguous(), slots=pc.slots, k_pad=pc.k_pad,
                            real_cin=pc.real_cin)
            pk["tails.primary"] = {h: dict(tail(h, [], 2), **first(h, [feat_src])) for h in primary}
            if radar:
                pk["tails.secondary"] = {h: dict(tail(h, [2, 4], 6), **first(h, [feat_src, pc_src]))
                                         for h in SECONDARY_HEADS}
        self._packed = pk

    def _heads_bf(self):
        """The heads run as fused split-operand launches (cf_head_fused on 16x16x32 fragments): model.heads_bf16 and every head's
        output count fits that kernel's one 16-row output tile; otherwise the exact-fp32 layer-by-layer heads."""
        return bool(self.heads_bf16) and all(int(n) <= 16 for n in self.config.heads.values())

    # ----------------------------------------------------------------------------- dynamic range
    def _range_groups(self):
        """{layer name: the layer names that share ONE activation pre-scale with it} for every layer of the packed model whose
        operands are split to fp16 (f16x3 convolutions / DCNs, the fused stem, the heads' mx first layers).  Names are the
        unfused plan's: `X.tree1.conv2` with its Tree's `X.project` when the projection rides in conv2's accumulators."""
        pk = self._packed
        groups = {}
        for name, pc in pk.items():
            if isinstance(pc, (dict, tuple)) or getattr(pc, "out_scale", 0.0) <= 0 or name == "base.stem":
                continue
            g = self._group_of(name)
            for n in g:
                groups[n] = g
        if "base.stem" in pk:
            for n in ("base.base_layer", "base.level0", "base.level1"):
                groups[n] = (n,)
        if self._mx_active:
            g = ("heads.primary.0",) + (("heads.secondary.0",) if "tails.secondary" in pk else ())
            for n in g:
                groups[n] = g
        return groups

    def _group_of(self, name):
        """(name,) or, where a Tree's `project` rides in its tree1.conv2's accumulators, that pair: one pre-scale for both."""
        pk = self._packed or {}
        if name.endswith(".project"):
            c2 = name[:-len(".project")] + ".tree1.conv2"
            if getattr(pk.get(c2), "proj_k", 0) > 0:
                return (c2, name)
        if getattr(pk.get(name), "proj_k", 0) > 0:
            return (name, name[:-len(".tree1.conv2")] + ".project")
        return (name,)

    def _scale(self, name):
        """The activation pre-scale of layer `name` under the current calibration (None = the kernels' default, 16)."""
        if self._ranges is None:
            return None
        vals = [self._ranges[n] for n in self._group_of(name) if n in self._ranges]
        return ops.in_scale_for(max(vals), self.range_headroom) if vals else None

    def activation_ranges(self):
        """{layer name: max |x| over the layer's input tensors} read from the RESIDENT buffers of the live plans, i.e. for
        the most recent forward of each plan: one reduction launch per buffer on the current stream, one device -> host
        copy - off the hot path, no re-run.  Covers every operand that exists in HBM; the operands fused launches keep on
        the chip (the stem's two full-resolution maps, x2 of a conv2 + Root launch) are not seen - `hidden_layers()` names
        them, `measure_ranges` sees them too.  NaN / inf inputs come out as nan / inf."""
        items = [(name, t) for plan in self._all_plans() for name, ts in plan.inputs.items() for t in ts]
        if not items:
            return {}
        lib, st = _lib.load(), _lib.stream_ptr()
        dev = items[0][1].device
        with torch.cuda.device(dev):
            out = torch.zeros(len(items), device=dev, dtype=torch.float32)
            for i, (_, t) in enumerate(items):
                c = t.shape[-1]
                _lib.check(lib.cf_absmax_f32(t.data_ptr(), t.numel() // c, c, c, out.data_ptr() + 4 * i, st), "cf_absmax_f32")
            vals = out.cpu().tolist()
        r = {}
        for (name, _), v in zip(items, vals):
            old = r.get(name, 0.0)
            r[name] = old if old != old else (v if v != v else max(old, v))      # (a NaN sticks)
        return r

    def hidden_layers(self):
        """Layer names with operands `activation_ranges` cannot see in the live plans (kept in LDS / registers by a fused launch)."""
        return sorted(set().union(*[p.hidden for p in self._all_plans()])) if self._all_plans() else []

    def measure_ranges(self, images, pc_dep=None, calib=None):
        """{layer name: max |x| over that layer's inputs} for THIS batch, every layer, measured on a shadow of the model
        that runs the exact-fp32 kernels with nothing fused (every intermediate in HBM, no fp16 split anywhere, so no value
        measured here can itself be a clamped one).  One slow forward (~30 ms at bs=16) + one reduction per buffer; the
        shadow's plans are freed on return.  Reference semantics this protects: the fp32 convolutions of
        model/networks/dla.py:124-159 accept any magnitude."""
        if not images.is_cuda:
            raise _lib.CfHipError("measure_ranges needs device tensors: the HIP path has no CPU fallback")
        with self._lock, torch.cuda.device(images.device), torch.no_grad():
            with torch.random.fork_rng(devices=[]):            # (the shadow's throw-away initialisation draws random numbers)
                shadow = DLASeg(34, 3, self.config)
            shadow.load_state_dict(self.state_dict())
            shadow.to(images.device)
            shadow.conv_f16, shadow.heads_bf16, shadow.streams, shadow.lanes, shadow.use_graph = False, False, 1, False, False
            shadow(images, pc_dep=pc_dep, calib=calib)
            r = shadow.activation_ranges()
            del shadow
        return r

    def range_violations(self, ranges, fraction=0.5):
        """[(layer, max |input|, pre-scale)] for every fp16-split layer whose inputs, under the CURRENT pre-scales, reach
        `fraction` of the fp16 limit (65504) or are not finite.  ranges: from measure_ranges / activation_ranges."""
        if self._packed is None:
            raise _lib.CfHipError("range_violations: run a forward (or check_ranges) first - the weights are not packed yet")
        out = []
        for name, g in sorted(self._range_groups().items()):
            vals = [ranges[n] for n in g if n in ranges]
            if not vals:
                continue
            m = float("nan") if any(v != v for v in vals) else max(vals)
            s = (self._feat_scale if name.startswith("heads.") else self._scale(name)) or ops.DEFAULT_IN_SCALE
            if not (m * s < fraction * ops.F16_MAX):
                out.append((name, m, s))
        return out

    def _raise_on(self, viol, how):
        if viol:
            worst = ", ".join(f"{n}: max |input| {m:.4g} x pre-scale {s:g}" for n, m, s in viol[:6])
            raise _lib.CfHipError(
                f"activation range guard ({how}): {len(viol)} layer(s) would be clamped at +-65504 / pre-scale by the split-fp16 "
                f"kernels ({worst}{', ...' if len(viol) > 6 else ''}).  Call model.calibrate(images, pc_dep=..., calib=...) "
                "on representative batches (per-layer power-of-two pre-scales), or Detector(..., range_policy='calibrate').")

    def check_ranges(self, images, pc_dep=None, calib=None):
        """Range guard: measure this batch's activation ranges (`measure_ranges`) and raise `CfHipError` naming every layer
        whose inputs reach HALF the fp16 limit under the current pre-scales - 65504 / 16 / 2 = 2047 on an uncalibrated
        model - instead of ever returning a silently clamped map.  -> the ranges."""
        r = self.measure_ranges(images, pc_dep, calib)
        with self._lock, torch.cuda.device(images.device):
            if self._packed is None:
                self._prepare(images.device)
            self._raise_on(self.range_violations(r), "check_ranges")
        self._range_checked = True
        return r

    def check_resident_ranges(self):
        """The cheap form of the guard for a running service: the same test on `activation_ranges()` - the buffers the last
        forwards left in HBM - without a second forward.  Blind to `hidden_layers()`."""
        r = self.activation_ranges()
        self._raise_on(self.range_violations(r), "check_resident_ranges")
        return r

    def calibrate(self, images, pc_dep=None, calib=None, reset=True):
        """Choose the per-layer activation pre-scales from this batch (accumulating over calls with reset=False): layers whose
        inputs stay below 1023 keep the default 16 (bit-identical results), the others get the largest power of two with
        max |input| * scale <= 65504 / range_headroom.  Plans and packed weights are rebuilt lazily.  -> the ranges."""
        r = self.measure_ranges(images, pc_dep, calib)
        bad = [n for n, v in r.items() if v != v or v == float("inf")]
        if bad:
            raise _lib.CfHipError(f"calibrate: non-finite activations in the inputs of {bad[:6]}")
        if not reset and self._ranges:
            r = {k: max(r.get(k, 0.0), self._ranges.get(k, 0.0)) for k in set(r) | set(self._ranges)}
        with self._lock:
            self._ranges = r
            self.invalidate()
        self._range_checked = True
        return r

    def calibration(self):
        """The ranges `calibrate` holds (None if uncalibrated) - save them beside a checkpoint, restore with `set_calibration`."""
        return None if self._ranges is None else dict(self._ranges)

    def set_calibration(self, ranges):
        with self._lock:
            self._ranges = None if ranges is None else {str(k): float(v) for k, v in ranges.items()}
            self.invalidate()
        self._range_checked = ranges is not None

    def activation_scales(self):
        """{layer: pre-scale} of every fp16-split layer under the current calibration (after the weights were packed)."""
        if self._packed is None:
            return {}
        return {n: ((self._feat_scale if n.startswith("heads.") else self._scale(n)) or ops.DEFAULT_IN_SCALE)
                for n in self._range_groups()}

    # ----------------------------------------------------------------------------------- forward
    def forward(self, x, pc_hm=None, pc_dep=None, calib=None):
        if self.training:
            raise NotImplementedError("training is outside the hot path; call model.eval()")
        if not x.is_cuda:
            raise _lib.CfHipError("DLASeg.forward needs device tensors: the HIP path has no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"images must be (B,3,H,W) with H,W multiples of 32, got {tuple(x.shape)}")
        B, _, H, W = x.shape
        dev = x.device
        x = x.float().contiguous()
        if self.isRadarEnabled:
            if pc_dep is None or calib is None:
                raise ValueError("radar model: pc_dep and calib are required")
            if tuple(pc_dep.shape) != (B, 3, H // 4, W // 4):
                raise ValueError(f"pc_dep must be {(B, 3, H // 4, W // 4)}, got {tuple(pc_dep.shape)}")
            if pc_dep.dtype != torch.float32 or not pc_dep.is_contiguous():
                raise ValueError("pc_dep must be contiguous float32")
            calib = calib.reshape(B, 3, 4).float().contiguous()
        # One model may be driven from several HIP streams and host threads.  A plan owns its intermediate buffers, so
        # plans are keyed by the stream the call is issued on (two forwards in flight on two streams never share a
        # buffer); building, patching the per-call pointers into the argument blocks and issuing the launches happen
        # under the model's lock (kernel arguments are copied at launch, so the blocks are free again on return).
        with self._lock, torch.cuda.device(dev):
            if self._packed is None:
                self._prepare(dev)
            sid = torch.cuda.current_stream(dev).cuda_stream
            if self.use_graph:
                return self._forward_graph(x, pc_dep, calib, B, H, W, dev, sid)
            return self._forward_eager(x, pc_dep, calib, B, H, W, dev, sid)

    def _forward_eager(self, x, pc_dep, calib, B, H, W, dev, sid, store=None):
        # (min_sub_batch counts 448 x 800 frames: a sub-batch of a larger input carries proportionally more work)
        if self.streams > 1 and B % self.streams == 0 and (B // self.streams) * H * W >= self.min_sub_batch * 448 * 800:
            return self._forward_concurrent(x, pc_dep, calib, B, H, W, dev, sid, store)
        plan = self._plan((B, H, W, dev, sid), lambda: _Plan(self, B, H, W, dev), store)
        return plan.run(self, x, pc_dep, calib)

    def _forward_graph(self, x, pc_dep, calib, B, H, W, dev, sid):
        """model.use_graph: the whole forward - including the fork into the trunk streams and the join in front of
        the heads - captured ONCE as a HIP graph over static input / output buffers and replayed per call (one
        hipGraphLaunch instead of ~100-350 launches from Python: with 4 trunk streams the eager path is bound by the
        host's launch rate).  Semantics are those of the eager forward: fresh output tensors every call (copies out
        of the static ones), `pc_hm_in` a view of the CALLER's pc_dep, `calib` the caller's tensor."""
        key = (B, H, W, dev, sid, "graph", self.streams)
        g = self._graphs.pop(key, None)
        if g is None:
            # warm-up (one-time attribute calls, the stream probe) with a throw-away plan set: its buffers are freed
            # again before the capture allocates the set the graph keeps
            self._forward_eager(x, pc_dep, calib, B, H, W, dev, sid, store={})
            torch.cuda.synchronize(dev)
            gx = x.clone()
            gpc = pc_dep.clone() if pc_dep is not None else None
            gcal = calib.clone() if calib is not None else None
            graph = torch.cuda.CUDAGraph()
            plans = {}                                                 # owned by the graph (see _plan)
            # No garbage collection may run inside the capture: a collected object with device-side teardown (an older
            # captured graph of a model that is itself garbage, say) aborts the process when its destructor runs while a
            # stream is capturing - and torch.cuda.graph() no longer collects on entry by default.  Collect now, then hold
            # the collector off until the capture has ended.
            import gc
            with _CAPTURE_LOCK:
                gc.collect()
                gc_was_on = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(graph):
                        cap_sid = torch.cuda.current_stream(dev).cuda_stream   # plans of the capture stream (own buffers)
                        gout = self._forward_eager(gx, gpc, gcal, B, H, W, dev, cap_sid, store=plans)[0]
                finally:
                    if gc_was_on:
                        gc.enable()
            g = (graph, gx, gpc, gcal, gout, plans)
        self._graphs[key] = g                                          # most recently used last
        while len(self._graphs) > max(1, int(self.max_plan_sets)):
            with _CAPTURE_LOCK:
                self._graphs.pop(next(iter(self._graphs)))
        graph, gx, gpc, gcal, gout, _plans = g
        gx.copy_(x)
        if gpc is not None:
            gpc.copy_(pc_dep)
        if gcal is not None:
            gcal.copy_(calib)
        graph.replay()
        y, fresh = {}, {}
        for k, v in gout.items():
            if k == "calib":
                y[k] = calib
            elif k == "pc_hm_in":
                y[k] = pc_dep[:, :1]
            else:                                                   # aliases in gout stay aliases (depthMap / pc_hm views)
                base = v._base if v._base is not None else v
                if id(base) not in fresh:
                    fresh[id(base)] = base.clone()
                nb = fresh[id(base)]
                y[k] = nb if v._base is None else nb.as_strided(v.size(), v.stride(), v.storage_offset() - base.storage_offset())
        return [y]

    def _forward_concurrent(self, x, pc_dep, calib, B, H, W, dev, sid, store=None):
        """Backbone + neck as `self.streams` sub-batches, each with its OWN trunk plan (own intermediate buffers)
        on its OWN HIP stream: several of those layers cannot fill the chip on their own (level4: 175 workgroups
        for 256 CUs, the 14x25 / 28x50 maps of the neck, the tail round of most grids) and the other sub-batch's
        launches fill the holes.  The trunks write their slices of ONE full-batch feature map; the heads, top-k,
        frustum association and secondary heads then run for the whole batch on the caller's stream (their grids
        fill the chip, and a launch there overlaps nothing - HIP-event and rocprofv3 durations of the head kernels
        mean the same thing as in the single-stream forward).  Frames are independent, so the result is the
        single-stream one bit for bit."""
        n = self.streams
        k = B // n
        h4, w4 = H // 4, W // 4
        cur = torch.cuda.current_stream(dev)
        pool = _side_streams(dev, sid, n)
        bf = self._heads_bf()

        def heads_plan():
            feat = torch.empty((B, h4, w4, 64), device=dev, dtype=torch.float32)
            if bf and self._mx_active:
                feat_in = torch.empty((B, h4, w4, packing.MX_ROW), device=dev, dtype=torch.uint8)
            else:
                feat_in = torch.empty((B, h4, w4, 2, 64), device=dev, dtype=torch.bfloat16) if bf else feat
            return _Plan(self, B, H, W, dev, part="heads", feat=feat, feat_in=feat_in)

        hplan = self._plan((B, H, W, dev, sid, "heads", n), heads_plan, store)
        spans = []
        # model.trunk_on_caller: the LAST trunk is issued on the caller's stream itself (behind the forks of the others), so the
        # heads follow its last launch in stream order and wait for n - 1 events instead of n
        on_cur = bool(self.trunk_on_caller)
        for i in range(n):
            sl = slice(i * k, (i + 1) * k)
            tplan = self._plan((B, H, W, dev, sid, "trunk", i, n),
                               lambda: _Plan(self, k, H, W, dev, part="trunk", feat=hplan.feat[sl],
                                             feat_in=hplan.feat_in[sl] if bf else None), store)
            s = cur if (on_cur and i == n - 1) else pool[i]
            if s is not cur:
                s.wait_stream(cur)
            with torch.cuda.stream(s):
                if self.record_spans:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                tplan.run_trunk(x[i * k:(i + 1) * k])
                if self.record_spans:
                    e1.record(s)
                    spans.append((e0, e1))
        for s in (pool[:n - 1] if on_cur else pool):
            cur.wait_stream(s)
        if self.record_spans:
            self.trunk_spans = spans
        return hplan.run(self, None, pc_dep, calib)

    def trunk_overlap(self):
        """With `record_spans` set: the last concurrent forward's trunk spans -> (ms each trunk took, ms during which
        the first two ran at the same time).  Streams on one hardware queue give ~0 overlap.  Syncs."""
        torch.cuda.synchronize()
        (a0, a1), (b0, b1) = self.trunk_spans[:2]
        la, lb = a0.elapsed_time(a1), b0.elapsed_time(b1)
        start_b = a0.elapsed_time(b0)                       # b's start relative to a's
        return [la, lb], max(0.0, min(la, start_b + lb) - max(0.0, start_b))


    # ------------------------------------------------------------------------- instrumentation
    def time_launch(self, name, on=True):
        """Bracket the named launch with HIP events (recorded on the launch stream) in every existing plan that
        holds it; read back with launch_times().  Names: 'tails.primary', 'base.level2.root', ..."""
        for plan in self._all_plans():
            idx = plan.step_index.get(name)
            if idx is None:
                continue
            if on:
                plan.timed.setdefault(idx, [])
            else:
                plan.timed.pop(idx, None)

    def launch_times(self, name):
        """-> (list of ms per recorded launch, algorithmic FLOPs of one launch); syncs."""
        torch.cuda.synchronize()
        out, flops = [], 0.0
        for plan in self._all_plans():
            idx = plan.step_index.get(name)
            if idx is None:
                continue
            flops = plan.step_flops[name]
            for e0, e1 in plan.timed.get(idx, []):
                out.append(e0.elapsed_time(e1))
            if idx in plan.timed:
                plan.timed[idx] = []
        return out, flops

    def time_all(self, on=True):
        """Bracket EVERY launch of every plan with HIP events (dev tool: tools/layer_times.py)."""
        for plan in self._all_plans():
            plan.timed = {i: [] for i in range(len(plan.steps))} if on else {}

    def all_launch_times(self):
        """-> [(step name or kernel entry point, mean ms, algorithmic FLOPs)] in launch order (single-plan forward)."""
        torch.cuda.synchronize()
        plan = list(self._plans.values())[-1]
        names = {v: k for k, v in plan.step_index.items()}
        out = []
        for i, step in enumerate(plan.steps):
            ev = plan.timed.get(i, [])
            if not ev:
                continue
            ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            nm = names.get(i, step[0].__name__)
            out.append((nm, ms, plan.step_flops.get(names.get(i, ""), 0.0)))
        return out

    def conv_flops_per_forward(self):
        """Algorithmic FLOPs (2*MACs) of all conv / DCN launches of one forward (the plans in use)."""
        return sum(sum(p.step_flops.values()) for p in self._all_plans())


_network_factory = {"dla": DLASeg}


def getModel(config):
    """model/model.py:18-44."""
    arch = config.MODEL.ARCH
    num_layers = arch[arch.find("_") + 1:] if "_" in arch else 0
    arch = arch[:arch.find("_")] if "_" in arch else arch
    return _network_factory[arch](num_layers, in_channels=3, config=config)
