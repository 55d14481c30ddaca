"""DEV-ONLY: generate tests/golden/*.npz by running the REFERENCE's own Python
(/root/reference, read-only) on the deterministic inputs of tests/golden/cases.py.

Runs only in the build container (the reference does not exist on the GPU box).  Nothing from
the reference is copied: the committed artefacts are input seeds and output tensors.

Import recipe = SURVEY.md Appendix D: packages the reference imports but that are absent from
this image (nuscenes-devkit, pyquaternion, lightning, yacs, torchvision) are registered as
inert module objects before the import; `torchvision.ops.deform_conv2d` is served by
oracle/dcn_ref.py (so the DCN arithmetic itself stays "parity unpinned" - it is held by the
known-answer tests - while everything around it is the reference's own code).  For utils/postProcess.py
the absent `cv2` is a module object whose only attribute is a numpy 3-point solve standing for
cv2.getAffineTransform (so that solve is likewise outside the pin).

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"


def _install_inert_modules():
    from oracle import dcn_ref

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def defrost(self):
            pass

        def freeze(self):
            pass

    mod("nuscenes"); mod("nuscenes.utils")
    mod("nuscenes.utils.data_classes", RadarPointCloud=type("RadarPointCloud", (), {}))
    mod("nuscenes.utils.geometry_utils", view_points=None, transform_matrix=None)
    mod("pyquaternion", Quaternion=object)
    mod("lightning"); mod("lightning.pytorch")
    mod("lightning.pytorch.utilities", rank_zero_only=lambda f: f)
    mod("yacs"); mod("yacs.config", CfgNode=CfgNode)
    mod("torchvision"); mod("torchvision.ops", deform_conv2d=dcn_ref.deform_conv2d)

    def get_affine_transform(src, dst):      # cv2.getAffineTransform: exact 3-point solve, float64 (2,3)
        A = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], axis=1)
        return np.linalg.solve(A, np.asarray(dst, np.float64)).T.copy()
    mod("cv2", getAffineTransform=get_affine_transform)


def reference_config(radar, H, W):
    from config.default import _Cfg as cfg
    from config.utils import updateConfigHeads, updateConfigHeadsWeights, updateConvNumOfHeads
    cfg.DATASET.RADAR_PC = radar
    cfg.MODEL.FUSION_STRATEGY = "middle" if radar else None
    cfg.MODEL.FRUSTUM = radar
    cfg.MODEL.LOAD_DIR = "offline-placeholder"        # never download ImageNet weights
    cfg.DATASET.NUM_CLASSES = 10
    cfg.MODEL.INPUT_SIZE = (H, W)
    cfg.MODEL.OUTPUT_SIZE = (H // 4, W // 4)
    cfg.DATASET.PC_REVERSE = True
    updateConfigHeads(cfg)
    updateConfigHeadsWeights(cfg)
    updateConvNumOfHeads(cfg)
    return cfg


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def sample_idx(n, m=4096, seed=123):
    return np.random.RandomState(seed).randint(0, n, size=m)


def main():
    import matplotlib
    matplotlib.use("Agg")
    _install_inert_modules()
    sys.path[:0] = [REF, os.path.join(REF, "lib")]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from tests.golden import cases
    from model import getModel, fusionDecode
    from utils.pointcloud import getPcFrustumHeatmap, cvtPcDepthToHeatmap
    from model.utils import topk as ref_topk, nms as ref_nms

    # ---- 1. model forward, small resolution, both configs -----------------------------------
    for tag, radar, B, H, W in (("centerfusion_small", True, 2, 128, 160),
                                ("centernet_small", False, 1, 96, 128)):
        cfg = reference_config(radar, H, W)
        model = getModel(cfg).eval()
        sd = cases.tuned_state_dict(radar=radar, seed=0)
        assert set(sd.keys()) == set(model.state_dict().keys()), \
            set(sd.keys()) ^ set(model.state_dict().keys())
        model.load_state_dict(sd, strict=True)
        x, pc_dep, calib = cases.model_inputs(B, H, W, seed=1, radar=radar)
        with torch.no_grad():
            y = model(x, pc_dep=pc_dep.clone() if radar else None, calib=calib)[0]
        keys = [k for k in y if k != "calib" and y[k] is not None]
        arrays = {f"out_{k}": y[k] for k in keys}
        arrays["key_order"] = np.array(list(y.keys()))
        # per-stage outputs of the reference's own sub-modules (dla.py:627-635): backbone levels,
        # DLA-up outputs, the IDA-up feature map - stored as sampled values + a float64 sum each
        with torch.no_grad():
            levels = model.base(x)
            ups = model.dla_up([t.clone() for t in levels])
            yy = [ups[i].clone() for i in range(3)]
            model.ida_up(yy, 0, len(yy))
        for name, t in ([(f"y{i}", t) for i, t in enumerate(levels)] + [(f"up{i}", t) for i, t in enumerate(ups)]
                        + [("feat", yy[-1])]):
            flat = t.reshape(-1)
            idx = sample_idx(flat.numel(), m=2048, seed=321)
            arrays[f"stage_idx_{name}"] = idx
            arrays[f"stage_val_{name}"] = flat[idx]
            arrays[f"stage_sum_{name}"] = np.array(float(flat.double().sum()))
            arrays[f"stage_shape_{name}"] = np.array(t.shape)
        if radar:
            arrays["n_painted"] = np.array(int((y["pc_hm"] != 0).sum()))
            print("  frustum painted pixels:", int(arrays["n_painted"]))
        save(f"model_{tag}.npz", **arrays)

    # ---- 2. full resolution, bs=1, sampled values -------------------------------------------
    cfg = reference_config(True, 448, 800)
    model = getModel(cfg).eval()
    model.load_state_dict(cases.tuned_state_dict(radar=True, seed=0), strict=True)
    x, pc_dep, calib = cases.model_inputs(1, 448, 800, seed=2, radar=True, n_points=(80, 200))
    with torch.no_grad():
        y = model(x, pc_dep=pc_dep.clone(), calib=calib)[0]
    arrays = {}
    for k, v in y.items():
        if k == "calib":
            continue
        flat = v.reshape(-1)
        idx = sample_idx(flat.numel())
        arrays[f"idx_{k}"] = idx
        arrays[f"val_{k}"] = flat[idx]
        arrays[f"sum_{k}"] = np.array(float(flat.double().sum()))
    arrays["n_painted"] = np.array(int((y["pc_hm"] != 0).sum()))
    print("  full-res frustum painted pixels:", int(arrays["n_painted"]))
    with torch.no_grad():
        det = fusionDecode([dict(y)], outputSize=(112, 200), K=100, norm2d=False)
    for k, v in det.items():
        arrays[f"det_{k}"] = v
    save("model_centerfusion_fullres.npz", **arrays)

    # ---- 3. frustum association on hand-built head dicts ------------------------------------
    cfg = reference_config(True, 448, 800)
    for seed in (0, 1, 2):
        y, pc_dep, calib = cases.frustum_case(seed)
        yy = {k: v.clone() for k, v in y.items()}
        pc_hm = getPcFrustumHeatmap(yy, pc_dep.clone(), calib, cfg)
        s, inds, cls, ys, xs = ref_topk(y["heatmap"], K=100)
        nz = torch.nonzero(pc_hm.reshape(-1)).reshape(-1)
        save(f"frustum_{seed}.npz", nz_idx=nz.int(), nz_val=pc_hm.reshape(-1)[nz],
             shape=np.array(pc_hm.shape), topk_scores=s, topk_inds=inds.int(), topk_cls=cls,
             topk_ys=ys.int(), topk_xs=xs.int())
        print(f"  frustum seed {seed}: {nz.numel()} painted values")

    # the two hand cases of SURVEY.md Appendix B.5 (reference function called directly)
    pc_dep = torch.zeros(3, 112, 200)
    pc_dep[0, 40:60, 0:10] = 10.0
    pc_dep[1, 40:60, 0:10] = 1.5
    pc_dep[2, 40:60, 0:10] = -2.5
    res = {}
    for name, box in (("neg", (-1.5, 42.0, 7.5, 58.0)), ("pos", (0.5, 42.0, 7.5, 58.0))):
        pc_hm = torch.zeros(3, 112, 200)
        cvtPcDepthToHeatmap(pc_hm, pc_dep, torch.tensor(10.5), torch.tensor(box),
                            torch.tensor(2.0), 60.0)
        nz = torch.nonzero(pc_hm[0])
        res[f"{name}_nz"] = nz.int()
        res[f"{name}_val"] = pc_hm[:, nz[:, 0], nz[:, 1]] if nz.numel() else np.zeros((3, 0))
    save("frustum_handcases.npz", **res)

    # ---- 4. decode -----------------------------------------------------------------------
    for seed, radar in ((0, True), (1, False)):
        out = cases.decode_case(seed, radar=radar)
        det = fusionDecode([{k: v.clone() for k, v in out.items()}], outputSize=(112, 200),
                           K=100, norm2d=False)
        save(f"decode_{seed}.npz", **{k: v for k, v in det.items()})
    out = cases.decode_case(2, radar=True)
    det = fusionDecode([{k: v.clone() for k, v in out.items()}], outputSize=(112, 200), K=100,
                       norm2d=True)
    save("decode_2_norm2d.npz", **{k: v for k, v in det.items()})
    # ---- 5. postProcess (2D -> 3D) on decoded detections --------------------------------------
    from utils.postProcess import postProcess
    for seed in (0, 1):
        out = cases.decode_case(seed, radar=True)
        out["depth2"] = out["depth2"].abs() * 20 + 2          # plausible positive depths
        out["dimension"] = out["dimension"].abs() + 0.1
        if seed == 1:
            out["dimension"][:, 1] -= 0.6                     # some non-positive dims -> zeroed boxes
        det = fusionDecode([{k: v.clone() for k, v in out.items()}], outputSize=(112, 200), K=100)
        calibs = cases.model_inputs(2, 448, 800, seed=0)[2]
        pp = postProcess({k: v.clone() for k, v in det.items()}, np.array([800.0, 450.0], np.float32), 1600.0,
                         112, 200, calibs)
        save(f"postprocess_{seed}.npz", **{k: v for k, v in pp.items()})

    # tie-heavy: only the strictly-distinct prefix and the selected *set* are well defined
    out = cases.decode_case(3, radar=True, tie_heavy=True)
    heat = ref_nms(out["heatmap"])
    s, inds, cls, ys, xs = ref_topk(heat, K=100)
    save("decode_3_ties.npz", scores=s, inds=inds.int(), cls=cls)


if __name__ == "__main__":
    main()
