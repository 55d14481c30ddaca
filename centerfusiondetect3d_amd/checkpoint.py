"""Checkpoint loading with the reference's legacy parameter names (model/model.py:47-250:
`loadModel`, `elasticLoadStateDict`, `toggleWeightName`).

Checkpoints of the original CenterFusion release and of earlier revisions of the reference name the
heads `hm. / wh. / dep. / dim. / rot. / amodel_offset. / dep_sec. / rot_sec.` (v1) or by their bare new
names without the `detectHead_0.` scope (v2), wrap the deformable convolutions in a `.conv` scope
(`proj_1.conv.weight`, `proj_1.conv.conv_offset_mask.weight`, `up_1.conv.weight`), call the BN after them
`actf`, and may carry a DataParallel `module.` prefix.  `to_new_name` maps any of these to the key of
`DLASeg.state_dict()`; tests/golden/legacy_keys.npz (generated with the reference's own
`toggleWeightName`) pins the mapping for all 434 keys in both directions."""
import re

import torch

_HEADS_V1 = {"dep_sec.": "depth2.", "rot_sec.": "rotation2.", "hm.": "heatmap.", "wh.": "widthHeight.",
             "dep.": "depth.", "dim.": "dimension.", "rot.": "rotation.", "amodel_offset.": "amodal_offset."}
_HEADS_NEW = ["reg", "depth2", "rotation2", "heatmap", "widthHeight", "depth", "rotation", "dimension",
              "amodal_offset", "nuscenes_att", "velocity"]
_UP_NODE = re.compile(r"^(.*_up.*_\d)\.conv\.(weight|bias)$")


def to_new_name(name: str) -> str:
    """legacy (v1 / v2 / DataParallel) or current parameter name -> current name."""
    if name.startswith("module.") and not name.startswith("module_list"):
        name = name[7:]
    m = _UP_NODE.match(name)
    if m:                                                   # deformable / upsample weights lost their .conv scope
        return f"{m.group(1)}.{m.group(2)}"
    name = name.replace(".conv.conv_offset_mask.", ".conv_offset_mask.").replace(".actf.", ".activation.")
    if name.startswith("detectHead_0."):
        return name
    for old, new in _HEADS_V1.items():                      # v1 head scopes
        if name.startswith(old):
            return "detectHead_0." + new + name[len(old):]
    for h in _HEADS_NEW:                                    # v2: new head names without the detectHead_0 scope
        if name.startswith(h + "."):
            return "detectHead_0." + name
    return name


def to_old_name(name: str, version: int = 1) -> str:
    """current name -> legacy name (version 1: hm./dep_sec. ...; version 2: bare head names)."""
    m = re.match(r"^(.*_up.*_\d)\.(weight|bias)$", name)
    if m:
        return f"{m.group(1)}.conv.{m.group(2)}"
    if ".conv_offset_mask." in name:
        return name.replace(".conv_offset_mask.", ".conv.conv_offset_mask.")
    if ".activation." in name:
        return name.replace(".activation.", ".actf.")
    if name.startswith("detectHead_0."):
        rest = name[len("detectHead_0."):]
        if version == 1:
            for old, new in _HEADS_V1.items():
                if rest.startswith(new):
                    return old + rest[len(new):]
            return name
        return rest
    return name


def elastic_load_state_dict(model, state_dict, verbose=False):
    """model/model.py:58-134: rename, drop unknown keys, keep the model's own tensor where a shape differs or a
    key is missing.  -> (model, report dict: loaded / skipped_shape / dropped / missing)."""
    own = model.state_dict()
    final, report = {}, {"loaded": [], "skipped_shape": [], "dropped": [], "missing": []}
    for k, v in state_dict.items():
        nk = to_new_name(k)
        if nk not in own:
            report["dropped"].append(k)
            continue
        if tuple(v.shape) != tuple(own[nk].shape):
            report["skipped_shape"].append(k)
            final[nk] = own[nk]
        else:
            final[nk] = v
            report["loaded"].append(nk)
    for k in own:
        if k not in final:
            report["missing"].append(k)
            final[k] = own[k]
    model.load_state_dict(final, strict=False)
    if verbose:
        print({k: len(v) for k, v in report.items()})
    return model, report


def loadModel(model, config):
    """model/model.py:137-166 for inference: -> (checkpoint, model, start_epoch)."""
    checkpoint = torch.load(config.MODEL.LOAD_DIR, map_location="cpu")
    start_epoch = 1
    if "epoch" in checkpoint and getattr(getattr(config, "TRAIN", None), "RESUME", False):
        start_epoch = checkpoint["epoch"] + 1
    sd = checkpoint["state_dict"]
    if sd.keys() != model.state_dict().keys():
        model, _ = elastic_load_state_dict(model, sd)
    else:
        model.load_state_dict(sd, strict=False)
    return checkpoint, model, start_epoch
