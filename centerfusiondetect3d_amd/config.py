"""Minimal configuration object with the attribute names the reference's yacs config exposes on
the hot path (config/default.py:3-88, derived fields of config/utils.py:69-204).  Any object with
the same attributes (e.g. the reference's own CfgNode) is accepted by the model; this module only
exists so the package works without yacs.
"""
import copy


class CfgNode(dict):
    """dict with attribute access; defrost()/freeze() are accepted and ignored."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return CfgNode({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def defrost(self):
        pass

    def freeze(self):
        pass


def _base():
    c = CfgNode()
    c.NAME = "CenterFusion"
    c.DATASET = CfgNode(DATASET="nuscenes", RADAR_PC=True, MAX_PC=1000, MAX_PC_DIST=60.0,
                        PC_Z_OFFSET=0.0, PC_ROI_METHOD="pillars", PILLAR_DIMS=(1.5, 0.2, 0.2),
                        ONE_HOT_PC=False, NUM_CLASSES=10, PC_REVERSE=True)
    c.MODEL = CfgNode(LOAD_DIR="", ARCH="dla_34", FREEZE_BACKBONE=False, NORM_EVAL=False,
                      NORM_2D=False, FUSION_STRATEGY="middle", FRUSTUM=True, K=100,
                      INPUT_SIZE=(448, 800), OUTPUT_SIZE=(112, 200),
                      DLA=CfgNode(NODE="DeformConv"))
    c.TRAIN = CfgNode(UNCERTAINTY_LOSS=False)
    c.TEST = CfgNode(BATCH_SIZE=1)
    return c


def update_heads(config):
    """heads / head_conv exactly as config/utils.py:69-166 derives them."""
    heads = {"heatmap": config.DATASET.NUM_CLASSES, "reg": 2, "widthHeight": 2, "depth": 1,
             "rotation": 8, "dimension": 3, "amodal_offset": 2}
    if config.DATASET.DATASET == "nuscenes":
        heads.update({"nuscenes_att": 8, "velocity": 3})
    middle = config.DATASET.RADAR_PC and config.MODEL.FUSION_STRATEGY == "middle"
    if middle:
        heads.update({"depth2": 1, "rotation2": 8})
    if config.TRAIN.UNCERTAINTY_LOSS:
        heads.update({"uncertainty": 1})
    head_conv = {h: [256] for h in heads}
    if config.DATASET.RADAR_PC:
        if middle:
            head_conv.update({"depth2": [256, 256, 256], "rotation2": [256, 256, 256]})
        if config.DATASET.DATASET == "nuscenes":
            head_conv.update({"velocity": [256, 256, 256], "nuscenes_att": [256, 256, 256]})
    config.heads = CfgNode(heads)
    config.head_conv = CfgNode(head_conv)
    return config


def centerfusion_middle_config(input_size=(448, 800)):
    """configs/Centerfusion_Middle.yaml: DLA-34 + radar middle fusion with frustum association."""
    c = _base()
    c.NAME = "CenterFusion_Middle"
    c.MODEL.INPUT_SIZE = tuple(input_size)
    c.MODEL.OUTPUT_SIZE = (input_size[0] // 4, input_size[1] // 4)
    return update_heads(c)


def centernet_config(input_size=(448, 800)):
    """configs/CenterNet.yaml: camera-only DLA-34."""
    c = _base()
    c.NAME = "CenterNet"
    c.DATASET.RADAR_PC = False
    c.MODEL.FUSION_STRATEGY = None
    c.MODEL.FRUSTUM = False
    c.MODEL.INPUT_SIZE = tuple(input_size)
    c.MODEL.OUTPUT_SIZE = (input_size[0] // 4, input_size[1] // 4)
    return update_heads(c)
