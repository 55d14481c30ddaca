"""centerfusiondetect3d_amd - the CenterFusion inference forward path as hand-written HIP kernels
for AMD MI355X (gfx950), behind the reference project's model / decode interfaces.

    from centerfusiondetect3d_amd import getModel, fusionDecode, centerfusion_middle_config
    model = getModel(centerfusion_middle_config()).cuda().eval()
    outputs = model(images, pc_dep=pc_dep, calib=calib)      # [ {head: (B,C,112,200)} ]
    dets = fusionDecode(outputs, outputSize=(112, 200), K=100)
"""
from .config import CfgNode, centerfusion_middle_config, centernet_config, update_heads
from .model import DLASeg, getModel
from .decode import fusionDecode, decode_packed, decode_post_packed, unpack_detections, DET_FIELDS, DET_WIDTH
from .pointcloud import getPcFrustumHeatmap, getAffineTransform, process_point_cloud_batch, radar_to_pc_dep
from .postprocess import postProcess, post_process_packed, unpack_post, POST_FIELDS, POST_WIDTH
from .preprocess import preProcessImages
from .serialize import NuScenesResults, convert_eval_format
from .detector import Detector

__all__ = ["Detector", "NuScenesResults", "convert_eval_format", "decode_post_packed", "preProcessImages", "radar_to_pc_dep", "CfgNode", "centerfusion_middle_config", "centernet_config", "update_heads", "DLASeg",
           "getModel", "fusionDecode", "decode_packed", "unpack_detections", "DET_FIELDS",
           "DET_WIDTH", "getPcFrustumHeatmap", "getAffineTransform", "process_point_cloud_batch", "postProcess",
           "post_process_packed", "unpack_post", "POST_FIELDS", "POST_WIDTH"]
