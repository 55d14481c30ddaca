"""Image side of `Detector.pre_process` on the device (reference src/lib/detector.py:206-234,
SURVEY §8(f) rank 2): affine warp of the uint8 camera frames to the network input size
(cv2.warpAffine, INTER_LINEAR, in OpenCV's fixed-point arithmetic), /255, mean / std, HWC -> CHW.

    images = preProcessImages(frames, config.MODEL.INPUT_SIZE)       # (B, 3, inH, inW) fp32, on the GPU
replaces
    images = cv2.warpAffine(np.concatenate(frames, -1), transMatInput, (inW, inH), flags=cv2.INTER_LINEAR)
    images = ((images / 255.0 - mean) / std).astype(np.float32).transpose(2, 0, 1).reshape(-1, 3, inH, inW)
    images = torch.from_numpy(images) ... .to(device)
so only the raw bytes cross PCIe (4.3 MB/frame at 1600x900 instead of 4.3 MB of fp32 at 800x448 plus a
host-side warp), and nothing is computed on the host.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .pointcloud import getAffineTransform

NUSCENES_MEAN = np.array([0.40789654, 0.44719302, 0.47026115], dtype=np.float32)   # datasets/nuscenes.py:110
NUSCENES_STD = np.array([0.28863828, 0.27408164, 0.27809835], dtype=np.float32)    # datasets/nuscenes.py:111


def invert_affine(M):
    """The dst -> src map cv::warpAffine derives from the forward 2x3 matrix (float64, its operation order)."""
    M = np.asarray(M, np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def preProcessImages(imageOrigins, input_size, mean=NUSCENES_MEAN, std=NUSCENES_STD, transMat=None,
                     device=None, out=None):
    """imageOrigins: list of (H, W, 3) uint8 ndarrays / tensors of one size, or a (B, H, W, 3) uint8
    tensor (host or device).  input_size = (inH, inW).  transMat: forward 2x3 matrix; default = the
    detector's `getAffineTransform(center, max(H, W), 0, [inW, inH])` (detector.py:208-217)."""
    if isinstance(imageOrigins, (list, tuple)):
        frames = torch.stack([torch.as_tensor(np.ascontiguousarray(f) if isinstance(f, np.ndarray) else f)
                              for f in imageOrigins], 0)
    else:
        frames = torch.as_tensor(imageOrigins)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError(f"frames must be (B, H, W, 3) uint8, got {tuple(frames.shape)} {frames.dtype}")
    if device is None:
        device = frames.device if frames.is_cuda else torch.device("cuda", torch.cuda.current_device())
    if torch.device(device).type != "cuda":
        raise _lib.CfHipError("preProcessImages runs on the GPU: the HIP path has no CPU fallback")
    frames = frames.to(device, non_blocking=True).contiguous()
    B, Hs, Ws, _ = frames.shape
    inH, inW = int(input_size[0]), int(input_size[1])
    if transMat is None:
        center = np.array([Ws / 2.0, Hs / 2.0], dtype=np.float32)
        transMat = getAffineTransform(center, max(Hs, Ws) * 1.0, 0, [inW, inH])
    minv = (C.c_double * 6)(*invert_affine(transMat))
    mean_c = (C.c_float * 3)(*np.asarray(mean, np.float32).tolist())
    std_c = (C.c_float * 3)(*np.asarray(std, np.float32).tolist())
    if out is None:
        out = torch.empty((B, 3, inH, inW), device=device, dtype=torch.float32)
    with torch.cuda.device(device):
        _lib.check(_lib.load().cf_preprocess_images(frames.data_ptr(), B, Hs, Ws, minv, mean_c, std_c, inH, inW,
                                                    out.data_ptr(), _lib.stream_ptr()), "cf_preprocess_images")
    return out
