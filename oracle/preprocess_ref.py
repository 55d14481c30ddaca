"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - CPU restatement of the image side of
`Detector.pre_process` (reference src/lib/detector.py:206-234): affine warp of the uint8 camera
frame to the network input size, /255, mean / std, HWC -> CHW.

PARITY UNPINNED.  The warp is `cv2.warpAffine(..., flags=cv2.INTER_LINEAR)` - OpenCV is a third-party
dependency that is absent from /root/reference and from this image (the reference's requirements.txt
asks for `opencv-python` without a version; any 4.x release has the algorithm below).  What follows
restates OpenCV's published fixed-point algorithm for 8-bit images (modules/imgproc/src/imgwarp.cpp:
cv::warpAffine -> WarpAffineInvoker -> remapBilinear<FixedPtCast<int, uchar, 15>>):

  * the 2x3 matrix is inverted in float64 (dst -> src map), in OpenCV's operation order;
  * source coordinates are evaluated in fixed point with 10 fractional bits:
        X = (cvRound((M1*y + M2)*1024) + 16 + cvRound(M0*x*1024)) >> 5          (5 fractional bits left)
    (cvRound = round half to even; 16 = half of a 1/32 step; likewise Y);
  * sx = X >> 5, fx = X & 31; the four taps are weighted by the integer products
        (32-fx)(32-fy)*32, fx(32-fy)*32, (32-fx)fy*32, fx*fy*32       (they sum to 2^15 exactly)
    and the result is (sum + 2^14) >> 15;  taps outside the image read the constant border value 0.
    (OpenCV stores the weights as int16 and patches the one entry that does not fit, fx = fy = 0, to
     {32767, 0, 0, 1}; that rounds to the same byte for every input, so the exact products are used.)

It is held by known-answer tests (tests/test_oracle_preprocess.py): identity, integer translation, the
nuScenes 1600x900 -> 800x448 case (an exact 2x decimation), half-pixel averaging with round-half-up.
"""
import numpy as np

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, TAB = 1 << AB_BITS, 1 << INTER_BITS


def invert_affine(M):
    """cv::warpAffine's in-place inversion of the forward 2x3 matrix (float64, its operation order)."""
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


def _cv_round(v):
    return np.rint(v).astype(np.int64)          # round half to even, like cvRound / lrint


def warp_affine_u8(img, M, dsize):
    """img (H, W, C) uint8, M forward 2x3, dsize = (width, height) -> (height, width, C) uint8."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    Hs, Ws, _ = img.shape
    Wd, Hd = dsize
    Mi = invert_affine(M)
    x = np.arange(Wd, dtype=np.float64)
    y = np.arange(Hd, dtype=np.float64)
    adelta, bdelta = _cv_round(Mi[0] * x * AB_SCALE), _cv_round(Mi[3] * x * AB_SCALE)
    rd = AB_SCALE // TAB // 2
    X0 = _cv_round((Mi[1] * y + Mi[2]) * AB_SCALE) + rd
    Y0 = _cv_round((Mi[4] * y + Mi[5]) * AB_SCALE) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx, sy, fx, fy = X >> INTER_BITS, Y >> INTER_BITS, X & (TAB - 1), Y & (TAB - 1)
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < Hs) & (xx >= 0) & (xx < Ws)
        v = src[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)]
        return v * ok[..., None]

    w00, w01 = (TAB - fx) * (TAB - fy) * TAB, fx * (TAB - fy) * TAB
    w10, w11 = (TAB - fx) * fy * TAB, fx * fy * TAB
    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] +
           tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None])
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def pre_process_images(images, M, input_size, mean, std):
    """images: list of (H, W, 3) uint8 frames (detector.py:226-234) -> (B, 3, inH, inW) float32.
    ((warp / 255.0 - mean) / std) is evaluated in float64 with float32 mean / std, then cast."""
    inH, inW = input_size
    mean = np.asarray(mean, np.float32).astype(np.float64)
    std = np.asarray(std, np.float32).astype(np.float64)
    out = []
    for im in images:
        w = warp_affine_u8(im, M, (inW, inH))
        out.append(((w / 255.0 - mean) / std).astype(np.float32).transpose(2, 0, 1))
    return np.stack(out, 0)
