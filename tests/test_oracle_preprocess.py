"""Known-answer tests of oracle/preprocess_ref.py (cv2.warpAffine restatement; parity unpinned: cv2 is
not in the image) - each case is one whose exact bytes follow from the published algorithm."""
import numpy as np

from oracle import preprocess_ref as pp


def _img(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, size=(h, w, 3)).astype(np.uint8)


def test_identity_and_integer_translation():
    im = _img(20, 31)
    assert np.array_equal(pp.warp_affine_u8(im, [[1, 0, 0], [0, 1, 0]], (31, 20)), im)
    out = pp.warp_affine_u8(im, [[1, 0, 3], [0, 1, -2]], (31, 20))      # dst(x, y) = src(x - 3, y + 2)
    assert np.array_equal(out[:18, 3:], im[2:, :28])
    assert not out[:, :3].any() and not out[18:].any()                   # constant border 0


def test_nuscenes_case_is_exact_decimation():
    # 1600x900 -> 800x448 with centre (800, 450), scale 1600 (utils/image.py:43-83): M = [[.5, 0, 0], [0, .5, -1]]
    im = _img(90, 160, seed=1)                                           # same geometry, 1/10 size
    M = [[0.5, 0.0, 0.0], [0.0, 0.5, 45 * 0.5 - 45 + 22.0]]              # maps centre (80, 45) to (40, 22)
    out = pp.warp_affine_u8(im, M, (80, 44))
    assert np.array_equal(out, im[1:89:2, 0:160:2])                      # src = (2x, 2y + 1)


def test_half_pixel_shift_rounds_half_up():
    im = np.zeros((1, 4, 1), np.uint8)
    im[0, :, 0] = [10, 13, 200, 255]
    out = pp.warp_affine_u8(np.repeat(im, 3, 0), [[1, 0, -0.5], [0, 1, 0]], (3, 3))
    # dst x reads src x + 0.5: (a + b + 1) >> 1
    assert out[1, :, 0].tolist() == [(10 + 13 + 1) >> 1, (13 + 200 + 1) >> 1, (200 + 255 + 1) >> 1]


def test_partial_border_taps_read_zero():
    im = np.full((4, 4, 1), 200, np.uint8)
    out = pp.warp_affine_u8(im, [[1, 0, 0.5], [0, 1, 0]], (5, 4))       # dst x reads src x - 0.5
    assert out[0, :, 0].tolist() == [100, 200, 200, 200, 100]           # half of a border tap is 0


def test_inverse_matches_float64_linear_algebra():
    M = np.array([[0.43, -0.1, 12.5], [0.07, 0.52, -3.25]])
    A = np.vstack([M, [0, 0, 1]])
    np.testing.assert_allclose(pp.invert_affine(M).reshape(2, 3), np.linalg.inv(A)[:2], rtol=1e-12, atol=1e-12)


def test_normalisation_is_float64_then_cast():
    im = _img(8, 8, seed=3)
    mean, std = np.float32([0.40789654, 0.44719302, 0.47026115]), np.float32([0.28863828, 0.27408164, 0.27809835])
    out = pp.pre_process_images([im], [[1, 0, 0], [0, 1, 0]], (8, 8), mean, std)
    ref = ((im / 255.0 - mean) / std).astype(np.float32).transpose(2, 0, 1)[None]
    assert out.dtype == np.float32 and np.array_equal(out, ref)


def test_general_affine_agrees_with_an_independent_float_bilinear_warp():
    """Geometry cross-check against a DIFFERENT implementation: torch's `F.grid_sample` (float bilinear, zero padding,
    pixel-centre coordinates) warps the same smooth frame through the same forward matrix - a rotation + anisotropic scale
    + shift that no known-answer case covers.  OpenCV's arithmetic quantises the source position to 1/32 pixel and the
    weights to 15 bits, so on a smooth image (gradient <= 2 levels per pixel) the two must agree within one grey level
    wherever all four taps lie inside the frame, and within 255 / 32 + 1 levels where a tap reads the zero border (the
    1/32-pixel position step times the jump at the edge): direction of the map, inversion, tap order and border handling
    are all exercised."""
    import torch
    import torch.nn.functional as F
    Hs, Ws, Hd, Wd = 60, 90, 50, 70
    yy, xx = np.mgrid[0:Hs, 0:Ws].astype(np.float64)
    img = np.stack([100 + 60 * np.sin(xx / 9.0) + 50 * np.cos(yy / 7.0), 20 + 1.9 * xx + 0.5 * yy,
                    128 + 100 * np.sin((xx + yy) / 11.0)], axis=-1)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    th = np.deg2rad(17.0)
    M = np.array([[0.8 * np.cos(th), -0.9 * np.sin(th), 6.3], [0.8 * np.sin(th), 0.9 * np.cos(th), -4.7]])
    got = pp.warp_affine_u8(img, M, (Wd, Hd)).astype(np.float64)
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)                                                # dst -> src, plain float64 linear algebra
    yd, xd = np.mgrid[0:Hd, 0:Wd].astype(np.float64)
    sx = Ai[0, 0] * xd + Ai[0, 1] * yd + Ai[0, 2]
    sy = Ai[1, 0] * xd + Ai[1, 1] * yd + Ai[1, 2]
    grid = torch.from_numpy(np.stack([2 * sx / (Ws - 1) - 1, 2 * sy / (Hs - 1) - 1], -1))[None]
    src = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    ref = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0].permute(1, 2, 0).numpy()
    inside = (sx >= 0) & (sx <= Ws - 1) & (sy >= 0) & (sy <= Hs - 1)
    err = np.abs(got - ref).max(-1)
    assert float(err[inside].max()) <= 1.0 + 1e-9, float(err[inside].max())
    assert float(err.max()) <= 255.0 / 32 + 1.0, float(err.max())
    assert int(inside.sum()) > 1500 and int((~inside).sum()) > 300
    assert int((ref == 0).all(-1).sum()) > 50 and int((ref > 0).all(-1).sum()) > 1000      # borders AND interior present
