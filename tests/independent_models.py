"""Independent models the configs[2] oracle (oracle/augment.c) is pinned against.  None of them shares code or
structure with the oracle: they are the float64 textbook definitions the REFERENCE'S OWN TESTS compare its operators
with, restated in numpy because OpenCV is not in this image.

  gaussian_kernel   OpenCV's published cv::getGaussianKernel for sigma > 0 (what gaussian_blur_params_test.cc:33-57
                    holds FillGaussian to, 1e-7): exp(-x^2 / (2 sigma^2)) in double, normalised to sum 1, cast to float
  color_twist       the numpy model of dali/test/python/operator_1/test_color_twist.py:68-104 (hue rotation and
                    saturation scaling in YIQ, contrast about mid-grey, brightness), float64 throughout
  hsv               same with the value multiplier of fn.hsv (the reference's docs: "scaled based on the value and
                    saturation multipliers")
  warp_affine       exact bilinear (or nearest) sampling at M * (x + 0.5, y + 0.5) - 0.5 with a constant or clamped border: what
                    cv2.warpAffine(WARP_INVERSE_MAP | INTER_LINEAR) computes for the matrix ToCVMatrix builds in
                    operator_2/test_warp.py:47-52 (OpenCV itself interpolates with 5-bit fixed-point weights, which is why
                    the reference allows 8 LSB against it, test_warp.py:236; the exact model is tighter)
  convolve_reflect101 separable float64 convolution with cv2.BORDER_REFLECT_101 (operator_1/test_gaussian_blur.py:60-90)
"""
import math

import numpy as np


def gaussian_kernel(size, sigma):
    assert sigma > 0 and size % 2 == 1
    x = np.arange(size, dtype=np.float64) - (size - 1) * 0.5
    k = np.exp(-0.5 / (sigma * sigma) * x * x)
    return (k / k.sum()).astype(np.float32)


def opencv_sigma_of_size(size):
    return ((size - 1) * 0.5 - 1) * 0.3 + 0.8


def opencv_size_of_sigma(sigma):
    """DALI's rule (gaussian_blur_params.h:27-33): 2 * ceil(3 sigma) + 1."""
    return 2 * int(math.ceil(sigma * 3)) + 1


_RGB2YIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.321], [0.211, -0.523, 0.311]], np.float64)


def color_twist(img_u8, hue_deg=0.0, saturation=1.0, brightness=1.0, contrast=1.0, value=1.0):
    a = math.radians(hue_deg)
    rot = np.array([[1, 0, 0], [0, math.cos(a), math.sin(a)], [0, -math.sin(a), math.cos(a)]], np.float64)
    rot[1:] *= saturation
    m = np.linalg.inv(_RGB2YIQ) @ rot @ (_RGB2YIQ * value)
    px = img_u8.reshape(-1, 3).astype(np.float64) @ m.T
    px = ((px - 128.0) * contrast + 128.0) * brightness
    return np.round(np.clip(px, 0, 255)).astype(np.uint8).reshape(img_u8.shape)


def warp_affine(img_u8, matrix_dst_to_src, out_hw, fill=None, interp="linear"):
    """fill: None = clamp to the border pixels, else a constant (scalar or per channel)."""
    h, w, c = img_u8.shape
    oh, ow = out_hw
    m = np.asarray(matrix_dst_to_src, np.float64).reshape(2, 3)
    ys, xs = np.mgrid[0:oh, 0:ow].astype(np.float64)
    sx = m[0, 0] * (xs + 0.5) + m[0, 1] * (ys + 0.5) + m[0, 2]
    sy = m[1, 0] * (xs + 0.5) + m[1, 1] * (ys + 0.5) + m[1, 2]
    src = img_u8.astype(np.float64)
    fillv = None if fill is None else np.round(np.clip(np.broadcast_to(np.asarray(fill, np.float64), (c,)), 0, 255))

    def tap(ix, iy):
        inside = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
        v = src[np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)]
        if fillv is not None:
            v = np.where(inside[..., None], v, fillv)
        return v

    if interp == "nearest":
        return tap(np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)).astype(np.uint8)
    fx, fy = sx - 0.5, sy - 0.5
    x0, y0 = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    qx, qy = (fx - x0)[..., None], (fy - y0)[..., None]
    top = tap(x0, y0) * (1 - qx) + tap(x0 + 1, y0) * qx
    bot = tap(x0, y0 + 1) * (1 - qx) + tap(x0 + 1, y0 + 1) * qx
    return np.round(np.clip(top * (1 - qy) + bot * qy, 0, 255)).astype(np.uint8)


def rotate_output_size(angle_deg, in_hw):
    """The reference's own model of the rotated canvas (operator_2/test_rotate.py:39-60): the bounding box of the turned
    image, grown by one where its parity differs from the input side that dominates it."""
    # (the operator's angle is a float32 in radians: at exactly 45 degrees that decides which side "dominates")
    a = float(np.float32(-angle_deg) * np.float32(math.pi / 180))
    cosa, sina = abs(math.cos(a)), abs(math.sin(a))
    h, w = in_hw
    out_w, out_h = int(math.ceil(w * cosa + h * sina - 1e-2)), int(math.ceil(h * cosa + w * sina - 1e-2))
    ref_w, ref_h = (w, h) if sina <= cosa else (h, w)
    return out_h + (out_h % 2 != ref_h % 2), out_w + (out_w % 2 != ref_w % 2)


def rotate_matrix(angle_deg, in_hw, out_hw):
    """Destination -> source, float64: move the output centre to the origin, turn, move to the input centre
    (test_rotate.py:111-120)."""
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    (ih, iw), (oh, ow) = in_hw, out_hw
    return np.array([[c, -s, iw / 2 - c * ow / 2 + s * oh / 2], [s, c, ih / 2 - s * ow / 2 - c * oh / 2]], np.float64)


def convolve_reflect101(img_u8, win_x, win_y):
    def along(a, win, axis):
        r = (len(win) - 1) // 2
        pad = [(0, 0)] * a.ndim
        pad[axis] = (r, r)
        p = np.pad(a, pad, mode="reflect")
        out = np.zeros_like(a)
        for k, wk in enumerate(np.asarray(win, np.float64)):
            sl = [slice(None)] * a.ndim
            sl[axis] = slice(k, k + a.shape[axis])
            out += wk * p[tuple(sl)]
        return out
    a = along(img_u8.astype(np.float64), win_x, 1)
    return along(a, win_y, 0)
