"""ONNX reference Resize (linear, with and without antialias, half_pixel coordinates), restated in numpy.

The reference pins its resampling against golden arrays generated with ONNX's reference implementation
(`onnx.backend.test.case.node.resize.interpolate_nd` with `linear_coeffs` / `linear_coeffs_antialias`) on a 22 x 22
checkerboard of 2 x 2 squares resized to 17 x 13, atol 1 (/root/reference/dali/test/python/operator_2/test_resize.py:919-1029;
the arrays live in DALI_extra, which is not mounted, and `onnx` is not installed here).  This file restates that
public algorithm (onnx/reference/ops/op_resize.py) so the same golden arrays can be regenerated: a pin of the
oracle's filter geometry that does not depend on the oracle's own rounding model."""
import numpy as np


def linear_coeffs(ratio, scale=None):
    return np.array([1 - ratio, ratio])


def linear_coeffs_antialias(ratio, scale):
    scale = min(scale, 1.0)                      # antialiasing only when down-sampling
    start = int(np.floor(-1 / scale) + 1)
    footprint = 2 - 2 * start
    args = (np.arange(start, start + footprint) - ratio) * scale
    coeffs = np.clip(1 - np.abs(args), 0, 1)
    return np.array(coeffs) / sum(coeffs)


def _get_neighbor_idxes(x, n, limit):
    """the n indexes nearest to x in [0, limit), preferring the smaller index on ties"""
    idxes = sorted(range(limit), key=lambda idx: (abs(x - idx), idx))[:n]
    return np.array(sorted(idxes))


def _get_neighbor(x, n, data):
    pad_width = int(np.ceil(n / 2))
    padded = np.pad(data, pad_width, mode="edge")
    idxes = _get_neighbor_idxes(x + pad_width, n, len(padded))
    return idxes - pad_width, padded[idxes]


def _interpolate_1d_with_x(data, scale_factor, x, get_coeffs):
    x_ori = (x + 0.5) / scale_factor - 0.5      # half_pixel
    x_ori_int = int(np.floor(x_ori))
    ratio = 1 if float(x_ori).is_integer() else x_ori - x_ori_int   # in (0, 1]: the pixel left of x_ori is preferred
    coeffs = get_coeffs(ratio, scale_factor)
    _, points = _get_neighbor(x_ori, len(coeffs), data)
    return float(np.dot(coeffs, points))


def _interpolate_nd_with_x(data, n, scale_factors, x, get_coeffs):
    if n == 1:
        return _interpolate_1d_with_x(data, scale_factors[0], x[0], get_coeffs)
    res1d = [_interpolate_nd_with_x(data[i], n - 1, scale_factors[1:], x[1:], get_coeffs) for i in range(data.shape[0])]
    return _interpolate_1d_with_x(np.array(res1d), scale_factors[0], x[0], get_coeffs)


def interpolate_nd(data, get_coeffs, output_size):
    data = np.asarray(data, np.float64)
    scale_factors = np.array(output_size) / np.array(data.shape)
    out = np.zeros(output_size, np.float64)
    for x in np.ndindex(*output_size):
        out[x] = _interpolate_nd_with_x(data, data.ndim, scale_factors, list(x), get_coeffs)
    return out


def checkerboard_22_22():
    """22 x 22 checkerboard of 2 x 2 squares (DALI_extra db/imgproc/checkerboard_22_22.npy), 0 / 255."""
    yy, xx = np.mgrid[0:22, 0:22]
    return (((yy // 2 + xx // 2) % 2) * 255).astype(np.uint8)
