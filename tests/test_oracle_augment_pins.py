"""Pins for oracle/augment.c (BASELINE configs[2]) against INDEPENDENT models - the ones the reference's own tests hold
its operators to (tests/independent_models.py), at the reference's own bounds or tighter:

  Gaussian windows  gaussian_blur_params_test.cc:33-57: every (size, sigma) pair of that test, 1e-7 per coefficient
  Gaussian blur     operator_1/test_gaussian_blur.py:134,164: max_allowed_error = 1 against the float convolution
  color twist       operator_1/test_color_twist.py:107-118: abs 1 / rel 1/512 against the numpy model
  warp affine       operator_2/test_warp.py:217-236: the reference allows 8 against OpenCV; against the exact bilinear model
                    the oracle stays within 1 LSB on noise and is equal on all but a fraction of the pixels
  erase             operator_1/test_erase.py: integer region arithmetic, exact against a numpy slice assignment
The product kernels (HIP and host) equal the oracle bit for bit (tests/test_gpu_augment.py, test_cpu_backend.py), so these
bounds carry over to them."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from tests import independent_models as M
from tests.util import synth_image

# gaussian_blur_params_test.cc:34-37
SIZE_SIGMA = [(1, 0), (3, 0), (5, 0), (7, 0), (9, 0), (11, 0), (13, 0), (15, 0), (101, 0), (0, 0.025), (0, 0.25), (0, 0.5),
              (0, 0.75), (0, 1.0), (0, 1.25), (0, 1.5), (0, 2.0), (0, 3.0), (0, 5.0), (0, 16.0)]


@pytest.mark.parametrize("size,sigma", SIZE_SIGMA)
def test_gaussian_window_known_answers(size, sigma):
    w = O.gaussian_window(sigma=float(sigma), window_size=int(size))
    if size == 0:
        size = M.opencv_size_of_sigma(np.float32(sigma))
    elif sigma == 0:
        sigma = M.opencv_sigma_of_size(size)
    assert w.size == size
    ref = M.gaussian_kernel(size, float(np.float32(sigma)))
    assert np.abs(w - ref).max() <= 1e-7, (size, sigma, np.abs(w - ref).max())
    assert abs(float(w.astype(np.float64).sum()) - 1.0) < 1e-6 and np.array_equal(w, w[::-1])


@pytest.mark.parametrize("sigma,shape", [(3.0, (96, 120)), (1.0, (40, 33)), (0.5, (17, 64)), (5.0, (64, 48))])
def test_gaussian_blur_within_one_of_the_float_convolution(sigma, shape):
    rng = np.random.default_rng(int(sigma * 10))
    for img in (synth_image(rng, *shape), rng.integers(0, 256, (*shape, 3), dtype=np.uint8)):
        win = O.gaussian_window(sigma=sigma)
        got = O.gaussian_blur_u8(img, win)
        exact = M.convolve_reflect101(img, M.gaussian_kernel(win.size, sigma), M.gaussian_kernel(win.size, sigma))
        assert np.abs(got.astype(np.float64) - exact).max() <= 0.5 + 2e-3          # rounding of the exact value +- float32 error
        assert np.abs(got.astype(int) - np.round(exact).astype(int)).max() <= 1   # the reference's own bound


def test_color_twist_against_the_reference_numpy_model():
    """Parameter ranges of operator_1/test_color_twist.py:38-47 (hue +-... degrees, saturation / brightness / contrast in
    [0, 2]) on random and structured images; u8 -> u8."""
    rng = np.random.default_rng(2139)
    worst, differing, total = 0, 0, 0
    for it in range(40):
        img = rng.integers(0, 256, (128, 32, 3), dtype=np.uint8) if it % 2 else synth_image(rng, 64, 64)
        hue = float(rng.uniform(-180, 180)) if it else 0.0
        sat, bri, con = (float(rng.uniform(0, 2)) for _ in range(3))
        m, off = O.color_twist_matrix(hue, sat, 1.0, bri, con)
        got = O.linear_transform_u8(img, m, off)
        ref = M.color_twist(img, hue, sat, bri, con)
        d = np.abs(got.astype(int) - ref.astype(int))
        worst, differing, total = max(worst, int(d.max())), differing + int((d > 0).sum()), total + d.size
        assert np.allclose(got, ref, rtol=1 / 512, atol=1), (hue, sat, bri, con)
    assert worst <= 1 and differing / total < 5e-3, (worst, differing / total)   # only rounding ties flip


def test_hsv_value_and_pure_hue_rotations():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    for hue, sat, val in [(0, 1, 0.5), (120, 1, 1), (-37.5, 0.3, 1.4), (360, 1, 1), (90, 0, 1)]:
        m, off = O.color_twist_matrix(hue, sat, val, 1.0, 1.0)
        got = O.linear_transform_u8(img, m, off)
        ref = M.color_twist(img, hue, sat, 1.0, 1.0, value=val)
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1, (hue, sat, val)
    m, off = O.color_twist_matrix(360.0, 1.0, 1.0, 1.0, 1.0)          # a whole turn is the identity up to rounding
    assert np.abs(O.linear_transform_u8(img, m, off).astype(int) - img).max() <= 1
    m, off = O.color_twist_matrix(0.0, 0.0, 1.0, 1.0, 1.0)            # no saturation: every channel is the luma
    g = O.linear_transform_u8(img, m, off)
    assert np.abs(g[..., 0].astype(int) - g[..., 1]).max() <= 1 and np.abs(g[..., 1].astype(int) - g[..., 2]).max() <= 1


def _reference_transform(angle, zoom, dst_cx, dst_cy, src_cx, src_cy):
    """The family of operator_2/test_warp.py:31-46 (rotation about a centre, zoom, re-centring), built here directly."""
    c, s = math.cos(angle) / zoom, math.sin(angle) / zoom
    return np.array([[c, -s, src_cx - c * dst_cx + s * dst_cy], [s, c, src_cy - s * dst_cx - c * dst_cy]], np.float32)


WARP_MATRICES = [np.array([[0.1, 0.9, 10], [0.8, -0.2, -20]], np.float32)] + \
                [_reference_transform(math.radians(10 * i), 2, 160, 120, 100, 100) for i in range(0, 19, 3)]


@pytest.mark.parametrize("mi", range(len(WARP_MATRICES)))
@pytest.mark.parametrize("fill", [42.0, None])
def test_warp_affine_against_the_exact_bilinear_model(mi, fill):
    """Output 240 x 320 with fill 42 as test_warp.py:70-76; smooth + textured image and noise."""
    rng = np.random.default_rng(1009 + mi)
    m = WARP_MATRICES[mi]
    for img, frac in ((synth_image(rng, 200, 260), 0.06), (rng.integers(0, 256, (200, 260, 3), dtype=np.uint8), 0.06)):
        got = O.warp_affine_u8(img, m, out_hw=(240, 320), interp=1, fill=fill)
        ref = M.warp_affine(img, m, (240, 320), fill=fill)
        d = np.abs(got.astype(int) - ref.astype(int))
        # float32 incremental source coordinates against exact ones: 1e-4 px x up to 255 of contrast flips roundings only
        assert d.max() <= 1, (mi, d.max())
        assert (d > 0).mean() <= frac, (mi, (d > 0).mean())
        assert d.max() <= 8      # the reference's bound against OpenCV, for the record


@pytest.mark.parametrize("mi", [1, 3])     # (matrix 0 puts every tenth source coordinate exactly ON a pixel edge)
def test_warp_affine_nearest_and_forward_matrices(mi):
    rng = np.random.default_rng(7 + mi)
    img = synth_image(rng, 120, 150)
    m = WARP_MATRICES[mi]
    got = O.warp_affine_u8(img, m, out_hw=(100, 140), interp=0, fill=7.0)
    ref = M.warp_affine(img, m, (100, 140), fill=7.0, interp="nearest")
    assert (got != ref).mean() < 2e-3          # a source coordinate within 1e-4 of a pixel edge may pick the neighbour
    # inverse_map=False: the oracle inverts the matrix like affine_mat_inv; numpy's inverse is the independent answer
    fwd = np.vstack([np.linalg.inv(np.vstack([m.astype(np.float64), [0, 0, 1]]))[:2]]).astype(np.float32)
    back = O.affine_inverse(fwd)
    assert np.allclose(back, m, rtol=1e-5, atol=1e-4)


def test_erase_against_numpy_slices():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    got = O.erase_u8(img, [(10, 20), (50, 70)], [(15, 30), (40, 40)], fill=(1.0, 2.0, 3.0))
    ref = img.copy()
    ref[10:25, 20:50] = (1, 2, 3)
    ref[50:60, 70:80] = (1, 2, 3)
    assert np.array_equal(got, ref)
    got = O.erase_u8(img, [(0.5, 0.5)], [(0.5, 0.25)], fill=(9.0,), normalized_anchor=True, normalized_shape=True,
                     centered_anchor=True)
    ref = img.copy()
    ref[15:45, 30:50] = 9
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("angle", [0.0, 10.0, -33.3, 45.0, 90.0, 135.0, 180.0, 270.0, 301.5])
@pytest.mark.parametrize("hw", [(120, 161), (97, 64), (50, 50)])
def test_rotate_canvas_and_pixels(angle, hw):
    """fn.rotate: canvas size = the reference's own model (operator_2/test_rotate.py:39-60), pixels = exact bilinear sampling
    through the float64 matrix of test_rotate.py:111-120 (the reference allows 8 against OpenCV, :177)."""
    rng = np.random.default_rng(int(abs(angle) * 10) + hw[0])
    img = synth_image(rng, *hw)
    m, out_hw = O.rotate_params(angle, hw)
    assert out_hw == M.rotate_output_size(angle, hw), (angle, hw)
    assert np.allclose(m, M.rotate_matrix(angle, hw, out_hw), rtol=1e-5, atol=2e-4)
    got = O.rotate_u8(img, angle, fill=42.0)
    ref = M.warp_affine(img, M.rotate_matrix(angle, hw, out_hw), out_hw, fill=42.0)
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 1 or (d > 1).mean() < 2e-3, (d.max(), (d > 1).mean())   # (multiples of 90 degrees put coordinates ON pixel edges)
    assert (d > 0).mean() < 0.08
    # keep_size / explicit size keep the canvas and re-centre
    m2, hw2 = O.rotate_params(angle, hw, keep_size=True)
    assert hw2 == hw and np.allclose(m2, M.rotate_matrix(angle, hw, hw), rtol=1e-5, atol=2e-4)
    m3, hw3 = O.rotate_params(angle, hw, size=(70, 33))
    assert hw3 == (70, 33) and np.allclose(m3, M.rotate_matrix(angle, hw, (70, 33)), rtol=1e-5, atol=2e-4)
