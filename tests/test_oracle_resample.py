"""Oracle pinning for resampling: the reference's own self-contained known answers
(dali/kernels/test/resampling_test/resampling_impl_cpu_test.cc:27-90) and its PIL cross-check with the
tolerances of dali/test/python/operator_2/test_resize.py:96-121,582-589 (mean abs err <= 0.4, max <= 10 on
the interior; PIL is installed, so this oracle is live)."""
import numpy as np
import pytest
from PIL import Image

from oracle import oracle as O
from tests.util import synth_image


def test_triangular_filter_known_answers():
    # TEST(ResampleCPU, TriangularFilter): 479 -> 93
    w, in_w = 93, 479
    scale = np.float32(in_w) / np.float32(w)
    assert O.triangular_support(float(scale)) == 11
    idx, coeffs = O.init_triangular(w, 0.0, float(scale), float(scale))
    support = coeffs.shape[1]
    for i in range(w):
        src_i = int(np.floor(np.float32(i + 0.5) * scale))
        max_k = int(np.argmax(coeffs[i]))
        assert abs(max_k - support // 2) <= 1
        slope = coeffs[i, 1] - coeffs[i, 0]
        assert coeffs[i, 0] < slope and coeffs[i, support - 1] < slope, "filter misses a contributing pixel"
        assert idx[i] + max_k == src_i, "filter maximum expected to coincide with NN pixel"
        assert abs(coeffs[i].sum() - 1) < 1e-6


def test_filter_symmetry():
    # TEST(ResampleCPU, FilterSymmetry) restated for the triangular filter (10 -> 9): the weight output i puts
    # on source pixel j equals the weight output w-1-i puts on pixel in_w-1-j.  (The reference checks its Gaussian
    # tap-by-tap; a triangular window of non-integer width starts at ceil(), so taps shift by one between mirror
    # images and the comparison has to go through source indices.)
    w, in_w = 9, 10
    scale = np.float32(in_w) / np.float32(w)
    idx, coeffs = O.init_triangular(w, 0.0, float(scale), float(scale))
    support = coeffs.shape[1]
    for i in range(w):
        wi = {int(idx[i]) + k: coeffs[i, k] for k in range(support) if coeffs[i, k] != 0}
        wm = {in_w - 1 - (int(idx[w - 1 - i]) + k): coeffs[w - 1 - i, k] for k in range(support)
              if coeffs[w - 1 - i, k] != 0}
        assert wi.keys() == wm.keys(), "symmetry broken"
        for j in wi:
            assert abs(wi[j] - wm[j]) < 1e-6, "symmetry broken"
    mid = coeffs[w // 2]
    nz = mid[mid != 0]
    assert abs(nz[0] - nz[-1]) < 1e-6, "central pixel should have a symmetrical kernel"


def test_linear_is_triangular_radius_1():
    assert O.triangular_support(1.0) == 2
    idx, coeffs = O.init_triangular(8, 0.0, 0.5, 1.0)  # 2x upscale
    assert coeffs.shape[1] == 2
    # sample centres at 0.25, 0.75, ... -> lerp weights (0.75, 0.25) / (0.25, 0.75)
    assert np.allclose(coeffs[1], [0.75, 0.25]) and np.allclose(coeffs[2], [0.25, 0.75])


def _interior(a):
    return a[2:-2, 2:-2]


@pytest.mark.parametrize("in_hw,out_hw", [((300, 400), (150, 200)), ((375, 500), (224, 224)), ((480, 640), (93, 93)),
                                          ((100, 120), (224, 224)), ((333, 500), (256, 171))])
def test_resize_vs_pil_bilinear(in_hw, out_hw):
    rng = np.random.default_rng(in_hw[0] + out_hw[1])
    img = synth_image(rng, *in_hw)
    got = O.resample_u8(img, out_hw)  # DALI default: triangular (antialias) when shrinking, linear when enlarging
    ref = np.asarray(Image.fromarray(img).resize((out_hw[1], out_hw[0]), Image.BILINEAR))
    d = np.abs(_interior(got).astype(np.int32) - _interior(ref).astype(np.int32))
    assert d.mean() <= 0.4 and d.max() <= 10, (d.mean(), d.max())


def test_roi_resize_vs_pil_box():
    rng = np.random.default_rng(11)
    img = synth_image(rng, 375, 500)
    anchors, crops = O.rrc_batch(5, 0, [img.shape[:2]] * 6)
    for a, c in zip(anchors, crops):
        roi = (a[0], a[1], a[0] + c[0], a[1] + c[1])
        got = O.resample_u8(img, (224, 224), roi=roi)
        ref = np.asarray(Image.fromarray(img).resize((224, 224), Image.BILINEAR, box=(roi[1], roi[0], roi[3], roi[2])))
        d = np.abs(_interior(got).astype(np.int32) - _interior(ref).astype(np.int32))
        assert d.mean() <= 0.4 and d.max() <= 10, (roi, d.mean(), d.max())


def test_identity_and_constant_images():
    rng = np.random.default_rng(2)
    img = synth_image(rng, 50, 70)
    assert np.array_equal(O.resample_u8(img, (50, 70)), img)          # scale 1: exact copy
    flat = np.full((40, 30, 3), 77, np.uint8)
    assert (O.resample_u8(flat, (17, 91)) == 77).all()                # normalised coefficients
    assert (O.resample_u8(flat, (224, 224), roi=(3.5, 2.25, 30.75, 20.5)) == 77).all()


def test_pass_order_cost_model_and_rounding_modes():
    rng = np.random.default_rng(3)
    img = synth_image(rng, 480, 640)
    # strong vertical-only shrink -> vertical pass first; strong horizontal-only shrink -> horizontal first
    _, info = O.resample_u8(img, (60, 640), return_info=True)
    assert info[0] == 1
    _, info = O.resample_u8(img, (480, 80), return_info=True)
    assert info[0] == 0
    # the three rounding models differ by at most 1 LSB and only on exact ties
    a = O.resample_u8(img, (240, 320), round_mode=0)
    b = O.resample_u8(img, (240, 320), round_mode=1)
    c = O.resample_u8(img, (240, 320), round_mode=2)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1 and np.abs(a.astype(int) - c.astype(int)).max() <= 1


def test_gradient_image_recovers_roi():
    """Property test of dali/test/python/operator_2/test_random_resized_crop.py:29-90: a gradient image encodes
    coordinates, so the resized crop reveals which source window it came from."""
    H, W = 300, 400
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([xx * 255 // (W - 1), yy * 255 // (H - 1), np.zeros_like(xx)], -1).astype(np.uint8)
    roi = (40, 100, 220, 340)
    out = O.resample_u8(img, (224, 224), roi=roi).astype(np.float32)
    x0, x1 = out[112, 2, 0] * (W - 1) / 255, out[112, -3, 0] * (W - 1) / 255
    y0, y1 = out[2, 112, 1] * (H - 1) / 255, out[-3, 112, 1] * (H - 1) / 255
    assert abs(x0 - (roi[1] + 2.5 * 240 / 224)) < 3 and abs(x1 - (roi[3] - 2.5 * 240 / 224)) < 3
    assert abs(y0 - (roi[0] + 2.5 * 180 / 224)) < 3 and abs(y1 - (roi[2] - 2.5 * 180 / 224)) < 3


@pytest.mark.parametrize("antialias", [False, True])
def test_checkerboard_vs_onnx_reference(antialias):
    """The reference's golden pin for the linear filters (test_resize.py:919-1029): the 22 x 22 checkerboard resized to
    17 x 13 must be within atol 1 of the ONNX reference implementation (regenerated by tests/onnx_resize_ref.py; the
    golden .npy files themselves live in DALI_extra).  Compared on the FLOAT output like the reference test, so the
    oracle's u8 rounding model plays no part."""
    from tests import onnx_resize_ref as R
    board = R.checkerboard_22_22()
    ref = R.interpolate_nd(board, R.linear_coeffs_antialias if antialias else (lambda x, _: R.linear_coeffs(x)), (17, 13))
    got = O.resample_f32(board, (17, 13), antialias=antialias)[:, :, 0]
    assert got.shape == (17, 13)
    np.testing.assert_allclose(got, ref, atol=1)
    # the filters are the same continuous kernels: the agreement is in fact at float accuracy
    assert np.abs(got - ref).max() < 2e-3
    # and the rounded u8 result is the rounded reference wherever that is not a tie
    u8 = O.resample_u8(board, (17, 13), antialias=antialias)[:, :, 0]
    clear = np.abs(ref - np.floor(ref) - 0.5) > 1e-2
    assert np.array_equal(u8[clear], np.floor(ref + 0.5).astype(np.uint8)[clear])


def _cubic_coeffs(ratio, scale=None, A=-0.5):
    """ONNX reference cubic_coeffs (onnx/reference/ops/op_resize.py) with the Catmull-Rom parameter DALI's window uses."""
    return np.array([((A * (ratio + 1) - 5 * A) * (ratio + 1) + 8 * A) * (ratio + 1) - 4 * A,
                     ((A + 2) * ratio - (A + 3)) * ratio * ratio + 1,
                     ((A + 2) * (1 - ratio) - (A + 3)) * (1 - ratio) * (1 - ratio) + 1,
                     ((A * ((1 - ratio) + 1) - 5 * A) * ((1 - ratio) + 1) + 8 * A) * ((1 - ratio) + 1) - 4 * A])


def test_cubic_upscale_matches_onnx_reference_cubic():
    """The tabulated cubic window (129 entries, linear interpolation) against the closed-form Catmull-Rom weights of the
    ONNX reference on a checkerboard and on noise: the table's interpolation error stays far below half an LSB."""
    from tests import onnx_resize_ref as R
    rng = np.random.default_rng(4)
    board = (np.indices((11, 13)).sum(0) % 2 * 255).astype(np.uint8)
    noise = rng.integers(0, 256, (9, 12), dtype=np.uint8)
    for img, osz in [(board, (29, 40)), (noise, (31, 25))]:
        got = O.resample_f32(img[:, :, None], osz, min_filter=O.FILTER_CUBIC, mag_filter=O.FILTER_CUBIC, antialias=False)[:, :, 0]
        ref = R.interpolate_nd(img, _cubic_coeffs, osz)
        # ONNX excludes nothing at the borders (edge padding) and so does the reference's clamping of tap indices
        assert np.abs(got - ref).max() < 0.25, np.abs(got - ref).max()


def test_tabulated_filters_known_answers():
    """resampling_filters.cu:66-137: table sizes, unit centre, symmetric, support = 2 * radius (cubic, Lanczos3) and the
    Gaussian's 4 sqrt(2) sigma; nearest neighbour = pixel replication on an integer up-scale."""
    for ftype, radius, n, sup in [(O.FILTER_CUBIC, 2, 129, 4), (O.FILTER_LANCZOS3, 3, 193, 6), (O.FILTER_CUBIC, 5, 129, 10),
                                  (O.FILTER_GAUSSIAN, 1, 65, 2), (O.FILTER_GAUSSIAN, 3.3, 65, 7)]:
        co, scale, anchor, support = O.filter_table(ftype, radius)
        assert len(co) == n and support == sup, (ftype, radius, len(co), support)
        assert co[n // 2] == 1.0 and np.allclose(co, co[::-1], atol=1e-6) and abs(anchor - (n - 1) / scale / 2) < 1e-4
    img = np.random.default_rng(0).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    up = O.resample_u8(img, (15, 14), min_filter=O.FILTER_NN, mag_filter=O.FILTER_NN)
    assert np.array_equal(up, img.repeat(3, 0).repeat(2, 1))
    dn = O.resample_u8(up, (5, 7), min_filter=O.FILTER_NN, mag_filter=O.FILTER_NN)
    assert np.array_equal(dn, img)
