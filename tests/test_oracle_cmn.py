"""Oracle pinning for CropMirrorNormalize: the numpy formula of the reference's python test
(dali/test/python/operator_1/test_crop_mirror_normalize.py:225-286, eps 1e-5 for fp32; fp16 mean 0.3 / max 0.6
at :324 -- far looser than what we hold) and the sequential-data naive loop of
dali/kernels/slice/slice_flip_normalize_permute_pad_kernel_test.h:40-131."""
import numpy as np
import pytest

from oracle import oracle as O


def _numpy_cmn(img, anchor, crop, mirror, mean, std, scale=1.0, shift=0.0, layout="CHW", pad=False):
    (ay, ax), (ch, cw) = anchor, crop
    out = img[ay:ay + ch, ax:ax + cw].astype(np.float32)
    if mirror:
        out = out[:, ::-1]
    C = img.shape[2]
    inv_std = np.array([np.float32(1.0) / np.float32(s) for s in std], np.float32)
    mean = np.float32(mean)
    outc = 4 if pad and C == 3 else C
    o2 = np.zeros((ch, cw, outc), np.float32)
    o2[:, :, :C] = (out - mean) * inv_std * scale + shift
    return o2.transpose(2, 0, 1) if layout == "CHW" else o2


@pytest.mark.parametrize("layout", ["CHW", "HWC"])
@pytest.mark.parametrize("mirror", [False, True])
def test_cmn_fp32_matches_numpy_formula(layout, mirror):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    mean, std = [0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255]
    for scale, shift in [(1.0, 0.0), (2.0, 0.5), (0.25, -3.0)]:
        m, i = O.cmn_norm_args(mean, std, scale, shift)
        got = O.cmn_u8(img, (5, 9), (40, 50), mirror=mirror, mean=m, inv_std=i, layout=layout, dtype=O.F32)
        ref = _numpy_cmn(img, (5, 9), (40, 50), mirror, mean, std, scale, shift, layout)
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_cmn_sequential_data_naive_loop():
    # slice_flip_normalize_permute_pad_kernel_test.h: sequential input, per-channel mean/std, flip, permute
    shape = (7, 9, 3)
    img = (np.arange(np.prod(shape)) % 256).astype(np.uint8).reshape(shape)
    mean = np.array([3.0, 10.0, 100.0], np.float32)
    inv = np.array([1 / 2.0, 1 / 4.0, 1 / 8.0], np.float32)
    got = O.cmn_u8(img, (1, 2), (5, 6), mirror=True, mean=mean, inv_std=inv, layout="CHW", dtype=O.F32)
    for c in range(3):
        for y in range(5):
            for x in range(6):
                v = (np.float32(img[1 + y, 2 + (5 - x), c]) - mean[c]) * inv[c]
                assert got[c, y, x] == v


def test_cmn_identity_args_are_dropped_and_std1_exact():
    m, i = O.cmn_norm_args([0.0], [1.0])
    assert m.size == 0 and i.size == 0          # crop_mirror_normalize.h:142-148: skip normalisation
    m, i = O.cmn_norm_args([0.0, 0.0, 0.0], [1.0])
    assert m.size == 0
    m, i = O.cmn_norm_args([128.0], [1.0])
    assert list(m) == [128.0] and list(i) == [1.0]
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1)
    got = O.cmn_u8(img, (0, 0), (16, 16), mean=m, inv_std=i, layout="HWC", dtype=O.F16)
    assert np.array_equal(got.astype(np.float32).reshape(-1), np.arange(256, dtype=np.float32) - 128)


def test_fp16_rounding_is_ties_away_from_zero():
    """half_float round_to_nearest with HALF_ROUND_TIES_TO_EVEN == 0 (include/dali/util/half.hpp:231-243):
    agrees with IEEE round-to-nearest-even everywhere except exact ties, which go away from zero."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 3, 200000).astype(np.float32),
                        rng.uniform(-70000, 70000, 20000).astype(np.float32),
                        (rng.normal(0, 1, 20000) * 1e-6).astype(np.float32),
                        np.array([0.0, -0.0, 65504.0, 65520.0, 1e9, -1e9, np.inf, -np.inf, 6e-8, 2.98e-8, 5.96e-8],
                                 np.float32)])
    got = O.float2half(x).view(np.uint16)
    rne = x.astype(np.float16).view(np.uint16)
    diff = got != rne
    # every difference must be an exact tie resolved away from zero (got = rne + 1 ulp in magnitude)
    if diff.any():
        lo = np.minimum(got[diff], rne[diff]).view(np.float16).astype(np.float64)
        hi = np.maximum(got[diff], rne[diff]).view(np.float16).astype(np.float64)
        mid = (lo + hi) / 2
        assert np.array_equal(mid, x[diff].astype(np.float64))
        assert (np.abs(got[diff].view(np.float16).astype(np.float64)) >= np.abs(x[diff].astype(np.float64))).all()
    # explicit ties: 1 + 2^-11 is halfway between 1 and 1 + 2^-10 -> away (1 + 2^-10); RNE gives 1
    t = np.array([1.0 + 2.0 ** -11, -(1.0 + 2.0 ** -11), 1.0 + 3 * 2.0 ** -11], np.float32)
    h = O.float2half(t).astype(np.float64)
    assert list(h) == [1 + 2.0 ** -10, -(1 + 2.0 ** -10), 1 + 2 * 2.0 ** -10]
    assert np.isnan(O.float2half(np.array([np.nan], np.float32)).astype(np.float32)[0])


def test_pad_output_and_out_of_bounds():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (10, 12, 3), dtype=np.uint8)
    m, i = O.cmn_norm_args([1, 2, 3], [2, 2, 2])
    out = O.cmn_u8(img, (0, 0), (10, 12), mean=np.r_[m, 0], inv_std=np.r_[i, 0], pad_output=True, layout="HWC",
                   dtype=O.F32)
    assert out.shape == (10, 12, 4) and (out[:, :, 3] == 0).all()
    with pytest.raises(RuntimeError):
        O.cmn_u8(img, (5, 5), (10, 12), dtype=O.F32)           # out_of_bounds_policy="error"
    out = O.cmn_u8(img, (-2, -3), (14, 18), mean=m, inv_std=i, pad_oob=True, fill_values=(9.0,), dtype=O.F32)
    assert (out[:, :2, :] == 9).all() and (out[:, :, :3] == 9).all() and (out[:, 12:, :] == 9).all()
    assert out[0, 2, 3] == (np.float32(img[0, 0, 0]) - m[0]) * i[0]
