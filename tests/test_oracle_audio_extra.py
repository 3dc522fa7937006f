"""Pins for the oracle's MFCC and audio resampling (CPU): the DCT against scipy's (an independent implementation of
the same textbook definitions), the resampler against analytic signals."""
import numpy as np
import pytest
from scipy import fft as sfft

from oracle import audio as A


@pytest.mark.parametrize("dct_type", [1, 2, 3, 4])
@pytest.mark.parametrize("norm", [False, True])
def test_dct_matches_scipy(dct_type, norm):
    if dct_type == 1 and norm:
        pytest.skip("the reference ignores normalisation for type I")
    rng = np.random.default_rng(dct_type)
    x = rng.normal(0, 1, (40, 17)).astype(np.float32)
    got = A.mfcc(x, n_mfcc=13, dct_type=dct_type, normalize=norm, lifter=0.0)
    ref = sfft.dct(x.astype(np.float64), type=dct_type, axis=0, norm="ortho" if norm else None)[:13]
    if not norm:
        ref = ref / 2      # scipy's unnormalised transforms are twice the textbook sums (test_mfcc.py:78-81)
    assert got.shape == (13, 17)
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()


def test_lifter_formula_and_ndct_clamp():
    x = np.ones((8, 3), np.float32)
    out = A.mfcc(x, n_mfcc=20, dct_type=2, lifter=22.0)        # more coefficients than bands: clamped to 8
    assert out.shape == (8, 3)
    k = np.arange(8)
    want = (1 + 11.0 * np.sin(np.pi * (k + 1) / 22.0))[:, None] * A.mfcc(x, 20, 2, False, 0.0)
    assert np.allclose(out, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("in_rate,out_rate", [(44100, 16000), (16000, 44100), (8000, 8000), (22050, 16000)])
def test_resample_sine_keeps_frequency_and_amplitude(in_rate, out_rate):
    n = 4000
    f = 440.0
    t = np.arange(n) / in_rate
    x = np.sin(2 * np.pi * f * t).astype(np.float32)
    y = A.audio_resample(x, in_rate, out_rate, quality=50.0)
    assert y.shape[0] == int(np.ceil(n * out_rate / in_rate))
    tt = np.arange(y.shape[0]) / out_rate
    ref = np.sin(2 * np.pi * f * tt)
    inner = slice(64, y.shape[0] - 64)                          # the ends see a truncated filter
    assert np.abs(y[inner] - ref[inner]).max() < 2e-3
    # window properties of the reference: lobes from the quality, unit gain at the centre
    assert [A.resample_lobes(q) for q in (0, 50, 100)] == [3, 16, 64]
    lookup, scale, center = A.resample_window(16)
    assert lookup.size == 16 * 64 + 1 + 5 and lookup[0] == 0 and abs(lookup[int(center)] - 1) < 1e-6


def test_resample_channels_and_out_length():
    rng = np.random.default_rng(2)
    x = rng.normal(0, 0.3, (1000, 2)).astype(np.float32)
    y = A.audio_resample(x, 1000, 700, out_length=700)
    assert y.shape == (700, 2)
    for c in range(2):
        assert np.array_equal(y[:, c], A.audio_resample(x[:, c], 1000, 700, out_length=700))


def test_typed_resampling_oracle_reproduces_the_reference_known_answers():
    """oracle.audio.audio_resample_typed against the reference's conversion table
    (dali/test/python/operator_1/test_audio_resample.py:119-161): constant signals, scale 1, quality 0."""
    from tests.test_audio_resample_types import _conversion_cases, DYNAMIC_RANGES, _NP
    cases = _conversion_cases() + [(t, v, t, v, e) for t, v, e in DYNAMIC_RANGES]
    for src, in_values, dst, out_values, eps in cases:
        for x, want in zip(in_values, out_values):
            sig = np.full(64, x, _NP[src])
            got = A.audio_resample_typed(sig, 1.0, 1.0, quality=0.0, out_dtype=_NP[dst])
            ref = np.full(64, want, _NP[dst])
            assert got.dtype == _NP[dst]
            assert np.allclose(got.astype(np.float64), ref.astype(np.float64), 1e-6, eps), (src, dst, x, got[:3], want)
