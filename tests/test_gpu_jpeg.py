"""GPU parity: hybrid JPEG decode (host Huffman -> gfx950 IDCT / upsample / colour) vs the oracle.
Integer arithmetic => bit-exact."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image, synth_jpeg_batch

pytestmark = pytest.mark.gpu


def _decode_gpu(enc, **kw):
    from dali_amd import backend as B
    views, plan = B.decode_jpeg_batch(enc, device="cuda", **kw)
    torch.cuda.synchronize()
    return [v.cpu().numpy() for v in views]


def test_decode_matches_oracle_all_modes():
    rng = np.random.default_rng(7)
    enc = []
    for (h, w) in [(1, 1), (8, 8), (17, 23), (33, 47), (100, 75), (375, 500), (31, 17), (2, 3), (5, 64), (257, 255)]:
        for kw in [dict(subsampling="4:4:4"), dict(subsampling="4:2:2"), dict(subsampling="4:2:0"),
                   dict(subsampling="4:1:1"), dict(subsampling="4:2:0", progressive=True),
                   dict(subsampling="4:2:0", restart_marker_blocks=3), dict(subsampling="4:2:0", quality=100),
                   dict(subsampling="4:2:0", quality=5)]:
            enc.append(encode_jpeg(synth_image(rng, h, w), **({"quality": 85} | kw)))
        enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80))
        enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80, progressive=True))
    got = _decode_gpu(enc)
    for i, e in enumerate(enc):
        ref = O.jpeg_decode_rgb(e)
        assert got[i].shape == ref.shape
        assert np.array_equal(got[i], ref), f"sample {i}: max diff {np.abs(got[i].astype(int) - ref).max()}"


def test_decode_imagenet_like_batch_dense_and_padded_pitch():
    rng = np.random.default_rng(1234)
    enc = synth_jpeg_batch(rng, 48)
    refs = [O.jpeg_decode_rgb(e) for e in enc]
    for align in (1, 16, 64):
        got = _decode_gpu(enc, out_pitch_align=align)
        for i in range(len(enc)):
            assert np.array_equal(got[i], refs[i]), f"align {align} sample {i}"


def test_empty_batch():
    from dali_amd import backend as B
    views, plan = B.decode_jpeg_batch([], device="cuda")
    assert views == []


def test_corrupt_stream_raises():
    from dali_amd import backend as B
    from dali_amd._capi import DaliAmdError
    with pytest.raises(DaliAmdError):
        B.decode_jpeg_batch([b"not a jpeg at all"], device="cuda")
