"""Workloads for tools/hipemu/ldsprof.py in the shapes of the bench legs (not collected by the suites: the file name does
not match test_*.py; run as `python tools/hipemu/ldsprof.py -- tests/ldsprof_workloads.py [-k blur]`)."""
import io

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import synth_image

pytestmark = pytest.mark.gpu


def _dev(img):
    return torch.from_numpy(np.ascontiguousarray(img)).cuda()


def _rot(rng, size=512):
    t, s = np.deg2rad(rng.uniform(-30, 30)), rng.uniform(0.8, 1.2)
    c, sn = np.cos(t) / s, np.sin(t) / s
    m = np.array([[c, -sn, 0], [sn, c, 0]], np.float32)
    h = size / 2
    m[0, 2] = h - m[0, 0] * h - m[0, 1] * h
    m[1, 2] = h - m[1, 0] * h - m[1, 1] * h
    return m


def test_blur():      # configs[2]: 512 x 512 x 3, sigma 3
    from dali_amd import backend as B
    rng = np.random.default_rng(1)
    imgs = [synth_image(rng, 512, 512) for _ in range(3)]
    outs = B.gaussian_blur_batch([_dev(im) for im in imgs], sigma=3.0)
    assert np.array_equal(outs[0].cpu().numpy(), O.gaussian_blur_u8(imgs[0], O.gaussian_window(3.0)))


def test_warp():      # configs[2]: rotation by up to 30 degrees, scale 0.8 ... 1.2, bilinear, constant border
    from dali_amd import backend as B
    rng = np.random.default_rng(2)
    imgs = [synth_image(rng, 512, 512) for _ in range(3)]
    mats = [_rot(rng) for _ in imgs]
    outs = B.warp_affine_batch([_dev(im) for im in imgs], mats, interp=1, fill_value=0.0)
    assert np.array_equal(outs[0].cpu().numpy(), O.warp_affine_u8(imgs[0], mats[0], interp=1, fill=0.0))


def test_resample():  # configs[1]: random-resized-crop windows of ImageNet-like shapes -> 224 x 224
    from dali_amd import backend as B
    rng = np.random.default_rng(5)
    sizes = [(375, 500), (500, 375), (480, 640), (333, 500), (256, 384), (500, 500)]
    imgs = [synth_image(rng, *sizes[i % len(sizes)]) for i in range(12)]
    anchors, crops = O.rrc_batch(99, 0, [im.shape[:2] for im in imgs])
    rois = [(a[0], a[1], a[0] + c[0], a[1] + c[1]) for a, c in zip(anchors, crops)]
    out = B.resample_batch([_dev(im) for im in imgs], (224, 224), rois=rois).cpu().numpy()
    assert np.array_equal(out[0], O.resample_u8(imgs[0], (224, 224), roi=rois[0]))


def test_jpeg():      # configs[1]: the entropy decoder's passes and the colour kernel on 4:2:0 streams
    from PIL import Image
    from dali_amd import backend as B
    rng = np.random.default_rng(7)
    enc = []
    for h, w in [(375, 500), (500, 375), (480, 640), (333, 500)]:
        buf = io.BytesIO()
        Image.fromarray(synth_image(rng, h, w)).save(buf, format="JPEG", quality=90)
        enc.append(np.frombuffer(buf.getvalue(), np.uint8))
    outs = B.decode_jpeg_batch(enc, huffman="gpu")
    ref = O.decode_jpeg(enc[0]) if hasattr(O, "decode_jpeg") else None
    if ref is not None:
        assert np.array_equal(np.asarray(outs[0].cpu()), ref)
