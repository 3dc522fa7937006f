"""Shared helpers for the test-suite: seeded synthetic images / JPEG streams."""
import io

import numpy as np
from PIL import Image


def synth_image(rng, h, w, c=3):
    """Smooth multi-octave noise + gradients: compresses like a natural photo (8-10 %)."""
    acc = np.zeros((h, w, c), np.float32)
    for octave in range(4):
        gh, gw = max(2, h >> (5 - octave)), max(2, w >> (5 - octave))
        base = rng.integers(0, 256, (gh, gw, c)).astype(np.uint8)
        planes = [np.asarray(Image.fromarray(base[:, :, k]).resize((w, h), Image.BILINEAR), np.float32)
                  for k in range(c)]
        acc += np.stack(planes, -1) * (0.5 ** octave)
    acc /= 1.875
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(2):
        a, b = rng.uniform(-0.2, 0.2, 2)
        acc += (a * xx + b * yy)[:, :, None]
    acc += rng.normal(0, 2.0, acc.shape)
    img = np.clip(acc, 0, 255).astype(np.uint8)
    return img if c > 1 else img[:, :, 0]


def encode_jpeg(img, quality=85, subsampling="4:2:0", **kw):
    b = io.BytesIO()
    im = Image.fromarray(img)
    if im.mode == "L":
        im.save(b, "JPEG", quality=quality, **kw)
    else:
        im.save(b, "JPEG", quality=quality, subsampling=subsampling, **kw)
    return b.getvalue()


IMAGENET_LIKE_SIZES = [(375, 500), (500, 375), (480, 640), (500, 333), (500, 500), (384, 256), (768, 1024)]


def synth_jpeg_batch(rng, n, sizes=None, gray_frac=0.05):
    """n encoded streams drawn like SURVEY.md section 8(d): 80 % q75 / 20 % q90; 85 % 4:2:0, 10 % 4:4:4,
    5 % grayscale; sizes ImageNet-like."""
    sizes = sizes or IMAGENET_LIKE_SIZES
    out = []
    for _ in range(n):
        if sizes is IMAGENET_LIKE_SIZES:
            k = 6 if rng.random() < 0.05 else rng.integers(0, 6)
        else:
            k = rng.integers(0, len(sizes))
        h, w = sizes[k]
        q = 90 if rng.random() < 0.2 else 75
        r = rng.random()
        if r < gray_frac:
            out.append(encode_jpeg(synth_image(rng, h, w, 1), q))
        elif r < gray_frac + 0.10:
            out.append(encode_jpeg(synth_image(rng, h, w), q, "4:4:4"))
        else:
            out.append(encode_jpeg(synth_image(rng, h, w), q, "4:2:0"))
    return out
