"""Seeded synthetic inputs for tests and benchmarks: ImageNet-like images / JPEG streams (SURVEY.md section 8(d):
sizes from {500x375, 375x500, 640x480, 333x500, 500x500, 256x384, 1024x768 (5 %)}, multi-octave value noise +
linear gradients, quality 90 for 20 % / 75 for 80 %, 4:2:0 85 % / 4:4:4 10 % / grayscale 5 %, baseline Huffman).
Needs numpy + Pillow only (no torch, no GPU), so data sets can be generated in forked worker processes before the
device is initialised."""
import io

import numpy as np
from PIL import Image


def synth_image(rng, h, w, c=3, octaves=6, decay=0.85, noise=3.0, gradient=0.2):
    """1/f-like multi-octave value noise + linear gradients + sensor noise: compresses like a natural
    photograph (about 100 KB at ImageNet sizes with the q75/q90 mix of synth_jpeg_batch)."""
    acc = np.zeros((h, w, c), np.float32)
    for o in range(octaves):
        gh, gw = max(2, (h >> (octaves - 1 - o)) + 1), max(2, (w >> (octaves - 1 - o)) + 1)
        base = rng.integers(0, 256, (gh, gw, c)).astype(np.uint8)
        planes = [np.asarray(Image.fromarray(base[:, :, k]).resize((w, h), Image.BILINEAR), np.float32)
                  for k in range(c)]
        acc += (np.stack(planes, -1) - 128.0) * (decay ** o)
    acc = acc / np.sqrt((decay ** (2 * np.arange(octaves))).sum()) * 1.6 + 128.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(2):
        a, b = rng.uniform(-gradient, gradient, 2)
        acc += (a * xx + b * yy)[:, :, None]
    acc += rng.normal(0, noise, acc.shape)
    img = np.clip(acc, 0, 255).astype(np.uint8)
    return img if c > 1 else img[:, :, 0]


def encode_jpeg(img, quality=85, subsampling="4:2:0", **kw):
    b = io.BytesIO()
    im = Image.fromarray(img)
    if kw.get("optimize") or kw.get("progressive"):
        # Pillow hands the encoder ONE buffer for such a stream (it is written in one piece): make it large enough
        from PIL import ImageFile
        ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, int(img.size) + 65536)
    if im.mode == "L":
        im.save(b, "JPEG", quality=quality, **kw)
    else:
        im.save(b, "JPEG", quality=quality, subsampling=subsampling, **kw)
    return b.getvalue()


IMAGENET_LIKE_SIZES = [(375, 500), (500, 375), (480, 640), (500, 333), (500, 500), (384, 256), (768, 1024)]


def synth_jpeg_batch(rng, n, sizes=None, gray_frac=0.05, image_kw=None, **save_kw):
    """n encoded streams drawn like SURVEY.md section 8(d): 80 % q75 / 20 % q90; 85 % 4:2:0, 10 % 4:4:4,
    5 % grayscale; sizes ImageNet-like.  `save_kw`: passed to the encoder (optimize=True: Huffman tables optimised per
    image; progressive=True)."""
    sizes = sizes or IMAGENET_LIKE_SIZES
    image_kw = image_kw or {}
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
            out.append(encode_jpeg(synth_image(rng, h, w, 1, **image_kw), q, **save_kw))
        elif r < gray_frac + 0.10:
            out.append(encode_jpeg(synth_image(rng, h, w, **image_kw), q, "4:4:4", **save_kw))
        else:
            out.append(encode_jpeg(synth_image(rng, h, w, **image_kw), q, "4:2:0", **save_kw))
    return out


DATASET_VARIANTS = ("baseline", "distinct_dht", "mixed", "large", "flat")
LARGE_IMAGE_HW = (3000, 4000)     # 12 MP


def synth_dataset_image(index, seed=1234, variant="baseline"):
    """Image `index` of the synthetic data set: depends only on (seed, index, variant), so every rank of a sharded run
    generates exactly its own shard and all ranks agree on the whole set.  Variants (what real collections hold that the
    baseline set does not - VERDICT r05 missing 3; the reference takes all of them in one batch, image_decoder.h:613-880):
      baseline      the SURVEY 8(d) mix, encoder's default tables: ONE set of Huffman tables for the whole data set
      distinct_dht  the same images saved with optimize=True: every file brings its own four Huffman tables
      mixed         5 % progressive and 2 % CMYK (Adobe, four components) among the baseline streams (+ its 5 % grayscale)
      large         2 % 12-megapixel images (3000 x 4000) among the baseline sizes
      flat          2 % 12-megapixel frames that are mostly saturated (the position pass's worst case; tools only)"""
    rng = np.random.default_rng([seed, int(index)])
    if variant == "baseline":
        return synth_jpeg_batch(rng, 1)[0]
    pick = np.random.default_rng([seed, int(index), 77]).random()
    if variant == "distinct_dht":
        return synth_jpeg_batch(rng, 1, optimize=True)[0]
    if variant == "mixed":
        if pick < 0.05:
            return synth_jpeg_batch(rng, 1, gray_frac=0.0, progressive=True)[0]
        if pick < 0.07:
            h, w = IMAGENET_LIKE_SIZES[int(rng.integers(0, 6))]
            b = io.BytesIO()
            Image.fromarray(synth_image(rng, h, w)).convert("CMYK").save(b, "JPEG", quality=85)
            return b.getvalue()
        return synth_jpeg_batch(rng, 1)[0]
    if variant == "large":
        if pick < 0.02:
            # (the illumination gradients keep the brightness range of the small images: with their per-pixel slope a
            # 4000-pixel image saturates to black / white over most of its area - 0.5 bits per pixel, hundreds of empty
            # blocks per 256 bytes of stream, which no camera produces)
            return synth_jpeg_batch(rng, 1, sizes=[LARGE_IMAGE_HW], gray_frac=0.0,
                                    image_kw=dict(gradient=0.2 * 500 / max(LARGE_IMAGE_HW)))[0]
        return synth_jpeg_batch(rng, 1)[0]
    if variant == "flat":
        # 2 % 12-megapixel frames whose illumination gradients keep the per-pixel slope of the small images: most of the frame
        # saturates to black / white - 0.5 bits per pixel, hundreds of empty blocks per 256 bytes of stream.  Not what a camera
        # produces; what a scan, a screenshot or a product shot on a blank background looks like to the entropy decoder.
        if pick < 0.02:
            return synth_jpeg_batch(rng, 1, sizes=[LARGE_IMAGE_HW], gray_frac=0.0)[0]
        return synth_jpeg_batch(rng, 1)[0]
    raise ValueError(f"unknown data set variant {variant!r}")


def _synth_dataset_job(job):
    return synth_dataset_image(*job)


def synth_dataset(first_index, count, seed=1234, workers=0, variant="baseline"):
    """Encoded JPEGs [first_index, first_index + count) of the synthetic data set; `workers` > 1 forks that many
    generator processes (call before the GPU runtime is initialised in this process)."""
    jobs = [(i, seed, variant) for i in range(int(first_index), int(first_index) + int(count))]
    if workers and workers > 1 and count >= 2 * workers:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            # (chunks of a few images: the 12-megapixel ones of "large" take a hundred times as long as the others)
            return pool.map(_synth_dataset_job, jobs, chunksize=max(1, min(8, count // (4 * workers))))
    return [_synth_dataset_job(j) for j in jobs]
