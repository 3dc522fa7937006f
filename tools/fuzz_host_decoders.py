"""Mutation fuzzing of the host image parsers / decoders (untrusted bytes): valid JPEG (baseline 4:2:0 / 4:4:4 / grey,
progressive, restart markers, CMYK), PNG, BMP and PNM files are truncated, bit-flipped, spliced and fed to
decoders.image(device="cpu") through a CPU pipeline.  A decode either succeeds or raises; anything else (a crash, a
sanitizer report) is a bug.  Meant to run against the AddressSanitizer build:  tools/asan_fuzz.sh [iterations]"""
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
from PIL import Image


def seeds(rng):
    def img(h, w, mode="RGB"):
        a = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
        a = np.kron(a, np.ones((8, 8, 1), np.uint8))[:h, :w]
        im = Image.fromarray(a)
        return im.convert(mode) if mode != "RGB" else im
    out = []
    for kw in [dict(subsampling="4:2:0"), dict(subsampling="4:4:4"), dict(subsampling="4:2:2", progressive=True),
               dict(subsampling="4:2:0", restart_marker_blocks=3), dict(subsampling="4:2:0", optimize=True)]:
        b = io.BytesIO(); img(61, 83).save(b, "JPEG", quality=80, **kw); out.append(b.getvalue())
    b = io.BytesIO(); img(40, 50, "L").save(b, "JPEG", quality=70); out.append(b.getvalue())
    b = io.BytesIO(); img(40, 50, "CMYK").save(b, "JPEG", quality=70); out.append(b.getvalue())
    for fmt, mode in [("PNG", "RGB"), ("PNG", "L"), ("PNG", "P"), ("PNG", "RGBA"), ("BMP", "RGB"), ("BMP", "L"), ("PPM", "RGB"),
                      ("PPM", "L")]:
        b = io.BytesIO(); img(33, 47, mode).save(b, fmt); out.append(b.getvalue())
    b = io.BytesIO(); img(64, 64).save(b, "PNG", interlace=True) if False else img(64, 64).save(b, "PNG", compress_level=9)
    out.append(b.getvalue())
    return out


def mutate(rng, data, others):
    d = bytearray(data)
    kind = rng.integers(0, 6)
    if kind == 0 and len(d) > 4:                      # truncate
        del d[rng.integers(1, len(d)):]
    elif kind == 1:                                   # flip a few bits
        for _ in range(rng.integers(1, 8)):
            i = rng.integers(0, len(d)); d[i] ^= 1 << rng.integers(0, 8)
    elif kind == 2:                                   # overwrite a run with random bytes
        i = rng.integers(0, len(d)); n = min(len(d) - i, int(rng.integers(1, 64)))
        d[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 3:                                   # overwrite a run with 0xFF / 0x00 (marker and length fields)
        i = rng.integers(0, len(d)); n = min(len(d) - i, int(rng.integers(1, 8)))
        d[i:i + n] = bytes([0xFF if rng.integers(0, 2) else 0x00]) * n
    elif kind == 4:                                   # splice the tail of another file
        o = others[rng.integers(0, len(others))]
        i = rng.integers(0, len(d)); j = rng.integers(0, len(o))
        d[i:] = o[j:]
    else:                                             # duplicate / delete a chunk
        i = rng.integers(0, len(d)); n = min(len(d) - i, int(rng.integers(1, 256)))
        if rng.integers(0, 2): d[i:i] = d[i:i + n]
        else: del d[i:i + n]
    return bytes(d) if len(d) else b"\x00"


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(2024)
    pool = seeds(rng)
    bs = 1
    ok = bad = 0
    for out_type in (types.RGB, types.GRAY, types.YCbCr):
        def make():
            pipe = Pipeline(batch_size=bs, num_threads=2, device_id=None, prefetch_queue_depth=1)
            with pipe:
                enc = fn.external_source(name="enc")
                pipe.set_outputs(fn.decoders.image(enc, device="cpu", output_type=out_type),
                                 fn.decoders.image_random_crop(enc, device="cpu", output_type=out_type, seed=3))
            pipe.build()
            return pipe
        pipe = make()
        for it in range(iters):
            batch = [np.frombuffer(mutate(rng, pool[rng.integers(0, len(pool))], pool), np.uint8).copy() for _ in range(bs)]
            try:
                pipe.feed_input("enc", batch)
                pipe.run()
                ok += 1
            except RuntimeError:
                bad += 1
                pipe = make()   # like the reference's, a pipeline is not used again after a failed iteration
    print(f"fuzz: {ok} batches decoded, {bad} rejected with an error, no crash")


if __name__ == "__main__":
    main()
