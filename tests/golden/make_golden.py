"""Generates the committed JPEG golden fixtures: small synthetic streams + the sha256 of the RGB
pixels libjpeg-turbo (Pillow's bundled build) decodes them to.  Run once in the build container:
    python tests/golden/make_golden.py
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.util import encode_jpeg, synth_image  # noqa: E402

CASES = {
    "g_420_64x48.jpg": ((48, 64), dict(subsampling="4:2:0", quality=75)),
    "g_444_33x47.jpg": ((33, 47), dict(subsampling="4:4:4", quality=90)),
    "g_422_31x17.jpg": ((31, 17), dict(subsampling="4:2:2", quality=85)),
    "g_411_40x56.jpg": ((40, 56), dict(subsampling="4:1:1", quality=60)),
    "g_420_prog_57x75.jpg": ((57, 75), dict(subsampling="4:2:0", quality=80, progressive=True)),
    "g_420_rst_50x70.jpg": ((50, 70), dict(subsampling="4:2:0", quality=75, restart_marker_blocks=3)),
    "g_gray_29x43.jpg": ((29, 43), dict(gray=True, quality=80)),
    "g_420_q5_64x64.jpg": ((64, 64), dict(subsampling="4:2:0", quality=5)),
    "g_420_q100_24x24.jpg": ((24, 24), dict(subsampling="4:2:0", quality=100)),
    "g_420_1x1.jpg": ((1, 1), dict(subsampling="4:2:0", quality=75)),
}


def main():
    index = {}
    for k, (name, (size, kw)) in enumerate(sorted(CASES.items())):
        rng = np.random.default_rng(1000 + k)
        kw = dict(kw)
        gray = kw.pop("gray", False)
        data = encode_jpeg(synth_image(rng, size[0], size[1], 1 if gray else 3), **kw)
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(data)
        rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        index[name] = {"shape": list(rgb.shape), "sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
                       "decoder": f"libjpeg-turbo {features.version('libjpeg_turbo')} (Pillow {Image.__version__})"}
    with open(os.path.join(HERE, "jpeg_golden.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
    print("wrote", len(index), "fixtures")


if __name__ == "__main__":
    main()
