"""The GPU entropy decoder's algorithm on the CPU: tools/huff_model.cpp runs the SAME per-lane code as the kernels
(dali_amd/csrc/huff_core.h) and restates their orchestration lane by lane - relaxation with capped block-start lists
and the overflow path, dense per-segment start lists, segment hand-over with repair, block ordinals, DC prefix sums,
the task mapping of the block pass - then compares every coefficient with the host entropy decoder.  Shrunk
constants (tiny slices / segments / lists) force the paths real streams seldom take."""
import ctypes as C
import glob
import io
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from tests.util import encode_jpeg, synth_image, synth_jpeg_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = None                          # the constants of jpeg_huffman.hip
STRESS = [(64, 16, 2, 3, 64),          # slice_bytes, seg_threads, warm_lanes, list_cap, blocks_per_wg
          (32, 8, 1, 1, 64), (256, 32, 0, 33, 128), (128, 64, 12, 2, 768)]


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("huff_model") / "libhuff_model.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "dali_amd", "csrc"), os.path.join(ROOT, "tools", "huff_model.cpp"),
                           "-L", os.path.join(ROOT, "dali_amd", "lib"), "-ldali_amd_host",
                           "-Wl,-rpath," + os.path.join(ROOT, "dali_amd", "lib"), "-o", out])
    lib = C.CDLL(out)
    lib.huff_model_message.restype = C.c_char_p

    def check(data, params=None):
        buf = np.frombuffer(data, np.uint8)
        stats = (C.c_int * 7)()
        pr = (C.c_int * 5)(*params) if params else None
        rc = lib.huff_model_check(buf.ctypes.data_as(C.c_void_p), C.c_size_t(len(data)), pr, stats)
        assert rc != 1, lib.huff_model_message().decode()
        return rc, dict(zip(("segments", "rounds", "repairs", "overflow_lanes", "blocks", "starts", "restarts"), stats))
    return check


def _streams():
    rng = np.random.default_rng(77)
    out = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.jpg")))]
    for size in [(33, 47), (100, 75), (8, 8), (1, 1), (240, 320)]:
        for sub in ("4:2:0", "4:4:4", "4:2:2"):
            out.append(encode_jpeg(synth_image(rng, *size), int(rng.integers(30, 96)), sub))
        out.append(encode_jpeg(synth_image(rng, *size, 1), 70))
    out += synth_jpeg_batch(rng, 4)
    # content a wrongly started decoder does not recover from quickly: flat, periodic, very sparse blocks
    flat = np.full((256, 384, 3), 200, np.uint8)
    out.append(encode_jpeg(flat, 75))
    stripes = np.zeros((192, 256, 3), np.uint8)
    stripes[:, ::16] = 255
    out.append(encode_jpeg(stripes, 60))
    grad = np.tile(np.linspace(0, 255, 512).astype(np.uint8)[None, :, None], (320, 1, 3))
    out.append(encode_jpeg(grad, 90, "4:2:0"))
    b = io.BytesIO()
    Image.fromarray(synth_image(rng, 120, 160)).save(b, "JPEG", quality=85, optimize=True)   # per-image Huffman tables
    out.append(b.getvalue())
    return out


def _restart_streams():
    """Restart intervals (DRI + RSTn markers) of every size: one MCU, part of a row, rows, larger than a segment;
    gray streams whose flat blocks are shorter than a byte (an MCU fits into what could be padding); optimised
    tables; streams with bytes behind EOI (the decoder finds the end of the scan by itself)."""
    rng = np.random.default_rng(78)
    out = []
    for size, sub, blocks in [((100, 150), "4:2:0", 1), ((100, 150), "4:2:0", 3), ((64, 96), "4:4:4", 2), ((240, 320), "4:2:0", 7),
                              ((240, 320), "4:2:2", 20), ((480, 640), "4:2:0", 200), ((33, 47), "4:4:4", 1)]:
        out.append(encode_jpeg(synth_image(rng, *size), int(rng.integers(40, 96)), sub, restart_marker_blocks=blocks))
    for rows in (1, 2, 5):
        out.append(encode_jpeg(synth_image(rng, 200, 300), 80, "4:2:0", restart_marker_rows=rows))
    out.append(encode_jpeg(synth_image(rng, 120, 200, 1), 75, restart_marker_blocks=1))
    out.append(encode_jpeg(synth_image(rng, 120, 200, 1), 75, restart_marker_blocks=5, optimize=True))
    flat = np.full((64, 512), 128, np.uint8)              # 6-bit MCUs (DC difference 0 + end-of-block)
    flat[:, 300:] = rng.integers(0, 255, (64, 212))
    for blocks in (1, 3, 8):
        out.append(encode_jpeg(flat, 75, restart_marker_blocks=blocks))
    out.append(encode_jpeg(np.full((256, 384, 3), 77, np.uint8), 75, restart_marker_blocks=2))
    out.append(encode_jpeg(synth_image(rng, 96, 128), 85, restart_marker_blocks=4) + b"trailing bytes \xff\xd8 behind EOI" * 40)
    return out


@pytest.mark.parametrize("params", [KERNEL] + STRESS)
def test_model_decodes_restart_intervals(model, params):
    seen = dict(restarts=0, multi_segment=0)
    for data in _restart_streams():
        rc, st = model(data, params)
        assert rc == 0, "streams with restart intervals are eligible for the GPU decoder"
        assert st["restarts"] > 0
        seen["restarts"] += st["restarts"]
        seen["multi_segment"] += st["segments"] > 1
    assert seen["multi_segment"] > 0
    print(params, seen)


def test_model_takes_one_bit_codes_and_finds_the_end_of_the_scan(model):
    # optimised tables of a flat image: the DC table holds ONE symbol, coded with one bit
    flat = np.full((128, 192, 3), 90, np.uint8)
    data = encode_jpeg(flat, 75, optimize=True)
    for params in [KERNEL] + STRESS:
        assert model(data, params)[0] == 0
        assert model(data + bytes(1000), params)[0] == 0        # padding behind EOI
        assert model(data[:-2], params)[0] == 0                 # no EOI at all: the segment runs to the end of the file


@pytest.mark.parametrize("params", [KERNEL] + STRESS)
def test_model_matches_the_host_decoder(model, params):
    seen = dict(eligible=0, repairs=0, overflow=0, multi_segment=0, rounds=0)
    for data in _streams():
        rc, st = model(data, params)
        if rc == 2:
            continue   # progressive / multi-scan: host path
        seen["eligible"] += 1
        assert st["starts"] >= st["blocks"] + 1
        seen["repairs"] += st["repairs"]
        seen["overflow"] += st["overflow_lanes"]
        seen["multi_segment"] += st["segments"] > 1
        seen["rounds"] = max(seen["rounds"], st["rounds"])
    assert seen["eligible"] >= 30
    if params is not KERNEL:   # the stress constants must really reach the rare paths
        assert seen["multi_segment"] >= 10 and seen["overflow"] > 0
        if params[2] <= 2:
            assert seen["repairs"] > 0
    print(params, seen)
