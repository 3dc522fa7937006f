"""Host-side scan analysis feeding the GPU Huffman decoder (daliamdJpegAnalyzeScan): eligibility rules, table
extraction and the bounds of the entropy-coded segment.  CPU only."""
import ctypes as C
import glob
import os

import numpy as np

from dali_amd import _capi as capi
from tests.util import encode_jpeg, synth_image

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _analyze(e):
    host = capi.host()
    buf = np.frombuffer(e, np.uint8)
    info, scan = capi.JpegInfo(), capi.JpegScan()
    capi.check_host(host.daliamdJpegParse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info)))
    capi.check_host(host.daliamdJpegAnalyzeScan(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info),
                                                C.byref(scan)))
    return buf, info, scan


def _host_quant(buf, info):
    host = capi.host()
    coefs = [np.zeros(max(int(info.coef_elems[c]), 1), np.int16) for c in range(3)]
    ptrs = (C.c_void_p * 4)(*[c.ctypes.data for c in coefs], None)
    quant = np.zeros((3, 64), np.uint16)
    capi.check_host(host.daliamdJpegDecodeCoefficients(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size),
                                                       C.byref(info), ptrs, quant.ctypes.data_as(C.c_void_p)))
    return quant


def test_eligibility_rules():
    rng = np.random.default_rng(3)
    img = synth_image(rng, 64, 96)
    assert _analyze(encode_jpeg(img, 85))[2].eligible == 1
    assert _analyze(encode_jpeg(img, 85, optimize=True))[2].eligible == 1
    assert _analyze(encode_jpeg(img[..., 0], 85))[2].eligible == 1
    assert _analyze(encode_jpeg(img, 85, progressive=True))[2].eligible == 0
    rst = _analyze(encode_jpeg(img, 85, restart_marker_blocks=2))[2]
    assert rst.eligible == 1 and rst.restart_interval == 2


def test_scan_geometry_tables_and_segment_bounds():
    rng = np.random.default_rng(4)
    for sub, bpm, layout in [("4:4:4", 3, [0, 1, 2]), ("4:2:2", 4, [0, 0, 1, 2]), ("4:2:0", 6, [0, 0, 0, 0, 1, 2]),
                             ("4:1:1", 6, [0, 0, 0, 0, 1, 2])]:
        e = encode_jpeg(synth_image(rng, 50, 70), 80, subsampling=sub)
        buf, info, scan = _analyze(e)
        assert scan.eligible == 1 and scan.blocks_per_mcu == bpm
        assert list(scan.comp_of_block[:bpm]) == layout
        assert scan.mcus_x == -(-70 // (8 * info.hmax)) and scan.mcus_y == -(-50 // (8 * info.vmax))
        # the segment starts right after the SOS header and ends at the EOI marker
        o, l = scan.ecs_offset, scan.ecs_length
        assert bytes(buf[o + l:o + l + 2]) == b"\xff\xd9" and o + l + 2 == buf.size
        sos = e.index(b"\xff\xda")
        assert o == sos + 2 + int.from_bytes(e[sos + 2:sos + 4], "big")
        # quantisation tables: same values, same (column-major) order as the decoder reports
        q = np.ctypeslib.as_array(scan.quant)[:3]
        assert np.array_equal(q, _host_quant(buf, info))
        # DHT: every selected table is present and its code-length counts describe at most 256 symbols
        for c in range(3):
            assert 0 < sum(scan.dc_bits[scan.dc_sel[c]]) <= 16 and 0 < sum(scan.ac_bits[scan.ac_sel[c]]) <= 256


def test_golden_files_classification():
    seen = {}
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.jpg"))):
        e = open(path, "rb").read()
        seen[os.path.basename(path)] = _analyze(e)[2].eligible
    assert any(seen.values()) and not all(seen.values())
    for name, eligible in seen.items():
        if "prog" in name:
            assert eligible == 0, name
        if "rst" in name:     # restart intervals run on the GPU decoder since round 4
            assert eligible == 1, name


def test_header_analysis_agrees_with_the_walk():
    """daliamdJpegAnalyzeHeader (what decoders.image runs: one pass up to SOS) reports what daliamdJpegParse +
    daliamdJpegAnalyzeScan report, except that the segment is 'everything behind SOS'."""
    host = capi.host()
    rng = np.random.default_rng(8)
    streams = [open(f, "rb").read() for f in sorted(glob.glob(os.path.join(GOLDEN, "*.jpg")))]
    streams += [encode_jpeg(synth_image(rng, 40, 56), 85, restart_marker_rows=1),
                encode_jpeg(synth_image(rng, 40, 56), 85) + b"\x00" * 100]
    for e in streams:
        buf, info, scan = _analyze(e)
        info2, scan2 = capi.JpegInfo(), capi.JpegScan()
        capi.check_host(host.daliamdJpegAnalyzeHeader(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size), C.byref(info2),
                                                      C.byref(scan2)))
        assert bytes(info) == bytes(info2)
        if info.num_components == 4:
            assert scan2.eligible == 0
            continue
        assert scan.eligible == scan2.eligible
        if scan.eligible:
            assert scan2.length_is_upper_bound == 1 and scan.length_is_upper_bound == 0
            # everything behind SOS up to (and including) the file's last EOI marker: what trails it is not uploaded
            last_eoi = bytes(e).rfind(b"\xff\xd9")
            assert scan2.ecs_offset == scan.ecs_offset
            assert scan2.ecs_offset + scan2.ecs_length == (last_eoi + 2 if last_eoi >= scan.ecs_offset else buf.size)
            assert scan2.ecs_length >= scan.ecs_length
            scan2.ecs_length, scan2.length_is_upper_bound = scan.ecs_length, 0
            assert bytes(scan) == bytes(scan2)
