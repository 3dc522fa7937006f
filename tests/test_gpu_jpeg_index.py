"""Side information of resident streams (daliamdJpegHuffDesc.index / index_out, round 5): a decode that builds the index
of its streams, then decodes FROM the index - no un-stuffing, one decode per 256-byte slice from its recorded entry state,
DC predictors from the index, slices outside a region of interest skipped - must give the same bits as the full parse and
as the oracle.  Kernel level (the C ABI through dali_amd/backend.py); the operator level is tests/test_gpu_encoded_cache.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image

pytestmark = pytest.mark.gpu


def _streams(rng):
    enc = []
    for (h, w) in [(1, 1), (8, 8), (17, 23), (100, 75), (375, 500), (500, 375), (257, 255), (5, 640), (480, 640)]:
        for kw in [dict(subsampling="4:4:4"), dict(subsampling="4:2:2"), dict(subsampling="4:2:0"), dict(subsampling="4:1:1"),
                   dict(subsampling="4:2:0", quality=100), dict(subsampling="4:2:0", quality=5),
                   dict(subsampling="4:2:0", optimize=True)]:
            enc.append(encode_jpeg(synth_image(rng, h, w), **({"quality": 85} | kw)))
        enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80))
    # flat content (hundreds of blocks per slice), noise at q100 (a block spans slices), trailing bytes behind EOI, a
    # restart-interval stream (takes the ordinary path inside the same table) and a progressive one (host decoder)
    enc.append(encode_jpeg(np.full((480, 640, 3), (200, 30, 90), np.uint8), 90))
    enc.append(encode_jpeg(rng.integers(0, 256, (130, 262, 3), dtype=np.uint8), 100, subsampling="4:4:4"))
    enc.append(encode_jpeg(synth_image(rng, 120, 160), 85) + b"\x00\xff\xd8tail" * 9)
    enc.append(encode_jpeg(synth_image(rng, 200, 300), 85, subsampling="4:2:0", restart_marker_blocks=7))
    enc.append(encode_jpeg(synth_image(rng, 64, 80), 85, progressive=True))
    enc.append(encode_jpeg(synth_image(rng, 1200, 1600), 92, subsampling="4:2:0"))   # several segments
    return enc


@pytest.mark.parametrize("exact_scan", [True, False])
def test_indexed_decode_equals_full_parse_and_oracle(exact_scan):
    from dali_amd import backend as B
    enc = _streams(np.random.default_rng(21))
    first, plan = B.decode_jpeg_batch(enc, device="cuda", exact_scan=exact_scan, index="build")
    torch.cuda.synchronize()
    first = [v.cpu().numpy() for v in first]
    again, plan2 = B.decode_jpeg_batch(enc, device="cuda", exact_scan=exact_scan, index="use", index_from=plan)
    torch.cuda.synchronize()
    again = [v.cpu().numpy() for v in again]
    for i, e in enumerate(enc):
        ref = O.jpeg_decode_rgb(e)
        assert np.array_equal(first[i], ref), f"sample {i} (decode that builds the index)"
        assert np.array_equal(again[i], ref), f"sample {i} (decode from the index): {np.argwhere((again[i] != ref).any(2))[:4].tolist()}"
    # 12 bytes per 256-byte slice + the header: a few per cent of the stream
    assert plan.index_bytes < 1.10 * plan.stream_bytes + 1024 * len(enc)


def test_indexed_window_decode(tmp_path):
    """Windows (region-of-interest decode) from the index: only the slices that hold blocks of the window's MCU rectangle
    are decoded; every window - other ones than the decode that built the index saw - equals the crop of the oracle."""
    from dali_amd import backend as B
    rng = np.random.default_rng(22)
    enc = [e for e in _streams(rng)]
    shapes = [O.jpeg_decode_rgb(e).shape[:2] for e in enc]

    def windows(seed):
        r = np.random.default_rng(seed)
        out = []
        for (h, w) in shapes:
            wh, ww = int(r.integers(1, h + 1)), int(r.integers(1, w + 1))
            out.append((int(r.integers(0, h - wh + 1)), int(r.integers(0, w - ww + 1)), wh, ww))
        return out
    _, plan = B.decode_jpeg_batch(enc, device="cuda", exact_scan=False, index="build", rois=windows(1))
    torch.cuda.synchronize()
    for seed in (2, 3):
        rois = windows(seed)
        got, _ = B.decode_jpeg_batch(enc, device="cuda", exact_scan=False, index="use", index_from=plan, rois=rois)
        torch.cuda.synchronize()
        for i, (e, (y, x, h, w)) in enumerate(zip(enc, rois)):
            ref = O.jpeg_decode_rgb(e)[y:y + h, x:x + w]
            assert np.array_equal(got[i].cpu().numpy(), ref), f"sample {i} window {(y, x, h, w)}"


def test_setup_refuses_what_has_no_index():
    """The plain Setup entry point refuses tables with index pointers; restart-interval streams take neither."""
    import ctypes as C
    from dali_amd import _capi as capi, backend as B
    rng = np.random.default_rng(23)
    enc = [encode_jpeg(synth_image(rng, 64, 64), 85, restart_marker_blocks=3)]
    plan = B.JpegBatchPlan(enc)
    plan.set_index_mode("build", torch.device("cuda"))
    planes = torch.empty(plan.plane_bytes, dtype=torch.uint8, device="cuda")
    table = plan.huffman_descs(None, planes_dev=planes)[0]
    assert table["index_out"][0] == 0                       # (the driver leaves restart-interval streams alone)
    table = table.copy()
    table["index_out"] = plan._index_dev.data_ptr()
    a, b, c, k = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    rc = capi.kernels().daliamdJpegHuffmanSetupColor(table.ctypes.data_as(C.c_void_p), 1, C.byref(a), C.byref(b), C.byref(c), C.byref(k))
    assert rc != 0 and b"restart" in capi.kernels().daliamdGetLastErrorMessage()


def test_host_built_code_tables_are_the_device_built_ones():
    """daliamdJpegHuffDesc.tables: the code tables of a stream built on the host (daliamdJpegHuffmanTablesBuild) are byte for
    byte what PrepareKernel's table workgroups leave in the scratch, and a decode that brings them - the launch then builds
    nothing - gives the same pixels; with an index on top the launch starts at IndexedSyncKernel."""
    import ctypes as C
    import bench
    from dali_amd import _capi as capi, backend as B
    rng = np.random.default_rng(24)
    enc = [encode_jpeg(synth_image(rng, 200, 300), 85), encode_jpeg(synth_image(rng, 120, 160), 90, optimize=True),
           encode_jpeg(synth_image(rng, 64, 64, 1), 70), encode_jpeg(synth_image(rng, 96, 96), 60, subsampling="4:4:4", optimize=True),
           encode_jpeg(synth_image(rng, 200, 300), 85, subsampling="4:2:0", restart_marker_blocks=5)]
    plain, plan = B.decode_jpeg_batch(enc, device="cuda")
    torch.cuda.synchronize()
    lib = capi.kernels()
    nb = C.c_size_t(0)
    capi.check(lib.daliamdJpegHuffmanTablesBytes(C.byref(nb)))
    table = plan.huffman_descs(None, planes_dev=torch.empty(max(plan.plane_bytes, 1), dtype=torch.uint8, device="cuda"))[0]
    scratch = plan._huff_ws["scratch"].cpu().numpy()
    for j in range(len(enc)):
        if table["table_owner"][j] != j:
            continue        # (built in its owner's scratch)
        host = np.zeros(nb.value, np.uint8)
        capi.check(lib.daliamdJpegHuffmanTablesBuild(C.c_void_p(table.ctypes.data + j * table.dtype.itemsize), host.ctypes.data_as(C.c_void_p)))
        ecs_len, head = int(table["ecs_len"][j]), int(table["ecs"][j]) % 16
        off = int(plan._scratch_off[j]) + 16 + 16 * int(table["num_tiles"][j]) + (ecs_len + 256 + 15) // 16 * 16
        assert int(table["num_tiles"][j]) == max(1, -(-(head + ecs_len) // 8192))
        dev = scratch[off:off + nb.value]
        assert np.array_equal(dev, host), f"stream {j}: {np.nonzero(dev != host)[0][:8].tolist()}"
    bench.kernel_timing(True)
    brought, _ = B.decode_jpeg_batch(enc, device="cuda", host_tables=True)
    torch.cuda.synchronize()
    _, p1 = B.decode_jpeg_batch(enc[:4], device="cuda", host_tables=True, index="build")
    torch.cuda.synchronize()
    bench.kernel_timing()
    again, _ = B.decode_jpeg_batch(enc[:4], device="cuda", host_tables=True, index="use", index_from=p1)
    torch.cuda.synchronize()
    bench.kernel_timing(False)
    launched = bench.kernel_timing()
    assert "IndexedSyncKernel" in launched and "PrepareKernel" not in launched and "SyncKernel" not in launched, launched
    for i, e in enumerate(enc):
        ref = O.jpeg_decode_rgb(e)
        assert np.array_equal(plain[i].cpu().numpy(), ref) and np.array_equal(brought[i].cpu().numpy(), ref), i
        if i < 4:
            assert np.array_equal(again[i].cpu().numpy(), ref), i


@pytest.mark.parametrize("exact_scan", [True, False])
def test_host_built_index_is_the_device_built_one(exact_scan):
    """daliamdJpegHuffmanIndexBuildHost (round 6: the sidecar tools/jpeg2idx.py writes, so that a cold process and epoch 1 from
    files decode from the index too): byte for byte what IndexBuildKernel leaves behind a decode - header, clean stream with
    its all-ones padding, every slice entry and the sentinel - and a decode FROM the host-built entries gives the oracle's
    pixels.  (The device leaves the bytes between those regions unwritten; the host zeroes them.)"""
    from dali_amd import backend as B
    enc = _streams(np.random.default_rng(21))
    _, plan = B.decode_jpeg_batch(enc, device="cuda", exact_scan=exact_scan, index="build")
    torch.cuda.synchronize()
    dev = plan._index_dev.cpu().numpy()
    host_all = np.zeros_like(dev)
    checked = 0
    for j, i in enumerate(plan._huff_sel):
        off, ecs_len = int(plan._index_off[j]), int(plan._ecs_len[j])
        if plan.scan["restart_interval"][i] != 0:
            continue                                                     # (no index for restart-interval streams)
        host, status = plan.host_index(j)
        assert status == 0, (i, status)
        d = dev[off:off + host.size]
        clean_len, total_starts, num_slices = (int(x) for x in d[:12].view(np.int32))
        assert bytes(host[:64]) == bytes(d[:64]), f"stream {i}: header {host[:12].view(np.int32)} vs {d[:12].view(np.int32)}"
        stream_end = clean_len + 40                                      # the stream and its 40 bytes of all-ones padding
        assert np.array_equal(host[64:64 + stream_end], d[64:64 + stream_end]), f"stream {i}: clean stream"
        eoff = 64 + (ecs_len + 256 + 63) // 64 * 64
        cap = (ecs_len + 255) // 256
        he, de = host[eoff:eoff + 12 * (cap + 1)].view(np.uint32).reshape(-1, 3), d[eoff:eoff + 12 * (cap + 1)].view(np.uint32).reshape(-1, 3)
        bad = np.nonzero((he != de).any(1))[0]
        assert bad.size == 0, f"stream {i} ({num_slices} slices, {total_starts} starts): entries {bad[:6].tolist()} differ: {he[bad[:3]].tolist()} vs {de[bad[:3]].tolist()}"
        host_all[off:off + host.size] = host
        checked += 1
    assert checked >= len(enc) - 3
    # decode from the HOST-built entries
    plan._index_dev.copy_(torch.from_numpy(host_all))
    again, _ = B.decode_jpeg_batch(enc, device="cuda", exact_scan=exact_scan, index="use", index_from=plan)
    torch.cuda.synchronize()
    for i, e in enumerate(enc):
        assert np.array_equal(again[i].cpu().numpy(), O.jpeg_decode_rgb(e)), f"sample {i} (decode from the host-built index)"
