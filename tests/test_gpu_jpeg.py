"""GPU parity: JPEG decode (GPU or host Huffman -> gfx950 IDCT / upsample / colour) vs the oracle.
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


@pytest.mark.parametrize("huffman", ["gpu", "gpu-header-only", "host"])
def test_decode_matches_oracle_all_modes(huffman):
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
        enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80, restart_marker_blocks=2))
        enc.append(encode_jpeg(synth_image(rng, h, w), 70, subsampling="4:4:4", restart_marker_rows=1) + b"\xff\xd8tail" * 9)
    # "gpu-header-only": what decoders.image does - the host parses the headers only, the segment handed to the kernels is
    # everything behind SOS (EOI marker, trailing bytes) and the un-stuffing pass finds where it ends
    kw = dict(huffman="gpu", exact_scan=False) if huffman == "gpu-header-only" else dict(huffman=huffman)
    got = _decode_gpu(enc, **kw)
    for i, e in enumerate(enc):
        ref = O.jpeg_decode_rgb(e)
        assert got[i].shape == ref.shape
        assert np.array_equal(got[i], ref), f"sample {i}: max diff {np.abs(got[i].astype(int) - ref).max()}"


def test_fused_colour_output_equals_the_colour_kernel_and_the_oracle():
    """Round 4: YCbCr 4:2:0 streams leave the entropy decoder as RGB (chroma of a band of MCU rows in LDS, fancy
    upsampling + colour conversion in the luma lanes, the band seams finished by a second small launch).  Sizes that move
    the seams around (1 .. 24 MCU rows per band, widths up to the 2048-pixel limit and beyond it - those fall back), odd
    sizes, a restart-interval stream, a flat image; the result must equal the planes + colour-kernel path and the oracle."""
    from dali_amd import backend as B
    rng = np.random.default_rng(5)
    enc, want_fused = [], []
    for (h, w) in [(16, 16), (17, 33), (33, 17), (5, 5), (1, 6), (31, 2048), (47, 2047), (64, 2049), (48, 1025), (49, 1024),
                   (375, 500), (500, 375), (333, 517), (129, 640), (128, 641), (640, 129), (1080, 1920), (97, 131), (160, 8),
                   (240, 320), (1200, 40)]:
        enc.append(encode_jpeg(synth_image(rng, h, w), 85, subsampling="4:2:0"))
        want_fused.append(w <= 2048)
    enc.append(encode_jpeg(synth_image(rng, 375, 500), 90, subsampling="4:2:0", restart_marker_blocks=7)); want_fused.append(True)
    enc.append(encode_jpeg(np.full((480, 640, 3), (200, 30, 90), np.uint8), 90)); want_fused.append(True)
    enc.append(encode_jpeg(rng.integers(0, 256, (130, 262, 3), dtype=np.uint8), 100, subsampling="4:2:0")); want_fused.append(True)
    enc.append(encode_jpeg(synth_image(rng, 120, 160), 85, subsampling="4:2:0", optimize=True)); want_fused.append(True)
    # 4:4:4 (chroma block = luma block, nothing interpolated) and grayscale (R = G = B) also leave as RGB, up to 128 MCUs
    # = 1024 pixels wide; 4:2:2 and everything else keeps the colour kernel
    for (h, w) in [(100, 150), (375, 500), (8, 8), (1, 1), (65, 1024), (33, 1025), (130, 517)]:
        enc.append(encode_jpeg(synth_image(rng, h, w), 85, subsampling="4:4:4")); want_fused.append(w <= 1024)
        enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80)); want_fused.append(w <= 1024)
    enc.append(encode_jpeg(synth_image(rng, 100, 150), 85, subsampling="4:2:2")); want_fused.append(False)
    enc.append(encode_jpeg(synth_image(rng, 64, 4), 85, subsampling="4:2:0")); want_fused.append(False)   # chroma 2 wide: box
    views, plan = B.decode_jpeg_batch(enc, device="cuda", fuse_color=True)
    torch.cuda.synchronize()
    assert list(plan.fused_color) == want_fused
    fused = [v.cpu().numpy() for v in views]
    plain = _decode_gpu(enc, fuse_color=False)
    for i, e in enumerate(enc):
        assert np.array_equal(fused[i], plain[i]), \
            f"sample {i} {fused[i].shape}: rows {sorted(set(np.nonzero((fused[i] != plain[i]).any(axis=(1, 2)))[0]))[:12]} differ"
        assert np.array_equal(fused[i], O.jpeg_decode_rgb(e)), f"sample {i}"


def _coefficients(enc, huffman, exact_scan=True):
    """Raw entropy-decoder output (int16 coefficient arrays) of both decoders."""
    from dali_amd import backend as B
    plan = B.JpegBatchPlan(enc, exact_scan=exact_scan)
    if huffman == "gpu":
        coef = torch.full((max(plan.coef_elems, 1),), 0x5555, dtype=torch.int16, device="cuda")  # the decoder zero-fills
        status, sel = plan.entropy_decode_gpu(coef)
        torch.cuda.synchronize()
        plan.check_gpu_status(status)
        return coef.cpu().numpy(), plan
    coef = torch.empty(max(plan.coef_elems, 1), dtype=torch.int16)
    plan.entropy_decode(coef)
    return coef.numpy(), plan


@pytest.mark.parametrize("exact_scan", [True, False])
def test_gpu_huffman_coefficients_equal_host_decoder(exact_scan):
    """The GPU entropy decoder must reproduce the host decoder's coefficient arrays exactly: sizes from one
    block to 6 MP, every subsampling, optimised Huffman tables, flat images (which never self-synchronise and
    exercise the relaxation's worst case) and noise at quality 100 (long codes, many stuffed bytes)."""
    rng = np.random.default_rng(99)
    enc = []
    for (h, w) in [(1, 1), (8, 8), (16, 16), (17, 23), (64, 48), (375, 500), (500, 375), (1080, 1920)]:
        for sub in ("4:4:4", "4:2:2", "4:2:0", "4:1:1"):
            enc.append(encode_jpeg(synth_image(rng, h, w), 90, subsampling=sub))
        enc.append(encode_jpeg(synth_image(rng, h, w), 75, optimize=True))
        enc.append(encode_jpeg(synth_image(rng, h, w, 1), 80))
    enc.append(encode_jpeg(np.full((480, 640, 3), 128, np.uint8), 90))                      # flat: DC+EOB only
    enc.append(encode_jpeg(np.full((333, 517, 3), (255, 0, 31), np.uint8), 50, subsampling="4:4:4"))
    enc.append(encode_jpeg(rng.integers(0, 256, (512, 768, 3), dtype=np.uint8), 100))      # white noise, q100
    enc.append(encode_jpeg(rng.integers(0, 256, (2000, 3000, 3), dtype=np.uint8), 95))     # ~6 MP, long stream
    half = synth_image(rng, 300, 400)
    half[:, 200:] = 7                                                                       # half flat
    enc.append(encode_jpeg(half, 85))
    # restart intervals: one MCU, a few MCUs, rows, intervals longer than a 61 KB segment; flat gray content whose
    # MCUs are shorter than a byte; a flat image with optimised tables (a 1-bit code) - all on the GPU decoder
    for (h, w), kw in [((375, 500), dict(restart_marker_blocks=1)), ((375, 500), dict(restart_marker_blocks=5)),
                       ((500, 375), dict(restart_marker_rows=1, subsampling="4:4:4")),
                       ((1080, 1920), dict(restart_marker_rows=2)), ((1080, 1920), dict(restart_marker_rows=40, quality=95)),
                       ((64, 48), dict(restart_marker_blocks=2, subsampling="4:2:2", optimize=True))]:
        enc.append(encode_jpeg(synth_image(rng, h, w), **({"quality": 85} | kw)))
    flat_gray = np.full((256, 1024), 128, np.uint8)
    flat_gray[:, 700:] = rng.integers(0, 255, (256, 324))
    enc.append(encode_jpeg(flat_gray, 75, restart_marker_blocks=3))
    enc.append(encode_jpeg(np.full((128, 192, 3), 90, np.uint8), 75, optimize=True))
    enc.append(encode_jpeg(synth_image(rng, 300, 200), 85) + bytes(5000))                  # bytes behind EOI
    got, plan = _coefficients(enc, "gpu", exact_scan)
    assert plan.gpu_eligible.all(), "every baseline single-scan stream takes the GPU decoder"
    ref, _ = _coefficients(enc, "host")
    assert got.shape == ref.shape
    if not np.array_equal(got, ref):
        for i in range(plan.n):
            a, b = int(plan.coef_off[i, 0]), int(plan.coef_off[i + 1, 0]) if i + 1 < plan.n else got.size
            assert np.array_equal(got[a:b], ref[a:b]), \
                f"sample {i} ({len(enc[i])} B): {np.count_nonzero(got[a:b] != ref[a:b])} coefficients differ, " \
                f"first at {np.nonzero(got[a:b] != ref[a:b])[0][0]}"


def test_gpu_huffman_fused_idct_planes_equal_the_idct_kernel():
    """The entropy decoder's fused output (dequantise + IDCT inside ExpandKernel, planes instead of coefficients) is
    bit-identical to storing the coefficients and running JpegIdctKernel - full images and block rectangles."""
    from dali_amd import backend as B
    rng = np.random.default_rng(21)
    enc = []
    for (h, w), kw in [((375, 500), dict(subsampling="4:2:0")), ((96, 131), dict(subsampling="4:4:4")),
                       ((200, 333), dict(subsampling="4:2:2")), ((64, 48), dict(subsampling="4:1:1", quality=98)),
                       ((130, 70), dict(subsampling="4:2:0", quality=20)), ((1, 1), {}), ((8, 9), {})]:
        enc.append(encode_jpeg(synth_image(rng, h, w), **({"quality": 85} | kw)))
    enc.append(encode_jpeg(synth_image(rng, 77, 91, 1), 80))                      # grayscale
    for rois in (None, [(10, 20, 100, 200), (3, 5, 60, 100), None, (8, 8, 32, 24), None, None, None, (0, 0, 77, 91)]):
        plan = B.JpegBatchPlan(enc, 16, rois=rois)
        assert plan.analyze_scans().all()
        dev = torch.device("cuda")
        plan.upload_streams(dev)
        coef = torch.zeros(plan.coef_elems, dtype=torch.int16, device=dev)
        ref = torch.zeros(plan.plane_bytes, dtype=torch.uint8, device=dev)
        got = torch.zeros_like(ref)
        rgb = torch.empty(max(plan.out_bytes, 1), dtype=torch.uint8, device=dev)
        status = plan.run_gpu_huffman(coef)
        B.jpeg_gpu_stage(plan, coef, ref, rgb)                                    # IDCT kernel -> ref planes
        torch.cuda.synchronize()
        plan.check_gpu_status(status)
        status = plan.run_gpu_huffman(None, planes_dev=got)                        # fused -> planes, no coefficients
        torch.cuda.synchronize()
        plan.check_gpu_status(status)
        assert torch.equal(ref, got)
        assert int(ref.count_nonzero()) > 0


def test_gpu_huffman_eligibility_and_mixed_batches():
    rng = np.random.default_rng(5)
    img = synth_image(rng, 120, 160)
    enc = [encode_jpeg(img, 85), encode_jpeg(img, 85, progressive=True), encode_jpeg(img, 85, restart_marker_blocks=4),
           encode_jpeg(img[..., 0], 85), encode_jpeg(img, 85, optimize=True)]
    from dali_amd import backend as B
    plan = B.JpegBatchPlan(enc)
    assert plan.analyze_scans().tolist() == [True, False, True, True, True]
    got = _decode_gpu(enc, huffman="gpu")
    for i, e in enumerate(enc):
        assert np.array_equal(got[i], O.jpeg_decode_rgb(e)), f"sample {i}"


def test_gpu_huffman_truncated_stream_raises():
    from dali_amd import backend as B
    from dali_amd._capi import DaliAmdError
    rng = np.random.default_rng(6)
    e = encode_jpeg(synth_image(rng, 200, 300), 85)
    cut = e[:len(e) // 2] + b"\xff\xd9"
    with pytest.raises(DaliAmdError, match="corrupt JPEG data"):
        B.decode_jpeg_batch([cut], device="cuda", huffman="gpu")


@pytest.mark.parametrize("huffman", ["gpu", "host"])
def test_region_of_interest_decode_equals_decode_then_crop(huffman):
    """decoders.image_crop / image_random_crop semantics (dali/test/python/decoder/test_image.py:118-216): decoding a
    window must give exactly the pixels of the full decode, for every chroma layout (the window's edges need the
    neighbouring chroma samples for fancy upsampling) and for windows touching the image borders."""
    rng = np.random.default_rng(31)
    enc, rois = [], []
    for (h, w) in [(64, 64), (97, 131), (375, 500), (31, 17), (8, 8), (200, 333)]:
        for kw in [dict(subsampling="4:4:4"), dict(subsampling="4:2:2"), dict(subsampling="4:2:0"),
                   dict(subsampling="4:1:1"), dict(subsampling="4:2:0", progressive=True)]:
            e = encode_jpeg(synth_image(rng, h, w), 85, **kw)
            for _ in range(3):
                ch, cw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
                y0, x0 = int(rng.integers(0, h - ch + 1)), int(rng.integers(0, w - cw + 1))
                enc.append(e)
                rois.append((y0, x0, ch, cw))
            enc.append(e)
            rois.append((0, 0, h, w))                                   # the whole image as a window
            enc.append(e)
            rois.append((h - 1, w - 1, 1, 1))                            # the last pixel
        g = encode_jpeg(synth_image(rng, h, w, 1), 80)
        enc.append(g)
        rois.append((h // 3, w // 4, max(1, h // 2), max(1, w // 2)))
    enc.append(encode_jpeg(synth_image(rng, 120, 160), 90))
    rois.append(None)                                                    # mixed with an un-cropped sample
    got = _decode_gpu(enc, huffman=huffman, rois=rois)
    cache = {}
    for i, (e, r) in enumerate(zip(enc, rois)):
        if id(e) not in cache:
            cache[id(e)] = O.jpeg_decode_rgb(e)
        ref = cache[id(e)]
        if r is not None:
            ref = ref[r[0]:r[0] + r[2], r[1]:r[1] + r[3]]
        assert got[i].shape == ref.shape, (i, r)
        assert np.array_equal(got[i], ref), f"sample {i} roi {r}: max diff {np.abs(got[i].astype(int) - ref).max()}"


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


def test_slices_with_more_block_starts_than_their_lds_list_holds():
    """SyncKernel's write phase (round 5): a lane whose list of block starts outgrew its 17 LDS slots decodes again FROM THE
    LAST START ITS LIST HOLDS - block index inside the MCU advanced by the blocks that ended, one block fewer for the stream's
    very first slice (whose list begins with the start of block 0), from the slice's first bit for restart-interval streams.
    Flat and nearly flat content (2-6 bits per block: hundreds of starts per 256-byte slice) in every MCU structure, with
    texture in between so that overflowing and ordinary slices alternate inside one workgroup; several segments."""
    rng = np.random.default_rng(31)
    enc = []

    def striped(h, w, ch=3):
        # flat bands (hundreds of blocks per slice) between textured ones (a dozen)
        img = synth_image(rng, h, w, ch).copy()
        for y0 in range(0, h, 96):
            img[y0:y0 + 56] = rng.integers(0, 256, ch if ch > 1 else 1, dtype=np.uint8) if ch > 1 else rng.integers(0, 256)
        return img
    for sub in ("4:2:0", "4:2:2", "4:4:4", "4:1:1"):
        enc.append(encode_jpeg(np.full((264, 520, 3), (17, 200, 90), np.uint8), 90, subsampling=sub))     # flat from the first bit
        enc.append(encode_jpeg(striped(400, 600), 88, subsampling=sub))
        enc.append(encode_jpeg(striped(400, 600), 88, subsampling=sub, restart_marker_blocks=11))
        enc.append(encode_jpeg(striped(300, 420), 60, subsampling=sub, optimize=True))
    enc.append(encode_jpeg(np.full((512, 512), 77, np.uint8), 90))                                         # grayscale: one block per MCU
    enc.append(encode_jpeg(striped(480, 640, 1), 85))
    enc.append(encode_jpeg(striped(480, 640, 1), 85, restart_marker_rows=2))
    enc.append(encode_jpeg(striped(1600, 2000), 80, subsampling="4:2:0"))                                  # several segments
    for kw in (dict(huffman="gpu"), dict(huffman="gpu", exact_scan=False)):
        got = _decode_gpu(enc, **kw)
        for i, e in enumerate(enc):
            ref = O.jpeg_decode_rgb(e)
            assert np.array_equal(got[i], ref), f"sample {i} ({kw}): rows {sorted(set(np.nonzero((got[i] != ref).any(axis=(1, 2)))[0]))[:8]}"
    a, plan = _coefficients(enc, "gpu")
    b, _ = _coefficients(enc, "host")
    assert np.array_equal(a, b)
