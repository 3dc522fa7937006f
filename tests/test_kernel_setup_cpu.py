"""Host-side halves of the C ABI that need no GPU: descriptor validation / grid sizing (`*Setup`), scratch sizing and
the region-of-interest planner.  They are pure host code inside libdali_amd_kernels.so."""
import ctypes as C

import numpy as np
import pytest

from dali_amd import _capi as capi


def _plan(w, h, ncomp, hs, vs, orientation, y0, x0, hh, ww):
    lib = capi.kernels()
    plan = capi.JpegRoiPlan()
    rc = lib.daliamdJpegPlanRoi(w, h, ncomp, (C.c_int32 * 3)(*hs), (C.c_int32 * 3)(*vs), orientation, y0, x0, hh, ww,
                                C.byref(plan))
    return rc, plan


def test_roi_plan_identity_and_block_rectangles():
    # 4:2:0, 100 x 80 image, window rows 16..48, columns 32..64
    rc, p = _plan(100, 80, 3, (2, 1, 1), (2, 1, 1), 1, 16, 32, 32, 32)
    assert rc == 0
    assert (p.roi_x0, p.roi_y0, p.roi_w, p.roi_h, p.out_x0, p.out_y0) == (32, 16, 32, 32, 32, 16)
    # luma: exactly the co-sited samples -> blocks x 4..8, y 2..6
    assert list(p.rect[0]) == [4, 2, 8, 6]
    # chroma (half resolution, one neighbour on each side for the triangle filter): samples x 15..32, y 7..24
    assert list(p.rect[1]) == [1, 0, 5, 4] and list(p.rect[2]) == [1, 0, 5, 4]


def test_roi_plan_clamps_at_the_image_border_and_covers_whole_image():
    rc, p = _plan(100, 80, 3, (2, 1, 1), (2, 1, 1), 1, 0, 0, 80, 100)
    assert rc == 0
    assert list(p.rect[0]) == [0, 0, 13, 10]           # ceil(100/8), ceil(80/8)
    assert list(p.rect[1]) == [0, 0, 7, 5]              # chroma 50 x 40 samples


@pytest.mark.parametrize("orientation", range(1, 9))
def test_roi_plan_orientation_maps_the_window_back_to_the_source(orientation):
    """Brute force: push every source pixel through the EXIF mapping used by the colour kernel and check that the
    planned source window is exactly the pre-image of the upright window."""
    W, H = 37, 23                                     # source (stored) size
    UH, UW = (W, H) if orientation >= 5 else (H, W)
    y0, x0, hh, ww = 3, 5, 9, 11
    rc, p = _plan(W, H, 3, (1, 1, 1), (1, 1, 1), orientation, y0, x0, hh, ww)
    assert rc == 0 and (p.out_y0, p.out_x0) == (y0, x0)
    inside = np.zeros((H, W), bool)
    for y in range(H):
        for x in range(W):
            oy, ox = {1: (y, x), 2: (y, W - 1 - x), 3: (H - 1 - y, W - 1 - x), 4: (H - 1 - y, x), 5: (x, y),
                      6: (x, H - 1 - y), 7: (W - 1 - x, H - 1 - y), 8: (W - 1 - x, y)}[orientation]
            assert 0 <= oy < UH and 0 <= ox < UW
            inside[y, x] = y0 <= oy < y0 + hh and x0 <= ox < x0 + ww
    ys, xs = np.nonzero(inside)
    assert (p.roi_y0, p.roi_x0) == (ys.min(), xs.min())
    assert (p.roi_h, p.roi_w) == (ys.max() - ys.min() + 1, xs.max() - xs.min() + 1)
    assert inside[p.roi_y0:p.roi_y0 + p.roi_h, p.roi_x0:p.roi_x0 + p.roi_w].all()


def test_roi_plan_rejects_windows_outside_the_image():
    rc, _ = _plan(100, 80, 3, (2, 1, 1), (2, 1, 1), 1, 70, 0, 20, 10)
    assert rc != 0
    assert b"does not fit" in capi.kernels().daliamdGetLastErrorMessage()
    rc, _ = _plan(100, 80, 3, (2, 1, 1), (2, 1, 1), 6, 0, 0, 90, 10)   # rotated: the upright image is 100 x 80 (H x W)
    assert rc == 0


def test_huffman_scratch_grows_with_the_stream_and_the_block_count():
    lib = capi.kernels()
    sizes = []
    for ecs, blocks in [(0, 1), (1000, 6), (100_000, 4400), (100_000, 8800), (1_000_000, 4400)]:
        n = C.c_size_t(0)
        assert lib.daliamdJpegHuffmanScratchBytes(ecs, blocks, C.byref(n)) == 0
        sizes.append(n.value)
        assert n.value % 256 == 0 and n.value >= ecs + 64
    assert sizes == sorted(sizes)
    assert lib.daliamdJpegHuffmanScratchBytes(-1, 1, C.byref(C.c_size_t())) != 0


def _huff_desc(ecs_len=5000, total_blocks=600, bits0=0):
    d = capi.JpegHuffDesc()
    d.ecs, d.scratch, d.status = 0x1000, 0x2000, 0x3000     # fake device addresses: Setup never dereferences them
    for c in range(3):
        d.coef[c] = 0x10000 * (c + 1)
        d.blocks_x[c] = 10 if c else 20
        d.h_samp[c] = d.v_samp[c] = 1 if c else 2
    d.ecs_len, d.blocks_per_mcu, d.mcus_x, d.total_blocks = ecs_len, 6, 10, total_blocks
    for k, comp in enumerate([0, 0, 0, 0, 1, 2]):
        d.comp_of_block[k] = comp
    for t in range(4):
        d.bits[t][0] = bits0
        d.bits[t][1] = 1
    return d


def test_huffman_setup_sizes_the_three_grids_and_validates():
    lib = capi.kernels()
    descs = (capi.JpegHuffDesc * 3)(_huff_desc(5000, 600), _huff_desc(100_000, 6000), _huff_desc(40_000, 2400))
    tiles, segs, bwg = C.c_int(), C.c_int(), C.c_int()
    assert lib.daliamdJpegHuffmanSetup(descs, 3, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
    assert [d.tile_start for d in descs] == [0, 1, 14] and tiles.value == 19         # 8 KB tiles
    assert [d.seg_start for d in descs] == [0, 1, 3] and segs.value == 4             # 244 slices of 256 bytes
    # block-decoding workgroups: the same number of MCUs (a multiple of 32) per workgroup for streams of the same geometry
    counts = [descs[1].blk_wg_start - descs[0].blk_wg_start, descs[2].blk_wg_start - descs[1].blk_wg_start,
              bwg.value - descs[2].blk_wg_start]
    assert descs[0].blk_wg_start == 0
    assert any(counts == [-(-m // mpw) for m in (100, 1000, 400)] for mpw in range(32, 513, 32)), counts
    bad = (capi.JpegHuffDesc * 1)(_huff_desc(total_blocks=601))                      # not a whole number of MCUs
    assert lib.daliamdJpegHuffmanSetup(bad, 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) != 0
    one_bit = (capi.JpegHuffDesc * 1)(_huff_desc(bits0=1))                           # a 1-bit code is fine (round 4)
    assert lib.daliamdJpegHuffmanSetup(one_bit, 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
    bad = (capi.JpegHuffDesc * 1)(_huff_desc())
    bad[0].restart_interval = 70000                                                  # DRI is a 16-bit field
    assert lib.daliamdJpegHuffmanSetup(bad, 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) != 0
    # scratch: the restart boundaries (4 bytes per interval) come behind everything else
    plain, rst = C.c_size_t(), C.c_size_t()
    assert lib.daliamdJpegHuffmanScratchBytes(100_000, 6000, C.byref(plain)) == 0
    assert lib.daliamdJpegHuffmanScratchBytesRestart(100_000, 6000, 1000, C.byref(rst)) == 0
    assert 4000 - 256 <= rst.value - plain.value <= 4000 + 256                    # (the total is rounded up to 256 bytes)


def test_huffman_setup_shares_code_tables_between_streams_with_the_same_dht():
    """table_owner: streams whose DHT contents and MCU structure are identical use the tables the first of them builds."""
    lib = capi.kernels()
    descs = (capi.JpegHuffDesc * 5)(*[_huff_desc(5000 + 100 * i, 600) for i in range(5)])
    descs[2].vals[1][3] = 7                      # another symbol list
    descs[4].vals[1][3] = 7                      # ... the same as stream 2's
    descs[3].blocks_per_mcu, descs[3].total_blocks = 3, 600
    for k, comp in enumerate([0, 1, 2, 0, 0, 0]):
        descs[3].comp_of_block[k] = comp         # 4:4:4: another MCU structure with the standard DHT
    tiles, segs, bwg = C.c_int(), C.c_int(), C.c_int()
    assert lib.daliamdJpegHuffmanSetup(descs, 5, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
    assert [d.table_owner for d in descs] == [0, 0, 2, 3, 2]


def test_normalize_setup_views_and_grids():
    lib = capi.kernels()
    descs = (capi.NormalizeDesc * 3)()
    for d, (o, r, i) in zip(descs, [(1, 480 * 640, 3), (513, 1000, 1), (1, 300, 513)]):
        d.in_, d.out = 0x1000, 0x2000
        d.outer, d.reduced, d.inner = o, r, i
        d.in_dtype, d.out_dtype = capi.UINT8, capi.FLOAT
    sw, aw, bins = C.c_int(), C.c_int(), C.c_int64()
    assert lib.daliamdNormalizeSetup(descs, 3, C.byref(sw), C.byref(aw), C.byref(bins)) == 0
    assert bins.value == 513                                        # the largest outer*inner of the batch
    assert descs[0].stat_chunks == -(-480 * 640 * 3 // 65536)      # narrow inner: 64 K elements per workgroup
    assert descs[1].stat_chunks == 1                               # one workgroup per row of 1000
    assert descs[2].stat_chunks == -(-300 // 64) * 3               # wide inner: 64 rows x 256 columns per workgroup
    assert sw.value == descs[0].stat_chunks + 513 + descs[2].stat_chunks
    descs[0].in_dtype = capi.FLOAT16
    assert lib.daliamdNormalizeSetup(descs, 3, C.byref(sw), C.byref(aw), C.byref(bins)) != 0


def test_huffman_setup_fused_output_needs_aligned_planes_not_coefficients():
    """With plane[] set the decoder writes samples instead of coefficients: coef may be NULL, the planes must allow
    8-byte stores."""
    lib = capi.kernels()
    tiles, segs, bwg = C.c_int(), C.c_int(), C.c_int()

    def fused(pitch0=160, plane0=0x40000, with_coef=False):
        d = _huff_desc()
        for c in range(3):
            if not with_coef:
                d.coef[c] = 0
            d.plane[c] = 0x40000 * (c + 1)
            d.plane_pitch[c] = 80 if c else 160
        d.plane[0], d.plane_pitch[0] = plane0, pitch0
        return (capi.JpegHuffDesc * 1)(d)
    assert lib.daliamdJpegHuffmanSetup(fused(), 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
    assert lib.daliamdJpegHuffmanSetup(fused(with_coef=True), 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
    for bad in (fused(pitch0=164), fused(pitch0=152), fused(plane0=0x40004)):   # pitch % 8, pitch < blocks_x*8, alignment
        assert lib.daliamdJpegHuffmanSetup(bad, 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) != 0
        assert b"planes must be 8-byte aligned" in lib.daliamdGetLastErrorMessage()
    neither = _huff_desc()
    for c in range(3):
        neither.coef[c] = 0
    assert lib.daliamdJpegHuffmanSetup((capi.JpegHuffDesc * 1)(neither), 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) != 0


def test_huffman_setup_fused_colour_output():
    """rgb != NULL (round 4): only through SetupColor, only for 4:2:0 streams in the usual block order up to 128 MCUs
    wide without a block rectangle; the block grid is one workgroup per band of whole MCU rows."""
    lib = capi.kernels()
    tiles, segs, bwg, kinds = C.c_int(), C.c_int(), C.c_int(), C.c_int()

    def color(mcus_x=10, mcus_y=10, width=None, height=None, rgb=0x80000, pitch=None):
        d = _huff_desc(total_blocks=mcus_x * mcus_y * 6)
        d.mcus_x = mcus_x
        for k, (ho, vo) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (0, 0)]):
            d.h_of_block[k], d.v_of_block[k] = ho, vo
        d.width = mcus_x * 16 - 3 if width is None else width
        d.height = mcus_y * 16 - 5 if height is None else height
        d.rgb, d.rgb_pitch = rgb, (3 * d.width + 7) // 8 * 8 if pitch is None else pitch
        return d

    def setup(*descs):
        arr = (capi.JpegHuffDesc * len(descs))(*descs)
        rc = lib.daliamdJpegHuffmanSetupColor(arr, len(descs), C.byref(tiles), C.byref(segs), C.byref(bwg), C.byref(kinds))
        return rc, arr

    assert lib.daliamdJpegHuffmanColorFusable(C.byref(color())) == 1
    rc, arr = setup(color(10, 10), _huff_desc(), color(128, 3), color(33, 7), color(64, 5))
    # bits 0 / 1: plane and fused colour outputs; round 5: 4 = streams parsed in the launch, 32 = code tables built in it
    assert rc == 0 and kinds.value == (3 | 4 | 32)
    # bands: 12 MCU rows of 10 MCUs -> 1 band; the plain stream 100 MCUs -> its own count; 128 wide: one row per band;
    # 33 wide: 3 rows per band -> 3 bands; 64 wide: 2 rows per band -> 3 bands
    starts = [d.blk_wg_start for d in arr] + [bwg.value]
    counts = [b - a for a, b in zip(starts, starts[1:])]
    assert counts[0] == 1 and counts[2] == 3 and counts[3] == 3 and counts[4] == 3, counts
    rc, _ = setup(_huff_desc())
    assert rc == 0 and kinds.value == (1 | 4 | 32)
    # the plain entry points refuse a table with rgb outputs
    assert lib.daliamdJpegHuffmanSetup((capi.JpegHuffDesc * 1)(color()), 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) != 0
    assert b"daliamdJpegHuffmanSetupColor" in lib.daliamdGetLastErrorMessage()
    # geometry the fused output does not take
    wide = color(129, 2)
    assert lib.daliamdJpegHuffmanColorFusable(C.byref(wide)) == 0 and setup(wide)[0] != 0
    sub = color()
    sub.h_samp[0] = 1
    assert lib.daliamdJpegHuffmanColorFusable(C.byref(sub)) == 0
    rect = color()
    rect.rect[0][2] = 4
    assert lib.daliamdJpegHuffmanColorFusable(C.byref(rect)) == 0
    order = color()
    order.comp_of_block[4], order.comp_of_block[5] = 2, 1
    assert lib.daliamdJpegHuffmanColorFusable(C.byref(order)) == 0
    # buffer rules: alignment, pitch, size consistent with the MCU grid, more than 4 pixels wide
    for bad in (color(rgb=0x80004), color(pitch=3 * 157 + 5), color(pitch=8), color(width=170), color(height=200),
                color(1, 4, width=4)):
        assert setup(bad)[0] != 0, "accepted a bad descriptor"


def test_huffman_setup_sizes_the_block_grid_from_the_region_of_interest():
    """Region-of-interest decode (daliamdJpegHuffDesc.rect): the block kernel's workgroups walk the bounding rectangle of
    the MCUs that hold a wanted block of any component, so the grid is ceil(rectangle MCUs / MCUs per workgroup) - checked
    against a brute-force walk over the MCUs for random rectangles of a 4:2:0 frame (the planner's shapes: luma rect in
    blocks, chroma rect = luma / 2 with a margin)."""
    lib = capi.kernels()
    rng = np.random.default_rng(5)
    tiles, segs, bwg = C.c_int(), C.c_int(), C.c_int()
    full = _huff_desc(total_blocks=40 * 30 * 6)
    full.mcus_x = 40
    assert lib.daliamdJpegHuffmanSetup((capi.JpegHuffDesc * 1)(full), 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
    mpw = -(-40 * 30 // bwg.value)            # MCUs per workgroup (a multiple of 32), from the whole-frame grid
    mpw = next(m for m in range(32, 513, 32) if -(-40 * 30 // m) == bwg.value)
    for _ in range(200):
        mx, my = 40, 30
        x0, y0 = int(rng.integers(0, 2 * mx - 1)), int(rng.integers(0, 2 * my - 1))
        x1, y1 = int(rng.integers(x0 + 1, 2 * mx + 1)), int(rng.integers(y0 + 1, 2 * my + 1))
        d = _huff_desc(total_blocks=mx * my * 6)
        d.mcus_x = mx
        for k, (ho, vo) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (0, 0)]):
            d.h_of_block[k], d.v_of_block[k] = ho, vo
        rects = {0: (x0, y0, x1, y1),
                 1: (max(x0 // 2 - 1, 0), max(y0 // 2 - 1, 0), min((x1 + 1) // 2 + 1, mx), min((y1 + 1) // 2 + 1, my))}
        rects[2] = rects[1]
        for c, r in rects.items():
            for j in range(4):
                d.rect[c][j] = r[j]
        assert lib.daliamdJpegHuffmanSetup((capi.JpegHuffDesc * 1)(d), 1, C.byref(tiles), C.byref(segs), C.byref(bwg)) == 0
        need = np.zeros((my, mx), bool)       # MCUs with a wanted block
        for yy in range(my):
            for xx in range(mx):
                luma = any(rects[0][0] <= 2 * xx + ho < rects[0][2] and rects[0][1] <= 2 * yy + vo < rects[0][3]
                           for ho in (0, 1) for vo in (0, 1))
                chroma = rects[1][0] <= xx < rects[1][2] and rects[1][1] <= yy < rects[1][3]
                need[yy, xx] = luma or chroma
        ys, xs = np.nonzero(need)
        box = (xs.max() - xs.min() + 1) * (ys.max() - ys.min() + 1)
        assert bwg.value == -(-box // mpw), (rects, box, bwg.value)
