"""GPU end-to-end: the DALI-style pipeline (readers.file -> decoders.image(mixed) -> random_resized_crop ->
crop_mirror_normalize) through the C++ host framework, checked against the oracle composition, plus the
iterator and operator-fusion behaviour."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image, ImageOps

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image, synth_jpeg_batch

pytestmark = pytest.mark.gpu

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


@pytest.fixture(scope="module")
def jpeg_dir(tmp_path_factory):
    root = tmp_path_factory.mktemp("jpegs")
    rng = np.random.default_rng(21)
    files = []
    enc = synth_jpeg_batch(rng, 24, sizes=[(120, 160), (160, 120), (200, 200), (97, 131), (240, 320)])
    for i, e in enumerate(enc):
        d = root / f"class_{i % 3}"
        os.makedirs(d, exist_ok=True)
        p = d / f"img_{i:03d}.jpg"
        p.write_bytes(e)
    for c in range(3):
        for f in sorted(os.listdir(root / f"class_{c}")):
            files.append((str(root / f"class_{c}" / f), c))
    return str(root), files


def _train_pipe(root, bs, fused=True, depth=2, **reader_kw):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=bs, num_threads=4, device_id=0, seed=3, prefetch_queue_depth=depth)
    with pipe:
        jpegs, labels = fn.readers.file(file_root=root, name="Reader", **reader_kw)
        images = fn.decoders.image(jpegs, device="mixed", output_type=types.RGB)
        crops = fn.random_resized_crop(images, size=[64, 80], seed=1234)
        flip = fn.random.coin_flip(probability=0.5, seed=4321)
        out = fn.crop_mirror_normalize(crops, dtype=types.FLOAT16, output_layout="CHW", mean=MEAN, std=STD, mirror=flip)
        if fused:
            pipe.set_outputs(out, labels)
        else:
            pipe.set_outputs(out, labels, crops)
    pipe.build()
    return pipe


def _oracle_batch(files, picks, it, bs, out_hw=(64, 80)):
    imgs = [O.jpeg_decode_rgb(open(files[k][0], "rb").read()) for k in picks]
    anchors, crops = O.rrc_batch(1234, it, [im.shape[:2] for im in imgs])
    mirror = O.coin_flip_batch(4321, it, bs, 0.5)
    mean, inv = O.cmn_norm_args(MEAN, STD)
    u8, f16 = [], []
    for i, im in enumerate(imgs):
        roi = (anchors[i][0], anchors[i][1], anchors[i][0] + crops[i][0], anchors[i][1] + crops[i][1])
        r = O.resample_u8(im, out_hw, roi=roi)
        u8.append(r)
        f16.append(O.cmn_u8(r, (0, 0), out_hw, mirror=bool(mirror[i]), mean=mean, inv_std=inv, dtype=O.F16))
    return np.stack(u8), np.stack(f16)


@pytest.mark.parametrize("fused", [True, False])
def test_train_pipeline_matches_oracle(jpeg_dir, fused):
    root, files = jpeg_dir
    bs = 8
    pipe = _train_pipe(root, bs, fused=fused)
    for it in range(4):
        outs = pipe.run()
        data = outs[0].as_tensor().cpu().numpy()
        labels = outs[1].as_array().reshape(-1)
        picks = [(it * bs + i) % len(files) for i in range(bs)]
        assert list(labels) == [files[k][1] for k in picks]
        ref_u8, ref_f16 = _oracle_batch(files, picks, it, bs)
        assert data.shape == (bs, 3, 64, 80) and data.dtype == np.float16
        assert np.array_equal(data.view(np.uint16), ref_f16.view(np.uint16)), f"iteration {it}"
        kernels = pipe.executed_kernels()
        if fused:
            assert "fused_resample_cmn" in kernels and "resample" not in kernels and "cmn" not in kernels
        else:
            assert "resample" in kernels and "cmn" in kernels and "fused_resample_cmn" not in kernels
            assert np.array_equal(outs[2].as_tensor().cpu().numpy(), ref_u8)
        # (the stand-alone IDCT kernel only runs for streams the host entropy decoder took: the GPU decoder's block
        # output is already dequantised and inverse-transformed)
        # (... and, round 4, usually colour-converted as well: "jpeg_huffman_rgb"; the colour launch is for the rest)
        assert "jpeg_huffman" in kernels and ("jpeg_color" in kernels or "jpeg_huffman_rgb" in kernels)


def test_train_pipeline_with_the_fused_colour_output(jpeg_dir, monkeypatch):
    """DALI_AMD_FUSE_COLOR=1: decoders.image hands the entropy decoder the output images (daliamdJpegHuffDesc.rgb) and
    launches the colour kernel only for what is left; the batches stay bit-identical to the oracle."""
    monkeypatch.setenv("DALI_AMD_FUSE_COLOR", "1")
    root, files = jpeg_dir
    bs = 8
    pipe = _train_pipe(root, bs, fused=False)
    for it in range(4):
        outs = pipe.run()
        picks = [(it * bs + i) % len(files) for i in range(bs)]
        ref_u8, ref_f16 = _oracle_batch(files, picks, it, bs)
        assert np.array_equal(outs[0].as_tensor().cpu().numpy().view(np.uint16), ref_f16.view(np.uint16)), f"iteration {it}"
        assert np.array_equal(outs[2].as_tensor().cpu().numpy(), ref_u8)
        kernels = pipe.executed_kernels()
        assert "jpeg_huffman_rgb" in kernels and "jpeg_color" not in kernels


@pytest.mark.parametrize("depth,streams", [(5, None), (7, None), (5, "0"), (4, "2")])
def test_deep_prefetch_shares_compute_streams_and_stays_exact(jpeg_dir, depth, streams, monkeypatch):
    # more iterations in flight than the executor's three compute streams (pipeline.cpp: slot s on stream s mod 3;
    # DALI_AMD_PIPELINE_STREAMS=0: one per slot): every iteration of two epochs still matches the oracle bit for bit
    root, files = jpeg_dir
    if streams is not None:
        monkeypatch.setenv("DALI_AMD_PIPELINE_STREAMS", streams)
    bs = 8
    pipe = _train_pipe(root, bs, depth=depth)
    for it in range(2 * len(files) // bs + depth):
        outs = pipe.run()
        data = outs[0].as_tensor().cpu().numpy()
        picks = [(it * bs + i) % len(files) for i in range(bs)]
        assert list(outs[1].as_array().reshape(-1)) == [files[k][1] for k in picks]
        _, ref_f16 = _oracle_batch(files, picks, it, bs)
        assert np.array_equal(data.view(np.uint16), ref_f16.view(np.uint16)), f"iteration {it}"


def test_decoder_output_and_exif_orientation(tmp_path):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(2)
    files = []
    for o in range(1, 9):
        img = Image.fromarray(synth_image(rng, 40 + o, 64))
        exif = Image.Exif()
        exif[0x0112] = o
        p = tmp_path / f"o{o}.jpg"
        img.save(p, "JPEG", quality=90, exif=exif)
        files.append(str(p))
    for adjust in (True, False):
        pipe = Pipeline(batch_size=8, num_threads=2, device_id=0)
        with pipe:
            enc, _ = fn.readers.file(files=files)
            pipe.set_outputs(fn.decoders.image(enc, device="mixed", adjust_orientation=adjust))
        (out,) = pipe.run()
        assert out.layout() == "HWC"
        for i, f in enumerate(files):
            im = Image.open(f)
            ref = np.asarray((ImageOps.exif_transpose(im) if adjust else im).convert("RGB"))
            got = out[i].as_cpu()
            assert got.shape == ref.shape, (i, adjust)
            assert np.array_equal(got, ref), (i, adjust)


def test_decoder_error_names_the_file(tmp_path):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    good = tmp_path / "good.jpg"
    good.write_bytes(encode_jpeg(synth_image(np.random.default_rng(0), 32, 32)))
    bad = tmp_path / "broken.jpg"
    bad.write_bytes(b"\xff\xd8 this is not a jpeg")
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=[str(good), str(bad)])
        pipe.set_outputs(fn.decoders.image(enc, device="mixed"))
    with pytest.raises(RuntimeError, match="broken.jpg"):
        pipe.run()


def test_decoder_gpu_and_host_huffman_paths_agree(tmp_path):
    """Baseline streams take the GPU entropy decoder, progressive / restart-marker streams the host one, and an
    explicit hybrid_huffman_threshold sends small images to the host as in the reference; all bit-exact."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(8)
    encs = [encode_jpeg(synth_image(rng, 96, 128), 85), encode_jpeg(synth_image(rng, 96, 128), 85, progressive=True),
            encode_jpeg(synth_image(rng, 200, 150), 75, restart_marker_blocks=5), encode_jpeg(synth_image(rng, 64, 64, 1), 80),
            encode_jpeg(synth_image(rng, 300, 400), 90, subsampling="4:4:4"), encode_jpeg(synth_image(rng, 17, 9), 60)]
    files = []
    for i, e in enumerate(encs):
        p = tmp_path / f"f{i}.jpg"
        p.write_bytes(e)
        files.append(str(p))
    refs = [O.jpeg_decode_rgb(e) for e in encs]
    for kw, expect_gpu in [({}, True), ({"hybrid_huffman_threshold": 10 ** 9}, False), ({"hybrid_huffman_threshold": 20000}, True)]:
        pipe = Pipeline(batch_size=len(files), num_threads=3, device_id=0)
        with pipe:
            enc, _ = fn.readers.file(files=files)
            pipe.set_outputs(fn.decoders.image(enc, device="mixed", **kw))
        for _ in range(3):   # the ring of staging buffers comes round
            (out,) = pipe.run()
            assert ("jpeg_huffman" in pipe.executed_kernels()) == expect_gpu
            for i, ref in enumerate(refs):
                assert np.array_equal(out[i].as_cpu(), ref), (kw, i)


def test_decoder_truncated_stream_is_reported_with_the_file_name(tmp_path):
    """A stream that parses but runs out of entropy-coded data is detected by the GPU decoder's status word; the
    error surfaces when the outputs of that iteration are requested and names the sample."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(9)
    good = encode_jpeg(synth_image(rng, 120, 160), 85)
    cut = encode_jpeg(synth_image(rng, 240, 320), 85)
    cut = cut[:len(cut) // 2] + b"\xff\xd9"
    (tmp_path / "fine.jpg").write_bytes(good)
    (tmp_path / "truncated.jpg").write_bytes(cut)
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(files=[str(tmp_path / "fine.jpg"), str(tmp_path / "truncated.jpg")])
        pipe.set_outputs(fn.decoders.image(enc, device="mixed"))
    with pytest.raises(RuntimeError, match="truncated.jpg.*corrupt JPEG data"):
        pipe.run()


def test_standalone_cmn_with_crop_and_pad(jpeg_dir):
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    root, files = jpeg_dir
    pipe = Pipeline(batch_size=6, num_threads=2, device_id=0)
    with pipe:
        enc, _ = fn.readers.file(file_root=root)
        img = fn.decoders.image(enc, device="mixed")
        out = fn.crop_mirror_normalize(img, crop=(90, 100), dtype=types.FLOAT, output_layout="HWC", mean=[128.0],
                                       std=[64.0], pad_output=True, crop_pos_x=0.25, crop_pos_y=1.0)
        pipe.set_outputs(out)
    (out,) = pipe.run()
    got = out.as_tensor().cpu().numpy()
    mean, inv = O.cmn_norm_args([128.0], [64.0])
    for i in range(6):
        im = O.jpeg_decode_rgb(open(files[i][0], "rb").read())
        ay, ax = O.crop_anchor(1.0, 90, im.shape[0]), O.crop_anchor(0.25, 100, im.shape[1])
        ref = O.cmn_u8(im, (ay, ax), (90, 100), mean=mean, inv_std=inv, layout="HWC", pad_output=True, dtype=O.F32)
        assert np.array_equal(got[i].view(np.uint32), ref.view(np.uint32))
    assert pipe.executed_kernels().count("cmn") == 1


def test_iterator_two_shards(jpeg_dir):
    from dali_amd.plugin.pytorch import DALIGenericIterator, LastBatchPolicy
    root, files = jpeg_dir
    pipes = [_train_pipe(root, 4, shard_id=k, num_shards=2, pad_last_batch=True) for k in range(2)]
    it = DALIGenericIterator(pipes, ["data", "label"], reader_name="Reader", last_batch_policy=LastBatchPolicy.PARTIAL,
                             auto_reset=True)
    seen = [[], []]
    for batch in it:
        for g in range(2):
            assert batch[g]["data"].is_cuda and batch[g]["data"].dtype == torch.float16
            assert batch[g]["data"].shape[1:] == (3, 64, 80)
            seen[g] += [int(v) for v in batch[g]["label"].reshape(-1)]
    assert len(seen[0]) == 12 and len(seen[1]) == 12
    assert seen[0] + seen[1] == [f[1] for f in files]
    # second epoch: shards rotate
    seen2 = [[], []]
    for batch in it:
        for g in range(2):
            seen2[g] += [int(v) for v in batch[g]["label"].reshape(-1)]
    assert seen2[0] == seen[1] and seen2[1] == seen[0]


@pytest.mark.parametrize("depth", [1, 2, 5])
def test_iterator_stream_ordered_handover(jpeg_dir, depth):
    """DALIGenericIterator hands the batch over in stream order (no host wait): with the consumer's stream kept busy, so
    that its copy of batch i is still pending while the pipeline wants to write the ring slot again, every batch must
    still equal what pipeline.run() returns for the same iteration - the slot's reuse waits for the release event."""
    from dali_amd.plugin.pytorch import DALIGenericIterator
    root, _ = jpeg_dir
    iters = 3 * (depth + 1) + 2
    ref_pipe = _train_pipe(root, 6, depth=depth)
    want = []
    for _ in range(iters):
        out, labels = ref_pipe.run()
        want.append((out.as_tensor().clone().cpu(), labels.as_array().copy()))
    del ref_pipe
    it = DALIGenericIterator([_train_pipe(root, 6, depth=depth)], ["data", "label"])
    busy = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    got = []
    for _ in range(iters):
        for _ in range(6):          # a few ms of work in front of the hand-over on the consumer's stream
            busy.mul_(1.0001)
        got.append(next(it)[0])
    torch.cuda.synchronize()
    for k, (g, (w, wl)) in enumerate(zip(got, want)):
        assert torch.equal(g["data"].cpu(), w), f"iteration {k}"
        assert np.array_equal(g["label"].numpy().reshape(-1), wl.reshape(-1)), f"labels of iteration {k}"


def test_iterator_on_a_side_stream(jpeg_dir):
    """The consumer may iterate under its own stream context: the tensors belong to that stream."""
    from dali_amd.plugin.pytorch import DALIGenericIterator
    root, _ = jpeg_dir
    ref_pipe = _train_pipe(root, 4, depth=2)
    want = [ref_pipe.run()[0].as_tensor().clone().cpu() for _ in range(5)]
    del ref_pipe
    it = DALIGenericIterator([_train_pipe(root, 4, depth=2)], ["data", "label"])
    side = torch.cuda.Stream()
    sums = []
    with torch.cuda.stream(side):
        for k in range(5):
            d = next(it)[0]["data"]
            sums.append((d.float().sum(), want[k].float().sum()))
            assert torch.equal(d.cpu(), want[k])
    side.synchronize()


def test_heavy_augmentation_pipeline_matches_oracle(monkeypatch):
    """configs[2]: warp_affine + gaussian_blur(sigma=3) + color_twist + erase on 512x512 images, random parameters
    from fn.random.uniform / external_source, compared with the oracle chain bit for bit (the blur with its VALU kernel, the
    CPU order of roundings: DALI_AMD_BLUR_MFMA=0; the default matrix-core kernel is held to <= 1 LSB in test_gpu_augment.py)."""
    monkeypatch.setenv("DALI_AMD_BLUR_MFMA", "0")
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(5)
    bs = 6
    imgs = [synth_image(rng, 512, 512) for _ in range(bs)]
    mats = []
    for _ in range(bs):
        t, s = np.deg2rad(rng.uniform(-30, 30)), rng.uniform(0.8, 1.2)
        c, sn = np.cos(t) / s, np.sin(t) / s
        m = np.array([[c, -sn, 0], [sn, c, 0]], np.float32)
        m[0, 2] = 256 - m[0, 0] * 256 - m[0, 1] * 256
        m[1, 2] = 256 - m[1, 0] * 256 - m[1, 1] * 256
        mats.append(m.reshape(6))
    pipe = Pipeline(batch_size=bs, num_threads=2, device_id=0, seed=17, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="images", layout="HWC")
        m = fn.external_source(name="matrix")
        hue = fn.random.uniform(range=[-30.0, 30.0], seed=1)
        sat = fn.random.uniform(range=[0.7, 1.3], seed=2)
        bri = fn.random.uniform(range=[0.8, 1.2], seed=3)
        con = fn.random.uniform(range=[0.8, 1.2], seed=4)
        anchor = fn.random.uniform(range=[0.0, 0.7], shape=[2], seed=5)
        shape = fn.random.uniform(range=[0.1, 0.3], shape=[2], seed=6)
        y = fn.warp_affine(x.gpu(), matrix=m, fill_value=0.0, interp_type=types.INTERP_LINEAR)
        y = fn.gaussian_blur(y, sigma=3.0)
        y = fn.color_twist(y, hue=hue, saturation=sat, brightness=bri, contrast=con)
        y = fn.erase(y, anchor=anchor, shape=shape, normalized=True, fill_value=0.0)
        pipe.set_outputs(y, hue, sat, bri, con, anchor, shape)
    pipe.build()
    pipe.feed_input("images", imgs, layout="HWC")
    pipe.feed_input("matrix", mats)
    out, hue, sat, bri, con, anchor, shape = pipe.run()
    assert pipe.executed_kernels() == ["h2d_copy", "warp_affine", "gaussian_blur", "color_twist+erase"]
    win = O.gaussian_window(3.0)
    for i in range(bs):
        ref = O.warp_affine_u8(imgs[i], mats[i], interp=1, fill=0.0)
        ref = O.gaussian_blur_u8(ref, win)
        mm, off = O.color_twist_matrix(float(hue.at(i)), float(sat.at(i)), 1.0, float(bri.at(i)), float(con.at(i)))
        ref = O.linear_transform_u8(ref, mm, off)
        ref = O.erase_u8(ref, anchor.at(i), shape.at(i), fill=(0.0,), normalized_anchor=True, normalized_shape=True)
        got = out[i].as_cpu()
        assert np.array_equal(got, ref), f"sample {i}: max diff {np.abs(got.astype(int) - ref).max()}"


def test_gpu_tensor_dlpack_zero_copy_and_device_feed():
    """TensorGPU.__dlpack__ is a zero-copy view (same address as the __cuda_array_interface__ view) on the ROCm device;
    a CUDA torch tensor can be fed to external_source through DLPack."""
    import torch
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    x = torch.arange(2 * 5 * 6, dtype=torch.float32, device="cuda").reshape(2, 5, 6)
    pipe = Pipeline(batch_size=2, num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        pipe.set_outputs(fn.external_source(name="x").gpu())
    pipe.feed_input("x", x)
    (out,) = pipe.run()
    t = out[1]
    dev_type, dev_id = t.__dlpack_device__()
    assert dev_id == 0 and dev_type in (2, 10)          # kDLCUDA / kDLROCM
    view = torch.from_dlpack(t)
    assert view.data_ptr() == t.data_ptr() and view.is_cuda
    assert torch.equal(view, x[1])
