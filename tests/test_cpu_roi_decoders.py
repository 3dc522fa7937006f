"""decoders.image_crop / image_random_crop / image_slice with device="cpu": the window arithmetic of the mixed operators
(the same code) around the host decoder - "decode, then crop" of the oracle, bit for bit
(dali/operators/imgcodec/roi_image_decoder.h:47-96, dali/test/python/decoder/test_image.py:118-216)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import encode_jpeg, synth_image


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    root = tmp_path_factory.mktemp("roi_cpu")
    rng = np.random.default_rng(78)
    out = []
    specs = [((120, 160), dict(subsampling="4:2:0")), ((200, 150), dict(subsampling="4:4:4")),
             ((97, 131), dict(subsampling="4:2:2")), ((240, 320), dict(subsampling="4:2:0", progressive=True)),
             ((64, 48), dict(subsampling="4:1:1")), ((75, 211), dict(subsampling="4:2:0", optimize=True))]
    for i, (hw, kw) in enumerate(specs):
        p = root / f"img{i}.jpg"
        p.write_bytes(encode_jpeg(synth_image(rng, *hw), 85, **kw))
        out.append(str(p))
    return out


def _decoded(files):
    return [O.jpeg_decode_rgb(open(f, "rb").read()) for f in files]


def _llround(v):
    return int(np.floor(v + 0.5)) if v >= 0 else -int(np.floor(-v + 0.5))


def test_image_random_crop_cpu_equals_decode_then_crop_and_checkpoints(files):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)

    def make(checkpoint=None):
        pipe = Pipeline(batch_size=bs, num_threads=3, device_id=None, prefetch_queue_depth=1, enable_checkpointing=True,
                        checkpoint=checkpoint)
        with pipe:
            enc, _ = fn.readers.file(files=files)
            pipe.set_outputs(fn.decoders.image_random_crop(enc, seed=1234, random_area=[0.1, 0.9]))
        return pipe

    pipe = make()
    ref = _decoded(files)
    for it in range(3):
        (out,) = pipe.run()
        assert pipe.executed_kernels() == ["host_jpeg_decode_roi"]
        anchors, crops = O.rrc_batch(1234, it, [r.shape[:2] for r in ref], area=(0.1, 0.9))
        for i in range(bs):
            (y0, x0), (h, w) = anchors[i], crops[i]
            got = out.at(i)
            assert got.shape == (h, w, 3), (it, i)
            assert np.array_equal(got, ref[i][y0:y0 + h, x0:x0 + w]), (it, i)
    # the generator state travels with the checkpoint like the mixed operator's
    state = pipe.checkpoint()
    (nxt,) = pipe.run()
    (rep,) = make(checkpoint=state).run()
    for i in range(bs):
        assert np.array_equal(rep.at(i), nxt.at(i))

def test_image_crop_cpu_fixed_window_and_per_sample_anchor(files):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    pos_x = np.linspace(0, 1, bs).astype(np.float32)
    pipe = Pipeline(batch_size=bs, num_threads=2, device_id=None)
    with pipe:
        enc, _ = fn.readers.file(files=files)
        px = fn.external_source(source=lambda: [np.array(v, np.float32) for v in pos_x], batch=True)
        pipe.set_outputs(fn.decoders.image_crop(enc, crop=(40, 33), crop_pos_x=px, crop_pos_y=0.3))
    (out,) = pipe.run()
    for i, r in enumerate(_decoded(files)):
        y0 = O.crop_anchor(0.3, 40, r.shape[0])
        x0 = O.crop_anchor(float(pos_x[i]), 33, r.shape[1])
        assert np.array_equal(out.at(i), r[y0:y0 + 40, x0:x0 + 33]), i
    bad = Pipeline(batch_size=bs, num_threads=2, device_id=None)
    with bad:
        enc, _ = fn.readers.file(files=files)
        bad.set_outputs(fn.decoders.image_crop(enc, crop=(1000, 10)))
    with pytest.raises(RuntimeError, match="out of the bounds"):
        bad.run()


@pytest.mark.parametrize("case", ["rel_start_rel_shape", "start_end", "positional_absolute_int"])
def test_image_slice_cpu_equals_decode_then_slice(files, case):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    bs = len(files)
    ref = _decoded(files)
    rng = np.random.default_rng(5)
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=None, prefetch_queue_depth=1)
    expect = []
    with pipe:
        enc, _ = fn.readers.file(files=files)
        if case == "rel_start_rel_shape":     # default axis order "WH"
            out = fn.decoders.image_slice(enc, rel_start=[0.25, 0.1], rel_shape=[0.5, 0.6])
            for r in ref:
                H, W = r.shape[:2]
                x0, x1 = _llround(0.25 * W), _llround((0.25 + 0.5) * W)
                y0, y1 = _llround(np.float32(0.1) * H), _llround((np.float64(np.float32(0.1)) + np.float64(np.float32(0.6))) * H)
                expect.append(r[y0:y1, x0:x1])
        elif case == "start_end":
            out = fn.decoders.image_slice(enc, start=[10, 5], end=[40, 47])
            expect = [r[5:47, 10:40] for r in ref]
        else:
            anchors = rng.integers(0, 20, (bs, 2)).astype(np.int32)
            shapes = rng.integers(8, 28, (bs, 2)).astype(np.int32)
            a = fn.external_source(name="a")
            s = fn.external_source(name="s")
            out = fn.decoders.image_slice(enc, a, s, axis_names="HW")
            expect = [r[an[0]:an[0] + sh[0], an[1]:an[1] + sh[1]] for r, an, sh in zip(ref, anchors, shapes)]
        pipe.set_outputs(out)
    pipe.build()
    if case.startswith("positional"):
        pipe.feed_input("a", anchors)
        pipe.feed_input("s", shapes)
    (res,) = pipe.run()
    for i in range(bs):
        got = res.at(i)
        assert got.shape == expect[i].shape, (case, i, got.shape, expect[i].shape)
        assert np.array_equal(got, expect[i]), (case, i)
