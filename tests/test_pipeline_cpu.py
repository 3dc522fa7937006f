"""Host framework tests that need no GPU: schema registry -> fn generation, graph validation, the
readers.file shard / shuffle / padding semantics, random operators vs the oracle, checkpointing, iterator
epoch accounting.  (Reference behaviour: dali/operators/reader/loader/loader.h:78-503, loader.cc:78-87,
dali/python/nvidia/dali/plugin/base_iterator.py.)"""
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    """3 classes, 23 files; file content = global index as 4 little-endian bytes."""
    root = tmp_path_factory.mktemp("ds")
    files = []
    k = 0
    for c, n in (("apple", 9), ("banana", 6), ("cherry", 8)):
        os.makedirs(root / c)
        for i in range(n):
            p = root / c / f"img_{i:03d}.jpg"
            p.write_bytes(np.int32(k).tobytes())
            files.append((str(p), {"apple": 0, "banana": 1, "cherry": 2}[c]))
            k += 1
    (root / "apple" / "notes.txt").write_text("ignored: extension not in the default filters")
    return str(root), files


def _reader_pipe(root, bs, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=bs, num_threads=2, device_id=0, seed=11, prefetch_queue_depth=kw.pop("depth", 2))
    with pipe:
        data, label = fn.readers.file(file_root=root, name="Reader", **kw)
        pipe.set_outputs(data, label)
    pipe.build()
    return pipe


def _run_indices(pipe, iters):
    idx, lab = [], []
    for _ in range(iters):
        d, l = pipe.run()
        idx.append([int(np.frombuffer(d.at(i).tobytes(), np.int32)[0]) for i in range(len(d))])
        lab.append([int(l.at(i)[0]) for i in range(len(l))])
    return np.array(idx), np.array(lab)


def test_fn_namespace_is_generated_from_schemas():
    from dali_amd import fn, ops
    for path in ("readers.file", "decoders.image", "random.coin_flip", "random.uniform", "random_resized_crop",
                 "crop_mirror_normalize", "external_source"):
        obj = fn
        for p in path.split("."):
            obj = getattr(obj, p)
        assert callable(obj)
    assert "file_root" in fn.readers.file.__doc__ and "random_area" in fn.random_resized_crop.__doc__
    assert ops.readers.File.schema_name == "readers__File" and ops.RandomResizedCrop.schema_name == "RandomResizedCrop"
    assert fn._to_snake_case("RandomResizedCrop") == "random_resized_crop"
    assert fn._to_snake_case("CoinFlip") == "coin_flip" and fn._to_snake_case("BBoxPaste") == "bbox_paste"


def test_reader_labels_sorted_dirs_and_sequential_order(dataset):
    root, files = dataset
    pipe = _reader_pipe(root, 5)
    idx, lab = _run_indices(pipe, 5)
    flat = idx.reshape(-1)
    assert list(flat[:23]) == list(range(23))          # alphabetical dirs, then files; .txt ignored
    assert list(flat[23:]) == [0, 1]                   # wraps into the next epoch without padding
    assert [files[i][1] for i in flat] == list(lab.reshape(-1))
    meta = pipe.reader_meta("Reader")
    assert meta == dict(epoch_size=23, epoch_size_padded=23, number_of_shards=1, shard_id=0, pad_last_batch=0,
                        stick_to_shard=0)


def test_reader_shards_partition_the_dataset(dataset):
    root, _ = dataset
    seen = []
    for k in range(4):
        pipe = _reader_pipe(root, 1, shard_id=k, num_shards=4, stick_to_shard=True)
        start, end = 23 * k // 4, 23 * (k + 1) // 4     # loader.cc:78-82
        idx, _ = _run_indices(pipe, end - start + 2)
        flat = list(idx.reshape(-1))
        assert flat[: end - start] == list(range(start, end))
        assert flat[end - start:] == [start, start + 1]  # stick_to_shard: wraps inside its own shard
        seen += flat[: end - start]
    assert sorted(seen) == list(range(23))


def test_reader_rotates_shards_between_epochs(dataset):
    root, _ = dataset
    pipe = _reader_pipe(root, 1, shard_id=1, num_shards=2)
    idx, _ = _run_indices(pipe, 23)
    flat = list(idx.reshape(-1))
    assert flat[:12] == list(range(11, 23))             # epoch 0: shard 1 = [11, 23)
    assert flat[12:] == list(range(0, 11))              # epoch 1: moves on to shard 0


def test_reader_pad_last_batch(dataset):
    root, _ = dataset
    # shard 0 of 2 has 11 samples, shard 1 has 12; with padding both yield ceil(23/2) = 12 per epoch and the last
    # batch is padded by repeating the final sample
    pipe = _reader_pipe(root, 5, shard_id=0, num_shards=2, pad_last_batch=True, stick_to_shard=True)
    idx, _ = _run_indices(pipe, 3)
    flat = list(idx.reshape(-1))
    assert flat[:11] == list(range(11)) and flat[11:15] == [10, 10, 10, 10]
    meta = pipe.reader_meta("Reader")
    assert meta["epoch_size"] == 23 and meta["epoch_size_padded"] == 24 and meta["pad_last_batch"] == 1


def test_reader_random_shuffle_is_a_permutation_and_seeded(dataset):
    root, _ = dataset
    a, _ = _run_indices(_reader_pipe(root, 23, random_shuffle=True, initial_fill=8, seed=5), 2)
    b, _ = _run_indices(_reader_pipe(root, 23, random_shuffle=True, initial_fill=8, seed=5), 2)
    c, _ = _run_indices(_reader_pipe(root, 23, random_shuffle=True, initial_fill=8, seed=6), 2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    for epoch in a:
        assert sorted(epoch) == list(range(23))         # every epoch is a permutation
    assert list(a[0]) != list(range(23))
    # a reservoir of 8 cannot move a sample more than ~8 + position forward: sample k appears after >= k - 8 draws
    pos = {v: i for i, v in enumerate(a[0])}
    assert all(pos[k] >= k - 8 for k in range(23))


def test_reader_file_list_and_files_args(dataset, tmp_path):
    root, files = dataset
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(f"{os.path.relpath(p, root)} {100 + l}" for p, l in files[:4]) + "\n")
    pipe = Pipeline(batch_size=4, num_threads=1, device_id=0)
    with pipe:
        d, l = fn.readers.file(file_root=root, file_list=str(lst))
        pipe.set_outputs(d, l)
    d, l = pipe.run()
    assert [int(l.at(i)[0]) for i in range(4)] == [100, 100, 100, 100]
    pipe = Pipeline(batch_size=3, num_threads=1, device_id=0)
    with pipe:
        d, l = fn.readers.file(files=[files[5][0], files[20][0], files[9][0]], labels=[7, 8, 9])
        pipe.set_outputs(d, l)
    d, l = pipe.run()
    assert [int(np.frombuffer(d.at(i).tobytes(), np.int32)[0]) for i in range(3)] == [5, 20, 9]
    assert [int(l.at(i)[0]) for i in range(3)] == [7, 8, 9]


def test_graph_validation_errors(dataset):
    root, _ = dataset
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=2, num_threads=1, device_id=0)
    with pipe:
        d, l = fn.readers.file(file_root=root)
        with pytest.raises(TypeError, match="unexpected keyword"):
            fn.random.coin_flip(probabilty=0.3)
        with pytest.raises(RuntimeError, match="not available for device \"gpu\""):
            fn.random.coin_flip(device="gpu")          # an operator runs where it is registered: no silent fallback
        with pytest.raises(ValueError, match="expects between"):
            fn.random_resized_crop(size=[8, 8], device="gpu")
        with pytest.raises(ValueError, match="cannot take a GPU input"):
            fn.decoders.image(d.gpu(), device="mixed")
        pipe.set_outputs(d, l)
    with pytest.raises(RuntimeError, match="num_shards needs to be greater than shard_id"):
        p2 = Pipeline(batch_size=2, num_threads=1, device_id=0)
        with p2:
            p2.set_outputs(*fn.readers.file(file_root=root, shard_id=3, num_shards=2))
        p2.build()
    with pytest.raises(RuntimeError, match="required argument"):
        p3 = Pipeline(batch_size=2, num_threads=1, device_id=0)
        with p3:
            d, l = fn.readers.file(file_root=root)
            img = fn.decoders.image(d, device="mixed")
            p3.set_outputs(fn.random_resized_crop(img))   # `size` is required
        p3.build()


def test_random_ops_match_oracle_and_checkpoint_roundtrip():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline, pipeline_def

    @pipeline_def(batch_size=16, num_threads=2, device_id=0, seed=99, prefetch_queue_depth=1)
    def pipe_fn(p):
        return fn.random.coin_flip(probability=p, seed=1234), fn.random.uniform(range=[-2.0, 3.0], seed=77)

    pipe = pipe_fn(0.3)
    # outputs are views of pipeline-owned buffers, valid until the next run(): copy them out right away
    outs = [tuple(o.as_array() for o in pipe.run()) for _ in range(4)]
    for it, (flip, uni) in enumerate(outs):
        assert np.array_equal(flip.reshape(-1), O.coin_flip_batch(1234, it, 16, 0.3))
        u = uni.reshape(-1)
        assert u.dtype == np.float32 and (u >= -2).all() and (u < 3).all()
    # uniform = fma(u32, (nextafter(3,-2) - (-2)) * 2^-32, -2): restated from random_dist.h:175-205
    g = O.Philox(77, 0, 0)
    r = g.next()
    mx = np.nextafter(np.float32(3), np.float32(-2))
    factor = np.float32((mx - np.float32(-2)) * np.float32(2.0 ** -32))
    expect = min(np.float32(np.float64(np.float32(r)) * np.float64(factor) + np.float64(np.float32(-2))), mx)
    assert outs[0][1].reshape(-1)[0] == expect
    # checkpoint after 4 iterations, restore into a fresh pipeline: the streams continue identically
    cpt = pipe.checkpoint()
    nxt = tuple(o.as_array() for o in pipe.run())
    pipe2 = Pipeline(batch_size=16, num_threads=2, device_id=0, seed=99, prefetch_queue_depth=1, checkpoint=cpt)
    with pipe2:
        pipe2.set_outputs(fn.random.coin_flip(probability=0.3, seed=1234), fn.random.uniform(range=[-2.0, 3.0], seed=77))
    res = pipe2.run()
    assert np.array_equal(res[0].as_array(), nxt[0]) and np.array_equal(res[1].as_array(), nxt[1])


def test_pipeline_seed_assigns_distinct_reproducible_op_seeds():
    from dali_amd import fn
    from dali_amd.pipeline import pipeline_def

    @pipeline_def(batch_size=64, num_threads=1, device_id=0)
    def p():
        return fn.random.coin_flip(), fn.random.coin_flip()

    a = [o.as_array() for o in p(seed=5).run()]
    b = [o.as_array() for o in p(seed=5).run()]
    c = [o.as_array() for o in p(seed=6).run()]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], a[1])      # the two operators got different seeds
    assert not np.array_equal(a[0], c[0])


def test_external_source_feed_and_callback():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=3, num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        pipe.set_outputs(x)
    pipe.build()
    with pytest.raises(RuntimeError, match="No data was provided"):
        pipe.run()
    pipe = Pipeline(batch_size=3, num_threads=1, device_id=0, prefetch_queue_depth=1)
    batches = [[np.full((2, i + 1), 10 * k + i, np.int32) for i in range(3)] for k in range(3)]
    with pipe:
        pipe.set_outputs(fn.external_source(source=iter(batches), layout="AB"))
    for k in range(3):
        (out,) = pipe.run()
        assert out.layout() == "AB" and out.shape() == [[2, 1], [2, 2], [2, 3]]
        assert all((out.at(i) == 10 * k + i).all() for i in range(3))


def test_iterator_epochs_with_reader(dataset):
    root, _ = dataset
    from dali_amd.plugin.pytorch import DALIGenericIterator, LastBatchPolicy

    def labels_of(it):
        return [[int(v) for v in b[0]["label"].reshape(-1)] for b in it]

    # FILL without padding: 23 samples, batch 5 -> 5 batches; the tail wraps into the next epoch
    it = DALIGenericIterator([_reader_pipe(root, 5)], ["data", "label"], reader_name="Reader", auto_reset=True)
    assert len(it) == 5
    e0 = labels_of(it)
    assert len(e0) == 5 and all(len(b) == 5 for b in e0)
    e1 = labels_of(it)
    assert len(e1) == 5   # 2 samples were read ahead: 21 left -> still 5 batches
    # PARTIAL: last batch trimmed to the 3 real samples
    it = DALIGenericIterator([_reader_pipe(root, 5, pad_last_batch=True)], ["data", "label"], reader_name="Reader",
                             last_batch_policy=LastBatchPolicy.PARTIAL, auto_reset=True)
    e = labels_of(it)
    assert [len(b) for b in e] == [5, 5, 5, 5, 3]
    assert [len(b) for b in labels_of(it)] == [5, 5, 5, 5, 3]
    # DROP: only full batches
    it = DALIGenericIterator([_reader_pipe(root, 5)], ["data", "label"], reader_name="Reader",
                             last_batch_policy=LastBatchPolicy.DROP, auto_reset=True)
    assert len(it) == 4 and [len(b) for b in labels_of(it)] == [5, 5, 5, 5]
    # two pipelines = two shards, like one per GPU
    pipes = [_reader_pipe(root, 4, shard_id=k, num_shards=2, pad_last_batch=True) for k in range(2)]
    it = DALIGenericIterator(pipes, ["data", "label"], reader_name="Reader", auto_reset=True)
    n = 0
    for batch in it:
        assert len(batch) == 2 and batch[0]["label"].shape[0] == 4
        n += 1
    assert n == 3   # ceil(23/2) = 12 per shard -> 3 batches of 4


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("pad", [False, True])
def test_reader_checkpoint_roundtrip(dataset, shuffle, pad):
    """A restored reader continues with exactly the samples the saved one returns next - across an epoch boundary,
    with samples sitting in the shuffle buffer, into a fresh pipeline and into one that has already run
    (reference: loader.h:279,335,485-503 checkpointing of the loader position + rng)."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    root, files = dataset
    kw = dict(random_shuffle=shuffle, initial_fill=7, pad_last_batch=pad, num_shards=2, shard_id=1)

    def make(cpt=None):
        pipe = Pipeline(batch_size=5, num_threads=2, device_id=0, seed=11, prefetch_queue_depth=1, checkpoint=cpt)
        with pipe:
            data, label = fn.readers.file(file_root=root, name="Reader", **kw)
            pipe.set_outputs(data, label)
        pipe.build()
        return pipe

    pipe = make()
    _run_indices(pipe, 3)                      # 15 samples: past the first epoch of a 12-sample shard
    cpt = pipe.checkpoint()
    expect, expect_lab = _run_indices(pipe, 6)
    fresh = make(cpt)
    got, got_lab = _run_indices(fresh, 6)
    assert np.array_equal(got, expect) and np.array_equal(got_lab, expect_lab)
    # restoring into a reader that has already run ahead replays the same stream again
    used = make()
    _run_indices(used, 5)
    used._backend.restore(cpt)
    again, _ = _run_indices(used, 6)
    assert np.array_equal(again, expect)


def test_dlpack_export_and_external_source_import():
    """TensorCPU.__dlpack__ / __dlpack_device__ (backend_impl.cc:623-740) and DLPack producers as external_source input
    (external_source.py:95-130): a torch tensor feeds the pipeline, the outputs come back through np.from_dlpack."""
    import torch
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    batch = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    pipe = Pipeline(batch_size=2, num_threads=1, device_id=None, prefetch_queue_depth=1)
    with pipe:
        pipe.set_outputs(fn.external_source(name="x"))
    pipe.feed_input("x", batch)                       # one DLPack tensor = the batch
    (out,) = pipe.run()
    assert out.at(0).shape == (3, 4)
    t0 = out[0]
    assert t0.__dlpack_device__() == (1, 0)
    assert np.array_equal(np.from_dlpack(t0), batch[0].numpy())
    assert torch.equal(torch.from_dlpack(out[1]), batch[1])
    pipe.feed_input("x", [batch[1], batch[0]])        # a list of DLPack tensors = the samples
    (out,) = pipe.run()
    assert np.array_equal(out.at(0), batch[1].numpy())


def _wav(x, sr):
    import io, struct
    pcm = np.round(np.clip(x, -1, 1) * 32767).astype("<i2")
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    b = io.BytesIO()
    b.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVE")
    b.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * 2 * ch, 2 * ch, 16))
    b.write(b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes())
    return b.getvalue()


def test_audio_decoder_resamples_and_cpu_audio_resample_match_oracle():
    """decoders.audio(sample_rate=...) = decode (+ downmix) then the windowed-sinc resampler
    (audio_decoder_impl.cc:38-120); fn.audio_resample on the CPU backend runs the same host kernel."""
    from oracle import audio as A
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(7)
    t = np.arange(6000) / 22050.0
    mono = 0.4 * np.sin(2 * np.pi * 330 * t) + rng.normal(0, 0.01, t.size)
    stereo = np.stack([mono, 0.5 * mono[::-1]], 1)
    wavs = [_wav(mono, 22050), _wav(stereo, 22050)]
    for downmix in (False, True):
        pipe = Pipeline(batch_size=2, num_threads=2, device_id=None, prefetch_queue_depth=1)
        with pipe:
            enc = fn.external_source(name="wav")
            audio, rate = fn.decoders.audio(enc, sample_rate=16000.0, quality=50.0, downmix=downmix)
            plain, _ = fn.decoders.audio(enc, downmix=downmix)
            again = fn.audio_resample(plain, in_rate=22050.0, out_rate=16000.0)
            pipe.set_outputs(audio, rate, plain, again)
        pipe.feed_input("wav", wavs)
        audio, rate, plain, again = pipe.run()
        for i in range(2):
            src = plain.at(i)
            ref = A.audio_resample(src, 22050.0, 16000.0, 50.0)
            got = audio.at(i)
            assert got.shape == ref.shape and float(rate.at(i)) == 16000.0, (got.shape, ref.shape)
            assert np.abs(got - ref).max() <= 1e-4 and np.abs(got - ref).mean() <= 1e-6
            assert np.array_equal(again.at(i), got), "audio_resample(cpu) must equal the decoder's resampling"
        assert (audio.at(1).ndim == 1) == downmix


@pytest.mark.parametrize("in_t,out_t", [(np.int16, None), (np.int16, np.float32), (np.float32, np.int16), (np.uint8, np.int16),
                                        (np.int16, np.uint16), (np.float32, np.uint8), (np.int32, np.int32),
                                        (np.uint32, np.int8), (np.int8, np.uint32)])
def test_cpu_audio_resample_integer_sample_types(in_t, out_t):
    """Integer samples are normalised to floats around the float resampler and converted back with saturation
    (dali/operators/audio/resample.cc:142-192, convert.h:262-350): against the oracle's restatement.  The float
    filter sum differs from the oracle's float64 sum in the last bits, which can move an integer result by one."""
    from oracle import audio as A
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(5)
    t = np.arange(3000) / 16000.0
    sig = 0.8 * np.sin(2 * np.pi * 440 * t) + rng.normal(0, 0.05, t.size)
    sig2 = np.stack([sig, -0.5 * sig], 1)

    def as_type(x):
        if in_t == np.float32:
            return (x * 1.3).astype(np.float32)                       # beyond [-1, 1]: the integer results saturate
        info = np.iinfo(in_t)
        if info.min < 0:
            return np.clip(np.round(x * info.max), info.min, info.max).astype(in_t)
        return np.clip(np.round((x * 0.5 + 0.5) * info.max), 0, info.max).astype(in_t)

    samples = [as_type(sig2), as_type(sig2[::-1].copy())] if out_t is not None else [as_type(sig), as_type(-sig)]
    to_dali = {np.int8: types.INT8, np.uint8: types.UINT8, np.int16: types.INT16, np.uint16: types.UINT16,
               np.int32: types.INT32, np.uint32: types.UINT32, np.float32: types.FLOAT}
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=None, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        kw = {} if out_t is None else {"dtype": to_dali[out_t]}
        pipe.set_outputs(fn.audio_resample(x, in_rate=16000.0, out_rate=11025.0, **kw))
    pipe.feed_input("x", samples)
    (out,) = pipe.run()
    want_t = np.dtype(in_t if out_t is None else out_t)
    for i, s in enumerate(samples):
        ref = A.audio_resample_typed(s, 16000.0, 11025.0, out_dtype=want_t)
        got = out.at(i)
        assert got.dtype == want_t and got.shape == ref.shape, (got.dtype, got.shape, ref.shape)
        if want_t == np.float32:
            assert np.abs(got - ref).max() <= 1e-4
        else:
            d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
            full = float(np.iinfo(want_t).max)
            assert d.max() <= max(1.0, 2e-4 * full), (d.max(), full)    # the reference's own cpu-vs-gpu bound, scaled
            assert (d > max(1.0, 2e-6 * full)).mean() < 0.01
            assert got.min() >= np.iinfo(want_t).min and got.max() <= np.iinfo(want_t).max


@pytest.mark.parametrize("dtype", ["INT16", "INT32", "FLOAT"])
@pytest.mark.parametrize("downmix", [False, True])
@pytest.mark.parametrize("resample", [False, True])
def test_audio_decoder_output_types(dtype, downmix, resample):
    """decoders.audio(dtype=...): DecodeAudio<T> of the reference (audio_decoder_impl.cc:49-120) - raw frames when nothing
    else happens, otherwise float frames, equal-weight downmix, resampling, ConvertSatNorm at the end."""
    from oracle import audio as A
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(9)
    t = np.arange(5000) / 22050.0
    a = 0.9 * np.sin(2 * np.pi * 220 * t) + rng.normal(0, 0.02, t.size)
    frames = [np.stack([a, 0.7 * a[::-1], -0.4 * a], 1), np.stack([a[:3000], -a[:3000], 0.1 * a[:3000]], 1)]
    pcm = [np.clip(np.round(f * 32767), -32768, 32767).astype(np.int16) for f in frames]
    wavs = [_wav(p.astype(np.float64) / 32767.0, 22050) for p in pcm]
    kw = dict(dtype=getattr(types, dtype), downmix=downmix)
    if resample:
        kw["sample_rate"] = 16000.0
    pipe = Pipeline(batch_size=2, num_threads=2, device_id=None, prefetch_queue_depth=1)
    with pipe:
        enc = fn.external_source(name="wav")
        raw, _ = fn.decoders.audio(enc, dtype=types.INT16)
        audio, rate = fn.decoders.audio(enc, **kw)
        pipe.set_outputs(raw, audio, rate)
    pipe.feed_input("wav", wavs)
    raw, audio, rate = pipe.run()
    np_t = {"INT16": np.int16, "INT32": np.int32, "FLOAT": np.float32}[dtype]
    for i in range(2):
        stored = raw.at(i)                      # the file's own PCM16 frames
        assert stored.dtype == np.int16 and stored.shape == pcm[i].shape
        ref = A.decode_audio(stored, 22050, np_t, downmix=downmix, sample_rate=16000.0 if resample else None)
        got = audio.at(i)
        assert got.dtype == np_t and got.shape == ref.shape, (got.dtype, got.shape, ref.shape)
        assert float(rate.at(i)) == (16000.0 if resample else 22050.0)
        if not resample:
            assert np.array_equal(got, ref)     # decode, shift and the weighted downmix are exact restatements
        elif np_t == np.float32:
            assert np.abs(got - ref).max() <= 1e-4
        else:
            full = float(np.iinfo(np_t).max)
            assert np.abs(got.astype(np.int64) - ref.astype(np.int64)).max() <= max(1.0, 2e-4 * full)


@pytest.mark.parametrize("kw", [dict(), dict(nfft=512, wl=400, step=160, nfilter=64, formula="htk"),
                                dict(power=1, center=False, nfilter=128), dict(reflect=False, normalize=False, nfilter=23)])
def test_audio_feature_operators_on_the_cpu_backend(kw):
    """configs[3] on the CPU backend: decoders.audio -> spectrogram -> mel_filter_bank -> to_decibels (+ mfcc) on the host
    kernels against the oracle: the spectrogram within 1e-4 of the largest bin (the reference's own FFT tolerance,
    test_spectrogram.py:188) and tighter on the loud bins, the mel product exactly (same order of float operations) on
    the operator's own spectrogram, decibels and MFCC to float rounding."""
    from oracle import audio as A
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(21)
    sigs = []
    for secs in (0.4, 1.1, 0.1):
        t = np.arange(int(16000 * secs)) / 16000.0
        sigs.append((0.5 * np.sin(2 * np.pi * (300 + 900 * t) * t) + rng.normal(0, 0.05, t.size)).astype(np.float64))
    wavs = [_wav(s, 16000) for s in sigs]
    pipe = Pipeline(batch_size=len(wavs), num_threads=3, device_id=None, prefetch_queue_depth=1)
    with pipe:
        enc = fn.external_source(name="wav")
        audio, _ = fn.decoders.audio(enc, downmix=True)
        spec = fn.spectrogram(audio, nfft=kw.get("nfft", 1024), window_length=kw.get("wl", 1024), window_step=kw.get("step", 256),
                              power=kw.get("power", 2), center_windows=kw.get("center", True), reflect_padding=kw.get("reflect", True))
        mel = fn.mel_filter_bank(spec, nfilter=kw.get("nfilter", 80), sample_rate=16000.0, freq_high=8000.0,
                                 mel_formula=kw.get("formula", "slaney"), normalize=kw.get("normalize", True))
        db = fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0)
        mfcc = fn.mfcc(db, n_mfcc=13, lifter=22.0)
        pipe.set_outputs(audio, spec, mel, db, mfcc)
    pipe.feed_input("wav", wavs)
    audio, spec, mel, db, mfcc = pipe.run()
    assert pipe.executed_kernels() == ["host_spectrogram", "host_mel_filter_bank", "host_to_decibels", "host_mfcc_dct"]
    for i in range(len(wavs)):
        x = audio.at(i)
        ref_spec = A.spectrogram(x, nfft=kw.get("nfft", 1024), window_length=kw.get("wl", 1024), window_step=kw.get("step", 256),
                                 power=kw.get("power", 2), center_windows=kw.get("center", True),
                                 reflect_padding=kw.get("reflect", True))
        got_spec = spec.at(i)
        assert got_spec.shape == ref_spec.shape, (got_spec.shape, ref_spec.shape)
        assert np.abs(got_spec - ref_spec).max() <= 1e-6 * max(1.0, ref_spec.max())
        ref_mel = A.mel_filter_bank(got_spec, nfilter=kw.get("nfilter", 80), sample_rate=16000.0, freq_high=8000.0,
                                    mel_formula=kw.get("formula", "slaney"), normalize=kw.get("normalize", True))
        assert np.array_equal(mel.at(i), ref_mel)
        ref_db = A.to_decibels(mel.at(i), multiplier=10.0, cutoff_db=-80.0)
        assert np.abs(db.at(i) - ref_db).max() <= 1e-4
        ref_mfcc = A.mfcc(db.at(i), n_mfcc=13, lifter=22.0)
        assert mfcc.at(i).shape == ref_mfcc.shape
        assert np.abs(mfcc.at(i) - ref_mfcc).max() <= 1e-5 * max(1.0, np.abs(ref_mfcc).max())


@pytest.mark.parametrize("interp,fill", [("INTERP_LINEAR", 0.0), ("INTERP_NN", 17.0), ("INTERP_LINEAR", None)])
def test_heavy_augmentation_operators_on_the_cpu_backend(interp, fill):
    """configs[2] on the CPU backend: warp_affine + gaussian_blur + color_twist + erase through the host kernels, compared
    with the oracle chain bit for bit (constant border, clamp border, nearest and bilinear sampling, random parameters)."""
    from oracle import oracle as O
    from tests.util import synth_image
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(15)
    bs = 4
    imgs = [synth_image(rng, h, w) for (h, w) in [(200, 300), (257, 190), (64, 520), (128, 128)]]
    mats = []
    for im in imgs:
        t, s = np.deg2rad(rng.uniform(-30, 30)), rng.uniform(0.8, 1.2)
        c, sn = np.cos(t) / s, np.sin(t) / s
        cx, cy = im.shape[1] / 2, im.shape[0] / 2
        m = np.array([[c, -sn, 0], [sn, c, 0]], np.float32)
        m[0, 2] = cx - m[0, 0] * cx - m[0, 1] * cy
        m[1, 2] = cy - m[1, 0] * cx - m[1, 1] * cy
        mats.append(m.reshape(6))
    pipe = Pipeline(batch_size=bs, num_threads=3, device_id=None, seed=17, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="images", layout="HWC")
        m = fn.external_source(name="matrix")
        hue = fn.random.uniform(range=[-30.0, 30.0], seed=1)
        sat = fn.random.uniform(range=[0.7, 1.3], seed=2)
        bri = fn.random.uniform(range=[0.8, 1.2], seed=3)
        con = fn.random.uniform(range=[0.8, 1.2], seed=4)
        anchor = fn.random.uniform(range=[0.0, 0.7], shape=[2], seed=5)
        shape = fn.random.uniform(range=[0.1, 0.3], shape=[2], seed=6)
        kw = {} if fill is None else {"fill_value": fill}
        y = fn.warp_affine(x, matrix=m, interp_type=getattr(types, interp), **kw)
        y = fn.gaussian_blur(y, sigma=2.0)
        y = fn.color_twist(y, hue=hue, saturation=sat, brightness=bri, contrast=con)
        y = fn.erase(y, anchor=anchor, shape=shape, normalized=True, fill_value=3.0)
        pipe.set_outputs(y, hue, sat, bri, con, anchor, shape)
    pipe.build()
    pipe.feed_input("images", imgs, layout="HWC")
    pipe.feed_input("matrix", mats)
    out, hue, sat, bri, con, anchor, shape = pipe.run()
    assert pipe.executed_kernels() == ["host_warp_affine", "host_gaussian_blur", "host_color_twist", "host_erase"]
    win = O.gaussian_window(2.0)
    for i in range(bs):
        ref = O.warp_affine_u8(imgs[i], mats[i], interp=1 if interp == "INTERP_LINEAR" else 0, fill=fill)
        ref = O.gaussian_blur_u8(ref, win)
        mm, off = O.color_twist_matrix(float(hue.at(i)), float(sat.at(i)), 1.0, float(bri.at(i)), float(con.at(i)))
        ref = O.linear_transform_u8(ref, mm, off)
        ref = O.erase_u8(ref, anchor.at(i), shape.at(i), fill=(3.0,), normalized_anchor=True, normalized_shape=True)
        got = out.at(i)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"sample {i}: max diff {np.abs(got.astype(int) - ref).max()}"


@pytest.mark.parametrize("device", ["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def test_rotate_and_the_colour_twist_siblings(device):
    """fn.rotate (rotate.cc:19-44: canvas inferred / keep_size / explicit size, per-sample angles) and fn.hsv / fn.hue /
    fn.saturation (color_twist.cc:26-103,142-144: the ColorTwist class under schemas with fewer arguments) against the oracle
    bit for bit, on the host kernels and on the device kernels."""
    from oracle import oracle as O
    from tests.util import synth_image
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(23)
    imgs = [synth_image(rng, h, w) for (h, w) in [(120, 161), (97, 64), (50, 50), (200, 131)]]
    angles = [np.float32(a) for a in (10.0, -33.3, 90.0, 217.5)]
    pipe = Pipeline(batch_size=4, num_threads=3, device_id=None if device == "cpu" else 0, seed=3, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="images", layout="HWC")
        a = fn.external_source(name="angle")
        xd = x if device == "cpu" else x.gpu()
        r0 = fn.rotate(xd, angle=a, fill_value=42.0)
        r1 = fn.rotate(xd, angle=a, keep_size=True)
        r2 = fn.rotate(xd, angle=25.0, size=[70, 33], interp_type=types.INTERP_NN, fill_value=0.0)
        h0 = fn.hsv(xd, hue=120.0, saturation=0.5, value=1.25)
        h1 = fn.hue(xd, hue=-45.0)
        h2 = fn.saturation(xd, saturation=1.7)
        pipe.set_outputs(r0, r1, r2, h0, h1, h2)
    pipe.build()
    pipe.feed_input("images", imgs, layout="HWC")
    pipe.feed_input("angle", angles)
    outs = pipe.run()
    get = (lambda tl, i: tl.at(i)) if device == "cpu" else (lambda tl, i: tl[i].as_cpu())
    for i, im in enumerate(imgs):
        assert np.array_equal(get(outs[0], i), O.rotate_u8(im, float(angles[i]), fill=42.0)), i
        assert np.array_equal(get(outs[1], i), O.rotate_u8(im, float(angles[i]), keep_size=True)), i
        assert np.array_equal(get(outs[2], i), O.rotate_u8(im, 25.0, size=(70, 33), interp=0, fill=0.0)), i
        for k, (hue, sat, val) in enumerate([(120.0, 0.5, 1.25), (-45.0, 1.0, 1.0), (0.0, 1.7, 1.0)]):
            m, off = O.color_twist_matrix(hue, sat, val, 1.0, 1.0)
            assert np.array_equal(get(outs[3 + k], i), O.linear_transform_u8(im, m, off)), (i, k)
    with pytest.raises(TypeError, match="unexpected keyword"):
        with Pipeline(batch_size=1, num_threads=1, device_id=None):
            fn.hue(fn.external_source(name="q"), saturation=2.0)
