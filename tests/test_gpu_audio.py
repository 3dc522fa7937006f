"""GPU parity for the audio path (BASELINE.json configs[3]): decoders.audio -> spectrogram(nfft=1024) ->
mel_filter_bank(80) -> to_decibels, against the numpy oracle (float64 FFT).

Stated tolerances (the reference's own, dali/test/python/operator_2/test_spectrogram.py:188,
operator_1/test_mel_filter_bank.py:199, operator_2/test_to_decibels.py:120):
  spectrogram  |got - ref| <= 1e-4 * max(ref) + 1e-6        (f32 radix-4 FFT vs f64 FFT)
  mel          |got - ref| <= 1e-3 relative to the row scale  (we hold 1e-5: fma chain vs mul+add)
  decibels     |got - ref| <= 1e-4 * |ref| + 1e-3 dB
"""
import io
import os
import struct

import numpy as np
import pytest

from oracle import audio as A

pytestmark = pytest.mark.gpu


def synth_signal(rng, seconds, sr=16000):
    """band-limited noise + 3 chirps."""
    n = int(seconds * sr)
    t = np.arange(n) / sr
    x = rng.normal(0, 0.05, n)
    x = np.convolve(x, np.ones(8) / 8, mode="same")
    for _ in range(3):
        f0, f1 = rng.uniform(100, 3000), rng.uniform(200, 7000)
        x += 0.2 * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / seconds))
    return np.clip(x, -0.99, 0.99)


def to_wav(x, sr=16000):
    pcm = np.round(x * 32767).astype("<i2")
    b = io.BytesIO()
    b.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVE")
    b.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16))
    b.write(b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes())
    return b.getvalue()


def _audio_pipe(bs, **kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=bs, num_threads=4, device_id=0, prefetch_queue_depth=1)
    with pipe:
        enc = fn.external_source(name="wav")
        audio, rate = fn.decoders.audio(enc, downmix=True)
        spec = fn.spectrogram(audio.gpu(), nfft=kw.get("nfft", 1024), window_length=kw.get("wl", 1024),
                              window_step=kw.get("step", 256), power=kw.get("power", 2),
                              center_windows=kw.get("center", True), reflect_padding=kw.get("reflect", True))
        mel = fn.mel_filter_bank(spec, nfilter=kw.get("nfilter", 80), sample_rate=16000.0, freq_high=8000.0,
                                 mel_formula=kw.get("formula", "slaney"), normalize=kw.get("normalize", True))
        db = fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0)
        pipe.set_outputs(audio, rate, spec, mel, db)
    pipe.build()
    return pipe


@pytest.mark.parametrize("kw", [dict(), dict(nfft=512, wl=400, step=160, nfilter=64, formula="htk"),
                                dict(power=1, center=False, nfilter=128), dict(reflect=False, normalize=False, nfilter=23)])
def test_audio_pipeline_matches_oracle(kw):
    rng = np.random.default_rng(11)
    sigs = [synth_signal(rng, s) for s in (1.3, 2.0, 0.7, 3.1)]
    wavs = [to_wav(s) for s in sigs]
    pipe = _audio_pipe(len(wavs), **kw)
    pipe.feed_input("wav", wavs)
    audio, rate, spec, mel, db = pipe.run()
    assert pipe.executed_kernels() == ["h2d_copy", "spectrogram", "mel_filter_bank_banded", "to_decibels"]
    for i, w in enumerate(wavs):
        ref_audio, sr = A.decode_wav(w)
        assert np.array_equal(audio.at(i), ref_audio) and float(rate.at(i)) == sr == 16000.0
        ref_spec = A.spectrogram(ref_audio, nfft=kw.get("nfft", 1024), window_length=kw.get("wl", 1024),
                                 window_step=kw.get("step", 256), power=kw.get("power", 2),
                                 center_windows=kw.get("center", True), reflect_padding=kw.get("reflect", True))
        got_spec = spec[i].as_cpu()
        assert got_spec.shape == ref_spec.shape
        err = np.abs(got_spec - ref_spec).max()
        assert err <= 1e-4 * ref_spec.max() + 1e-6, f"spectrogram sample {i}: {err} vs max {ref_spec.max()}"
        # mel / dB are checked on the GPU's own spectrogram so each stage has its own tolerance
        ref_mel = A.mel_filter_bank(got_spec, kw.get("nfilter", 80), 16000.0, 0.0, 8000.0, kw.get("normalize", True),
                                    kw.get("formula", "slaney"))
        got_mel = mel[i].as_cpu()
        scale = np.abs(ref_mel).max(axis=1, keepdims=True) + 1e-12
        assert (np.abs(got_mel - ref_mel) / scale).max() <= 1e-5, f"mel sample {i}"
        ref_db = A.to_decibels(got_mel, 10.0, 0.0, -80.0)
        got_db = db[i].as_cpu()
        assert np.abs(got_db - ref_db).max() <= 1e-3, f"dB sample {i}"
        assert got_db.max() <= 1e-4 and got_db.min() >= -80.0 - 1e-3


def _pcm_chain(bs, nfft, wl, step, center, reflect, dtype=None):
    """decoders.audio -> gpu -> spectrogram -> mel -> dB with the decoded audio used by the spectrogram ONLY."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=bs, num_threads=4, device_id=0, prefetch_queue_depth=1)
    with pipe:
        enc = fn.external_source(name="wav")
        audio, rate = fn.decoders.audio(enc, downmix=True, **({"dtype": dtype} if dtype is not None else {}))
        spec = fn.spectrogram(audio.gpu(), nfft=nfft, window_length=wl, window_step=step, center_windows=center,
                              reflect_padding=reflect)
        mel = fn.mel_filter_bank(spec, nfilter=80, sample_rate=16000.0, freq_high=8000.0)
        pipe.set_outputs(fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0), rate)
    pipe.build()
    return pipe


@pytest.mark.parametrize("nfft,wl,step,center,reflect", [(1024, 1024, 256, True, True), (512, 400, 160, True, False),
                                                         (1024, 800, 200, False, True), (256, 256, 64, True, True),
                                                         (1024, 1024, 255, True, True)])
def test_pcm16_crosses_the_bus_as_int16_and_gives_the_same_bits(nfft, wl, step, center, reflect, monkeypatch):
    """Round 4 graph-level fusion: when the decoded audio feeds nothing but the copy in front of a gpu Spectrogram, a
    batch of mono 16-bit streams travels as int16 and the spectrogram kernel's load divides by 32768.  The results must be
    the bits of the unfused graph (DALI_AMD_NO_PCM16_FUSION=1): odd lengths (unaligned sample pairs), odd hops, every
    padding mode, the generic kernel (nfft 256) and the register-resident one; a batch with a stereo file falls back."""
    rng = np.random.default_rng(3)
    sigs = [synth_signal(rng, s) for s in (1.3, 0.40006, 2.1, 0.9)]          # (6401 samples: an odd length)
    wavs = [to_wav(s) for s in sigs]
    outs = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("DALI_AMD_NO_PCM16_FUSION", "1")
        pipe = _pcm_chain(len(wavs), nfft, wl, step, center, reflect)
        res = []
        for _ in range(3):                                                    # (ring slots are reused)
            pipe.feed_input("wav", wavs)
            db, rate = pipe.run()
            res.append([db[i].as_cpu().copy() for i in range(len(wavs))])
            assert [float(rate.at(i)) for i in range(len(wavs))] == [16000.0] * len(wavs)
        outs[fused] = res
    for a, b in zip(outs[True], outs[False]):
        for i, (x, y) in enumerate(zip(a, b)):
            assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"sample {i}"


def test_pcm16_fusion_with_flac_streams_in_the_batch(monkeypatch):
    """Mono 16-bit FLAC next to WAV: the batch still travels as int16 (the FLAC samples are decoded to int16 on the host, so
    the output is a copy, not a view of the files); same bits as the float path."""
    here = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(8)
    batch = [open(os.path.join(here, "golden", "flac", "mono16_fixed.flac"), "rb").read(), to_wav(synth_signal(rng, 0.9)),
             open(os.path.join(here, "golden", "flac", "nolength.flac"), "rb").read()]
    outs = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("DALI_AMD_NO_PCM16_FUSION", "1")
        pipe = _pcm_chain(len(batch), 1024, 1024, 256, True, True)
        pipe.feed_input("wav", batch)
        db, _ = pipe.run()
        outs.append([db[i].as_cpu().copy() for i in range(len(batch))])
    for i, (x, y) in enumerate(zip(*outs)):
        assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"sample {i}"


def test_pcm16_fusion_falls_back_when_a_stream_needs_the_host_path(monkeypatch):
    """A stereo file (downmixed on the host in float), a float32 WAV or a resampled stream in the batch: the whole batch is
    decoded to float as before."""
    rng = np.random.default_rng(4)
    a, b = synth_signal(rng, 0.8), synth_signal(rng, 0.8)
    pcm = np.round(np.stack([a, b], axis=1) * 32767).astype("<i2")
    buf = io.BytesIO()
    buf.write(b"RIFF" + struct.pack("<I", 36 + pcm.nbytes) + b"WAVE")
    buf.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 16000, 16000 * 4, 4, 16))
    buf.write(b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes())
    wavs = [to_wav(a), buf.getvalue()]
    pipe = _pcm_chain(2, 1024, 1024, 256, True, True)
    pipe.feed_input("wav", wavs)
    db, _ = pipe.run()
    monkeypatch.setenv("DALI_AMD_NO_PCM16_FUSION", "1")
    ref = _pcm_chain(2, 1024, 1024, 256, True, True)
    ref.feed_input("wav", wavs)
    rdb, _ = ref.run()
    for i in range(2):
        assert np.array_equal(db[i].as_cpu(), rdb[i].as_cpu())


def test_mel_weights_match_oracle_and_reference_properties():
    import ctypes as C
    from dali_amd import _capi as capi
    lib = capi.kernels()
    for nf, nfft, sr, lo, hi, norm, formula in [(80, 1024, 16000.0, 0.0, 8000.0, True, "slaney"),
                                               (128, 2048, 44100.0, 0.0, 0.0, True, "slaney"),
                                               (40, 512, 16000.0, 300.0, 7000.0, False, "htk")]:
        W = np.zeros((nf, nfft // 2 + 1), np.float32)
        rc = lib.daliamdMelFilterBankWeights(nf, nfft, C.c_float(sr), C.c_float(lo), C.c_float(hi), int(norm),
                                             1 if formula == "htk" else 0, W.ctypes.data_as(C.c_void_p))
        assert rc == 0
        ref = A.mel_weights(nf, nfft, sr, lo, hi, norm, formula)
        assert np.array_equal(W, ref)
        # triangular filters: every row is non-negative, unimodal, and neighbours overlap
        assert (W >= 0).all() and (W.sum(1) > 0).all()
        peaks = W.argmax(1)
        assert (np.diff(peaks) > 0).all()
        if not norm:   # un-normalised triangles sum to 1 across filters inside the covered band
            inside = slice(peaks[0] + 1, peaks[-1])
            assert np.allclose(W[:, inside].sum(0), 1.0, atol=1e-5)


def test_to_decibels_reference_and_silence():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(3)
    data = [np.abs(rng.normal(0, 1, (7, 33))).astype(np.float32), np.zeros((4, 5), np.float32),
            (rng.uniform(0, 1e-12, (3, 3))).astype(np.float32)]
    for ref, cutoff, mult in [(None, -200.0, 10.0), (2.5, -60.0, 20.0)]:
        pipe = Pipeline(batch_size=3, num_threads=1, device_id=0, prefetch_queue_depth=1)
        with pipe:
            x = fn.external_source(name="x")
            kw = {} if ref is None else {"reference": ref}
            pipe.set_outputs(fn.to_decibels(x.gpu(), multiplier=mult, cutoff_db=cutoff, **kw))
        pipe.feed_input("x", data)
        (out,) = pipe.run()
        for i, d in enumerate(data):
            want = A.to_decibels(d, mult, 0.0 if ref is None else ref, cutoff)
            got = out[i].as_cpu()
            assert np.abs(got - want).max() <= 1e-3 + 1e-4 * np.abs(want).max(), (i, ref)


@pytest.mark.parametrize("nfft,wl,step", [(2, 2, 1), (8, 5, 3), (64, 64, 16), (128, 100, 50), (256, 256, 64),
                                         (2048, 2048, 512), (4096, 3000, 1000)])
def test_spectrogram_every_fft_size(nfft, wl, step):
    """Every template instance of the half-length radix-4 FFT (odd and even log2, fewer butterflies than lanes, the
    two-frames-per-wave nfft=4096 layout), with windows shorter than nfft and both padding modes."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(nfft)
    sigs = [rng.normal(0, 0.3, n).astype(np.float32) for n in (max(wl, 37), 5000, 12345)]
    for reflect, center, power in [(True, True, 2), (False, True, 1), (True, False, 2)]:
        pipe = Pipeline(batch_size=len(sigs), num_threads=2, device_id=0, prefetch_queue_depth=1)
        with pipe:
            x = fn.external_source(name="x")
            pipe.set_outputs(fn.spectrogram(x.gpu(), nfft=nfft, window_length=wl, window_step=step, power=power,
                                            center_windows=center, reflect_padding=reflect))
        pipe.feed_input("x", sigs)
        (spec,) = pipe.run()
        for i, s in enumerate(sigs):
            ref = A.spectrogram(s, nfft=nfft, window_length=wl, window_step=step, power=power, center_windows=center,
                                reflect_padding=reflect)
            got = spec[i].as_cpu()
            assert got.shape == ref.shape, (got.shape, ref.shape)
            err = np.abs(got - ref).max()
            assert err <= 1e-4 * ref.max() + 1e-6, (nfft, reflect, center, power, i, err, ref.max())


def test_spectrogram_quiet_bins_relative():
    """a bound relative to max(ref) hides errors in quiet bins.  A signal with 60 dB of dynamic range
    across the band: every bin above a floor of 1e-6 of the maximum is held to 2e-3 relative."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(21)
    n = 20000
    t = np.arange(n) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 5e-4 * np.sin(2 * np.pi * 3000 * t) + rng.normal(0, 2e-5, n)).astype(np.float32)
    pipe = Pipeline(batch_size=1, num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        s = fn.external_source(name="x")
        pipe.set_outputs(fn.spectrogram(s.gpu(), nfft=1024, window_length=1024, window_step=256, power=2))
    pipe.feed_input("x", [x])
    (spec,) = pipe.run()
    got = spec[0].as_cpu().astype(np.float64)
    ref = A.spectrogram(x, nfft=1024, window_length=1024, window_step=256, power=2).astype(np.float64)
    loud = ref > 1e-6 * ref.max()
    assert loud.mean() > 0.02
    rel = np.abs(got[loud] - ref[loud]) / ref[loud]
    assert rel.max() <= 2e-3, rel.max()


@pytest.mark.parametrize("kw", [dict(), dict(n_mfcc=13, dct_type=2, normalize=True, lifter=22.0),
                                dict(n_mfcc=40, dct_type=3), dict(n_mfcc=8, dct_type=4, normalize=True),
                                dict(n_mfcc=10, dct_type=1, lifter=5.0), dict(n_mfcc=200)])
def test_mfcc_matches_oracle(kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(31)
    mels = [np.abs(rng.normal(0, 1, (64, t))).astype(np.float32) for t in (1, 37, 129, 400)]
    pipe = Pipeline(batch_size=len(mels), num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        m = fn.external_source(name="mel", layout="ft")
        pipe.set_outputs(fn.mfcc(m.gpu(), **kw))
    pipe.feed_input("mel", mels, layout="ft")
    (out,) = pipe.run()
    assert pipe.executed_kernels() == ["h2d_copy", "mfcc_dct"]
    for i, mel in enumerate(mels):
        ref = A.mfcc(mel, n_mfcc=kw.get("n_mfcc", 20), dct_type=kw.get("dct_type", 2), normalize=kw.get("normalize", False),
                     lifter=kw.get("lifter", 0.0))
        got = out[i].as_cpu()
        assert got.shape == ref.shape, (got.shape, ref.shape)
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (kw, i)


@pytest.mark.parametrize("shape,axis", [((50, 64), 1), ((3, 64, 41), 1), ((2, 5, 32), 2), ((40, 7, 3), 0)])
def test_mfcc_along_any_axis(shape, axis):
    """The transform runs along `axis`, every other extent is kept (mfcc.cc:128-172)."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(33)
    x = np.abs(rng.normal(0, 1, shape)).astype(np.float32)
    pipe = Pipeline(batch_size=2, num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        m = fn.external_source(name="mel")
        pipe.set_outputs(fn.mfcc(m.gpu(), axis=axis, n_mfcc=12, lifter=3.0))
    pipe.feed_input("mel", [x, x[::-1].copy()])
    (out,) = pipe.run()
    for i, src in enumerate([x, x[::-1]]):
        flat = np.moveaxis(src, axis, 0).reshape(shape[axis], -1)
        ref = A.mfcc(flat, n_mfcc=12, lifter=3.0).reshape((12,) + tuple(np.delete(shape, axis)))
        ref = np.moveaxis(ref, 0, axis)
        got = out[i].as_cpu()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_mfcc_rejects_what_the_reference_rejects():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    for kw, msg in [(dict(dct_type=5), "Unsupported DCT type"), (dict(dct_type=1, normalize=True), "not supported")]:
        pipe = Pipeline(batch_size=1, num_threads=1, device_id=0)
        with pipe:
            m = fn.external_source(name="mel")
            pipe.set_outputs(fn.mfcc(m.gpu(), **kw))
        with pytest.raises(Exception, match=msg):
            pipe.build()


@pytest.mark.parametrize("kw", [dict(in_rate=44100.0, out_rate=16000.0), dict(in_rate=16000.0, out_rate=22050.0, quality=0.0),
                                dict(scale=0.37, quality=100.0), dict(out_length=777), dict(scale=1.0)])
def test_audio_resample_matches_oracle(kw):
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(41)
    sigs = [rng.normal(0, 0.3, n).astype(np.float32) for n in (1, 300, 5000)]
    sigs.append(rng.normal(0, 0.3, (2000, 2)).astype(np.float32))
    for group in (sigs[:3], sigs[3:]):
        pipe = Pipeline(batch_size=len(group), num_threads=1, device_id=0, prefetch_queue_depth=1)
        with pipe:
            s = fn.external_source(name="x")
            pipe.set_outputs(fn.audio_resample(s.gpu(), **kw))
        pipe.feed_input("x", group)
        (out,) = pipe.run()
        assert pipe.executed_kernels() == ["h2d_copy", "audio_resample"]
        for i, x in enumerate(group):
            if "out_length" in kw:
                ref = A.audio_resample(x, x.shape[0], kw["out_length"], kw.get("quality", 50.0), out_length=kw["out_length"])
            elif "scale" in kw:
                ref = A.audio_resample(x, 1.0, float(np.float32(kw["scale"])), kw.get("quality", 50.0))
            else:
                ref = A.audio_resample(x, kw["in_rate"], kw["out_rate"], kw.get("quality", 50.0))
            got = out[i].as_cpu()
            assert got.shape == ref.shape, (got.shape, ref.shape, kw)
            # the reference's own cpu-vs-gpu bound (test_audio_resample.py:60): mean 1e-6, max 1e-4
            err = np.abs(got - ref)
            assert err.max() <= 1e-4 and err.mean() <= 1e-6, (kw, i, err.max(), err.mean())


def test_audio_resample_argument_errors():
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    for kw, msg in [(dict(), "No resampling factor"), (dict(in_rate=1.0), "must be specified together"),
                    (dict(scale=2.0, out_length=5), "cannot be used together"), (dict(scale=1.0, quality=101.0), "out of range")]:
        pipe = Pipeline(batch_size=1, num_threads=1, device_id=0)
        with pipe:
            s = fn.external_source(name="x")
            pipe.set_outputs(fn.audio_resample(s.gpu(), **kw))
        with pytest.raises(Exception, match=msg):
            pipe.build()


@pytest.mark.parametrize("in_t,out_t", [(np.int16, None), (np.float32, np.int16), (np.uint8, np.float32), (np.int16, np.uint16),
                                        (np.int32, np.int8)])
def test_audio_resample_integer_sample_types_on_the_gpu(in_t, out_t):
    """Typed samples: normalise -> float resampler -> saturating conversion, three launches; against the oracle with
    the bound the CPU backend's test uses."""
    from dali_amd import fn, types
    from dali_amd.pipeline import Pipeline
    rng = np.random.default_rng(6)
    t = np.arange(4000) / 16000.0
    sig = np.stack([0.8 * np.sin(2 * np.pi * 440 * t) + rng.normal(0, 0.05, t.size), 0.3 * np.cos(2 * np.pi * 90 * t)], 1)

    def as_type(x):
        if in_t == np.float32:
            return (x * 1.3).astype(np.float32)
        info = np.iinfo(in_t)
        if info.min < 0:
            return np.clip(np.round(x * info.max), info.min, info.max).astype(in_t)
        return np.clip(np.round((x * 0.5 + 0.5) * info.max), 0, info.max).astype(in_t)

    samples = [as_type(sig), as_type(sig[::-1].copy()), as_type(sig[:1])]
    to_dali = {np.int8: types.INT8, np.uint8: types.UINT8, np.int16: types.INT16, np.uint16: types.UINT16,
               np.int32: types.INT32, np.uint32: types.UINT32, np.float32: types.FLOAT}
    pipe = Pipeline(batch_size=3, num_threads=1, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        kw = {} if out_t is None else {"dtype": to_dali[out_t]}
        pipe.set_outputs(fn.audio_resample(x.gpu(), in_rate=16000.0, out_rate=22050.0, **kw))
    for _ in range(2):
        pipe.feed_input("x", samples)
        (out,) = pipe.run()
    want_t = np.dtype(in_t if out_t is None else out_t)
    names = pipe.executed_kernels()
    assert "audio_resample" in names and ("audio_samples_to_float" in names) == (in_t != np.float32 or want_t.kind == "u")
    assert ("audio_samples_from_float" in names) == (want_t != np.float32)
    for i, s in enumerate(samples):
        ref = A.audio_resample_typed(s, 16000.0, 22050.0, out_dtype=want_t)
        got = out[i].as_cpu()
        assert got.dtype == want_t and got.shape == ref.shape
        if want_t == np.float32:
            assert np.abs(got - ref).max() <= 1e-4
        else:
            full = float(np.iinfo(want_t).max)
            d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
            assert d.max() <= max(1.0, 2e-4 * full), (d.max(), full)
            assert (d > max(1.0, 2e-6 * full)).mean() < 0.01


def _chain_pipe(bs, outputs, reference=None, **kw):
    """spectrogram -> mel_filter_bank -> to_decibels with ONLY `outputs` leaving the pipeline: a chain whose spectrogram
    (and mel energies) have a single consumer is fused into one launch (ops.h: DeferredAudio)."""
    from dali_amd import fn
    from dali_amd.pipeline import Pipeline
    pipe = Pipeline(batch_size=bs, num_threads=2, device_id=0, prefetch_queue_depth=1)
    with pipe:
        x = fn.external_source(name="x")
        spec = fn.spectrogram(x.gpu(), nfft=kw.get("nfft", 1024), window_length=kw.get("wl", 1024),
                              window_step=kw.get("step", 256), power=kw.get("power", 2),
                              center_windows=kw.get("center", True), reflect_padding=kw.get("reflect", True))
        mel = fn.mel_filter_bank(spec, nfilter=kw.get("nfilter", 80), sample_rate=16000.0, freq_high=8000.0,
                                 mel_formula=kw.get("formula", "slaney"), normalize=kw.get("normalize", True))
        ref = {} if reference is None else {"reference": reference}
        db = fn.to_decibels(mel, multiplier=10.0, cutoff_db=-80.0, **ref)
        nodes = dict(spec=spec, mel=mel, db=db)
        pipe.set_outputs(*[nodes[o] for o in outputs])
    pipe.build()
    return pipe


@pytest.mark.parametrize("kw", [dict(), dict(nfft=512, wl=400, step=160, nfilter=64, formula="htk"),
                                dict(power=1, center=False, nfilter=128), dict(reflect=False, normalize=False, nfilter=23)])
@pytest.mark.parametrize("variant", ["mfma", "valu"])
def test_fused_spectrogram_mel_decibels_equals_the_three_launches(kw, variant, monkeypatch):
    """The fused launch (power tile -> filter bank in LDS, on the f32 matrix cores or as the banded VALU product) gives
    what the three separate kernels give: the mel energies are the same ascending-k fma chain (f32 MFMA accumulates
    exactly like fmaf; zero weights add exact zeros), so the comparison is bit for bit; the unfused chain itself is held
    to the oracle by test_audio_pipeline_matches_oracle."""
    monkeypatch.setenv("DALI_AMD_MEL_VALU", "1" if variant == "valu" else "0")
    rng = np.random.default_rng(21)
    sigs = [synth_signal(rng, s).astype(np.float32) for s in (1.3, 2.0, 0.05 if kw.get("center", True) else 0.7, 3.1, 0.9)]
    name = "spectrogram_mel_fused" + ("_mfma" if variant == "mfma" else "")
    plain = _chain_pipe(len(sigs), ["spec", "mel", "db"], **kw)
    plain.feed_input("x", sigs)
    _, mel0, db0 = plain.run()
    assert plain.executed_kernels() == ["h2d_copy", "spectrogram", "mel_filter_bank_banded", "to_decibels"]
    # 1. mel only
    p1 = _chain_pipe(len(sigs), ["mel"], **kw)
    p1.feed_input("x", sigs)
    (mel1,) = p1.run()
    assert p1.executed_kernels() == ["h2d_copy", name]
    # 2. decibels against the sample's maximum: fused launch (collects the maxima) + the in-place element-wise pass
    p2 = _chain_pipe(len(sigs), ["db"], **kw)
    p2.feed_input("x", sigs)
    (db2,) = p2.run()
    assert p2.executed_kernels() == ["h2d_copy", name, "to_decibels"]
    # 3. decibels against a given reference: one launch
    plain3 = _chain_pipe(len(sigs), ["mel", "db"], reference=0.37, **kw)
    plain3.feed_input("x", sigs)
    _, db3_ref = plain3.run()
    p3 = _chain_pipe(len(sigs), ["db"], reference=0.37, **kw)
    p3.feed_input("x", sigs)
    (db3,) = p3.run()
    assert p3.executed_kernels() == ["h2d_copy", name]
    for i in range(len(sigs)):
        a, b = mel0[i].as_cpu(), mel1[i].as_cpu()
        assert a.shape == b.shape == (kw.get("nfilter", 80), a.shape[1])
        assert np.array_equal(a, b), f"mel sample {i}: max rel diff {(np.abs(a - b) / (np.abs(a).max() + 1e-30)).max()}"
        assert np.array_equal(db0[i].as_cpu(), db2[i].as_cpu()), f"dB (max reference) sample {i}"
        assert np.array_equal(db3_ref[i].as_cpu(), db3[i].as_cpu()), f"dB (reference 0.37) sample {i}"


def test_chains_the_fused_kernel_does_not_cover_run_unfused():
    rng = np.random.default_rng(4)
    sigs = [synth_signal(rng, 0.5).astype(np.float32) for _ in range(2)]
    pipe = _chain_pipe(2, ["db"], nfft=2048, wl=2048, step=512)       # only nfft 512 / 1024 have the register-resident FFT
    pipe.feed_input("x", sigs)
    (db,) = pipe.run()
    assert pipe.executed_kernels() == ["h2d_copy", "spectrogram", "mel_filter_bank_banded", "to_decibels"]
    ref = A.to_decibels(A.mel_filter_bank(A.spectrogram(sigs[0], nfft=2048, window_length=2048, window_step=512), 80, 16000.0, 0.0,
                                          8000.0, True, "slaney"), 10.0, 0.0, -80.0)
    assert np.abs(db[0].as_cpu() - ref).max() <= 2e-2
