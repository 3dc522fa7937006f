"""ORACLE (test infrastructure only -- never imported by the product package).

numpy restatement of the reference's audio feature path (BASELINE.json configs[3]):

  Hann window (periodic, half-sample shifted)   dali/kernels/signal/window/window_functions.h:25-33
  window extraction, centring, reflect-101 pad  dali/kernels/signal/window/extract_windows_cpu.cc:96-145,
                                                extract_windows_args.h:38-43 (num_windows)
  FFT size / window centring inside nfft        dali/kernels/signal/fft/fft_cpu_impl_ffts.cc:105-111
  spectrum type (power / magnitude)             dali/operators/signal/fft/spectrogram.cc:128-146
  mel filter bank (weights in double, Slaney / HTK scales, area normalisation)
                                                dali/kernels/audio/mel_scale/mel_scale.h:27-130,
                                                mel_filter_bank_cpu.cc:44-111
  decibels                                      dali/kernels/signal/decibel/decibel_calculator.h:25-52,
                                                to_decibels_cpu.cc:54-66
  WAV decode (PCM16 -> float in [-1, 1))        dali/operators/decoder/audio/generic_decoder.cc:140-220 (libsndfile)

The reference's CPU FFT is the un-vendored FFTS library; like the reference's own tests
(dali/test/python/operator_2/test_spectrogram.py:188, eps 1e-4) the oracle uses a float64 FFT (numpy.fft.rfft)
and the comparison is tolerance-based: PARITY for the FFT stage is "within 1e-4 relative", not bit-exact.
"""
import io
import struct

import numpy as np


def hann_window(n):
    a = 2 * np.pi / n
    t = np.arange(n, dtype=np.float64)
    return (0.5 * (1.0 - np.cos(a * (t + 0.5)))).astype(np.float32)


def num_windows(length, window_size, step, centered):
    if not centered:
        length -= window_size
    return length // step + 1


def _reflect101(idx, size):
    if size < 2:
        return np.full_like(idx, size - 1)
    idx = idx.copy()
    while True:
        lo, hi = idx < 0, idx >= size
        if not (lo.any() or hi.any()):
            return idx
        idx[lo] = -idx[lo]
        idx[hi] = 2 * size - 2 - idx[hi]


def spectrogram(x, nfft=None, window_length=512, window_step=256, window_fn=None, power=2, center_windows=True,
                reflect_padding=True):
    """x: float32 [L].  Returns float32 [nfft/2+1, T] ("ft" layout)."""
    x = np.asarray(x, np.float32)
    nfft = window_length if nfft is None else nfft
    assert window_length <= nfft
    win = hann_window(window_length) if window_fn is None else np.asarray(window_fn, np.float32)
    L = x.shape[0]
    T = num_windows(L, window_length, window_step, center_windows)
    center = window_length // 2 if center_windows else 0
    starts = np.arange(T, dtype=np.int64) * window_step - center
    idx = starts[:, None] + np.arange(window_length, dtype=np.int64)[None, :]
    if reflect_padding:
        frames = x[_reflect101(idx, L)] * win[None, :]           # float32 product, like the CPU kernel
    else:
        ok = (idx >= 0) & (idx < L)
        frames = np.where(ok, x[np.clip(idx, 0, L - 1)] * win[None, :], np.float32(0))
    frames = frames.astype(np.float32)
    buf = np.zeros((T, nfft), np.float64)
    s = (nfft - window_length) // 2
    buf[:, s:s + window_length] = frames
    X = np.fft.rfft(buf, axis=1)
    p = X.real ** 2 + X.imag ** 2
    if power == 1:
        p = np.sqrt(p)
    return p.T.astype(np.float32)


def _hz_to_mel(hz, formula):
    hz = float(hz)
    if formula == "htk":
        return 1127.0 * np.log(1.0 + hz / 700.0)
    fsp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, step_log = (min_log_hz - 0.0) / fsp, 0.068751777
    return min_log_mel + np.log(hz / min_log_hz) / step_log if hz >= min_log_hz else (hz - 0.0) / fsp


def _mel_to_hz(mel, formula):
    mel = float(mel)
    if formula == "htk":
        return 700.0 * (np.exp(mel / 1127.0) - 1.0)
    fsp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, step_log = (min_log_hz - 0.0) / fsp, 0.068751777
    return min_log_hz * np.exp(step_log * (mel - min_log_mel)) if mel >= min_log_mel else 0.0 + mel * fsp


def mel_weights(nfilter, nfft, sample_rate, freq_low=0.0, freq_high=0.0, normalize=True, mel_formula="slaney"):
    """Dense [nfilter, nfft/2+1] float32 matrix equivalent to MelFilterImplBase + ComputeFreqMajor."""
    if freq_high <= 0:
        freq_high = sample_rate / 2
    mel_low, mel_high = _hz_to_mel(freq_low, mel_formula), _hz_to_mel(freq_high, mel_formula)
    hz_step = float(sample_rate) / nfft
    mel_delta = (mel_high - mel_low) / (nfilter + 1)
    nbin = nfft // 2 + 1
    inv = 1.0 / hz_step
    b0 = int(np.ceil(freq_low * inv))
    b1 = min(int(np.ceil(freq_high * inv)), nbin)
    weights_down = np.zeros(nbin, np.float32)
    norm = np.ones(nfilter, np.float32)
    intervals = np.full(nbin, -1, np.int64)
    mel0, mel1 = mel_low, mel_low + mel_delta
    fftbin = b0
    f = fftbin * hz_step
    for interval in range(nfilter + 1):
        if interval == nfilter:
            mel1 = mel_high
        f0, f1 = _mel_to_hz(mel0, mel_formula), _mel_to_hz(mel1, mel_formula)
        if normalize and interval < nfilter:
            f2 = _mel_to_hz(mel1 + mel_delta, mel_formula)
            norm[interval] = np.float32(2.0 / (f2 - f0))
        slope = 1.0 / (f1 - f0)
        while fftbin < b1 and f < f1:
            weights_down[fftbin] = np.float32((f1 - f) * slope)
            intervals[fftbin] = interval
            fftbin += 1
            f = fftbin * hz_step
        mel0, mel1 = mel1, mel1 + mel_delta
    W = np.zeros((nfilter, nbin), np.float32)
    for b in range(b0, b1):
        up = intervals[b]
        wd = weights_down[b]
        wu = np.float32(1) - wd
        down = up - 1
        if down >= 0:
            W[down, b] = wd * norm[down] if normalize else wd
        if 0 <= up < nfilter:
            W[up, b] = wu * norm[up] if normalize else wu
    return W


def mel_filter_bank(spec, nfilter=128, sample_rate=44100.0, freq_low=0.0, freq_high=0.0, normalize=True,
                    mel_formula="slaney"):
    """spec: float32 [nfft/2+1, T] -> float32 [nfilter, T]; accumulation over bins in increasing order, float32,
    multiply then add (ComputeFreqMajor)."""
    spec = np.asarray(spec, np.float32)
    nfft = 2 * (spec.shape[0] - 1)
    W = mel_weights(nfilter, nfft, sample_rate, freq_low, freq_high, normalize, mel_formula)
    out = np.zeros((nfilter, spec.shape[1]), np.float32)
    for b in range(spec.shape[0]):
        nz = np.nonzero(W[:, b])[0]
        for m in nz:
            out[m] += W[m, b] * spec[b]
    return out


def to_decibels(x, multiplier=10.0, reference=0.0, cutoff_db=-200.0):
    x = np.asarray(x, np.float32)
    min_ratio = np.float32(10.0 ** (cutoff_db / multiplier))
    if min_ratio == 0:
        min_ratio = np.nextafter(np.float32(0), np.float32(1))
    if reference == 0.0:
        s_ref = np.float32(x.max()) if x.size else np.float32(1)
        if s_ref == 0:
            s_ref = np.float32(1)
    else:
        s_ref = np.float32(reference)
    inv = np.float32(1) if s_ref == 1 else np.float32(1) / s_ref
    mul_log2 = np.float32(np.float32(multiplier) * np.float32(0.3010299956639812))
    return (mul_log2 * np.log2(np.maximum(min_ratio, x * inv))).astype(np.float32)


def decode_wav(data):
    """PCM WAV -> (float32 [T] or [T, C] in [-1, 1), sample_rate) like sf_readf_float."""
    b = io.BytesIO(data)
    riff, _, wave = struct.unpack("<4sI4s", b.read(12))
    assert riff == b"RIFF" and wave == b"WAVE"
    fmt = None
    while True:
        hdr = b.read(8)
        if len(hdr) < 8:
            raise ValueError("no data chunk")
        cid, size = struct.unpack("<4sI", hdr)
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", b.read(16))
            b.read(size - 16)
        elif cid == b"data":
            raw = b.read(size)
            break
        else:
            b.read(size + (size & 1))
    tag, ch, rate, _, _, bits = fmt
    if tag == 1 and bits in (8, 16, 24, 32):
        a = pcm_to_float(wav_pcm(raw, bits), bits)
    elif tag == 3 and bits == 32:
        a = np.frombuffer(raw, "<f4").astype(np.float32)
    else:
        raise ValueError("unsupported WAV encoding")
    return (a.reshape(-1, ch) if ch > 1 else a), float(rate)


def wav_pcm(raw, bits):
    """The integer samples of a PCM data chunk in their own range (8-bit WAV is unsigned: offset 128)."""
    if bits == 8:
        return np.frombuffer(raw, np.uint8).astype(np.int32) - 128
    if bits == 16:
        return np.frombuffer(raw, "<i2").astype(np.int32)
    if bits == 32:
        return np.frombuffer(raw, "<i4").astype(np.int32)
    b = np.frombuffer(raw[:len(raw) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
    return ((b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) ^ 0x800000) - 0x800000


# libsndfile's reads of `bits`-wide integer samples (what GenericAudioDecoder gets, generic_decoder.cc:170-183):
#   sf_readf_float  x / 2^(bits-1);   sf_readf_short  the top 16 bits (narrower samples shifted up);   sf_readf_int  x << (32 - bits)
def pcm_to_float(x, bits):
    return x.astype(np.float32) * np.float32(1.0 / (1 << (bits - 1)))


def pcm_to_int16(x, bits):
    return (x >> (bits - 16) if bits > 16 else x << (16 - bits)).astype(np.int16)


def pcm_to_int32(x, bits):
    return (x.astype(np.int64) << (32 - bits)).astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------
# FLAC (RFC 9639), plain Python: the checker of dali_amd/host/flac_decode.cpp on the committed fixtures
# (tests/golden/flac, written by tests/golden/make_flac_golden.py - a third, independent piece of code).  Slow by design.
# ---------------------------------------------------------------------------------------------------------------
class _FlacBits:
    def __init__(self, data, pos):
        self.d, self.pos = data, pos * 8

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.pos >> 3] >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def s(self, n):
        v = self.u(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self):
        z = 0
        while self.u(1) == 0:
            z += 1
        return z


def _flac_residual(br, blocksize, order):
    method = br.u(2)
    assert method < 2
    pbits = 4 + method
    porder = br.u(4)
    out = []
    for part in range(1 << porder):
        count = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.u(pbits)
        if k == (1 << pbits) - 1:
            raw = br.u(5)
            out += [br.s(raw) for _ in range(count)]
        else:
            for _ in range(count):
                v = (br.unary() << k) | br.u(k)
                out.append((v >> 1) ^ -(v & 1))
    return out


def _flac_subframe(br, blocksize, bps):
    assert br.u(1) == 0
    kind = br.u(6)
    wasted = br.unary() + 1 if br.u(1) else 0
    bps -= wasted
    if kind == 0:
        x = [br.s(bps)] * blocksize
    elif kind == 1:
        x = [br.s(bps) for _ in range(blocksize)]
    elif 8 <= kind <= 12:
        order = kind - 8
        x = [br.s(bps) for _ in range(order)]
        taps = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        for r in _flac_residual(br, blocksize, order):
            x.append(r + sum(t * x[-1 - j] for j, t in enumerate(taps)))
    elif kind >= 32:
        order = kind - 31
        x = [br.s(bps) for _ in range(order)]
        prec = br.u(4) + 1
        shift = br.s(5)
        taps = [br.s(prec) for _ in range(order)]
        for r in _flac_residual(br, blocksize, order):
            x.append(r + (sum(t * x[-1 - j] for j, t in enumerate(taps)) >> shift))
    else:
        raise ValueError("reserved subframe type")
    return [v << wasted for v in x]


def decode_flac(data):
    """-> (int32 [frames][channels] in the stream's own range, bits per sample, sample rate)."""
    assert data[:4] == b"fLaC"
    pos, last = 4, False
    while not last:
        last, kind = data[pos] >> 7, data[pos] & 0x7F
        size = int.from_bytes(data[pos + 1:pos + 4], "big")
        if kind == 0:
            v = int.from_bytes(data[pos + 4 + 10:pos + 4 + 18], "big")
            rate, channels, bps = v >> 44, ((v >> 41) & 7) + 1, ((v >> 36) & 31) + 1
        pos += 4 + size
    chans = [[] for _ in range(channels)]
    while pos + 2 <= len(data):
        br = _FlacBits(data, pos)
        assert br.u(14) == 0x3FFE and br.u(1) == 0
        br.u(1)
        bs_code, sr_code, ch_code, ss_code = br.u(4), br.u(4), br.u(4), br.u(3)
        assert br.u(1) == 0
        first = br.u(8)
        extra = 0
        while first & (0x80 >> extra):
            extra += 1
        for _ in range(max(0, extra - 1)):
            br.u(8)
        if bs_code == 1:
            blocksize = 192
        elif bs_code == 6:
            blocksize = br.u(8) + 1
        elif bs_code == 7:
            blocksize = br.u(16) + 1
        else:
            blocksize = (576 << (bs_code - 2)) if bs_code <= 5 else (256 << (bs_code - 8))
        if sr_code == 12:
            br.u(8)
        elif sr_code in (13, 14):
            br.u(16)
        br.u(8)   # CRC-8 (the product checks it; here the frame is taken on trust)
        sub = []
        for c in range(channels):
            side = (ch_code == 8 and c == 1) or (ch_code == 9 and c == 0) or (ch_code == 10 and c == 1)
            sub.append(_flac_subframe(br, blocksize, bps + (1 if side else 0)))
        if ch_code == 8:
            sub[1] = [a - b for a, b in zip(sub[0], sub[1])]
        elif ch_code == 9:
            sub[0] = [a + b for a, b in zip(sub[0], sub[1])]
        elif ch_code == 10:
            mid = [(m << 1) | (sd & 1) for m, sd in zip(sub[0], sub[1])]
            sub = [[(m + sd) >> 1 for m, sd in zip(mid, sub[1])], [(m - sd) >> 1 for m, sd in zip(mid, sub[1])]]
        for c in range(channels):
            chans[c] += sub[c]
        pos = ((br.pos + 7) >> 3) + 2
    return np.array(chans, np.int64).T.astype(np.int32), bps, float(rate)


# ---------------------------------------------------------------------------------------------------------------
# MFCC: DCT along the first axis + liftering (dali/kernels/signal/dct/table.h:26-96, dct_cpu.cc:75-110,
# dali/operators/audio/mfcc/mfcc.h:43-48, mfcc.cc:41-60).  The table is built in double and rounded to float like
# the reference's; the sum runs in float64 here (the comparison is tolerance-based: the reference's own test compares
# with librosa at 1e-3, test_mfcc.py:153).
# ---------------------------------------------------------------------------------------------------------------
def dct_table(dct_type, normalize, n_in, ndct):
    k = np.arange(ndct, dtype=np.float64)[:, None]
    n = np.arange(n_in, dtype=np.float64)[None, :]
    if dct_type == 1:
        t = np.cos(np.pi / (n_in - 1) * k * n)
        t[:, 0] = 0.5
        t[:, n_in - 1] = np.where(np.arange(ndct) % 2 == 0, 0.5, -0.5)
    elif dct_type == 2:
        t = np.cos(np.pi / n_in * (n + 0.5) * k)
        if normalize:
            t *= np.where(np.arange(ndct)[:, None] == 0, 1.0 / np.sqrt(n_in), np.sqrt(2.0 / n_in))
    elif dct_type == 3:
        f0, fi = (1.0 / np.sqrt(n_in), np.sqrt(2.0 / n_in)) if normalize else (0.5, 1.0)
        t = fi * np.cos(np.pi / n_in * n * (k + 0.5))
        t[:, 0] = f0
    elif dct_type == 4:
        t = (np.sqrt(2.0 / n_in) if normalize else 1.0) * np.cos(np.pi / n_in * (n + 0.5) * (k + 0.5))
    else:
        raise ValueError(f"Unsupported DCT type: {dct_type}")
    return t.astype(np.float32)


def lifter_coeffs(lifter, n):
    if lifter == 0:
        return np.ones(n, np.float32)
    i = np.arange(n, dtype=np.float32)
    return (np.float32(1) + np.float32(lifter / 2) * np.sin(np.float32(np.pi) / np.float32(lifter) * (i + 1))).astype(np.float32)


def mfcc(mel, n_mfcc=20, dct_type=2, normalize=False, lifter=0.0):
    """mel: float32 [n_in, T] -> float32 [ndct, T]."""
    mel = np.asarray(mel, np.float32)
    n_in = mel.shape[0]
    ndct = n_in if n_mfcc <= 0 or n_mfcc > n_in else n_mfcc
    t = dct_table(dct_type, normalize and dct_type != 1, n_in, ndct).astype(np.float64)
    out = t @ mel.astype(np.float64)
    return (lifter_coeffs(lifter, ndct).astype(np.float64)[:, None] * out).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# Audio resampling (dali/kernels/signal/resampling.h:33-106, resampling_cpu.cc:129-172, resampling_params.h:27-30).
# The window table and the positions follow the reference in float32 (the table entries, the block start in double,
# the position inside a block of 256 outputs by repeated float additions); the filter sum runs in float64.
# ---------------------------------------------------------------------------------------------------------------
def resample_lobes(quality):
    return int(round(0.007 * quality * quality - 0.09 * quality + 3))


def resample_window(lobes):
    coeffs = lobes * 64 + 1
    scale = np.float32(2.0 * lobes / (coeffs - 1))
    scale_env = np.float32(2.0 / coeffs)
    center = int((coeffs - 1) * 0.5)
    i = np.arange(coeffs, dtype=np.float32)
    x = ((i - center) * scale).astype(np.float32)
    y = ((i - center) * scale_env).astype(np.float32)
    xp = (x.astype(np.float64) * np.pi).astype(np.float32)       # `x *= M_PI`: the product is formed in double
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(np.abs(xp) < 1e-5, np.float32(1) - xp * xp * np.float32(1.0 / 6), np.sin(xp) / xp).astype(np.float32)
    hann = 0.5 * (1 + np.cos(y.astype(np.float64) * np.pi))
    lookup = np.zeros(coeffs + 5, np.float32)
    lookup[1:coeffs + 1] = (sinc * hann).astype(np.float32)
    return lookup, np.float32(1) / scale, np.float32(center + 1)


def audio_resample(x, in_rate, out_rate, quality=50.0, out_length=None):
    """x: float32 [L] or [L, C].  Returns float32 [ceil(L * out_rate / in_rate)(, C)]."""
    x = np.asarray(x, np.float32)
    squeeze = x.ndim == 1
    if squeeze:
        x = x[:, None]
    n_in = x.shape[0]
    lobes = resample_lobes(quality)
    lookup, wscale, wcenter = resample_window(lobes)
    n_out = int(np.ceil(n_in * out_rate / in_rate)) if out_length is None else int(out_length)
    scale = float(in_rate) / float(out_rate)
    fscale = np.float32(scale)
    out = np.zeros((n_out, x.shape[1]), np.float32)
    xd = x.astype(np.float64)
    for out_block in range(0, n_out, 256):
        nb = min(256, n_out - out_block)
        in_block_f = out_block * scale
        in_block_i = int(np.floor(in_block_f))
        steps = np.full(nb, fscale, np.float32)
        steps[0] = np.float32(in_block_f - in_block_i)
        in_pos = np.add.accumulate(steps, dtype=np.float32)     # p0, p0 + fscale, (p0 + fscale) + fscale, ...
        for j in range(nb):
            p = in_pos[j]
            xc = int(np.ceil(p))
            i0, i1 = xc - lobes, xc + lobes
            i0 = max(i0, -in_block_i)
            i1 = min(i1, n_in - in_block_i)
            if i1 <= i0:
                continue
            xs = (np.arange(i0, i1, dtype=np.float32) - p).astype(np.float32)      # i - in_pos, then x++ (exact here)
            fi = (xs * wscale + wcenter).astype(np.float32)
            fl = np.floor(fi)
            di = (fi - fl).astype(np.float32)
            li = fl.astype(np.int64)
            w = (lookup[li] + di * (lookup[li + 1] - lookup[li])).astype(np.float32)
            out[out_block + j] = (xd[in_block_i + i0:in_block_i + i1] * w.astype(np.float64)[:, None]).sum(0)
    return out[:, 0] if squeeze else out


# ---------------------------------------------------------------------------------------------------------------
# Typed audio resampling (dali/operators/audio/resample.cc:142-192, include/dali/core/convert.h:262-350): integer
# samples are normalised to floats (x / max of the type), squeezed to [0, 1] when a signed source feeds an unsigned
# result or stretched to [-1, 1] the other way round, resampled as floats and converted back with
# clamp(round_half_away(f * max)).
# ---------------------------------------------------------------------------------------------------------------
_NORM_MAX = {np.dtype(np.int8): 127.0, np.dtype(np.uint8): 255.0, np.dtype(np.int16): 32767.0, np.dtype(np.uint16): 65535.0,
             np.dtype(np.int32): 2147483647.0, np.dtype(np.uint32): 4294967295.0}


def convert_norm_to_float(x, out_dtype):
    """ConvertInput: x of any audio sample type -> float32, as the operator prepares it for a result of `out_dtype`."""
    x = np.asarray(x)
    out_dtype = np.dtype(out_dtype)
    out_unsigned = out_dtype.kind == "u"
    if x.dtype == np.float32:
        f = x
        return ((f + np.float32(1)) * np.float32(0.5)).astype(np.float32) if out_unsigned else f
    inv = np.float32(1) / np.float32(_NORM_MAX[x.dtype])          # Out(1) / max_value<In>(), the maximum converted to float
    f = (x.astype(np.float32) * inv).astype(np.float32)
    if out_unsigned and x.dtype.kind == "i":
        return ((f + np.float32(1)) * np.float32(0.5)).astype(np.float32)
    if not out_unsigned and x.dtype.kind == "u":
        return (f * np.float32(2) - np.float32(1)).astype(np.float32)
    return f


def convert_norm_from_float(f, out_dtype):
    """ConvertSatNorm<Out>(float): clamp(round(f * max)); halves away from zero; float results pass through."""
    out_dtype = np.dtype(out_dtype)
    f = np.asarray(f, np.float32)
    if out_dtype == np.float32:
        return f
    fmax = np.float32(_NORM_MAX[out_dtype])
    v = (f * fmax).astype(np.float32).astype(np.float64)
    r = np.sign(v) * np.floor(np.abs(v) + 0.5)
    info = np.iinfo(out_dtype)
    return np.clip(r, info.min, info.max).astype(out_dtype)


def audio_resample_typed(x, in_rate, out_rate, quality=50.0, out_length=None, out_dtype=None):
    x = np.asarray(x)
    out_dtype = np.dtype(x.dtype if out_dtype is None else out_dtype)
    f = convert_norm_to_float(x, out_dtype)
    y = audio_resample(f, in_rate, out_rate, quality=quality, out_length=out_length)
    return convert_norm_from_float(y, out_dtype)


def decode_audio(pcm16, rate, out_dtype=np.float32, downmix=False, sample_rate=None, quality=50.0):
    """DecodeAudio<T> (dali/operators/decoder/audio/audio_decoder_impl.cc:49-120) for PCM16 frames [L, C]:
    straight to T when nothing else happens (int16 as stored, int32 = value << 16, float = value / 32768); otherwise
    floats, equal-weight downmix (kernels/signal/downmixing.h:50-76) and / or resampling, the last step converting to T."""
    pcm16 = np.asarray(pcm16, np.int16)
    if pcm16.ndim == 1:
        pcm16 = pcm16[:, None]
    out_dtype = np.dtype(out_dtype)
    ch = pcm16.shape[1]
    resample = sample_rate is not None and sample_rate > 0 and float(rate) != float(sample_rate)
    mix = downmix and ch > 1
    mono = downmix or ch == 1
    if not resample and not mix:
        x = pcm16[:, 0] if mono else pcm16
        if out_dtype == np.int16:
            return x.copy()
        if out_dtype == np.int32:
            return x.astype(np.int32) << 16
        return (x.astype(np.float32) * np.float32(1.0 / 32768)).astype(np.float32)
    f = (pcm16.astype(np.float32) * np.float32(1.0 / 32768)).astype(np.float32)
    if mix:
        w = np.float32(1.0) / np.float32(ch)
        acc = (f[:, 0] * w).astype(np.float32)
        for c in range(1, ch):
            acc = (acc + (f[:, c] * w).astype(np.float32)).astype(np.float32)
        f = acc
    elif mono:
        f = f[:, 0]
    if resample:
        f = audio_resample(f, float(rate), float(sample_rate), quality=quality)
    return convert_norm_from_float(f, out_dtype)
