#!/usr/bin/env python
"""Writes the FLAC fixtures under tests/golden/flac/ (committed together with this script).

No FLAC tool or library exists in this image (no libFLAC, libsndfile, ffmpeg, soundfile), so the streams are made by the
small ENCODER below, written from the published format (RFC 9639) independently of the two decoders that read them
(the product's dali_amd/host/flac_decode.cpp and the oracle's oracle/audio.py:decode_flac).  The ground truth of a
fixture is the PCM the encoder was given (<name>.npz, int32 [frames][channels]); lossless means every decoder must
return exactly that.  Each fixture pins a part of the format:

  mono16_fixed      16-bit mono, fixed predictors of orders 0-4, partitioned Rice residuals, a short last
                    frame, block size 4096 (code 1100) then an explicit 16-bit block size
  stereo16_modes    16-bit stereo, one frame each of left/right, left/side, side/right and mid/side, block size 1152
  lpc24             24-bit stereo, LPC subframes of orders 1-12 with 12-15-bit coefficients, Rice parameters > 14 (5-bit method)
  pcm8_const_verb   8-bit mono: constant and verbatim subframes, block size 192 (code 0001) and an 8-bit explicit size
  wasted_escape     16-bit stereo whose samples are multiples of 4 / 16 (wasted bits) + partitions stored raw (escape code)
  nolength          like mono16_fixed with "total samples = 0" in STREAMINFO (decoders must walk the frames)
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "flac")


class Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        if nbits == 0:
            return
        value &= (1 << nbits) - 1
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, zeros):
        while zeros >= 32:
            self.put(0, 32)
            zeros -= 32
        self.put(1, zeros + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out, first_bits = [], 6
    while v >= (1 << first_bits):
        out.append(0x80 | (v & 0x3F))
        v >>= 6
        first_bits -= 1
    lead = (0xFF << (first_bits + 1)) & 0xFF
    return bytes([lead | v] + out[::-1])


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def residual_of(x, coefs, shift):
    order = len(coefs)
    res = []
    for i in range(order, len(x)):
        pred = sum(int(coefs[j]) * int(x[i - 1 - j]) for j in range(order)) >> shift
        res.append(int(x[i]) - pred)
    return res


def write_residual(bw, res, blocksize, order, porder, force_escape=()):
    fold = [(r << 1) if r >= 0 else ((-r << 1) - 1) for r in res]
    parts = 1 << porder
    # parameter per partition: the best of 0..30 for its mean
    lens = [(blocksize >> porder) - (order if p == 0 else 0) for p in range(parts)]
    chunks, at = [], 0
    for n in lens:
        chunks.append(fold[at:at + n])
        at += n
    ks = []
    for ch in chunks:
        mean = (sum(ch) / len(ch)) if ch else 0
        ks.append(max(0, int(mean).bit_length() - 1))
    method = 1 if max(ks + [0]) >= 15 or force_escape == "five" else 0
    bw.put(method, 2)
    bw.put(porder, 4)
    pbits, esc = (5, 31) if method else (4, 15)
    at = 0
    for p, ch in enumerate(chunks):
        if p in force_escape if isinstance(force_escape, (set, tuple, list)) else False:
            raw = max([abs(r).bit_length() + 1 for r in res[at:at + len(ch)]] + [1])
            bw.put(esc, pbits)
            bw.put(raw, 5)
            for r in res[at:at + len(ch)]:
                bw.put(r, raw)
        else:
            k = min(ks[p], esc - 1)
            bw.put(k, pbits)
            for u in ch:
                bw.unary(u >> k)
                bw.put(u, k)
        at += len(ch)


def write_subframe(bw, x, bps, kind, **kw):
    x = [int(v) for v in x]
    wasted = kw.get("wasted", 0)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
        bps -= wasted
    bw.put(0, 1)
    if kind == "constant":
        bw.put(0, 6)
    elif kind == "verbatim":
        bw.put(1, 6)
    elif kind == "fixed":
        bw.put(8 + kw["order"], 6)
    else:
        bw.put(31 + len(kw["coefs"]), 6)
    if wasted:
        bw.put(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    if kind == "constant":
        assert len(set(x)) == 1
        bw.put(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            bw.put(v, bps)
    elif kind == "fixed":
        order = kw["order"]
        for v in x[:order]:
            bw.put(v, bps)
        write_residual(bw, residual_of(x, FIXED[order], 0), len(x), order, kw.get("porder", 0), kw.get("escape", ()))
    else:
        coefs, shift, prec = kw["coefs"], kw["shift"], kw["precision"]
        for v in x[:len(coefs)]:
            bw.put(v, bps)
        bw.put(prec - 1, 4)
        bw.put(shift, 5)
        for c in coefs:
            bw.put(c, prec)
        write_residual(bw, residual_of(x, coefs, shift), len(x), len(coefs), kw.get("porder", 0), kw.get("escape", ()))


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def write_frame(number, channels_pcm, bps, rate_code, mode, subframes, explicit_bs=None):
    """channels_pcm: list of per-channel lists (already decorrelated when mode >= 8); subframes: per channel (kind, kwargs)."""
    blocksize = len(channels_pcm[0])
    bw = Bits()
    bw.put(0x3FFE, 14)
    bw.put(0, 1)
    bw.put(0, 1)          # fixed block size stream: the number is the frame number
    if explicit_bs == 8:
        bs_code = 6
    elif explicit_bs == 16:
        bs_code = 7
    else:
        bs_code = BS_CODES[blocksize]
    bw.put(bs_code, 4)
    bw.put(rate_code, 4)
    bw.put(mode if mode >= 8 else len(channels_pcm) - 1, 4)
    bw.put(SS_CODES[bps], 3)
    bw.put(0, 1)
    for b in utf8_number(number):
        bw.put(b, 8)
    if bs_code == 6:
        bw.put(blocksize - 1, 8)
    elif bs_code == 7:
        bw.put(blocksize - 1, 16)
    bw.put(crc8(bw.bytes()), 8)
    for c, (x, (kind, kw)) in enumerate(zip(channels_pcm, subframes)):
        side = (mode == 8 and c == 1) or (mode == 9 and c == 0) or (mode == 10 and c == 1)
        write_subframe(bw, x, bps + (1 if side else 0), kind, **kw)
    bw.align()
    body = bw.bytes()
    return body + struct.pack(">H", crc16(body))


def stream(frames, channels, bps, rate, total, blocksize):
    info = struct.pack(">HH", blocksize, blocksize) + b"\0\0\0" + b"\0\0\0"
    v = (rate << 44) | ((channels - 1) << 41) | ((bps - 1) << 36) | total
    info += struct.pack(">Q", v) + b"\0" * 16
    # a padding block in front of STREAMINFO's successor position exercises the metadata walk
    return b"fLaC" + bytes([0, 0, 0, 34]) + info + bytes([0x81, 0, 0, 6]) + b"\0" * 6 + b"".join(frames)


def decorrelate(left, right, mode):
    left, right = [int(v) for v in left], [int(v) for v in right]
    if mode == 8:
        return [left, [a - b for a, b in zip(left, right)]]
    if mode == 9:
        return [[a - b for a, b in zip(left, right)], right]
    if mode == 10:
        return [[(a + b) >> 1 for a, b in zip(left, right)], [a - b for a, b in zip(left, right)]]
    return [left, right]


def tone(rng, n, bps, f=0.01, noise=0.002):
    t = np.arange(n)
    x = 0.6 * np.sin(2 * np.pi * f * t + rng.uniform(0, 6)) + 0.2 * np.sin(2 * np.pi * 3.1 * f * t) + noise * rng.standard_normal(n)
    return np.clip(np.round(x * (1 << (bps - 1))), -(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int64)


def save(name, data, pcm):
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, name + ".flac"), "wb").write(data)
    if pcm is not None:
        np.savez_compressed(os.path.join(OUT, name + ".npz"), pcm=np.asarray(pcm, np.int32))


def main():
    rng = np.random.default_rng(2024)
    # ---- mono16_fixed (+ nolength)
    x = tone(rng, 4096 * 3 + 777, 16)
    frames, at, num = [], 0, 0
    for order in (0, 3, 4):
        frames.append(write_frame(num, [x[at:at + 4096]], 16, 5, 0, [("fixed", dict(order=order, porder=order + 1))]))
        at += 4096
        num += 1
    frames.append(write_frame(num, [x[at:]], 16, 5, 0, [("fixed", dict(order=2, porder=0))], explicit_bs=16))
    save("mono16_fixed", stream(frames, 1, 16, 16000, len(x), 4096), x[:, None])
    save("nolength", stream(frames, 1, 16, 16000, 0, 4096), None)      # same samples as mono16_fixed
    # ---- stereo16_modes
    left, right = tone(rng, 1152 * 4, 16, 0.013), tone(rng, 1152 * 4, 16, 0.0131)
    right = np.clip(left + (right >> 3), -32768, 32767)
    frames = []
    for k, mode in enumerate((1, 8, 9, 10)):
        sl = slice(1152 * k, 1152 * (k + 1))
        chans = decorrelate(left[sl], right[sl], mode)
        frames.append(write_frame(k, chans, 16, 9, mode, [("fixed", dict(order=2, porder=2)), ("fixed", dict(order=1, porder=3))]))
    save("stereo16_modes", stream(frames, 2, 16, 44100, 1152 * 4, 1152), np.stack([left, right], 1))
    # ---- lpc24
    n = 1024 * 3
    left, right = tone(rng, n, 24, 0.004, 0.0005), tone(rng, n, 24, 0.0057, 0.0005)
    frames = []
    for k, order in enumerate((1, 8, 12)):
        sl = slice(1024 * k, 1024 * (k + 1))
        prec, shift = 12 + k, 9 + k
        # a stable made-up predictor: decaying alternating taps; the residual absorbs whatever it misses
        taps = [int(round((0.9 ** j) * (1 if j % 2 == 0 else -0.5) * (1 << shift) / (1 + 0.3 * order))) for j in range(order)]
        taps = [max(-(1 << (prec - 1)), min((1 << (prec - 1)) - 1, t)) for t in taps]
        kw = dict(coefs=taps, shift=shift, precision=prec, porder=k)
        frames.append(write_frame(k, [left[sl], right[sl]], 24, 10, 1, [("lpc", kw), ("lpc", dict(kw, porder=4))]))
    save("lpc24", stream(frames, 2, 24, 48000, n, 1024), np.stack([left, right], 1))
    # ---- pcm8_const_verb
    a = tone(rng, 192, 8, 0.05, 0.02)
    b = np.full(200, -37, np.int64)
    c = tone(rng, 192, 8, 0.02, 0.05)
    frames = [write_frame(0, [a], 8, 4, 0, [("verbatim", {})]),
              write_frame(1, [b], 8, 4, 0, [("constant", {})], explicit_bs=8),
              write_frame(2, [c], 8, 4, 0, [("fixed", dict(order=1, porder=1))])]
    save("pcm8_const_verb", stream(frames, 1, 8, 8000, 192 + 200 + 192, 192), np.concatenate([a, b, c])[:, None])
    # ---- wasted_escape
    left = (tone(rng, 2048, 16, 0.02) >> 2) << 2
    right = (tone(rng, 2048, 16, 0.017) >> 4) << 4
    right[700:764] = rng.integers(-30000, 30000, 64) >> 4 << 4      # a burst: its partition is cheaper raw
    frames = [write_frame(0, [left[:1024], right[:1024]], 16, 5, 1,
                          [("fixed", dict(order=2, porder=4, wasted=2)), ("fixed", dict(order=1, porder=4, wasted=4, escape=(11,)))]),
              write_frame(1, [left[1024:], right[1024:]], 16, 5, 1,
                          [("verbatim", dict(wasted=2)), ("fixed", dict(order=3, porder=2, wasted=4, escape=(0, 3)))])]
    save("wasted_escape", stream(frames, 2, 16, 16000, 2048, 1024), np.stack([left, right], 1))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
