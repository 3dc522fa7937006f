// FLAC stream decoder (host): what decoders.audio needs for LibriSpeech-shaped data, which ships as FLAC.
//
// Reference counterpart: GenericAudioDecoder over libsndfile (dali/operators/decoder/audio/generic_decoder.cc:180-206),
// whose FLAC support is libFLAC - not vendored in /root/reference.  The format is the published one (RFC 9639): a
// "fLaC" marker, metadata blocks (STREAMINFO first), then frames of up to 8 channels; every channel of a frame is one
// subframe - constant, verbatim, fixed polynomial predictor of order 0-4, or LPC of order 1-32 with quantised
// coefficients - whose residual is Rice-coded in 2^n partitions; stereo frames may carry left/side, side/right or
// mid/side instead of left/right.  Decoding is exact integer arithmetic, so the result is the encoder's input bit for
// bit; every frame header carries a CRC-8 and every frame a CRC-16, both checked.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "dali_amd_host.h"
#include "host_common.h"

namespace {

using daliamd_host::Fail;

struct BitReader {
  const uint8_t *p;
  size_t n, pos = 0;   // pos in bits
  bool ok = true;
  uint32_t Bits(int count) {   // count <= 32
    uint64_t v = 0;
    for (int got = 0; got < count;) {
      const size_t byte = pos >> 3;
      if (byte >= n) { ok = false; return 0; }
      const int avail = 8 - (int)(pos & 7), take = std::min(avail, count - got);
      v = (v << take) | ((p[byte] >> (avail - take)) & ((1u << take) - 1));
      pos += take;
      got += take;
    }
    return (uint32_t)v;
  }
  int32_t Signed(int count) {
    if (count == 0) return 0;
    const uint32_t v = Bits(count);
    return count == 32 ? (int32_t)v : (int32_t)(v << (32 - count)) >> (32 - count);
  }
  uint32_t Unary() {   // zeros in front of the next one bit
    uint32_t zeros = 0;
    for (;;) {
      const size_t byte = pos >> 3;
      if (byte >= n) { ok = false; return 0; }
      const int avail = 8 - (int)(pos & 7);
      const uint32_t rest = p[byte] & ((1u << avail) - 1);
      if (rest) {
        const int lead = avail - 1 - (31 - __builtin_clz(rest));
        zeros += lead;
        pos += lead + 1;
        return zeros;
      }
      zeros += avail;
      pos += avail;
    }
  }
  void AlignByte() { pos = (pos + 7) & ~(size_t)7; }
};

uint8_t Crc8(const uint8_t *p, size_t n) {   // polynomial x^8 + x^2 + x + 1, initial value 0
  uint8_t c = 0;
  for (size_t i = 0; i < n; i++) {
    c ^= p[i];
    for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}
uint16_t Crc16(const uint8_t *p, size_t n) {  // polynomial x^16 + x^15 + x^2 + 1, initial value 0
  static uint16_t table[256];
  static bool init = [] {
    for (int i = 0; i < 256; i++) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      table[i] = c;
    }
    return true;
  }();
  (void)init;
  uint16_t c = 0;
  for (size_t i = 0; i < n; i++) c = (uint16_t)((c << 8) ^ table[(c >> 8) ^ p[i]]);
  return c;
}

bool ReadResidual(BitReader &br, int blocksize, int order, int64_t *out) {
  const int method = (int)br.Bits(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5, escape = method == 0 ? 15 : 31;
  const int porder = (int)br.Bits(4);
  const int parts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
  int i = order;
  for (int part = 0; part < parts; part++) {
    int count = (blocksize >> porder) - (part == 0 ? order : 0);
    if (count < 0) return false;
    const int k = (int)br.Bits(pbits);
    if (k == escape) {
      const int raw = (int)br.Bits(5);
      for (; count > 0; count--) out[i++] = br.Signed(raw);
    } else {
      for (; count > 0; count--) {
        const uint32_t q = br.Unary();
        const uint32_t u = (q << k) | (k ? br.Bits(k) : 0);
        out[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (!br.ok) return false;
  }
  return i == blocksize;
}

bool ReadSubframe(BitReader &br, int blocksize, int bps, int64_t *out) {
  if (br.Bits(1) != 0) return false;
  const int type = (int)br.Bits(6);
  int wasted = 0;
  if (br.Bits(1)) wasted = (int)br.Unary() + 1;
  bps -= wasted;
  if (bps <= 0 || bps > 33) return false;
  auto sample = [&](int bits) -> int64_t {   // up to 33 bits (the side channel of 32-bit stereo)
    if (bits <= 32) return br.Signed(bits);
    const int64_t hi = br.Signed(bits - 32);
    return (hi << 32) | br.Bits(32);
  };
  if (type == 0) {
    const int64_t v = sample(bps);
    for (int i = 0; i < blocksize; i++) out[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < blocksize; i++) out[i] = sample(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; i++) out[i] = sample(bps);
    if (!ReadResidual(br, blocksize, order, out)) return false;
    for (int i = order; i < blocksize; i++) {
      int64_t pred = 0;
      switch (order) {
        case 1: pred = out[i - 1]; break;
        case 2: pred = 2 * out[i - 1] - out[i - 2]; break;
        case 3: pred = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
        case 4: pred = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
        default: break;
      }
      out[i] += pred;
    }
  } else if (type >= 32) {
    const int order = type - 31;
    if (order > blocksize) return false;
    for (int i = 0; i < order; i++) out[i] = sample(bps);
    const int precision = (int)br.Bits(4) + 1;
    if (precision == 16) return false;
    const int shift = br.Signed(5);
    if (shift < 0) return false;
    int32_t coef[32];
    for (int j = 0; j < order; j++) coef[j] = br.Signed(precision);
    if (!ReadResidual(br, blocksize, order, out)) return false;
    for (int i = order; i < blocksize; i++) {
      int64_t acc = 0;
      for (int j = 0; j < order; j++) acc += (int64_t)coef[j] * out[i - 1 - j];
      out[i] += acc >> shift;
    }
  } else {
    return false;   // reserved subframe type
  }
  if (wasted)
    for (int i = 0; i < blocksize; i++) out[i] = out[i] * ((int64_t)1 << wasted);
  return br.ok;
}

struct StreamInfo { int channels = 0, bps = 0; double rate = 0; int64_t total = 0; size_t first_frame = 0; int max_block = 0; };

// A leading ID3v2 tag (libFLAC / libsndfile skip it): "ID3", version (2 bytes), flags, size as four 7-bit bytes;
// flag 0x10 = a 10-byte footer behind the tag.
size_t SkipId3v2(const uint8_t *p, size_t n) {
  size_t pos = 0;
  while (n - pos >= 10 && !memcmp(p + pos, "ID3", 3) && !((p[pos + 6] | p[pos + 7] | p[pos + 8] | p[pos + 9]) & 0x80)) {
    const size_t len = ((size_t)p[pos + 6] << 21) | ((size_t)p[pos + 7] << 14) | ((size_t)p[pos + 8] << 7) | p[pos + 9];
    const size_t total = 10 + len + ((p[pos + 5] & 0x10) ? 10 : 0);
    if (total > n - pos) break;
    pos += total;
  }
  return pos;
}

int ParseHeader(const uint8_t *p, size_t n, StreamInfo *si) {
  const size_t lead = SkipId3v2(p, n);
  if (n - lead < 4 + 4 + 34 || memcmp(p + lead, "fLaC", 4)) return Fail("not a FLAC stream");
  size_t pos = lead + 4;
  bool last = false, have_info = false;
  while (!last) {
    if (pos + 4 > n) return Fail("FLAC: truncated metadata");
    last = (p[pos] & 0x80) != 0;
    const int type = p[pos] & 0x7f;
    const size_t len = ((size_t)p[pos + 1] << 16) | ((size_t)p[pos + 2] << 8) | p[pos + 3];
    pos += 4;
    if (pos + len > n) return Fail("FLAC: truncated metadata block");
    if (type == 0) {
      if (len < 34) return Fail("FLAC: short STREAMINFO");
      const uint8_t *b = p + pos;
      si->max_block = (b[2] << 8) | b[3];
      const uint64_t v = ((uint64_t)b[10] << 56) | ((uint64_t)b[11] << 48) | ((uint64_t)b[12] << 40) | ((uint64_t)b[13] << 32) |
                         ((uint64_t)b[14] << 24) | ((uint64_t)b[15] << 16) | ((uint64_t)b[16] << 8) | b[17];
      si->rate = (double)(v >> 44);
      si->channels = (int)((v >> 41) & 7) + 1;
      si->bps = (int)((v >> 36) & 31) + 1;
      si->total = (int64_t)(v & ((1ull << 36) - 1));
      have_info = true;
    }
    pos += len;
  }
  if (!have_info || si->rate <= 0) return Fail("FLAC: no STREAMINFO block");
  si->first_frame = pos;
  // The 36-bit sample count comes from the file: a frame is at least 10 bytes (header 5 + CRC-8, a constant subframe
  // of two bytes, CRC-16) and carries at most 65 536 samples, so `n` bytes cannot hold more than this - a crafted
  // header must not make the caller allocate terabytes.
  const int64_t most = (int64_t)((n - pos) / 10 + 1) * 65536;
  if (si->total > most) return Fail("FLAC: STREAMINFO promises %lld samples per channel, the file's %zu bytes cannot hold them",
                                    (long long)si->total, n);
  return 0;
}

// One frame at `pos`; appends blocksize * channels interleaved samples to `pcm` (when not null).  Returns the frame's
// block size, 0 at the end of the stream, -1 on a broken frame.
int DecodeFrame(const uint8_t *p, size_t n, size_t *pos, const StreamInfo &si, std::vector<int64_t> &work, int32_t *pcm,
                int64_t written, int64_t capacity) {
  if (*pos + 2 > n) return 0;
  BitReader br{p + *pos, n - *pos};
  // no sync code behind the first frame: trailing data (an ID3v1 tag, padding) - the stream ends here, as in libFLAC
  if (br.Bits(14) != 0x3ffe) return *pos > si.first_frame ? 0 : -1;
  if (br.Bits(1) != 0) return -1;
  br.Bits(1);   // blocking strategy: only changes the meaning of the coded number
  const int bs_code = (int)br.Bits(4), sr_code = (int)br.Bits(4), ch_code = (int)br.Bits(4), ss_code = (int)br.Bits(3);
  if (br.Bits(1) != 0) return -1;
  {  // frame / sample number, coded like UTF-8 (up to 7 bytes)
    const uint32_t first = br.Bits(8);
    int extra = 0;
    if (first & 0x80) {
      for (uint32_t m = 0x40; m && (first & m); m >>= 1) extra++;
      if (extra == 0 || extra > 6) return -1;
    }
    for (int i = 0; i < extra; i++)
      if ((br.Bits(8) & 0xc0) != 0x80) return -1;
  }
  int blocksize;
  switch (bs_code) {
    case 0: return -1;
    case 1: blocksize = 192; break;
    case 6: blocksize = (int)br.Bits(8) + 1; break;
    case 7: blocksize = (int)br.Bits(16) + 1; break;
    default: blocksize = bs_code <= 5 ? 576 << (bs_code - 2) : 256 << (bs_code - 8);
  }
  if (sr_code == 12) br.Bits(8);
  else if (sr_code == 13 || sr_code == 14) br.Bits(16);
  else if (sr_code == 15) return -1;
  static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
  int bps = kBits[ss_code];
  if (bps < 0) return -1;
  if (bps == 0) bps = si.bps;
  const int channels = ch_code < 8 ? ch_code + 1 : 2;
  if (ch_code > 10 || channels != si.channels || bps != si.bps) return -1;   // (mid-stream format changes: not taken)
  if (!br.ok || (br.pos & 7)) return -1;
  const size_t hdr_bytes = br.pos >> 3;
  if (hdr_bytes + 1 > br.n || Crc8(br.p, hdr_bytes) != br.p[hdr_bytes]) return -1;
  br.Bits(8);
  work.resize((size_t)blocksize * channels);
  for (int c = 0; c < channels; c++) {
    const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
    if (!ReadSubframe(br, blocksize, bps + (side ? 1 : 0), work.data() + (size_t)c * blocksize)) return -1;
  }
  br.AlignByte();
  const size_t body = br.pos >> 3;
  if (body + 2 > br.n) return -1;
  if (Crc16(br.p, body) != (uint16_t)((br.p[body] << 8) | br.p[body + 1])) return -1;
  *pos += body + 2;
  if (pcm) {
    int64_t *a = work.data(), *b = work.data() + blocksize;
    for (int i = 0; i < blocksize && channels == 2 && ch_code >= 8; i++) {
      if (ch_code == 8) b[i] = a[i] - b[i];                 // left, side -> right = left - side
      else if (ch_code == 9) a[i] = a[i] + b[i];            // side, right -> left = side + right
      else {                                                // mid, side
        const int64_t side = b[i], mid = (a[i] * 2) | (side & 1);
        a[i] = (mid + side) >> 1;
        b[i] = (mid - side) >> 1;
      }
    }
    for (int i = 0; i < blocksize; i++) {
      if (written + i >= capacity) break;
      for (int c = 0; c < channels; c++) pcm[(written + i) * channels + c] = (int32_t)work[(size_t)c * blocksize + i];
    }
  }
  return blocksize;
}

}  // namespace

extern "C" {

int daliamdFlacProbe(const uint8_t *data, size_t size, daliamdAudioStreamInfo *info) {
  if (!data || !info) return Fail("daliamdFlacProbe: NULL argument");
  StreamInfo si;
  if (int rc = ParseHeader(data, size, &si)) return rc;
  if (si.total == 0) {   // unknown length in STREAMINFO: walk the frames
    std::vector<int64_t> work;
    size_t pos = si.first_frame;
    for (;;) {
      const int bs = DecodeFrame(data, size, &pos, si, work, nullptr, 0, 0);
      if (bs < 0) return Fail("FLAC: broken frame at byte %zu", pos);
      if (bs == 0) break;
      si.total += bs;
    }
  }
  info->channels = si.channels;
  info->bits_per_sample = si.bps;
  info->sample_rate = si.rate;
  info->frames = si.total;
  return 0;
}

int daliamdFlacDecode(const uint8_t *data, size_t size, int32_t *pcm, int64_t frames) {
  if (!data || (!pcm && frames > 0)) return Fail("daliamdFlacDecode: NULL argument");
  StreamInfo si;
  if (int rc = ParseHeader(data, size, &si)) return rc;
  std::vector<int64_t> work;
  size_t pos = si.first_frame;
  int64_t written = 0;
  while (written < frames) {
    const int bs = DecodeFrame(data, size, &pos, si, work, pcm, written, frames);
    if (bs < 0) return Fail("FLAC: broken frame at byte %zu (bad sync code, CRC mismatch or reserved field)", pos);
    if (bs == 0) break;
    written += bs;
  }
  if (written < frames) return Fail("FLAC: the stream holds %lld samples per channel, STREAMINFO promised %lld", (long long)written,
                                    (long long)frames);
  return 0;
}

}  // extern "C"
