// Host-side JPEG stream parser and Huffman entropy decoder (baseline + progressive, 8 bit).
//
// Role in the pipeline: the CPU half of the hybrid decoder.  It turns the compressed scan(s)
// into un-dequantised DCT coefficient blocks laid out exactly as the gfx950 IDCT kernel reads
// them (per component [blocks_y][blocks_x][64] int16, each block column-major), normally
// straight into pinned host memory; everything after that (dequantisation, IDCT, chroma
// upsampling, colour conversion, resize, normalise) runs on the GPU.
// Reference counterpart: ImageDecoder::ParseSample + the nvImageCodec/nvJPEG hybrid Huffman stage
// (dali/operators/imgcodec/image_decoder.h:473-500,810-815); algorithm: ITU-T T.81 Annex F/G.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "dali_amd_host.h"
#include "host_common.h"

namespace daliamd_host {

namespace {

constexpr int kLookBits = 9;

// zigzag position -> index inside a COLUMN-MAJOR block (col*8 + row); 16 guard entries
struct ZigZagT {
  uint8_t v[64 + 16];
  constexpr ZigZagT() : v() {
    const uint8_t nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    for (int i = 0; i < 64; i++) v[i] = (uint8_t)((nat[i] & 7) * 8 + (nat[i] >> 3));
    for (int i = 64; i < 80; i++) v[i] = 63;
  }
};
constexpr ZigZagT kZZ;

struct HuffTable {
  bool present = false;
  uint8_t bits[17] = {};
  uint8_t vals[256] = {};
  uint16_t look[1 << kLookBits];      // (nbits << 8) | symbol, 0 = longer than kLookBits
  int16_t fast_ac[1 << kLookBits];    // (value << 8) | (run << 4) | total_bits, 0 = none
  int32_t maxcode[18];
  int32_t valoff[17];

  // lookup = false: only the canonical-code bookkeeping and its validation (the scan analysis for the GPU decoder
  // hands on bits[] / vals[] and never decodes with these tables)
  bool Build(bool is_ac, bool lookup = true) {
    uint32_t code = 0;
    int p = 0;
    uint32_t codes[257];
    uint8_t sizes[257];
    for (int l = 1; l <= 16; l++)
      for (int i = 0; i < bits[l]; i++) {
        if (p >= 256) return false;
        sizes[p++] = (uint8_t)l;
      }
    int n = p;
    p = 0;
    for (int l = 1; l <= 16; l++) {
      valoff[l] = p - (int)code;
      for (int i = 0; i < bits[l]; i++) codes[p++] = code++;
      if (code > (1u << l)) return false;
      maxcode[l] = bits[l] ? (int32_t)code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
    if (!lookup) return true;
    memset(look, 0, sizeof(look));
    for (int i = 0; i < n; i++) {
      int l = sizes[i];
      if (l > kLookBits) continue;
      int first = codes[i] << (kLookBits - l);
      for (int j = 0; j < (1 << (kLookBits - l)); j++) look[first + j] = (uint16_t)((l << 8) | vals[i]);
    }
    memset(fast_ac, 0, sizeof(fast_ac));
    if (is_ac) {
      for (int i = 0; i < (1 << kLookBits); i++) {
        int e = look[i];
        if (!e) continue;
        int len = e >> 8, sym = e & 255;
        int run = sym >> 4, mag = sym & 15;
        if (mag && len + mag <= kLookBits) {
          int k = ((i << len) & ((1 << kLookBits) - 1)) >> (kLookBits - mag);
          int m = 1 << (mag - 1);
          if (k < m) k += (int)((~0u) << mag) + 1;
          if (k >= -128 && k <= 127) fast_ac[i] = (int16_t)((k * 256) + (run * 16) + (len + mag));
        }
      }
    }
    present = true;
    return true;
  }
};

struct BitReader {
  const uint8_t *p = nullptr, *end = nullptr;
  uint64_t buf = 0;
  int cnt = 0;
  bool marker = false;

  void Reset(const uint8_t *b, const uint8_t *e) { p = b; end = e; buf = 0; cnt = 0; marker = false; }

  inline void Refill() {
    if (!marker && end - p >= 8) {
      uint64_t w;
      memcpy(&w, p, 8);
      w = __builtin_bswap64(w);
      uint64_t t = ~w;  // byte == 0xFF  <=>  ~byte == 0
      if (!((t - 0x0101010101010101ull) & ~t & 0x8080808080808080ull)) {
        int k = (64 - cnt) >> 3;
        if (k == 8) { buf = w; } else { buf |= (w >> cnt) & ~((1ull << (64 - cnt - 8 * k)) - 1); }
        p += k;
        cnt += 8 * k;
        return;
      }
    }
    while (cnt <= 56) {
      uint32_t c = 0;
      if (!marker && p < end) {
        c = *p;
        if (c == 0xFF) {
          uint32_t c2 = p + 1 < end ? p[1] : 0xD9;
          if (c2 == 0) p += 2;
          else { marker = true; c = 0; }
        } else {
          p++;
        }
      } else {
        marker = true;
      }
      buf |= (uint64_t)c << (56 - cnt);
      cnt += 8;
    }
  }
  // at least 32 valid bits: a symbol (<= 16) and its extra bits (<= 15) without looking at the stream again
  inline void Need32() { if (cnt < 32) Refill(); }
  inline uint32_t Peek(int n) { return (uint32_t)(buf >> (64 - n)); }
  inline void Drop(int n) { buf <<= n; cnt -= n; }
  inline int Get(int n) { uint32_t v = Peek(n); Drop(n); return (int)v; }
  inline int GetBit() { int v = (int)(buf >> 63); Drop(1); return v; }
};

inline int Extend(int v, int s) { return v < (1 << (s - 1)) ? v + (int)((~0u) << s) + 1 : v; }

inline int DecodeSymbol(BitReader &br, const HuffTable &t) {
  int e = t.look[br.Peek(kLookBits)];
  if (e) { br.Drop(e >> 8); return e & 255; }
  uint32_t code = br.Peek(16);
  for (int l = kLookBits + 1; l <= 16; l++) {
    int c = (int)(code >> (16 - l));
    if (c <= t.maxcode[l]) { br.Drop(l); return t.vals[(c + t.valoff[l]) & 255]; }
  }
  br.Drop(16);
  return 0;
}

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int bx = 0, by = 0, dw = 0, dh = 0;
};

struct Decoder {
  const uint8_t *data;
  size_t size;
  size_t pos = 0;
  int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
  bool progressive = false, have_frame = false;
  Component comp[4];
  uint16_t qt[4][64];  // zigzag order as transmitted
  bool qt_present[4] = {};
  HuffTable dc[4], ac[4];
  int restart_interval = 0;
  bool jfif = false, adobe = false;
  int adobe_transform = 0, orientation = 1;
  // scan
  int ns = 0, scomp[4], std_[4], sta[4], Ss = 0, Se = 63, Ah = 0, Al = 0;
  int last_dc[4], eobrun = 0;
  int16_t *coef[4] = {};
  const daliamdJpegInfo *expect = nullptr;  // geometry the caller sized the coefficient arrays for
  bool analyze_only = false;                // stop at the first SOS and report it instead of decoding
  int num_scans = 0;
  size_t first_ecs = 0;
  // what the markers in front of SOF said (daliamdJpegParse stops there; a run that goes on to SOS reports the same)
  struct AtSof { bool jfif, adobe; int adobe_transform, orientation, restart_interval; } at_sof{false, false, 0, 1, 0};

  static int R16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

  void ParseExif(const uint8_t *p, int len) {
    if (len < 14 || memcmp(p, "Exif\0\0", 6)) return;
    const uint8_t *t = p + 6;
    uint32_t n = len - 6;
    bool le;
    if (t[0] == 'I' && t[1] == 'I') le = true; else if (t[0] == 'M' && t[1] == 'M') le = false; else return;
    auto r16 = [&](uint32_t o) -> uint32_t { return le ? (t[o] | (t[o + 1] << 8)) : ((t[o] << 8) | t[o + 1]); };
    auto r32 = [&](uint32_t o) -> uint32_t {
      return le ? (r16(o) | (r16(o + 2) << 16)) : ((r16(o) << 16) | r16(o + 2));
    };
    if (r16(2) != 42) return;
    // every offset comes from the stream: compare without wrap-around (64-bit arithmetic)
    const uint64_t off = r32(4);
    if (off + 2 > n) return;
    const uint32_t cnt = r16((uint32_t)off);
    for (uint32_t i = 0; i < cnt; i++) {
      const uint64_t e64 = off + 2 + 12ull * i;
      if (e64 + 12 > n) return;
      const uint32_t e = (uint32_t)e64;
      if (r16(e) == 0x0112) {
        uint32_t v = r16(e + 8);
        if (v >= 1 && v <= 8) orientation = (int)v;
        return;
      }
    }
  }

  int ParseSof(const uint8_t *p, int len, bool prog) {
    if (len < 6) return Fail("truncated SOF");
    if (p[0] != 8) return Fail("only 8-bit JPEG is supported (got %d-bit)", p[0]);
    height = R16(p + 1); width = R16(p + 3); ncomp = p[5];
    progressive = prog;
    if (ncomp < 1 || ncomp > 4 || len < 6 + 3 * ncomp) return Fail("bad component count %d", ncomp);
    if (width <= 0 || height <= 0) return Fail("empty image %dx%d", width, height);
    hmax = vmax = 1;
    for (int i = 0; i < ncomp; i++) {
      Component &c = comp[i];
      c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i] & 3;
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return Fail("bad sampling factors");
      hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
    }
    int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < ncomp; i++) {
      Component &c = comp[i];
      c.bx = mcux * c.h; c.by = mcuy * c.v;
      c.dw = (width * c.h + hmax - 1) / hmax; c.dh = (height * c.v + vmax - 1) / vmax;
    }
    if (expect) {
      bool same = expect->width == width && expect->height == height && expect->num_components == ncomp;
      for (int i = 0; same && i < ncomp; i++)
        same = expect->blocks_x[i] == comp[i].bx && expect->blocks_y[i] == comp[i].by;
      if (!same) return Fail("daliamdJpegDecodeCoefficients: info does not describe this stream");
    }
    have_frame = true;
    return 0;
  }

  int ParseDqt(const uint8_t *p, int len) {
    while (len > 0) {
      int pq = p[0] >> 4, tq = p[0] & 15;
      if (tq > 3) return Fail("bad DQT index");
      p++; len--;
      int need = pq ? 128 : 64;
      if (len < need) return Fail("truncated DQT");
      for (int i = 0; i < 64; i++) qt[tq][i] = pq ? (uint16_t)R16(p + 2 * i) : p[i];
      p += need; len -= need;
      qt_present[tq] = true;
    }
    return 0;
  }

  int ParseDht(const uint8_t *p, int len) {
    while (len > 0) {
      if (len < 17) return Fail("truncated DHT");
      int tc = p[0] >> 4, th = p[0] & 15;
      if (tc > 1 || th > 3) return Fail("bad DHT index");
      HuffTable &t = tc ? ac[th] : dc[th];
      int count = 0;
      t.bits[0] = 0;
      for (int i = 1; i <= 16; i++) { t.bits[i] = p[i]; count += p[i]; }
      p += 17; len -= 17;
      if (count > 256 || count > len) return Fail("bad DHT counts");
      memset(t.vals, 0, sizeof(t.vals));
      memcpy(t.vals, p, count);
      p += count; len -= count;
      if (!t.Build(tc == 1, !analyze_only)) return Fail("invalid Huffman table");
    }
    return 0;
  }

  // ---- block decoders ----
  inline void BlockSeq(BitReader &br, int16_t *blk, int s_idx) {
    int ci = scomp[s_idx];
    const HuffTable &dt = dc[std_[s_idx]], &at = ac[sta[s_idx]];
    br.Need32();
    int s = DecodeSymbol(br, dt) & 15;  // a (corrupt) table may list categories > 15: keep the bit count sane (the
                                        // GPU decoder's table entries mask the symbol the same way)
    if (s) s = Extend(br.Get(s), s);
    last_dc[ci] += s;
    blk[0] = (int16_t)last_dc[ci];
    for (int k = 1; k < 64;) {
      br.Need32();
      int fa = at.fast_ac[br.Peek(kLookBits)];
      if (fa) {
        k += (fa >> 4) & 15;
        br.Drop(fa & 15);
        blk[kZZ.v[k++]] = (int16_t)(fa >> 8);
        continue;
      }
      int rs = DecodeSymbol(br, at);
      int r = rs >> 4;
      s = rs & 15;
      if (s) {
        k += r;
        blk[kZZ.v[k++]] = (int16_t)Extend(br.Get(s), s);  // >= 16 bits left after the symbol
      } else {
        if (r != 15) break;
        k += 16;
      }
    }
  }
  inline void BlockDcFirst(BitReader &br, int16_t *blk, int s_idx) {
    int ci = scomp[s_idx];
    br.Refill();
    int s = DecodeSymbol(br, dc[std_[s_idx]]) & 15;
    if (s) { br.Refill(); s = Extend(br.Get(s), s); }
    last_dc[ci] += s;
    blk[0] = (int16_t)(last_dc[ci] * (1 << Al));
  }
  inline void BlockDcRefine(BitReader &br, int16_t *blk) {
    br.Refill();
    if (br.GetBit()) blk[0] |= (int16_t)(1 << Al);
  }
  inline void BlockAcFirst(BitReader &br, int16_t *blk) {
    if (eobrun > 0) { eobrun--; return; }
    const HuffTable &at = ac[sta[0]];
    for (int k = Ss; k <= Se; k++) {
      br.Refill();
      int rs = DecodeSymbol(br, at);
      int r = rs >> 4, s = rs & 15;
      if (s) {
        k += r;
        br.Refill();
        blk[kZZ.v[k]] = (int16_t)(Extend(br.Get(s), s) * (1 << Al));
      } else if (r == 15) {
        k += 15;
      } else {
        eobrun = 1 << r;
        if (r) { br.Refill(); eobrun += br.Get(r); }
        eobrun--;
        break;
      }
    }
  }
  inline void RefineNonZero(BitReader &br, int16_t *c, int p1, int m1) {
    br.Refill();
    if (br.GetBit()) {
      if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
    }
  }
  inline void BlockAcRefine(BitReader &br, int16_t *blk) {
    const HuffTable &at = ac[sta[0]];
    int p1 = 1 << Al, m1 = (int)((~0u) << Al);
    int k = Ss;
    if (eobrun == 0) {
      for (; k <= Se; k++) {
        br.Refill();
        int rs = DecodeSymbol(br, at);
        int r = rs >> 4, s = rs & 15;
        if (s) {
          br.Refill();
          s = br.GetBit() ? p1 : m1;
        } else if (r != 15) {
          eobrun = 1 << r;
          if (r) { br.Refill(); eobrun += br.Get(r); }
          break;
        }
        do {
          int16_t *c = blk + kZZ.v[k];
          if (*c != 0) {
            RefineNonZero(br, c, p1, m1);
          } else if (--r < 0) {
            break;
          }
          k++;
        } while (k <= Se);
        if (s) blk[kZZ.v[k]] = (int16_t)s;
      }
    }
    if (eobrun > 0) {
      for (; k <= Se; k++) {
        int16_t *c = blk + kZZ.v[k];
        if (*c != 0) RefineNonZero(br, c, p1, m1);
      }
      eobrun--;
    }
  }

  template <int MODE>  // 0 seq, 1 dc first, 2 dc refine, 3 ac first, 4 ac refine
  inline void Block(BitReader &br, int16_t *blk, int s_idx) {
    if (MODE == 0) BlockSeq(br, blk, s_idx);
    else if (MODE == 1) BlockDcFirst(br, blk, s_idx);
    else if (MODE == 2) BlockDcRefine(br, blk);
    else if (MODE == 3) BlockAcFirst(br, blk);
    else BlockAcRefine(br, blk);
  }

  void Restart(BitReader &br) {
    const uint8_t *q = br.p;
    const uint8_t *e = data + size;
    while (q + 1 < e) {
      if (q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) { q += 2; break; }
      if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) break;
      q++;
    }
    br.Reset(q, e);
    memset(last_dc, 0, sizeof(last_dc));
    eobrun = 0;
  }

  template <int MODE>
  void ScanLoop(BitReader &br) {
    int left = restart_interval;
    if (ns == 1) {
      const Component &c = comp[scomp[0]];
      int bw = (c.dw + 7) / 8, bh = (c.dh + 7) / 8;
      int16_t *base = coef[scomp[0]];
      for (int y = 0; y < bh; y++) {
        int16_t *row = base + (size_t)y * c.bx * 64;
        for (int x = 0; x < bw; x++) {
          if (restart_interval) { if (left == 0) { Restart(br); left = restart_interval; } left--; }
          Block<MODE>(br, row + x * 64, 0);
        }
      }
    } else {
      int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
      for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
          if (restart_interval) { if (left == 0) { Restart(br); left = restart_interval; } left--; }
          for (int s = 0; s < ns; s++) {
            const Component &c = comp[scomp[s]];
            int16_t *base = coef[scomp[s]];
            for (int v = 0; v < c.v; v++)
              for (int h = 0; h < c.h; h++)
                Block<MODE>(br, base + ((size_t)(my * c.v + v) * c.bx + (mx * c.h + h)) * 64, s);
          }
        }
    }
  }

  int DecodeScan() {
    for (int s = 0; s < ns; s++) {
      bool need_dc = !progressive || Ss == 0, need_ac = !progressive || Ss > 0;
      if (need_dc && !(progressive && Ah) && !dc[std_[s]].present) return Fail("missing DC Huffman table");
      if (need_ac && !ac[sta[s]].present) return Fail("missing AC Huffman table");
    }
    if (progressive && Ss > 0 && ns != 1) return Fail("interleaved progressive AC scan");
    BitReader br;
    br.Reset(data + pos, data + size);
    memset(last_dc, 0, sizeof(last_dc));
    eobrun = 0;
    if (!progressive) ScanLoop<0>(br);
    else if (Ss == 0) { if (Ah == 0) ScanLoop<1>(br); else ScanLoop<2>(br); }
    else { if (Ah == 0) ScanLoop<3>(br); else ScanLoop<4>(br); }
    // resynchronise on the next marker
    const uint8_t *q = br.p;
    const uint8_t *e = data + size;
    // the reader may have buffered bytes ahead of the consumed bits: step back over them
    q -= std::min<ptrdiff_t>(br.cnt / 8, q - (data + pos));
    while (q + 1 < e) {
      if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7)) break;
      q++;
    }
    pos = q - data;
    return 0;
  }

  // headers_only: stop after SOF (+ the APPn seen before it)
  int Run(bool headers_only) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return Fail("not a JPEG stream (no SOI)");
    pos = 2;
    for (;;) {
      while (pos < size && data[pos] != 0xFF) pos++;
      while (pos < size && data[pos] == 0xFF) pos++;
      if (pos >= size) break;
      int m = data[pos++];
      if (m == 0xD9) break;
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
      if (pos + 2 > size) break;
      int len = R16(data + pos);
      if (len < 2 || pos + len > size) break;
      const uint8_t *p = data + pos + 2;
      int plen = len - 2;
      pos += len;
      int rc = 0;
      switch (m) {
        case 0xC0: case 0xC1: case 0xC2:
          if (have_frame) return Fail("multiple SOF markers");
          rc = ParseSof(p, plen, m == 0xC2);
          at_sof = AtSof{jfif, adobe, adobe_transform, orientation, restart_interval};
          if (!rc && headers_only) return 0;
          break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE:
        case 0xCF:
          return Fail("unsupported JPEG process (SOF%d: lossless/arithmetic/hierarchical)", m - 0xC0);
        case 0xC4: if (!headers_only) rc = ParseDht(p, plen); break;
        case 0xDB: if (!headers_only) rc = ParseDqt(p, plen); break;
        case 0xDD: if (plen >= 2) restart_interval = R16(p); break;
        case 0xE0: if (plen >= 5 && !memcmp(p, "JFIF", 5)) jfif = true; break;
        case 0xE1: ParseExif(p, plen); break;
        case 0xEE: if (plen >= 12 && !memcmp(p, "Adobe", 5)) { adobe = true; adobe_transform = p[11]; } break;
        case 0xDA: {
          if (!have_frame) return Fail("SOS before SOF");
          if (plen < 1) return Fail("truncated SOS");
          ns = p[0];
          if (ns < 1 || ns > ncomp || plen < 1 + 2 * ns + 3) return Fail("bad SOS");
          for (int i = 0; i < ns; i++) {
            int found = -1;
            for (int j = 0; j < ncomp; j++) if (comp[j].id == p[1 + 2 * i]) found = j;
            if (found < 0) return Fail("SOS references an unknown component");
            scomp[i] = found; std_[i] = (p[2 + 2 * i] >> 4) & 3; sta[i] = p[2 + 2 * i] & 3;
          }
          Ss = p[1 + 2 * ns]; Se = p[2 + 2 * ns]; Ah = p[3 + 2 * ns] >> 4; Al = p[3 + 2 * ns] & 15;
          if (!progressive) { Ss = 0; Se = 63; Ah = Al = 0; }
          if (Se > 63 || Ss > Se || Al > 13) return Fail("bad spectral selection / approximation");
          num_scans++;
          if (analyze_only) { first_ecs = pos; return 0; }
          rc = DecodeScan();
          break;
        }
        default: break;
      }
      if (rc) return rc;
    }
    if (!have_frame) return Fail("no SOF marker found");
    return 0;
  }
};

void FillInfo(const Decoder &d, daliamdJpegInfo *info) {
  memset(info, 0, sizeof(*info));
  info->width = d.width; info->height = d.height; info->num_components = d.ncomp;
  info->progressive = d.progressive; info->hmax = d.hmax; info->vmax = d.vmax;
  info->orientation = d.at_sof.orientation; info->restart_interval = d.at_sof.restart_interval;
  for (int i = 0; i < d.ncomp; i++) {
    info->h_samp[i] = d.comp[i].h; info->v_samp[i] = d.comp[i].v;
    info->blocks_x[i] = d.comp[i].bx; info->blocks_y[i] = d.comp[i].by;
    info->down_w[i] = d.comp[i].dw; info->down_h[i] = d.comp[i].dh;
    info->coef_elems[i] = (int64_t)d.comp[i].bx * d.comp[i].by * 64;
  }
  if (d.ncomp == 1) info->color = 0;
  else if (d.ncomp == 3) {
    bool rgb;
    if (d.at_sof.jfif) rgb = false;
    else if (d.at_sof.adobe) rgb = d.at_sof.adobe_transform == 0;
    else rgb = d.comp[0].id == 'R' && d.comp[1].id == 'G' && d.comp[2].id == 'B';
    info->color = rgb ? 2 : 1;
  } else if (d.ncomp == 4) {
    // four components: CMYK, or YCCK when the Adobe marker says transform 2 (jdapimin.c default_decompress_parms);
    // samples written by Adobe software are stored inverted (flag 8)
    info->color = (d.at_sof.adobe && d.at_sof.adobe_transform == 2 ? 4 : 3) | (d.at_sof.adobe ? 8 : 0);
  } else {
    info->color = -1;
  }
}

}  // namespace
}  // namespace daliamd_host

extern "C" {

int daliamdJpegParse(const uint8_t *data, size_t size, daliamdJpegInfo *info) {
  using namespace daliamd_host;
  if (!data || !info) return Fail("daliamdJpegParse: NULL argument");
  Decoder d;
  d.data = data; d.size = size;
  int rc = d.Run(true);
  if (rc) return rc;
  // colour-space markers (APP14) may legally follow SOF only in exotic files; the common
  // APPn-before-SOF order is what the header probe covers.
  FillInfo(d, info);
  return 0;
}

namespace daliamd_host {
namespace {
// What the GPU entropy decoder needs to know about the (first) scan: eligibility, MCU structure, DHT / DQT contents,
// where the entropy-coded segment starts.  `d` has run up to the SOS header (analyze_only).
void FillScan(const Decoder &d, daliamdJpegScan *scan) {
  memset(scan, 0, sizeof(*scan));
  if (d.num_scans != 1 || d.progressive || d.ns != d.ncomp || (d.ncomp != 1 && d.ncomp != 3)) return;  // not eligible
  for (int s = 0; s < d.ns; s++)
    if (d.scomp[s] != s) return;  // components out of order: leave it to the host decoder
  int bpm = 0;
  for (int c = 0; c < d.ncomp; c++) {
    if (d.std_[c] > 1 || d.sta[c] > 1) return;  // the kernel keeps two DC + two AC tables (baseline limit)
    if (!d.dc[d.std_[c]].present || !d.ac[d.sta[c]].present || !d.qt_present[d.comp[c].tq]) return;
    for (int v = 0; v < d.comp[c].v; v++)
      for (int h = 0; h < d.comp[c].h; h++) {
        if (bpm >= 10) return;
        scan->comp_of_block[bpm] = (uint8_t)c; scan->h_of_block[bpm] = (uint8_t)h; scan->v_of_block[bpm] = (uint8_t)v;
        bpm++;
      }
    scan->dc_sel[c] = (uint8_t)d.std_[c]; scan->ac_sel[c] = (uint8_t)d.sta[c];
    for (int k = 0; k < 64; k++) scan->quant[c][kZZ.v[k]] = d.qt[d.comp[c].tq][k];
  }
  for (int t = 0; t < 4; t++) {
    if (d.dc[t].present) { memcpy(scan->dc_bits[t], d.dc[t].bits + 1, 16); memcpy(scan->dc_vals[t], d.dc[t].vals, 256); }
    if (d.ac[t].present) { memcpy(scan->ac_bits[t], d.ac[t].bits + 1, 16); memcpy(scan->ac_vals[t], d.ac[t].vals, 256); }
  }
  scan->blocks_per_mcu = bpm;
  scan->mcus_x = (d.width + 8 * d.hmax - 1) / (8 * d.hmax);
  scan->mcus_y = (d.height + 8 * d.vmax - 1) / (8 * d.vmax);
  scan->restart_interval = d.restart_interval;
  scan->ecs_offset = (int64_t)d.first_ecs;
  // Everything behind the SOS header, cut at the LAST end-of-image marker of the file (a backwards look that ends after
  // two bytes on nearly every file): what follows an EOI - padding, an appended thumbnail's tail, trailing data - is not
  // scan data, and without the cut it would be uploaded every iteration and held by the encoded-stream cache.
  // Still an upper bound: the un-stuffing kernel finds where the scan really ends (an EOI further in front, if there is one).
  size_t end = d.size;
  for (size_t p = d.size; p >= d.first_ecs + 2; p--)
    if (d.data[p - 2] == 0xFF && d.data[p - 1] == 0xD9) { end = p; break; }   // (the marker stays in: it ends the segment)
  scan->ecs_length = (int64_t)(end - d.first_ecs);
  scan->length_is_upper_bound = 1;
  scan->eligible = 1;
}
}  // namespace
}  // namespace daliamd_host

int daliamdJpegAnalyzeScan(const uint8_t *data, size_t size, const daliamdJpegInfo *info, daliamdJpegScan *scan) {
  using namespace daliamd_host;
  if (!data || !info || !scan) return Fail("daliamdJpegAnalyzeScan: NULL argument");
  memset(scan, 0, sizeof(*scan));
  Decoder d;
  d.data = data; d.size = size;
  d.analyze_only = true;
  int rc = d.Run(false);
  if (rc) return rc;
  FillScan(d, scan);
  if (!scan->eligible) return 0;
  // entropy-coded segment: from the end of the SOS header to the next real marker (RSTn belong to the segment)
  size_t q = d.first_ecs;
  bool rst_seen = false;
  while (q + 1 < size) {  // memchr: about one byte in 256 is 0xFF, the rest is skipped at memory speed
    const void *hit = memchr(data + q, 0xFF, size - 1 - q);
    if (!hit) { q = size; break; }
    q = (size_t)(static_cast<const uint8_t *>(hit) - data);
    const uint8_t m = data[q + 1];
    if (m >= 0xD0 && m <= 0xD7) rst_seen = true;
    else if (m != 0 && m != 0xFF) break;
    q++;
  }
  if (q + 1 >= size) q = size;
  if (rst_seen && d.restart_interval == 0) { scan->eligible = 0; return 0; }  // RST without DRI: host path
  scan->ecs_length = (int64_t)(q - d.first_ecs);
  scan->length_is_upper_bound = 0;
  return 0;
}

int daliamdJpegAnalyzeHeader(const uint8_t *data, size_t size, daliamdJpegInfo *info, daliamdJpegScan *scan) {
  using namespace daliamd_host;
  if (!data || !info || !scan) return Fail("daliamdJpegAnalyzeHeader: NULL argument");
  memset(scan, 0, sizeof(*scan));
  Decoder d;
  d.data = data; d.size = size;
  d.analyze_only = true;
  int rc = d.Run(false);
  if (rc) return rc;
  FillInfo(d, info);
  if (d.num_scans == 1 && info->num_components != 4) FillScan(d, scan);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Baseline re-encoding of decoded coefficients (round 6).  The GPU entropy decoder takes baseline streams with one
// interleaved scan; a progressive (multi-scan, spectral-selection / successive-approximation) stream holds the SAME
// quantised coefficients in another order.  Writing them out again as one sequential Huffman scan - what `jpegtran` does,
// lossless by construction - gives a stream the GPU decodes to the identical blocks, so the encoded-stream cache can keep
// such a file resident in that form: its host decode (4-5 ms of CPU per ImageNet-sized image, every epoch) is paid once.
// Code tables: the typical tables of ITU-T T.81 Annex K.3 (what libjpeg writes by default) - one table set for every
// re-encoded stream, shared with ordinary files in the decoder's table store.
namespace daliamd_host {
namespace {
const uint8_t kStdDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kStdDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kStdDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kStdAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kStdAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kStdAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kStdAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable {
  uint16_t code[256];
  uint8_t size[256];   // 0: the symbol has no code
  EncTable(const uint8_t *bits16, const uint8_t *vals, int nvals) {
    memset(code, 0, sizeof(code));
    memset(size, 0, sizeof(size));
    uint32_t c = 0;
    int p = 0;
    for (int l = 1; l <= 16; l++) {
      for (int i = 0; i < bits16[l - 1] && p < nvals; i++, p++) { code[vals[p]] = (uint16_t)c++; size[vals[p]] = (uint8_t)l; }
      c <<= 1;
    }
  }
};

struct BitWriter {
  uint8_t *p, *end;
  uint64_t acc = 0;
  int n = 0;          // bits held in acc (< 32 between calls)
  bool overflow = false;
  BitWriter(uint8_t *b, uint8_t *e) : p(b), end(e) {}
  inline void Put(uint32_t bits, int len) {   // len <= 26
    acc = (acc << len) | (bits & ((1u << len) - 1));
    n += len;
    while (n >= 8) {
      const uint8_t byte = (uint8_t)(acc >> (n - 8));
      n -= 8;
      if (end - p < 2) { overflow = true; return; }
      *p++ = byte;
      if (byte == 0xFF) *p++ = 0;   // byte stuffing (T.81 F.1.2.3)
    }
  }
  void Flush() {   // pad the last byte with 1-bits
    if (n > 0) Put((1u << (8 - n)) - 1, 8 - n);
  }
};

inline int BitSize(int v) {   // category of |v| (T.81 Table F.1 / F.2)
  v = v < 0 ? -v : v;
  return v ? 32 - __builtin_clz((unsigned)v) : 0;
}
}  // namespace
}  // namespace daliamd_host

int daliamdJpegEncodeBaselineScan(const daliamdJpegInfo *info, const int16_t *const coef[4], const uint16_t *quant, uint8_t *out,
                                  size_t capacity, size_t *length, daliamdJpegScan *scan) {
  using namespace daliamd_host;
  if (!info || !coef || !quant || !out || !length || !scan) return Fail("daliamdJpegEncodeBaselineScan: NULL argument");
  const int nc = info->num_components;
  if (nc != 1 && nc != 3) return Fail("daliamdJpegEncodeBaselineScan: %d components (1 or 3)", nc);
  static const EncTable dc_lum(kStdDcLumBits, kStdDcVals, 12), dc_chr(kStdDcChrBits, kStdDcVals, 12);
  static const EncTable ac_lum(kStdAcLumBits, kStdAcLumVals, 162), ac_chr(kStdAcChrBits, kStdAcChrVals, 162);
  memset(scan, 0, sizeof(*scan));
  int bpm = 0;
  for (int c = 0; c < nc; c++) {
    if (!coef[c]) return Fail("daliamdJpegEncodeBaselineScan: coef[%d] is NULL", c);
    const int H = nc == 1 ? 1 : info->h_samp[c], V = nc == 1 ? 1 : info->v_samp[c];
    for (int v = 0; v < V; v++)
      for (int h = 0; h < H; h++) {
        if (bpm >= 10) return Fail("daliamdJpegEncodeBaselineScan: more than 10 blocks per MCU");
        scan->comp_of_block[bpm] = (uint8_t)c; scan->h_of_block[bpm] = (uint8_t)h; scan->v_of_block[bpm] = (uint8_t)v;
        bpm++;
      }
    scan->dc_sel[c] = scan->ac_sel[c] = (uint8_t)(c == 0 ? 0 : 1);
    memcpy(scan->quant[c], quant + 64 * c, 128);
  }
  memcpy(scan->dc_bits[0], kStdDcLumBits, 16); memcpy(scan->dc_vals[0], kStdDcVals, 12);
  memcpy(scan->ac_bits[0], kStdAcLumBits, 16); memcpy(scan->ac_vals[0], kStdAcLumVals, 162);
  if (nc == 3) {
    memcpy(scan->dc_bits[1], kStdDcChrBits, 16); memcpy(scan->dc_vals[1], kStdDcVals, 12);
    memcpy(scan->ac_bits[1], kStdAcChrBits, 16); memcpy(scan->ac_vals[1], kStdAcChrVals, 162);
  }
  const int hmax = nc == 1 ? 1 : info->hmax, vmax = nc == 1 ? 1 : info->vmax;
  scan->blocks_per_mcu = bpm;
  scan->mcus_x = (info->width + 8 * hmax - 1) / (8 * hmax);
  scan->mcus_y = (info->height + 8 * vmax - 1) / (8 * vmax);
  for (int c = 0; c < nc; c++) {   // the arrays must hold the MCU-padded grid (daliamdJpegDecodeCoefficients' layout)
    const int H = nc == 1 ? 1 : info->h_samp[c], V = nc == 1 ? 1 : info->v_samp[c];
    if (info->blocks_x[c] < scan->mcus_x * H || info->blocks_y[c] < scan->mcus_y * V)
      return Fail("daliamdJpegEncodeBaselineScan: component %d holds %d x %d blocks, the MCU grid needs %d x %d", c,
                  info->blocks_x[c], info->blocks_y[c], scan->mcus_x * H, scan->mcus_y * V);
  }
  BitWriter bw(out, out + capacity);
  int pred[4] = {0, 0, 0, 0};
  for (int my = 0; my < scan->mcus_y; my++)
    for (int mx = 0; mx < scan->mcus_x; mx++)
      for (int b = 0; b < bpm; b++) {
        const int c = scan->comp_of_block[b];
        const int H = nc == 1 ? 1 : info->h_samp[c], V = nc == 1 ? 1 : info->v_samp[c];
        const int by = my * V + scan->v_of_block[b], bx = mx * H + scan->h_of_block[b];
        const int16_t *blk = coef[c] + ((size_t)by * info->blocks_x[c] + bx) * 64;
        const EncTable &dct = c == 0 ? dc_lum : dc_chr, &act = c == 0 ? ac_lum : ac_chr;
        // DC difference (F.1.2.1)
        const int diff = blk[0] - pred[c];
        pred[c] = blk[0];
        const int ds = BitSize(diff);
        if (ds > 11) return Fail("daliamdJpegEncodeBaselineScan: DC difference %d outside the baseline range", diff);
        bw.Put(dct.code[ds], dct.size[ds]);
        if (ds) bw.Put((uint32_t)(diff < 0 ? diff - 1 : diff), ds);
        // AC coefficients in zig-zag order (F.1.2.2)
        int run = 0;
        for (int k = 1; k < 64; k++) {
          const int v = blk[kZZ.v[k]];
          if (v == 0) { run++; continue; }
          while (run > 15) { bw.Put(act.code[0xF0], act.size[0xF0]); run -= 16; }
          const int sz = BitSize(v);
          if (sz > 10) return Fail("daliamdJpegEncodeBaselineScan: AC coefficient %d outside the baseline range", v);
          const int sym = (run << 4) | sz;
          bw.Put(act.code[sym], act.size[sym]);
          bw.Put((uint32_t)(v < 0 ? v - 1 : v), sz);
          run = 0;
        }
        if (run > 0) bw.Put(act.code[0], act.size[0]);
        if (bw.overflow) return Fail("daliamdJpegEncodeBaselineScan: the output buffer (%zu bytes) is too small", capacity);
      }
  bw.Flush();
  if (bw.overflow) return Fail("daliamdJpegEncodeBaselineScan: the output buffer (%zu bytes) is too small", capacity);
  *length = (size_t)(bw.p - out);
  scan->ecs_offset = 0;
  scan->ecs_length = (int64_t)*length;
  scan->length_is_upper_bound = 0;
  scan->restart_interval = 0;
  scan->eligible = 1;
  return 0;
}

int daliamdJpegDecodeCoefficients(const uint8_t *data, size_t size, const daliamdJpegInfo *info,
                                  int16_t *const coef[4], uint16_t *quant) {
  using namespace daliamd_host;
  if (!data || !info || !coef || !quant) return Fail("daliamdJpegDecodeCoefficients: NULL argument");
  if (info->num_components != 1 && info->num_components != 3 && info->num_components != 4)
    return Fail("JPEG with %d components is not supported", info->num_components);
  Decoder d;
  d.data = data; d.size = size;
  d.expect = info;
  for (int c = 0; c < info->num_components; c++) {
    if (!coef[c]) return Fail("daliamdJpegDecodeCoefficients: coef[%d] is NULL", c);
    d.coef[c] = coef[c];
    memset(coef[c], 0, sizeof(int16_t) * (size_t)info->coef_elems[c]);
  }
  int rc = d.Run(false);
  if (rc) return rc;
  for (int c = 0; c < d.ncomp; c++) {
    if (!d.qt_present[d.comp[c].tq]) return Fail("missing quantisation table %d", d.comp[c].tq);
    for (int k = 0; k < 64; k++) quant[c * 64 + kZZ.v[k]] = d.qt[d.comp[c].tq][k];
  }
  return 0;
}

}  // extern "C"
