// Host decoders for the loss-less container formats decoders.image also accepts next to JPEG: PNG, BMP, PNM
// (reference: "Supported formats: JPEG, JPEG 2000, TIFF, PNG, BMP, PNM, PPM, PGM, PBM, WebP",
// dali/operators/imgcodec/decoder_schema.cc:149; decoded there by nvImageCodec's libpng / OpenCV back ends on the
// CPU for both the cpu and the mixed operator).  They are not on the hot path - ImageNet holds one PNG - so this is a
// plain, careful host implementation: the mixed operator decodes such samples on its thread pool into page-locked
// memory and uploads them next to the GPU-decoded JPEGs.  Output is always 8-bit RGB (gray is replicated, alpha is
// dropped, 16-bit samples keep their high byte) - the same pixels Pillow's Image.convert("RGB") gives for 8-bit
// sources, which is what the tests pin.
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "dali_amd_host.h"
#include "host_common.h"

namespace daliamd_host {
namespace {

struct Window { int y0, x0, h, w; };

inline uint32_t Be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t Le32(const uint8_t *p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
inline uint32_t Le16(const uint8_t *p) { return ((uint32_t)p[1] << 8) | p[0]; }

// ------------------------------------------------------------------------------------------------ PNG
const uint8_t kPngSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

struct PngHeader {
  uint32_t width = 0, height = 0;
  int depth = 0, color = 0, interlace = 0;
  int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
};

int ParsePngHeader(const uint8_t *d, size_t n, PngHeader *h) {
  if (n < 33 || memcmp(d, kPngSig, 8) != 0) return Fail("not a PNG stream");
  if (Be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) return Fail("PNG: the first chunk is not IHDR");
  h->width = Be32(d + 16);
  h->height = Be32(d + 20);
  h->depth = d[24];
  h->color = d[25];
  h->interlace = d[28];
  if (h->width == 0 || h->height == 0 || h->width > (1u << 24) || h->height > (1u << 24)) return Fail("PNG: invalid image size");
  if (d[26] != 0 || d[27] != 0 || h->interlace > 1) return Fail("PNG: unknown compression / filter / interlace method");
  const int dp = h->depth, c = h->color;
  const bool ok = (c == 0 && (dp == 1 || dp == 2 || dp == 4 || dp == 8 || dp == 16)) ||
                  (c == 3 && (dp == 1 || dp == 2 || dp == 4 || dp == 8)) ||
                  ((c == 2 || c == 4 || c == 6) && (dp == 8 || dp == 16));
  if (!ok) return Fail("PNG: invalid colour type %d / bit depth %d", c, dp);
  return 0;
}

inline int Paeth(int a, int b, int c) {
  int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reverses the scanline filters of one (sub-)image of `rows` lines of `row_bytes` bytes in place; every line is
// preceded by its filter-type byte.  Returns the number of bytes consumed or -1.
int64_t Unfilter(uint8_t *p, int64_t avail, int rows, int64_t row_bytes, int bpp) {
  if (rows == 0 || row_bytes == 0) return 0;
  if (avail < (int64_t)rows * (row_bytes + 1)) return -1;
  const uint8_t *prev = nullptr;
  for (int y = 0; y < rows; y++) {
    const int type = p[0];
    uint8_t *cur = p + 1;
    switch (type) {
      case 0: break;
      case 1:
        for (int64_t i = bpp; i < row_bytes; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        break;
      case 2:
        if (prev) for (int64_t i = 0; i < row_bytes; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);
        break;
      case 3:
        for (int64_t i = 0; i < row_bytes; i++) {
          int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
          cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (int64_t i = 0; i < row_bytes; i++) {
          int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
          cur[i] = (uint8_t)(cur[i] + Paeth(a, b, c));
        }
        break;
      default: return -1;
    }
    prev = cur;
    p += row_bytes + 1;
  }
  return (int64_t)rows * (row_bytes + 1);
}

// sample `x` of a packed line -> 8-bit RGB
inline void PngPixel(const PngHeader &h, const uint8_t *line, uint32_t x, const uint8_t *plte, int nplte, uint8_t rgb[3]) {
  const int dp = h.depth;
  if (h.color == 3) {
    int idx = dp == 8 ? line[x] : (line[(x * dp) >> 3] >> (8 - dp - ((x * dp) & 7))) & ((1 << dp) - 1);
    if (idx >= nplte) idx = 0;  // libpng reports an error for it only in strict mode; black otherwise
    rgb[0] = plte[3 * idx]; rgb[1] = plte[3 * idx + 1]; rgb[2] = plte[3 * idx + 2];
    return;
  }
  const int ch = h.channels();
  if (dp < 8) {  // gray 1 / 2 / 4 bits: scaled to the full range (bit replication = v * 255 / max)
    const int v = (line[(x * dp) >> 3] >> (8 - dp - ((x * dp) & 7))) & ((1 << dp) - 1);
    rgb[0] = rgb[1] = rgb[2] = (uint8_t)(v * 255 / ((1 << dp) - 1));
    return;
  }
  const int step = dp / 8;  // 16-bit samples: the high byte comes first
  const uint8_t *s = line + (size_t)x * ch * step;
  if (ch <= 2) rgb[0] = rgb[1] = rgb[2] = s[0];
  else { rgb[0] = s[0]; rgb[1] = s[step]; rgb[2] = s[2 * step]; }
}

int DecodePng(const uint8_t *d, size_t n, uint8_t *out, int64_t pitch, const Window &win) {
  PngHeader h;
  if (ParsePngHeader(d, n, &h)) return 1;
  std::vector<uint8_t> idat;
  uint8_t plte[768];
  int nplte = 0;
  size_t pos = 8;
  bool end = false;
  while (!end) {
    if (pos + 12 > n) return Fail("PNG: truncated stream (no IEND chunk)");
    const uint32_t len = Be32(d + pos);
    const uint8_t *type = d + pos + 4, *body = d + pos + 8;
    if (len > n - pos - 12) return Fail("PNG: chunk exceeds the stream");
    if ((uint32_t)crc32(crc32(0, type, 4), body, len) != Be32(body + len)) return Fail("PNG: CRC error in chunk %.4s", type);
    if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!memcmp(type, "PLTE", 4)) {
      if (len % 3 != 0 || len > 768) return Fail("PNG: invalid palette");
      memcpy(plte, body, len);
      nplte = (int)(len / 3);
    } else if (!memcmp(type, "IEND", 4)) end = true;
    else if (!(type[0] & 0x20)) {
      if (memcmp(type, "IHDR", 4)) return Fail("PNG: unknown critical chunk %.4s", type);
    }
    pos += 12 + (size_t)len;
  }
  if (h.color == 3 && nplte == 0) return Fail("PNG: palette image without PLTE chunk");
  const int bits = h.channels() * h.depth;
  const int bpp = bits >= 8 ? bits / 8 : 1;
  auto line_bytes = [&](uint32_t w) { return ((int64_t)w * bits + 7) / 8; };
  // Adam7 passes: {x0, y0, dx, dy}; not interlaced = one pass over everything
  static const int kAdam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  int64_t total = 0;
  const int npass = h.interlace ? 7 : 1;
  for (int p = 0; p < npass; p++) {
    const uint32_t pw = h.interlace ? (h.width + kAdam7[p][2] - 1 - kAdam7[p][0]) / kAdam7[p][2] : h.width;
    const uint32_t ph = h.interlace ? (h.height + kAdam7[p][3] - 1 - kAdam7[p][1]) / kAdam7[p][3] : h.height;
    if (pw && ph) total += (int64_t)ph * (line_bytes(pw) + 1);
  }
  if (total > ((int64_t)1 << 33)) return Fail("PNG: image too large");
  std::vector<uint8_t> raw((size_t)total);
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit(&zs) != Z_OK) return Fail("PNG: zlib initialisation failed");
  if (idat.size() > 0xFFFFFFFFu || raw.size() > 0xFFFFFFFFu) { inflateEnd(&zs); return Fail("PNG: image too large"); }
  zs.next_in = idat.data();
  zs.avail_in = (uInt)idat.size();
  zs.next_out = raw.data();
  zs.avail_out = (uInt)raw.size();
  const int zr = inflate(&zs, Z_FINISH);
  const bool complete = zs.avail_out == 0 && (zr == Z_STREAM_END || zr == Z_OK || zr == Z_BUF_ERROR);
  inflateEnd(&zs);
  if (!complete) return Fail("PNG: corrupt or truncated image data");
  uint8_t *p = raw.data();
  int64_t left = total;
  for (int pass = 0; pass < npass; pass++) {
    const int x0 = h.interlace ? kAdam7[pass][0] : 0, y0 = h.interlace ? kAdam7[pass][1] : 0;
    const int dx = h.interlace ? kAdam7[pass][2] : 1, dy = h.interlace ? kAdam7[pass][3] : 1;
    const uint32_t pw = (h.width + dx - 1 - x0) / dx, ph = (h.height + dy - 1 - y0) / dy;
    if (!pw || !ph) continue;
    const int64_t lb = line_bytes(pw);
    const int64_t used = Unfilter(p, left, (int)ph, lb, bpp);
    if (used < 0) return Fail("PNG: invalid filter type");
    for (uint32_t r = 0; r < ph; r++) {
      const int y = y0 + (int)r * dy;
      if (y < win.y0 || y >= win.y0 + win.h) continue;
      const uint8_t *line = p + (int64_t)r * (lb + 1) + 1;
      uint8_t *orow = out + (int64_t)(y - win.y0) * pitch;
      for (uint32_t c = 0; c < pw; c++) {
        const int x = x0 + (int)c * dx;
        if (x < win.x0 || x >= win.x0 + win.w) continue;
        PngPixel(h, line, c, plte, nplte, orow + 3 * (x - win.x0));
      }
    }
    p += used;
    left -= used;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ BMP
struct BmpHeader {
  int width = 0, height = 0, bits = 0, ncolors = 0;
  bool bottom_up = true;
  uint32_t pixel_offset = 0, palette_offset = 0, palette_entry = 4, compression = 0;
  uint32_t mask[3] = {0, 0, 0};
};

int ParseBmpHeader(const uint8_t *d, size_t n, BmpHeader *h) {
  if (n < 26 || d[0] != 'B' || d[1] != 'M') return Fail("not a BMP stream");
  h->pixel_offset = Le32(d + 10);
  const uint32_t hs = Le32(d + 14);
  if (hs == 12) {  // OS/2 core header
    h->width = (int)Le16(d + 18);
    h->height = (int)Le16(d + 20);
    h->bits = (int)Le16(d + 24);
    h->palette_entry = 3;
  } else if (hs >= 40 && n >= 14 + (size_t)hs) {
    h->width = (int32_t)Le32(d + 18);
    int32_t hh = (int32_t)Le32(d + 22);
    h->bottom_up = hh > 0;
    h->height = hh > 0 ? hh : -hh;
    h->bits = (int)Le16(d + 28);
    h->compression = Le32(d + 30);
    h->ncolors = (int)Le32(d + 46);
    if (h->compression == 3) {  // BI_BITFIELDS: masks follow the 40-byte header (or sit inside a V4/V5 header)
      if (n < 14 + 40 + 12) return Fail("BMP: truncated header");
      for (int c = 0; c < 3; c++) h->mask[c] = Le32(d + 14 + 40 + 4 * c);
    }
  } else {
    return Fail("BMP: unsupported header size %u", hs);
  }
  h->palette_offset = 14 + hs + (h->compression == 3 && hs == 40 ? 12 : 0);
  if (h->width <= 0 || h->height <= 0 || h->width > (1 << 24) || h->height > (1 << 24)) return Fail("BMP: invalid image size");
  if (h->compression != 0 && h->compression != 3) return Fail("BMP: compressed (RLE / embedded) bitmaps are not supported");
  if (h->compression == 3 && h->bits != 16 && h->bits != 32) return Fail("BMP: bit fields need 16 or 32 bits per pixel");
  if (h->bits != 1 && h->bits != 4 && h->bits != 8 && h->bits != 16 && h->bits != 24 && h->bits != 32)
    return Fail("BMP: unsupported bit count %d", h->bits);
  if (h->compression == 3) {
    for (int c = 0; c < 3; c++) {  // a channel mask is one contiguous run of bits
      const uint32_t m = h->mask[c];
      if (m) {
        const uint32_t low = m & (0u - m);  // lowest set bit; m + low clears the run if it is contiguous
        if (((uint64_t)m + low) & m) return Fail("BMP: channel mask 0x%08X is not contiguous", m);
      }
    }
  }
  if (h->bits <= 8 && h->ncolors == 0) h->ncolors = 1 << h->bits;
  if (h->bits <= 8 && (h->ncolors < 0 || h->ncolors > 256)) return Fail("BMP: invalid palette size");
  return 0;
}

inline uint8_t MaskedTo8(uint32_t v, uint32_t mask) {
  if (!mask) return 0;
  int shift = 0, width = 0;
  while (!((mask >> shift) & 1)) shift++;
  while (shift + width < 32 && ((mask >> (shift + width)) & 1)) width++;
  // 64-bit: a 32-bit wide mask makes (1u << width) undefined and x * 255 overflows above 24 bits
  const uint64_t x = (v & mask) >> shift, mx = (1ull << width) - 1;
  return (uint8_t)(x * 255 / mx);  // truncating, like Pillow's BGR;15 / BGR;16 unpackers
}

int DecodeBmp(const uint8_t *d, size_t n, uint8_t *out, int64_t pitch, const Window &win) {
  BmpHeader h;
  if (ParseBmpHeader(d, n, &h)) return 1;
  const int64_t stride = (((int64_t)h.width * h.bits + 31) / 32) * 4;
  if ((uint64_t)h.pixel_offset + (uint64_t)stride * h.height > n) return Fail("BMP: truncated pixel data");
  const uint8_t *pal = d + h.palette_offset;
  if (h.bits <= 8 && (size_t)h.palette_offset + (size_t)h.ncolors * h.palette_entry > n) return Fail("BMP: truncated palette");
  uint32_t mask[3] = {h.mask[0], h.mask[1], h.mask[2]};
  if (h.bits == 16 && h.compression == 0) { mask[0] = 0x7C00; mask[1] = 0x03E0; mask[2] = 0x001F; }
  for (int y = win.y0; y < win.y0 + win.h; y++) {
    const uint8_t *row = d + h.pixel_offset + (int64_t)(h.bottom_up ? h.height - 1 - y : y) * stride;
    uint8_t *o = out + (int64_t)(y - win.y0) * pitch;
    for (int x = win.x0; x < win.x0 + win.w; x++, o += 3) {
      if (h.bits <= 8) {
        int idx = h.bits == 8 ? row[x] : (row[(x * h.bits) >> 3] >> (8 - h.bits - ((x * h.bits) & 7))) & ((1 << h.bits) - 1);
        if (idx >= h.ncolors) idx = 0;
        const uint8_t *e = pal + (size_t)idx * h.palette_entry;  // stored blue, green, red
        o[0] = e[2]; o[1] = e[1]; o[2] = e[0];
      } else if (h.bits == 24 || (h.bits == 32 && h.compression == 0)) {
        const uint8_t *s = row + (size_t)x * (h.bits / 8);
        o[0] = s[2]; o[1] = s[1]; o[2] = s[0];
      } else {
        const uint32_t v = h.bits == 16 ? Le16(row + 2 * (size_t)x) : Le32(row + 4 * (size_t)x);
        o[0] = MaskedTo8(v, mask[0]); o[1] = MaskedTo8(v, mask[1]); o[2] = MaskedTo8(v, mask[2]);
      }
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ PNM
struct PnmHeader { int kind = 0, width = 0, height = 0, maxval = 1; size_t data = 0; };

bool PnmToken(const uint8_t *d, size_t n, size_t *pos, int *value) {
  size_t p = *pos;
  for (;;) {
    while (p < n && (d[p] == ' ' || d[p] == '\t' || d[p] == '\r' || d[p] == '\n' || d[p] == '\v' || d[p] == '\f')) p++;
    if (p < n && d[p] == '#') { while (p < n && d[p] != '\n') p++; continue; }
    break;
  }
  if (p >= n || d[p] < '0' || d[p] > '9') return false;
  int64_t v = 0;
  while (p < n && d[p] >= '0' && d[p] <= '9') { v = v * 10 + (d[p] - '0'); if (v > (1 << 30)) return false; p++; }
  *pos = p;
  *value = (int)v;
  return true;
}

int ParsePnmHeader(const uint8_t *d, size_t n, PnmHeader *h) {
  if (n < 3 || d[0] != 'P' || d[1] < '1' || d[1] > '6') return Fail("not a PNM stream");
  h->kind = d[1] - '0';
  size_t pos = 2;
  if (!PnmToken(d, n, &pos, &h->width) || !PnmToken(d, n, &pos, &h->height)) return Fail("PNM: malformed header");
  if (h->kind != 1 && h->kind != 4) {
    if (!PnmToken(d, n, &pos, &h->maxval)) return Fail("PNM: malformed header");
    if (h->maxval < 1 || h->maxval > 65535) return Fail("PNM: invalid maximum value %d", h->maxval);
  }
  if (h->width <= 0 || h->height <= 0 || h->width > (1 << 24) || h->height > (1 << 24)) return Fail("PNM: invalid image size");
  h->data = pos + 1;  // exactly one white-space character separates the header from binary data
  return 0;
}

int DecodePnm(const uint8_t *d, size_t n, uint8_t *out, int64_t pitch, const Window &win) {
  PnmHeader h;
  if (ParsePnmHeader(d, n, &h)) return 1;
  const int ch = (h.kind == 3 || h.kind == 6) ? 3 : 1;
  auto scale = [&](int v) { return (uint8_t)(h.maxval == 255 ? v : (v * 255 + h.maxval / 2) / h.maxval); };
  auto put = [&](int y, int x, int c, int v) {
    if (y < win.y0 || y >= win.y0 + win.h || x < win.x0 || x >= win.x0 + win.w) return;
    uint8_t *o = out + (int64_t)(y - win.y0) * pitch + 3 * (int64_t)(x - win.x0);
    if (ch == 1) o[0] = o[1] = o[2] = (uint8_t)v; else o[c] = (uint8_t)v;
  };
  if (h.kind <= 3) {  // plain (ASCII) variants
    size_t pos = h.data - 1;
    for (int y = 0; y < h.height; y++)
      for (int x = 0; x < h.width; x++)
        for (int c = 0; c < ch; c++) {
          int v;
          if (h.kind == 1) {  // bits may follow each other without white space
            while (pos < n && d[pos] != '0' && d[pos] != '1') {
              if (d[pos] == '#') while (pos < n && d[pos] != '\n') pos++;
              else pos++;
            }
            if (pos >= n) return Fail("PNM: truncated pixel data");
            v = d[pos++] == '1' ? 0 : 255;
          } else {
            if (!PnmToken(d, n, &pos, &v) || v > h.maxval) return Fail("PNM: truncated or invalid pixel data");
            v = scale(v);
          }
          put(y, x, c, v);
        }
    return 0;
  }
  const int bytes = h.maxval > 255 ? 2 : 1;
  const int64_t row = h.kind == 4 ? (h.width + 7) / 8 : (int64_t)h.width * ch * bytes;
  if (h.data > n || (uint64_t)row * h.height > n - h.data) return Fail("PNM: truncated pixel data");
  for (int y = win.y0; y < win.y0 + win.h; y++) {
    const uint8_t *r = d + h.data + (int64_t)y * row;
    for (int x = win.x0; x < win.x0 + win.w; x++)
      for (int c = 0; c < ch; c++) {
        int v;
        if (h.kind == 4) v = ((r[x >> 3] >> (7 - (x & 7))) & 1) ? 0 : 255;
        else if (bytes == 1) v = scale(r[(int64_t)x * ch + c]);
        else { const uint8_t *s = r + ((int64_t)x * ch + c) * 2; v = scale((s[0] << 8) | s[1]); }
        put(y, x, c, v);
      }
  }
  return 0;
}

}  // namespace
}  // namespace daliamd_host

using namespace daliamd_host;

extern "C" {

int daliamdImageProbe(const uint8_t *data, size_t size, daliamdImageFormat *format, int32_t *width, int32_t *height) {
  if (!data || !format || !width || !height) return Fail("daliamdImageProbe: NULL argument");
  *format = DALIAMD_IMAGE_UNKNOWN;
  *width = *height = 0;
  if (size >= 3 && data[0] == 0xFF && data[1] == 0xD8 && data[2] == 0xFF) {
    *format = DALIAMD_IMAGE_JPEG;  // dimensions: daliamdJpegParse
    return 0;
  }
  if (size >= 8 && memcmp(data, kPngSig, 8) == 0) {
    PngHeader h;
    if (ParsePngHeader(data, size, &h)) return 1;
    *format = DALIAMD_IMAGE_PNG; *width = (int32_t)h.width; *height = (int32_t)h.height;
    return 0;
  }
  if (size >= 2 && data[0] == 'B' && data[1] == 'M') {
    BmpHeader h;
    if (ParseBmpHeader(data, size, &h)) return 1;
    *format = DALIAMD_IMAGE_BMP; *width = h.width; *height = h.height;
    return 0;
  }
  if (size >= 2 && data[0] == 'P' && data[1] >= '1' && data[1] <= '6') {
    PnmHeader h;
    if (ParsePnmHeader(data, size, &h)) return 1;
    *format = DALIAMD_IMAGE_PNM; *width = h.width; *height = h.height;
    return 0;
  }
  return Fail("unrecognised image format (supported: JPEG, PNG, BMP, PNM)");
}

int daliamdImageDecodeRgb(const uint8_t *data, size_t size, uint8_t *out, int64_t pitch, int32_t y0, int32_t x0,
                          int32_t h, int32_t w) {
  daliamdImageFormat fmt;
  int32_t W, H;
  if (!out) return Fail("daliamdImageDecodeRgb: NULL output");
  if (daliamdImageProbe(data, size, &fmt, &W, &H)) return 1;
  if (fmt == DALIAMD_IMAGE_JPEG) return Fail("daliamdImageDecodeRgb: JPEG streams take the JPEG path (daliamdJpeg*)");
  if (h == 0 && w == 0) { y0 = x0 = 0; h = H; w = W; }
  if (y0 < 0 || x0 < 0 || h <= 0 || w <= 0 || y0 + h > H || x0 + w > W)
    return Fail("daliamdImageDecodeRgb: window [%d:%d, %d:%d] does not fit the %dx%d image", y0, y0 + h, x0, x0 + w, H, W);
  if (pitch < 3 * (int64_t)w) return Fail("daliamdImageDecodeRgb: pitch %lld < 3 * width", (long long)pitch);
  const Window win{y0, x0, h, w};
  switch (fmt) {
    case DALIAMD_IMAGE_PNG: return DecodePng(data, size, out, pitch, win);
    case DALIAMD_IMAGE_BMP: return DecodeBmp(data, size, out, pitch, win);
    default: return DecodePnm(data, size, out, pitch, win);
  }
}

}  // extern "C"
