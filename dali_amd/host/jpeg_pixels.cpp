// Host half of the JPEG pixel pipeline: dequantisation + inverse DCT, chroma upsampling, colour conversion and EXIF
// orientation on the CPU - what decoders.image(device="cpu") runs after the host entropy decoder (jpeg_entropy.cpp).
//
// Reference counterpart: ImageDecoder<CPUBackend> (dali/operators/imgcodec/image_decoder.h:613-880, registered in
// host_decoder.cc:35-48) -> nvImageCodec's libjpeg_turbo_decoder extension -> libjpeg-turbo (un-vendored) with
// fancy upsampling always on (image_decoder.h:297-305) and the accurate integer IDCT (:290-291).  The arithmetic below
// is the published libjpeg-turbo arithmetic (jidctint.c "islow", jdsample.c triangle filters, jdcolor.c 16-bit
// fixed-point BT.601), the same the device kernels implement (csrc/jpeg_idct_math.h, csrc/jpeg_color.hip): both
// paths produce the same bytes.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "dali_amd_host.h"
#include "host_common.h"

namespace daliamd_host {
namespace {

constexpr int kConstBits = 13, kPass1Bits = 2;
constexpr int32_t F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270,
                  F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137,
                  F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

inline int32_t Descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 8-point pass of the islow butterfly (jidctint.c), not yet descaled
inline void Butterfly8(const int32_t in[8], int32_t out[8]) {
  int32_t z2 = in[2], z3 = in[6];
  int32_t z1 = (z2 + z3) * F_0_541196100;
  int32_t tmp2 = z1 + z3 * (-F_1_847759065);
  int32_t tmp3 = z1 + z2 * F_0_765366865;
  int32_t tmp0 = (in[0] + in[4]) * (1 << kConstBits);
  int32_t tmp1 = (in[0] - in[4]) * (1 << kConstBits);
  const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int32_t z4 = tmp1 + tmp3;
  const int32_t z5 = (z3 + z4) * F_1_175875602;
  tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
  z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = tmp10 + tmp3; out[7] = tmp10 - tmp3;
  out[1] = tmp11 + tmp2; out[6] = tmp11 - tmp2;
  out[2] = tmp12 + tmp1; out[5] = tmp12 - tmp1;
  out[3] = tmp13 + tmp0; out[4] = tmp13 - tmp0;
}
// range_limit[x & RANGE_MASK]: 10-bit signed wrap, +128, clamp
inline uint8_t RangeLimit(int32_t x) {
  const int32_t v = ((x & 1023) ^ 512) - 512 + 128;
  return (uint8_t)std::min(std::max(v, 0), 255);
}

// coef: [blocks_y][blocks_x][64] column-major blocks; quant: 64 values in the same element order -> plane rows of
// blocks_x * 8 samples
void IdctComponent(const int16_t *coef, const uint16_t *quant, int blocks_x, int blocks_y, uint8_t *plane) {
  const int pitch = blocks_x * 8;
  for (int by = 0; by < blocks_y; by++)
    for (int bx = 0; bx < blocks_x; bx++) {
      const int16_t *b = coef + ((size_t)by * blocks_x + bx) * 64;
      int32_t ws[8][8];  // [row][column] after pass 1
      for (int col = 0; col < 8; col++) {
        int32_t in[8], o[8];
        for (int r = 0; r < 8; r++) in[r] = (int32_t)b[col * 8 + r] * (int32_t)quant[col * 8 + r];
        Butterfly8(in, o);
        for (int r = 0; r < 8; r++) ws[r][col] = Descale(o[r], kConstBits - kPass1Bits);
      }
      for (int r = 0; r < 8; r++) {
        int32_t o[8];
        Butterfly8(ws[r], o);
        uint8_t *dst = plane + (size_t)(by * 8 + r) * pitch + bx * 8;
        for (int c = 0; c < 8; c++) dst[c] = RangeLimit(Descale(o[c], kConstBits + kPass1Bits + 3));
      }
    }
}

enum UpsampleMode { kFull, kH2V1, kH2V2, kH1V2, kBox };
inline int ClampI(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }

struct Comp {
  const uint8_t *plane;
  int pitch, mode, hx, vx, dw, dh;
};

// one output row of one component, up-sampled to the image width (jdsample.c: h2v1 / h2v2 / h1v2 fancy, box otherwise;
// the edge columns / rows are the general formulas with the neighbour index clamped)
void UpsampleRow(const Comp &c, int y, int width, uint8_t *out) {
  switch (c.mode) {
    case kFull:
      memcpy(out, c.plane + (size_t)y * c.pitch, width);
      break;
    case kH2V1: {
      const uint8_t *p = c.plane + (size_t)y * c.pitch;
      for (int x = 0; x < width; x++) {
        const int k = x >> 1;
        const int s = p[k];
        out[x] = (x & 1) ? (uint8_t)((s * 3 + p[ClampI(k + 1, 0, c.dw - 1)] + 2) >> 2)
                         : (uint8_t)((s * 3 + p[ClampI(k - 1, 0, c.dw - 1)] + 1) >> 2);
      }
      break;
    }
    case kH2V2: {
      const int r = y >> 1, r1 = ClampI((y & 1) ? r + 1 : r - 1, 0, c.dh - 1);
      const uint8_t *p0 = c.plane + (size_t)r * c.pitch, *p1 = c.plane + (size_t)r1 * c.pitch;
      for (int x = 0; x < width; x++) {
        const int k = x >> 1;
        const int kn = ClampI((x & 1) ? k + 1 : k - 1, 0, c.dw - 1);
        const int v = p0[k] * 3 + p1[k], vn = p0[kn] * 3 + p1[kn];
        out[x] = (uint8_t)((v * 3 + vn + ((x & 1) ? 7 : 8)) >> 4);
      }
      break;
    }
    case kH1V2: {
      const int r = y >> 1, r1 = ClampI((y & 1) ? r + 1 : r - 1, 0, c.dh - 1);
      const int bias = (y & 1) ? 2 : 1;
      const uint8_t *p0 = c.plane + (size_t)r * c.pitch, *p1 = c.plane + (size_t)r1 * c.pitch;
      for (int x = 0; x < width; x++) out[x] = (uint8_t)((p0[x] * 3 + p1[x] + bias) >> 2);
      break;
    }
    default: {
      const uint8_t *p = c.plane + (size_t)(y / c.vx) * c.pitch;
      for (int x = 0; x < width; x++) out[x] = p[std::min(x / c.hx, c.pitch - 1)];
    }
  }
}

int ModeOf(const daliamdJpegInfo &info, int c) {
  const int h = info.h_samp[c], v = info.v_samp[c];
  if (h == info.hmax && v == info.vmax) return kFull;
  if (h * 2 == info.hmax && v == info.vmax && info.down_w[c] > 2) return kH2V1;
  if (h == info.hmax && v * 2 == info.vmax) return kH1V2;
  if (h * 2 == info.hmax && v * 2 == info.vmax && info.down_w[c] > 2) return kH2V2;
  return kBox;
}

constexpr int kScaleBits = 16;
constexpr int32_t kOneHalf = 1 << (kScaleBits - 1);
constexpr int32_t Fix(double x) { return (int32_t)(x * (1L << kScaleBits) + 0.5); }
inline uint8_t Clamp8(int v) { return (uint8_t)std::min(std::max(v, 0), 255); }

// jdcolor.c ycc_rgb_convert: 16-bit fixed-point BT.601, full range
inline void YccToRgb(int y, int cb, int cr, uint8_t *rgb) {
  const int u = cb - 128, v = cr - 128;
  rgb[0] = Clamp8(y + ((Fix(1.40200) * v + kOneHalf) >> kScaleBits));
  rgb[1] = Clamp8(y + (((-Fix(0.34414)) * u + kOneHalf + (-Fix(0.71414)) * v) >> kScaleBits));
  rgb[2] = Clamp8(y + ((Fix(1.77200) * u + kOneHalf) >> kScaleBits));
}

}  // namespace

// The image in its stored orientation -> the upright position EXIF orientation `o` asks for.
// (H, W) = stored size; returns the position in the upright image, whose size is (W, H) for o >= 5.
static inline void UprightPos(int o, int y, int x, int H, int W, int *oy, int *ox) {
  switch (o) {
    case 2: *oy = y; *ox = W - 1 - x; break;          // mirrored horizontally
    case 3: *oy = H - 1 - y; *ox = W - 1 - x; break;  // rotated by 180 degrees
    case 4: *oy = H - 1 - y; *ox = x; break;          // mirrored vertically
    case 5: *oy = x; *ox = y; break;                  // transposed
    case 6: *oy = x; *ox = H - 1 - y; break;          // needs a clockwise quarter turn
    case 7: *oy = W - 1 - x; *ox = H - 1 - y; break;  // transverse
    case 8: *oy = W - 1 - x; *ox = y; break;          // needs a counter-clockwise quarter turn
    default: *oy = y; *ox = x;
  }
}

}  // namespace daliamd_host

using namespace daliamd_host;

extern "C" int daliamdJpegDecodeRgbHost(const uint8_t *data, size_t size, const daliamdJpegInfo *info, int orientation,
                                        uint8_t *out, int64_t pitch) {
  if (!data || !info || !out) return Fail("daliamdJpegDecodeRgbHost: NULL argument");
  const int nc = info->num_components;
  if (nc != 1 && nc != 3) return Fail("JPEG with %d components (CMYK/YCCK) is not supported", nc);
  const int W = info->width, H = info->height;
  const bool turned = orientation >= 5 && orientation <= 8;
  if (pitch < (int64_t)3 * (turned ? H : W)) return Fail("daliamdJpegDecodeRgbHost: pitch too small");
  // entropy decode
  std::vector<int16_t> coef_store;
  size_t total = 0;
  for (int c = 0; c < nc; c++) total += (size_t)info->coef_elems[c];
  coef_store.resize(total);
  int16_t *coef[4] = {nullptr, nullptr, nullptr, nullptr};
  {
    size_t off = 0;
    for (int c = 0; c < nc; c++) { coef[c] = coef_store.data() + off; off += (size_t)info->coef_elems[c]; }
  }
  uint16_t quant[4 * 64];
  if (daliamdJpegDecodeCoefficients(data, size, info, coef, quant) != 0) return 1;  // message already set
  // planes
  std::vector<uint8_t> plane_store(total);
  Comp comps[3];
  {
    size_t off = 0;
    for (int c = 0; c < nc; c++) {
      uint8_t *plane = plane_store.data() + off;
      IdctComponent(coef[c], quant + 64 * c, info->blocks_x[c], info->blocks_y[c], plane);
      comps[c] = Comp{plane, info->blocks_x[c] * 8, ModeOf(*info, c), info->hmax / std::max(1, info->h_samp[c]),
                      info->vmax / std::max(1, info->v_samp[c]), info->down_w[c], info->down_h[c]};
      off += (size_t)info->coef_elems[c];
    }
  }
  // rows: upsample, convert, place
  std::vector<uint8_t> rows((size_t)3 * W), px((size_t)3 * W);
  for (int y = 0; y < H; y++) {
    for (int c = 0; c < nc; c++) UpsampleRow(comps[c], y, W, rows.data() + (size_t)c * W);
    const uint8_t *r0 = rows.data(), *r1 = rows.data() + W, *r2 = rows.data() + 2 * (size_t)W;
    if (nc == 1) {
      for (int x = 0; x < W; x++) px[3 * x] = px[3 * x + 1] = px[3 * x + 2] = r0[x];
    } else if (info->color == 2) {  // stored as RGB (Adobe transform 0)
      for (int x = 0; x < W; x++) { px[3 * x] = r0[x]; px[3 * x + 1] = r1[x]; px[3 * x + 2] = r2[x]; }
    } else {
      for (int x = 0; x < W; x++) YccToRgb(r0[x], r1[x], r2[x], &px[3 * x]);
    }
    if (orientation <= 1 || orientation > 8) {
      memcpy(out + (size_t)y * pitch, px.data(), (size_t)3 * W);
    } else {
      for (int x = 0; x < W; x++) {
        int oy, ox;
        UprightPos(orientation, y, x, H, W, &oy, &ox);
        uint8_t *d = out + (size_t)oy * pitch + (size_t)3 * ox;
        d[0] = px[3 * x]; d[1] = px[3 * x + 1]; d[2] = px[3 * x + 2];
      }
    }
  }
  return 0;
}
