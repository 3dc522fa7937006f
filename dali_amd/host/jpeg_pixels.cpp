// Host half of the JPEG pixel pipeline: dequantisation + inverse DCT, chroma upsampling, colour conversion and EXIF
// orientation on the CPU - what decoders.image(device="cpu") runs after the host entropy decoder (jpeg_entropy.cpp).
//
// Reference counterpart: ImageDecoder<CPUBackend> (dali/operators/imgcodec/image_decoder.h:613-880, registered in
// host_decoder.cc:35-48) -> nvImageCodec's libjpeg_turbo_decoder extension -> libjpeg-turbo (un-vendored) with
// fancy upsampling always on (image_decoder.h:297-305) and the accurate integer IDCT (:290-291).  The arithmetic below
// is the published libjpeg-turbo arithmetic (jidctint.c "islow", jdsample.c triangle filters, jdcolor.c 16-bit
// fixed-point BT.601), the same the device kernels implement (csrc/jpeg_idct_math.h, csrc/jpeg_color.hip): both
// paths produce the same bytes.
#include <immintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
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
// (T = int32_t, or eight of them side by side: wrap-around integer arithmetic lane by lane either way)
template <typename T>
__attribute__((always_inline)) inline void Butterfly8(const T in[8], T out[8]) {
  T z2 = in[2], z3 = in[6];
  T z1 = (z2 + z3) * F_0_541196100;
  T tmp2 = z1 + z3 * (-F_1_847759065);
  T tmp3 = z1 + z2 * F_0_765366865;
  T tmp0 = (in[0] + in[4]) * (1 << kConstBits);
  T tmp1 = (in[0] - in[4]) * (1 << kConstBits);
  const T tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  T z4 = tmp1 + tmp3;
  const T z5 = (z3 + z4) * F_1_175875602;
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

typedef int32_t v8i __attribute__((vector_size(32)));

// 8 x 8 transpose of 32-bit elements
__attribute__((target("avx2"), always_inline)) inline void Transpose8(const v8i r[8], v8i out[8]) {
  const __m256i t0 = _mm256_unpacklo_epi32((__m256i)r[0], (__m256i)r[1]), t1 = _mm256_unpackhi_epi32((__m256i)r[0], (__m256i)r[1]);
  const __m256i t2 = _mm256_unpacklo_epi32((__m256i)r[2], (__m256i)r[3]), t3 = _mm256_unpackhi_epi32((__m256i)r[2], (__m256i)r[3]);
  const __m256i t4 = _mm256_unpacklo_epi32((__m256i)r[4], (__m256i)r[5]), t5 = _mm256_unpackhi_epi32((__m256i)r[4], (__m256i)r[5]);
  const __m256i t6 = _mm256_unpacklo_epi32((__m256i)r[6], (__m256i)r[7]), t7 = _mm256_unpackhi_epi32((__m256i)r[6], (__m256i)r[7]);
  const __m256i u0 = _mm256_unpacklo_epi64(t0, t2), u1 = _mm256_unpackhi_epi64(t0, t2);
  const __m256i u2 = _mm256_unpacklo_epi64(t1, t3), u3 = _mm256_unpackhi_epi64(t1, t3);
  const __m256i u4 = _mm256_unpacklo_epi64(t4, t6), u5 = _mm256_unpackhi_epi64(t4, t6);
  const __m256i u6 = _mm256_unpacklo_epi64(t5, t7), u7 = _mm256_unpackhi_epi64(t5, t7);
  out[0] = (v8i)_mm256_permute2x128_si256(u0, u4, 0x20);
  out[1] = (v8i)_mm256_permute2x128_si256(u1, u5, 0x20);
  out[2] = (v8i)_mm256_permute2x128_si256(u2, u6, 0x20);
  out[3] = (v8i)_mm256_permute2x128_si256(u3, u7, 0x20);
  out[4] = (v8i)_mm256_permute2x128_si256(u0, u4, 0x31);
  out[5] = (v8i)_mm256_permute2x128_si256(u1, u5, 0x31);
  out[6] = (v8i)_mm256_permute2x128_si256(u2, u6, 0x31);
  out[7] = (v8i)_mm256_permute2x128_si256(u3, u7, 0x31);
}

// The same two passes with the eight columns (then the eight rows) of a block side by side in one register: the
// integer arithmetic of every lane is that of the scalar code below
__attribute__((target("avx2")))
void IdctComponentAvx2(const int16_t *coef, const uint16_t *quant, int blocks_x, int blocks_y, uint8_t *plane) {
  const int pitch = blocks_x * 8;
  v8i q[8];
  for (int col = 0; col < 8; col++) q[col] = (v8i)_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i *)(quant + col * 8)));
  for (int by = 0; by < blocks_y; by++)
    for (int bx = 0; bx < blocks_x; bx++) {
      const int16_t *b = coef + ((size_t)by * blocks_x + bx) * 64;
      v8i v[8], in[8], o[8], ws[8];
      // v[col] = the dequantised column (lanes = rows); in[row] = lanes over the columns
      for (int col = 0; col < 8; col++)
        v[col] = (v8i)_mm256_mullo_epi32(_mm256_cvtepi16_epi32(_mm_loadu_si128((const __m128i *)(b + col * 8))), (__m256i)q[col]);
      Transpose8(v, in);
      Butterfly8(in, o);
      for (int r = 0; r < 8; r++) ws[r] = (o[r] + (1 << (kConstBits - kPass1Bits - 1))) >> (kConstBits - kPass1Bits);
      Transpose8(ws, in);   // in[column] = lanes over the rows
      Butterfly8(in, o);    // o[column] = lanes over the rows
      for (int c = 0; c < 8; c++) {
        const v8i x = (o[c] + (1 << (kConstBits + kPass1Bits + 3 - 1))) >> (kConstBits + kPass1Bits + 3);
        const v8i w = ((x & 1023) ^ 512) - 512 + 128;
        o[c] = (v8i)_mm256_min_epi32(_mm256_max_epi32((__m256i)w, _mm256_setzero_si256()), _mm256_set1_epi32(255));
      }
      Transpose8(o, ws);    // ws[row] = lanes over the columns
      for (int r = 0; r < 8; r++) {
        const __m128i p16 = _mm_packs_epi32(_mm256_castsi256_si128((__m256i)ws[r]), _mm256_extracti128_si256((__m256i)ws[r], 1));
        _mm_storel_epi64((__m128i *)(plane + (size_t)(by * 8 + r) * pitch + bx * 8), _mm_packus_epi16(p16, p16));
      }
    }
}

// coef: [blocks_y][blocks_x][64] column-major blocks; quant: 64 values in the same element order -> plane rows of
// blocks_x * 8 samples
void IdctComponent(const int16_t *coef, const uint16_t *quant, int blocks_x, int blocks_y, uint8_t *plane) {
  static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("DALI_AMD_HOST_NO_AVX2");   // (the variable: tests)
  if (avx2) return IdctComponentAvx2(coef, quant, blocks_x, blocks_y, plane);
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

// Row kernels of the fancy up-sampling, written for the vectoriser: `t` holds the (vertically blended) chroma row
// with one copy of the edge sample on either side - the clamped neighbour index of the edge columns - and the row is
// produced for all 2 * dw columns (the caller's row buffers have room for the odd one past the image width)
__attribute__((target_clones("avx2", "default")))
void H2V2Row(const uint8_t *p0, const uint8_t *p1, int dw, int16_t *t, uint8_t *out) {
  for (int k = 0; k < dw; k++) t[k + 1] = (int16_t)(p0[k] * 3 + p1[k]);
  t[0] = t[1];
  t[dw + 1] = t[dw];
  for (int k = 0; k < dw; k++) {
    out[2 * k] = (uint8_t)((t[k + 1] * 3 + t[k] + 8) >> 4);
    out[2 * k + 1] = (uint8_t)((t[k + 1] * 3 + t[k + 2] + 7) >> 4);
  }
}
__attribute__((target_clones("avx2", "default")))
void H2V1Row(const uint8_t *p, int dw, int16_t *t, uint8_t *out) {
  for (int k = 0; k < dw; k++) t[k + 1] = p[k];
  t[0] = t[1];
  t[dw + 1] = t[dw];
  for (int k = 0; k < dw; k++) {
    out[2 * k] = (uint8_t)((t[k + 1] * 3 + t[k] + 1) >> 2);
    out[2 * k + 1] = (uint8_t)((t[k + 1] * 3 + t[k + 2] + 2) >> 2);
  }
}
__attribute__((target_clones("avx2", "default")))
void H1V2Row(const uint8_t *p0, const uint8_t *p1, int bias, int width, uint8_t *out) {
  for (int x = 0; x < width; x++) out[x] = (uint8_t)((p0[x] * 3 + p1[x] + bias) >> 2);
}

// one output row of one component, up-sampled to the image width (jdsample.c: h2v1 / h2v2 / h1v2 fancy, box otherwise;
// the edge columns / rows are the general formulas with the neighbour index clamped)
// `out` has room for width + 1 samples, `scratch` for dw + 2 values
void UpsampleRow(const Comp &c, int y, int width, uint8_t *out, int16_t *scratch) {
  switch (c.mode) {
    case kFull:
      memcpy(out, c.plane + (size_t)y * c.pitch, width);
      break;
    case kH2V1:
      H2V1Row(c.plane + (size_t)y * c.pitch, c.dw, scratch, out);
      break;
    case kH2V2: {
      const int r = y >> 1, r1 = ClampI((y & 1) ? r + 1 : r - 1, 0, c.dh - 1);
      H2V2Row(c.plane + (size_t)r * c.pitch, c.plane + (size_t)r1 * c.pitch, c.dw, scratch, out);
      break;
    }
    case kH1V2: {
      const int r = y >> 1, r1 = ClampI((y & 1) ? r + 1 : r - 1, 0, c.dh - 1);
      H1V2Row(c.plane + (size_t)r * c.pitch, c.plane + (size_t)r1 * c.pitch, (y & 1) ? 2 : 1, width, out);
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

__attribute__((target_clones("avx2", "default")))
void YccRowToRgb(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int n, uint8_t *rgb) {
  for (int x = 0; x < n; x++) {
    const int u = cb[x] - 128, v = cr[x] - 128, yy = y[x];
    const int r = yy + ((Fix(1.40200) * v + kOneHalf) >> kScaleBits);
    const int g = yy + (((-Fix(0.34414)) * u + kOneHalf + (-Fix(0.71414)) * v) >> kScaleBits);
    const int b = yy + ((Fix(1.77200) * u + kOneHalf) >> kScaleBits);
    rgb[3 * x] = (uint8_t)std::min(std::max(r, 0), 255);
    rgb[3 * x + 1] = (uint8_t)std::min(std::max(g, 0), 255);
    rgb[3 * x + 2] = (uint8_t)std::min(std::max(b, 0), 255);
  }
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

// ---- output formats (decoders.image `output_type`) -------------------------------------------------------------
// The decoder produces RGB (or the luma plane alone for GRAY: nvImageCodec is asked for P_Y then,
// dali/operators/imgcodec/image_decoder.h:538-541) and ConvertCPU turns RGB into the other formats
// (dali/operators/imgcodec/util/convert.h:140-192,260-280): BGR = channel swap, YCbCr = ITU-R BT.601 with head / foot
// room (kernels/imgproc/color_manipulation/color_space_conversion_impl.h:62-103), float arithmetic, ConvertSat rounding.
static inline uint8_t SatRound(float v) {  // ConvertSat<uint8_t>(float): clamp, round half away from zero
  if (!(v > 0.0f)) return 0;
  if (v >= 255.0f) return 255;
  return (uint8_t)(v + 0.5f);
}
static inline void RgbToYcbcr601(const uint8_t *rgb, uint8_t *out) {
  const float r = rgb[0], g = rgb[1], b = rgb[2];
  out[0] = SatRound(0.25678823529f * r + 0.50412941176f * g + 0.09790588235f * b + 16.0f);
  out[1] = SatRound(-0.14822289945f * r + -0.29099278682f * g + 0.43921568627f * b + 128.0f);
  out[2] = SatRound(0.43921568627f * r + -0.36778831435f * g + -0.07142737192f * b + 128.0f);
}
static inline uint8_t RgbToGray(const uint8_t *rgb) {  // jpeg::rgb_to_y (:157-163)
  return SatRound(0.299f * rgb[0] + 0.587f * rgb[1] + 0.114f * rgb[2]);
}
// Pillow's CMYK -> RGB (libImaging/Convert.c cmyk2rgb) on samples as an Adobe file stores them (inverted): the pin for
// the four-component streams, whose conversion in the reference lives in un-vendored nvImageCodec.
static inline uint8_t MulDiv255(int a, int b) {
  const int t = a * b + 128;
  return (uint8_t)(((t >> 8) + t) >> 8);
}
static inline void CmykToRgb(int c, int m, int y, int k, bool inverted, uint8_t *rgb) {
  if (inverted) { c = 255 - c; m = 255 - m; y = 255 - y; k = 255 - k; }
  const int nk = 255 - k;
  rgb[0] = Clamp8(nk - MulDiv255(c, nk));
  rgb[1] = Clamp8(nk - MulDiv255(m, nk));
  rgb[2] = Clamp8(nk - MulDiv255(y, nk));
}

extern "C" int daliamdJpegOutputChannels(int num_components, int output_type) {
  if (output_type == DALIAMD_IMAGE_GRAY) return 1;
  if (output_type == DALIAMD_IMAGE_ANY) return num_components == 1 ? 1 : 3;
  return 3;
}

extern "C" int daliamdJpegDecodeHost(const uint8_t *data, size_t size, const daliamdJpegInfo *info, int orientation,
                                     int output_type, uint8_t *out, int64_t pitch) {
  if (!data || !info || !out) return Fail("daliamdJpegDecodeHost: NULL argument");
  const int nc = info->num_components;
  if (nc != 1 && nc != 3 && nc != 4) return Fail("JPEG with %d components is not supported", nc);
  if (output_type != DALIAMD_IMAGE_RGB && output_type != DALIAMD_IMAGE_BGR && output_type != DALIAMD_IMAGE_GRAY &&
      output_type != DALIAMD_IMAGE_YCBCR && output_type != DALIAMD_IMAGE_ANY)
    return Fail("decoders.image: unsupported output_type %d", output_type);
  const int oc = daliamdJpegOutputChannels(nc, output_type);
  const int W = info->width, H = info->height;
  const bool turned = orientation >= 5 && orientation <= 8;
  if (pitch < (int64_t)oc * (turned ? H : W)) return Fail("daliamdJpegDecodeHost: pitch too small");
  // entropy decode
  // per-thread work buffers, kept between calls: a fresh half-megabyte allocation per image is a round of page faults
  static thread_local std::vector<int16_t> coef_store, scratch;
  static thread_local std::vector<uint8_t> plane_store, rows, rgb, px;
  size_t total = 0;
  for (int c = 0; c < nc; c++) total += (size_t)info->coef_elems[c];
  if (coef_store.size() < total) coef_store.resize(total);
  int16_t *coef[4] = {nullptr, nullptr, nullptr, nullptr};
  {
    size_t off = 0;
    for (int c = 0; c < nc; c++) { coef[c] = coef_store.data() + off; off += (size_t)info->coef_elems[c]; }
  }
  uint16_t quant[4 * 64];
  if (daliamdJpegDecodeCoefficients(data, size, info, coef, quant) != 0) return 1;  // message already set
  // planes: the luma plane alone is enough for a gray output of a YCbCr / gray stream (jdcolor.c grayscale_convert)
  const bool luma_only = oc == 1 && (nc == 1 || (nc == 3 && info->color != 2));
  const int nplanes = luma_only ? 1 : nc;
  if (plane_store.size() < total) plane_store.resize(total);
  Comp comps[4];
  {
    size_t off = 0;
    for (int c = 0; c < nplanes; c++) {
      uint8_t *plane = plane_store.data() + off;
      IdctComponent(coef[c], quant + 64 * c, info->blocks_x[c], info->blocks_y[c], plane);
      comps[c] = Comp{plane, info->blocks_x[c] * 8, ModeOf(*info, c), info->hmax / std::max(1, info->h_samp[c]),
                      info->vmax / std::max(1, info->v_samp[c]), info->down_w[c], info->down_h[c]};
      off += (size_t)info->coef_elems[c];
    }
  }
  const int base_color = info->color & 7;
  const bool inverted = (info->color & 8) != 0;
  // rows: upsample, convert to RGB, to the output format, place
  const size_t rs = (size_t)W + 16;   // row stride of the component rows: room for the odd column of a 2x row
  if (rows.size() < 4 * rs) rows.resize(4 * rs);
  if (rgb.size() < (size_t)3 * W) { rgb.resize((size_t)3 * W); px.resize((size_t)3 * W); }
  if (scratch.size() < rs) scratch.resize(rs);
  for (int y = 0; y < H; y++) {
    for (int c = 0; c < nplanes; c++) UpsampleRow(comps[c], y, W, rows.data() + c * rs, scratch.data());
    const uint8_t *r0 = rows.data(), *r1 = r0 + rs, *r2 = r1 + rs, *r3 = r2 + rs;
    if (luma_only) {
      memcpy(px.data(), r0, W);
    } else {
      if (nc == 1) {
        for (int x = 0; x < W; x++) rgb[3 * x] = rgb[3 * x + 1] = rgb[3 * x + 2] = r0[x];
      } else if (nc == 3 && base_color == 2) {  // stored as RGB (Adobe transform 0)
        for (int x = 0; x < W; x++) { rgb[3 * x] = r0[x]; rgb[3 * x + 1] = r1[x]; rgb[3 * x + 2] = r2[x]; }
      } else if (nc == 3) {
        YccRowToRgb(r0, r1, r2, W, rgb.data());
      } else if (base_color == 4) {  // YCCK -> CMYK (jdcolor.c ycck_cmyk_convert: C, M, Y = 255 - R, G, B; K as it is)
        for (int x = 0; x < W; x++) {
          uint8_t t[3];
          YccToRgb(r0[x], r1[x], r2[x], t);
          CmykToRgb(255 - t[0], 255 - t[1], 255 - t[2], r3[x], inverted, &rgb[3 * x]);
        }
      } else {
        for (int x = 0; x < W; x++) CmykToRgb(r0[x], r1[x], r2[x], r3[x], inverted, &rgb[3 * x]);
      }
      switch (output_type) {
        case DALIAMD_IMAGE_BGR:
          for (int x = 0; x < W; x++) { px[3 * x] = rgb[3 * x + 2]; px[3 * x + 1] = rgb[3 * x + 1]; px[3 * x + 2] = rgb[3 * x]; }
          break;
        case DALIAMD_IMAGE_YCBCR:
          for (int x = 0; x < W; x++) RgbToYcbcr601(&rgb[3 * x], &px[3 * x]);
          break;
        case DALIAMD_IMAGE_GRAY:
          for (int x = 0; x < W; x++) px[x] = RgbToGray(&rgb[3 * x]);
          break;
        default:
          memcpy(px.data(), rgb.data(), (size_t)3 * W);
      }
    }
    if (orientation <= 1 || orientation > 8) {
      memcpy(out + (size_t)y * pitch, px.data(), (size_t)oc * W);
    } else {
      for (int x = 0; x < W; x++) {
        int oy, ox;
        UprightPos(orientation, y, x, H, W, &oy, &ox);
        uint8_t *d = out + (size_t)oy * pitch + (size_t)oc * ox;
        for (int c = 0; c < oc; c++) d[c] = px[oc * x + c];
      }
    }
  }
  return 0;
}

extern "C" int daliamdJpegDecodeRgbHost(const uint8_t *data, size_t size, const daliamdJpegInfo *info, int orientation,
                                        uint8_t *out, int64_t pitch) {
  return daliamdJpegDecodeHost(data, size, info, orientation, DALIAMD_IMAGE_RGB, out, pitch);
}

// In-place-free conversion of decoded RGB rows (raster formats decoded by daliamdImageDecodeRgb) to `output_type`.
extern "C" int daliamdConvertRgbRows(const uint8_t *rgb, int64_t in_pitch, int width, int height, int output_type,
                                     uint8_t *out, int64_t out_pitch) {
  if (!rgb || !out) return Fail("daliamdConvertRgbRows: NULL argument");
  for (int y = 0; y < height; y++) {
    const uint8_t *s = rgb + (size_t)y * in_pitch;
    uint8_t *d = out + (size_t)y * out_pitch;
    switch (output_type) {
      case DALIAMD_IMAGE_BGR:
        for (int x = 0; x < width; x++) { const uint8_t r = s[3 * x], b = s[3 * x + 2]; d[3 * x] = b; d[3 * x + 1] = s[3 * x + 1]; d[3 * x + 2] = r; }
        break;
      case DALIAMD_IMAGE_YCBCR:
        for (int x = 0; x < width; x++) { uint8_t t[3]; RgbToYcbcr601(&s[3 * x], t); d[3 * x] = t[0]; d[3 * x + 1] = t[1]; d[3 * x + 2] = t[2]; }
        break;
      case DALIAMD_IMAGE_GRAY:
        for (int x = 0; x < width; x++) d[x] = RgbToGray(&s[3 * x]);
        break;
      case DALIAMD_IMAGE_RGB:
      case DALIAMD_IMAGE_ANY:
        if (d != s) memmove(d, s, (size_t)3 * width);
        break;
      default:
        return Fail("decoders.image: unsupported output_type %d", output_type);
    }
  }
  return 0;
}
