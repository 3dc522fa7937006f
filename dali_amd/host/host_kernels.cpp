// CPU execution of the per-sample descriptors the device kernels consume: separable resampling (+ the fused
// CropMirrorNormalize epilogue) and stand-alone CropMirrorNormalize.  The operators build ONE set of descriptors
// (daliamdResampleSetup / the CropMirrorNormalize argument code) whatever the backend; device="gpu" uploads them and
// launches the HIP kernel, device="cpu" hands each sample to these functions on the operator's thread pool - one
// task per sample, like the reference's CPU operators (dali/operators/image/resize/resize_op_impl_cpu.h:84-107,
// dali/operators/image/crop/crop_mirror_normalize.cc:116-144).
//
// Arithmetic = the reference CPU kernels (and therefore = csrc/resample.hip, csrc/cmn.hip, which follow them):
//   coefficient tables   InitializeResamplingFilter   dali/kernels/imgproc/resample/resampling_impl_cpu.cc:22-47
//   filter evaluation    ResamplingFilter::operator() dali/kernels/imgproc/resample/resampling_filters.cuh:48-67
//   two passes, fp32 tmp SeparableResampleCPU         dali/kernels/imgproc/resample/separable_cpu.h:152-241,
//                                                     resampling_impl_cpu.h:50-123,362-390
//   u8 rounding          SSE2 body half-even, scalar tail half-away (kernels/common/simd.h:53-56, core/convert.h:306-321)
//   CMN                  slice_flip_normalize_permute_pad_cpu.h:37-64, half.hpp:231-243 (fp16 ties away from zero)
// This file is built with -ffp-contract=off: multiply and add are rounded separately, as on baseline x86-64.
#include <algorithm>
#include <cmath>
#include <emmintrin.h>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "dali_amd_host.h"
#include "dali_amd_resample_filters.h"
#include "host_common.h"

namespace daliamd_host {
namespace {

// the 3-entry triangular table {0, 1, 0}, linearly interpolated (host branch of ResamplingFilter::operator())
inline float TriEval(float x) {
  if (!(x > -1)) return 0;
  if (x >= 3) return 0;
  const int x0 = (int)std::floor(x), x1 = x0 + 1;
  const float d = x - x0;
  const float f0 = x0 < 0.0f ? 0.0f : (x0 == 1 ? 1.0f : 0.0f);
  const float f1 = x1 >= 3 ? 0.0f : (x1 == 1 ? 1.0f : 0.0f);
  return f0 + d * (f1 - f0);
}

struct AxisTable {
  std::vector<int32_t> first;  // first tap of every output position
  std::vector<float> coef;     // [out][support], normalised
};
const float *FilterTables() {  // the tabulated windows, built once (InitFilters, resampling_filters.cu:66-108)
  static const std::vector<float> tables = [] {
    std::vector<float> t(DALIAMD_RF_TOTAL);
    daliamdBuildFilterTables(t.data());
    return t;
  }();
  return tables.data();
}

AxisTable BuildTable(int axis, int kind, int out_size, float origin, float scale, float fanchor, float fscale, int support) {
  AxisTable t;
  t.first.resize(out_size);
  t.coef.resize((size_t)out_size * support);
  if (kind == DALIAMD_FK_NN) {  // one tap of weight 1 on the nearest source pixel (ResampleNN)
    for (int o = 0; o < out_size; o++) {
      t.first[o] = daliamdNearestIndex(axis, o, origin, scale);
      t.coef[o] = 1.0f;
    }
    return t;
  }
  const int ncoef = daliamdFilterTableSize(kind);
  const float *ftab = kind == DALIAMD_FK_TRIANGULAR ? nullptr : FilterTables() + daliamdFilterTableOffset(kind);
  float start = origin;
  start += 0.5f * scale - 0.5f - fanchor;
  for (int o = 0; o < out_size; o++) {
    const float sx0f = o * scale + start;
    const int sx0 = (int)std::ceil(sx0f);
    const float f0 = sx0 - sx0f;
    float *co = &t.coef[(size_t)o * support];
    float sum = 0;
    for (int k = 0; k < support; k++) {
      co[k] = ftab ? daliamdFilterEval(ftab, ncoef, (f0 + k) * fscale) : TriEval((f0 + k) * fscale);
      sum += co[k];
    }
    if (sum)
      for (int k = 0; k < support; k++) co[k] /= sum;
    t.first[o] = sx0;
  }
  return t;
}

inline int ClampI(int v, int lo, int hi) { return std::min(std::max(v, lo), hi); }

inline uint32_t RoundU8(float v, bool half_even) {
  if (half_even) {
    const float c = std::fmin(std::fmax(v, 0.0f), 255.0f);  // NaN -> 0
    return (uint32_t)std::nearbyint(c);                     // default rounding mode: to nearest even
  }
  if (!(v > 0.0f)) return 0;
  float r = std::floor(v);
  r += (v - r >= 0.5f) ? 1.0f : 0.0f;
  return (uint32_t)std::fmin(r, 255.0f);
}

// half_float::detail::float2half_impl<round_to_nearest> with ties away from zero (util/half.hpp:464-536)
uint16_t Float2HalfAway(float f) {
  uint32_t bits;
  memcpy(&bits, &f, 4);
  const uint32_t e = (bits >> 23) & 0xff, sign = (bits >> 16) & 0x8000, mant = bits & 0x7FFFFF;
  uint32_t base;
  int shift;
  if (e < 103) { base = 0; shift = 24; }
  else if (e < 113) { base = 0x0400u >> (113 - e); shift = 126 - (int)e; }
  else if (e < 143) { base = (e - 112) << 10; shift = 13; }
  else if (e < 255) { base = 0x7C00; shift = 24; }
  else { base = 0x7C00; shift = 13; }
  const uint32_t h = (base | sign) + (mant >> shift);
  const uint32_t rnd = ((mant >> (shift - 1)) | (e == 102 ? 1u : 0u)) & ((h & 0x7C00) != 0x7C00 ? 1u : 0u);
  return (uint16_t)(h + rnd);
}

inline float RoundAway(float v) {  // std::round
  float r = std::trunc(v);
  const float d = v - r;
  if (d >= 0.5f) r += 1.0f;
  else if (d <= -0.5f) r -= 1.0f;
  return r;
}

// epilogue of the resampling kernel: the rounded u8 value -> plain / normalised output element
struct Epilogue {
  void *out;
  int out_h, out_w, channels, dtype, layout, normalize, mirror;
  size_t Base(int y, int x, size_t *cstride) const {
    const int xo = mirror ? out_w - 1 - x : x;
    if (layout == DALIAMD_LAYOUT_CHW) { *cstride = (size_t)out_h * out_w; return (size_t)y * out_w + xo; }
    *cstride = 1;
    return ((size_t)y * out_w + xo) * channels;
  }
  void Store(size_t o, uint32_t v, float mean, float inv_std) const {
    float f = (float)v;
    if (dtype == DALIAMD_UINT8) {
      if (normalize) f = (float)RoundU8((f - mean) * inv_std, false);
      static_cast<uint8_t *>(out)[o] = (uint8_t)f;
    } else {
      if (normalize) f = (f - mean) * inv_std;
      if (dtype == DALIAMD_FLOAT16) static_cast<uint16_t *>(out)[o] = Float2HalfAway(f);
      else static_cast<float *>(out)[o] = f;
    }
  }
};

}  // namespace
}  // namespace daliamd_host

using namespace daliamd_host;

namespace {
inline float LoadElem(const uint8_t *row, int dtype, size_t elem) {
  switch (dtype) {
    case DALIAMD_UINT8: return (float)row[elem];
    case DALIAMD_INT16: { int16_t v; memcpy(&v, row + 2 * elem, 2); return (float)v; }
    case DALIAMD_UINT16: { uint16_t v; memcpy(&v, row + 2 * elem, 2); return (float)v; }
    default: { float v; memcpy(&v, row + 4 * elem, 4); return v; }
  }
}
// ConvertSat of the scalar tails (clamp(std::round)) or the SIMD store (clamp, then round half to even)
inline float RoundTyped(float v, int dtype, bool even) {
  if (dtype == DALIAMD_FLOAT) return v;
  const float lo = dtype == DALIAMD_INT16 ? -32768.0f : 0.0f;
  const float hi = dtype == DALIAMD_UINT8 ? 255.0f : dtype == DALIAMD_INT16 ? 32767.0f : 65535.0f;
  if (even) return std::nearbyint(std::fmin(std::fmax(v, lo), hi));
  const float r = RoundAway(v);
  if (!(r > lo)) return lo;
  return std::fmin(r, hi);
}
inline void StoreTyped(void *out, int dtype, size_t o, float r) {
  switch (dtype) {
    case DALIAMD_UINT8: static_cast<uint8_t *>(out)[o] = (uint8_t)r; break;
    case DALIAMD_INT16: static_cast<int16_t *>(out)[o] = (int16_t)r; break;
    case DALIAMD_UINT16: static_cast<uint16_t *>(out)[o] = (uint16_t)r; break;
    default: static_cast<float *>(out)[o] = r; break;
  }
}

// i16 / u16 / f32 input or the unrounded float result: the same two passes, element by element (the device side's
// ResampleGenericKernel)
int ResampleGenericHost(const daliamdResampleDesc &d) {
  const int C = d.channels, sup_x = d.support[0], sup_y = d.support[1];
  const AxisTable tx = BuildTable(0, d.filter_kind[0], d.out_w, d.origin[0], d.scale[0], d.fanchor[0], d.fscale[0], sup_x);
  const AxisTable ty = BuildTable(1, d.filter_kind[1], d.out_h, d.origin[1], d.scale[1], d.fanchor[1], d.fscale[1], sup_y);
  const int ex = d.ext[0] - 1, ey = d.ext[1] - 1;
  const uint8_t *in = d.in;
  const int tmp_w = d.tmp_w, tmp_h = d.tmp_h;
  std::vector<float> tmp((size_t)tmp_w * tmp_h * C);
  const bool vfirst = d.first_axis == 1;
  for (int y = 0; y < tmp_h; y++)
    for (int x = 0; x < tmp_w; x++)
      for (int c = 0; c < C; c++) {
        float acc = 0;
        if (vfirst) {
          for (int k = 0; k < sup_y; k++) {
            const int sy = ClampI(ty.first[y] + k, 0, ey) + d.lo[1];
            acc += LoadElem(in + (size_t)sy * d.in_pitch, d.in_dtype, (size_t)(d.lo[0] + x) * C + c) * ty.coef[(size_t)y * sup_y + k];
          }
        } else {
          const uint8_t *row = in + (size_t)(d.lo[1] + y) * d.in_pitch;
          for (int k = 0; k < sup_x; k++) {
            const int sx = ClampI(tx.first[x] + k, 0, ex) + d.lo[0];
            acc += tx.coef[(size_t)x * sup_x + k] * LoadElem(row, d.in_dtype, (size_t)sx * C + c);
          }
        }
        tmp[((size_t)y * tmp_w + x) * C + c] = acc;
      }
  // rounding regions of an H-last pass for any width, with the SIMD width of the output type
  std::vector<uint8_t> even_col(d.out_w, 0);
  const int lanes = d.round_lanes;
  if (vfirst) {
    const int ow = d.out_w, in_w = d.ext[0];
    const bool flipped = tx.first[ow - 1] < tx.first[0];
    int first_regular = 0, last_regular = ow - 1;
    if (flipped) {
      while (first_regular < ow && tx.first[first_regular] + sup_x > in_w) first_regular++;
      while (last_regular >= 0 && tx.first[last_regular] < 0) last_regular--;
    } else {
      while (first_regular < ow && tx.first[first_regular] < 0) first_regular++;
      while (last_regular >= 0 && tx.first[last_regular] + sup_x > in_w) last_regular--;
    }
    const int bounds[5] = {0, std::min(first_regular, last_regular + 1), first_regular, last_regular + 1, ow};
    int x = 0;
    for (int r = 0; r < 4; r++) {
      const int ox1 = bounds[r + 1];
      for (; x + lanes <= ox1; x += lanes)
        for (int l = 0; l < lanes; l++) even_col[x + l] = 1;
      x = std::max(x, ox1);
    }
  }
  const int flat_w = d.out_w * C;
  for (int y = 0; y < d.out_h; y++)
    for (int x = 0; x < d.out_w; x++)
      for (int c = 0; c < C; c++) {
        float acc = 0;
        bool even;
        if (vfirst) {
          for (int k = 0; k < sup_x; k++) {
            const int sx = ClampI(tx.first[x] + k, 0, tmp_w - 1);
            acc += tx.coef[(size_t)x * sup_x + k] * tmp[((size_t)y * tmp_w + sx) * C + c];
          }
          even = even_col[x] != 0;
        } else {
          for (int k = 0; k < sup_y; k++) {
            const int sy = ClampI(ty.first[y] + k, 0, tmp_h - 1);
            acc += tmp[((size_t)sy * tmp_w + x) * C + c] * ty.coef[(size_t)y * sup_y + k];
          }
          const int f = x * C + c, t0 = f & ~255;
          even = f < t0 + ((std::min(t0 + 256, flat_w) - t0) / lanes) * lanes;
        }
        const int xo = d.mirror ? d.out_w - 1 - x : x;
        const size_t o = d.out_layout == DALIAMD_LAYOUT_CHW ? ((size_t)c * d.out_h + y) * d.out_w + xo : ((size_t)y * d.out_w + xo) * C + c;
        StoreTyped(d.out, d.out_dtype, o, RoundTyped(acc, d.out_dtype, even));
      }
  return 0;
}
}  // namespace

namespace {
// clamp, then round half to even: cvtss2si under the default MXCSR rounding mode (= std::nearbyint, without the call)
inline uint32_t RoundU8Even(float v) {
  const float c = std::fmin(std::fmax(v, 0.0f), 255.0f);  // NaN -> 0
  return (uint32_t)_mm_cvtss_si32(_mm_set_ss(c));
}
inline uint32_t RoundU8Fast(float v, bool half_even) { return half_even ? RoundU8Even(v) : RoundU8(v, false); }

// The epilogue's output element is a function of (channel, rounded u8 value): 256 results per channel (the device
// kernel keeps the same table in LDS), as raw bits of the output type
struct EpilogueLut {
  uint32_t bits[4][256];
  void *out;
  int dtype;
  EpilogueLut(const Epilogue &ep, const float *mean, const float *inv_std) : out(ep.out), dtype(ep.dtype) {
    for (int c = 0; c < ep.channels; c++)
      for (uint32_t v = 0; v < 256; v++) {
        float f = (float)v;
        if (ep.dtype == DALIAMD_UINT8) {
          bits[c][v] = ep.normalize ? RoundU8((f - mean[c]) * inv_std[c], false) : v;
        } else {
          if (ep.normalize) f = (f - mean[c]) * inv_std[c];
          if (ep.dtype == DALIAMD_FLOAT16) bits[c][v] = Float2HalfAway(f);
          else memcpy(&bits[c][v], &f, 4);
        }
      }
  }
  inline void Store(size_t o, int c, uint32_t v) const {
    const uint32_t b = bits[c][v];
    if (dtype == DALIAMD_FLOAT16) static_cast<uint16_t *>(out)[o] = (uint16_t)b;
    else if (dtype == DALIAMD_UINT8) static_cast<uint8_t *>(out)[o] = (uint8_t)b;
    else static_cast<uint32_t *>(out)[o] = b;
  }
};

// A finished row of `n` float results -> rounded u8, written contiguously (u8 output, no normalisation, HWC, not
// mirrored): 16 elements at a time where all 16 round half to even - clamp, cvtps2dq, pack: the reference's SSE2 store -
// and element by element elsewhere
inline void StoreRowU8(const float *v, const uint8_t *even, int n, uint8_t *dst) {
  const __m128 lo = _mm_setzero_ps(), hi = _mm_set1_ps(255.0f);
  int e = 0;
  while (e < n) {
    if (e + 16 <= n && _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(even + e)), _mm_set1_epi8(1))) == 0xffff) {
      __m128i q[4];
      // max(v, 0) with v as the FIRST operand returns 0 for a NaN, like std::fmax(v, 0) in RoundU8Even
      for (int j = 0; j < 4; j++) q[j] = _mm_cvtps_epi32(_mm_min_ps(_mm_max_ps(_mm_loadu_ps(v + e + 4 * j), lo), hi));
      _mm_storeu_si128((__m128i *)(dst + e), _mm_packus_epi16(_mm_packs_epi32(q[0], q[1]), _mm_packs_epi32(q[2], q[3])));
      e += 16;
    } else {
      dst[e] = (uint8_t)RoundU8Fast(v[e], even[e] != 0);
      e++;
    }
  }
}

// One row of the horizontal pass: dst[x][c] = sum_k coef[x][k] * src[xoff[x][k] + c], taps in order, product then sum
// (ResampleCol, resampling_impl_cpu.h:50-86).  Four lanes per pixel whatever C is: src and dst carry 3 floats of padding.
inline void HorzRow(const float *src, const int *xoff, const float *coef, int out_w, int sup, int C, float *dst) {
  for (int x = 0; x < out_w; x++) {
    const int *xo = xoff + (size_t)x * sup;
    const float *co = coef + (size_t)x * sup;
    __m128 acc = _mm_setzero_ps();
    for (int k = 0; k < sup; k++) acc = _mm_add_ps(acc, _mm_mul_ps(_mm_set1_ps(co[k]), _mm_loadu_ps(src + xo[k])));
    _mm_storeu_ps(dst + (size_t)x * C, acc);
  }
}

// dst[e] += (float)row[e] * w over a source row / dst[e] += row[e] * w over an intermediate row: one tap of the
// vertical pass for every element (ResampleVert, resampling_impl_cpu.h:362-390: taps in order, product then sum)
__attribute__((target_clones("avx2", "default")))
void VertTapU8(float *dst, const uint8_t *row, float w, int n) {
  for (int e = 0; e < n; e++) dst[e] += (float)row[e] * w;
}
__attribute__((target_clones("avx2", "default")))
void VertTapF32(float *dst, const float *row, float w, int n) {
  for (int e = 0; e < n; e++) dst[e] += row[e] * w;
}
__attribute__((target_clones("avx2", "default")))
void RowToFloat(float *dst, const uint8_t *row, int n) {
  for (int e = 0; e < n; e++) dst[e] = (float)row[e];
}
}  // namespace

extern "C" int daliamdResampleRunHost(const daliamdResampleDesc *desc) {
  if (!desc || !desc->in || !desc->out) return Fail("daliamdResampleRunHost: NULL descriptor or buffer");
  const daliamdResampleDesc &d = *desc;
  const int C = d.channels, pitch = d.in_pitch;
  if (C < 1 || C > 4) return Fail("daliamdResampleRunHost: %d channels (supported: 1..4)", C);
  if (d.generic == 1) return ResampleGenericHost(d);   // (2: a u8 sample the DEVICE runs in two launches; here it is an ordinary one)
  const int sup_x = d.support[0], sup_y = d.support[1];
  const AxisTable tx = BuildTable(0, d.filter_kind[0], d.out_w, d.origin[0], d.scale[0], d.fanchor[0], d.fscale[0], sup_x);
  const AxisTable ty = BuildTable(1, d.filter_kind[1], d.out_h, d.origin[1], d.scale[1], d.fanchor[1], d.fscale[1], sup_y);
  // taps are clamped to [0, ext) inside the frame that starts at lo (first-pass axis: the whole image; second-pass
  // axis: the source region the first pass produced)
  const int ex = d.ext[0] - 1, ey = d.ext[1] - 1;
  const uint8_t *base = d.in + (size_t)d.lo[1] * pitch + (size_t)d.lo[0] * C;
  Epilogue ep{d.out, d.out_h, d.out_w, C, d.out_dtype, d.out_layout, d.normalize, d.mirror};
  const EpilogueLut lut(ep, d.mean, d.inv_std);
  // element offsets of the clamped taps of every output column
  std::vector<int> xoff((size_t)d.out_w * sup_x);
  for (int x = 0; x < d.out_w; x++)
    for (int k = 0; k < sup_x; k++) xoff[(size_t)x * sup_x + k] = ClampI(tx.first[x] + k, 0, ex) * C;
  const int rowlen = d.out_w * C;
  const bool plain_u8 = d.out_dtype == DALIAMD_UINT8 && !d.normalize && !d.mirror && d.out_layout != DALIAMD_LAYOUT_CHW;
  if (d.first_axis == 1) {
    // vertical pass into one intermediate row [ext_x * C], horizontal pass on it, epilogue: row by row
    const int NB = d.ext[0] * C;
    std::vector<float> trow(NB + 4), orow(rowlen + 4);
    // Rounding of an H-last pass (ResampleHorz, resampling_impl_cpu.h:286-336,477-485): the row is cut into the
    // left-clamped / regular / right-clamped column regions, every region runs 16 columns at a time through the SSE2
    // body (half to even) and finishes in a scalar tail (half away from zero).  The descriptor carries this as a
    // 256-column bit mask for the kernel; here it is derived for any width.
    std::vector<uint8_t> even_col(d.out_w, 0);
    {
      const int ow = d.out_w, in_w = d.ext[0];
      const bool flipped = tx.first[ow - 1] < tx.first[0];
      int first_regular = 0, last_regular = ow - 1;
      if (flipped) {
        while (first_regular < ow && tx.first[first_regular] + sup_x > in_w) first_regular++;
        while (last_regular >= 0 && tx.first[last_regular] < 0) last_regular--;
      } else {
        while (first_regular < ow && tx.first[first_regular] < 0) first_regular++;
        while (last_regular >= 0 && tx.first[last_regular] + sup_x > in_w) last_regular--;
      }
      const int bounds[5] = {0, std::min(first_regular, last_regular + 1), first_regular, last_regular + 1, ow};
      int x = 0;
      for (int r = 0; r < 4; r++) {
        const int ox1 = bounds[r + 1];
        for (; x + 16 <= ox1; x += 16)
          for (int l = 0; l < 16; l++) even_col[x + l] = 1;
        x = std::max(x, ox1);
      }
    }
    std::vector<uint8_t> even_el(rowlen);
    for (int x = 0; x < d.out_w; x++)
      for (int c = 0; c < C; c++) even_el[(size_t)x * C + c] = even_col[x];
    for (int y = 0; y < d.out_h; y++) {
      const float *co = &ty.coef[(size_t)y * sup_y];
      std::fill(trow.begin(), trow.end(), 0.0f);
      for (int k = 0; k < sup_y; k++) VertTapU8(trow.data(), base + (size_t)ClampI(ty.first[y] + k, 0, ey) * pitch, co[k], NB);
      HorzRow(trow.data(), xoff.data(), tx.coef.data(), d.out_w, sup_x, C, orow.data());
      if (plain_u8) { StoreRowU8(orow.data(), even_el.data(), rowlen, static_cast<uint8_t *>(d.out) + (size_t)y * rowlen); continue; }
      for (int x = 0; x < d.out_w; x++) {
        size_t cs;
        const size_t o = ep.Base(y, x, &cs);
        const bool even = even_col[x] != 0;
        for (int c = 0; c < C; c++) lut.Store(o + c * cs, c, RoundU8Fast(orow[(size_t)x * C + c], even));
      }
    }
  } else {
    // horizontal pass: tmp[ext_y][out_w * C], then vertical
    const int nrows = d.ext[1], NB = d.ext[0] * C;
    const size_t tpitch = (size_t)rowlen + 4;
    std::vector<float> tmp((size_t)nrows * tpitch), frow(NB + 4), acc(rowlen);
    // ResampleVert: 256-element tiles, 16-lane SIMD body (half to even) then scalar tail (half away)
    std::vector<uint8_t> even_el(rowlen);
    for (int fi = 0; fi < rowlen; fi++) {
      const int t0 = fi & ~255, tend = std::min(t0 + 256, rowlen);
      even_el[fi] = fi < t0 + ((tend - t0) & ~15);
    }
    for (int r = 0; r < nrows; r++) {
      RowToFloat(frow.data(), base + (size_t)r * pitch, NB);
      HorzRow(frow.data(), xoff.data(), tx.coef.data(), d.out_w, sup_x, C, &tmp[(size_t)r * tpitch]);
    }
    for (int y = 0; y < d.out_h; y++) {
      const float *co = &ty.coef[(size_t)y * sup_y];
      std::fill(acc.begin(), acc.end(), 0.0f);
      for (int k = 0; k < sup_y; k++) VertTapF32(acc.data(), &tmp[(size_t)ClampI(ty.first[y] + k, 0, ey) * tpitch], co[k], rowlen);
      if (plain_u8) { StoreRowU8(acc.data(), even_el.data(), rowlen, static_cast<uint8_t *>(d.out) + (size_t)y * rowlen); continue; }
      for (int x = 0; x < d.out_w; x++) {
        size_t cs;
        const size_t o = ep.Base(y, x, &cs);
        for (int c = 0; c < C; c++) {
          const int fi = x * C + c;
          // ResampleVert: 256-element tiles, 16-lane SIMD body (half to even) then scalar tail (half away)
          const int t0 = fi & ~255, tend = std::min(t0 + 256, rowlen);
          const bool even = fi < t0 + ((tend - t0) & ~15);
          lut.Store(o + c * cs, c, RoundU8Fast(acc[fi], even));
        }
      }
    }
  }
  return 0;
}

namespace {
// ConvertSat<Out> of one CropMirrorNormalize value, as the bits of the output element
inline uint32_t CmnElemBits(float v, int dtype) {
  switch (dtype) {
    case DALIAMD_FLOAT: { uint32_t b; memcpy(&b, &v, 4); return b; }
    case DALIAMD_FLOAT16: return Float2HalfAway(v);
    case DALIAMD_UINT8: {
      const float r = RoundAway(v);
      return (uint8_t)(!(r > 0.0f) ? 0.0f : std::fmin(r, 255.0f));
    }
    default: {
      float r = RoundAway(v);
      r = r != r ? 0.0f : std::fmin(std::fmax(r, -128.0f), 127.0f);
      return (uint8_t)(int8_t)r;
    }
  }
}

// Every output element is a function of (channel, u8 sample) alone: 256 results per channel, computed with the
// element-wise arithmetic above, then the crop is a table look-up per element.  T = the output element as raw bits.
template <typename T>
void CmnRowsLut(const daliamdCmnDesc &d) {
  const int cw = d.crop_w, ch = d.crop_h, C = d.channels, Co = d.out_channels;
  const bool chw = d.out_layout == DALIAMD_LAYOUT_CHW;
  T lut[4][256], fillv[4];
  for (int c = 0; c < 4; c++) {
    fillv[c] = (T)CmnElemBits(d.fill[c], d.out_dtype);
    for (int v = 0; v < 256; v++) {
      float f = (float)v;
      if (d.normalize) f = (f - d.mean[c]) * d.inv_std[c];
      lut[c][v] = (T)CmnElemBits(f, d.out_dtype);
    }
  }
  // columns whose source pixel lies inside the image: [x_lo, x_hi)
  int x_lo, x_hi;
  if (d.mirror) { x_lo = std::max(0, d.anchor_x + cw - d.in_w); x_hi = std::min(cw, d.anchor_x + cw); }
  else { x_lo = std::max(0, -d.anchor_x); x_hi = std::min(cw, d.in_w - d.anchor_x); }
  x_hi = std::max(x_hi, x_lo);
  T *out = static_cast<T *>(d.out);
  const size_t plane = (size_t)ch * cw;
  for (int y = 0; y < ch; y++) {
    const int sy = d.anchor_y + y;
    const bool row_in = sy >= 0 && sy < d.in_h;
    const int lo = row_in ? x_lo : cw, hi = row_in ? x_hi : cw;   // outside rows: fill only
    auto fill = [&](int x0, int x1) {
      for (int x = x0; x < x1; x++)
        for (int c = 0; c < Co; c++) out[chw ? c * plane + (size_t)y * cw + x : ((size_t)y * cw + x) * Co + c] = fillv[c];
    };
    fill(0, std::min(lo, cw));
    fill(std::max(hi, lo), cw);
    if (!row_in || lo >= hi) continue;
    const int step = d.mirror ? -C : C;
    const uint8_t *px = d.in + (size_t)sy * d.in_pitch + (size_t)(d.mirror ? d.anchor_x + (cw - 1 - lo) : d.anchor_x + lo) * C;
    if (chw && C == 3 && Co == 3) {
      T *o0 = out + (size_t)y * cw, *o1 = o0 + plane, *o2 = o1 + plane;
      for (int x = lo; x < hi; x++, px += step) {
        o0[x] = lut[0][px[0]];
        o1[x] = lut[1][px[1]];
        o2[x] = lut[2][px[2]];
      }
    } else {
      for (int x = lo; x < hi; x++, px += step)
        for (int c = 0; c < Co; c++)
          out[chw ? c * plane + (size_t)y * cw + x : ((size_t)y * cw + x) * Co + c] = c < C ? lut[c][px[c]] : fillv[c];
    }
  }
}
}  // namespace

extern "C" int daliamdCmnRunHost(const daliamdCmnDesc *desc) {
  if (!desc || !desc->in || !desc->out) return Fail("daliamdCmnRunHost: NULL descriptor or buffer");
  const daliamdCmnDesc &d = *desc;
  switch (d.out_dtype) {
    case DALIAMD_FLOAT: CmnRowsLut<uint32_t>(d); break;
    case DALIAMD_FLOAT16: CmnRowsLut<uint16_t>(d); break;
    default: CmnRowsLut<uint8_t>(d); break;
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// Audio resampling on the host (decoders.audio(sample_rate=...), audio_resample(device="cpu")): windowed sinc with a
// Hann envelope, coefficients from a table with linear interpolation, blocks of 256 outputs whose position advances by
// float additions (dali/kernels/signal/resampling.h:33-106, resampling_cpu.cc:129-236, resampling_params.h:27-30).
// The reference sums four partial sums in its SSE2 body; here the taps are added in order (the device kernel does the
// same), which stays inside the reference's own cpu-vs-gpu bound (test_audio_resample.py:60).
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct SincWindow {
  std::vector<float> lookup;
  float scale = 1, center = 1;
  int lobes = 0;
};
const SincWindow &GetSincWindow(int lobes) {
  static std::mutex mu;
  static std::map<int, SincWindow> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(lobes);
  if (it != cache.end()) return it->second;
  SincWindow w;
  const int coeffs = lobes * 64 + 1;
  const float wscale = 2.0f * lobes / (coeffs - 1), scale_envelope = 2.0f / coeffs;
  w.lookup.assign(coeffs + 5, 0.0f);
  const int c = (int)((coeffs - 1) * 0.5f);
  for (int i = 0; i < coeffs; i++) {
    float x = (i - c) * wscale, y = (i - c) * scale_envelope;
    float xp = (float)(x * M_PI);   // `x *= M_PI`: the product is formed in double
    float sinc = std::abs(xp) < 1e-5f ? 1.0f - xp * xp * (1.0f / 6) : std::sin(xp) / xp;
    w.lookup[i + 1] = (float)(sinc * (0.5 * (1 + std::cos((double)y * M_PI))));
  }
  w.center = (float)(c + 1);
  w.scale = 1 / wscale;
  w.lobes = lobes;
  return cache.emplace(lobes, std::move(w)).first->second;
}
}  // namespace

extern "C" int daliamdAudioResampleHost(const float *in, int64_t in_length, int channels, double in_rate, double out_rate,
                                        float quality, float *out, int64_t out_length) {
  if (!in || !out || channels < 1 || in_length < 0 || out_length < 0) return Fail("daliamdAudioResampleHost: invalid argument");
  if (!(in_rate > 0) || !(out_rate > 0)) return Fail("Sampling rate must be positive");
  if (!(quality >= 0 && quality <= 100)) return Fail("``quality`` out of range: %g\nValid range is [0..100].", (double)quality);
  const int lobes = (int)std::round(0.007 * quality * quality - 0.09 * quality + 3);
  const SincWindow &w = GetSincWindow(lobes);
  const float *lookup = w.lookup.data();
  const double scale = in_rate / out_rate;
  const float fscale = (float)scale;
  for (int64_t out_block = 0; out_block < out_length; out_block += 256) {
    const int64_t block_end = std::min<int64_t>(out_block + 256, out_length);
    const double in_block_f = out_block * scale;
    const int64_t in_block_i = (int64_t)std::floor(in_block_f);
    float in_pos = (float)(in_block_f - in_block_i);
    const float *blk = in + in_block_i * channels;
    for (int64_t out_pos = out_block; out_pos < block_end; out_pos++, in_pos += fscale) {
      const int xc = (int)std::ceil(in_pos);
      int i0 = xc - lobes, i1 = xc + lobes;
      if (i0 + in_block_i < 0) i0 = (int)-in_block_i;
      if (i1 + in_block_i > in_length) i1 = (int)(in_length - in_block_i);
      for (int c = 0; c < channels; c++) {
        float f = 0;
        float x = i0 - in_pos;
        for (int i = i0; i < i1; i++, x++) {
          const float fi = x * w.scale + w.center;
          const float fl = std::floor(fi);
          const float di = fi - fl;
          const int li = (int)fl;
          f += blk[(int64_t)i * channels + c] * (lookup[li] + di * (lookup[li + 1] - lookup[li]));
        }
        out[out_pos * channels + c] = f;
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Normalised sample-type conversion (ConvertSatNorm<float>(int), ConvertSatNorm<int>(float): convert.h:262-350)
// ---------------------------------------------------------------------------------------------
namespace {
template <typename T> float NormToFloat(T v, float inv_max) { return (float)v * inv_max; }
template <typename T> T NormFromFloat(float v, float fmax, float fmin, T tmin, T tmax) {
  const float r = std::round(v * fmax);
  return r <= fmin ? tmin : r >= fmax ? tmax : (T)r;
}
float LoadNormHost(const void *p, int dtype, int64_t i) {
  switch (dtype) {
    case DALIAMD_INT8: return NormToFloat(((const int8_t *)p)[i], 1.0f / 127.0f);
    case DALIAMD_UINT8: return NormToFloat(((const uint8_t *)p)[i], 1.0f / 255.0f);
    case DALIAMD_INT16: return NormToFloat(((const int16_t *)p)[i], 1.0f / 32767.0f);
    case DALIAMD_UINT16: return NormToFloat(((const uint16_t *)p)[i], 1.0f / 65535.0f);
    case DALIAMD_INT32: return NormToFloat(((const int32_t *)p)[i], 1.0f / 2147483648.0f);
    case DALIAMD_UINT32: return NormToFloat(((const uint32_t *)p)[i], 1.0f / 4294967296.0f);
    default: return ((const float *)p)[i];
  }
}
void StoreNormHost(void *p, int dtype, int64_t i, float v) {
  switch (dtype) {
    case DALIAMD_INT8: ((int8_t *)p)[i] = NormFromFloat<int8_t>(v, 127.0f, -128.0f, -128, 127); break;
    case DALIAMD_UINT8: ((uint8_t *)p)[i] = NormFromFloat<uint8_t>(v, 255.0f, 0.0f, 0, 255); break;
    case DALIAMD_INT16: ((int16_t *)p)[i] = NormFromFloat<int16_t>(v, 32767.0f, -32768.0f, -32768, 32767); break;
    case DALIAMD_UINT16: ((uint16_t *)p)[i] = NormFromFloat<uint16_t>(v, 65535.0f, 0.0f, 0, 65535); break;
    case DALIAMD_INT32: ((int32_t *)p)[i] = NormFromFloat<int32_t>(v, 2147483648.0f, -2147483648.0f, INT32_MIN, INT32_MAX); break;
    case DALIAMD_UINT32: ((uint32_t *)p)[i] = NormFromFloat<uint32_t>(v, 4294967296.0f, 0.0f, 0u, UINT32_MAX); break;
    default: ((float *)p)[i] = v;
  }
}
}  // namespace

extern "C" int daliamdConvertNormHost(const void *in, int in_dtype, void *out, int out_dtype, int64_t count, int mode) {
  auto known = [](int t) {
    return t == DALIAMD_INT8 || t == DALIAMD_UINT8 || t == DALIAMD_INT16 || t == DALIAMD_UINT16 || t == DALIAMD_INT32 ||
           t == DALIAMD_UINT32 || t == DALIAMD_FLOAT;
  };
  if (count < 0 || (count > 0 && (!in || !out)) || !known(in_dtype) || !known(out_dtype) || mode < 0 || mode > 2)
    return Fail("daliamdConvertNormHost: invalid argument");
  for (int64_t i = 0; i < count; i++) {
    float f = LoadNormHost(in, in_dtype, i);
    if (mode == 1) f = (f + 1.0f) * 0.5f;
    else if (mode == 2) f = f * 2.0f - 1.0f;
    StoreNormHost(out, out_dtype, i, f);
  }
  return 0;
}
