// Batched separable resampling (triangular / linear, ROI) with a fused CropMirrorNormalize
// epilogue, for gfx950.
//
// What it replaces: SeparableResamplingGPUImpl (dali/kernels/imgproc/resample/separable_impl.h:90-190,
// resampling_batch.cu:25-109, resampling_impl.cuh:58-376) and, when `normalize` is set, the
// SliceHwc2HwcChwNormalize fast path (dali/kernels/slice/slice_hwc2chw_normalize_gpu.cu:631-990).
//
// Arithmetic follows the reference's CPU backend so that results can be compared with it
// element by element:
//   * per-output index/coefficient tables exactly as InitializeResamplingFilter
//     (resampling_impl_cpu.cc:22-47): coefficients pre-normalised by division;
//   * taps accumulated in increasing k with separately rounded multiply and add (this file is
//     built with -ffp-contract=off), fp32 intermediate between the two passes
//     (separable_cpu.h:152-241, resampling_impl_cpu.h:74-84,116-121);
//   * pass order from the reference cost model (resampling_setup.cc:131-201);
//   * u8 rounding as the SSE2 build does it: half-to-even inside the 16-lane SIMD body,
//     half-away-from-zero in the scalar tails (common/simd.h:53-56, core/convert.h:306-321);
//   * fused epilogue = CMN CPU arithmetic (slice_flip_normalize_permute_pad_cpu.h:41-42):
//     (float(u8) - mean) * inv_std, fp16 stored round-to-nearest ties-away (util/half.hpp:231-243).
//
// MI355X design: ONE launch per batch; a workgroup owns a TILE_H x TILE_W output tile of one
// sample (descriptor table + binary search, XCD-aware block remap so all tiles of a sample share
// one XCD's L2).  Pass 1 reads the u8 source straight from global memory (dword loads when the
// row pitch allows) and leaves its fp32 result in LDS; pass 2 reads LDS only.  The fp32
// intermediate and the 224x224 u8 image of the unfused pipeline never touch HBM:
// algorithmic traffic = source ROI bytes + output bytes.
#include <cmath>
#include <cstring>
#include "common.h"

namespace daliamd {

constexpr int kResampleThreads = 256;
constexpr int kMaxLds = 60 * 1024;

// ---------------------------------------------------------------------------------------------
// shared host/device arithmetic
// ---------------------------------------------------------------------------------------------
// ResamplingFilter::operator() for the 3-entry triangular table {0,1,0}
// (resampling_filters.cuh:48-67, host branch)
__host__ __device__ inline float TriEval(float x) {
  if (!(x > -1)) return 0;
  if (x >= 3) return 0;
  int x0 = (int)floorf(x);
  int x1 = x0 + 1;
  float d = x - x0;
  float f0 = x0 < 0.0f ? 0.0f : (x0 == 1 ? 1.0f : 0.0f);
  float f1 = x1 >= 3 ? 0.0f : (x1 == 1 ? 1.0f : 0.0f);
  return f0 + d * (f1 - f0);
}

__host__ __device__ inline float FilterStart(float origin, float scale, float anchor) {
  float s = origin;
  s += 0.5f * scale - 0.5f - anchor;
  return s;
}

__host__ __device__ inline int FirstTap(int o, float scale, float start, float *f0) {
  float sx0f = o * scale + start;
  int sx0 = (int)ceilf(sx0f);
  *f0 = sx0 - sx0f;
  return sx0;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ClampI(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ uint32_t RoundU8(float v, bool half_even) {
  if (half_even) {
    float c = fminf(fmaxf(v, 0.0f), 255.0f);  // NaN -> 0 (fmaxf returns the non-NaN operand)
    return (uint32_t)rintf(c);
  }
  if (!(v > 0.0f)) return 0;
  float r = floorf(v);
  r += (v - r >= 0.5f) ? 1.0f : 0.0f;
  return (uint32_t)fminf(r, 255.0f);
}

// half_float::detail::float2half_impl<round_to_nearest>, ties away from zero (half.hpp:464-536)
__device__ __forceinline__ uint16_t Float2HalfAway(float f) {
  uint32_t bits = __float_as_uint(f);
  uint32_t e = (bits >> 23) & 0xff;
  uint32_t sign = (bits >> 16) & 0x8000;
  uint32_t mant = bits & 0x7FFFFF;
  uint32_t base;
  int shift;
  if (e < 103) { base = 0; shift = 24; }
  else if (e < 113) { base = 0x0400u >> (113 - e); shift = 126 - (int)e; }
  else if (e < 143) { base = (e - 112) << 10; shift = 13; }
  else if (e < 255) { base = 0x7C00; shift = 24; }
  else { base = 0x7C00; shift = 13; }
  uint32_t h = (base | sign) + (mant >> shift);
  uint32_t rnd = ((mant >> (shift - 1)) | (e == 102 ? 1u : 0u)) & ((h & 0x7C00) != 0x7C00 ? 1u : 0u);
  return (uint16_t)(h + rnd);
}

struct Epilogue {
  void *out;
  int out_h, out_w, channels;
  int dtype, layout, normalize, mirror;
  float mean[4], inv_std[4];

  __device__ __forceinline__ void Store(int y, int x, int c, uint32_t v) const {
    int xo = mirror ? out_w - 1 - x : x;
    size_t o = layout == DALIAMD_LAYOUT_CHW ? ((size_t)c * out_h + y) * out_w + xo
                                            : ((size_t)y * out_w + xo) * channels + c;
    if (dtype == DALIAMD_UINT8) {
      float f = (float)v;
      if (normalize) f = RoundU8(((float)v - mean[c]) * inv_std[c], false);
      reinterpret_cast<uint8_t *>(out)[o] = (uint8_t)f;
    } else {
      float f = (float)v;
      if (normalize) f = (f - mean[c]) * inv_std[c];
      if (dtype == DALIAMD_FLOAT16) reinterpret_cast<uint16_t *>(out)[o] = Float2HalfAway(f);
      else reinterpret_cast<float *>(out)[o] = f;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kResampleThreads) void ResampleKernel(const daliamdResampleDesc *__restrict__ descs,
                                                                   int ndesc, int total_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdResampleDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int tid = threadIdx.x;
  const int C = d.channels;
  const int TH = d.tile_h, TW = d.tile_w;
  int t = wg - d.wg_start;
  int ty = t / d.tiles_x, tx = t - ty * d.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int th = min(TH, d.out_h - oy0), tw = min(TW, d.out_w - ox0);
  const int sup_x = d.support[0], sup_y = d.support[1];

  float *cy = lds;                  // [TH][sup_y]
  float *cx = cy + TH * sup_y;      // [TW][sup_x]
  int *iy = reinterpret_cast<int *>(cx + TW * sup_x);  // [TH]
  int *ix = iy + TH;                // [TW]
  float *tmp = reinterpret_cast<float *>(ix + TW);
  tmp = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(tmp) + 15) & ~(uintptr_t)15);

  // ---- index / coefficient tables (InitializeResamplingFilter) ----
  for (int i = tid; i < th + tw; i += kResampleThreads) {
    int axis = i < th ? 1 : 0;
    int o = axis ? oy0 + i : ox0 + (i - th);
    int sup = axis ? sup_y : sup_x;
    float *co = axis ? cy + i * sup_y : cx + (i - th) * sup_x;
    float start = FilterStart(d.origin[axis], d.scale[axis], d.fanchor[axis]);
    float f0;
    int s0 = FirstTap(o, d.scale[axis], start, &f0);
    float sum = 0;
    for (int k = 0; k < sup; k++) {
      float c = TriEval((f0 + k) * d.fscale[axis]);
      co[k] = c;
      sum += c;
    }
    if (sum) {
      for (int k = 0; k < sup; k++) co[k] /= sum;
    }
    if (axis) iy[i] = s0; else ix[i - th] = s0;
  }
  __syncthreads();

  Epilogue ep;
  ep.out = d.out; ep.out_h = d.out_h; ep.out_w = d.out_w; ep.channels = C;
  ep.dtype = d.out_dtype; ep.layout = d.out_layout; ep.normalize = d.normalize; ep.mirror = d.mirror;
#pragma unroll
  for (int c = 0; c < 4; c++) { ep.mean[c] = d.mean[c]; ep.inv_std[c] = d.inv_std[c]; }

  const uint8_t *__restrict__ in = d.in;
  const int pitch = d.in_pitch;

  if (d.first_axis == 1) {
    // ================= vertical pass first (source rows -> LDS), then horizontal =================
    // second axis = x: taps are clamped to the ROI window [lo_x, lo_x + ext_x)
    const int ext_x = d.ext[0], lo_x = d.lo[0];
    int a0 = ClampI(ix[0], 0, ext_x - 1), a1 = ClampI(ix[0] + sup_x - 1, 0, ext_x - 1);
    int b0 = ClampI(ix[tw - 1], 0, ext_x - 1), b1 = ClampI(ix[tw - 1] + sup_x - 1, 0, ext_x - 1);
    const int c_lo = min(min(a0, a1), min(b0, b1));
    const int c_hi = max(max(a0, a1), max(b0, b1)) + 1;
    const int nbytes = (c_hi - c_lo) * C;      // tmp row length in elements
    const int in_h = d.ext[1];                 // first axis: clamp to the whole image
    const uint8_t *row0 = in + (size_t)(lo_x + c_lo) * C;
    if ((pitch & 3) == 0) {
      // dword path: every row has the same alignment
      const int shift = (int)(reinterpret_cast<uintptr_t>(row0) & 3);
      const uint8_t *arow0 = row0 - shift;
      const int ndw = (nbytes + shift + 3) >> 2;
      // bytes of the image buffer reachable from arow0 (never read past in + in_h * pitch)
      const ptrdiff_t buf_end = (ptrdiff_t)in_h * pitch - (arow0 - in);
      for (int i = tid; i < th * ndw; i += kResampleThreads) {
        int y = i / ndw, j = i - y * ndw;
        const float *co = cy + y * sup_y;
        int r0 = iy[y];
        float a[4] = {0, 0, 0, 0};
        for (int k = 0; k < sup_y; k++) {
          int r = ClampI(r0 + k, 0, in_h - 1);
          ptrdiff_t off = (ptrdiff_t)r * pitch + 4 * j;
          uint32_t v;
          if (off >= -(arow0 - in) && off + 4 <= buf_end) {
            v = *reinterpret_cast<const uint32_t *>(arow0 + off);
          } else {  // first/last dword of the buffer: assemble from the in-bounds bytes
            v = 0;
            for (int b = 0; b < 4; b++) {
              ptrdiff_t o = off + b;
              if (o >= -(arow0 - in) && o < buf_end) v |= (uint32_t)arow0[o] << (8 * b);
            }
          }
          float w = co[k];
#pragma unroll
          for (int b = 0; b < 4; b++) a[b] += (float)((v >> (8 * b)) & 255) * w;
        }
        float *trow = tmp + y * nbytes;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          int e = 4 * j + b - shift;
          if (e >= 0 && e < nbytes) trow[e] = a[b];
        }
      }
    } else {
      for (int i = tid; i < th * nbytes; i += kResampleThreads) {
        int y = i / nbytes, j = i - y * nbytes;
        const float *co = cy + y * sup_y;
        int r0 = iy[y];
        float a = 0;
        for (int k = 0; k < sup_y; k++) {
          int r = ClampI(r0 + k, 0, in_h - 1);
          a += (float)row0[(size_t)r * pitch + j] * co[k];
        }
        tmp[y * nbytes + j] = a;
      }
    }
    __syncthreads();
    const int n_out = th * tw * C;
    for (int i = tid; i < n_out; i += kResampleThreads) {
      int c = i % C;
      int xy = i / C;
      int x = xy % tw, y = xy / tw;
      const float *co = cx + x * sup_x;
      const float *trow = tmp + y * nbytes;
      int s0 = ix[x];
      float a = 0;
      for (int k = 0; k < sup_x; k++) {
        int sx = ClampI(s0 + k, 0, ext_x - 1) - c_lo;
        a += co[k] * trow[sx * C + c];
      }
      int gx = ox0 + x;
      bool even = (d.even_mask[(gx >> 5) & 7] >> (gx & 31)) & 1;
      ep.Store(oy0 + y, gx, c, RoundU8(a, even));
    }
  } else {
    // ================= horizontal pass first (gather along rows -> LDS), then vertical =================
    const int ext_y = d.ext[1], lo_y = d.lo[1];
    int a0 = ClampI(iy[0], 0, ext_y - 1), a1 = ClampI(iy[0] + sup_y - 1, 0, ext_y - 1);
    int b0 = ClampI(iy[th - 1], 0, ext_y - 1), b1 = ClampI(iy[th - 1] + sup_y - 1, 0, ext_y - 1);
    const int r_lo = min(min(a0, a1), min(b0, b1));
    const int r_hi = max(max(a0, a1), max(b0, b1)) + 1;
    const int nrows = r_hi - r_lo;
    const int in_w = d.ext[0];
    const int rowlen = tw * C;
    const uint8_t *base = in + (size_t)(lo_y + r_lo) * pitch;
    for (int i = tid; i < nrows * rowlen; i += kResampleThreads) {
      int r = i / rowlen, j = i - r * rowlen;
      int x = j / C, c = j - x * C;
      const float *co = cx + x * sup_x;
      const uint8_t *row = base + (size_t)r * pitch + c;
      int s0 = ix[x];
      float a = 0;
      for (int k = 0; k < sup_x; k++) {
        int sx = ClampI(s0 + k, 0, in_w - 1);
        a += co[k] * (float)row[sx * C];
      }
      tmp[i] = a;
    }
    __syncthreads();
    const int n_out = th * rowlen;
    const int flat_w = d.out_w * C;
    for (int i = tid; i < n_out; i += kResampleThreads) {
      int y = i / rowlen, j = i - y * rowlen;
      int x = j / C, c = j - x * C;
      const float *co = cy + y * sup_y;
      int s0 = iy[y];
      float a = 0;
      for (int k = 0; k < sup_y; k++) {
        int sy = ClampI(s0 + k, 0, ext_y - 1) - r_lo;
        a += tmp[sy * rowlen + j] * co[k];
      }
      // ResampleVert: 256-element tiles, 16-lane SIMD body then scalar tail
      int fi = (ox0 + x) * C + c;
      int t0 = fi & ~255;
      int tend = min(t0 + 256, flat_w);
      bool even = fi < t0 + ((tend - t0) & ~15);
      ep.Store(oy0 + y, ox0 + x, c, RoundU8(a, even));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host-side setup: SeparableResamplingSetup<2>::SetupSample restated for the fused kernel
// (resampling_setup.cc:46-122,131-201,271-337; params.h:43-60; resampling_filters.cuh:38-46)
// ---------------------------------------------------------------------------------------------
struct HostFilter {
  int num_coeffs = 0;
  float anchor = 0, scale = 1;
  void Rescale(float support) {
    float old_scale = scale;
    scale = (num_coeffs - 1) / support;
    anchor = anchor * old_scale / scale;
  }
  int Support() const { return (int)ceilf((num_coeffs - 1) / scale); }
};

static HostFilter Triangular(float radius) {
  HostFilter f;
  f.num_coeffs = 3;
  f.anchor = 1;
  f.scale = (3 - 1) * 0.5f;
  f.Rescale(std::max(1.0f, 2 * radius));
  return f;
}

static int SetupOne(const daliamdResampleArgs &a, daliamdResampleDesc &d, int index) {
  DALIAMD_REQUIRE(a.in_h > 0 && a.in_w > 0 && a.out_h > 0 && a.out_w > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: sample %d has an empty input or output", index);
  DALIAMD_REQUIRE(a.channels >= 1 && a.channels <= 4, DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdResampleSetup: sample %d: %d channels (supported: 1..4)", index, a.channels);
  DALIAMD_REQUIRE(a.in_pitch >= a.in_w * a.channels, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: sample %d: pitch %d < row bytes", index, a.in_pitch);
  DALIAMD_REQUIRE(a.min_filter != DALIAMD_INTERP_NN && a.mag_filter != DALIAMD_INTERP_NN,
                  DALIAMD_ERROR_UNSUPPORTED, "daliamdResampleSetup: nearest-neighbour filter not supported");
  DALIAMD_REQUIRE(a.out_dtype == DALIAMD_UINT8 || a.out_dtype == DALIAMD_FLOAT16 || a.out_dtype == DALIAMD_FLOAT,
                  DALIAMD_ERROR_UNSUPPORTED, "daliamdResampleSetup: unsupported output type %d", a.out_dtype);
  memset(&d, 0, sizeof(d));
  d.in = a.in; d.out = a.out;
  d.in_h = a.in_h; d.in_w = a.in_w; d.channels = a.channels; d.in_pitch = a.in_pitch;
  d.out_h = a.out_h; d.out_w = a.out_w;
  d.out_dtype = a.out_dtype; d.out_layout = a.out_layout; d.normalize = a.normalize; d.mirror = a.mirror;
  for (int c = 0; c < 4; c++) { d.mean[c] = a.mean[c]; d.inv_std[c] = a.inv_std[c]; }

  const int in_size[2] = {a.in_w, a.in_h};
  const int out_size[2] = {a.out_w, a.out_h};
  int roi_lo[2], roi_hi[2];
  for (int dim = 0; dim < 2; dim++) {  // dim 0 = H, 1 = W; axis: 0 = x, 1 = y
    int axis = 1 - dim;
    float roi_start = 0, roi_end = (float)in_size[axis];
    if (a.use_roi) {
      roi_start = dim == 0 ? a.roi_y0 : a.roi_x0;
      roi_end = dim == 0 ? a.roi_y1 : a.roi_x1;
    }
    float in_sz = a.use_roi ? std::abs(roi_end - roi_start) : (float)in_size[axis];
    int type = out_size[axis] < in_sz ? a.min_filter : a.mag_filter;
    bool aa = a.antialias != 0;
    if (aa && type == DALIAMD_INTERP_LINEAR) type = DALIAMD_INTERP_TRIANGULAR;
    else if (!aa && type == DALIAMD_INTERP_TRIANGULAR) type = DALIAMD_INTERP_LINEAR;
    float radius = 1;
    if (type == DALIAMD_INTERP_TRIANGULAR) {
      bool shrink = aa && (in_sz > out_size[axis]);
      radius = shrink ? in_sz / out_size[axis] : 1;
    }
    HostFilter f = Triangular(type == DALIAMD_INTERP_LINEAR ? 1.0f : radius);
    d.origin[axis] = roi_start;
    d.scale[axis] = (roi_end - roi_start) / out_size[axis];
    int support = f.Support();
    float lo, hi;
    if (roi_start <= roi_end) {
      lo = roi_start - f.anchor;
      hi = roi_end - f.anchor + support;
    } else {
      lo = roi_end - f.anchor;
      hi = roi_start - f.anchor + support;
    }
    roi_lo[axis] = std::max<int>(0, std::min<int>(in_size[axis], (int)std::floor(lo)));
    roi_hi[axis] = std::max<int>(0, std::min<int>(in_size[axis], (int)std::ceil(hi)));
    d.fscale[axis] = f.scale;
    d.fanchor[axis] = f.anchor;
    d.support[axis] = std::max(1, support);
  }
  DALIAMD_REQUIRE(roi_hi[0] > roi_lo[0] && roi_hi[1] > roi_lo[1], DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: sample %d: region of interest lies outside the image", index);
  // processing order (cost model)
  float best = 1e+30f;
  for (int first = 0; first < 2; first++) {
    int cur[2] = {roi_hi[0] - roi_lo[0], roi_hi[1] - roi_lo[1]};
    int ax[2] = {first, 1 - first};
    float total = 0;
    for (int p = 0; p < 2; p++) {
      int ai = ax[p];
      cur[ai] = out_size[ai];
      int64_t vol = (int64_t)cur[0] * cur[1];
      float base = (float)(d.support[ai] * vol);
      float mul = ai == 0 ? 1.4f : 1.0f;
      total = total + (mul * base + vol * 3.0f);
    }
    if (total < best) { best = total; d.first_axis = first; }
  }
  // clamp windows: first-pass axis sees the whole image; the other is cut to the source ROI and
  // its origin becomes ROI-relative (resampling_setup.cc:326-336)
  int second = 1 - d.first_axis;
  d.lo[d.first_axis] = 0; d.ext[d.first_axis] = in_size[d.first_axis];
  d.lo[second] = roi_lo[second]; d.ext[second] = roi_hi[second] - roi_lo[second];
  d.origin[second] -= roi_lo[second];

  // rounding mask for an H-last pass: columns inside the SSE2 16-lane groups round half-to-even
  if (d.first_axis == 1) {
    if (a.out_w <= 256) {
      int ow = a.out_w, in_w = d.ext[0], sup = d.support[0];
      std::vector<int> idx(ow);
      float start = FilterStart(d.origin[0], d.scale[0], d.fanchor[0]);
      for (int x = 0; x < ow; x++) { float f0; idx[x] = FirstTap(x, d.scale[0], start, &f0); }
      bool flipped = idx[ow - 1] < idx[0];
      int first_regular = 0, last_regular = ow - 1;
      if (flipped) {
        while (first_regular < ow && idx[first_regular] + sup > in_w) first_regular++;
        while (last_regular >= 0 && idx[last_regular] < 0) last_regular--;
      } else {
        while (first_regular < ow && idx[first_regular] < 0) first_regular++;
        while (last_regular >= 0 && idx[last_regular] + sup > in_w) last_regular--;
      }
      int bounds[5] = {0, std::min(first_regular, last_regular + 1), first_regular, last_regular + 1, ow};
      int x = 0;
      for (int r = 0; r < 4; r++) {
        int ox1 = bounds[r + 1];
        for (; x + 16 <= ox1; x += 16)
          for (int l = 0; l < 16; l++) d.even_mask[(x + l) >> 5] |= 1u << ((x + l) & 31);
        for (; x < ox1; x++) {}
      }
    } else {
      for (int i = 0; i < 8; i++) d.even_mask[i] = 0xffffffffu;  // wide outputs: SIMD rounding everywhere
    }
  }

  // tile selection: keep tmp + tables inside the LDS budget
  int tw = 32, th = 16;
  auto lds_need = [&](int tw_, int th_) -> size_t {
    size_t tables = (size_t)th_ * d.support[1] + (size_t)tw_ * d.support[0] + th_ + tw_;
    size_t tmp_elems;
    if (d.first_axis == 1) {
      size_t ncols = (size_t)std::ceil(tw_ * std::abs(d.scale[0])) + d.support[0] + 2;
      tmp_elems = (size_t)th_ * ncols * a.channels;
    } else {
      size_t nrows = (size_t)std::ceil(th_ * std::abs(d.scale[1])) + d.support[1] + 2;
      tmp_elems = nrows * (size_t)tw_ * a.channels;
    }
    return (tables + tmp_elems) * 4 + 16;
  };
  while (lds_need(tw, th) > (size_t)kMaxLds && (tw > 1 || th > 1)) {
    bool shrink_h;
    if (d.first_axis == 1) shrink_h = th > 1;   // tmp = th x cols: rows are the cheap thing to drop
    else shrink_h = !(tw > 1);                  // tmp = rows x tw: columns are the cheap thing to drop
    if (shrink_h) th >>= 1; else tw >>= 1;
  }
  DALIAMD_REQUIRE(lds_need(tw, th) <= (size_t)kMaxLds, DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdResampleSetup: sample %d: scale %g x %g needs more LDS than available", index,
                  d.scale[0], d.scale[1]);
  d.tile_w = tw; d.tile_h = th;
  d.tiles_x = (a.out_w + tw - 1) / tw;
  d.tiles_y = (a.out_h + th - 1) / th;
  d.lds_bytes = (int)lds_need(tw, th);
  return DALIAMD_SUCCESS;
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdResampleSetup(const daliamdResampleArgs *args, int n, daliamdResampleDesc *descs,
                                     int *num_workgroups, int *lds_bytes) {
  DALIAMD_REQUIRE(args && descs && num_workgroups && lds_bytes && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: NULL argument");
  int wg = 0, lds = 0;
  for (int i = 0; i < n; i++) {
    int rc = daliamd::SetupOne(args[i], descs[i], i);
    if (rc != DALIAMD_SUCCESS) return (daliamdResult_t)rc;
    descs[i].wg_start = wg;
    wg += descs[i].tiles_x * descs[i].tiles_y;
    lds = lds > descs[i].lds_bytes ? lds : descs[i].lds_bytes;
  }
  *num_workgroups = wg;
  *lds_bytes = lds;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdResampleRun(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                   int num_workgroups, int lds_bytes) {
  if (n == 0 || num_workgroups == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_workgroups > 0 && lds_bytes >= 0 && lds_bytes <= daliamd::kMaxLds,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdResampleRun: invalid argument");
  hipLaunchKernelGGL(daliamd::ResampleKernel, dim3(daliamd::XcdGrid(num_workgroups)),
                     dim3(daliamd::kResampleThreads), lds_bytes, (hipStream_t)stream, descs_dev, n,
                     num_workgroups);
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
