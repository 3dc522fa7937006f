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
constexpr int kTargetLds = 26 * 1024;

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

#define SEL4(c, a0, a1, a2, a3) ((c) == 0 ? (a0) : (c) == 1 ? (a1) : (c) == 2 ? (a2) : (a3))
typedef float floatx2 __attribute__((ext_vector_type(2)));

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
__device__ __noinline__ uint16_t Float2HalfAwaySlow(float f) {
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

// Fast path: in the normal half range the ties-away result is the hardware round-to-nearest-even
// result, plus one unit in the last place exactly when the dropped bits are 0x1000 (a tie) and RNE
// rounded down (kept LSB even).
__device__ __forceinline__ uint16_t Float2HalfAway(float f) {
  uint32_t bits = __float_as_uint(f);
  uint32_t e = (bits >> 23) & 0xff;
  if (e >= 113 && e < 142) {
    _Float16 hf = (_Float16)f;  // v_cvt_f16_f32, round to nearest even
    uint16_t h = __builtin_bit_cast(uint16_t, hf);
    bool tie_down = ((bits & 0x1FFF) == 0x1000) && ((bits & 0x2000) == 0);
    return tie_down ? (uint16_t)(h + 1) : h;
  }
  if ((bits & 0x7fffffff) == 0) return (uint16_t)(bits >> 16);
  return Float2HalfAwaySlow(f);
}

struct Epilogue {
  void *out;
  int out_h, out_w, channels;
  int dtype, layout, normalize, mirror;
  const uint16_t *lut;  // LDS: [channels][256] fp16 results of the normalisation, or null

  // element offset of (y, x, channel 0) and the per-channel stride
  __device__ __forceinline__ size_t Base(int y, int x, size_t *cstride) const {
    int xo = mirror ? out_w - 1 - x : x;
    if (layout == DALIAMD_LAYOUT_CHW) { *cstride = (size_t)out_h * out_w; return (size_t)y * out_w + xo; }
    *cstride = 1;
    return ((size_t)y * out_w + xo) * channels;
  }
  // mean / inv_std are passed by value: a runtime-indexed member array would live in scratch memory
  __device__ __forceinline__ void Store(size_t o, int c, uint32_t v, float mean, float inv_std) const {
    if (lut) {
      ((uint16_t __attribute__((address_space(1))) *)out)[o] = lut[c * 256 + v];
      return;
    }
    float f = (float)v;
    if (dtype == DALIAMD_UINT8) {
      if (normalize) f = RoundU8((f - mean) * inv_std, false);
      ((uint8_t __attribute__((address_space(1))) *)out)[o] = (uint8_t)f;
    } else {
      if (normalize) f = (f - mean) * inv_std;
      // (explicit global address space: generic stores would also count against the LDS counter)
      if (dtype == DALIAMD_FLOAT16) ((uint16_t __attribute__((address_space(1))) *)out)[o] = Float2HalfAway(f);
      else ((float __attribute__((address_space(1))) *)out)[o] = f;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// Per-sample tables (workspace words at desc.table_off), filled by ResampleTablesKernel once per sample instead of
// once per tile:   xi[out_w] | xc[out_w][sup_x] | yi[out_h] | yc[out_h][sup_y] | lut: u16 [channels][256]
// xi / yi = first tap of each output column / row, xc / yc its normalised coefficients (InitializeResamplingFilter);
// lut[c][v] = fp16((v - mean[c]) * inv_std[c]), ties away: the fused epilogue of an fp16 output is one look-up, the
// rounded u8 value being the index.
struct TableLayout { int xi, xc, yi, yc, lut, words; };
__host__ __device__ inline TableLayout MakeTableLayout(const daliamdResampleDesc &d) {
  TableLayout l;
  l.xi = 0;
  l.xc = l.xi + d.out_w;
  l.yi = l.xc + d.out_w * d.support[0];
  l.yc = l.yi + d.out_h;
  l.lut = l.yc + d.out_h * d.support[1];
  l.words = l.lut + (d.use_lut ? d.channels * 128 : 0);
  return l;
}
__host__ __device__ inline int TableEntries(const daliamdResampleDesc &d) {
  return d.out_w + d.out_h + (d.use_lut ? d.channels * 256 : 0);
}

using GU32 = uint32_t __attribute__((address_space(1)));
using GI32 = int32_t __attribute__((address_space(1)));
using GF32 = float __attribute__((address_space(1)));
using GU16 = uint16_t __attribute__((address_space(1)));
using GBytes = const uint8_t __attribute__((address_space(1)));

constexpr int kTableThreads = 256;
__global__ __launch_bounds__(kTableThreads) void ResampleTablesKernel(const daliamdResampleDesc *__restrict__ descs, int ndesc,
                                                                      int total_entries, uint8_t *__restrict__ workspace) {
  const int e = blockIdx.x * kTableThreads + threadIdx.x;
  if (e >= total_entries) return;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tab_start <= e) lo = mid; else hi = mid - 1;
  }
  const daliamdResampleDesc &d = descs[lo];
  const TableLayout L = MakeTableLayout(d);
  GU32 *tab = (GU32 *)(workspace + d.table_off);
  int local = e - d.tab_start;
  if (local >= d.out_w + d.out_h) {  // epilogue look-up entry
    const int q = local - d.out_w - d.out_h, c = q >> 8, v = q & 255;
    const float f = ((float)v - SEL4(c, d.mean[0], d.mean[1], d.mean[2], d.mean[3])) *
                    SEL4(c, d.inv_std[0], d.inv_std[1], d.inv_std[2], d.inv_std[3]);
    ((GU16 *)(tab + L.lut))[q] = Float2HalfAway(f);
    return;
  }
  const int axis = local < d.out_w ? 0 : 1;
  const int o = axis ? local - d.out_w : local;
  const int sup = d.support[axis];
  GF32 *co = (GF32 *)(tab + (axis ? L.yc : L.xc)) + (size_t)o * sup;
  const float start = FilterStart(d.origin[axis], d.scale[axis], d.fanchor[axis]);
  float f0;
  const int s0 = FirstTap(o, d.scale[axis], start, &f0);
  float sum = 0;
  for (int k = 0; k < sup; k++) {
    float c = TriEval((f0 + k) * d.fscale[axis]);
    co[k] = c;
    sum += c;
  }
  if (sum) {
    for (int k = 0; k < sup; k++) co[k] /= sum;
  }
  ((GI32 *)tab)[(axis ? L.yi : L.xi) + o] = s0;
}

// Descriptor lookup: tile counts are usually identical across the batch, so first try the uniform
// guess (two independent loads); fall back to the binary search.
__device__ __forceinline__ int FindResampleDesc(const daliamdResampleDesc *descs, int n, int wg, int total_wg) {
  int g = (int)(((long long)wg * n) / total_wg);
  if (descs[g].wg_start <= wg && (g + 1 == n || wg < descs[g + 1].wg_start)) return g;
  return FindDesc(descs, n, wg);
}

__device__ __forceinline__ void MinMax4(int a, int b, int c, int d, int *lo, int *hi) {
  *lo = min(min(a, b), min(c, d));
  *hi = max(max(a, b), max(c, d));
}

__global__ __launch_bounds__(kResampleThreads) void ResampleKernel(const daliamdResampleDesc *__restrict__ descs,
                                                                   int ndesc, int total_wg,
                                                                   const uint8_t *__restrict__ workspace) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdResampleDesc &d = descs[FindResampleDesc(descs, ndesc, wg, total_wg)];
  const int tid = threadIdx.x;
  const int C = d.channels;
  const int TH = d.tile_h, TW = d.tile_w;       // powers of two
  const int tw_log2 = 31 - __clz(TW), th_log2 = 31 - __clz(TH);
  int t = wg - d.wg_start;
  int ty = t / d.tiles_x, tx = t - ty * d.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int th = min(TH, d.out_h - oy0), tw = min(TW, d.out_w - ox0);
  const int sup_x = d.support[0], sup_y = d.support[1];
  const bool use_lut = d.use_lut != 0;

  // LDS carve-up: coefficients and per-tap source offsets (tap-major: entry k * TILE + i), the epilogue look-up
  // table, the staged window, tmp.  Byte offsets from the LDS base, never a round trip through an integer: that
  // would make every access behind it a generic one - flat loads that wait on both memory counters.
  float *cy = lds;                                        // [sup_y][TH]
  float *cx = cy + TH * sup_y;                            // [sup_x][TW]
  int *yt = reinterpret_cast<int *>(cx + TW * sup_x);     // [sup_y][TH] per-tap row offset
  int *xt = yt + TH * sup_y;                              // [sup_x][TW] per-tap column offset (elements)
  uint16_t *lut = reinterpret_cast<uint16_t *>(xt + TW * sup_x);  // [C][256]
  const size_t table_words = (size_t)2 * (TH * sup_y + TW * sup_x) + (use_lut ? C * 128 : 0);
  uint8_t *stage = reinterpret_cast<uint8_t *>(lds) + ((table_words * sizeof(float) + 15) & ~(size_t)15);

  const TableLayout L = MakeTableLayout(d);
  GU32 *tab = (GU32 *)(workspace + d.table_off);
  GI32 *xi = (GI32 *)tab + L.xi, *yi = (GI32 *)tab + L.yi;
  GF32 *xc = (GF32 *)(tab + L.xc), *yc = (GF32 *)(tab + L.yc);

  Epilogue ep;
  ep.out = d.out; ep.out_h = d.out_h; ep.out_w = d.out_w; ep.channels = C;
  ep.dtype = d.out_dtype; ep.layout = d.out_layout; ep.normalize = d.normalize; ep.mirror = d.mirror;
  ep.lut = use_lut ? lut : nullptr;
  const float mean0 = d.mean[0], mean1 = d.mean[1], mean2 = d.mean[2], mean3 = d.mean[3];
  const float inv0 = d.inv_std[0], inv1 = d.inv_std[1], inv2 = d.inv_std[2], inv3 = d.inv_std[3];

  const uint8_t *__restrict__ in = d.in;
  const int pitch = d.in_pitch;
  const bool vfirst = d.first_axis == 1;
  const bool staged = d.staged != 0;  // 0: source window too large for LDS, read it from global memory

  // ---- source window of this tile: rows [y_lo, y_hi] x columns [x_lo, x_hi] (in each axis' clamp frame) ----
  // first-pass axis: taps clamped to the whole image; second-pass axis: to the ROI window [lo, lo+ext)
  const int ex = d.ext[0] - 1, ey = d.ext[1] - 1;
  int x_lo, x_hi, y_lo, y_hi;
  {
    const int ix_a = xi[ox0], ix_b = xi[ox0 + tw - 1], iy_a = yi[oy0], iy_b = yi[oy0 + th - 1];
    MinMax4(ClampI(ix_a, 0, ex), ClampI(ix_a + sup_x - 1, 0, ex), ClampI(ix_b, 0, ex), ClampI(ix_b + sup_x - 1, 0, ex),
            &x_lo, &x_hi);
    MinMax4(ClampI(iy_a, 0, ey), ClampI(iy_a + sup_y - 1, 0, ey), ClampI(iy_b, 0, ey), ClampI(iy_b + sup_y - 1, 0, ey),
            &y_lo, &y_hi);
  }
  const int ncols = x_hi - x_lo + 1, nrows = y_hi - y_lo + 1;
  const int NB = ncols * C;                              // bytes per window row
  const int LP = (NB + 15 + 15) & ~15;                   // LDS row pitch (room for the alignment shift)
  const uint8_t *win = in + (size_t)(d.lo[1] + y_lo) * pitch + (size_t)(d.lo[0] + x_lo) * C;
  const uintptr_t win_addr = reinterpret_cast<uintptr_t>(win);
  float *tmp = reinterpret_cast<float *>(stage + (staged ? (size_t)nrows * LP : 0));
  const int rowlen = tw * C;                             // H-first tmp row length

  // ---- coefficient / per-tap offset tables of the tile, from the per-sample tables ----
  //   xt: element offset of tap k of column x inside a window row
  //   yt: V-first: byte offset of element 0 of the tapped row inside `stage` (or `win` when not staged)
  //       H-first: element offset of the tapped row inside tmp
  for (int i = tid; i < TW * sup_x; i += kResampleThreads) {
    const int x = i & (TW - 1), k = i >> tw_log2;
    if (x < tw) {
      cx[i] = xc[(size_t)(ox0 + x) * sup_x + k];
      xt[i] = (ClampI(xi[ox0 + x] + k, 0, ex) - x_lo) * C;
    }
  }
  for (int i = tid; i < TH * sup_y; i += kResampleThreads) {
    const int y = i & (TH - 1), k = i >> th_log2;
    if (y < th) {
      cy[i] = yc[(size_t)(oy0 + y) * sup_y + k];
      const int r = ClampI(yi[oy0 + y] + k, 0, ey) - y_lo;
      int v;
      if (!vfirst) v = r * rowlen;
      else if (staged) v = r * LP + (int)((win_addr + (size_t)r * pitch) & 15);
      else v = r * pitch;
      yt[i] = v;
    }
  }
  if (use_lut) {
    GU32 *src = tab + L.lut;
    uint32_t *dst = reinterpret_cast<uint32_t *>(lut);
    for (int i = tid; i < C * 128; i += kResampleThreads) dst[i] = src[i];
  }

  // ---- stage the window in LDS with 16-byte coalesced loads (each row keeps its own alignment shift) ----
  if (staged) {
    const uintptr_t buf_lo = reinterpret_cast<uintptr_t>(in);
    const uintptr_t buf_hi = buf_lo + (size_t)d.in_h * pitch;
    for (int r = tid >> 4; r < nrows; r += kResampleThreads / 16) {
      uintptr_t ra = win_addr + (size_t)r * pitch;
      int sh = (int)(ra & 15);
      int nch = (sh + NB + 15) >> 4;
      uint8_t *dst = stage + r * LP;
      for (int q = tid & 15; q < nch; q += 16) {
        uintptr_t g = ra - sh + 16 * q;
        uint4 v;
        if (g >= buf_lo && g + 16 <= buf_hi) {
          typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
          const u32x4_t t4 = *(const u32x4_t __attribute__((address_space(1))) *)g;  // global, not generic: the loads
          v = make_uint4(t4.x, t4.y, t4.z, t4.w);                                     // of a row may overlap the LDS stores
        } else {  // chunk straddles the buffer boundary: assemble from the in-bounds bytes
          uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
          for (int b = 0; b < 16; b++) {
            uintptr_t a = g + b;
            uint32_t byte = (a >= buf_lo && a < buf_hi) ? (uint32_t)(*(GBytes *)a) : 0u;
            byte <<= 8 * (b & 3);
            if (b < 4) w0 |= byte; else if (b < 8) w1 |= byte; else if (b < 12) w2 |= byte; else w3 |= byte;
          }
          v = make_uint4(w0, w1, w2, w3);
        }
        *reinterpret_cast<uint4 *>(dst + 16 * q) = v;
      }
    }
  }
  __syncthreads();
  GBytes *gwin = (GBytes *)win;  // source rows when the window is not staged

  if (vfirst) {
    // ================= vertical pass (window rows -> tmp[th][NB]), then horizontal =================
    if (staged && (pitch & 3) == 0) {
      // every row has the same shift modulo 4: produce 4 consecutive elements from one LDS dword per tap; the
      // (row, dword) items are spread evenly over the threads
      const int s4 = (int)(win_addr & 3);
      const int ndw = (NB + s4 + 3) >> 2;
      const float inv_ndw = 1.0f / (float)ndw;
      for (int item = tid; item < th * ndw; item += kResampleThreads) {
        const int y = (int)(((float)item + 0.5f) * inv_ndw);   // exact: item < 2^16
        const int j = item - y * ndw;
        const float *co = cy + y;
        const int *ro = yt + y;
        floatx2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
        const uint8_t *col = stage + 4 * j - s4;
        for (int k = 0; k < sup_y; k++) {
          const uint32_t v = *reinterpret_cast<const uint32_t *>(col + ro[k * TH]);
          const float w = co[k * TH];
          a01 += floatx2{(float)(v & 255), (float)((v >> 8) & 255)} * w;
          a23 += floatx2{(float)((v >> 16) & 255), (float)(v >> 24)} * w;
        }
        float *trow = tmp + y * NB;
        const int e = 4 * j - s4;
        if (e >= 0 && e + 3 < NB) {
          trow[e] = a01.x; trow[e + 1] = a01.y; trow[e + 2] = a23.x; trow[e + 3] = a23.y;
        } else {
          if (e >= 0 && e < NB) trow[e] = a01.x;
          if (e + 1 >= 0 && e + 1 < NB) trow[e + 1] = a01.y;
          if (e + 2 >= 0 && e + 2 < NB) trow[e + 2] = a23.x;
          if (e + 3 >= 0 && e + 3 < NB) trow[e + 3] = a23.y;
        }
      }
    } else {
      for (int y = tid >> 6; y < th; y += kResampleThreads / 64) {
        const float *co = cy + y;
        const int *ro = yt + y;
        for (int e = tid & 63; e < NB; e += 64) {
          float a = 0;
          if (staged) {
            for (int k = 0; k < sup_y; k++) a += (float)stage[ro[k * TH] + e] * co[k * TH];
          } else {
            for (int k = 0; k < sup_y; k++) a += (float)gwin[ro[k * TH] + e] * co[k * TH];
          }
          tmp[y * NB + e] = a;
        }
      }
    }
    __syncthreads();
    const int x = tid & (TW - 1);
    if (x < tw) {
      const float *co = cx + x;
      const int *xo = xt + x;
      const int gx = ox0 + x;
      const bool even = (d.even_mask[(gx >> 5) & 7] >> (gx & 31)) & 1;
      for (int y = tid >> tw_log2; y < th; y += kResampleThreads >> tw_log2) {
        const float *trow = tmp + y * NB;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int k = 0; k < sup_x; k++) {
          float w = co[k * TW];
          const float *p = trow + xo[k * TW];
          a0 += w * p[0];
          if (C > 1) a1 += w * p[1];
          if (C > 2) a2 += w * p[2];
          if (C > 3) a3 += w * p[3];
        }
        size_t cs, o = ep.Base(oy0 + y, gx, &cs);
        ep.Store(o, 0, RoundU8(a0, even), mean0, inv0);
        if (C > 1) ep.Store(o + cs, 1, RoundU8(a1, even), mean1, inv1);
        if (C > 2) ep.Store(o + 2 * cs, 2, RoundU8(a2, even), mean2, inv2);
        if (C > 3) ep.Store(o + 3 * cs, 3, RoundU8(a3, even), mean3, inv3);
      }
    }
  } else {
    // ================= horizontal pass (window rows -> tmp[nrows][tw*C]), then vertical =================
    const int x = tid & (TW - 1);
    if (x < tw) {
      const float *co = cx + x;
      const int *xo = xt + x;
      for (int r = tid >> tw_log2; r < nrows; r += kResampleThreads >> tw_log2) {
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (staged) {
          const uint8_t *srow = stage + r * LP + (int)((win_addr + (size_t)r * pitch) & 15);
          for (int k = 0; k < sup_x; k++) {
            float w = co[k * TW];
            const uint8_t *p = srow + xo[k * TW];
            a0 += w * (float)p[0];
            if (C > 1) a1 += w * (float)p[1];
            if (C > 2) a2 += w * (float)p[2];
            if (C > 3) a3 += w * (float)p[3];
          }
        } else {
          GBytes *srow = gwin + (size_t)r * pitch;
          for (int k = 0; k < sup_x; k++) {
            float w = co[k * TW];
            GBytes *p = srow + xo[k * TW];
            a0 += w * (float)p[0];
            if (C > 1) a1 += w * (float)p[1];
            if (C > 2) a2 += w * (float)p[2];
            if (C > 3) a3 += w * (float)p[3];
          }
        }
        float *tp = tmp + r * rowlen + x * C;
        tp[0] = a0;
        if (C > 1) tp[1] = a1;
        if (C > 2) tp[2] = a2;
        if (C > 3) tp[3] = a3;
      }
    }
    __syncthreads();
    const int flat_w = d.out_w * C;
    if (x < tw) {
      for (int y = tid >> tw_log2; y < th; y += kResampleThreads >> tw_log2) {
        const float *co = cy + y;
        const int *ro = yt + y;
        const float *tcol = tmp + x * C;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int k = 0; k < sup_y; k++) {
          float w = co[k * TH];
          const float *p = tcol + ro[k * TH];
          a0 += p[0] * w;
          if (C > 1) a1 += p[1] * w;
          if (C > 2) a2 += p[2] * w;
          if (C > 3) a3 += p[3] * w;
        }
        // ResampleVert: 256-element tiles, 16-lane SIMD body then scalar tail
        int fi = (ox0 + x) * C;
        size_t cs, o = ep.Base(oy0 + y, ox0 + x, &cs);
#define VLAST_EVEN(f) ((f) < ((f) & ~255) + ((min(((f) & ~255) + 256, flat_w) - ((f) & ~255)) & ~15))
        ep.Store(o, 0, RoundU8(a0, VLAST_EVEN(fi)), mean0, inv0);
        if (C > 1) ep.Store(o + cs, 1, RoundU8(a1, VLAST_EVEN(fi + 1)), mean1, inv1);
        if (C > 2) ep.Store(o + 2 * cs, 2, RoundU8(a2, VLAST_EVEN(fi + 2)), mean2, inv2);
        if (C > 3) ep.Store(o + 3 * cs, 3, RoundU8(a3, VLAST_EVEN(fi + 3)), mean3, inv3);
#undef VLAST_EVEN
      }
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
  d.use_lut = a.normalize && a.out_dtype == DALIAMD_FLOAT16;  // fused CMN to fp16: the epilogue is a 256-entry look-up

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

  // tile selection: keep tables + staged source window + tmp inside the LDS budget.  When even small tiles
  // cannot hold their source window (extreme down-scaling) fall back to reading the source from global memory.
  auto lds_need = [&](int tw_, int th_, bool staged) -> size_t {
    size_t tables = 2 * ((size_t)th_ * d.support[1] + (size_t)tw_ * d.support[0]) + (d.use_lut ? a.channels * 128 : 0);
    size_t ncols = (size_t)std::ceil(tw_ * std::abs(d.scale[0])) + d.support[0] + 2;
    size_t nrows = (size_t)std::ceil(th_ * std::abs(d.scale[1])) + d.support[1] + 2;
    ncols = std::min<size_t>(ncols, a.in_w);
    nrows = std::min<size_t>(nrows, a.in_h);
    size_t lp = (ncols * a.channels + 15 + 15) & ~(size_t)15;
    size_t stage = staged ? nrows * lp : 0;
    size_t tmp_elems = d.first_axis == 1 ? (size_t)th_ * ncols * a.channels : nrows * (size_t)tw_ * a.channels;
    return tables * 4 + 16 + stage + tmp_elems * 4;
  };
  auto shrink = [&](int &tw_, int &th_, bool staged, int min_area, size_t budget) {
    tw_ = 32; th_ = 16;
    while (lds_need(tw_, th_, staged) > budget && tw_ * th_ > min_area) {
      double fx = tw_ * std::abs(d.scale[0]) + d.support[0], fy = th_ * std::abs(d.scale[1]) + d.support[1];
      bool shrink_h = th_ > 1 && (fy >= fx || tw_ == 1);
      if (shrink_h) th_ >>= 1; else tw_ >>= 1;
    }
    return lds_need(tw_, th_, staged) <= budget;
  };
  // 1) comfortable budget: 6 workgroups per CU (160 KiB LDS) so memory latency stays hidden;
  // 2) whole LDS budget with smaller tiles; 3) no staging at all.
  int tw, th;
  // (measured on the bench set: 32x16 tiles inside 26 KB beat both larger tiles / fewer resident workgroups and
  // smaller tiles / more workgroups)
  bool staged = shrink(tw, th, true, 128, kTargetLds) || shrink(tw, th, true, 64, kMaxLds);
  if (!staged) {
    bool ok = shrink(tw, th, false, 1, kMaxLds);
    DALIAMD_REQUIRE(ok, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdResampleSetup: sample %d: scale %g x %g needs more LDS than available", index,
                    d.scale[0], d.scale[1]);
  }
  d.staged = staged ? 1 : 0;
  d.tile_w = tw; d.tile_h = th;
  d.tiles_x = (a.out_w + tw - 1) / tw;
  d.tiles_y = (a.out_h + th - 1) / th;
  d.lds_bytes = (int)lds_need(tw, th, staged);
  return DALIAMD_SUCCESS;
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdResampleSetup(const daliamdResampleArgs *args, int n, daliamdResampleDesc *descs,
                                     int *num_workgroups, int *lds_bytes, size_t *workspace_bytes, int *table_entries) {
  DALIAMD_REQUIRE(args && descs && num_workgroups && lds_bytes && workspace_bytes && table_entries && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: NULL argument");
  int wg = 0, lds = 0, entries = 0;
  size_t ws = 0;
  for (int i = 0; i < n; i++) {
    int rc = daliamd::SetupOne(args[i], descs[i], i);
    if (rc != DALIAMD_SUCCESS) return (daliamdResult_t)rc;
    descs[i].wg_start = wg;
    wg += descs[i].tiles_x * descs[i].tiles_y;
    lds = lds > descs[i].lds_bytes ? lds : descs[i].lds_bytes;
    descs[i].table_off = (int64_t)ws;
    descs[i].tab_start = entries;
    ws += ((size_t)daliamd::MakeTableLayout(descs[i]).words * 4 + 15) & ~(size_t)15;
    entries += daliamd::TableEntries(descs[i]);
  }
  *num_workgroups = wg;
  *lds_bytes = lds;
  *workspace_bytes = ws;
  *table_entries = entries;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdResampleRun(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                   int num_workgroups, int lds_bytes, void *workspace_dev, size_t workspace_bytes,
                                   int table_entries) {
  if (n == 0 || num_workgroups == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_workgroups > 0 && lds_bytes >= 0 && lds_bytes <= daliamd::kMaxLds,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdResampleRun: invalid argument");
  DALIAMD_REQUIRE(workspace_dev && workspace_bytes > 0 && table_entries > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleRun: the table workspace is missing (size it with daliamdResampleSetup)");
  {
    daliamd::KernelTimer timer("ResampleTablesKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(daliamd::ResampleTablesKernel, dim3((table_entries + daliamd::kTableThreads - 1) / daliamd::kTableThreads),
                       dim3(daliamd::kTableThreads), 0, (hipStream_t)stream, descs_dev, n, table_entries,
                       static_cast<uint8_t *>(workspace_dev));
  }
  {
    daliamd::KernelTimer timer("ResampleKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(daliamd::ResampleKernel, dim3(daliamd::XcdGrid(num_workgroups)),
                       dim3(daliamd::kResampleThreads), lds_bytes, (hipStream_t)stream, descs_dev, n,
                       num_workgroups, static_cast<const uint8_t *>(workspace_dev));
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
