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
// MI355X design (round 2): two launches per batch.  ResampleTablesKernel fills, per SAMPLE, the first tap and the
// normalised coefficients of every output column / row (and the 256-entry fp16 table of the fused normalisation) and,
// per TILE, a 128-byte record with everything a tile's prologue needs.  ResampleKernel: a workgroup takes consecutive
// 32 x 16 output tiles (XCD-aware remap: the tiles of a sample share one XCD's L2), stages the u8 source window in LDS
// with 16-byte loads while the previous tile computes, runs pass 1 into an fp32 tile in LDS and pass 2 from LDS - two
// pixels per thread on the common paths - and stores the (normalised) result.  The fp32 intermediate and the 224 x 224
// u8 image of the unfused pipeline never touch HBM: algorithmic traffic = source ROI bytes + output bytes.  Other
// element types and the unrounded float result take a plain two-launch path (ResampleGenericKernel).
#include <cmath>
#include <cstring>
#include <mutex>
#include "common.h"
#include "dali_amd_resample_filters.h"

namespace daliamd {

constexpr int kResampleThreads = 256;
constexpr int kMaxLds = 60 * 1024;
constexpr int kTargetLds = 26 * 1024;
// Pitch of a staged window row in LDS: room for the row at any 16-byte phase (NB + 15 + 15).  DALIAMD_RS_ROW_PAD=1 makes it
// congruent to 64 modulo 128 bytes: the vertical first pass reads 16 consecutive dwords per output row, two output rows
// per 32-lane group of a ds_read_b32 (32 banks); when these tap neighbouring window rows, a pitch of 16 dwords modulo 32
// puts the two runs on disjoint banks.  Measured (round 4, gpurun_out/r4i_pmc): SQ_LDS_BANK_CONFLICT 8.68 M -> 8.44 M per
// launch, no change in time - rows two apart then collide instead, and the conflicts that count are elsewhere: the four
// dword stores per item of the first pass (lane stride 4 dwords: two-way) and the second pass's gathers from tmp (16 pixel
// pairs of a row spread over ~125 dwords: three to four lanes per bank).  Off: it costs 20 % more window LDS.
#ifndef DALIAMD_RS_ROW_PAD
#define DALIAMD_RS_ROW_PAD 0
#endif
__host__ __device__ inline int StagedRowPitch(int nb) {
  const int need = (nb + 15 + 15) & ~15;
  return DALIAMD_RS_ROW_PAD ? ((need - 64 + 127) & ~127) + 64 : need;
}

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
typedef int intx2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ClampI(int v, int lo, int hi) { return min(max(v, lo), hi); }

// u8 rounding of the reference's two code paths: half-to-even (the SIMD body: cvtps with the default rounding mode) or
// half-away-from-zero (ConvertSat in the scalar tails); NaN and negatives -> 0.  Both from one round-to-nearest-even:
// it differs from half-away only at a tie that was rounded DOWN, where clamped - rounded is exactly +0.5.
// half-to-even only (the common tile: every column in a SIMD body): one instruction - v_cvt_pk_u8_f32 converts with the
// current rounding mode (nearest even) and saturates to 0..255 (NaN -> 0); the clamp + rint + convert above cost four
__device__ __forceinline__ uint32_t RoundU8Even(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u);
#else
  const float c = fminf(fmaxf(v, 0.0f), 255.0f);
  return (uint32_t)rintf(c);
#endif
}
__device__ __forceinline__ uint32_t RoundU8Slow(float v, bool half_even);
__device__ __forceinline__ uint32_t RoundU8(float v, bool half_even) { return RoundU8Slow(v, half_even); }
__device__ __forceinline__ uint32_t RoundU8Slow(float v, bool half_even) {
  const float c = fminf(fmaxf(v, 0.0f), 255.0f);  // NaN -> 0 (fmaxf returns the non-NaN operand)
  float r = rintf(c);
  if (!half_even && c - r == 0.5f) r += 1.0f;
  return (uint32_t)r;
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
  // Two neighbouring pixels (x even, x + 1) of a 3-channel fp16 CHW output through the look-up table: one dword per
  // channel plane.  Needs an even out_w and a 4-byte aligned output (the caller checks).
  __device__ __forceinline__ void StorePair(int y, int x, uint32_t r0, uint32_t r1, uint32_t g0, uint32_t g1, uint32_t b0,
                                            uint32_t b1) const {
    const int xo = mirror ? out_w - 2 - x : x;
    const uint32_t plane = (uint32_t)(out_h * out_w);
    uint16_t __attribute__((address_space(1))) *o = (uint16_t __attribute__((address_space(1))) *)out + ((size_t)y * out_w + xo);
    const uint32_t l0 = lut[r0], l1 = lut[r1], l2 = lut[256 + g0], l3 = lut[256 + g1], l4 = lut[512 + b0], l5 = lut[512 + b1];
    *(uint32_t __attribute__((address_space(1))) *)o = mirror ? (l1 | (l0 << 16)) : (l0 | (l1 << 16));
    *(uint32_t __attribute__((address_space(1))) *)(o + plane) = mirror ? (l3 | (l2 << 16)) : (l2 | (l3 << 16));
    *(uint32_t __attribute__((address_space(1))) *)(o + 2 * (size_t)plane) = mirror ? (l5 | (l4 << 16)) : (l4 | (l5 << 16));
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
// does output column x of an H-last pass round half to even?  (descriptor regions, see daliamdResampleDesc)
__host__ __device__ inline bool RoundsEven(const daliamdResampleDesc &d, int x) {
  return (x >= d.round_lo[0] && x < d.round_hi[0]) || (x >= d.round_lo[1] && x < d.round_hi[1]) ||
         (x >= d.round_lo[2] && x < d.round_hi[2]) || (x >= d.round_lo[3] && x < d.round_hi[3]);
}
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

// Tile record (workspace, 128 bytes per tile, written by ResampleTablesKernel): everything the prologue of a tile needs,
// so that the staging loads can be issued after ONE dependent memory round trip (the record) instead of four
// (descriptor index -> descriptor -> first taps of the tile's corners -> window).
struct TileRec {
  uint64_t win, out, tab;   // window origin, the sample's output, the sample's tables
  // what the epilogue needs to know (round 5: in the record, so that the common path - fp16 CHW through the look-up table -
  // never waits for the descriptor behind desc_idx): bit 0 mirror, bit 1 CHW, bit 2 normalize, bits 4-7 output type
  uint32_t epi, reserved;
  int32_t pitch, nrows, NB, LP;
  int32_t x_lo, y_lo, tmp_bytes, desc_idx;
  int32_t ox0, oy0, tw, th;
  int32_t TW, TH, sup_x, sup_y;
  int32_t ex, ey, out_w, out_h;
  int32_t flags, channels, rowlen;
  uint32_t even_bits;   // bit x: column ox0 + x of an H-last pass rounds half to even (TW <= 32)
};
static_assert(sizeof(TileRec) == 128, "layout");
enum { kRecVFirst = 1, kRecStaged = 2, kRecUseLut = 4, kRecPrefetch = 8, kRecInBounds = 16 };
constexpr int kLutLdsBytes = 2048;      // [4][256] fp16 at the start of the LDS block, whatever the tile
constexpr int kPrefetchChunks = 4;      // staged rows per thread held in registers across the previous tile's passes
#ifndef DALIAMD_RS_TILES
#define DALIAMD_RS_TILES 2
#endif
constexpr int kTilesPerWg = DALIAMD_RS_TILES;

__device__ __forceinline__ void MinMax4(int a, int b, int c, int d, int *lo, int *hi) {
  *lo = min(min(a, b), min(c, d));
  *hi = max(max(a, b), max(c, d));
}

// Elements per row of the fp32 intermediate of a V-first tile.  The packed vertical pass produces the four elements of
// one staged DWORD per item: with the rows laid out on the window's dword grid (element e of the tile at index e + s4, s4 =
// the window's byte offset inside its first dword) an item's result is ONE aligned 16-byte store - the four dword stores
// at a lane stride of four dwords it replaces hit every bank four times per wave.
__host__ __device__ inline int TmpRowPitch(int NB, bool vfirst, bool staged, int pitch, uint64_t win) {
  return vfirst && staged && (pitch & 3) == 0 ? (NB + (int)(win & 3) + 3) & ~3 : NB;
}

// one thread per tile: the record
__device__ void MakeTileRec(const daliamdResampleDesc *descs, int ndesc, int tile, uint8_t *workspace, TileRec *out) {
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].wg_start <= tile) lo = mid; else hi = mid - 1;
  }
  const daliamdResampleDesc &d = descs[lo];
  TileRec r;
  const int C = d.channels, TH = d.tile_h, TW = d.tile_w;
  const int t = tile - d.wg_start;
  const int ty = t / d.tiles_x, tx = t - ty * d.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int th = min(TH, d.out_h - oy0), tw = min(TW, d.out_w - ox0);
  const int sup_x = d.support[0], sup_y = d.support[1];
  const int ex = d.ext[0] - 1, ey = d.ext[1] - 1;
  float f0;
  const float start_x = FilterStart(d.origin[0], d.scale[0], d.fanchor[0]);
  const float start_y = FilterStart(d.origin[1], d.scale[1], d.fanchor[1]);
  int ix_a, ix_b, iy_a, iy_b;
  if (d.filter_kind[0] == DALIAMD_FK_NN) {
    ix_a = daliamdNearestIndex(0, ox0, d.origin[0], d.scale[0]);
    ix_b = daliamdNearestIndex(0, ox0 + tw - 1, d.origin[0], d.scale[0]);
  } else {
    ix_a = FirstTap(ox0, d.scale[0], start_x, &f0);
    ix_b = FirstTap(ox0 + tw - 1, d.scale[0], start_x, &f0);
  }
  if (d.filter_kind[1] == DALIAMD_FK_NN) {
    iy_a = daliamdNearestIndex(1, oy0, d.origin[1], d.scale[1]);
    iy_b = daliamdNearestIndex(1, oy0 + th - 1, d.origin[1], d.scale[1]);
  } else {
    iy_a = FirstTap(oy0, d.scale[1], start_y, &f0);
    iy_b = FirstTap(oy0 + th - 1, d.scale[1], start_y, &f0);
  }
  int x_lo, x_hi, y_lo, y_hi;
  MinMax4(ClampI(ix_a, 0, ex), ClampI(ix_a + sup_x - 1, 0, ex), ClampI(ix_b, 0, ex), ClampI(ix_b + sup_x - 1, 0, ex), &x_lo, &x_hi);
  MinMax4(ClampI(iy_a, 0, ey), ClampI(iy_a + sup_y - 1, 0, ey), ClampI(iy_b, 0, ey), ClampI(iy_b + sup_y - 1, 0, ey), &y_lo, &y_hi);
  const int ncols = x_hi - x_lo + 1, nrows = y_hi - y_lo + 1;
  const int NB = ncols * C, LP = StagedRowPitch(NB);
  const bool vfirst = d.first_axis == 1, staged = d.staged != 0;
  r.win = reinterpret_cast<uint64_t>(d.in + (size_t)(d.lo[1] + y_lo) * d.in_pitch + (size_t)(d.lo[0] + x_lo) * C);
  const uint64_t buf_lo = reinterpret_cast<uint64_t>(d.in), buf_hi = buf_lo + (size_t)d.in_h * d.in_pitch;
  r.out = reinterpret_cast<uint64_t>(d.out);
  r.epi = (d.mirror ? 1u : 0u) | (d.out_layout == DALIAMD_LAYOUT_CHW ? 2u : 0u) | (d.normalize ? 4u : 0u) | ((uint32_t)d.out_dtype << 4);
  r.reserved = 0;
  r.tab = reinterpret_cast<uint64_t>(workspace + d.table_off);
  r.pitch = d.in_pitch; r.nrows = nrows; r.NB = NB; r.LP = LP;
  r.x_lo = x_lo; r.y_lo = y_lo;
  r.tmp_bytes = 4 * (vfirst ? th * TmpRowPitch(NB, vfirst, staged, d.in_pitch, r.win) : nrows * tw * C);
  r.desc_idx = lo;
  r.ox0 = ox0; r.oy0 = oy0; r.tw = tw; r.th = th;
  r.TW = TW; r.TH = TH; r.sup_x = sup_x; r.sup_y = sup_y;
  r.ex = ex; r.ey = ey; r.out_w = d.out_w; r.out_h = d.out_h;
  const bool prefetch = staged && NB <= 241 - 15 && nrows <= 16 * kPrefetchChunks && TW * sup_x <= kResampleThreads &&
                        TH * sup_y <= kResampleThreads && d.in_pitch < (1 << 23);
  // every 16-byte chunk of every window row lies inside the source buffer: no per-chunk bounds checks
  const bool in_bounds = r.win >= buf_lo + 16 && r.win + (uint64_t)(nrows - 1) * d.in_pitch + NB + 32 <= buf_hi;
  r.flags = (vfirst ? kRecVFirst : 0) | (staged ? kRecStaged : 0) | (d.use_lut ? kRecUseLut : 0) | (prefetch ? kRecPrefetch : 0) |
            (in_bounds ? kRecInBounds : 0);
  r.channels = C; r.rowlen = tw * C;
  r.even_bits = 0;
  for (int x = 0; x < tw; x++) r.even_bits |= (RoundsEven(d, ox0 + x) ? 1u : 0u) << x;
  *out = r;
}

constexpr int kTableThreads = 256;
__global__ __launch_bounds__(kTableThreads) void ResampleTablesKernel(const daliamdResampleDesc *__restrict__ descs, int ndesc,
                                                                      int sample_entries, int total_tiles,
                                                                      uint8_t *__restrict__ workspace, size_t tile_rec_off,
                                                                      const float *__restrict__ filter_tables) {
  const int e = blockIdx.x * kTableThreads + threadIdx.x;
  if (e >= sample_entries + total_tiles) return;
  if (e >= sample_entries) {  // one thread per tile: its record
    const int tile = e - sample_entries;
    MakeTileRec(descs, ndesc, tile, workspace, reinterpret_cast<TileRec *>(workspace + tile_rec_off) + tile);
    return;
  }
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
  const int kind = d.filter_kind[axis];
  if (kind == DALIAMD_FK_NN) {  // one tap of weight 1 on the nearest source pixel
    co[0] = 1.0f;
    ((GI32 *)tab)[(axis ? L.yi : L.xi) + o] = daliamdNearestIndex(axis, o, d.origin[axis], d.scale[axis]);
    return;
  }
  const float start = FilterStart(d.origin[axis], d.scale[axis], d.fanchor[axis]);
  float f0;
  const int s0 = FirstTap(o, d.scale[axis], start, &f0);
  float sum = 0;
  const int ncoef = daliamdFilterTableSize(kind);
  GF32 *ftab = (GF32 *)filter_tables + (kind == DALIAMD_FK_TRIANGULAR ? 0 : daliamdFilterTableOffset(kind));
  for (int k = 0; k < sup; k++) {
    float c = kind == DALIAMD_FK_TRIANGULAR ? TriEval((f0 + k) * d.fscale[axis])
                                            : daliamdFilterEval(ftab, ncoef, (f0 + k) * d.fscale[axis]);
    co[k] = c;
    sum += c;
  }
  if (sum) {
    for (int k = 0; k < sup; k++) co[k] /= sum;
  }
  ((GI32 *)tab)[(axis ? L.yi : L.xi) + o] = s0;
}

// What a thread fetches from global memory for one tile ahead of time: its chunks of the staged window and its entry
// of the coefficient / first-tap tables of either axis.
struct TilePrefetch {
  uint4 chunk[kPrefetchChunks];   // (a slot is written to LDS under the condition it was loaded under: never initialised otherwise)
  float cxv, cyv;
  int xiv, yiv;
};

__device__ __forceinline__ uint4 LoadChunk(uintptr_t g, uintptr_t buf_lo, uintptr_t buf_hi) {
  if (g >= buf_lo && g + 16 <= buf_hi) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t t4 = *(const u32x4_t __attribute__((address_space(1))) *)g;
    return make_uint4(t4.x, t4.y, t4.z, t4.w);
  }
  // chunk straddles the buffer boundary: assemble from the in-bounds bytes
  uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
  for (int b = 0; b < 16; b++) {
    uintptr_t a = g + b;
    uint32_t byte = (a >= buf_lo && a < buf_hi) ? (uint32_t)(*(GBytes *)a) : 0u;
    byte <<= 8 * (b & 3);
    if (b < 4) w0 |= byte; else if (b < 8) w1 |= byte; else if (b < 12) w2 |= byte; else w3 |= byte;
  }
  return make_uint4(w0, w1, w2, w3);
}

__device__ __forceinline__ void FetchTile(const TileRec &r, const daliamdResampleDesc *descs, int tid, TilePrefetch &p) {
  if (!(r.flags & kRecPrefetch)) return;
  if (r.flags & kRecInBounds) {
    // 32-bit offsets from a uniform base (16 bytes in front of the window, so that they stay non-negative)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const uint8_t __attribute__((address_space(1))) *base = (const uint8_t __attribute__((address_space(1))) *)(r.win - 16);
    const uint32_t lo4 = (uint32_t)r.win & 15u;
#pragma unroll
    for (int i = 0; i < kPrefetchChunks; i++) {
      const int row = (tid >> 4) + 16 * i, q = tid & 15;
      if (16 * i >= r.nrows) continue;   // wave-uniform: a 31-row window skips the two upper chunk slots entirely
      if (row < r.nrows) {
        const uint32_t ro = (uint32_t)__mul24(row, r.pitch);
        const uint32_t sh = (lo4 + ro) & 15u;
        if (q < (int)((sh + (uint32_t)r.NB + 15u) >> 4)) {
          const u32x4_t t4 = *(const u32x4_t __attribute__((address_space(1))) *)(base + (ro + 16u - sh + 16u * (uint32_t)q));
          p.chunk[i] = make_uint4(t4.x, t4.y, t4.z, t4.w);
        }
      }
    }
  } else {
    // (a window at the edge of its buffer: the bounds come from the descriptor - the rare path pays the round trip)
    const daliamdResampleDesc &d = descs[r.desc_idx];
    const uintptr_t buf_lo = (uintptr_t)d.in, buf_hi = buf_lo + (size_t)d.in_h * d.in_pitch;
#pragma unroll
    for (int i = 0; i < kPrefetchChunks; i++) {
      const int row = (tid >> 4) + 16 * i, q = tid & 15;
      if (row < r.nrows) {
        const uintptr_t ra = (uintptr_t)r.win + (size_t)row * r.pitch;
        const int sh = (int)(ra & 15);
        if (q < ((sh + r.NB + 15) >> 4)) p.chunk[i] = LoadChunk(ra - sh + 16 * q, buf_lo, buf_hi);
      }
    }
  }
  GU32 *tab = (GU32 *)r.tab;
  const int tw_log2 = 31 - __clz(r.TW), th_log2 = 31 - __clz(r.TH);
  p.cxv = 0.0f; p.xiv = 0; p.cyv = 0.0f; p.yiv = 0;
  {
    const int x = tid & (r.TW - 1), k = tid >> tw_log2;
    if (k < r.sup_x && x < r.tw) {
      p.xiv = ((GI32 *)tab)[r.ox0 + x];
      p.cxv = ((GF32 *)(tab + r.out_w))[(size_t)(r.ox0 + x) * r.sup_x + k];
    }
  }
  {
    const int y = tid & (r.TH - 1), k = tid >> th_log2;
    if (k < r.sup_y && y < r.th) {
      const int yi_off = r.out_w + r.out_w * r.sup_x;
      p.yiv = ((GI32 *)tab)[yi_off + r.oy0 + y];
      p.cyv = ((GF32 *)(tab + yi_off + r.out_h))[(size_t)(r.oy0 + y) * r.sup_y + k];
    }
  }
}

// Workgroup barrier that orders LDS accesses only: __syncthreads() also waits for every global load in flight
// (its release fence covers all address spaces), which would make the next tile's prefetch wait at the first barrier.
__device__ __forceinline__ void LdsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Four bytes at any byte offset of a 4-byte aligned LDS buffer: two aligned dwords and a byte funnel shift (an unaligned
// ds_read_b32 works on gfx950 but is slow: the pass built on it measured 16 us slower per batch).
__device__ __forceinline__ uint32_t LdsDwordAt(const uint8_t *base, int off) {
  const uint32_t *q = reinterpret_cast<const uint32_t *>(base + (off & ~3));
  return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)off & 3u);
}

// Vertical first pass over a staged window whose rows share their alignment modulo 4.  16 threads per output row; a
// thread owns the dwords j, j + 16, ... (N of them) of that row's window: the row's tap offset and coefficient are read
// once per tap for all of them, every dword gives 4 elements (packed multiply / add, separately rounded).
#ifndef DALIAMD_RS_LPR
#define DALIAMD_RS_LPR 16
#endif
constexpr int kVLanesPerRow = DALIAMD_RS_LPR;   // lanes that share one output row of the vertical first pass
template <int N>
__device__ __forceinline__ void VPassItems(const uint8_t *stage, float *tmp, const float *cy, const int *yt, int tid, int th,
                                           int sup_y, int s4, int ndw, int jbase) {
  const int j0 = jbase + (tid & (kVLanesPerRow - 1));
  const uint8_t *col[N];
#pragma unroll
  for (int i = 0; i < N; i++) col[i] = stage + 4 * min(j0 + kVLanesPerRow * i, ndw - 1) - s4;   // (clamped: no out-of-range reads)
  for (int y = tid / kVLanesPerRow; y < th; y += kResampleThreads / kVLanesPerRow) {
    const float *co = cy + y * sup_y;
    const int *ro = yt + y * sup_y;
    floatx2 lo[N], hi[N];
#pragma unroll
    for (int i = 0; i < N; i++) lo[i] = hi[i] = floatx2{0.0f, 0.0f};
    const int nk = sup_y;
    int off_next = ro[0];     // one tap ahead: the offset / coefficient reads travel with the previous tap's data reads
    float w_next = co[0];
    for (int k = 0; k < nk; k++) {
      const int off = off_next;
      const float w = w_next;
      uint32_t v[N];
#pragma unroll
      for (int i = 0; i < N; i++) v[i] = *reinterpret_cast<const uint32_t *>(col[i] + off);
      const int kn = min(k + 1, nk - 1);
      off_next = ro[kn];
      w_next = co[kn];
#pragma unroll
      for (int i = 0; i < N; i++) {
        lo[i] += floatx2{(float)(v[i] & 255), (float)((v[i] >> 8) & 255)} * w;
        hi[i] += floatx2{(float)((v[i] >> 16) & 255), (float)(v[i] >> 24)} * w;
      }
    }
    float4 *trow = reinterpret_cast<float4 *>(tmp + y * (4 * ndw));   // rows on the dword grid: TmpRowPitch
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int j = j0 + kVLanesPerRow * i;
      if (j < ndw) trow[j] = make_float4(lo[i].x, lo[i].y, hi[i].x, hi[i].y);
    }
  }
}


__global__ __launch_bounds__(kResampleThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void ResampleKernel(const daliamdResampleDesc *__restrict__ descs,
                                                                   int ndesc, int total_wg, int total_tiles,
                                                                   const uint8_t *__restrict__ workspace,
                                                                   size_t tile_rec_off) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const int tid = threadIdx.x;
  const TileRec *recs = reinterpret_cast<const TileRec *>(workspace + tile_rec_off);
  const int t_begin = wg * kTilesPerWg, t_end = min(t_begin + kTilesPerWg, total_tiles);
  uint16_t *lut = reinterpret_cast<uint16_t *>(lds);   // fixed place: survives from tile to tile of the same sample
  int lut_desc = -1;

  TilePrefetch pf;
  FetchTile(recs[t_begin], descs, tid, pf);
  for (int tile = t_begin; tile < t_end; tile++) {
    // (re-read rather than carried over from the previous round: two live records are 64 scalar registers, and the
    // second read of a record comes from the scalar cache)
    const TileRec r = recs[tile];
    const daliamdResampleDesc &d = descs[r.desc_idx];
    const int C = r.channels, TH = r.TH, TW = r.TW;
    const int tw_log2 = 31 - __clz(TW), th_log2 = 31 - __clz(TH);
    const int oy0 = r.oy0, ox0 = r.ox0, th = r.th, tw = r.tw;
    const int sup_x = r.sup_x, sup_y = r.sup_y;
    const bool use_lut = (r.flags & kRecUseLut) != 0, vfirst = (r.flags & kRecVFirst) != 0, staged = (r.flags & kRecStaged) != 0;
    const int ex = r.ex, ey = r.ey, x_lo = r.x_lo, y_lo = r.y_lo;
    const int nrows = r.nrows, NB = r.NB, LP = r.LP, pitch = r.pitch, rowlen = r.rowlen;
    const uintptr_t win_addr = (uintptr_t)r.win;
    // V-first: pitch of a row of the fp32 intermediate and where element 0 sits in it (TmpRowPitch)
    const int NBp = TmpRowPitch(NB, vfirst, staged, pitch, r.win);
    const int tshift = NBp != NB || (vfirst && staged && (pitch & 3) == 0) ? (int)(win_addr & 3) : 0;

    // LDS carve-up (byte offsets from the LDS base, never a round trip through an integer: that would make every
    // access behind it a generic one): look-up table | staged window | tmp | coefficients and per-tap offsets
    // (tap-major: entry k * TILE + i)
    uint8_t *stage = reinterpret_cast<uint8_t *>(lds) + kLutLdsBytes;
    float *tmp = reinterpret_cast<float *>(stage + (staged ? (size_t)nrows * LP : 0));
    // (every table starts on 8 bytes: the pixel-pair passes read two columns' entries at once)
    const int ny_tab = (TH * sup_y + 1) & ~1, nx_tab = (TW * sup_x + 1) & ~1;
    float *cy = tmp + (((r.tmp_bytes >> 2) + 1) & ~1);      // [TH][sup_y] (row-major: a pass walks the taps of ONE row)
    float *cx = cy + ny_tab;                                // [sup_x][TW]
    int *yt = reinterpret_cast<int *>(cx + nx_tab);         // [TH][sup_y] per-tap row offset
    int *xt = yt + ny_tab;                                  // [sup_x][TW] per-tap column offset (elements)

    GU32 *tab = (GU32 *)r.tab;
    // ---- this tile's tables and window into LDS: from the registers filled one tile ago, or straight from memory ----
    //   xt: element offset of tap k of column x inside a window row
    //   yt: V-first: byte offset of element 0 of the tapped row inside `stage` (or `win` when not staged)
    //       H-first: element offset of the tapped row inside tmp
    auto y_offset = [&](int row) {
      if (!vfirst) return row * rowlen;
      if (staged) return row * LP + (int)((win_addr + (size_t)row * pitch) & 15);
      return row * pitch;
    };
    if (r.flags & kRecPrefetch) {
#pragma unroll
      for (int i = 0; i < kPrefetchChunks; i++) {
        const int row = (tid >> 4) + 16 * i, q = tid & 15;
        if (16 * i >= nrows) continue;   // wave-uniform
        if (row < nrows) {
          const int sh = (int)(((uint32_t)win_addr + (uint32_t)__mul24(row, pitch)) & 15u);
          if (q < ((sh + NB + 15) >> 4)) *reinterpret_cast<uint4 *>(stage + __mul24(row, LP) + 16 * q) = pf.chunk[i];
        }
      }
      {
        const int x = tid & (TW - 1), k = tid >> tw_log2;
        if (k < sup_x && x < tw) {
          cx[tid] = pf.cxv;
          xt[tid] = (ClampI(pf.xiv + k, 0, ex) - x_lo) * C + tshift;
        }
      }
      {
        const int y = tid & (TH - 1), k = tid >> th_log2;
        if (k < sup_y && y < th) {
          cy[y * sup_y + k] = pf.cyv;
          yt[y * sup_y + k] = y_offset(ClampI(pf.yiv + k, 0, ey) - y_lo);
        }
      }
    } else {
      GI32 *xi = (GI32 *)tab, *yi = (GI32 *)tab + r.out_w + r.out_w * sup_x;
      GF32 *xc = (GF32 *)(tab + r.out_w), *yc = (GF32 *)(tab + r.out_w + r.out_w * sup_x + r.out_h);
      for (int i = tid; i < TW * sup_x; i += kResampleThreads) {
        const int x = i & (TW - 1), k = i >> tw_log2;
        if (x < tw) {
          cx[i] = xc[(size_t)(ox0 + x) * sup_x + k];
          xt[i] = (ClampI(xi[ox0 + x] + k, 0, ex) - x_lo) * C + tshift;
        }
      }
      for (int i = tid; i < TH * sup_y; i += kResampleThreads) {
        const int y = i & (TH - 1), k = i >> th_log2;
        if (y < th) {
          cy[y * sup_y + k] = yc[(size_t)(oy0 + y) * sup_y + k];
          yt[y * sup_y + k] = y_offset(ClampI(yi[oy0 + y] + k, 0, ey) - y_lo);
        }
      }
      if (staged) {  // 16-byte coalesced loads, each row keeps its own alignment shift
        for (int row = tid >> 4; row < nrows; row += kResampleThreads / 16) {
          const uintptr_t ra = win_addr + (size_t)row * pitch;
          const int sh = (int)(ra & 15), nch = (sh + NB + 15) >> 4;
          for (int q = tid & 15; q < nch; q += 16)
            *reinterpret_cast<uint4 *>(stage + row * LP + 16 * q) =
                LoadChunk(ra - sh + 16 * q, (uintptr_t)d.in, (uintptr_t)d.in + (size_t)d.in_h * d.in_pitch);
        }
      }
    }
    if (use_lut && lut_desc != r.desc_idx) {   // a new sample: its normalisation look-up table
      const int words = C * 128;
      GU32 *src = tab + (r.out_w + r.out_w * sup_x + r.out_h + r.out_h * sup_y);
      uint32_t *dst = reinterpret_cast<uint32_t *>(lut);
      for (int i = tid; i < words; i += kResampleThreads) dst[i] = src[i];
      lut_desc = r.desc_idx;
    }
    // ---- the next tile's record and loads: in flight during this tile's passes ----
    if (tile + 1 < t_end) FetchTile(recs[tile + 1], descs, tid, pf);
    LdsBarrier();

    Epilogue ep;
    ep.out = (void *)(uintptr_t)r.out; ep.out_h = r.out_h; ep.out_w = r.out_w; ep.channels = C;
    ep.dtype = (int)(r.epi >> 4); ep.layout = (r.epi & 2u) ? DALIAMD_LAYOUT_CHW : DALIAMD_LAYOUT_HWC;
    ep.normalize = (r.epi & 4u) != 0; ep.mirror = (r.epi & 1u) != 0;
    ep.lut = use_lut ? lut : nullptr;
    // (with the look-up table nobody needs these: the descriptor is then not read at all on the common path)
    float mean0 = 0, mean1 = 0, mean2 = 0, mean3 = 0, inv0 = 1, inv1 = 1, inv2 = 1, inv3 = 1;
    if (!use_lut) {
      mean0 = d.mean[0]; mean1 = d.mean[1]; mean2 = d.mean[2]; mean3 = d.mean[3];
      inv0 = d.inv_std[0]; inv1 = d.inv_std[1]; inv2 = d.inv_std[2]; inv3 = d.inv_std[3];
    }
    GBytes *gwin = (GBytes *)win_addr;  // source rows when the window is not staged
    // the second pass two pixels per thread (packed multiply / add, one dword store per channel plane): the common
    // fused case - three channels into an fp16 CHW batch of even width
    const bool pair = C == 3 && use_lut && (r.epi & 2u) && (r.out_w & 1) == 0 && TW >= 2 && (r.out & 3) == 0;
    const int hw_log2 = tw_log2 - 1;

    if (vfirst) {
      // ================= vertical pass (window rows -> tmp[th][NB]), then horizontal =================
      if (staged && (pitch & 3) == 0) {
        // every row has the same shift modulo 4: 4 consecutive elements from one LDS dword per tap
        const int s4 = (int)(win_addr & 3);
        const int ndw = (NB + s4 + 3) >> 2;
        for (int jbase = 0; jbase < ndw; jbase += 3 * kVLanesPerRow) {
          const int rem = ndw - jbase;
          if (rem > 2 * kVLanesPerRow) VPassItems<3>(stage, tmp, cy, yt, tid, th, sup_y, s4, ndw, jbase);
          else if (rem > kVLanesPerRow) VPassItems<2>(stage, tmp, cy, yt, tid, th, sup_y, s4, ndw, jbase);
          else VPassItems<1>(stage, tmp, cy, yt, tid, th, sup_y, s4, ndw, jbase);
        }
      } else {
        for (int y = tid >> 6; y < th; y += kResampleThreads / 64) {
          const float *co = cy + y * sup_y;
          const int *ro = yt + y * sup_y;
          for (int e = tid & 63; e < NB; e += 64) {
            float a = 0;
            if (staged) {
              for (int k = 0; k < sup_y; k++) a += (float)stage[ro[k] + e] * co[k];
            } else {
              for (int k = 0; k < sup_y; k++) a += (float)gwin[ro[k] + e] * co[k];
            }
            tmp[y * NBp + e] = a;
          }
        }
      }
      LdsBarrier();
      const int x = pair ? (tid & ((TW >> 1) - 1)) * 2 : tid & (TW - 1);
      if (pair) {
        if (x < tw) {
          const floatx2 *co = reinterpret_cast<const floatx2 *>(cx + x);
          const intx2 *xo = reinterpret_cast<const intx2 *>(xt + x);
          const int gx = ox0 + x, half_tw = TW >> 1;
          const uint32_t em = r.even_bits >> x;   // the tile's rounding bits ride in its record (TW <= 32)
          const bool even0 = em & 1, even1 = (em >> 1) & 1;
          const uint32_t need = tw >= 32 ? 0xffffffffu : (1u << tw) - 1u;
          const bool tile_even = (r.even_bits & need) == need;
          const int nk = sup_x;
          for (int y = tid >> hw_log2; y < th; y += kResampleThreads >> hw_log2) {
            const float *trow = tmp + y * NBp;
            floatx2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f}, a2 = {0.0f, 0.0f};
            for (int k = 0; k < nk; k++) {
              const floatx2 w = co[k * half_tw];
              const intx2 o = xo[k * half_tw];
              const float *p = trow + o.x, *q = trow + o.y;
              a0 += w * floatx2{p[0], q[0]};
              a1 += w * floatx2{p[1], q[1]};
              a2 += w * floatx2{p[2], q[2]};
            }
            if (tile_even)   // (uniform) the usual case: every column of the tile rounds half-to-even
              ep.StorePair(oy0 + y, gx, RoundU8Even(a0.x), RoundU8Even(a0.y), RoundU8Even(a1.x), RoundU8Even(a1.y),
                           RoundU8Even(a2.x), RoundU8Even(a2.y));
            else
              ep.StorePair(oy0 + y, gx, RoundU8(a0.x, even0), RoundU8(a0.y, even1), RoundU8(a1.x, even0), RoundU8(a1.y, even1),
                           RoundU8(a2.x, even0), RoundU8(a2.y, even1));
          }
        }
      } else if (x < tw) {
        const float *co = cx + x;
        const int *xo = xt + x;
        const int gx = ox0 + x;
        const bool even = (r.even_bits >> x) & 1;
        for (int y = tid >> tw_log2; y < th; y += kResampleThreads >> tw_log2) {
          const float *trow = tmp + y * NBp;
          float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
          if (C == 3) {  // the common case without the per-tap channel tests
            for (int k = 0; k < sup_x; k++) {
              const float w = co[k * TW];
              const float *p = trow + xo[k * TW];
              a0 += w * p[0];
              a1 += w * p[1];
              a2 += w * p[2];
            }
          } else {
            for (int k = 0; k < sup_x; k++) {
              float w = co[k * TW];
              const float *p = trow + xo[k * TW];
              a0 += w * p[0];
              if (C > 1) a1 += w * p[1];
              if (C > 2) a2 += w * p[2];
              if (C > 3) a3 += w * p[3];
            }
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
      if (C == 3 && staged && TW >= 2 && (tw & 1) == 0) {
        // two output columns per thread: each tap is the 4 bytes at the column's offset (3 channels + 1), packed multiply / add
        const int x2 = (tid & ((TW >> 1) - 1)) * 2;
        if (x2 < tw) {
          const floatx2 *co = reinterpret_cast<const floatx2 *>(cx + x2);
          const intx2 *xo = reinterpret_cast<const intx2 *>(xt + x2);
          const int half_tw = TW >> 1;
          const int nk = sup_x;
          for (int row = tid >> hw_log2; row < nrows; row += kResampleThreads >> hw_log2) {
            const int so = row * LP + (int)((win_addr + (size_t)row * pitch) & 15);   // stage is 16-byte aligned
            floatx2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f}, a2 = {0.0f, 0.0f};
            for (int k = 0; k < nk; k++) {
              const floatx2 w = co[k * half_tw];
              const intx2 o = xo[k * half_tw];
              const uint32_t u = LdsDwordAt(stage, so + o.x), v = LdsDwordAt(stage, so + o.y);
              a0 += w * floatx2{(float)(u & 255), (float)(v & 255)};
              a1 += w * floatx2{(float)((u >> 8) & 255), (float)((v >> 8) & 255)};
              a2 += w * floatx2{(float)((u >> 16) & 255), (float)((v >> 16) & 255)};
            }
            floatx2 *tp = reinterpret_cast<floatx2 *>(tmp + row * rowlen + x2 * 3);   // rowlen = tw * 3 is even
            tp[0] = floatx2{a0.x, a1.x};
            tp[1] = floatx2{a2.x, a0.y};
            tp[2] = floatx2{a1.y, a2.y};
          }
        }
      } else if (x < tw) {
        const float *co = cx + x;
        const int *xo = xt + x;
        for (int row = tid >> tw_log2; row < nrows; row += kResampleThreads >> tw_log2) {
          float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
          if (staged) {
            const uint8_t *srow = stage + row * LP + (int)((win_addr + (size_t)row * pitch) & 15);
            for (int k = 0; k < sup_x; k++) {
              float w = co[k * TW];
              const uint8_t *p = srow + xo[k * TW];
              a0 += w * (float)p[0];
              if (C > 1) a1 += w * (float)p[1];
              if (C > 2) a2 += w * (float)p[2];
              if (C > 3) a3 += w * (float)p[3];
            }
          } else {
            GBytes *srow = gwin + (size_t)row * pitch;
            for (int k = 0; k < sup_x; k++) {
              float w = co[k * TW];
              GBytes *p = srow + xo[k * TW];
              a0 += w * (float)p[0];
              if (C > 1) a1 += w * (float)p[1];
              if (C > 2) a2 += w * (float)p[2];
              if (C > 3) a3 += w * (float)p[3];
            }
          }
          float *tp = tmp + row * rowlen + x * C;
          tp[0] = a0;
          if (C > 1) tp[1] = a1;
          if (C > 2) tp[2] = a2;
          if (C > 3) tp[3] = a3;
        }
      }
      LdsBarrier();
      const int flat_w = r.out_w * C;
#define VLAST_EVEN(f) ((f) < ((f) & ~255) + ((min(((f) & ~255) + 256, flat_w) - ((f) & ~255)) & ~15))
      if (pair) {
        const int x2 = (tid & ((TW >> 1) - 1)) * 2;
        if (x2 < tw) {
          const int fi = (ox0 + x2) * 3;
          bool e0 = true, e1 = true, e2 = true, e3 = true, e4 = true, e5 = true;
          if (flat_w & 15) {   // a row whose last 256-element block has a scalar tail
            e0 = VLAST_EVEN(fi); e1 = VLAST_EVEN(fi + 1); e2 = VLAST_EVEN(fi + 2);
            e3 = VLAST_EVEN(fi + 3); e4 = VLAST_EVEN(fi + 4); e5 = VLAST_EVEN(fi + 5);
          }
          const floatx2 *tcol = reinterpret_cast<const floatx2 *>(tmp + x2 * 3);   // 6 floats: both pixels
          const int nk = sup_y;
          for (int y = tid >> hw_log2; y < th; y += kResampleThreads >> hw_log2) {
            const float *co = cy + y * sup_y;
            const int *ro = yt + y * sup_y;
            floatx2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f}, a45 = {0.0f, 0.0f};
            for (int k = 0; k < nk; k++) {
              const float w = co[k];
              const floatx2 *p = tcol + (ro[k] >> 1);   // rowlen = tw * 3 is even
              a01 += p[0] * w;
              a23 += p[1] * w;
              a45 += p[2] * w;
            }
            if (!(flat_w & 15))
              ep.StorePair(oy0 + y, ox0 + x2, RoundU8Even(a01.x), RoundU8Even(a23.y), RoundU8Even(a01.y),
                           RoundU8Even(a45.x), RoundU8Even(a23.x), RoundU8Even(a45.y));
            else
              ep.StorePair(oy0 + y, ox0 + x2, RoundU8(a01.x, e0), RoundU8(a23.y, e3), RoundU8(a01.y, e1), RoundU8(a45.x, e4),
                           RoundU8(a23.x, e2), RoundU8(a45.y, e5));
          }
        }
      } else if (x < tw) {
        for (int y = tid >> tw_log2; y < th; y += kResampleThreads >> tw_log2) {
          const float *co = cy + y * sup_y;
          const int *ro = yt + y * sup_y;
          const float *tcol = tmp + x * C;
          float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
          for (int k = 0; k < sup_y; k++) {
            float w = co[k];
            const float *p = tcol + ro[k];
            a0 += p[0] * w;
            if (C > 1) a1 += p[1] * w;
            if (C > 2) a2 += p[2] * w;
            if (C > 3) a3 += p[3] * w;
          }
          // ResampleVert: 256-element tiles, 16-lane SIMD body then scalar tail
          int fi = (ox0 + x) * C;
          size_t cs, o = ep.Base(oy0 + y, ox0 + x, &cs);
          ep.Store(o, 0, RoundU8(a0, VLAST_EVEN(fi)), mean0, inv0);
          if (C > 1) ep.Store(o + cs, 1, RoundU8(a1, VLAST_EVEN(fi + 1)), mean1, inv1);
          if (C > 2) ep.Store(o + 2 * cs, 2, RoundU8(a2, VLAST_EVEN(fi + 2)), mean2, inv2);
          if (C > 3) ep.Store(o + 3 * cs, 3, RoundU8(a3, VLAST_EVEN(fi + 3)), mean3, inv3);
        }
      }
#undef VLAST_EVEN
    }
    LdsBarrier();   // everyone is done with this tile's LDS before the next one's data lands there
  }
}

// ---------------------------------------------------------------------------------------------
// The other element types (i16, u16, f32 input; the unrounded float result): a plain two-launch version of the same
// arithmetic - pass 0 resamples the first axis into an fp32 intermediate in the workspace (the reference's tmp surface:
// the source ROI with the first-pass axis replaced by the output size), pass 1 the second axis into the output.  One
// thread per element, coefficients and first taps from the per-sample tables.  Correct, not tuned.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float LoadElem(const void *base, int dtype, size_t byte_off, size_t elem) {
  const uint8_t __attribute__((address_space(1))) *p = (const uint8_t __attribute__((address_space(1))) *)base + byte_off;
  switch (dtype) {
    case DALIAMD_UINT8: return (float)p[elem];
    case DALIAMD_INT16: return (float)((const int16_t __attribute__((address_space(1))) *)p)[elem];
    case DALIAMD_UINT16: return (float)((const uint16_t __attribute__((address_space(1))) *)p)[elem];
    default: return ((const float __attribute__((address_space(1))) *)p)[elem];
  }
}

// ConvertSat of the scalar tails (clamp(std::round)) or the SIMD store (clamp, then round half to even)
__device__ __forceinline__ float RoundTyped(float v, int dtype, bool even) {
  if (dtype == DALIAMD_FLOAT) return v;
  const float lo = dtype == DALIAMD_INT16 ? -32768.0f : 0.0f;
  const float hi = dtype == DALIAMD_UINT8 ? 255.0f : dtype == DALIAMD_INT16 ? 32767.0f : 65535.0f;
  if (even) return rintf(fminf(fmaxf(v, lo), hi));   // NaN -> lo
  float r = roundf(v);
  if (!(r > lo)) return lo;
  return fminf(r, hi);
}

__global__ __launch_bounds__(256) void ResampleGenericKernel(const daliamdResampleDesc *__restrict__ descs, int ndesc, int pass,
                                                             int total_items, uint8_t *__restrict__ workspace) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= total_items) return;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].gen_start[pass] <= item) lo = mid; else hi = mid - 1;
  }
  // samples without work in this pass share their successor's start: step forward to the one that owns the item
  const daliamdResampleDesc &d = descs[lo];
  if (!d.generic) return;
  const int local = item - (int)d.gen_start[pass];
  const int C = d.channels;
  const TableLayout L = MakeTableLayout(d);
  GU32 *tab = (GU32 *)(workspace + d.table_off);
  GI32 *xi = (GI32 *)tab + L.xi, *yi = (GI32 *)tab + L.yi;
  GF32 *xc = (GF32 *)(tab + L.xc), *yc = (GF32 *)(tab + L.yc);
  GF32 *tmp = (GF32 *)(workspace + d.tmp_off);
  const int sup_x = d.support[0], sup_y = d.support[1];
  const int ex = d.ext[0] - 1, ey = d.ext[1] - 1;
  const bool vfirst = d.first_axis == 1;
  const int tmp_w = d.tmp_w;
  if (pass == 0 && d.generic == 2) {
    // u8 samples (extreme down-scaling): the first pass is where the work is - 25 taps for every element of an intermediate
    // as wide (or as tall) as the source region.  A thread takes FOUR neighbouring bytes of a source row per tap (vertical
    // pass first) or the C channel bytes of one source pixel (horizontal pass first) out of the aligned dword pair that
    // holds them - one address, two loads, one funnel shift instead of four byte loads with their address arithmetic -,
    // the taps in increasing k, multiply and add rounded separately: the bits of the element-by-element form below.
    // The launch is sized for one thread per element; the threads behind the last group leave at once.
    const size_t extent = (size_t)d.in_h * (size_t)d.in_pitch;           // bytes of the source buffer the descriptor vouches for
    GBytes *src = (GBytes *)d.in;
    auto four_bytes = [&](size_t off) -> uint32_t {                         // bytes off .. off + 3 (little endian)
      if (off + 8 <= extent) {
        const size_t a = off & ~(size_t)3;
        const uint32_t w0 = *(const GU32 *)(src + a), w1 = *(const GU32 *)(src + a + 4);
        return (uint32_t)((((uint64_t)w1 << 32) | w0) >> ((uint32_t)(off & 3) * 8u));   // (one v_alignbit_b32)
      }
      uint32_t v = 0;
      for (int b = 0; b < 4; b++)
        if (off + b < extent) v |= (uint32_t)src[off + b] << (8 * b);
      return v;
    };
    if (vfirst) {
      const int row_elems = tmp_w * C, ne4 = (row_elems + 3) >> 2;
      if (local >= d.tmp_h * ne4) return;
      const int y = local / ne4, e0 = (local - y * ne4) * 4;
      const size_t col = (size_t)d.lo[0] * C + e0;
      float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      const int first = yi[y];
      GF32 *co = yc + (size_t)y * sup_y;
      for (int k = 0; k < sup_y; k++) {
        const int sy = ClampI(first + k, 0, ey) + d.lo[1];
        const uint32_t v = four_bytes((size_t)sy * d.in_pitch + col);
        const float w = co[k];
        a0 += (float)(v & 255u) * w;
        a1 += (float)((v >> 8) & 255u) * w;
        a2 += (float)((v >> 16) & 255u) * w;
        a3 += (float)(v >> 24) * w;
      }
      GF32 *o = tmp + (size_t)y * row_elems + e0;
      o[0] = a0;
      if (e0 + 1 < row_elems) o[1] = a1;
      if (e0 + 2 < row_elems) o[2] = a2;
      if (e0 + 3 < row_elems) o[3] = a3;
    } else {
      if (local >= d.tmp_h * tmp_w) return;
      const int y = local / tmp_w, x = local - y * tmp_w;
      const size_t row_off = (size_t)(d.lo[1] + y) * d.in_pitch;
      float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      const int first = xi[x];
      GF32 *co = xc + (size_t)x * sup_x;
      for (int k = 0; k < sup_x; k++) {
        const int sx = ClampI(first + k, 0, ex) + d.lo[0];
        const uint32_t v = four_bytes(row_off + (size_t)sx * C);
        const float w = co[k];
        a0 += w * (float)(v & 255u);
        a1 += w * (float)((v >> 8) & 255u);
        a2 += w * (float)((v >> 16) & 255u);
        a3 += w * (float)(v >> 24);
      }
      GF32 *o = tmp + ((size_t)y * tmp_w + x) * C;
      o[0] = a0;
      if (C > 1) o[1] = a1;
      if (C > 2) o[2] = a2;
      if (C > 3) o[3] = a3;
    }
    return;
  }
  if (pass == 0) {
    const int c = local % C, x = (local / C) % tmp_w, y = local / (C * tmp_w);
    float acc = 0;
    if (vfirst) {  // rows of the whole image, columns of the source ROI
      const size_t col = (size_t)(d.lo[0] + x) * C + c;
      for (int k = 0; k < sup_y; k++) {
        const int sy = ClampI(yi[y] + k, 0, ey) + d.lo[1];
        acc += LoadElem(d.in, d.in_dtype, (size_t)sy * d.in_pitch, col) * yc[(size_t)y * sup_y + k];
      }
    } else {       // columns of the whole image, rows of the source ROI
      const size_t row_off = (size_t)(d.lo[1] + y) * d.in_pitch;
      for (int k = 0; k < sup_x; k++) {
        const int sx = ClampI(xi[x] + k, 0, ex) + d.lo[0];
        acc += xc[(size_t)x * sup_x + k] * LoadElem(d.in, d.in_dtype, row_off, (size_t)sx * C + c);
      }
    }
    tmp[local] = acc;
    return;
  }
  const int c = local % C, x = (local / C) % d.out_w, y = local / (C * d.out_w);
  float acc = 0;
  bool even;
  if (vfirst) {  // horizontal pass over the intermediate; rounding regions of ResampleHorz
    for (int k = 0; k < sup_x; k++) {
      const int sx = ClampI(xi[x] + k, 0, tmp_w - 1);
      acc += xc[(size_t)x * sup_x + k] * tmp[((size_t)y * tmp_w + sx) * C + c];
    }
    even = RoundsEven(d, x);
  } else {       // vertical pass; ResampleVert: 256-element tiles, SIMD body then scalar tail
    for (int k = 0; k < sup_y; k++) {
      const int sy = ClampI(yi[y] + k, 0, d.tmp_h - 1);
      acc += tmp[((size_t)sy * tmp_w + x) * C + c] * yc[(size_t)y * sup_y + k];
    }
    const int flat_w = d.out_w * C, f = x * C + c, t0 = f & ~255;
    even = f < t0 + ((min(t0 + 256, flat_w) - t0) / d.round_lanes) * d.round_lanes;
  }
  const int xo = d.mirror ? d.out_w - 1 - x : x;
  const size_t o = d.out_layout == DALIAMD_LAYOUT_CHW ? ((size_t)c * d.out_h + y) * d.out_w + xo : ((size_t)y * d.out_w + xo) * C + c;
  if (d.generic == 2) {
    // u8 samples the tile kernel serves badly (extreme down-scaling, SetupOne): the resampled value is rounded to u8 by the
    // rule of its column, then the fused CropMirrorNormalize epilogue - the arithmetic of Epilogue::Store, whose look-up
    // table holds exactly these results
    const uint32_t v = RoundU8Slow(acc, even);
    Epilogue ep{d.out, d.out_h, d.out_w, C, d.out_dtype, d.out_layout, d.normalize, d.mirror, nullptr};
    ep.Store(o, c, v, SEL4(c, d.mean[0], d.mean[1], d.mean[2], d.mean[3]), SEL4(c, d.inv_std[0], d.inv_std[1], d.inv_std[2], d.inv_std[3]));
    return;
  }
  const float r = RoundTyped(acc, d.out_dtype, even);
  switch (d.out_dtype) {
    case DALIAMD_UINT8: ((uint8_t __attribute__((address_space(1))) *)d.out)[o] = (uint8_t)r; break;
    case DALIAMD_INT16: ((int16_t __attribute__((address_space(1))) *)d.out)[o] = (int16_t)r; break;
    case DALIAMD_UINT16: ((uint16_t __attribute__((address_space(1))) *)d.out)[o] = (uint16_t)r; break;
    default: ((float __attribute__((address_space(1))) *)d.out)[o] = r; break;
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
// the tabulated filters: {size, anchor 1, scale (size - 1) / 2}, Lanczos rescaled to 6 and cubic to 4 at creation
// (InitFilters, resampling_filters.cu:66-108), then the per-use rescale (resampling_filters.cu:115-137)
static HostFilter Tabulated(int size) {
  HostFilter f;
  f.num_coeffs = size;
  f.anchor = 1;
  f.scale = (size - 1) * 0.5f;
  return f;
}
static HostFilter Gaussian(float sigma) {
  HostFilter f = Tabulated(DALIAMD_RF_GAUSSIAN_SIZE);
  f.Rescale(std::max(1.0f, static_cast<float>(4 * M_SQRT2) * sigma));
  return f;
}
static HostFilter Lanczos3(float radius) {
  HostFilter f = Tabulated(DALIAMD_RF_LANCZOS_SIZE);
  f.Rescale(6);
  f.Rescale(2.0f * std::max(3.0f, radius));
  return f;
}
static HostFilter Cubic(float radius) {
  HostFilter f = Tabulated(DALIAMD_RF_CUBIC_SIZE);
  f.Rescale(4);
  f.Rescale(2.0f * std::max(2.0f, radius));
  return f;
}

static int SetupOne(const daliamdResampleArgs &a, daliamdResampleDesc &d, int index) {
  DALIAMD_REQUIRE(a.in_h > 0 && a.in_w > 0 && a.out_h > 0 && a.out_w > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: sample %d has an empty input or output", index);
  DALIAMD_REQUIRE(a.channels >= 1 && a.channels <= 4, DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdResampleSetup: sample %d: %d channels (supported: 1..4)", index, a.channels);
  {
    const int esize = a.in_dtype == DALIAMD_UINT8 ? 1 : a.in_dtype == DALIAMD_FLOAT ? 4 : 2;
    DALIAMD_REQUIRE(a.in_pitch >= a.in_w * a.channels * esize, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdResampleSetup: sample %d: pitch %d < row bytes", index, a.in_pitch);
  }
  DALIAMD_REQUIRE(a.min_filter >= DALIAMD_INTERP_NN && a.min_filter <= DALIAMD_INTERP_GAUSSIAN &&
                  a.mag_filter >= DALIAMD_INTERP_NN && a.mag_filter <= DALIAMD_INTERP_GAUSSIAN, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleSetup: sample %d: unknown interpolation type", index);
  bool generic = a.in_dtype != DALIAMD_UINT8 || a.unrounded;
  if (!generic) {
    DALIAMD_REQUIRE(a.out_dtype == DALIAMD_UINT8 || a.out_dtype == DALIAMD_FLOAT16 || a.out_dtype == DALIAMD_FLOAT,
                    DALIAMD_ERROR_UNSUPPORTED, "daliamdResampleSetup: unsupported output type %d", a.out_dtype);
  } else {
    DALIAMD_REQUIRE(a.in_dtype == DALIAMD_UINT8 || a.in_dtype == DALIAMD_INT16 || a.in_dtype == DALIAMD_UINT16 ||
                    a.in_dtype == DALIAMD_FLOAT, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdResampleSetup: sample %d: unsupported input type %d (u8, i16, u16, f32)", index, a.in_dtype);
    DALIAMD_REQUIRE(a.unrounded ? a.out_dtype == DALIAMD_FLOAT : a.out_dtype == a.in_dtype, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdResampleSetup: sample %d: the output type must be the input type, or FLOAT with `unrounded`", index);
    DALIAMD_REQUIRE(!a.normalize, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdResampleSetup: sample %d: the fused normalisation needs u8 input", index);
  }
  memset(&d, 0, sizeof(d));
  d.in_dtype = a.in_dtype; d.unrounded = a.unrounded; d.generic = generic ? 1 : 0;
  // lanes of the reference's SIMD store for the output type: 16 bytes / sizeof(Out) (simd.h:314-327)
  // (the fused path rounds to u8 before its normalisation epilogue, whatever the final type)
  d.round_lanes = !generic || a.out_dtype == DALIAMD_UINT8 ? 16 : (a.out_dtype == DALIAMD_INT16 || a.out_dtype == DALIAMD_UINT16) ? 8 : 4;
  d.in = a.in; d.out = a.out;
  d.in_h = a.in_h; d.in_w = a.in_w; d.channels = a.channels; d.in_pitch = a.in_pitch;
  d.out_h = a.out_h; d.out_w = a.out_w;
  d.out_dtype = a.out_dtype; d.out_layout = a.out_layout; d.normalize = a.normalize; d.mirror = a.mirror;
  for (int c = 0; c < 4; c++) { d.mean[c] = a.mean[c]; d.inv_std[c] = a.inv_std[c]; }
  d.use_lut = a.normalize && a.out_dtype == DALIAMD_FLOAT16;  // fused CMN to fp16: the epilogue is a 256-entry look-up

  // (a.full_h > 0: the buffer is a window of the image the region of interest refers to - the arithmetic below is that
  // image's, only the addresses are the window's)
  const bool windowed = a.full_h > 0;
  if (windowed)
    DALIAMD_REQUIRE(a.use_roi && a.full_w > 0 && a.org_y >= 0 && a.org_x >= 0 && a.org_y + a.in_h <= a.full_h &&
                    a.org_x + a.in_w <= a.full_w, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdResampleSetup: sample %d: a %d x %d window at (%d, %d) of a %d x %d image needs a region of interest "
                    "and must lie inside the image", index, a.in_h, a.in_w, a.org_y, a.org_x, a.full_h, a.full_w);
  const int in_size[2] = {windowed ? a.full_w : a.in_w, windowed ? a.full_h : a.in_h};
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
    // DefaultFilterRadius (params.h:40-55), GetResamplingFilter (resampling_setup.cc:27-44)
    const bool shrink = aa && (in_sz > out_size[axis]);
    const float ratio = in_sz / out_size[axis];
    HostFilter f;
    int kind;
    switch (type) {
      case DALIAMD_INTERP_NN: f.num_coeffs = 0; f.anchor = 0; f.scale = 1; kind = DALIAMD_FK_NN; break;
      case DALIAMD_INTERP_LINEAR: f = Triangular(1.0f); kind = DALIAMD_FK_TRIANGULAR; break;
      case DALIAMD_INTERP_TRIANGULAR: f = Triangular(shrink ? ratio : 1); kind = DALIAMD_FK_TRIANGULAR; break;
      case DALIAMD_INTERP_GAUSSIAN: f = Gaussian((float)((shrink ? ratio : 1) * 0.5f / M_SQRT2)); kind = DALIAMD_FK_GAUSSIAN; break;
      case DALIAMD_INTERP_CUBIC: f = Cubic(shrink ? 2 * ratio : 2); kind = DALIAMD_FK_CUBIC; break;
      default: f = Lanczos3(shrink ? 3 * ratio : 3); kind = DALIAMD_FK_LANCZOS3; break;
    }
    d.filter_kind[axis] = kind;
    d.origin[axis] = roi_start;
    d.scale[axis] = (roi_end - roi_start) / out_size[axis];
    int support = f.num_coeffs ? f.Support() : 1;
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
  DALIAMD_REQUIRE((d.filter_kind[0] == DALIAMD_FK_NN) == (d.filter_kind[1] == DALIAMD_FK_NN), DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdResampleSetup: sample %d: nearest-neighbour resampling on one axis only is not supported", index);
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
  if (windowed) {
    const int org[2] = {a.org_x, a.org_y}, win[2] = {a.in_w, a.in_h};
    for (int axis = 0; axis < 2; axis++) {
      // [roi_lo, roi_hi) is every source position a tap can take on the axis after the clamp to the image
      DALIAMD_REQUIRE(org[axis] <= roi_lo[axis] && roi_hi[axis] <= org[axis] + win[axis], DALIAMD_ERROR_INVALID_ARGUMENT,
                      "daliamdResampleSetup: sample %d: the window [%d, %d) does not hold the source range [%d, %d) of axis %d",
                      index, org[axis], org[axis] + win[axis], roi_lo[axis], roi_hi[axis], axis);
      d.lo[axis] -= org[axis];
    }
  }

  // rounding regions of an H-last pass: the reference's row loop runs over four column regions (left border, the
  // overlap of both borders, regular, right border); inside each, whole groups of `round_lanes` columns take the SIMD
  // store (half to even), the rest of the region the scalar tail (half away).  Any width.
  for (int r = 0; r < 4; r++) d.round_lo[r] = d.round_hi[r] = 0;
  if (d.first_axis == 1) {
    int ow = a.out_w, in_w = d.ext[0], sup = d.support[0];
    float start = FilterStart(d.origin[0], d.scale[0], d.fanchor[0]);
    auto first_tap = [&](int x) { float f0; return FirstTap(x, d.scale[0], start, &f0); };
    bool flipped = first_tap(ow - 1) < first_tap(0);
    int first_regular = 0, last_regular = ow - 1;
    if (flipped) {
      while (first_regular < ow && first_tap(first_regular) + sup > in_w) first_regular++;
      while (last_regular >= 0 && first_tap(last_regular) < 0) last_regular--;
    } else {
      while (first_regular < ow && first_tap(first_regular) < 0) first_regular++;
      while (last_regular >= 0 && first_tap(last_regular) + sup > in_w) last_regular--;
    }
    int bounds[5] = {0, std::min(first_regular, last_regular + 1), first_regular, last_regular + 1, ow};
    int x = 0;
    for (int r = 0; r < 4; r++) {
      const int ox1 = bounds[r + 1], lanes = d.round_lanes;
      d.round_lo[r] = x;
      if (ox1 > x) x += (ox1 - x) / lanes * lanes;
      d.round_hi[r] = x;
      x = std::max(x, ox1);
    }
  }

  // tile selection: keep tables + staged source window + tmp inside the LDS budget.  When even small tiles
  // cannot hold their source window (extreme down-scaling) fall back to reading the source from global memory.
  auto lds_need = [&](int tw_, int th_, bool staged) -> size_t {
    size_t tables = 2 * ((size_t)th_ * d.support[1] + (size_t)tw_ * d.support[0]);
    size_t ncols = (size_t)std::ceil(tw_ * std::abs(d.scale[0])) + d.support[0] + 2;
    size_t nrows = (size_t)std::ceil(th_ * std::abs(d.scale[1])) + d.support[1] + 2;
    ncols = std::min<size_t>(ncols, a.in_w);
    nrows = std::min<size_t>(nrows, a.in_h);
    size_t lp = (size_t)StagedRowPitch((int)(ncols * a.channels));
    size_t stage = staged ? nrows * lp : 0;
    // (V-first: rows of the intermediate on the window's dword grid, TmpRowPitch: up to 3 + 3 elements more)
    size_t tmp_elems = d.first_axis == 1 ? (size_t)th_ * (ncols * a.channels + 6) : nrows * (size_t)tw_ * a.channels;
    return kLutLdsBytes + tables * 4 + 32 + stage + tmp_elems * 4;
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
  // Extreme down-scaling (round 6; a 12-megapixel photograph into 224 x 224: scale 10-13, 25 taps per axis): the source
  // window of even a handful of output pixels fills the LDS budget - tiles of 4 x 4 outputs whose windows overlap three
  // times, 3 000 workgroup passes per image, 1.2 ms per batch for five such images among 251 ordinary ones.  Such a
  // sample takes the two-launch path instead (first-axis pass of the whole region into an fp32 intermediate in the
  // workspace, second pass + the fused epilogue from it: generic = 2), about as much arithmetic and no overlap.
  // DALI_AMD_RESAMPLE_TWO_PASS_AREA: tiles of at most this many outputs go there (default 64; 0 = never, large = always).
  static const int two_pass_area = [] { const char *e = getenv("DALI_AMD_RESAMPLE_TWO_PASS_AREA"); return e ? atoi(e) : 64; }();
  if (!generic && (tw * th <= two_pass_area || (!staged && two_pass_area > 0))) {
    generic = true;
    d.generic = 2;
  }
  if (generic) {  // the two-launch path: no tiles, an fp32 intermediate of the reference's shape in the workspace
    d.tiles_x = d.tiles_y = 0;
    d.tmp_w = d.first_axis == 0 ? a.out_w : d.ext[0];
    d.tmp_h = d.first_axis == 1 ? a.out_h : d.ext[1];
    d.use_lut = 0;
  }
  d.lds_bytes = (int)lds_need(tw, th, staged);
  return DALIAMD_SUCCESS;
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdResampleSetup(const daliamdResampleArgs *args, int n, daliamdResampleDesc *descs,
                                     daliamdResamplePlan *plan) {
  DALIAMD_REQUIRE(args && descs && plan && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdResampleSetup: NULL argument");
  int wg = 0, lds = 0, entries = 0;
  size_t ws = 0;
  int64_t gen[2] = {0, 0};
  for (int i = 0; i < n; i++) {
    int rc = daliamd::SetupOne(args[i], descs[i], i);
    if (rc != DALIAMD_SUCCESS) return (daliamdResult_t)rc;
    descs[i].wg_start = wg;
    wg += descs[i].tiles_x * descs[i].tiles_y;
    if (!descs[i].generic) lds = lds > descs[i].lds_bytes ? lds : descs[i].lds_bytes;
    descs[i].table_off = (int64_t)ws;
    descs[i].tab_start = entries;
    ws += ((size_t)daliamd::MakeTableLayout(descs[i]).words * 4 + 15) & ~(size_t)15;
    entries += daliamd::TableEntries(descs[i]);
    descs[i].gen_start[0] = gen[0];
    descs[i].gen_start[1] = gen[1];
    if (descs[i].generic) {
      descs[i].tmp_off = (int64_t)ws;
      ws += ((size_t)descs[i].tmp_w * descs[i].tmp_h * descs[i].channels * 4 + 15) & ~(size_t)15;
      gen[0] += (int64_t)descs[i].tmp_w * descs[i].tmp_h * descs[i].channels;
      gen[1] += (int64_t)descs[i].out_w * descs[i].out_h * descs[i].channels;
    }
  }
  DALIAMD_REQUIRE(gen[0] < (1ll << 31) && gen[1] < (1ll << 31), DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdResampleSetup: too many elements for the generic path in one batch");
  plan->num_tiles = wg;
  plan->lds_bytes = lds;
  plan->table_entries = entries;
  // ... and behind the tables (and intermediates) one 128-byte record per tile
  plan->workspace_bytes = ((ws + 127) & ~(size_t)127) + (size_t)wg * sizeof(daliamd::TileRec);
  plan->generic_items[0] = gen[0];
  plan->generic_items[1] = gen[1];
  return DALIAMD_SUCCESS;
}

// The tabulated filter windows live in device memory, one copy per device, uploaded synchronously at the first Run
// of the process on that device (1.5 KB; built with the host's libm like the reference's InitFilters).
static const float *DeviceFilterTables() {
  static std::mutex mu;
  static float *tables[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tables[dev]) {
    float host[DALIAMD_RF_TOTAL];
    daliamdBuildFilterTables(host);
    float *p = nullptr;
    if (hipMalloc(&p, sizeof(host)) != hipSuccess) return nullptr;
    if (hipMemcpy(p, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    tables[dev] = p;
  }
  return tables[dev];
}

// The two halves of daliamdResampleRun.  The first needs nothing but the descriptor table: a caller can run it on a side
// stream while `stream` is still busy with the kernels that produce the source images (order the two with an event).
static daliamdResult_t CheckResamplePlan(const daliamdResampleDesc *descs_dev, int n, const daliamdResamplePlan *plan,
                                         void *workspace_dev, bool *nothing) {
  *nothing = true;
  if (n == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && plan && n > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdResampleRun: invalid argument");
  if (plan->num_tiles == 0 && plan->generic_items[1] == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(plan->lds_bytes >= 0 && plan->lds_bytes <= daliamd::kMaxLds, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleRun: invalid plan");
  DALIAMD_REQUIRE(workspace_dev && plan->workspace_bytes > 0 && plan->table_entries > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleRun: the table workspace is missing (size it with daliamdResampleSetup)");
  DALIAMD_REQUIRE(plan->workspace_bytes >= (size_t)plan->num_tiles * sizeof(daliamd::TileRec), DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdResampleRun: workspace too small");
  *nothing = false;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdResampleRunTables(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                         const daliamdResamplePlan *plan, void *workspace_dev) {
  bool nothing;
  daliamdResult_t rc = CheckResamplePlan(descs_dev, n, plan, workspace_dev, &nothing);
  if (rc != DALIAMD_SUCCESS || nothing) return rc;
  const float *filter_tables = DeviceFilterTables();
  DALIAMD_REQUIRE(filter_tables, DALIAMD_ERROR_HIP, "daliamdResampleRun: could not set up the filter tables on the device");
  const int num_tiles = plan->num_tiles, table_entries = plan->table_entries;
  const size_t tile_rec_off = plan->workspace_bytes - (size_t)num_tiles * sizeof(daliamd::TileRec);   // as laid out by Setup
  daliamd::KernelTimer timer("ResampleTablesKernel", (hipStream_t)stream);
  const int total = table_entries + num_tiles;
  hipLaunchKernelGGL(daliamd::ResampleTablesKernel, dim3((total + daliamd::kTableThreads - 1) / daliamd::kTableThreads),
                     dim3(daliamd::kTableThreads), 0, (hipStream_t)stream, descs_dev, n, table_entries, num_tiles,
                     static_cast<uint8_t *>(workspace_dev), tile_rec_off, filter_tables);
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdResampleRunPasses(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                         const daliamdResamplePlan *plan, void *workspace_dev) {
  bool nothing;
  daliamdResult_t rc = CheckResamplePlan(descs_dev, n, plan, workspace_dev, &nothing);
  if (rc != DALIAMD_SUCCESS || nothing) return rc;
  const int num_tiles = plan->num_tiles;
  const size_t tile_rec_off = plan->workspace_bytes - (size_t)num_tiles * sizeof(daliamd::TileRec);
  uint8_t *ws = static_cast<uint8_t *>(workspace_dev);
  if (num_tiles > 0) {
    daliamd::KernelTimer timer("ResampleKernel", (hipStream_t)stream);
    const int wgs = (num_tiles + daliamd::kTilesPerWg - 1) / daliamd::kTilesPerWg;
    hipLaunchKernelGGL(daliamd::ResampleKernel, dim3(daliamd::XcdGrid(wgs)), dim3(daliamd::kResampleThreads), plan->lds_bytes,
                       (hipStream_t)stream, descs_dev, n, wgs, num_tiles, static_cast<const uint8_t *>(ws), tile_rec_off);
  }
  if (plan->generic_items[1] > 0) {  // samples of other element types: first-axis pass into the workspace, then the second
    daliamd::KernelTimer timer("ResampleGenericKernel", (hipStream_t)stream);
    for (int pass = 0; pass < 2; pass++) {
      const int items = (int)plan->generic_items[pass];
      if (items == 0) continue;
      hipLaunchKernelGGL(daliamd::ResampleGenericKernel, dim3((items + 255) / 256), dim3(256), 0, (hipStream_t)stream, descs_dev,
                         n, pass, items, ws);
    }
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdResampleRun(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                   const daliamdResamplePlan *plan, void *workspace_dev) {
  daliamdResult_t rc = daliamdResampleRunTables(stream, descs_dev, n, plan, workspace_dev);
  return rc != DALIAMD_SUCCESS ? rc : daliamdResampleRunPasses(stream, descs_dev, n, plan, workspace_dev);
}

}  // extern "C"
