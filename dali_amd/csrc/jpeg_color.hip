// Batched chroma upsampling + YCbCr->RGB (or gray->RGB) for gfx950.
//
// Arithmetic: libjpeg-turbo jdsample.c "fancy" triangle upsampling (h2v1, h2v2, h1v2), box
// replication for the other integral ratios, jdcolor.c 16-bit fixed-point BT.601 full-range
// conversion -- what the reference's CPU decoder produces (fancy upsampling is always on for
// the CPU backend: dali/operators/imgcodec/image_decoder.h:297-305).  Integer => bit-exact.
// The first/last-column special cases of jdsample.c are algebraically the general formula with
// the neighbour index clamped to [0, downsampled_width-1]; rows above/below the component are
// the replicated edge rows jdmainct.c provides as context => clamp to [0, downsampled_height-1].
//
// Mapping: a thread produces 8 consecutive output pixels (24 bytes) of kRowsPerThread rows: an 8-byte luma load and
// <= 3 small chroma loads per chroma row, three 8-byte stores per row when the output pitch allows (pitch % 8 == 0),
// byte stores otherwise; the common 4:2:0 case issues all its loads before the first use (ColorRows420).
// Workgroup = 32 x 8 threads = a 256 x (8 * kRowsPerThread) pixel tile.
// HBM traffic per pixel: 1 + 2/(h*v ratio) bytes read, 3 bytes written.
#include "common.h"
#include "jpeg_color_math.h"

namespace daliamd {

constexpr int kColorThreads = 256;
constexpr int kTileW = 256;  // pixels
#ifndef DALIAMD_COLOR_ROWS
#define DALIAMD_COLOR_ROWS 8
#endif
constexpr int kRowsPerThread = DALIAMD_COLOR_ROWS;  // consecutive rows per thread (even): the per-image set-up is paid once for all of them
constexpr int kTileH = 8 * kRowsPerThread;

// Fetches the 8 upsampled samples of one component for pixels x0..x0+7 of output row y (any x0: with a region of
// interest the 8-pixel groups are aligned to the region, not to the image).
__device__ __forceinline__ void UpsampleRow8(GBytes *__restrict__ plane, int pitch, int mode, int hx,
                                             int vx, int dw, int dh, int x0, int y, int out[8]) {
  if (mode == kFull) {
    GBytes *p = plane + (size_t)y * pitch;
    if ((x0 & 7) == 0) {
      u32x2 v = *reinterpret_cast<GPair *>(p + x0);  // planes are padded to 8-sample blocks
#pragma unroll
      for (int i = 0; i < 4; i++) { out[i] = (v.x >> (8 * i)) & 255; out[4 + i] = (v.y >> (8 * i)) & 255; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) out[i] = p[min(x0 + i, pitch - 1)];
    }
  } else if (mode == kH2V1) {
    GBytes *p = plane + (size_t)y * pitch;
    int k0 = x0 >> 1;
    int s[7];
    if (((k0 | pitch) & 3) == 0 && (reinterpret_cast<uintptr_t>(plane) & 3) == 0) {
      LoadSamples7(p, pitch, k0, s);
      ClampRight7(s, k0, dw);
    } else {
#pragma unroll
      for (int i = 0; i < 7; i++) s[i] = p[ClampI(k0 - 1 + i, 0, dw - 1)];
    }
    TriangleX8<2, 1, 2>(s, x0 & 1, out);
  } else if (mode == kH2V2) {
    int r = y >> 1;
    int r1 = ClampI((y & 1) ? r + 1 : r - 1, 0, dh - 1);
    GBytes *p0 = plane + (size_t)r * pitch, *p1 = plane + (size_t)r1 * pitch;
    int k0 = x0 >> 1;
    int s[7];
    if (((k0 | pitch) & 3) == 0 && (reinterpret_cast<uintptr_t>(plane) & 3) == 0) {
      int a[7], b[7];
      LoadSamples7(p0, pitch, k0, a);
      LoadSamples7(p1, pitch, k0, b);
#pragma unroll
      for (int i = 0; i < 7; i++) s[i] = a[i] * 3 + b[i];
      ClampRight7(s, k0, dw);
    } else {
#pragma unroll
      for (int i = 0; i < 7; i++) {
        int k = ClampI(k0 - 1 + i, 0, dw - 1);
        s[i] = p0[k] * 3 + p1[k];
      }
    }
    TriangleX8<4, 8, 7>(s, x0 & 1, out);
  } else if (mode == kH1V2) {
    int r = y >> 1;
    int r1 = ClampI((y & 1) ? r + 1 : r - 1, 0, dh - 1);
    int bias = (y & 1) ? 2 : 1;
    GBytes *p0 = plane + (size_t)r * pitch, *p1 = plane + (size_t)r1 * pitch;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      int x = min(x0 + i, pitch - 1);
      out[i] = (p0[x] * 3 + p1[x] + bias) >> 2;
    }
  } else {
    GBytes *p = plane + (size_t)(y / vx) * pitch;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = p[min((x0 + i) / hx, pitch - 1)];
  }
}

__device__ __forceinline__ int ModeOf(const daliamdJpegColorDesc &d, int c, int hmax, int vmax) {
  int h = d.h_samp[c], v = d.v_samp[c];
  if (h == hmax && v == vmax) return kFull;
  if (h * 2 == hmax && v == vmax && d.down_w[c] > 2) return kH2V1;
  if (h == hmax && v * 2 == vmax) return kH1V2;
  if (h * 2 == hmax && v * 2 == vmax && d.down_w[c] > 2) return kH2V2;
  return kBox;
}

// ---- the common case in one piece: YCbCr 4:2:0 (luma full, both chroma planes h2v2), 8-pixel groups aligned to the
// planes, upright output.  A thread produces 8 x ROWS pixels; the output rows y0..y0+ROWS-1 (y0 even) need the chroma
// rows r-1 .. r+ROWS/2 (r = y0/2).  ALL loads of the thread (2 luma + 6 chroma dwords per row pair, + 12) are issued
// before the first use, so one memory round trip covers all its rows (the row-by-row form cannot overlap them: the
// stores may alias the planes), and the descriptor walk in front of them is paid once per 8 x ROWS pixels.
template <int ROWS>
__device__ __forceinline__ void ColorRows420(const daliamdJpegColorDesc &d, int x0, int y0, int rx1, int ry1, int out_x0,
                                             int out_y0) {
  static_assert(ROWS >= 2 && ROWS % 2 == 0, "row pairs");
  constexpr int CR = ROWS / 2 + 2;
  const int k0 = x0 >> 1, r = y0 >> 1;
  uint32_t ca[2][CR], cb[2][CR], cc[2][CR];  // [component][chroma row r-1+j]: dwords left of / at / right of k0
  u32x2 luma[ROWS];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    GBytes *plane = (GBytes *)d.plane[1 + c];
    const int pitch = d.pitch[1 + c], dh = d.down_h[1 + c];
    const int ka = max(k0 - 4, 0), kc = min(k0 + 4, pitch - 4);
#pragma unroll
    for (int j = 0; j < CR; j++) {
      GBytes *row = plane + (size_t)ClampI(r - 1 + j, 0, dh - 1) * pitch;
      ca[c][j] = *reinterpret_cast<GWords *>(row + ka);
      cb[c][j] = *reinterpret_cast<GWords *>(row + k0);
      cc[c][j] = *reinterpret_cast<GWords *>(row + kc);
    }
  }
  {
    GBytes *plane = (GBytes *)d.plane[0];
    const int pitch = d.pitch[0];
#pragma unroll
    for (int j = 0; j < ROWS; j++) luma[j] = *reinterpret_cast<GPair *>(plane + (size_t)min(y0 + j, ry1 - 1) * pitch + x0);
  }
  const int npx = min(8, rx1 - x0);
  const int dw1 = d.down_w[1], dw2 = d.down_w[2];
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int y = y0 + j;
    if (y >= ry1) break;
    // output row y0+j: the nearer chroma row is r + (j >> 1), the further one the row above it (j even) or below (j odd)
    const int near = 1 + (j >> 1), far = (j & 1) ? near + 1 : near - 1;
    int up[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      int n7[7], f7[7], s[7];
      Chroma7(ca[c][near], cb[c][near], cc[c][near], k0 == 0, n7);
      Chroma7(ca[c][far], cb[c][far], cc[c][far], k0 == 0, f7);
#pragma unroll
      for (int i = 0; i < 7; i++) s[i] = n7[i] * 3 + f7[i];
      ClampRight7(s, k0, c == 0 ? dw1 : dw2);
      TriangleX8<4, 8, 7>(s, false, up[c]);
    }
    uint32_t px[24];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int yy = (int)(((i < 4 ? luma[j].x : luma[j].y) >> (8 * (i & 3))) & 255);
      const int u = up[0][i] - 128, v = up[1][i] - 128;
      const int rr = yy + ((FIXC(1.40200) * v + ONE_HALF) >> SCALEBITS);
      const int gg = yy + (((-FIXC(0.34414)) * u + ONE_HALF + (-FIXC(0.71414)) * v) >> SCALEBITS);
      const int bb = yy + ((FIXC(1.77200) * u + ONE_HALF) >> SCALEBITS);
      px[3 * i] = Clamp8(rr); px[3 * i + 1] = Clamp8(gg); px[3 * i + 2] = Clamp8(bb);
    }
    GOutBytes *o = (GOutBytes *)d.out + (size_t)(y - out_y0) * d.out_pitch + (size_t)(x0 - out_x0) * 3;
    if (npx == 8) {
      uint32_t w[6];
#pragma unroll
      for (int q = 0; q < 6; q++) w[q] = px[4 * q] | (px[4 * q + 1] << 8) | (px[4 * q + 2] << 16) | (px[4 * q + 3] << 24);
      GOutPair *o2 = reinterpret_cast<GOutPair *>(o);
      o2[0] = u32x2{w[0], w[1]};
      o2[1] = u32x2{w[2], w[3]};
      o2[2] = u32x2{w[4], w[5]};
    } else {
      for (int i = 0; i < npx * 3; i++) o[i] = (uint8_t)px[i];
    }
  }
}

// The same for a window that starts anywhere (region-of-interest decode: the 8-pixel groups follow the window's origin so
// that the output rows keep their store alignment; x0 may be odd, y0 may be odd).  The chroma samples k0 - 1 .. k0 + 5 of a
// row come from the three aligned dwords that hold them and a byte funnel shift, the 8 luma samples likewise; which chroma
// row is the nearer one follows the parity of the row (jdsample.c h2v2_fancy_upsample).  Same arithmetic, same bits; a
// random crop window met the generic row-by-row path before (0.147 ms for half the pixels of what the aligned path does
// in 0.119: tools/roi_kernel_times.py).
template <int ROWS>
__device__ __forceinline__ void ColorRows420Any(const daliamdJpegColorDesc &d, int x0, int y0, int rx1, int ry1, int out_x0,
                                                int out_y0) {
  static_assert(ROWS >= 2 && ROWS % 2 == 0, "row pairs");
  constexpr int CR = ROWS / 2 + 2;
  const int k0 = x0 >> 1, r = y0 >> 1;
  const int cbase = (k0 - 1) & ~3;                 // dword that holds sample k0 - 1 (-4 when k0 == 0)
  const uint32_t csh = (uint32_t)((k0 - 1) & 3);
  uint32_t c0[2][CR], c1[2][CR], c2[2][CR];        // [component][chroma row r-1+j]: the three dwords from cbase
  uint32_t l0[ROWS], l1[ROWS], l2[ROWS];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    GBytes *plane = (GBytes *)d.plane[1 + c];
    const int pitch = d.pitch[1 + c], dh = d.down_h[1 + c];
    const int ka = max(cbase, 0), kb = min(cbase + 4, pitch - 4), kc = min(cbase + 8, pitch - 4);
#pragma unroll
    for (int j = 0; j < CR; j++) {
      GBytes *row = plane + (size_t)ClampI(r - 1 + j, 0, dh - 1) * pitch;
      c0[c][j] = *reinterpret_cast<GWords *>(row + ka);
      c1[c][j] = *reinterpret_cast<GWords *>(row + kb);
      c2[c][j] = *reinterpret_cast<GWords *>(row + kc);
    }
  }
  const int lbase = x0 & ~3;
  const uint32_t lsh = (uint32_t)(x0 & 3);
  {
    GBytes *plane = (GBytes *)d.plane[0];
    const int pitch = d.pitch[0];
    const int la = lbase, lb = min(lbase + 4, pitch - 4), lc = min(lbase + 8, pitch - 4);
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
      GBytes *row = plane + (size_t)min(y0 + j, ry1 - 1) * pitch;
      l0[j] = *reinterpret_cast<GWords *>(row + la);
      l1[j] = *reinterpret_cast<GWords *>(row + lb);
      l2[j] = *reinterpret_cast<GWords *>(row + lc);
    }
  }
  const int npx = min(8, rx1 - x0);
  const int dw1 = d.down_w[1], dw2 = d.down_w[2];
  const bool odd_x = x0 & 1;
  auto window7 = [&](uint32_t a, uint32_t b, uint32_t cc, int s7[7]) {
    const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, csh), hi = __builtin_amdgcn_alignbyte(cc, b, csh);
    s7[0] = k0 == 0 ? (int)((lo >> 8) & 255) : (int)(lo & 255);   // the left neighbour of sample 0 is sample 0
    s7[1] = (int)((lo >> 8) & 255); s7[2] = (int)((lo >> 16) & 255); s7[3] = (int)(lo >> 24);
    s7[4] = (int)(hi & 255); s7[5] = (int)((hi >> 8) & 255); s7[6] = (int)((hi >> 16) & 255);
  };
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int y = y0 + j;
    if (y >= ry1) break;
    // output row y: the nearer chroma row is y >> 1, the further one the row above it (y even) or below it (y odd)
    const int near = (y >> 1) - (r - 1), far = (y & 1) ? near + 1 : near - 1;
    int up[2][8];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      int n7[7], f7[7], s[7];
      uint32_t na = c0[c][0], nb = c1[c][0], nc = c2[c][0], fa = na, fb = nb, fc = nc;
#pragma unroll
      for (int q = 0; q < CR; q++) {   // (a select chain: the row indices depend on the parity of y0)
        if (q == near) { na = c0[c][q]; nb = c1[c][q]; nc = c2[c][q]; }
        if (q == far) { fa = c0[c][q]; fb = c1[c][q]; fc = c2[c][q]; }
      }
      window7(na, nb, nc, n7);
      window7(fa, fb, fc, f7);
#pragma unroll
      for (int i = 0; i < 7; i++) s[i] = n7[i] * 3 + f7[i];
      ClampRight7(s, k0, c == 0 ? dw1 : dw2);
      TriangleX8<4, 8, 7>(s, odd_x, up[c]);
    }
    const uint32_t ylo = __builtin_amdgcn_alignbyte(l1[j], l0[j], lsh), yhi = __builtin_amdgcn_alignbyte(l2[j], l1[j], lsh);
    int yy[8];
#pragma unroll
    for (int i = 0; i < 8; i++) yy[i] = (int)(((i < 4 ? ylo : yhi) >> (8 * (i & 3))) & 255);
    uint32_t px[24];
    YccToRgb8(yy, up, px);
    StoreRgb8((GOutBytes *)d.out + (size_t)(y - out_y0) * d.out_pitch + (size_t)(x0 - out_x0) * 3, px, npx);
  }
}

// The common case: YCbCr 4:2:0 to RGB, upright, aligned planes.
__host__ __device__ inline bool Fast420(const daliamdJpegColorDesc &d) {
  if (d.out_format != DALIAMD_JPEG_OUT_RGB || d.color != DALIAMD_JPEG_YCC || d.orientation > 1) return false;
  const int hmax = max(d.h_samp[0], max(d.h_samp[1], d.h_samp[2])), vmax = max(d.v_samp[0], max(d.v_samp[1], d.v_samp[2]));
  if (d.h_samp[0] != hmax || d.v_samp[0] != vmax) return false;
  for (int c = 1; c < 3; c++)
    if (d.h_samp[c] * 2 != hmax || d.v_samp[c] * 2 != vmax || d.down_w[c] <= 2) return false;
  // (the window may start anywhere: ColorRows420Any)
  return ((d.out_pitch & 7) == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 7) == 0) &&
         ((d.pitch[0] & 7) | (d.pitch[1] & 3) | (d.pitch[2] & 3)) == 0 && (reinterpret_cast<uintptr_t>(d.plane[0]) & 7) == 0 &&
         ((reinterpret_cast<uintptr_t>(d.plane[1]) | reinterpret_cast<uintptr_t>(d.plane[2])) & 3) == 0;
}

// kConvert: samples whose RGB result is converted on to BGR / YCbCr / gray (the conversion code costs 100 registers)
__host__ __device__ inline bool NeedsConvert(const daliamdJpegColorDesc &d) {
  return d.out_format != DALIAMD_JPEG_OUT_RGB && !(d.out_format == DALIAMD_JPEG_OUT_GRAY && d.color != DALIAMD_JPEG_RGB);
}

// ---- full-resolution components (YCbCr 4:4:4, grayscale) to RGB, upright, 8-pixel groups aligned to the planes: nothing
// is interpolated, so the thread's ROWS x (1 or 3) 8-byte loads go out together and the rows follow (the row-by-row code
// below walks its mode tests and one memory round trip per row: the 15 % of a mixed batch that took it cost as much as
// the 85 % on the 4:2:0 fast path).
template <int ROWS>
__device__ __forceinline__ void ColorRowsFull(const daliamdJpegColorDesc &d, int x0, int y0, int rx1, int ry1, int out_x0,
                                              int out_y0) {
  const bool gray = d.color == DALIAMD_JPEG_GRAY;
  u32x2 v[3][ROWS];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    if (c > 0 && gray) break;
    GBytes *plane = (GBytes *)d.plane[c];
    const int pitch = d.pitch[c];
#pragma unroll
    for (int j = 0; j < ROWS; j++) v[c][j] = *reinterpret_cast<GPair *>(plane + (size_t)min(y0 + j, ry1 - 1) * pitch + x0);
  }
  const int npx = min(8, rx1 - x0);
#pragma unroll
  for (int j = 0; j < ROWS; j++) {
    const int y = y0 + j;
    if (y >= ry1) break;
    int yy[8];
#pragma unroll
    for (int i = 0; i < 8; i++) yy[i] = (int)(((i < 4 ? v[0][j].x : v[0][j].y) >> (8 * (i & 3))) & 255);
    uint32_t px[24];
    if (gray) {
#pragma unroll
      for (int i = 0; i < 8; i++) px[3 * i] = px[3 * i + 1] = px[3 * i + 2] = (uint32_t)yy[i];
    } else {
      int up[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 8; i++) up[c][i] = (int)(((i < 4 ? v[1 + c][j].x : v[1 + c][j].y) >> (8 * (i & 3))) & 255);
      YccToRgb8(yy, up, px);
    }
    StoreRgb8((GOutBytes *)d.out + (size_t)(y - out_y0) * d.out_pitch + (size_t)(x0 - out_x0) * 3, px, npx);
  }
}
__host__ __device__ inline bool FastFull(const daliamdJpegColorDesc &d) {
  if (d.out_format != DALIAMD_JPEG_OUT_RGB || d.orientation > 1 || d.color == DALIAMD_JPEG_RGB) return false;
  const int nc = d.color == DALIAMD_JPEG_GRAY ? 1 : 3;
  for (int c = 0; c < nc; c++)
    if (d.h_samp[c] != d.h_samp[0] || d.v_samp[c] != d.v_samp[0] || (d.pitch[c] & 7) != 0 ||
        (reinterpret_cast<uintptr_t>(d.plane[c]) & 7) != 0)
      return false;
  return (d.out_pitch & 7) == 0 && (reinterpret_cast<uintptr_t>(d.out) & 7) == 0 && (d.roi_w <= 0 || (d.roi_x0 & 7) == 0);
}

// does the 4:2:0 fast path's window start where its 8 x 2 pixel groups are aligned to the planes?
__host__ __device__ inline bool AlignedOrigin(const daliamdJpegColorDesc &d) {
  return d.roi_w <= 0 || ((d.roi_x0 & 7) | (d.roi_y0 & 1)) == 0;
}

// which instance of the non-converting kernel takes the sample (see kPath)
__host__ __device__ inline int ColorPath(const daliamdJpegColorDesc &d) {
  if (Fast420(d)) return AlignedOrigin(d) ? 1 : 2;
  return FastFull(d) ? 1 : 0;
}

// kPath: 1 = the fast paths for planes-aligned windows - 4:2:0 (ColorRows420) and full-resolution components
// (ColorRowsFull) - in one instance: what a batch of camera / ImageNet JPEGs takes, in ONE launch, at 100 registers;
// 0 = every other sampling / orientation, row by row (159 registers); 2 = the 4:2:0 fast path for windows that start
// anywhere (ColorRows420Any; region-of-interest decode).  (Before the full-resolution path existed the 4:4:4 / gray samples of
// a mixed batch needed the row-by-row instance: two launches per batch, 2.5 % slower than everything in one kernel.)
template <bool kConvert, int kPath = 0>
__global__ __launch_bounds__(kColorThreads) void JpegColorKernel(const daliamdJpegColorDesc *__restrict__ descs,
                                                                 int ndesc, int total_wg) {
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  int di = FindDesc(descs, ndesc, wg);
  const daliamdJpegColorDesc &d = descs[di];
  if (NeedsConvert(d) != kConvert) return;
  if (!kConvert && ColorPath(d) != kPath) return;
  // region of the (un-rotated) image to produce; the 8-pixel groups are aligned to its origin so that the output
  // rows keep their 8-byte store alignment
  const bool roi = d.roi_w > 0;
  const int rx0 = roi ? d.roi_x0 : 0, ry0 = roi ? d.roi_y0 : 0;
  const int rx1 = roi ? d.roi_x0 + d.roi_w : d.width, ry1 = roi ? d.roi_y0 + d.roi_h : d.height;
  int tiles_x = (rx1 - rx0 + kTileW - 1) / kTileW;
  int t = wg - d.wg_start;
  int ty = t / tiles_x, tx = t - ty * tiles_x;
  int x0 = rx0 + tx * kTileW + (threadIdx.x & 31) * 8;
  const int y_first = ry0 + ty * kTileH + (threadIdx.x >> 5) * kRowsPerThread;
  if (x0 >= rx1 || y_first >= ry1) return;

  const int nstored = d.color == DALIAMD_JPEG_GRAY ? 1 : 3;
  int hmax = 1, vmax = 1;
  for (int c = 0; c < nstored; c++) { hmax = max(hmax, d.h_samp[c]); vmax = max(vmax, d.v_samp[c]); }
  // output format (decoders.image output_type): a gray output of a gray / YCbCr stream is its luma plane alone
  // (libjpeg-turbo's JCS_GRAYSCALE output, which the reference asks nvImageCodec for: image_decoder.h:538-541)
  const int fmt = d.out_format;
  const int oc = fmt == DALIAMD_JPEG_OUT_GRAY ? 1 : 3;
  const bool luma_only = fmt == DALIAMD_JPEG_OUT_GRAY && d.color != DALIAMD_JPEG_RGB;
  const int ncomp = luma_only ? 1 : nstored;
  int mode[3] = {kFull, kFull, kFull};
  for (int c = 0; c < ncomp; c++) mode[c] = ModeOf(d, c, hmax, vmax);
  const int npx = min(8, rx1 - x0);
  const int out_x0 = roi ? d.out_x0 : 0, out_y0 = roi ? d.out_y0 : 0;
  const bool wide_ok = ((d.out_pitch & 7) == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 7) == 0);
  const bool wide_stores = npx == 8 && wide_ok && oc == 3;
  if constexpr (kPath == 2) {
    ColorRows420Any<kRowsPerThread>(d, x0, y_first, rx1, ry1, out_x0, out_y0);
    return;
  } else if constexpr (kPath == 1) {  // wave-uniform: the whole image takes one path
    if (Fast420(d)) ColorRows420<kRowsPerThread>(d, x0, y_first, rx1, ry1, out_x0, out_y0);
    else ColorRowsFull<kRowsPerThread>(d, x0, y_first, rx1, ry1, out_x0, out_y0);
    return;
  }

  for (int row = 0; row < kRowsPerThread; row++) {
    const int y = y_first + row;
    if (y >= ry1) break;
    int s[3][8];
    for (int c = 0; c < ncomp; c++)
      UpsampleRow8((GBytes *)d.plane[c], d.pitch[c], mode[c], hmax / d.h_samp[c], vmax / d.v_samp[c], d.down_w[c], d.down_h[c], x0, y,
                   s[c]);
    uint32_t px[24];
    if (luma_only) {
#pragma unroll
      for (int i = 0; i < 8; i++) px[i] = (uint32_t)s[0][i];
    } else if (d.color == DALIAMD_JPEG_GRAY) {
#pragma unroll
      for (int i = 0; i < 8; i++) px[3 * i] = px[3 * i + 1] = px[3 * i + 2] = (uint32_t)s[0][i];
    } else if (d.color == DALIAMD_JPEG_RGB) {
#pragma unroll
      for (int i = 0; i < 8; i++) { px[3 * i] = s[0][i]; px[3 * i + 1] = s[1][i]; px[3 * i + 2] = s[2][i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int yy = s[0][i], cb = s[1][i] - 128, cr = s[2][i] - 128;
        int r = yy + ((FIXC(1.40200) * cr + ONE_HALF) >> SCALEBITS);
        int g = yy + (((-FIXC(0.34414)) * cb + ONE_HALF + (-FIXC(0.71414)) * cr) >> SCALEBITS);
        int b = yy + ((FIXC(1.77200) * cb + ONE_HALF) >> SCALEBITS);
        px[3 * i] = Clamp8(r); px[3 * i + 1] = Clamp8(g); px[3 * i + 2] = Clamp8(b);
      }
    }
    // RGB -> the requested format (ConvertCPU / ConvertGPU of the reference, operators/imgcodec/util/convert.h:140-192:
    // BGR = swap; YCbCr = ITU-R BT.601 with head room, float, ConvertSat; gray = 0.299 R + 0.587 G + 0.114 B)
    if (kConvert) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint32_t r8 = px[3 * i], g8 = px[3 * i + 1], b8 = px[3 * i + 2];
        const float r = (float)r8, g = (float)g8, b = (float)b8;
        if (fmt == DALIAMD_JPEG_OUT_BGR) {
          px[3 * i] = b8; px[3 * i + 2] = r8;
        } else if (fmt == DALIAMD_JPEG_OUT_YCBCR) {
          px[3 * i] = SatRound8(0.25678823529f * r + 0.50412941176f * g + 0.09790588235f * b + 16.0f);
          px[3 * i + 1] = SatRound8(-0.14822289945f * r + -0.29099278682f * g + 0.43921568627f * b + 128.0f);
          px[3 * i + 2] = SatRound8(0.43921568627f * r + -0.36778831435f * g + -0.07142737192f * b + 128.0f);
        } else {  // gray from a stream stored as RGB
          px[i] = SatRound8(0.299f * r + 0.587f * g + 0.114f * b);
        }
      }
    }
    if (d.orientation > 1) {
      // undo the EXIF orientation: source pixel (y, x) lands at (oy, ox) of the upright image
      const int W = d.width, H = d.height;
      for (int i = 0; i < npx; i++) {
        int x = x0 + i, oy, ox;
        switch (d.orientation) {
          case 2: oy = y; ox = W - 1 - x; break;
          case 3: oy = H - 1 - y; ox = W - 1 - x; break;
          case 4: oy = H - 1 - y; ox = x; break;
          case 5: oy = x; ox = y; break;
          case 6: oy = x; ox = H - 1 - y; break;
          case 7: oy = W - 1 - x; ox = H - 1 - y; break;
          default: oy = W - 1 - x; ox = y; break;  // 8
        }
        GOutBytes *p = (GOutBytes *)d.out + (size_t)(oy - out_y0) * d.out_pitch + (size_t)(ox - out_x0) * oc;
        if (oc == 1) { p[0] = (uint8_t)px[i]; }
        else { p[0] = (uint8_t)px[3 * i]; p[1] = (uint8_t)px[3 * i + 1]; p[2] = (uint8_t)px[3 * i + 2]; }
      }
      continue;
    }
    GOutBytes *o = (GOutBytes *)d.out + (size_t)(y - out_y0) * d.out_pitch + (size_t)(x0 - out_x0) * oc;
    if (wide_stores) {
      uint32_t w[6];
#pragma unroll
      for (int j = 0; j < 6; j++)
        w[j] = px[4 * j] | (px[4 * j + 1] << 8) | (px[4 * j + 2] << 16) | (px[4 * j + 3] << 24);
      GOutPair *o2 = reinterpret_cast<GOutPair *>(o);
      o2[0] = u32x2{w[0], w[1]};
      o2[1] = u32x2{w[2], w[3]};
      o2[2] = u32x2{w[4], w[5]};
    } else {
      for (int i = 0; i < npx * oc; i++) o[i] = (uint8_t)px[i];
    }
  }
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdJpegColorSetup(daliamdJpegColorDesc *descs, int n, int *num_workgroups, int *kernel_mask) {
  DALIAMD_REQUIRE(descs && num_workgroups && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegColorSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    auto &d = descs[i];
    DALIAMD_REQUIRE(d.width > 0 && d.height > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegColorSetup: desc %d has empty image", i);
    DALIAMD_REQUIRE(d.orientation >= 0 && d.orientation <= 8, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegColorSetup: desc %d: invalid EXIF orientation %d", i, d.orientation);
    DALIAMD_REQUIRE(d.roi_w >= 0 && d.roi_h >= 0 && (d.roi_w == 0 || (d.roi_h > 0 && d.roi_x0 >= 0 && d.roi_y0 >= 0 &&
                                                                       d.roi_x0 + d.roi_w <= d.width &&
                                                                       d.roi_y0 + d.roi_h <= d.height)),
                    DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegColorSetup: desc %d: region of interest out of bounds", i);
    DALIAMD_REQUIRE(d.roi_w > 0 || (d.out_x0 == 0 && d.out_y0 == 0), DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegColorSetup: desc %d: output origin without a region of interest", i);
    const int pw = d.roi_w > 0 ? d.roi_w : d.width, ph = d.roi_w > 0 ? d.roi_h : d.height;  // produced source pixels
    DALIAMD_REQUIRE(d.out_format >= DALIAMD_JPEG_OUT_RGB && d.out_format <= DALIAMD_JPEG_OUT_YCBCR,
                    DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegColorSetup: desc %d: invalid output format %d", i, d.out_format);
    const int oc = d.out_format == DALIAMD_JPEG_OUT_GRAY ? 1 : 3;
    DALIAMD_REQUIRE(d.out_pitch >= oc * (d.orientation >= 5 ? ph : pw), DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegColorSetup: desc %d out_pitch %d < %d*width", i, d.out_pitch, oc);
    int ncomp = d.color == DALIAMD_JPEG_GRAY ? 1 : 3;
    int hmax = 1, vmax = 1;
    for (int c = 0; c < ncomp; c++) {
      DALIAMD_REQUIRE(d.h_samp[c] >= 1 && d.h_samp[c] <= 4 && d.v_samp[c] >= 1 && d.v_samp[c] <= 4,
                      DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegColorSetup: bad sampling factors");
      hmax = hmax > d.h_samp[c] ? hmax : d.h_samp[c];
      vmax = vmax > d.v_samp[c] ? vmax : d.v_samp[c];
    }
    for (int c = 0; c < ncomp; c++)
      DALIAMD_REQUIRE(hmax % d.h_samp[c] == 0 && vmax % d.v_samp[c] == 0, DALIAMD_ERROR_UNSUPPORTED,
                      "daliamdJpegColorSetup: fractional chroma sampling ratios are not supported");
    d.wg_start = wg;
    wg += ((pw + daliamd::kTileW - 1) / daliamd::kTileW) * ((ph + daliamd::kTileH - 1) / daliamd::kTileH);
  }
  *num_workgroups = wg;
  if (kernel_mask) {
    int mask = 0;
    for (int i = 0; i < n; i++)
      mask |= daliamd::NeedsConvert(descs[i]) ? 4 : daliamd::ColorPath(descs[i]) == 1 ? 1 : daliamd::ColorPath(descs[i]) == 2 ? 8 : 2;
    *kernel_mask = mask;
  }
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegPlanRoi(int width, int height, int num_components, const int32_t *h_samp,
                                   const int32_t *v_samp, int orientation, int up_y0, int up_x0, int up_h, int up_w,
                                   daliamdJpegRoiPlan *plan) {
  DALIAMD_REQUIRE(plan && h_samp && v_samp && width > 0 && height > 0 && (num_components == 1 || num_components == 3),
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegPlanRoi: invalid argument");
  DALIAMD_REQUIRE(orientation >= 0 && orientation <= 8, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegPlanRoi: invalid EXIF orientation %d", orientation);
  const int W = width, H = height;
  const int UW = orientation >= 5 ? H : W, UH = orientation >= 5 ? W : H;  // upright image
  DALIAMD_REQUIRE(up_h > 0 && up_w > 0 && up_y0 >= 0 && up_x0 >= 0 && up_y0 + up_h <= UH && up_x0 + up_w <= UW,
                  DALIAMD_ERROR_OUT_OF_RANGE, "daliamdJpegPlanRoi: window [%d:%d, %d:%d] does not fit a %dx%d image", up_y0,
                  up_y0 + up_h, up_x0, up_x0 + up_w, UH, UW);
  // inverse of the orientation mapping of JpegColorKernel: upright window -> source window
  int sx0, sy0, sw, sh;
  switch (orientation) {
    case 2: sx0 = W - (up_x0 + up_w); sy0 = up_y0; sw = up_w; sh = up_h; break;
    case 3: sx0 = W - (up_x0 + up_w); sy0 = H - (up_y0 + up_h); sw = up_w; sh = up_h; break;
    case 4: sx0 = up_x0; sy0 = H - (up_y0 + up_h); sw = up_w; sh = up_h; break;
    case 5: sx0 = up_y0; sw = up_h; sy0 = up_x0; sh = up_w; break;
    case 6: sx0 = up_y0; sw = up_h; sy0 = H - (up_x0 + up_w); sh = up_w; break;
    case 7: sx0 = W - (up_y0 + up_h); sw = up_h; sy0 = H - (up_x0 + up_w); sh = up_w; break;
    case 8: sx0 = W - (up_y0 + up_h); sw = up_h; sy0 = up_x0; sh = up_w; break;
    default: sx0 = up_x0; sy0 = up_y0; sw = up_w; sh = up_h; break;
  }
  plan->roi_x0 = sx0; plan->roi_y0 = sy0; plan->roi_w = sw; plan->roi_h = sh;
  plan->out_x0 = up_x0; plan->out_y0 = up_y0;
  int hmax = 1, vmax = 1;
  for (int c = 0; c < num_components; c++) {
    DALIAMD_REQUIRE(h_samp[c] >= 1 && h_samp[c] <= 4 && v_samp[c] >= 1 && v_samp[c] <= 4, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegPlanRoi: bad sampling factors");
    hmax = hmax > h_samp[c] ? hmax : h_samp[c];
    vmax = vmax > v_samp[c] ? vmax : v_samp[c];
  }
  for (int c = 0; c < 3; c++) plan->rect[c][0] = plan->rect[c][1] = plan->rect[c][2] = plan->rect[c][3] = 0;
  for (int c = 0; c < num_components; c++) {
    // samples of this component the colour kernel reads for the window: the co-sited ones plus one neighbour on
    // each side where it interpolates (fancy upsampling); conservative for the box modes
    const int hf = hmax / h_samp[c], vf = vmax / v_samp[c];
    const int dw = (W * h_samp[c] + hmax - 1) / hmax, dh = (H * v_samp[c] + vmax - 1) / vmax;
    int lo_x = sx0 / hf, hi_x = (sx0 + sw - 1) / hf, lo_y = sy0 / vf, hi_y = (sy0 + sh - 1) / vf;
    if (hf > 1) { lo_x -= 1; hi_x += 1; }
    if (vf > 1) { lo_y -= 1; hi_y += 1; }
    lo_x = lo_x < 0 ? 0 : lo_x; lo_y = lo_y < 0 ? 0 : lo_y;
    hi_x = hi_x > dw - 1 ? dw - 1 : hi_x; hi_y = hi_y > dh - 1 ? dh - 1 : hi_y;
    plan->rect[c][0] = lo_x >> 3; plan->rect[c][1] = lo_y >> 3;
    plan->rect[c][2] = (hi_x >> 3) + 1; plan->rect[c][3] = (hi_y >> 3) + 1;
  }
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegColorRun(daliamdStream_t stream, const daliamdJpegColorDesc *descs_dev, int n,
                                    int num_workgroups, int kernel_mask) {
  if (n == 0 || num_workgroups == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_workgroups > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegColorRun: invalid argument");
  if (kernel_mask & 1)
    {
      daliamd::KernelTimer timer("JpegColorKernel", (hipStream_t)stream);
      hipLaunchKernelGGL((daliamd::JpegColorKernel<false, 1>), dim3(daliamd::XcdGrid(num_workgroups)),
                         dim3(daliamd::kColorThreads), 0, (hipStream_t)stream, descs_dev, n, num_workgroups);
    }
  if (kernel_mask & 2)
    {
      daliamd::KernelTimer timer("JpegColorKernel", (hipStream_t)stream);
      hipLaunchKernelGGL((daliamd::JpegColorKernel<false, 0>), dim3(daliamd::XcdGrid(num_workgroups)),
                         dim3(daliamd::kColorThreads), 0, (hipStream_t)stream, descs_dev, n, num_workgroups);
    }
  if (kernel_mask & 8)
    {
      daliamd::KernelTimer timer("JpegColorKernel", (hipStream_t)stream);
      hipLaunchKernelGGL((daliamd::JpegColorKernel<false, 2>), dim3(daliamd::XcdGrid(num_workgroups)),
                         dim3(daliamd::kColorThreads), 0, (hipStream_t)stream, descs_dev, n, num_workgroups);
    }
  if (kernel_mask & 4)
    {
      daliamd::KernelTimer timer("JpegColorKernel", (hipStream_t)stream);
      hipLaunchKernelGGL(daliamd::JpegColorKernel<true>, dim3(daliamd::XcdGrid(num_workgroups)),
                         dim3(daliamd::kColorThreads), 0, (hipStream_t)stream, descs_dev, n, num_workgroups);
    }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
