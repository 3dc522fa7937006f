// Arithmetic shared by the colour stage (jpeg_color.hip) and the entropy decoder's fused colour output
// (jpeg_huffman.hip): libjpeg-turbo jdsample.c "fancy" triangle upsampling and the jdcolor.c 16-bit fixed-point
// BT.601 full-range conversion.  Integer => bit-exact whichever kernel runs it.
#ifndef DALI_AMD_CSRC_JPEG_COLOR_MATH_H_
#define DALI_AMD_CSRC_JPEG_COLOR_MATH_H_
#include "common.h"

namespace daliamd {

// Explicit global address space: the pointers come out of a descriptor in memory, and the generic ("flat") accesses
// the compiler would otherwise emit are slower and tie up the LDS counter as well.
using GBytes = const uint8_t __attribute__((address_space(1)));
using GWords = const uint32_t __attribute__((address_space(1)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
using GPair = const u32x2 __attribute__((address_space(1)));
using GOutBytes = uint8_t __attribute__((address_space(1)));
using GOutPair = u32x2 __attribute__((address_space(1)));

#define SCALEBITS 16
#define ONE_HALF (1 << (SCALEBITS - 1))
#define FIXC(x) ((int32_t)((x) * (1L << SCALEBITS) + 0.5))

__device__ __forceinline__ uint32_t Clamp8(int v) { return (uint32_t)min(max(v, 0), 255); }
// ConvertSat<uint8_t>(float): clamp, round half away from zero
__device__ __forceinline__ uint32_t SatRound8(float v) { return !(v > 0.0f) ? 0u : v >= 255.0f ? 255u : (uint32_t)(v + 0.5f); }

enum UpsampleMode { kFull = 0, kH2V1 = 1, kH2V2 = 2, kH1V2 = 3, kBox = 4 };

__device__ __forceinline__ int ClampI(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Triangle filter along x for the pixels x0..x0+7 from the (already vertically combined) samples around them.
// sv[i] = sample (x0 >> 1) - 1 + i (clamped), i = 0..6; SHIFT/BIAS as in jdsample.c (h2v1: 2 / 1,2; h2v2: 4 / 8,7).
template <int SHIFT, int BIAS_EVEN, int BIAS_ODD>
__device__ __forceinline__ void TriangleX8(const int sv[7], bool odd, int out[8]) {
  if (!odd) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      out[2 * i] = (sv[1 + i] * 3 + sv[i] + BIAS_EVEN) >> SHIFT;
      out[2 * i + 1] = (sv[1 + i] * 3 + sv[2 + i] + BIAS_ODD) >> SHIFT;
    }
  } else {  // x0 odd: the first pixel is the odd half of sample (x0 >> 1)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      out[2 * i] = (sv[1 + i] * 3 + sv[2 + i] + BIAS_ODD) >> SHIFT;
      out[2 * i + 1] = (sv[2 + i] * 3 + sv[1 + i] + BIAS_EVEN) >> SHIFT;
    }
  }
}

// Samples k0-1 .. k0+5 of one row with three dword loads instead of seven byte loads (k0 and the row start are
// multiples of 4; rows are padded to 8-sample blocks, so the dword behind k0 is inside the row unless k0 is its last
// dword - then the samples it would hold are beyond the component anyway and ClampRight7 replaces them).
// The left neighbour of sample 0 is sample 0.
__device__ __forceinline__ void LoadSamples7(GBytes *__restrict__ row, int pitch, int k0, int s[7]) {
  const uint32_t a = *reinterpret_cast<GWords *>(row + max(k0 - 4, 0));
  const uint32_t b = *reinterpret_cast<GWords *>(row + k0);
  const uint32_t c = *reinterpret_cast<GWords *>(row + min(k0 + 4, pitch - 4));
  s[0] = k0 ? (int)(a >> 24) : (int)(b & 255);
  s[1] = (int)(b & 255); s[2] = (int)((b >> 8) & 255); s[3] = (int)((b >> 16) & 255); s[4] = (int)(b >> 24);
  s[5] = (int)(c & 255); s[6] = (int)((c >> 8) & 255);
}
// Neighbour indices are clamped to the last sample dw-1 (>= k0): every later entry repeats its predecessor.
__device__ __forceinline__ void ClampRight7(int s[7], int k0, int dw) {
  if (k0 + 5 > dw - 1) {
#pragma unroll
    for (int i = 2; i < 7; i++) s[i] = k0 - 1 + i > dw - 1 ? s[i - 1] : s[i];
  }
}

__device__ __forceinline__ void Chroma7(uint32_t a, uint32_t b, uint32_t c, bool first, int s[7]) {
  s[0] = first ? (int)(b & 255) : (int)(a >> 24);
  s[1] = (int)(b & 255); s[2] = (int)((b >> 8) & 255); s[3] = (int)((b >> 16) & 255); s[4] = (int)(b >> 24);
  s[5] = (int)(c & 255); s[6] = (int)((c >> 8) & 255);
}

// YCbCr -> RGB of 8 pixels (jdcolor.c ycc_rgb_convert): yy[i] luma, up[0][i] / up[1][i] the upsampled Cb / Cr.
__device__ __forceinline__ void YccToRgb8(const int yy[8], const int up[2][8], uint32_t px[24]) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int u = up[0][i] - 128, v = up[1][i] - 128;
    const int rr = yy[i] + ((FIXC(1.40200) * v + ONE_HALF) >> SCALEBITS);
    const int gg = yy[i] + (((-FIXC(0.34414)) * u + ONE_HALF + (-FIXC(0.71414)) * v) >> SCALEBITS);
    const int bb = yy[i] + ((FIXC(1.77200) * u + ONE_HALF) >> SCALEBITS);
    px[3 * i] = Clamp8(rr); px[3 * i + 1] = Clamp8(gg); px[3 * i + 2] = Clamp8(bb);
  }
}
// 8 RGB pixels (or the first npx of them) to o, which is 8-byte aligned when npx == 8
__device__ __forceinline__ void StoreRgb8(GOutBytes *o, const uint32_t px[24], int npx) {
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

}  // namespace daliamd
#endif  // DALI_AMD_CSRC_JPEG_COLOR_MATH_H_
