// Host kernels of the heavy-augmentation operators (configs[2]) for the CPU backend: warp_affine, gaussian_blur, colour
// twist / erase.  Product code, one sample per call, driven by the same descriptors as the device kernels
// (include/dali_amd_kernels.h); the arithmetic of the reference's CPU kernels operation by operation:
//   warp    dali/kernels/imgproc/warp_cpu.h:143-178 (source coordinates advanced incrementally, re-anchored every
//           256 pixels), sampler.h:60-175 (nearest, border), :258-338 (bilinear: s0 + (s1 - s0) * qy)
//   blur    dali/kernels/imgproc/convolution/convolution_cpu.h:241-340, separable_convolution_cpu.h:91-110 (W pass
//           then H pass, float intermediate, reflect-101 border, taps accumulated in order)
//   twist   dali/kernels/imgproc/pointwise/linear_transformation_cpu.h:57-77 (M * px + offset, ConvertSat)
//   erase   dali/kernels/erase/erase_cpu.h (copy + fill of the clipped regions)
// Built with -ffp-contract=off like the rest of the host library: multiply and add are rounded separately.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "dali_amd_host.h"
#include "host_common.h"

using daliamd_host::Fail;

namespace {

// ConvertSat<uint8_t>(float) = clamp(roundf(v)), halves away from zero - without the library call: everything that
// rounds to <= 0 (NaN included) gives 0, everything from 255.5 on gives 255, and in between the truncation is the floor
// and the fraction v - floor(v) is exact.
inline uint8_t SatU8(float v) {
  if (!(v > 0)) return 0;
  if (v >= 255.5f) return 255;
  const int i = (int)v;
  return (uint8_t)(i + ((v - (float)i) >= 0.5f ? 1 : 0));
}
// the same function without branches, for loops the compiler turns into vector code: NaN and everything <= 0 become 0,
// everything from 255.5 on becomes exactly 255
inline uint8_t SatU8Flat(float v) {
  float c = v > 0.0f ? v : 0.0f;
  c = c < 255.5f ? c : 255.0f;
  const int i = (int)c;
  return (uint8_t)(i + ((c - (float)i) >= 0.5f));
}
// (int)std::floor(f) without the library call (baseline x86-64 has no rounding instruction); far outside the range of
// image coordinates the library call decides
inline int FloorI(float f) {
  if (!(f > -1e9f && f < 1e9f)) return (int)std::floor(f);
  const int i = (int)f;
  return i - ((float)i > f);
}

// whole-row loops, one clone per instruction set: the element arithmetic is the scalar code's
// dst[e] = sum over k of rows[k][e] * w[k], taps in order, product then sum - sixteen elements at a time with the
// running sums in registers (every element still sees acc = acc + sample * weight for k = 0, 1, ...)
__attribute__((target_clones("avx2", "default")))
void TapSum(float *dst, const float *const *rows, const float *w, int taps, size_t n) {
  size_t e = 0;
  for (; e + 16 <= n; e += 16) {
    float a[16];
    for (int j = 0; j < 16; j++) a[j] = 0.0f;
    for (int k = 0; k < taps; k++) {
      const float *r = rows[k] + e;
      const float wk = w[k];
      for (int j = 0; j < 16; j++) a[j] += r[j] * wk;
    }
    for (int j = 0; j < 16; j++) dst[e + j] = a[j];
  }
  for (; e < n; e++) {
    float a = 0.0f;
    for (int k = 0; k < taps; k++) a += rows[k][e] * w[k];
    dst[e] = a;
  }
}
__attribute__((target_clones("avx2", "default")))
void RowU8ToFloat(float *dst, const uint8_t *src, size_t n) {
  for (size_t e = 0; e < n; e++) dst[e] = (float)src[e];
}
__attribute__((target_clones("avx2", "default")))
void RowFloatToU8(uint8_t *dst, const float *src, size_t n) {
  for (size_t e = 0; e < n; e++) dst[e] = SatU8Flat(src[e]);
}
__attribute__((target_clones("avx2", "default")))
void TwistRow(uint8_t *dst, const uint8_t *src, int w, const float *m, const float *off) {
  for (int x = 0; x < w; x++) {
    const float v0 = src[3 * x], v1 = src[3 * x + 1], v2 = src[3 * x + 2];
    float s0 = m[0] * v0; s0 += m[1] * v1; s0 += m[2] * v2;
    float s1 = m[3] * v0; s1 += m[4] * v1; s1 += m[5] * v2;
    float s2 = m[6] * v0; s2 += m[7] * v1; s2 += m[8] * v2;
    dst[3 * x] = SatU8Flat(s0 + off[0]);
    dst[3 * x + 1] = SatU8Flat(s1 + off[1]);
    dst[3 * x + 2] = SatU8Flat(s2 + off[2]);
  }
}

inline int Reflect101(int idx, int size) {
  if (size < 2) return size - 1;
  for (;;) {
    if (idx < 0) idx = -idx;
    else if (idx >= size) idx = 2 * size - 2 - idx;
    else break;
  }
  return idx;
}

}  // namespace

extern "C" int daliamdWarpAffineHost(const daliamdWarpAffineDesc *d) {
  if (!d || !d->in || !d->out || d->in_h <= 0 || d->in_w <= 0 || d->channels < 1 || d->channels > 4)
    return Fail("daliamdWarpAffineHost: invalid descriptor");
  if (d->interp != DALIAMD_INTERP_NN && d->interp != DALIAMD_INTERP_LINEAR) return Fail("daliamdWarpAffineHost: unsupported interpolation");
  const int H = d->in_h, W = d->in_w, C = d->channels;
  const float *m = d->matrix;
  float border[4];
  for (int c = 0; c < 4; c++) border[c] = (float)SatU8(d->fill[c]);   // ConvertSat<In>(border value)
  auto fetch = [&](int x, int y, int c) -> float {
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) return d->in[(size_t)y * d->in_pitch + (size_t)x * C + c];
    if (!d->border_clamp) return border[c];
    x = x < 0 ? 0 : x > W - 1 ? W - 1 : x;
    y = y < 0 ? 0 : y > H - 1 ? H - 1 : y;
    return d->in[(size_t)y * d->in_pitch + (size_t)x * C + c];
  };
  const float dsdx_x = m[0], dsdx_y = m[3];
  const int tile_w = 256;
  const float dtx = tile_w * dsdx_x, dty = tile_w * dsdx_y;
  for (int y = 0; y < d->out_h; y++) {
    const float vx = 0 + 0.5f, vy = y + 0.5f;   // map_coords: the affine map of the pixel centre
    float tx = m[2]; tx += m[0] * vx; tx += m[1] * vy;
    float ty = m[5]; ty += m[3] * vx; ty += m[4] * vy;
    for (int x_tile = 0; x_tile < d->out_w; x_tile += tile_w, tx += dtx, ty += dty) {
      const int x_end = x_tile + tile_w < d->out_w ? x_tile + tile_w : d->out_w;
      float sx = tx, sy = ty;
      for (int x = x_tile; x < x_end; x++, sx += dsdx_x, sy += dsdx_y) {
        uint8_t *o = d->out + (size_t)y * d->out_pitch + (size_t)x * C;
        if (d->interp == DALIAMD_INTERP_NN) {
          const int ix = FloorI(sx), iy = FloorI(sy);
          if ((unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) {
            const uint8_t *p = d->in + (size_t)iy * d->in_pitch + (size_t)ix * C;
            for (int c = 0; c < C; c++) o[c] = p[c];
          } else {
            for (int c = 0; c < C; c++) o[c] = (uint8_t)fetch(ix, iy, c);
          }
        } else {
          const float fx = sx - 0.5f, fy = sy - 0.5f;
          const int x0 = FloorI(fx), y0 = FloorI(fy);
          const float qx = fx - x0, px = 1 - qx, qy = fy - y0;
          if (x0 >= 0 && y0 >= 0 && x0 + 1 < W && y0 + 1 < H) {   // all four taps inside: no border logic per tap
            const uint8_t *p0 = d->in + (size_t)y0 * d->in_pitch + (size_t)x0 * C, *p1 = p0 + d->in_pitch;
            for (int c = 0; c < C; c++) {
              const float s00 = p0[c], s01 = p0[C + c], s10 = p1[c], s11 = p1[C + c];
              const float s0 = s00 * px + s01 * qx;
              const float s1 = s10 * px + s11 * qx;
              o[c] = SatU8(s0 + (s1 - s0) * qy);
            }
          } else {
            for (int c = 0; c < C; c++) {
              const float s00 = fetch(x0, y0, c), s01 = fetch(x0 + 1, y0, c), s10 = fetch(x0, y0 + 1, c), s11 = fetch(x0 + 1, y0 + 1, c);
              const float s0 = s00 * px + s01 * qx;
              const float s1 = s10 * px + s11 * qx;
              o[c] = SatU8(s0 + (s1 - s0) * qy);
            }
          }
        }
      }
    }
  }
  return 0;
}

extern "C" int daliamdGaussianBlurHost(const daliamdGaussianBlurDesc *d) {
  if (!d || !d->in || !d->out || d->h <= 0 || d->w <= 0 || d->channels < 1) return Fail("daliamdGaussianBlurHost: invalid descriptor");
  if (d->size_x < 1 || d->size_y < 1 || d->size_x > DALIAMD_MAX_BLUR_WINDOW || d->size_y > DALIAMD_MAX_BLUR_WINDOW ||
      !(d->size_x & 1) || !(d->size_y & 1))
    return Fail("daliamdGaussianBlurHost: window sizes must be odd and at most %d", DALIAMD_MAX_BLUR_WINDOW);
  const int H = d->h, W = d->w, C = d->channels, rx = (d->size_x - 1) / 2, ry = (d->size_y - 1) / 2;
  // Both passes: sixteen elements of a row at a time, their taps in order (acc = acc + sample * weight, multiply and
  // add rounded separately) with the sums in registers.
  const size_t rowlen = (size_t)W * C;
  std::vector<float> tmp((size_t)H * rowlen), pad((size_t)(W + 2 * rx) * C), acc(rowlen);
  std::vector<const float *> taps(DALIAMD_MAX_BLUR_WINDOW);
  for (int y = 0; y < H; y++) {
    const uint8_t *row = d->in + (size_t)y * d->in_pitch;
    // the row with its reflected borders, as floats
    RowU8ToFloat(pad.data() + (size_t)rx * C, row, rowlen);
    for (int side = 0; side < 2; side++)
      for (int x = side ? W : -rx; x < (side ? W + rx : 0); x++) {
        const uint8_t *src = row + (size_t)Reflect101(x, W) * C;
        float *dst = pad.data() + (size_t)(x + rx) * C;
        for (int c = 0; c < C; c++) dst[c] = (float)src[c];
      }
    for (int k = 0; k < d->size_x; k++) taps[k] = pad.data() + (size_t)k * C;
    TapSum(tmp.data() + (size_t)y * rowlen, taps.data(), d->window_x, d->size_x, rowlen);
  }
  for (int y = 0; y < H; y++) {
    for (int k = 0; k < d->size_y; k++) taps[k] = tmp.data() + (size_t)Reflect101(y - ry + k, H) * rowlen;
    TapSum(acc.data(), taps.data(), d->window_y, d->size_y, rowlen);
    RowFloatToU8(d->out + (size_t)y * d->out_pitch, acc.data(), rowlen);
  }
  return 0;
}

extern "C" int daliamdPointwiseHost(const daliamdPointwiseDesc *d) {
  if (!d || !d->in || !d->out || d->h <= 0 || d->w <= 0 || d->channels < 1 || d->channels > 4)
    return Fail("daliamdPointwiseHost: invalid descriptor");
  if (d->transform && d->channels != 3) return Fail("daliamdPointwiseHost: the colour transformation needs 3 channels");
  if (d->num_regions < 0 || d->num_regions > DALIAMD_MAX_ERASE_REGIONS) return Fail("daliamdPointwiseHost: too many regions");
  const int C = d->channels;
  for (int y = 0; y < d->h; y++) {
    const uint8_t *src = d->in + (size_t)y * d->in_pitch;
    uint8_t *dst = d->out + (size_t)y * d->out_pitch;
    if (!d->transform) {
      if (dst != src) std::memmove(dst, src, (size_t)d->w * C);
      continue;
    }
    TwistRow(dst, src, d->w, d->matrix, d->offset);
  }
  uint8_t fill[4];
  for (int c = 0; c < 4; c++) fill[c] = SatU8(d->fill[c]);
  for (int r = 0; r < d->num_regions; r++) {
    const int y0 = d->region[r][0] < 0 ? 0 : d->region[r][0], x0 = d->region[r][1] < 0 ? 0 : d->region[r][1];
    const int y1 = d->region[r][2] > d->h ? d->h : d->region[r][2], x1 = d->region[r][3] > d->w ? d->w : d->region[r][3];
    for (int y = y0; y < y1; y++) {
      uint8_t *dst = d->out + (size_t)y * d->out_pitch;
      for (int x = x0; x < x1; x++)
        for (int c = 0; c < C; c++) dst[(size_t)x * C + c] = fill[c];
    }
  }
  return 0;
}
