/*
 * ORACLE (test infrastructure only -- never linked or imported by the product path).
 *
 * CPU restatement of the reference's CropMirrorNormalize (CPU backend):
 *
 *   normalisation args (double -> float)     dali/operators/image/crop/crop_mirror_normalize.h:120-149
 *   slice/flip/permute/pad arg mapping       crop_mirror_normalize.h:43-90,
 *                                            dali/kernels/slice/slice_flip_normalize_permute_pad_common.h:28-140
 *   inner arithmetic                         dali/kernels/slice/slice_flip_normalize_permute_pad_cpu.h:37-64
 *                                            out = ConvertSat<Out>((float(in) - mean) * inv_std)
 *   fp16 store                               include/dali/util/half.hpp:231-243,464-536
 *                                            (round to nearest, ties AWAY from zero)
 *   float -> integer store                   include/dali/core/convert.h:306-321 (std::round + clamp)
 *
 * Pinning: the numpy formula used by the reference's own python test
 * (dali/test/python/operator_1/test_crop_mirror_normalize.py:255-286) and the sequential-data
 * naive loop of dali/kernels/slice/slice_flip_normalize_permute_pad_kernel_test.h:40-131
 * are restated in tests/test_oracle_cmn.py.
 * Must be compiled with -ffp-contract=off (sub then mul, never fma).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* half_float::detail::float2half_impl<round_to_nearest>, HALF_ROUND_TIES_TO_EVEN == 0.
 * base_table/shift_table (half.hpp:471-522) written as the arithmetic that generated them. */
uint16_t orc_float2half(float value) {
  uint32_t bits;
  memcpy(&bits, &value, 4);
  uint32_t idx = bits >> 23;        /* sign + exponent, 0..511 */
  uint32_t e = idx & 0xff;
  uint16_t sign = (uint16_t)((idx & 0x100) << 7);
  uint16_t base;
  int shift;
  if (e < 103) { base = 0; shift = 24; }
  else if (e < 113) { base = (uint16_t)(0x0400 >> (113 - e)); shift = 126 - (int)e; }
  else if (e < 143) { base = (uint16_t)((e - 112) << 10); shift = 13; }
  else if (e < 255) { base = 0x7C00; shift = 24; }
  else { base = 0x7C00; shift = 13; }
  base |= sign;
  uint32_t mant = bits & 0x7FFFFF;
  uint16_t hbits = (uint16_t)(base + (uint16_t)(mant >> shift));
  hbits = (uint16_t)(hbits + (((mant >> (shift - 1)) | (e == 102)) & ((hbits & 0x7C00) != 0x7C00)));
  return hbits;
}

void orc_float2half_array(const float *in, uint16_t *out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = orc_float2half(in[i]);
}

/* ProcessNormArgs, crop_mirror_normalize.h:120-149.  mean/std come in as float vectors
 * (ArgValue<float>), scale/shift as float; evaluated in double, stored as float.
 * Returns the number of normalisation entries (0 when normalisation is the identity). */
int orc_cmn_norm_args(const float *mean, int nmean, const float *stdv, int nstd, float scale,
                      float shift, float *mean_out, float *inv_std_out) {
  int n = nmean > nstd ? nmean : nstd;
  int all_identity = 1;
  for (int d = 0; d < n; d++) {
    double mean_val = mean[d % nmean];
    double std_val = stdv[d % nstd];
    mean_out[d] = (float)fma(-(double)shift, std_val / (double)scale, mean_val);
    inv_std_out[d] = (float)((double)scale / std_val);
    if (!(mean_out[d] == 0.0f) || !(inv_std_out[d] == 1.0f)) all_identity = 0;
  }
  return all_identity ? 0 : n;
}

static uint8_t sat_u8(float v) {
  float r = roundf(v);
  if (!(r > 0)) return 0;
  if (r > 255) return 255;
  return (uint8_t)r;
}
static int8_t sat_i8(float v) {
  float r = roundf(v);
  if (r < -128) return -128;
  if (r > 127) return 127;
  if (r != r) return 0;
  return (int8_t)r;
}

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_U8 = 2, ORC_I8 = 3 };

static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

/*
 * in: u8 HWC [H][W][C].  Crop window anchor (ay, ax), shape (ch, cw); window may extend
 * out of bounds only if pad_oob != 0 (out_of_bounds_policy="pad"), where fill_values apply.
 * mirror: reverse W inside the crop.  layout_chw: 1 => CHW output, 0 => HWC.
 * pad_output: channels padded to next pow2 (mean = inv_std = 0 => padding elements are 0
 * after normalisation; with no normalisation they take fill_values).
 * nnorm: 0 (no normalisation), 1 (scalar) or C.
 * Output element type per `dtype`.  Returns 0 on success, 1 on an out-of-bounds window with
 * policy "error".
 */
int orc_cmn_u8(const uint8_t *in, int H, int W, int C, int ay, int ax, int ch, int cw, int mirror,
               const float *mean, const float *inv_std, int nnorm, int layout_chw, int pad_output,
               int pad_oob, const float *fill_values, int nfill, int dtype, void *out) {
  int oob = ay < 0 || ax < 0 || ay + ch > H || ax + cw > W;
  if (oob && !pad_oob) return 1;
  int Cout = pad_output ? next_pow2(C) : C;
  for (int y = 0; y < ch; y++) {
    for (int x = 0; x < cw; x++) {
      int sy = ay + y;
      int sx = mirror ? ax + (cw - 1 - x) : ax + x;
      int inside = sy >= 0 && sy < H && sx >= 0 && sx < W;
      for (int c = 0; c < Cout; c++) {
        int64_t o = layout_chw ? ((int64_t)c * ch + y) * cw + x : ((int64_t)y * cw + x) * Cout + c;
        /* fill value per channel: ProcessArgs, slice_flip_normalize_permute_pad_common.h:118-127 */
        float fill = nfill == 0 ? 0.0f : nfill == 1 ? fill_values[0] : (c < nfill ? fill_values[c] : 0.0f);
        float v;
        if (c >= C || !inside) {
          /* every out-of-bounds element (spatial or padded channel) is written with the fill
           * value, un-normalised: slice_flip_normalize_permute_pad_cpu.h:100-145 */
          v = fill;
        } else {
          float e = (float)in[((int64_t)sy * W + sx) * C + c];
          if (nnorm) {
            float m = mean[nnorm > 1 ? c : 0], s = inv_std[nnorm > 1 ? c : 0];
            v = (e - m) * s;
          } else {
            v = e;
          }
        }
        switch (dtype) {
          case ORC_F32: ((float *)out)[o] = v; break;
          case ORC_F16: ((uint16_t *)out)[o] = orc_float2half(v); break;
          case ORC_U8: ((uint8_t *)out)[o] = sat_u8(v); break;
          case ORC_I8: ((int8_t *)out)[o] = sat_i8(v); break;
        }
      }
    }
  }
  return 0;
}
