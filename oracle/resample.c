/*
 * ORACLE (test infrastructure only -- never linked or imported by the product path).
 *
 * CPU restatement of the reference's separable 2-D resampling (CPU backend) as used by
 * RandomResizedCrop / Resize:
 *
 *   filter table + rescale/support/eval  dali/kernels/imgproc/resample/resampling_filters.cuh:32-68
 *                                        dali/kernels/imgproc/resample/resampling_filters.cu:66-142
 *   filter choice, default radius        dali/kernels/imgproc/resample/resampling_setup.cc:27-76
 *                                        dali/kernels/imgproc/resample/params.h:43-60
 *   scale / origin / source ROI          resampling_setup.cc:84-122
 *   pass-order cost model                resampling_setup.cc:131-201
 *   per-sample setup (tmp shape, origin shift on the non-first axis)  resampling_setup.cc:271-337
 *   coefficient/index tables             dali/kernels/imgproc/resample/resampling_impl_cpu.cc:22-47
 *   horizontal / vertical passes         dali/kernels/imgproc/resample/resampling_impl_cpu.h:50-390
 *   two passes through an fp32 tmp       dali/kernels/imgproc/resample/separable_cpu.h:152-241
 *   final rounding                       SSE2 body: _mm_cvtps_epi32 after clamp (half-even),
 *                                        dali/kernels/common/simd.h:53-56; scalar tail:
 *                                        clamp(std::round) (half-away), include/dali/core/convert.h:306-321
 *
 * The arithmetic order is the reference's: coefficients pre-normalised by division, products
 * formed and added separately (no fma: the reference is built for baseline x86-64), taps
 * accumulated in increasing k.  This file must be compiled with -ffp-contract=off.
 *
 * Pinning: filter-support / symmetry known answers from the reference's own unit test
 * (dali/kernels/test/resampling_test/resampling_impl_cpu_test.cc:27-90) and the
 * resize-vs-PIL tolerance of dali/test/python/operator_2/test_resize.py:96-121,582-589
 * are checked in tests/test_oracle_resample.py.
 */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_SQRT2
#define M_SQRT2 1.41421356237309504880
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* CUBIC / LANCZOS3 / GAUSSIAN: the reference's tabulated windows (resampling_filters.cu:38-142, resampling_windows.h:44-65) */
enum { ORC_FILTER_NN = 0, ORC_FILTER_LINEAR = 1, ORC_FILTER_TRIANGULAR = 2, ORC_FILTER_CUBIC = 3, ORC_FILTER_LANCZOS3 = 4,
       ORC_FILTER_GAUSSIAN = 5 };
#define ORC_MAX_COEFFS 193

typedef struct {
  int num_coeffs;
  float anchor, scale;
  float coeffs[ORC_MAX_COEFFS];
} orc_filter;

/* resampling_filters.cuh:38-42 */
static void filter_rescale(orc_filter *f, float support) {
  float old_scale = f->scale;
  f->scale = (f->num_coeffs - 1) / support;
  f->anchor = f->anchor * old_scale / f->scale;
}

/* resampling_filters.cuh:44-46 */
static int filter_support(const orc_filter *f) { return (int)ceilf((f->num_coeffs - 1) / f->scale); }

/* resampling_filters.cuh:48-67 (host branch) */
static float filter_eval(const orc_filter *f, float x) {
  if (!(x > -1)) return 0;
  if (x >= f->num_coeffs) return 0;
  int x0 = (int)floorf(x);
  int x1 = x0 + 1;
  float d = x - x0;
  float f0 = x0 < 0.0f ? 0 : f->coeffs[x0];
  float f1 = x1 >= f->num_coeffs ? 0.0f : f->coeffs[x1];
  return f0 + d * (f1 - f0);
}

/* resampling_filters.cu:81-86,104-107,138-142: Triangular = {0,1,0}, anchor 1, scale 1 */
static orc_filter filter_triangular(float radius) {
  orc_filter f;
  f.num_coeffs = 3;
  f.anchor = 1;
  f.scale = (3 - 1) * 0.5f;
  f.coeffs[0] = 0; f.coeffs[1] = 1; f.coeffs[2] = 0;
  float s = 2 * radius;
  filter_rescale(&f, s > 1.0f ? s : 1.0f);
  return f;
}

/* include/dali/core/math_util.h:188-194 (float overload) */
static float sincf_(float x) {
  x = (float)(x * M_PI);   /* `x *= M_PI`: the product is formed in double */
  if (fabsf(x) < 1e-5f) return 1.0f - x * x * (1.0f / 6);
  return sinf(x) / x;
}
/* resampling_windows.h:44-48 */
static float lanczos_window(float x, float a) {
  if (fabsf(x) >= a) return 0.0f;
  return sincf_(x) * sincf_(x / a);
}
/* resampling_windows.h:54-65 */
static float cubic_window(float x) {
  x = fabsf(x);
  if (x >= 2) return 0;
  float x2 = x * x, x3 = x2 * x;
  if (x > 1) return -0.5f * x3 + 2.5f * x2 - 4.0f * x + 2.0f;
  return 1.5f * x3 - 2.5f * x2 + 1.0f;
}
/* InitFilters, resampling_filters.cu:66-108: {coeffs, size, scale 1, anchor (size - 1) / 2}, Lanczos rescaled to 6,
 * cubic to 4; then the per-use rescale of resampling_filters.cu:115-137 */
static orc_filter filter_base(int size) {
  orc_filter f;
  memset(&f, 0, sizeof(f));
  f.num_coeffs = size;
  f.anchor = 1;               /* add_filter pushes {base, size, 1, (size - 1) * 0.5f}: the struct order is */
  f.scale = (size - 1) * 0.5f; /* {coeffs, num_coeffs, anchor, scale}, so anchor = 1 and scale = (size - 1) / 2 */
  return f;
}
static orc_filter filter_gaussian(float sigma) {
  orc_filter f = filter_base(65);
  for (int i = 0; i < 65; i++) {
    float x = 4 * (i - (65 - 1) * 0.5f) / (65 - 1);
    f.coeffs[i] = expf(-x * x);
  }
  float s = (float)(4 * M_SQRT2) * sigma;
  filter_rescale(&f, s > 1.0f ? s : 1.0f);
  return f;
}
static orc_filter filter_lanczos3(float radius) {
  const int size = 2 * 3 * 32 + 1;
  orc_filter f = filter_base(size);
  for (int i = 0; i < size; i++) {
    float x = 2 * 3.0f * (i - (size - 1) * 0.5f) / (size - 1);
    f.coeffs[i] = lanczos_window(x, 3.0f);
  }
  filter_rescale(&f, 6);
  filter_rescale(&f, 2.0f * (radius > 3.0f ? radius : 3.0f));
  return f;
}
static orc_filter filter_cubic(float radius) {
  orc_filter f = filter_base(129);
  for (int i = 0; i < 129; i++) {
    float x = 4 * (i - (129 - 1) * 0.5f) / (129 - 1);
    f.coeffs[i] = cubic_window(x);
  }
  filter_rescale(&f, 4);
  filter_rescale(&f, 2.0f * (radius > 2.0f ? radius : 2.0f));
  return f;
}

/* resampling_impl_cpu.cc:22-47 */
static void init_resampling_filter(int32_t *out_indices, float *out_coeffs, int out_size,
                                   float srcx_0, float scale, const orc_filter *filter) {
  srcx_0 += 0.5f * scale - 0.5f - filter->anchor;
  int support = filter_support(filter);
  for (int x = 0; x < out_size; x++) {
    float sx0f = x * scale + srcx_0;
    int sx0 = (int)ceilf(sx0f);
    out_indices[x] = sx0;
    const float f0 = sx0 - sx0f;
    float sum = 0;
    for (int k = 0; k < support; k++) {
      float c = filter_eval(filter, (f0 + k) * filter->scale);
      out_coeffs[support * x + k] = c;
      sum += c;
    }
    if (sum) {
      for (int k = 0; k < support; k++) out_coeffs[support * x + k] /= sum;
    }
  }
}

typedef struct {
  /* per axis: 0 = x (W), 1 = y (H); vec order as in the reference */
  int filter_type[2];
  orc_filter filter[2];
  float origin[2], scale[2];
  int roi_lo[2], roi_hi[2];
  int in_size[2], out_size[2];
  int order[2];
  int support[2];
} orc_resample_setup;

/* Horizontal pass for one row, Out = float (no rounding) or u8.
 * resampling_impl_cpu.h:50-86 (ResampleCol), 126-226 (SIMD body), 286-313 (region split).
 * `simd_mask[x]`, when not NULL, receives 1 for columns handled by the SSE2 body. */
static void horz_regions(int out_w, int in_w, const int32_t *idx, int support, int lanes,
                         uint8_t *simd_mask) {
  /* GetFirstAndLastRegularCol, resampling_impl_cpu.h:246-266 */
  int flipped = idx[out_w - 1] < idx[0];
  int first_regular = 0, last_regular = out_w - 1;
  if (flipped) {
    while (first_regular < out_w && idx[first_regular] + support > in_w) first_regular++;
    while (last_regular >= 0 && idx[last_regular] < 0) last_regular--;
  } else {
    while (first_regular < out_w && idx[first_regular] < 0) first_regular++;
    while (last_regular >= 0 && idx[last_regular] + support > in_w) last_regular--;
  }
  int max_one_sided = first_regular < last_regular + 1 ? first_regular : last_regular + 1;
  int bounds[5] = {0, max_one_sided, first_regular, last_regular + 1, out_w};
  memset(simd_mask, 0, (size_t)out_w);
  int x = 0;
  for (int r = 0; r < 4; r++) {
    int ox1 = bounds[r + 1];
    /* impl.run(out,in,x,ox1,...): SIMD while x + lanes <= ox1, then scalar up to ox1;
     * returns x (unchanged when x >= ox1) */
    for (; x + lanes <= ox1; x += lanes)
      for (int l = 0; l < lanes; l++) simd_mask[x + l] = 1;
    for (; x < ox1; x++) simd_mask[x] = 0;
  }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* ---- one full resample, u8 HWC in -> u8 HWC out (float tmp) ---- */

static void setup_sample(orc_resample_setup *s, int H, int W, int use_roi, const float *roi_yx0yx1,
                         int outH, int outW, int min_filter, int mag_filter, int antialias) {
  /* params[dim]: dim 0 = H, dim 1 = W;  axis = 1 - dim */
  s->in_size[0] = W; s->in_size[1] = H;
  s->out_size[0] = outW; s->out_size[1] = outH;
  for (int dim = 0; dim < 2; dim++) {
    int axis = 1 - dim;
    float roi_start = 0, roi_end = (float)s->in_size[axis];
    if (use_roi) { roi_start = roi_yx0yx1[dim]; roi_end = roi_yx0yx1[2 + dim]; }
    /* SetFilters, resampling_setup.cc:46-76 */
    float in_size = use_roi ? fabsf(roi_end - roi_start) : (float)s->in_size[axis];
    int type = s->out_size[axis] < in_size ? min_filter : mag_filter;
    int aa = antialias;
    if (type != ORC_FILTER_NN) {
      if (aa && type == ORC_FILTER_LINEAR) type = ORC_FILTER_TRIANGULAR;
      else if (!aa && type == ORC_FILTER_TRIANGULAR) type = ORC_FILTER_LINEAR;
    }
    /* DefaultFilterRadius, params.h:43-60 */
    float radius = 1;
    {
      int a = aa && (in_size > s->out_size[axis]);
      float ratio = in_size / s->out_size[axis];
      if (type == ORC_FILTER_TRIANGULAR || type == ORC_FILTER_GAUSSIAN) radius = a ? ratio : 1;
      else if (type == ORC_FILTER_CUBIC) radius = a ? 2 * ratio : 2;
      else if (type == ORC_FILTER_LANCZOS3) radius = a ? 3 * ratio : 3;
    }
    s->filter_type[axis] = type;
    if (type == ORC_FILTER_LINEAR) s->filter[axis] = filter_triangular(1);
    else if (type == ORC_FILTER_TRIANGULAR) s->filter[axis] = filter_triangular(radius);
    else if (type == ORC_FILTER_GAUSSIAN) s->filter[axis] = filter_gaussian((float)(radius * 0.5f / M_SQRT2));  /* GetResamplingFilter */
    else if (type == ORC_FILTER_CUBIC) s->filter[axis] = filter_cubic(radius);
    else if (type == ORC_FILTER_LANCZOS3) s->filter[axis] = filter_lanczos3(radius);
    else { memset(&s->filter[axis], 0, sizeof(orc_filter)); s->filter[axis].num_coeffs = 0; s->filter[axis].anchor = 0; s->filter[axis].scale = 1; }

    /* ComputeScaleAndROI, resampling_setup.cc:84-122 */
    s->origin[axis] = roi_start;
    s->scale[axis] = (roi_end - roi_start) / s->out_size[axis];
    int support = s->filter[axis].num_coeffs ? filter_support(&s->filter[axis]) : 1;
    float lo, hi;
    if (roi_start <= roi_end) {
      lo = roi_start - s->filter[axis].anchor;
      hi = roi_end - s->filter[axis].anchor + support;
    } else {
      lo = roi_end - s->filter[axis].anchor;
      hi = roi_start - s->filter[axis].anchor + support;
    }
    int l = (int)floorf(lo), h = (int)ceilf(hi);
    s->roi_lo[axis] = clampi(l, 0, s->in_size[axis]);
    s->roi_hi[axis] = clampi(h, 0, s->in_size[axis]);
    int sup = s->filter[axis].num_coeffs ? filter_support(&s->filter[axis]) : -1;
    s->support[axis] = sup > 1 ? sup : 1;
  }
  /* GetProcessingOrder, resampling_setup.cc:131-201 (2-D: two candidate orders, DFS with a<b first) */
  float best = 1e+30f;
  for (int first = 0; first < 2; first++) {
    int second = 1 - first;
    int cur[2] = {s->roi_hi[0] - s->roi_lo[0], s->roi_hi[1] - s->roi_lo[1]};
    float total = 0;
    int ax[2] = {first, second};
    for (int p = 0; p < 2; p++) {
      int a = ax[p];
      cur[a] = s->out_size[a];
      int64_t vol = (int64_t)cur[0] * cur[1];
      float base = (float)(s->support[a] * vol); /* integer product, then to float */
      float mul = a == 0 ? 1.4f : 1.0f;
      float cost = mul * base + vol * 3.0f;
      total = total + cost;
    }
    if (total < best) { best = total; s->order[0] = first; s->order[1] = second; }
  }
}

/*
 * in:  u8 [H][W][C];  out: u8 [outH][outW][C]
 * roi: {y0, x0, y1, x1} in source pixels (floats), used when use_roi != 0
 * info (optional, 8 ints): order0, order1, support_x, support_y, tmp_w, tmp_h, roi_lo_x, roi_lo_y
 * round_mode: 0 = reference CPU (SIMD body half-even, scalar tail half-away),
 *             1 = half-away everywhere, 2 = half-even everywhere
 * Returns 0 on success.
 */
/* element types of the typed entry point (the reference resamples u8 / i16 / u16 / f32, resampling_batch.cu:125-152) */
enum { ORC_T_U8 = 0, ORC_T_I16 = 1, ORC_T_U16 = 2, ORC_T_F32 = 3 };

/* store() of a SIMD body: clamp in float, then cvtps = round half to even (simd.h:53-56,228-267);
 * scalar tails: ConvertSat = clamp(std::round) (convert.h:306-321).  Float outputs are not rounded. */
static float sat_typed(float v, int type, int even) {
  float lo = type == ORC_T_I16 ? -32768.0f : 0.0f;
  float hi = type == ORC_T_U8 ? 255.0f : type == ORC_T_I16 ? 32767.0f : 65535.0f;
  if (type == ORC_T_F32) return v;
  if (even) {
    float c = v;
    if (!(c > lo)) c = lo;
    if (c > hi) c = hi;
    return (float)lrintf(c);
  }
  float r = roundf(v);
  if (!(r > lo)) return lo;
  if (r > hi) return hi;
  return r;
}

/* core: input as float values (every u8 / i16 / u16 / f32 element is exactly a float), `out_vals` receives the output
 * elements as floats (integers for the integer types), `out_f32` the unrounded sums. */
static int resample_impl(const float *in, int H, int W, int C, int use_roi, const float *roi,
                         int outH, int outW, int min_filter, int mag_filter, int antialias,
                         int round_mode, int out_type, float *out, float *out_f32 /* optional: unrounded result */,
                         float *tmp_out /* optional */, int *info) {
  const int lanes = out_type == ORC_T_U8 ? 16 : out_type == ORC_T_F32 ? 4 : 8;  /* 16 bytes / sizeof(Out) */
  if (H <= 0 || W <= 0 || outH <= 0 || outW <= 0 || C <= 0) return 1;
  orc_resample_setup s;
  setup_sample(&s, H, W, use_roi, roi, outH, outW, min_filter, mag_filter, antialias);
  if ((s.filter_type[0] == ORC_FILTER_NN) != (s.filter_type[1] == ORC_FILTER_NN)) return 2;  /* mixed: not restated */

  int first = s.order[0], second = s.order[1];
  /* SetupSample tail, resampling_setup.cc:326-336: the non-first-pass axis is cut to the source
   * ROI: origin shifted, base pointer offset, extent = roi extent */
  int in_ext[2] = {W, H};
  int in_off[2] = {0, 0};
  float origin[2] = {s.origin[0], s.origin[1]};
  {
    int a = second;
    origin[a] -= s.roi_lo[a];
    in_off[a] = s.roi_lo[a];
    in_ext[a] = s.roi_hi[a] - s.roi_lo[a];
  }
  /* tmp shape: roi extent with the first-pass axis replaced by the output size */
  int tmp_size[2] = {s.roi_hi[0] - s.roi_lo[0], s.roi_hi[1] - s.roi_lo[1]};
  tmp_size[first] = s.out_size[first];
  /* Note: for the first-pass axis the input extent is the whole image, but the pass only
   * produces the tmp extent on the other axis (= in_ext[second]). */
  int tmp_w = tmp_size[0], tmp_h = tmp_size[1];
  if (info) {
    info[0] = first; info[1] = second; info[2] = s.support[0]; info[3] = s.support[1];
    info[4] = tmp_w; info[5] = tmp_h; info[6] = s.roi_lo[0]; info[7] = s.roi_lo[1];
  }
  if (tmp_w <= 0 || tmp_h <= 0) { if (out) memset(out, 0, sizeof(float) * (size_t)outH * outW * C); return 0; }
  if (s.filter_type[0] == ORC_FILTER_NN) {
    /* ResampleNN (resampling_impl_cpu.h:523-606) on the surface SetupSample leaves: whole image on the first-pass
     * axis, the source ROI on the other.  Rows advance by repeated float additions, columns are computed directly;
     * scale.x == 1 is the copy path with repeated borders. */
    const float *base = in + ((size_t)in_off[1] * W + in_off[0]) * C;
    float sy = origin[1] + 0.5f * s.scale[1];
    int sx0 = (int)floorf(origin[0] + 0.5f);
    for (int y = 0; y < outH; y++, sy += s.scale[1]) {
      int srcy = clampi((int)floorf(sy), 0, in_ext[1] - 1);
      for (int x = 0; x < outW; x++) {
        int srcx;
        if (s.scale[0] == 1) srcx = sx0 + x;
        else srcx = (int)floorf(origin[0] + (x + 0.5f) * s.scale[0]);
        srcx = clampi(srcx, 0, in_ext[0] - 1);
        for (int c = 0; c < C; c++) {
          float v = base[((size_t)srcy * W + srcx) * C + c];
          if (out) out[((size_t)y * outW + x) * C + c] = v;
          if (out_f32) out_f32[((size_t)y * outW + x) * C + c] = v;
        }
      }
    }
    return 0;
  }

  float *tmp = (float *)malloc(sizeof(float) * (size_t)tmp_w * tmp_h * C);
  int max_out = outW > outH ? outW : outH;
  int max_sup = s.support[0] > s.support[1] ? s.support[0] : s.support[1];
  int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_out);
  float *coef = (float *)malloc(sizeof(float) * (size_t)max_out * max_sup);
  uint8_t *mask = (uint8_t *)malloc((size_t)(outW > 16 ? outW : 16));
  (void)lanes;

  /* ---------- pass 0: u8 -> float tmp ---------- */
  {
    int axis = first;
    int sup = filter_support(&s.filter[axis]);
    init_resampling_filter(idx, coef, s.out_size[axis], origin[axis], s.scale[axis], &s.filter[axis]);
    const float *base = in + ((size_t)in_off[1] * W + in_off[0]) * C;
    if (axis == 0) { /* horizontal: rows = in_ext[1] (ROI rows), in width = W (whole) */
      for (int y = 0; y < tmp_h; y++) {
        const float *row = base + (size_t)y * W * C;
        float *orow = tmp + (size_t)y * tmp_w * C;
        for (int x = 0; x < tmp_w; x++) {
          int x0 = idx[x];
          for (int c = 0; c < C; c++) {
            float acc = 0;
            for (int k = 0; k < sup; k++) {
              int sx = clampi(x0 + k, 0, in_ext[0] - 1);
              acc += coef[x * sup + k] * row[sx * C + c];
            }
            orow[x * C + c] = acc;
          }
        }
      }
    } else { /* vertical: cols = in_ext[0] (ROI cols), in height = H (whole) */
      for (int y = 0; y < tmp_h; y++) {
        float *orow = tmp + (size_t)y * tmp_w * C;
        for (int i = 0; i < tmp_w * C; i++) {
          float acc = 0;
          for (int k = 0; k < sup; k++) {
            int sy = clampi(idx[y] + k, 0, in_ext[1] - 1);
            acc += base[(size_t)sy * W * C + i] * coef[y * sup + k];
          }
          orow[i] = acc;
        }
      }
    }
  }
  if (tmp_out) memcpy(tmp_out, tmp, sizeof(float) * (size_t)tmp_w * tmp_h * C);

  /* ---------- pass 1: float tmp -> u8 out ---------- */
  {
    int axis = second;
    int sup = filter_support(&s.filter[axis]);
    init_resampling_filter(idx, coef, s.out_size[axis], origin[axis], s.scale[axis], &s.filter[axis]);
    if (axis == 0) { /* horizontal over tmp rows (tmp_h == outH) */
      horz_regions(outW, tmp_w, idx, sup, lanes, mask);
      for (int y = 0; y < outH; y++) {
        const float *row = tmp + (size_t)y * tmp_w * C;
        float *orow = out ? out + (size_t)y * outW * C : NULL;
        for (int x = 0; x < outW; x++) {
          int x0 = idx[x];
          int even = round_mode == 2 || (round_mode == 0 && mask[x]);
          for (int c = 0; c < C; c++) {
            float acc = 0;
            for (int k = 0; k < sup; k++) {
              int sx = clampi(x0 + k, 0, tmp_w - 1);
              acc += coef[x * sup + k] * row[sx * C + c];
            }
            if (out_f32) out_f32[((size_t)y * outW + x) * C + c] = acc;
            if (out) orow[x * C + c] = sat_typed(acc, out_type, even);
          }
        }
      }
    } else { /* vertical over tmp cols (tmp_w == outW) */
      int flat_w = outW * C;
      for (int y = 0; y < outH; y++) {
        float *orow = out ? out + (size_t)y * outW * C : NULL;
        for (int x0 = 0; x0 < flat_w; x0 += 256) { /* ResampleVert tile, resampling_impl_cpu.h:362-390 */
          int end = x0 + 256 <= flat_w ? x0 + 256 : flat_w;
          int i = x0;
          int simd_end = x0 + ((end - x0) / lanes) * lanes;
          for (; i < end; i++) {
            float acc = 0;
            for (int k = 0; k < sup; k++) {
              int sy = clampi(idx[y] + k, 0, tmp_h - 1);
              acc += tmp[(size_t)sy * tmp_w * C + i] * coef[y * sup + k];
            }
            int even = round_mode == 2 || (round_mode == 0 && i < simd_end);
            if (out_f32) out_f32[(size_t)y * outW * C + i] = acc;
            if (out) orow[i] = sat_typed(acc, out_type, even);
          }
        }
      }
    }
  }
  free(tmp); free(idx); free(coef); free(mask);
  return 0;
}

static float *to_float_u8(const uint8_t *in, size_t n) {
  float *f = (float *)malloc(sizeof(float) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) f[i] = in[i];
  return f;
}

int orc_resample_u8(const uint8_t *in, int H, int W, int C, int use_roi, const float *roi,
                    int outH, int outW, int min_filter, int mag_filter, int antialias,
                    int round_mode, uint8_t *out, float *tmp_out /* optional */, int *info) {
  if (H <= 0 || W <= 0 || outH <= 0 || outW <= 0 || C <= 0) return 1;
  size_t n_out = (size_t)outH * outW * C;
  float *inf = to_float_u8(in, (size_t)H * W * C);
  float *outf = (float *)malloc(sizeof(float) * n_out);
  int rc = resample_impl(inf, H, W, C, use_roi, roi, outH, outW, min_filter, mag_filter, antialias, round_mode, ORC_T_U8,
                         outf, NULL, tmp_out, info);
  if (rc == 0) for (size_t i = 0; i < n_out; i++) out[i] = (uint8_t)outf[i];
  free(inf); free(outf);
  return rc;
}

/* Same resampling with the float result of the second pass returned as it is (fn.resize(dtype=FLOAT):
 * the final ConvertSat<float> is the identity).  out: float [outH][outW][C]. */
int orc_resample_u8_to_f32(const uint8_t *in, int H, int W, int C, int use_roi, const float *roi,
                           int outH, int outW, int min_filter, int mag_filter, int antialias, float *out) {
  if (H <= 0 || W <= 0 || outH <= 0 || outW <= 0 || C <= 0) return 1;
  float *inf = to_float_u8(in, (size_t)H * W * C);
  int rc = resample_impl(inf, H, W, C, use_roi, roi, outH, outW, min_filter, mag_filter, antialias, 0, ORC_T_U8, NULL, out, NULL,
                         NULL);
  free(inf);
  return rc;
}

/* Typed entry point: `in` holds the elements of a u8 / i16 / u16 / f32 image as floats, `out` receives the elements of
 * an image of type out_type (ORC_T_*; ORC_T_F32 = the unrounded result) as floats. */
int orc_resample_typed(const float *in, int H, int W, int C, int use_roi, const float *roi, int outH, int outW,
                       int min_filter, int mag_filter, int antialias, int out_type, float *out) {
  return resample_impl(in, H, W, C, use_roi, roi, outH, outW, min_filter, mag_filter, antialias, 0, out_type, out, NULL, NULL,
                       NULL);
}

/* coefficients of the tabulated filters for known-answer tests: type = ORC_FILTER_*, returns num_coeffs and fills
 * coeffs[<= 193], *scale, *anchor for the given radius (sigma derived like GetResamplingFilter) */
int orc_filter_table(int type, float radius, float *coeffs, float *scale, float *anchor, int *support) {
  orc_filter f;
  if (type == ORC_FILTER_GAUSSIAN) f = filter_gaussian((float)(radius * 0.5f / M_SQRT2));
  else if (type == ORC_FILTER_CUBIC) f = filter_cubic(radius);
  else if (type == ORC_FILTER_LANCZOS3) f = filter_lanczos3(radius);
  else f = filter_triangular(radius);
  memcpy(coeffs, f.coeffs, sizeof(float) * (size_t)f.num_coeffs);
  *scale = f.scale; *anchor = f.anchor; *support = filter_support(&f);
  return f.num_coeffs;
}

int orc_triangular_support(float radius) {
  orc_filter f = filter_triangular(radius);
  return filter_support(&f);
}

/* Fills idx[out_size], coeffs[out_size*support] for a triangular filter of the given radius;
 * returns support. */
int orc_init_triangular(int out_size, float srcx0, float scale, float radius, int32_t *idx,
                        float *coeffs) {
  orc_filter f = filter_triangular(radius);
  init_resampling_filter(idx, coeffs, out_size, srcx0, scale, &f);
  return filter_support(&f);
}
