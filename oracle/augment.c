/*
 * ORACLE (test infrastructure only -- never linked or imported by the product path).
 *
 * CPU restatement of the reference's "heavy augmentation" kernels (BASELINE.json configs[2]):
 *
 *   warp_affine (bilinear / nearest, constant or clamp border)
 *       dali/kernels/imgproc/warp_cpu.h:143-178 (incremental source coordinates, re-anchored every 256 px)
 *       dali/kernels/imgproc/warp/map_coords.h:32-40, include/dali/core/geom/transform.h:132-145 (affine())
 *       dali/kernels/imgproc/sampler.h:60-175 (nearest + border), :258-338 (bilinear)
 *   rotate (the same warp; matrix and canvas of dali/operators/image/remap/rotate_params.h)
 *   gaussian_blur (separable, reflect-101 border, float intermediate)
 *       dali/operators/image/convolution/gaussian_blur_params.h:27-83 (diameter, window)
 *       dali/kernels/imgproc/convolution/convolution_cpu.h:152-186,241-340, separable_convolution_cpu.h:70-113
 *   color_twist
 *       dali/operators/image/color/color_twist.h:50-83,156-170 (matrix composition),
 *       include/dali/core/geom/mat.h:260-299 (products), :551-612 (Gauss-Jordan inverse with fma),
 *       dali/kernels/imgproc/pointwise/linear_transformation_cpu.h:57-77
 *   erase
 *       dali/operators/generic/erase/erase_utils.h:52-165 (region arithmetic), dali/kernels/erase/erase_cpu.h
 *
 * Must be compiled with -ffp-contract=off.  Where the reference calls an unqualified exp()/cos()/sin() on a float
 * (overload resolution depends on which headers its TU sees) this file evaluates in double and rounds to float;
 * the two choices differ in at most the last bit of a window / matrix coefficient.
 *
 * PINS (tests/test_oracle_augment_pins.py, against the independent models of tests/independent_models.py - the ones
 * the reference's own tests use, at its own bounds or tighter):
 *   Gaussian windows   every (size, sigma) pair of gaussian_blur_params_test.cc:33-57 against OpenCV's published
 *                      getGaussianKernel, 1e-7 per coefficient (that test's bound)
 *   Gaussian blur      float64 reflect-101 convolution, <= 1 LSB (operator_1/test_gaussian_blur.py:134,164)
 *   color twist / hsv  the numpy model of operator_1/test_color_twist.py:68-104, abs 1 / rel 1/512 (its bound)
 *   warp affine        exact float64 bilinear sampling with the matrices of operator_2/test_warp.py:31-52,197, <= 1 LSB
 *                      (the reference allows 8 against OpenCV's fixed-point interpolation, test_warp.py:236)
 *   rotate             canvas sizes against the reference's own model (operator_2/test_rotate.py:39-60), pixels against
 *                      the exact bilinear model with the float64 matrix of test_rotate.py:111-120, <= 1 LSB
 *   erase              numpy slice assignment, exact
 * What stays unobservable here is the last-bit behaviour of a shipped DALI binary (libm version, contraction choices);
 * the reference's own tests do not pin it either.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static uint8_t sat_u8(float v) {
  float r = roundf(v);
  if (!(r > 0)) return 0;
  if (r > 255) return 255;
  return (uint8_t)r;
}

/* ---------------------------------------------------------------------------------------------- warp_affine */
static inline int floor_int(float x) { return (int)floorf(x); }

/* border: fill != NULL -> constant border (per channel); NULL -> clamp */
static inline float fetch(const uint8_t *in, int H, int W, int C, int x, int y, int c, const float *fill) {
  if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) return in[((size_t)y * W + x) * C + c];
  if (fill) return (float)sat_u8(fill[c]); /* ConvertSat<In>(border value) */
  x = x < 0 ? 0 : x > W - 1 ? W - 1 : x;
  y = y < 0 ? 0 : y > H - 1 ? H - 1 : y;
  return in[((size_t)y * W + x) * C + c];
}

/* m: 2x3 row-major, maps DESTINATION (x, y) to SOURCE (x, y).  interp: 0 nearest, 1 linear. */
void orc_warp_affine_u8(const uint8_t *in, int H, int W, int C, const float *m, int outH, int outW, int interp,
                        const float *fill /* NULL = clamp */, uint8_t *out) {
  const float dsdx_x = m[0], dsdx_y = m[3];
  const int tile_w = 256;
  const float dtx = tile_w * dsdx_x, dty = tile_w * dsdx_y;
  for (int y = 0; y < outH; y++) {
    /* map_coords(mapping, ivec2(0, y)) = affine(M, (0 + 0.5, y + 0.5)) */
    float vx = 0 + 0.5f, vy = y + 0.5f;
    float tx = m[2]; tx += m[0] * vx; tx += m[1] * vy;
    float ty = m[5]; ty += m[3] * vx; ty += m[4] * vy;
    for (int x_tile = 0; x_tile < outW; x_tile += tile_w, tx += dtx, ty += dty) {
      int x_end = x_tile + tile_w < outW ? x_tile + tile_w : outW;
      float sx = tx, sy = ty;
      for (int x = x_tile; x < x_end; x++, sx += dsdx_x, sy += dsdx_y) {
        uint8_t *o = out + ((size_t)y * outW + x) * C;
        if (interp == 0) {
          int ix = floor_int(sx), iy = floor_int(sy);
          for (int c = 0; c < C; c++) o[c] = (uint8_t)fetch(in, H, W, C, ix, iy, c, fill);
        } else {
          float fx = sx - 0.5f, fy = sy - 0.5f;
          int x0 = floor_int(fx), y0 = floor_int(fy);
          float qx = fx - x0, px = 1 - qx, qy = fy - y0;
          for (int c = 0; c < C; c++) {
            float s00 = fetch(in, H, W, C, x0, y0, c, fill), s01 = fetch(in, H, W, C, x0 + 1, y0, c, fill);
            float s10 = fetch(in, H, W, C, x0, y0 + 1, c, fill), s11 = fetch(in, H, W, C, x0 + 1, y0 + 1, c, fill);
            float s0 = s00 * px + s01 * qx;
            float s1 = s10 * px + s11 * qx;
            o[c] = sat_u8(s0 + (s1 - s0) * qy);
          }
        }
      }
    }
  }
}

/* affine_mat_inv for a 2x3 matrix (dst->src from src->dst): used when inverse_map == False.
 * include/dali/core/geom/transform.h affine_mat_inv: invert the 3x3 extension with Gauss-Jordan (mat.h:551-612). */
static void solve_gauss3(float A[3][3], float B[3][3]) {
  for (int v = 0; v < 3; v++) {
    float max = fabsf(A[v][v]);
    int maxr = v;
    for (int i = v + 1; i < 3; i++) {
      float q = fabsf(A[i][v]);
      if (q > max) { max = q; maxr = i; }
    }
    if (!max) return;
    if (maxr != v) {
      for (int j = 0; j < 3; j++) { float t = A[v][j]; A[v][j] = A[maxr][j]; A[maxr][j] = t; }
      for (int j = 0; j < 3; j++) { float t = B[v][j]; B[v][j] = B[maxr][j]; B[maxr][j] = t; }
    }
    float x = 1.0f / A[v][v];
    A[v][v] = 1;
    for (int j = v + 1; j < 3; j++) A[v][j] *= x;
    for (int j = 0; j < 3; j++) B[v][j] *= x;
    for (int i = 0; i < 3; i++) {
      if (i == v) continue;
      float c = -A[i][v];
      A[i][v] = 0;
      for (int j = v + 1; j < 3; j++) A[i][j] = fmaf(c, A[v][j], A[i][j]);
      for (int j = 0; j < 3; j++) B[i][j] = fmaf(c, B[v][j], B[i][j]);
    }
  }
}

void orc_mat3_inverse(const float *a, float *out) {
  float A[3][3], B[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  memcpy(A, a, sizeof(A));
  solve_gauss3(A, B);
  memcpy(out, B, sizeof(B));
}

/* affine_mat_inv (include/dali/core/geom/transform.h:166-174): m = inverse(2x2) [mat.h:610-622: adjugate / det],
 * t = -m * t */
void orc_affine_inverse_2x3(const float *m, float *out) {
  float det = m[0] * m[4] - m[1] * m[3];
  float i00 = m[4] / det, i01 = -m[1] / det, i10 = -m[3] / det, i11 = m[0] / det;
  float n00 = -i00, n01 = -i01, n10 = -i10, n11 = -i11;   /* (-m) */
  float t0 = n00 * m[2]; t0 += n01 * m[5];                /* mat * vec: s = m[i][0]*v[0]; s += m[i][1]*v[1] */
  float t1 = n10 * m[2]; t1 += n11 * m[5];
  out[0] = i00; out[1] = i01; out[2] = t0;
  out[3] = i10; out[4] = i11; out[5] = t1;
}

/* ---------------------------------------------------------------------------------------------- rotate */
/* fn.rotate, two-dimensional: the destination->source matrix and canvas of dali/operators/image/remap/rotate_params.h
 * (:33-52 RotatedCanvasSize, :126-127 sign of the angle, :216-228 AdjustParams, :273-291 parity correction for ONE
 * frame), then the warp above.  out_hw: in = explicit size (or 0 0), out = the size used; keep_size: the input's. */
void orc_rotate_params(float angle_deg, int in_h, int in_w, int keep_size, int *out_hw, float *matrix) {
  const float a = -angle_deg;
  const float rad = a * (float)(M_PI / 180);
  if (!(out_hw[0] > 0 && out_hw[1] > 0)) {
    if (keep_size) {
      out_hw[0] = in_h; out_hw[1] = in_w;
    } else {
      const double eps = 1e-2, abs_cos = fabs(cos((double)rad)), abs_sin = fabs(sin((double)rad));
      int w_out = (int)ceil(abs_cos * in_w + abs_sin * in_h - eps), h_out = (int)ceil(abs_cos * in_h + abs_sin * in_w - eps);
      const int par_w = abs_sin <= abs_cos ? in_w % 2 : in_h % 2, par_h = abs_sin <= abs_cos ? in_h % 2 : in_w % 2;
      w_out += (w_out % 2) ^ par_w;
      h_out += (h_out % 2) ^ par_h;
      out_hw[0] = h_out; out_hw[1] = w_out;
    }
  }
  /* translation(in / 2) * rotation2D(-a) * translation(-out / 2); mat.h product order */
  const float c = cosf(-rad), sn = sinf(-rad);
  const float tx = in_w * 0.5f, ty = in_h * 0.5f, ox = -(out_hw[1] * 0.5f), oy = -(out_hw[0] * 0.5f);
  float m02 = c * ox; m02 += (-sn) * oy; m02 += tx;
  float m12 = sn * ox; m12 += c * oy; m12 += ty;
  matrix[0] = c; matrix[1] = -sn; matrix[2] = m02;
  matrix[3] = sn; matrix[4] = c; matrix[5] = m12;
}

/* ---------------------------------------------------------------------------------------------- gaussian_blur */
int orc_gaussian_diameter(float sigma) { return 2 * (int)ceilf(sigma * 3) + 1; }
float orc_gaussian_sigma_from_diameter(int diameter) {
  int radius = (diameter - 1) / 2;
  return (float)((radius - 1) * 0.3 + 0.8);
}

/* FillGaussian, gaussian_blur_params.h:60-83 */
void orc_gaussian_window(float sigma, int diameter, float *window) {
  int r = (diameter - 1) / 2;
  float exp_scale = 0.5f / (sigma * sigma);
  float sum = 0.f;
  for (int x = -r; x < 0; x++) {
    window[x + r] = (float)exp((double)(-(x * x * exp_scale)));
    sum += window[x + r];
  }
  sum *= 2.;
  sum += 1.0;
  float scale = 1.f / sum;
  window[r] = scale;
  for (int x = 0; x < r; x++) {
    window[x] *= scale;
    window[2 * r - x] = window[x];
  }
}

static int reflect101(int idx, int size) {
  if (size < 2) return size - 1;
  for (;;) {
    if (idx < 0) idx = -idx;
    else if (idx >= size) idx = 2 * size - 2 - idx;
    else break;
  }
  return idx;
}

/* in/out: u8 HWC.  window_x applies along W (innermost, first), window_y along H. */
void orc_gaussian_blur_u8(const uint8_t *in, int H, int W, int C, const float *window_x, int dx, const float *window_y,
                          int dy, uint8_t *out) {
  float *tmp = (float *)malloc(sizeof(float) * (size_t)H * W * C);
  int rx = (dx - 1) / 2, ry = (dy - 1) / 2;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      for (int c = 0; c < C; c++) {
        float acc = 0;
        for (int k = 0; k < dx; k++) {
          int sx = reflect101(x - rx + k, W);
          acc += in[((size_t)y * W + sx) * C + c] * window_x[k];
        }
        tmp[((size_t)y * W + x) * C + c] = acc * 1.0f;
      }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      for (int c = 0; c < C; c++) {
        float acc = 0;
        for (int k = 0; k < dy; k++) {
          int sy = reflect101(y - ry + k, H);
          acc += window_y[k] * tmp[((size_t)sy * W + x) * C + c];
        }
        out[((size_t)y * W + x) * C + c] = sat_u8(acc * 1.0f);
      }
  free(tmp);
}

/* ---------------------------------------------------------------------------------------------- color_twist */
static void mat3_mul(const float a[9], const float b[9], float out[9]) {
  float r[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float s = a[3 * i] * b[j];
      s += a[3 * i + 1] * b[3 + j];
      s += a[3 * i + 2] * b[6 + j];
      r[3 * i + j] = s;
    }
  memcpy(out, r, sizeof(r));
}
static void mat3_diag(float v, float out[9]) {
  memset(out, 0, 9 * sizeof(float));
  out[0] = out[4] = out[8] = v;
}

/* ColorTwistBase::DetermineTransformation, color_twist.h:156-170.  half_range = 128 for integer input. */
void orc_color_twist_matrix(float hue, float saturation, float value, float brightness, float contrast, float *matrix,
                            float *offset) {
  const float rgb2yiq[9] = {.299f, .587f, .114f, .596f, -.274f, -.321f, .211f, -.523f, .311f};
  float yiq2rgb[9];
  orc_mat3_inverse(rgb2yiq, yiq2rgb);
  const float h_rad = (float)(hue * M_PI / 180);
  float hm[9], sm[9], t[9], d[9];
  mat3_diag(1, hm);
  hm[4] = (float)cos((double)h_rad); hm[8] = (float)cos((double)h_rad);
  hm[5] = (float)sin((double)h_rad); hm[7] = (float)-sin((double)h_rad);
  mat3_diag(1, sm);
  sm[4] = saturation; sm[8] = saturation;
  mat3_diag(brightness, t);
  mat3_diag(contrast, d);
  mat3_mul(t, d, t);
  mat3_mul(t, yiq2rgb, t);
  mat3_mul(t, hm, t);
  mat3_mul(t, sm, t);
  mat3_diag(value, d);
  mat3_mul(t, d, t);
  mat3_mul(t, rgb2yiq, t);
  memcpy(matrix, t, sizeof(t));
  const float half_range = 128.f;
  *offset = (half_range - half_range * contrast) * brightness;
}

/* LinearTransformationCpu::Run: out = ConvertSat(M * px + offset) */
void orc_linear_transform_u8(const uint8_t *in, int64_t npixels, const float *m, const float *offset3, uint8_t *out) {
  for (int64_t p = 0; p < npixels; p++) {
    float v[3] = {(float)in[3 * p], (float)in[3 * p + 1], (float)in[3 * p + 2]};
    for (int i = 0; i < 3; i++) {
      float s = m[3 * i] * v[0];
      s += m[3 * i + 1] * v[1];
      s += m[3 * i + 2] * v[2];
      out[3 * p + i] = sat_u8(s + offset3[i]);
    }
  }
}

/* ---------------------------------------------------------------------------------------------- erase */
/* regions: nregions x {anchor_y, anchor_x, shape_y, shape_x} already in pixels as floats (before truncation);
 * flags: bit0 normalized_anchor, bit1 normalized_shape, bit2 centered_anchor.  fill: C values (or 1, broadcast). */
void orc_erase_u8(const uint8_t *in, int H, int W, int C, const float *anchors_yx, const float *shapes_yx, int nregions,
                  int flags, const float *fill, int nfill, uint8_t *out) {
  memcpy(out, in, (size_t)H * W * C);
  for (int r = 0; r < nregions; r++) {
    int64_t a[2], s[2];
    int dims[2] = {H, W};
    for (int j = 0; j < 2; j++) {
      float anchor_val = (flags & 1) ? anchors_yx[2 * r + j] * dims[j] : anchors_yx[2 * r + j];
      float shape_val = (flags & 2) ? shapes_yx[2 * r + j] * dims[j] : shapes_yx[2 * r + j];
      if (flags & 4) anchor_val -= shape_val / 2;
      a[j] = (int64_t)anchor_val;
      int64_t end = (int64_t)(anchor_val + shape_val);
      s[j] = end - a[j];
    }
    int64_t y0 = a[0] < 0 ? 0 : a[0], y1 = a[0] + s[0] > H ? H : a[0] + s[0];
    int64_t x0 = a[1] < 0 ? 0 : a[1], x1 = a[1] + s[1] > W ? W : a[1] + s[1];
    for (int64_t y = y0; y < y1; y++)
      for (int64_t x = x0; x < x1; x++)
        for (int c = 0; c < C; c++) out[((size_t)y * W + x) * C + c] = sat_u8(fill[nfill > 1 ? c : 0]);
  }
}
