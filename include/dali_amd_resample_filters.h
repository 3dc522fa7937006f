/* Resampling filter tables shared by the kernel library (host side of csrc/resample.hip) and the CPU backend
 * (host/host_kernels.cpp): the reference's tabulated windows, built once with the host's libm exactly as
 * InitFilters does (dali/kernels/imgproc/resample/resampling_filters.cu:38-108, resampling_windows.h:44-65,
 * include/dali/core/math_util.h:188-194), and ResamplingFilter::operator() - linear interpolation in the table
 * (resampling_filters.cuh:48-67).  Header-only, host + device.  Not part of the C ABI. */
#ifndef DALI_AMD_RESAMPLE_FILTERS_H_
#define DALI_AMD_RESAMPLE_FILTERS_H_
#include <math.h>

#if defined(__HIPCC__)
#define DALIAMD_RF_HD __host__ __device__
#else
#define DALIAMD_RF_HD
#endif

/* filter kinds of daliamdResampleDesc.filter_kind[] */
enum { DALIAMD_FK_NN = 0, DALIAMD_FK_TRIANGULAR = 1, DALIAMD_FK_GAUSSIAN = 2, DALIAMD_FK_LANCZOS3 = 3, DALIAMD_FK_CUBIC = 4 };

enum {
  DALIAMD_RF_GAUSSIAN_SIZE = 65, DALIAMD_RF_LANCZOS_SIZE = 2 * 3 * 32 + 1, DALIAMD_RF_CUBIC_SIZE = 129,
  DALIAMD_RF_GAUSSIAN_OFF = 0, DALIAMD_RF_LANCZOS_OFF = 65, DALIAMD_RF_CUBIC_OFF = 65 + 193, DALIAMD_RF_TOTAL = 65 + 193 + 129
};

DALIAMD_RF_HD static inline int daliamdFilterTableOffset(int kind) {
  return kind == DALIAMD_FK_GAUSSIAN ? DALIAMD_RF_GAUSSIAN_OFF : kind == DALIAMD_FK_LANCZOS3 ? DALIAMD_RF_LANCZOS_OFF : DALIAMD_RF_CUBIC_OFF;
}
DALIAMD_RF_HD static inline int daliamdFilterTableSize(int kind) {
  return kind == DALIAMD_FK_TRIANGULAR ? 3 : kind == DALIAMD_FK_GAUSSIAN ? DALIAMD_RF_GAUSSIAN_SIZE
         : kind == DALIAMD_FK_LANCZOS3 ? DALIAMD_RF_LANCZOS_SIZE : kind == DALIAMD_FK_CUBIC ? DALIAMD_RF_CUBIC_SIZE : 0;
}

/* host only: fills table[DALIAMD_RF_TOTAL] */
static inline void daliamdBuildFilterTables(float *table) {
  for (int i = 0; i < DALIAMD_RF_GAUSSIAN_SIZE; i++) {
    float x = 4 * (i - (DALIAMD_RF_GAUSSIAN_SIZE - 1) * 0.5f) / (DALIAMD_RF_GAUSSIAN_SIZE - 1);
    table[DALIAMD_RF_GAUSSIAN_OFF + i] = expf(-x * x);
  }
  for (int i = 0; i < DALIAMD_RF_LANCZOS_SIZE; i++) {
    float x = 2 * 3.0f * (i - (DALIAMD_RF_LANCZOS_SIZE - 1) * 0.5f) / (DALIAMD_RF_LANCZOS_SIZE - 1);
    float w = 0.0f;
    if (fabsf(x) < 3.0f) {
      float a = (float)(x * M_PI), b = (float)((x / 3.0f) * M_PI);   /* `x *= M_PI`: the product is formed in double */
      float sa = fabsf(a) < 1e-5f ? 1.0f - a * a * (1.0f / 6) : sinf(a) / a;
      float sb = fabsf(b) < 1e-5f ? 1.0f - b * b * (1.0f / 6) : sinf(b) / b;
      w = sa * sb;
    }
    table[DALIAMD_RF_LANCZOS_OFF + i] = w;
  }
  for (int i = 0; i < DALIAMD_RF_CUBIC_SIZE; i++) {
    float x = fabsf(4 * (i - (DALIAMD_RF_CUBIC_SIZE - 1) * 0.5f) / (DALIAMD_RF_CUBIC_SIZE - 1));
    float w = 0.0f;
    if (x < 2) {
      float x2 = x * x, x3 = x2 * x;
      w = x > 1 ? -0.5f * x3 + 2.5f * x2 - 4.0f * x + 2.0f : 1.5f * x3 - 2.5f * x2 + 1.0f;
    }
    table[DALIAMD_RF_CUBIC_OFF + i] = w;
  }
}

/* ResamplingFilter::operator(): `coeffs` = the filter's table (num_coeffs entries) */
template <typename CoefPtr>
DALIAMD_RF_HD static inline float daliamdFilterEval(CoefPtr coeffs, int num_coeffs, float x) {
  if (!(x > -1)) return 0;
  if (x >= num_coeffs) return 0;
  int x0 = (int)floorf(x);
  int x1 = x0 + 1;
  float d = x - x0;
  float f0 = x0 < 0 ? 0.0f : coeffs[x0];
  float f1 = x1 >= num_coeffs ? 0.0f : coeffs[x1];
  return f0 + d * (f1 - f0);
}

/* first source pixel of output pixel o for a nearest-neighbour axis (ResampleNN, resampling_impl_cpu.h:523-606):
 * columns are computed directly (scale 1: the copy path), rows advance by repeated float additions */
DALIAMD_RF_HD static inline int daliamdNearestIndex(int axis, int o, float origin, float scale) {
  if (axis == 0) {
    if (scale == 1) return (int)floorf(origin + 0.5f) + o;
    return (int)floorf(origin + (o + 0.5f) * scale);
  }
  float sy = origin + 0.5f * scale;
  for (int y = 0; y < o; y++) sy += scale;
  return (int)floorf(sy);
}
#endif  /* DALI_AMD_RESAMPLE_FILTERS_H_ */
