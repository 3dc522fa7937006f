// fn.normalize on the host (CPU backend): the arithmetic of the device kernels (csrc/normalize.hip), i.e. of
// dali/operators/math/normalize/normalize.cc:209-296 and normalize_utils.h:133-220 - sums in double, the mean rounded to
// float before the squared differences are taken, inv_std = scale / sqrt(sum / (N - ddof) + epsilon) with zero kept zero,
// out = (x - mean) * inv_std + shift.  A sample is [outer][reduced][inner]; the samples of one call share the statistics
// (one sample, or the whole batch for batch=True).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "dali_amd_host.h"
#include "host_common.h"

using daliamd_host::Fail;

namespace {
inline float LoadF(const void *p, int dtype, int64_t i) {
  return dtype == DALIAMD_UINT8 ? (float)static_cast<const uint8_t *>(p)[i] : static_cast<const float *>(p)[i];
}
inline float SatRound(float v, float lo, float hi) { return std::fmin(std::fmax(std::nearbyint(v), lo), hi); }
}  // namespace

extern "C" int daliamdNormalizeHost(const daliamdNormalizeHostSample *samples, int n, int in_dtype, int out_dtype, int has_mean,
                                    float scalar_mean, int has_stddev, float scalar_inv_std, int ddof, float epsilon, float scale,
                                    float shift) {
  if (!samples || n < 1) return Fail("daliamdNormalizeHost: invalid argument");
  if (in_dtype != DALIAMD_UINT8 && in_dtype != DALIAMD_FLOAT) return Fail("daliamdNormalizeHost: the input must be uint8 or float");
  if (out_dtype != DALIAMD_FLOAT && out_dtype != DALIAMD_UINT8 && out_dtype != DALIAMD_INT8)
    return Fail("daliamdNormalizeHost: the output must be float, uint8 or int8");
  const int64_t outer = samples[0].outer, inner = samples[0].inner, bins = outer * inner;
  double count = 0;
  for (int s = 0; s < n; s++) {
    const auto &d = samples[s];
    if (!d.in || !d.out || d.outer < 1 || d.reduced < 1 || d.inner < 1) return Fail("daliamdNormalizeHost: sample %d: invalid shape", s);
    if (d.outer != outer || d.inner != inner)
      return Fail("Batch normalization requires that non-reduced dimensions have equal extent in all samples in the batch");
    count += (double)d.reduced;
  }
  std::vector<float> mean((size_t)bins, scalar_mean), inv_std((size_t)bins, scalar_inv_std);
  std::vector<double> acc;
  if (!has_mean) {
    acc.assign((size_t)bins, 0.0);
    for (int s = 0; s < n; s++) {
      const auto &d = samples[s];
      for (int64_t o = 0; o < outer; o++)
        for (int64_t r = 0; r < d.reduced; r++) {
          const int64_t base = (o * d.reduced + r) * inner;
          double *a = acc.data() + o * inner;
          for (int64_t i = 0; i < inner; i++) a[i] += (double)LoadF(d.in, in_dtype, base + i);
        }
    }
    for (int64_t p = 0; p < bins; p++) mean[p] = (float)(acc[p] * (count > 0 ? 1.0 / count : 0.0));
  }
  if (!has_stddev) {
    acc.assign((size_t)bins, 0.0);
    for (int s = 0; s < n; s++) {
      const auto &d = samples[s];
      for (int64_t o = 0; o < outer; o++)
        for (int64_t r = 0; r < d.reduced; r++) {
          const int64_t base = (o * d.reduced + r) * inner;
          for (int64_t i = 0; i < inner; i++) {
            const float dx = LoadF(d.in, in_dtype, base + i) - mean[o * inner + i];
            acc[o * inner + i] += (double)dx * (double)dx;
          }
        }
    }
    float rdiv = 0, mul = scale;   // FoldStdDev / SumSquare2InvStdDev
    if (count > ddof) rdiv = (float)(1.0 / (count - ddof));
    else if (epsilon == 0) { rdiv = 1; mul = 0; }
    for (int64_t p = 0; p < bins; p++) {
      const float x = (float)acc[p] * rdiv + epsilon;
      inv_std[p] = x != 0 ? mul / std::sqrt(x) : 0.0f;
    }
  }
  for (int s = 0; s < n; s++) {
    const auto &d = samples[s];
    for (int64_t o = 0; o < outer; o++)
      for (int64_t r = 0; r < d.reduced; r++) {
        const int64_t base = (o * d.reduced + r) * inner;
        for (int64_t i = 0; i < inner; i++) {
          const int64_t p = o * inner + i;
          const float v = (LoadF(d.in, in_dtype, base + i) - mean[p]) * inv_std[p] + shift;
          if (out_dtype == DALIAMD_FLOAT) static_cast<float *>(d.out)[base + i] = v;
          else if (out_dtype == DALIAMD_UINT8) static_cast<uint8_t *>(d.out)[base + i] = (uint8_t)SatRound(v, 0.0f, 255.0f);
          else static_cast<int8_t *>(d.out)[base + i] = (int8_t)SatRound(v, -128.0f, 127.0f);
        }
      }
  }
  return 0;
}
