/* ORACLE (test infrastructure only; never imported by dali_amd/).
 *
 * CPU restatement of the size / region arithmetic of DALI's `fn.resize`:
 *   AdjustOutputSize        dali/operators/image/resize/resize_attr_base.cc:88-188
 *   CalculateSampleParams   dali/operators/image/resize/resize_attr_base.h:50-116
 *   CalculateInputRoI       dali/operators/image/resize/resize_attr.cc:118-158
 * for 2-D images (dimension order H, W like the `size` argument).  Pinned by the worked examples in the
 * reference's own documentation strings (resize_attr_base.cc:33-41, :81-85) in tests/test_oracle_resize.py.
 */
#include <math.h>
#include <string.h>

enum { ORC_RESIZE_DEFAULT = 0, ORC_RESIZE_STRETCH = 1, ORC_RESIZE_NOT_LARGER = 2, ORC_RESIZE_NOT_SMALLER = 3 };

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* resize_attr_base.cc:88-188 */
static void adjust_output_size(float *out_size, const float *in_size, int ndim, int mode, const float *max_size) {
  double scale[3] = {1, 1, 1};
  int mask[3] = {0, 0, 0};
  int sizes_provided = 0;
  for (int d = 0; d < ndim; d++) {
    mask[d] = (out_size[d] != 0 && in_size[d] != 0);
    scale[d] = in_size[d] ? out_size[d] / in_size[d] : 1;   /* float division, widened afterwards (as the reference) */
    sizes_provided += mask[d];
  }
  if (sizes_provided == 0) {
    for (int d = 0; d < ndim; d++) out_size[d] = in_size[d];
    return;
  }
  if (mode == ORC_RESIZE_DEFAULT || mode == ORC_RESIZE_STRETCH) {
    if (sizes_provided < ndim) {
      double avg_scale = 1;
      if (mode == ORC_RESIZE_DEFAULT) {
        for (int d = 0; d < ndim; d++)
          if (mask[d]) avg_scale *= fabs(scale[d]);
        if (sizes_provided > 1) avg_scale = pow(avg_scale, 1.0 / sizes_provided);
      }
      for (int d = 0; d < ndim; d++)
        if (!mask[d]) {
          scale[d] = avg_scale;
          out_size[d] = mode == ORC_RESIZE_DEFAULT ? (float)(in_size[d] * scale[d]) : in_size[d];
        }
    }
    if (max_size)
      for (int d = 0; d < ndim; d++)
        if (max_size[d] > 0 && fabsf(out_size[d]) > max_size[d]) {
          out_size[d] = copysignf(max_size[d], out_size[d]);
          scale[d] = out_size[d] / in_size[d];
        }
    return;
  }
  /* NotLarger / NotSmaller */
  double final_scale = 0;
  int first = 1;
  for (int d = 0; d < ndim; d++)
    if (mask[d]) {
      float s = (float)fabs(scale[d]);
      if (first || (mode == ORC_RESIZE_NOT_SMALLER && s > final_scale) || (mode == ORC_RESIZE_NOT_LARGER && s < final_scale))
        final_scale = s;
      first = 0;
    }
  if (max_size)
    for (int d = 0; d < ndim; d++)
      if (max_size[d] > 0) {
        double s = (double)max_size[d] / in_size[d];
        if (s < final_scale) final_scale = s;
      }
  for (int d = 0; d < ndim; d++)
    if (!mask[d] || fabs(scale[d]) != final_scale) {
      scale[d] = copysign(final_scale, scale[d]);
      out_size[d] = (float)(in_size[d] * scale[d]);
    }
}

/* One sample.  in_hw: image size; requested[2]: requested H, W (0 = unspecified); has_roi/roi_relative/roi[4] =
 * (start_y, start_x, end_y, end_x); max_size[2] or NULL.  Outputs: out_hw[2], src_lo[2], src_hi[2] (y, x).
 * Returns 0, or 1 for "Cannot produce non-empty output from empty input". */
int orc_resize_params(const int *in_hw, const float *requested, int mode, const float *max_size, int subpixel_scale,
                      int has_roi, int roi_relative, const float *roi, int *out_hw, float *src_lo, float *src_hi) {
  const int ndim = 2;
  float in_lo[2], in_hi[2], in_size[2], req[2] = {requested[0], requested[1]};
  for (int d = 0; d < ndim; d++) { /* CalculateInputRoI */
    if (has_roi && in_hw[d] > 0) {
      double lo = roi[d], hi = roi[2 + d];
      if (roi_relative) { lo *= in_hw[d]; hi *= in_hw[d]; }
      const float min_size = 1e-3f;
      if (fabs(hi - lo) < min_size) {
        float offset = lo <= hi ? 0.5f * min_size : -0.5f * min_size;
        lo -= offset;
        hi += offset;
      }
      in_lo[d] = (float)lo;
      in_hi[d] = (float)hi;
    } else {
      in_lo[d] = 0;
      in_hi[d] = (float)in_hw[d];
    }
  }
  for (int d = 0; d < ndim; d++) { /* CalculateSampleParams */
    float sz = in_hi[d] - in_lo[d];
    if (sz < 0) {
      float t = in_hi[d]; in_hi[d] = in_lo[d]; in_lo[d] = t;
      req[d] = -req[d];
      sz = -sz;
    }
    in_size[d] = sz;
  }
  adjust_output_size(req, in_size, ndim, mode, max_size);
  for (int d = 0; d < ndim; d++)
    if (in_lo[d] == in_hi[d] && req[d] != 0) return 1;
  const int empty_input = in_hw[0] == 0 || in_hw[1] == 0;
  const int min_size = empty_input ? 0 : 1;
  for (int d = 0; d < ndim; d++) {
    float out_sz = req[d];
    int flip = out_sz < 0;
    int dst = (int)roundf(fabsf(out_sz));
    if (dst < min_size) dst = min_size;
    out_hw[d] = dst;
    float lo = in_lo[d], hi = in_hi[d];
    if (flip) { float t = lo; lo = hi; hi = t; }
    if (subpixel_scale && (float)dst != fabsf(out_sz)) {
      double adjustment = clampd((double)dst / fabsf(out_sz), -10.0, 10.0);
      double center = 0.5 * lo + 0.5 * hi;   /* alignment 0.5: the centre of the region stays put */
      double nlo = clampd(center + (lo - center) * adjustment, -1e+9, 1e+9);
      double nhi = clampd(center + (hi - center) * adjustment, -1e+9, 1e+9);
      lo = (float)nlo;
      hi = (float)nhi;
    }
    src_lo[d] = lo;
    src_hi[d] = hi;
  }
  return 0;
}
