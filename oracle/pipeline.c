/*
 * ORACLE (test infrastructure only).
 *
 * Whole-path CPU baseline: decode -> RandomResizedCrop -> CropMirrorNormalize for a batch, one task per sample on
 * an OpenMP team -- the shape of the reference's CPU backend, where every operator fans its batch out over
 * ws.GetThreadPool() one sample per task (dali/operators/image/resize/resize_op_impl_cpu.h:84-107,
 * dali/operators/image/crop/crop_mirror_normalize.cc:116-144, dali/operators/imgcodec/image_decoder.h:724-749).
 * Used only by bench.py's `cpu_baseline` leg and by tests.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint64_t key; uint64_t ctr[2]; int phase; uint32_t out[4]; } orc_philox;
void orc_philox_init(orc_philox *p, uint64_t key, uint64_t ctr_hi, uint64_t ctr_lo, int phase);
uint32_t orc_philox_next(orc_philox *p);
void orc_random_crop(orc_philox *g, int H, int W, float ar_lo, float ar_hi, float area_lo, float area_hi,
                     int num_attempts, int out_anchor_yx[2], int out_shape_hw[2]);
int orc_jpeg_info(const uint8_t *data, size_t size, int *info);
int orc_jpeg_decode_rgb(const uint8_t *data, size_t size, uint8_t *rgb, int16_t **coef_out, uint16_t *qt_out);
int orc_resample_u8(const uint8_t *in, int H, int W, int C, int use_roi, const float *roi, int outH, int outW,
                    int min_filter, int mag_filter, int antialias, int round_mode, uint8_t *out, float *tmp_out,
                    int *info);
int orc_cmn_u8(const uint8_t *in, int H, int W, int C, int ay, int ax, int ch, int cw, int mirror, const float *mean,
               const float *inv_std, int nnorm, int layout_chw, int pad_output, int pad_oob, const float *fill_values,
               int nfill, int dtype, void *out);

/* out: [n][3][out_h][out_w] fp16 (CHW).  Returns the number of failed samples. */
int orc_pipeline_batch(const uint8_t *const *jpegs, const size_t *sizes, int n, int64_t rrc_seed, int64_t flip_seed,
                       int64_t iteration, int out_h, int out_w, const float *mean, const float *inv_std,
                       uint16_t *out, int nthreads) {
  int failed = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : failed)
  for (int i = 0; i < n; i++) {
    int info[15];
    if (orc_jpeg_info(jpegs[i], sizes[i], info)) { failed++; continue; }
    int W = info[0], H = info[1];
    uint8_t *rgb = (uint8_t *)malloc((size_t)W * H * 3);
    uint8_t *u8 = (uint8_t *)malloc((size_t)out_h * out_w * 3);
    if (orc_jpeg_decode_rgb(jpegs[i], sizes[i], rgb, NULL, NULL)) { failed++; free(rgb); free(u8); continue; }
    orc_philox g;
    uint64_t ctr_hi = (uint64_t)iteration * (uint64_t)n + (uint64_t)i * 65537ull;
    orc_philox_init(&g, (uint64_t)rrc_seed ^ 0x12345678abcdefeULL, ctr_hi, 0, 0);
    int anchor[2], crop[2];
    orc_random_crop(&g, H, W, 3.0f / 4, 4.0f / 3, 0.08f, 1.0f, 10, anchor, crop);
    float roi[4] = {(float)anchor[0], (float)anchor[1], (float)(anchor[0] + crop[0]), (float)(anchor[1] + crop[1])};
    orc_resample_u8(rgb, H, W, 3, 1, roi, out_h, out_w, 1, 1, 1, 0, u8, NULL, NULL);
    orc_philox f;
    orc_philox_init(&f, (uint64_t)flip_seed, ctr_hi, 0, 0);
    int mirror = orc_philox_next(&f) <= 0x80000000u;
    orc_cmn_u8(u8, out_h, out_w, 3, 0, 0, out_h, out_w, mirror, mean, inv_std, 3, 1, 0, 0, NULL, 0, 1,
               out + (size_t)i * 3 * out_h * out_w);
    free(rgb);
    free(u8);
  }
  return failed;
}
