/*
 * ORACLE (test infrastructure only -- never linked or imported by the product path).
 *
 * CPU restatement of the reference's host-side random machinery for the
 * JPEG -> RandomResizedCrop -> CropMirrorNormalize path:
 *
 *   Philox4x32-10              include/dali/core/random/philox.h:27-160
 *                              dali/core/random/philox.cc:27-85
 *   per-sample RNG derivation  dali/operators/random/rng_base.h:46,55,105-140
 *                              dali/operators/image/crop/random_crop_attr.h:41-96
 *   RandomCropGenerator        dali/operators/image/crop/random_crop_generator_util.cc:36-105
 *   coin_flip / bernoulli      dali/operators/random/random_dist.h:293-312,
 *                              dali/operators/random/rng_base_cpu.h:50-60
 *   CropAttr::CalculateAnchor  dali/operators/image/crop/crop_attr.cc:224-240
 *
 * The reference draws through libstdc++'s std::uniform_real_distribution<float>
 * and std::uniform_int_distribution<int> over a 32-bit URBG.  Those are restated
 * here in plain C from the published libstdc++ (GCC 11) algorithms
 * (bits/random.tcc generate_canonical; bits/uniform_int_dist.h Lemire "_S_nd"),
 * so that this file is independent of the product host code, which uses the
 * std:: distributions directly.
 *
 * Pinning: Philox is checked against the Random123 known-answer vectors
 * (tests/test_oracle_rng.py).  The crop sequence itself has no golden vectors in the
 * reference (its tests check properties only, test_random_resized_crop.py:29-90).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define PHILOX_W32_0 0x9E3779B9u
#define PHILOX_W32_1 0xBB67AE85u
#define PHILOX_M4x32_0 0xD2511F53u
#define PHILOX_M4x32_1 0xCD9E8D57u

typedef struct {
  uint64_t key;
  uint64_t ctr[2]; /* ctr[0] = low (offset>>2), ctr[1] = high (sequence) */
  int phase;
  uint32_t out[4];
} orc_philox;

static void philox_round(uint32_t *x, uint32_t *y, uint32_t *z, uint32_t *w, uint32_t kx,
                         uint32_t ky) {
  uint64_t m0 = (uint64_t)PHILOX_M4x32_0 * *x;
  uint64_t m1 = (uint64_t)PHILOX_M4x32_1 * *z;
  uint32_t lo0 = (uint32_t)m0, hi0 = (uint32_t)(m0 >> 32);
  uint32_t lo1 = (uint32_t)m1, hi1 = (uint32_t)(m1 >> 32);
  uint32_t nx = hi1 ^ *y ^ kx, ny = lo1, nz = hi0 ^ *w ^ ky, nw = lo0;
  *x = nx; *y = ny; *z = nz; *w = nw;
}

/* philox.cc:48-85 */
static void philox_recalc(orc_philox *p) {
  uint32_t x = (uint32_t)p->ctr[0], y = (uint32_t)(p->ctr[0] >> 32);
  uint32_t z = (uint32_t)p->ctr[1], w = (uint32_t)(p->ctr[1] >> 32);
  uint32_t kx = (uint32_t)p->key, ky = (uint32_t)(p->key >> 32);
  for (int r = 0; r < 10; r++) {
    if (r) { kx += PHILOX_W32_0; ky += PHILOX_W32_1; }
    philox_round(&x, &y, &z, &w, kx, ky);
  }
  p->out[0] = x; p->out[1] = y; p->out[2] = z; p->out[3] = w;
}

void orc_philox_init(orc_philox *p, uint64_t key, uint64_t ctr_hi, uint64_t ctr_lo, int phase) {
  p->key = key; p->ctr[0] = ctr_lo; p->ctr[1] = ctr_hi; p->phase = phase & 3;
  philox_recalc(p);
}

/* philox.h:81-89 */
uint32_t orc_philox_next(orc_philox *p) {
  uint32_t ret = p->out[p->phase++];
  if (p->phase >= 4) {
    p->phase = 0;
    p->ctr[0] += 1;
    if (p->ctr[0] < 1) p->ctr[1]++;
    philox_recalc(p);
  }
  return ret;
}

/* philox.h:127-146 (skipahead) */
void orc_philox_skipahead(orc_philox *p, uint64_t n) {
  p->phase += (int)(n & 3);
  n >>= 2;
  if (p->phase > 3) { n++; p->phase -= 4; }
  if (n) {
    p->ctr[0] += n;
    if (p->ctr[0] < n) p->ctr[1]++;
    philox_recalc(p);
  }
}

void orc_philox_skipahead_sequence(orc_philox *p, uint64_t n) {
  p->ctr[1] += n;
  if (n) philox_recalc(p);
}

/* Raw block function for known-answer tests: ctr = {x,y,z,w}, key = {kx,ky}. */
void orc_philox_block(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc_philox p;
  p.key = ((uint64_t)key[1] << 32) | key[0];
  p.ctr[0] = ((uint64_t)ctr[1] << 32) | ctr[0];
  p.ctr[1] = ((uint64_t)ctr[3] << 32) | ctr[2];
  p.phase = 0;
  philox_recalc(&p);
  memcpy(out, p.out, sizeof(p.out));
}

/* ---- libstdc++ distributions over a 32-bit URBG, restated ---- */

/* generate_canonical<float, 24>(urbg): b = 24, r = 2^32, m = 1
 * (bits/random.tcc:3348-3380); float arithmetic throughout. */
static float canonical_f32(orc_philox *g) {
  float sum = (float)orc_philox_next(g) * 1.0f;
  float tmp = 1.0f * 4294967296.0f;
  float ret = sum / tmp;
  if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
  return ret;
}

/* uniform_real_distribution<float>(a,b)(g): (canonical * (b - a)) + a, random.h:1870 */
static float uniform_real_f32(orc_philox *g, float a, float b) {
  return (canonical_f32(g) * (b - a)) + a;
}

/* uniform_int_distribution<int>(a,b)(g), 32-bit URBG => Lemire _S_nd<uint64_t>,
 * bits/uniform_int_dist.h:241-270,300-312 */
static int uniform_int_i32(orc_philox *g, int a, int b) {
  uint32_t urange = (uint32_t)b - (uint32_t)a;
  uint32_t ret;
  if (urange == 0xffffffffu) {
    ret = orc_philox_next(g);
  } else {
    uint32_t range = urange + 1;
    uint64_t product = (uint64_t)orc_philox_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      uint32_t threshold = (uint32_t)(-range) % range;
      while (low < threshold) {
        product = (uint64_t)orc_philox_next(g) * (uint64_t)range;
        low = (uint32_t)product;
      }
    }
    ret = (uint32_t)(product >> 32);
  }
  return (int)(ret + (uint32_t)a);
}

/* ---- RandomCropGenerator::GenerateCropWindowImpl, random_crop_generator_util.cc:36-101 ---- */
void orc_random_crop(orc_philox *g, int H, int W, float ar_lo, float ar_hi, float area_lo,
                     float area_hi, int num_attempts, int out_anchor_yx[2], int out_shape_hw[2]) {
  int ch = 0, cw = 0;
  out_anchor_yx[0] = out_anchor_yx[1] = 0;
  out_shape_hw[0] = out_shape_hw[1] = 0;
  if (W <= 0 || H <= 0) return;
  float log_lo = logf(ar_lo), log_hi = logf(ar_hi); /* std::log(float) ctor args */

  float min_wh_ratio = ar_lo;
  float max_wh_ratio = ar_hi;
  float max_hw_ratio = 1 / ar_lo;
  float min_area = W * H * area_lo;
  int t1 = (int)(H * max_wh_ratio), t2 = (int)(W * max_hw_ratio);
  int maxW = t1 > 1 ? t1 : 1;
  int maxH = t2 > 1 ? t2 : 1;

  if (H * maxW < min_area) { /* image too wide */
    ch = H; cw = maxW;
  } else if (W * maxH < min_area) { /* image too tall */
    ch = maxH; cw = W;
  } else {
    int attempts_left = num_attempts;
    for (; attempts_left > 0; attempts_left--) {
      float scale = uniform_real_f32(g, area_lo, area_hi);
      size_t original_area = (size_t)(H * W);
      float target_area = scale * original_area;
      float ratio = expf(uniform_real_f32(g, log_lo, log_hi));
      int w = (int)roundf(sqrtf(target_area * ratio));
      int h = (int)roundf(sqrtf(target_area / ratio));
      if (w < 1) w = 1;
      if (h < 1) h = 1;
      ch = h; cw = w;
      ratio = (float)w / h;
      if (w <= W && h <= H && ratio >= min_wh_ratio && ratio <= max_wh_ratio) break;
    }
    if (attempts_left <= 0) {
      float max_area = area_hi * W * H;
      float ratio = (float)W / H;
      if (ratio > max_wh_ratio) { ch = H; cw = maxW; }
      else if (ratio < min_wh_ratio) { ch = maxH; cw = W; }
      else { ch = H; cw = W; }
      float s = max_area / (ch * cw);
      float scale = s < 1.0f ? s : 1.0f;
      /* std::sqrt(float) -> float; int * float -> float; then truncated by max<int> */
      int a = (int)(ch * sqrtf(scale)), b = (int)(cw * sqrtf(scale));
      ch = a > 1 ? a : 1;
      cw = b > 1 ? b : 1;
    }
  }
  out_shape_hw[0] = ch; out_shape_hw[1] = cw;
  out_anchor_yx[0] = uniform_int_i32(g, 0, H - ch);
  out_anchor_yx[1] = uniform_int_i32(g, 0, W - cw);
}

#define RANDOM_CROP_SEED_MOD 0x12345678abcdefeULL /* random_crop_attr.h:32 */
#define SKIPAHEAD_PER_SAMPLE 65537                /* rng_base.h:55 */
#define SKIPAHEAD_PER_ELEMENT 257                 /* rng_base.h:46 */

/* One RandomResizedCrop batch: iteration t, batch size B, op seed S.
 * master = Philox(S, 0, 0) advanced by B sequences per Run (rng_base.h:105,136-138);
 * sample i: master.skipahead_sequence(i*65537), key ^= modifier (random_crop_attr.h:87-95). */
void orc_rrc_batch(int64_t seed, int64_t iteration, int batch, const int *shapes_hw, float ar_lo,
                   float ar_hi, float area_lo, float area_hi, int num_attempts, int *anchors_yx,
                   int *crop_hw) {
  for (int i = 0; i < batch; i++) {
    orc_philox g;
    uint64_t ctr_hi = (uint64_t)iteration * (uint64_t)batch + (uint64_t)i * SKIPAHEAD_PER_SAMPLE;
    orc_philox_init(&g, (uint64_t)seed ^ RANDOM_CROP_SEED_MOD, ctr_hi, 0, 0);
    orc_random_crop(&g, shapes_hw[2 * i], shapes_hw[2 * i + 1], ar_lo, ar_hi, area_lo, area_hi,
                    num_attempts, anchors_yx + 2 * i, crop_hw + 2 * i);
  }
}

/* coin_flip: bernoulli_dist{p}(gen) = gen() <= uint32(p * 2^32)   (random_dist.h:296-308);
 * one element per sample => no per-element skipahead (rng_base_cpu.h:50-60). */
void orc_coin_flip_batch(int64_t seed, int64_t iteration, int batch, float probability,
                         int32_t *out) {
  float th = probability * 0x1p32f; /* float arithmetic, random_dist.h:297 */
  uint32_t threshold = th >= 0x1p32f ? 0xffffffffu : (uint32_t)th;
  for (int i = 0; i < batch; i++) {
    orc_philox g;
    uint64_t ctr_hi = (uint64_t)iteration * (uint64_t)batch + (uint64_t)i * SKIPAHEAD_PER_SAMPLE;
    orc_philox_init(&g, (uint64_t)seed, ctr_hi, 0, 0);
    out[i] = orc_philox_next(&g) <= threshold ? 1 : 0;
  }
}

/* CropAttr::CalculateAnchor, crop_attr.cc:224-240: anchor = round(norm * (in - crop)) in double */
int64_t orc_crop_anchor(double anchor_norm, int64_t crop, int64_t in, int round_mode) {
  double v = anchor_norm * (double)(in - crop);
  return round_mode ? (int64_t)round(v) : (int64_t)v;
}
