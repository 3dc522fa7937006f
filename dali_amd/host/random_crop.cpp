// Host-side random machinery of the hot path, bit-compatible with the reference:
//   Philox4x32_10                include/dali/core/random/philox.h:27-160, dali/core/random/philox.cc:27-108
//   RandomCropGenerator          dali/operators/image/crop/random_crop_generator_util.cc:22-105
//   per-sample stream derivation dali/operators/random/rng_base.h:95-140,
//                                dali/operators/image/crop/random_crop_attr.h:41-96
//   bernoulli_dist (coin_flip)   dali/operators/random/random_dist.h:293-312
//   CMN argument preparation     dali/operators/image/crop/crop_mirror_normalize.h:120-149
//   CropAttr::CalculateAnchor    dali/operators/image/crop/crop_attr.cc:224-240
// The crop generator deliberately draws through libstdc++'s std::uniform_real_distribution<float>
// and std::uniform_int_distribution<int> over a 32-bit URBG, exactly like the reference, so the
// sequence matches a reference build against the same libstdc++.  These decisions stay on the
// host (they are a few hundred integer ops per sample); only the resulting windows go to the GPU.
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include "dali_amd_host.h"
#include "host_common.h"

namespace daliamd_host {

class Philox4x32_10 {
 public:
  using result_type = uint32_t;
  static constexpr uint32_t min() { return 0; }
  static constexpr uint32_t max() { return 0xffffffffu; }

  explicit Philox4x32_10(const daliamdPhiloxState &s) : st_(s) { st_.phase &= 3; Recalc(); }
  uint32_t operator()() {
    uint32_t r = out_[st_.phase++];
    if (st_.phase >= 4) {
      st_.phase = 0;
      if (++st_.ctr[0] == 0) st_.ctr[1]++;
      Recalc();
    }
    return r;
  }
  const daliamdPhiloxState &state() const { return st_; }

 private:
  static inline void Round(uint32_t c[4], uint32_t kx, uint32_t ky) {
    uint64_t m0 = 0xD2511F53ull * c[0], m1 = 0xCD9E8D57ull * c[2];
    uint32_t n0 = (uint32_t)(m1 >> 32) ^ c[1] ^ kx, n1 = (uint32_t)m1;
    uint32_t n2 = (uint32_t)(m0 >> 32) ^ c[3] ^ ky, n3 = (uint32_t)m0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  void Recalc() {
    uint32_t c[4] = {(uint32_t)st_.ctr[0], (uint32_t)(st_.ctr[0] >> 32), (uint32_t)st_.ctr[1],
                     (uint32_t)(st_.ctr[1] >> 32)};
    uint32_t kx = (uint32_t)st_.key, ky = (uint32_t)(st_.key >> 32);
    for (int r = 0; r < 10; r++, kx += 0x9E3779B9u, ky += 0xBB67AE85u) Round(c, kx, ky);
    memcpy(out_, c, sizeof(out_));
  }
  daliamdPhiloxState st_;
  uint32_t out_[4];
};

constexpr uint64_t kRandomCropSeedModifier = 0x12345678abcdefeull;
constexpr uint64_t kSkipaheadPerSample = 65537;

struct CropWindow { int anchor[2], shape[2]; };

static CropWindow GenerateCropWindow(Philox4x32_10 &gen, int H, int W, float ar_lo, float ar_hi, float area_lo,
                                     float area_hi, int num_attempts) {
  CropWindow crop = {{0, 0}, {0, 0}};
  if (W <= 0 || H <= 0) return crop;
  std::uniform_real_distribution<float> aspect_ratio_log_dis(std::log(ar_lo), std::log(ar_hi));
  std::uniform_real_distribution<float> area_dis(area_lo, area_hi);

  float min_wh_ratio = ar_lo, max_wh_ratio = ar_hi, max_hw_ratio = 1 / ar_lo;
  float min_area = W * H * area_dis.a();
  int maxW = std::max<int>(1, H * max_wh_ratio);
  int maxH = std::max<int>(1, W * max_hw_ratio);

  if (H * maxW < min_area) {
    crop.shape[0] = H; crop.shape[1] = maxW;
  } else if (W * maxH < min_area) {
    crop.shape[0] = maxH; crop.shape[1] = W;
  } else {
    int attempts_left = num_attempts;
    for (; attempts_left > 0; attempts_left--) {
      float scale = area_dis(gen);
      size_t original_area = H * W;
      float target_area = scale * original_area;
      float ratio = std::exp(aspect_ratio_log_dis(gen));
      int w = static_cast<int>(std::roundf(sqrtf(target_area * ratio)));
      int h = static_cast<int>(std::roundf(sqrtf(target_area / ratio)));
      if (w < 1) w = 1;
      if (h < 1) h = 1;
      crop.shape[0] = h; crop.shape[1] = w;
      ratio = static_cast<float>(w) / h;
      if (w <= W && h <= H && ratio >= min_wh_ratio && ratio <= max_wh_ratio) break;
    }
    if (attempts_left <= 0) {
      float max_area = area_dis.b() * W * H;
      float ratio = static_cast<float>(W) / H;
      if (ratio > max_wh_ratio) { crop.shape[0] = H; crop.shape[1] = maxW; }
      else if (ratio < min_wh_ratio) { crop.shape[0] = maxH; crop.shape[1] = W; }
      else { crop.shape[0] = H; crop.shape[1] = W; }
      float scale = std::min(1.0f, max_area / (crop.shape[0] * crop.shape[1]));
      crop.shape[0] = std::max<int>(1, crop.shape[0] * std::sqrt(scale));
      crop.shape[1] = std::max<int>(1, crop.shape[1] * std::sqrt(scale));
    }
  }
  crop.anchor[0] = std::uniform_int_distribution<int>(0, H - crop.shape[0])(gen);
  crop.anchor[1] = std::uniform_int_distribution<int>(0, W - crop.shape[1])(gen);
  return crop;
}

}  // namespace daliamd_host

extern "C" {

using namespace daliamd_host;

int daliamdRandomCropBatch(const daliamdPhiloxState *master, int batch, const int32_t *shapes_hw, float aspect_lo,
                           float aspect_hi, float area_lo, float area_hi, int num_attempts, int32_t *anchors_yx,
                           int32_t *crops_hw) {
  if (!master || !shapes_hw || !anchors_yx || !crops_hw || batch < 0)
    return Fail("daliamdRandomCropBatch: invalid argument");
  if (!(aspect_lo <= aspect_hi) || !(aspect_lo > 0)) return Fail("Provided empty range (random_aspect_ratio)");
  if (!(area_lo <= area_hi)) return Fail("Provided empty range (random_area)");
  for (int i = 0; i < batch; i++) {
    daliamdPhiloxState s = *master;
    s.ctr[1] += (uint64_t)i * kSkipaheadPerSample;  // GetSampleRNG: skipahead_sequence(i * 65537)
    s.key ^= kRandomCropSeedModifier;               // OnLoadRandomState
    Philox4x32_10 gen(s);
    CropWindow w = GenerateCropWindow(gen, shapes_hw[2 * i], shapes_hw[2 * i + 1], aspect_lo, aspect_hi, area_lo,
                                      area_hi, num_attempts);
    anchors_yx[2 * i] = w.anchor[0]; anchors_yx[2 * i + 1] = w.anchor[1];
    crops_hw[2 * i] = w.shape[0]; crops_hw[2 * i + 1] = w.shape[1];
  }
  return 0;
}

int daliamdCoinFlipBatch(const daliamdPhiloxState *master, int batch, const float *probability,
                         int probability_stride, int32_t *out) {
  if (!master || !probability || !out || batch < 0) return Fail("daliamdCoinFlipBatch: invalid argument");
  for (int i = 0; i < batch; i++) {
    float p = probability[(size_t)i * probability_stride];
    float th = p * 0x1p32f;
    uint32_t threshold = th >= 0x1p32f ? 0xffffffffu : static_cast<uint32_t>(th);
    daliamdPhiloxState s = *master;
    s.ctr[1] += (uint64_t)i * kSkipaheadPerSample;
    Philox4x32_10 gen(s);  // element 0 of the sample: no per-element skipahead
    out[i] = gen() <= threshold ? 1 : 0;
  }
  return 0;
}

void daliamdPhiloxAdvanceSequence(daliamdPhiloxState *state, uint64_t n) { state->ctr[1] += n; }

void daliamdPhiloxGenerate(daliamdPhiloxState *state, uint32_t *out, int n) {
  Philox4x32_10 gen(*state);
  for (int i = 0; i < n; i++) out[i] = gen();
  *state = gen.state();
}

int daliamdPhiloxStateToString(const daliamdPhiloxState *state, char *buf, int buf_len) {
  if (!state || !buf) return Fail("daliamdPhiloxStateToString: NULL argument");
  int n = snprintf(buf, buf_len, "Philox_%016" PRIX64 "_%016" PRIX64 ":%016" PRIX64 "_%X", state->key,
                   state->ctr[1], state->ctr[0], (unsigned)state->phase);
  return n > 0 && n < buf_len ? 0 : Fail("daliamdPhiloxStateToString: buffer too small");
}

int daliamdPhiloxStateFromString(daliamdPhiloxState *state, const char *str) {
  if (!state || !str) return Fail("daliamdPhiloxStateFromString: NULL argument");
  uint64_t key, hi, lo;
  unsigned phase;
  if (sscanf(str, "Philox_%16" SCNx64 "_%16" SCNx64 ":%16" SCNx64 "_%1X", &key, &hi, &lo, &phase) != 4 || phase > 3)
    return Fail("Malformed Philox state string: %s", str);
  state->key = key; state->ctr[1] = hi; state->ctr[0] = lo; state->phase = (int)phase;
  return 0;
}

int daliamdCmnNormArgs(const float *mean, int nmean, const float *stddev, int nstd, float scale, float shift,
                       float *mean_out, float *inv_std_out) {
  if (!mean || !stddev || !mean_out || !inv_std_out || nmean < 1 || nstd < 1) {
    Fail("daliamdCmnNormArgs: invalid argument");
    return -1;
  }
  if (!(nmean == nstd || nmean == 1 || nstd == 1)) {
    Fail("``mean`` and ``std`` must either be of the same size, be scalars, or one of them can be a vector and "
         "the other a scalar.");
    return -1;
  }
  int n = std::max(nmean, nstd);
  bool identity = true;
  for (int d = 0; d < n; d++) {
    double mean_val = mean[d % nmean];
    double std_val = stddev[d % nstd];
    mean_out[d] = (float)std::fma(-(double)shift, std_val / (double)scale, mean_val);
    inv_std_out[d] = (float)((double)scale / std_val);
    identity = identity && mean_out[d] == 0.0f && inv_std_out[d] == 1.0f;
  }
  return identity ? 0 : n;
}

int64_t daliamdCropAnchor(float anchor_norm, int64_t crop, int64_t in, int round_half_away) {
  double anchor_f = static_cast<double>(anchor_norm) * (in - crop);
  return round_half_away ? (int64_t)std::round(anchor_f) : (int64_t)anchor_f;
}

}  // extern "C"
