// Host kernels of the audio feature operators (configs[3]) for the CPU backend: spectrogram, mel filter bank,
// decibels, DCT (MFCC).  Product code - the arithmetic of the reference's CPU kernels restated:
//   window extraction, centring, reflect-101 padding   dali/kernels/signal/window/extract_windows_cpu.cc:96-145
//   window centred inside nfft, power / magnitude        dali/kernels/signal/fft/fft_cpu_impl_ffts.cc:105-111,
//                                                        dali/operators/signal/fft/spectrogram.cc:128-146
//   mel filter bank, frequency-major                     dali/kernels/audio/mel_scale/mel_filter_bank_cpu.cc:77-111
//   decibels                                             dali/kernels/signal/decibel/decibel_calculator.h:25-57
//   DCT + liftering                                      dali/kernels/signal/dct/dct_cpu.cc:75-110, mfcc.cc:41-60
// The reference's FFT is the un-vendored FFTS library; this one is a radix-2 transform in double precision on the
// frame packed as nfft / 2 complex points (the comparison with the reference is tolerance-based there as well:
// test_spectrogram.py:188).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

#include "dali_amd_host.h"
#include "host_common.h"

using daliamd_host::Fail;

namespace {

int Reflect101(int64_t idx, int64_t size) {
  if (size < 2) return (int)(size - 1);
  while (idx < 0 || idx >= size) {
    if (idx < 0) idx = -idx;
    if (idx >= size) idx = 2 * size - 2 - idx;
  }
  return (int)idx;
}

// The transform of a real frame of nfft samples through one complex transform of n = nfft / 2 points on z[j] =
// x[2j] + i x[2j + 1] and the usual split of its result (the device kernel does the same): radix-2, decimation in time,
// real and imaginary parts in separate arrays so that the butterflies of a stage are plain loops over k, in double.
struct RealFftPlan {
  int n = 0;                         // complex points
  std::vector<int> rev;              // bit reversal
  std::vector<double> twr, twi;      // stage with half-length h: exp(-2 pi i k / (2h)), k < h, at offset h - 1
  std::vector<double> pr, pi;        // exp(-2 pi i k / nfft), k <= n / 2
  explicit RealFftPlan(int nfft) : n(nfft / 2), rev(n), twr(n > 1 ? n - 1 : 0), twi(n > 1 ? n - 1 : 0), pr(n / 2 + 1), pi(n / 2 + 1) {
    for (int i = 0, j = 0; i < n; i++) {
      rev[i] = j;
      int bit = n >> 1;
      for (; bit && (j & bit); bit >>= 1) j ^= bit;
      j ^= bit;
    }
    for (int h = 1; h < n; h <<= 1)
      for (int k = 0; k < h; k++) {
        const double a = -M_PI * k / h;
        twr[h - 1 + k] = std::cos(a);
        twi[h - 1 + k] = std::sin(a);
      }
    for (int k = 0; k <= n / 2; k++) {
      const double a = -2.0 * M_PI * k / nfft;
      pr[k] = std::cos(a);
      pi[k] = std::sin(a);
    }
  }
};

__attribute__((target_clones("avx2", "default")))
void FftStage(double *re, double *im, int n, int h, const double *wr, const double *wi) {
  for (int i = 0; i < n; i += 2 * h) {
    double *ar = re + i, *ai = im + i, *br = re + i + h, *bi = im + i + h;
    for (int k = 0; k < h; k++) {
      const double vr = br[k] * wr[k] - bi[k] * wi[k], vi = br[k] * wi[k] + bi[k] * wr[k];
      br[k] = ar[k] - vr;
      bi[k] = ai[k] - vi;
      ar[k] += vr;
      ai[k] += vi;
    }
  }
}

// re / im: the n points in bit-reversed order on entry; pw[k], k <= n: |X[k]|^2 of the real frame
void RealFftPower(const RealFftPlan &pl, double *re, double *im, double *pw) {
  const int n = pl.n;
  for (int h = 1; h < n; h <<= 1) FftStage(re, im, n, h, pl.twr.data() + h - 1, pl.twi.data() + h - 1);
  for (int k = 0; k <= n / 2; k++) {
    const int m = (n - k) & (n - 1);
    const double er = 0.5 * (re[k] + re[m]), ei = 0.5 * (im[k] - im[m]);
    const double orr = 0.5 * (im[k] + im[m]), oi = -0.5 * (re[k] - re[m]);
    const double tr = pl.pr[k] * orr - pl.pi[k] * oi, ti = pl.pr[k] * oi + pl.pi[k] * orr;
    const double ar = er + tr, ai = ei + ti, br = er - tr, bi = ei - ti;
    pw[k] = ar * ar + ai * ai;
    pw[n - k] = br * br + bi * bi;
  }
}

}  // namespace

extern "C" int daliamdSpectrogramHost(const float *in, int64_t length, const daliamdSpectrogramParams *p, const float *window,
                                      int64_t num_windows, float *out) {
  if (!in || !p || !window || !out || length <= 0 || num_windows < 0) return Fail("daliamdSpectrogramHost: invalid argument");
  const int nfft = p->nfft, wl = p->window_length;
  if (nfft < 2 || (nfft & (nfft - 1)) || wl <= 0 || wl > nfft || p->window_step <= 0 || (p->power != 1 && p->power != 2))
    return Fail("daliamdSpectrogramHost: unsupported parameters (nfft %d, window %d, step %d, power %d)", nfft, wl, p->window_step,
                p->power);
  const RealFftPlan plan(nfft);
  const int n = plan.n, nbins = n + 1, shift = (nfft - wl) / 2;
  const int64_t center = p->center_windows ? wl / 2 : 0;
  constexpr int kBlock = 16;   // frames per block: a bin's values of a block are one contiguous run of the [bin][frame] output
  std::vector<float> frame(nfft), blk((size_t)nbins * kBlock);
  std::vector<double> re(n), im(n), pw(nbins);
  for (int64_t t0 = 0; t0 < num_windows; t0 += kBlock) {
    const int cnt = (int)std::min<int64_t>(kBlock, num_windows - t0);
    for (int f = 0; f < cnt; f++) {
      const int64_t start = (t0 + f) * p->window_step - center;
      std::fill(frame.begin(), frame.end(), 0.0f);
      float *dst = frame.data() + shift;   // the product is formed in float, like the reference's window extraction
      if (start >= 0 && start + wl <= length) {
        for (int i = 0; i < wl; i++) dst[i] = in[start + i] * window[i];
      } else {
        for (int i = 0; i < wl; i++) {
          const int64_t idx = start + i;
          if (idx >= 0 && idx < length) dst[i] = in[idx] * window[i];
          else if (p->reflect_padding) dst[i] = in[Reflect101(idx, length)] * window[i];
        }
      }
      if (n == 1) {   // nfft == 2
        pw[0] = ((double)frame[0] + frame[1]) * ((double)frame[0] + frame[1]);
        pw[1] = ((double)frame[0] - frame[1]) * ((double)frame[0] - frame[1]);
      } else {
        for (int j = 0; j < n; j++) {
          re[plan.rev[j]] = (double)frame[2 * j];
          im[plan.rev[j]] = (double)frame[2 * j + 1];
        }
        RealFftPower(plan, re.data(), im.data(), pw.data());
      }
      for (int k = 0; k < nbins; k++) blk[(size_t)k * kBlock + f] = (float)(p->power == 1 ? std::sqrt(pw[k]) : pw[k]);
    }
    for (int k = 0; k < nbins; k++) std::copy_n(&blk[(size_t)k * kBlock], cnt, out + (int64_t)k * num_windows + t0);
  }
  return 0;
}

extern "C" int daliamdMelFilterBankHost(const float *spec, int nbins, int64_t frames, const float *weights, int nfilter, float *out) {
  if (!spec || !weights || !out || nbins <= 0 || frames < 0 || nfilter <= 0) return Fail("daliamdMelFilterBankHost: invalid argument");
  // every filter walks its own bins in increasing order: multiply, then add, in float (ComputeFreqMajor)
  for (int m = 0; m < nfilter; m++) {
    float *o = out + (int64_t)m * frames;
    std::fill(o, o + frames, 0.0f);
    const float *w = weights + (int64_t)m * nbins;
    for (int b = 0; b < nbins; b++) {
      if (w[b] == 0.0f) continue;
      const float wb = w[b];
      const float *s = spec + (int64_t)b * frames;
      for (int64_t t = 0; t < frames; t++) o[t] += wb * s[t];
    }
  }
  return 0;
}

extern "C" int daliamdToDecibelsHost(const float *in, int64_t size, float multiplier, float reference, float cutoff_db, float *out) {
  if (size < 0 || (size > 0 && (!in || !out))) return Fail("daliamdToDecibelsHost: invalid argument");
  float min_ratio = (float)std::pow(10.0, (double)cutoff_db / (double)multiplier);
  if (min_ratio == 0.0f) min_ratio = std::nextafter(0.0f, 1.0f);
  float s_ref = reference;
  if (!(reference > 0.0f)) {   // not given: the sample's maximum (1 when that is 0 or the sample is empty)
    s_ref = size ? *std::max_element(in, in + size) : 1.0f;
    if (s_ref == 0.0f) s_ref = 1.0f;
  }
  const float inv = s_ref == 1.0f ? 1.0f : 1.0f / s_ref;
  const float mul_log2 = multiplier * 0.3010299956639812f;
  for (int64_t i = 0; i < size; i++) out[i] = mul_log2 * std::log2(std::max(min_ratio, in[i] * inv));
  return 0;
}

extern "C" int daliamdDctHost(const float *in, int n_in, int64_t inner, const float *table, const float *lifter, int ndct, float *out) {
  if (!in || !table || !out || n_in <= 0 || inner < 0 || ndct <= 0) return Fail("daliamdDctHost: invalid argument");
  std::vector<double> acc((size_t)inner);
  for (int k = 0; k < ndct; k++) {
    std::fill(acc.begin(), acc.end(), 0.0);
    for (int n = 0; n < n_in; n++) {
      const double c = table[(int64_t)k * n_in + n];
      const float *s = in + (int64_t)n * inner;
      for (int64_t t = 0; t < inner; t++) acc[t] += c * (double)s[t];
    }
    const double l = lifter ? (double)lifter[k] : 1.0;
    float *o = out + (int64_t)k * inner;
    for (int64_t t = 0; t < inner; t++) o[t] = (float)(l * acc[t]);
  }
  return 0;
}
