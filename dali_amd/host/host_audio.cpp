// Host kernels of the audio feature operators (configs[3]) for the CPU backend: spectrogram, mel filter bank,
// decibels, DCT (MFCC).  Product code - the arithmetic of the reference's CPU kernels restated:
//   window extraction, centring, reflect-101 padding   dali/kernels/signal/window/extract_windows_cpu.cc:96-145
//   window centred inside nfft, power / magnitude        dali/kernels/signal/fft/fft_cpu_impl_ffts.cc:105-111,
//                                                        dali/operators/signal/fft/spectrogram.cc:128-146
//   mel filter bank, frequency-major                     dali/kernels/audio/mel_scale/mel_filter_bank_cpu.cc:77-111
//   decibels                                             dali/kernels/signal/decibel/decibel_calculator.h:25-57
//   DCT + liftering                                      dali/kernels/signal/dct/dct_cpu.cc:75-110, mfcc.cc:41-60
// The reference's FFT is the un-vendored FFTS library; this one is a radix-2 transform in double precision (the
// comparison with the reference is tolerance-based there as well: test_spectrogram.py:188).
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

// in-place radix-2 decimation-in-time FFT, n a power of two; tw[k] = exp(-2 pi i k / n), k < n / 2
void Fft(std::complex<double> *a, int n, const std::complex<double> *tw) {
  for (int i = 1, j = 0; i < n; i++) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < half; k++) {
        const std::complex<double> u = a[i + k], v = a[i + k + half] * tw[k * step];
        a[i + k] = u + v;
        a[i + k + half] = u - v;
      }
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
  std::vector<std::complex<double>> tw(nfft / 2), buf(nfft);
  for (int k = 0; k < nfft / 2; k++) tw[k] = std::polar(1.0, -2.0 * M_PI * k / nfft);
  const int nbins = nfft / 2 + 1, shift = (nfft - wl) / 2;
  const int64_t center = p->center_windows ? wl / 2 : 0;
  for (int64_t t = 0; t < num_windows; t++) {
    const int64_t start = t * p->window_step - center;
    std::fill(buf.begin(), buf.end(), std::complex<double>(0.0, 0.0));
    for (int i = 0; i < wl; i++) {
      const int64_t idx = start + i;
      float v;   // the product is formed in float, like the reference's window extraction
      if (idx >= 0 && idx < length) v = in[idx] * window[i];
      else if (p->reflect_padding) v = in[Reflect101(idx, length)] * window[i];
      else v = 0.0f;
      buf[shift + i] = std::complex<double>((double)v, 0.0);
    }
    Fft(buf.data(), nfft, tw.data());
    for (int k = 0; k < nbins; k++) {
      double pw = buf[k].real() * buf[k].real() + buf[k].imag() * buf[k].imag();
      if (p->power == 1) pw = std::sqrt(pw);
      out[(int64_t)k * num_windows + t] = (float)pw;
    }
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
