// Audio feature kernels for gfx950: spectrogram (fused windowing + in-LDS FFT + power), mel filter bank as an
// f32 MFMA GEMM, and to_decibels with a per-sample max reduction.  See include/dali_amd_kernels.h for the
// reference counterparts.  f32 throughout; parity with the reference's CPU path is tolerance-based for the FFT
// (its FFTS library is not available; tests use a float64 FFT like the reference's own tests do).
#include <cmath>
#include <vector>
#include "common.h"

namespace daliamd {

// =============================================================================================
// spectrogram
// =============================================================================================
constexpr int kSpecThreads = 256;
constexpr int kFramesPerWg = kSpecThreads / 64;  // one frame per wave

__device__ __forceinline__ long long Reflect101L(long long idx, long long size) {
  if (size < 2) return size - 1;
  for (;;) {
    if (idx < 0) idx = -idx;
    else if (idx >= size) idx = 2 * size - 2 - idx;
    else break;
  }
  return idx;
}

__global__ __launch_bounds__(kSpecThreads) void SpectrogramKernel(const daliamdSpectrogramDesc *__restrict__ descs, int ndesc,
                                                                  int total_wg, daliamdSpectrogramParams p,
                                                                  const float *__restrict__ window) {
  extern __shared__ __attribute__((aligned(16))) float2 spec_lds[];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdSpectrogramDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int nfft = p.nfft, half = nfft >> 1;
  const int log2n = 31 - __clz(nfft);
  float2 *tw = spec_lds;                       // [nfft/2] twiddles exp(-2*pi*i*k/nfft)
  float2 *buf = spec_lds + half;               // [kFramesPerWg][nfft]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < half; k += kSpecThreads) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)nfft, &s, &c);
    tw[k] = make_float2(c, s);
  }
  const int t0 = (wg - d.wg_start) * kFramesPerWg;
  const int frame = t0 + wave;
  float2 *fb = buf + (size_t)wave * nfft;
  // ---- windowed frame, bit-reversed placement (decimation in time) ----
  const int pad0 = (nfft - p.window_length) / 2;  // window centred inside nfft (fft_cpu_impl_ffts.cc:108-110)
  const long long start = (long long)frame * p.window_step - (p.center_windows ? p.window_length / 2 : 0);
  for (int i = lane; i < nfft; i += 64) {
    float v = 0.0f;
    int wi = i - pad0;
    if (frame < d.num_windows && wi >= 0 && wi < p.window_length) {
      long long idx = start + wi;
      if (p.reflect_padding) v = window[wi] * d.in[Reflect101L(idx, d.length)];
      else if (idx >= 0 && idx < d.length) v = window[wi] * d.in[idx];
    }
    int j = (int)(__brev((unsigned)i) >> (32 - log2n));
    fb[j] = make_float2(v, 0.0f);
  }
  __syncthreads();
  // ---- radix-2 butterflies ----
  for (int s = 0; s < log2n; s++) {
    const int hs = 1 << s;
    const int tstride = half >> s;
    for (int b = lane; b < half; b += 64) {
      int pos = b & (hs - 1);
      int i0 = ((b >> s) << (s + 1)) + pos, i1 = i0 + hs;
      float2 w = tw[pos * tstride];
      float2 x0 = fb[i0], x1 = fb[i1];
      float2 t = make_float2(w.x * x1.x - w.y * x1.y, w.x * x1.y + w.y * x1.x);
      fb[i0] = make_float2(x0.x + t.x, x0.y + t.y);
      fb[i1] = make_float2(x0.x - t.x, x0.y - t.y);
    }
    __syncthreads();
  }
  // ---- power / magnitude, frequency-major output: 4 consecutive frames per bin ----
  const int T = d.num_windows;
  const int nf = min(kFramesPerWg, T - t0);
  for (int b = tid; b <= half; b += kSpecThreads) {
    float *o = d.out + (size_t)b * T + t0;
    for (int f = 0; f < nf; f++) {
      float2 x = buf[(size_t)f * nfft + b];
      float pw = x.x * x.x + x.y * x.y;
      o[f] = p.power == 2 ? pw : sqrtf(pw);
    }
  }
}

// =============================================================================================
// mel filter bank: out[nfilter][T] = W[nfilter][K] * S[K][T] on v_mfma_f32_16x16x4_f32
// =============================================================================================
constexpr int kMelThreads = 256;  // 4 waves, each owns 16 frames (columns)
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MT>
__global__ __launch_bounds__(kMelThreads) void MelKernel(const daliamdMelDesc *__restrict__ descs, int ndesc, int total_wg,
                                                         const float *__restrict__ W, int nfilter, int K) {
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdMelDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = d.frames;
  const int t0 = ((wg - d.wg_start) * 4 + wave) * 16;
  if (t0 >= T) return;
  const int col = t0 + (lane & 15);      // B / D column owned by this lane
  const int kq = lane >> 4;              // k offset inside a 4-deep step (A and B)
  const int arow = lane & 15;            // A row inside a 16-row tile
  floatx4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) acc[m] = (floatx4){0, 0, 0, 0};
  const bool col_ok = col < T;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int kk = k0 + kq;
    const bool k_ok = kk < K;
    float b = (k_ok && col_ok) ? d.in[(size_t)kk * T + col] : 0.0f;
#pragma unroll
    for (int m = 0; m < MT; m++) {
      int i = 16 * m + arow;
      float a = (k_ok && i < nfilter) ? W[(size_t)i * K + kk] : 0.0f;
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
    }
  }
  if (!col_ok) return;
#pragma unroll
  for (int m = 0; m < MT; m++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int row = 16 * m + (lane >> 4) * 4 + r;
      if (row < nfilter) d.out[(size_t)row * T + col] = acc[m][r];
    }
  }
}

// =============================================================================================
// to_decibels
// =============================================================================================
constexpr int kDbThreads = 1024;

__global__ __launch_bounds__(kDbThreads) void DecibelKernel(const daliamdDecibelDesc *__restrict__ descs, float mul_log2,
                                                            float reference, float min_ratio) {
  __shared__ float red[kDbThreads / 64];
  const daliamdDecibelDesc &d = descs[blockIdx.x];
  const int tid = threadIdx.x;
  float s_ref = reference;
  if (!(reference > 0.0f)) {
    float m = 0.0f;
    for (int64_t i = tid; i < d.size; i += kDbThreads) m = fmaxf(m, d.in[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));  // wave64 shuffle reduction
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid < 64) {
      float v = tid < kDbThreads / 64 ? red[tid] : 0.0f;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
      if (tid == 0) red[0] = v;
    }
    __syncthreads();
    s_ref = red[0];
    if (s_ref == 0.0f) s_ref = 1.0f;
  }
  const float inv = s_ref == 1.0f ? 1.0f : 1.0f / s_ref;
  for (int64_t i = tid; i < d.size; i += kDbThreads) d.out[i] = mul_log2 * log2f(fmaxf(min_ratio, d.in[i] * inv));
}

// host: mel scales (mel_scale.h:27-73), all in double
static double HzToMel(double hz, int formula) {
  if (formula == 1) return 1127.0 * std::log(1.0 + hz / 700.0);
  const double fsp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = (min_log_hz - 0.0) / fsp, step_log = 0.068751777;
  return hz >= min_log_hz ? min_log_mel + std::log(hz / min_log_hz) / step_log : (hz - 0.0) / fsp;
}
static double MelToHz(double mel, int formula) {
  if (formula == 1) return 700.0 * (std::exp(mel / 1127.0) - 1.0);
  const double fsp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = (min_log_hz - 0.0) / fsp, step_log = 0.068751777;
  return mel >= min_log_mel ? min_log_hz * std::exp(step_log * (mel - min_log_mel)) : 0.0 + mel * fsp;
}

}  // namespace daliamd

extern "C" {

using namespace daliamd;

void daliamdHannWindow(int n, float *window) {  // window_functions.h:25-33
  double a = (2 * M_PI / n);
  for (int t = 0; t < n; t++) window[t] = static_cast<float>(0.5 * (1.0 - std::cos(a * (t + 0.5))));
}

daliamdResult_t daliamdSpectrogramSetup(daliamdSpectrogramDesc *descs, int n, const daliamdSpectrogramParams *p, int *nwg,
                                        int *lds_bytes) {
  DALIAMD_REQUIRE(descs && p && nwg && lds_bytes && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdSpectrogramSetup: NULL argument");
  DALIAMD_REQUIRE(p->nfft >= 2 && (p->nfft & (p->nfft - 1)) == 0 && p->nfft <= 4096, DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdSpectrogramSetup: nfft must be a power of two in [2, 4096], got %d", p->nfft);
  DALIAMD_REQUIRE(p->window_length > 0 && p->window_length <= p->nfft, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "Window length (%d) can't be bigger than the FFT size (%d)", p->window_length, p->nfft);
  DALIAMD_REQUIRE(p->window_step > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "window_step must be positive");
  DALIAMD_REQUIRE(p->power == 1 || p->power == 2, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "`power` can be only 1 (energy) or 2 (power), received %d", p->power);
  int wg = 0;
  for (int i = 0; i < n; i++) {
    auto &d = descs[i];
    int64_t len = d.length;
    if (!p->center_windows) len -= p->window_length;
    DALIAMD_REQUIRE(d.length > 0 && len >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdSpectrogramSetup: sample %d is shorter than the window", i);
    d.num_windows = (int32_t)(len / p->window_step + 1);  // extract_windows_args.h:38-43
    d.wg_start = wg;
    wg += (d.num_windows + kFramesPerWg - 1) / kFramesPerWg;
  }
  *nwg = wg;
  *lds_bytes = (p->nfft / 2 + kFramesPerWg * p->nfft) * (int)sizeof(float2);
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdSpectrogramRun(daliamdStream_t stream, const daliamdSpectrogramDesc *descs_dev, int n,
                                      const daliamdSpectrogramParams *p, const float *window_dev, int nwg, int lds_bytes) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && p && window_dev && n > 0 && nwg > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdSpectrogramRun: invalid argument");
  if (lds_bytes > 64 * 1024)
    DALIAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(SpectrogramKernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipLaunchKernelGGL(SpectrogramKernel, dim3(XcdGrid(nwg)), dim3(kSpecThreads), lds_bytes, (hipStream_t)stream, descs_dev, n,
                     nwg, *p, window_dev);
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdMelFilterBankWeights(int nfilter, int nfft, float sample_rate, float freq_low, float freq_high,
                                            int normalize, int formula, float *weights) {
  DALIAMD_REQUIRE(weights && nfilter > 0 && nfft > 0 && sample_rate > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdMelFilterBankWeights: invalid argument");
  if (freq_high <= 0) freq_high = sample_rate / 2;
  DALIAMD_REQUIRE(freq_low >= 0 && freq_low <= sample_rate / 2 && freq_high >= 0 && freq_high <= sample_rate / 2,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "freq_low / freq_high must lie in [0, sample_rate/2]");
  // MelFilterImplBase ctor (mel_scale.h:79-130) + MelFilterBankCpu::Impl (mel_filter_bank_cpu.cc:44-70)
  const int nbin = nfft / 2 + 1;
  double mel_low = HzToMel(freq_low, formula), mel_high = HzToMel(freq_high, formula);
  double hz_step = static_cast<double>(sample_rate) / nfft;
  double mel_delta = (mel_high - mel_low) / (nfilter + 1);
  double inv_hz_step = 1.0 / hz_step;
  int b0 = (int)std::ceil(freq_low * inv_hz_step), b1 = (int)std::ceil(freq_high * inv_hz_step);
  if (b1 > nbin) b1 = nbin;
  std::vector<float> wdown(nbin, 0.0f), norm(nfilter, 1.0f);
  std::vector<int> intervals(nbin, -1);
  double mel0 = mel_low, mel1 = mel_low + mel_delta;
  int fftbin = b0;
  double f = fftbin * hz_step;
  for (int interval = 0; interval <= nfilter; interval++, mel0 = mel1, mel1 += mel_delta) {
    if (interval == nfilter) mel1 = mel_high;
    double f0 = MelToHz(mel0, formula), f1 = MelToHz(mel1, formula);
    if (normalize && interval < nfilter) {
      double f2 = MelToHz(mel1 + mel_delta, formula);
      norm[interval] = (float)(2.0 / (f2 - f0));
    }
    double slope = 1. / (f1 - f0);
    for (; fftbin < b1 && f < f1; fftbin++, f = fftbin * hz_step) {
      wdown[fftbin] = (float)((f1 - f) * slope);
      intervals[fftbin] = interval;
    }
  }
  for (size_t i = 0; i < (size_t)nfilter * nbin; i++) weights[i] = 0.0f;
  for (int b = b0; b < b1; b++) {
    int up = intervals[b], down = up - 1;
    float wd = wdown[b], wu = 1.0f - wd;
    if (down >= 0) weights[(size_t)down * nbin + b] = normalize ? wd * norm[down] : wd;
    if (up >= 0 && up < nfilter) weights[(size_t)up * nbin + b] = normalize ? wu * norm[up] : wu;
  }
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdMelFilterBankSetup(daliamdMelDesc *descs, int n, int *nwg) {
  DALIAMD_REQUIRE(descs && nwg && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdMelFilterBankSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    DALIAMD_REQUIRE(descs[i].frames >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "negative frame count");
    descs[i].wg_start = wg;
    wg += (descs[i].frames + 63) / 64;
  }
  *nwg = wg;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdMelFilterBankRun(daliamdStream_t stream, const daliamdMelDesc *descs_dev, int n, int nwg,
                                        const float *W, int nfilter, int nbins) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && W && nfilter > 0 && nbins > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdMelFilterBankRun: invalid argument");
  int mt = (nfilter + 15) / 16;
  DALIAMD_REQUIRE(mt <= 16, DALIAMD_ERROR_UNSUPPORTED, "daliamdMelFilterBankRun: at most 256 filters are supported, got %d", nfilter);
  dim3 grid(XcdGrid(nwg)), block(kMelThreads);
  hipStream_t s = (hipStream_t)stream;
#define MEL_CASE(MT) case MT: hipLaunchKernelGGL(MelKernel<MT>, grid, block, 0, s, descs_dev, n, nwg, W, nfilter, nbins); break;
  switch (mt) {
    MEL_CASE(1) MEL_CASE(2) MEL_CASE(3) MEL_CASE(4) MEL_CASE(5) MEL_CASE(6) MEL_CASE(7) MEL_CASE(8)
    default: hipLaunchKernelGGL(MelKernel<16>, grid, block, 0, s, descs_dev, n, nwg, W, nfilter, nbins); break;
  }
#undef MEL_CASE
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdToDecibelsRun(daliamdStream_t stream, const daliamdDecibelDesc *descs_dev, int n, float multiplier,
                                     float reference, float cutoff_db) {
  if (n == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdToDecibelsRun: invalid argument");
  float min_ratio = std::pow(10.0f, cutoff_db / multiplier);        // to_decibels_op.h:41-49
  if (min_ratio == 0.0f) min_ratio = std::nextafter(0.0f, 1.0f);
  float mul_log2 = multiplier * 0.3010299956639812f;
  hipLaunchKernelGGL(DecibelKernel, dim3(n), dim3(kDbThreads), 0, (hipStream_t)stream, descs_dev, mul_log2, reference, min_ratio);
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
