// Audio feature kernels for gfx950: spectrogram (fused windowing + in-LDS radix-4 FFT + power), mel filter bank as a
// banded matrix product, and to_decibels with a per-sample max reduction.  See include/dali_amd_kernels.h for the
// reference counterparts.  f32 throughout; parity with the reference's CPU path is tolerance-based for the FFT
// (its FFTS library is not available; tests use a float64 FFT like the reference's own tests do).
#include <algorithm>
#include <cmath>
#include <vector>
#include "common.h"

namespace daliamd {

#pragma clang fp contract(fast)  // tolerance-based parity here (f32 FFT vs the oracle's f64): let the compiler form FMAs

// pointers read from a descriptor have no known address space: say "global" so that the loads do not become flat
// ones, which tie up the LDS counter as well
using GFloat = const float __attribute__((address_space(1)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
using GFloat2 = const floatx2 __attribute__((address_space(1)));
using GOutFloat = float __attribute__((address_space(1)));
using GInt16 = const int16_t __attribute__((address_space(1)));
using GWord = const uint32_t __attribute__((address_space(1)));
// daliamdSpectrogramParams.input_pcm16: the signal as 16-bit PCM, sample / 32768 (exact in float)
constexpr float kPcm16Scale = 1.0f / 32768.0f;
__device__ __forceinline__ float SampleAt(GFloat *in, bool pcm16, long long idx) {
  return pcm16 ? (float)((GInt16 *)in)[idx] * kPcm16Scale : in[idx];
}

// =============================================================================================
// spectrogram
//
// One frame per wave at a time, kFramesPerWave frames one after another.  The nfft real samples of a frame are packed
// into N = nfft/2 complex points z[n] = x[2n] + i x[2n+1]; a Stockham radix-4 (+ one radix-2 step when log2 N is odd)
// FFT runs in the wave's own LDS buffer - every lane keeps all the points of its butterflies in registers between the
// read and the write of a step, so one buffer is enough and the only synchronisation is wave-local.  The spectrum of
// the real signal is then X[k] = E[k] + W^k O[k], X[N-k] = conj(E[k] - W^k O[k]) with E / O the even / odd parts of Z.
// Powers go to an LDS tile [bin][frame] so that the frequency-major output is written 16 frames (64 bytes) at a time.
// =============================================================================================
constexpr int kSpecWaves = 4;
constexpr int kSpecThreads = 64 * kSpecWaves;
inline __host__ __device__ constexpr int SpecFramesPerWave(int nfft) { return nfft >= 4096 ? 2 : 4; }  // LDS budget
inline int SpecFramesPerWg(int nfft) { return kSpecWaves * SpecFramesPerWave(nfft); }

__device__ __forceinline__ long long Reflect101L(long long idx, long long size) {
  if (size < 2) return size - 1;
  for (;;) {
    if (idx < 0) idx = -idx;
    else if (idx >= size) idx = 2 * size - 2 - idx;
    else break;
  }
  return idx;
}

__device__ __forceinline__ void SpecWaveSync() {  // LDS accesses of one wave execute in order; compiler fence only
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 CMul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 CAdd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 CSub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// tw[k] = exp(-2 pi i k / (2N)), k < N; the second half of the circle is the negated first half
template <int N>
__device__ __forceinline__ float2 Twiddle(const float2 *tw, int idx) {
  float2 w = tw[idx & (N - 1)];
  return idx >= N ? make_float2(-w.x, -w.y) : w;
}

// One Stockham step of radix R over the N points in `work`: butterfly j reads work[j + r N/R], multiplies by
// exp(-2 pi i r k / (Ns R)), k = j mod Ns, and writes work[(j - k) R + k + r Ns].
template <int N, int NS>
__device__ __forceinline__ void Radix4Step(float2 *work, const float2 *tw, int lane) {
  constexpr int per = N / 4, UB = (per + 63) / 64;
  float2 v[UB][4];
#pragma unroll
  for (int u = 0; u < UB; u++) {
    const int j = lane + 64 * u;
    if (j < per) {
      const int k = j & (NS - 1);
      float2 a = work[j], b = work[j + per], c = work[j + 2 * per], d = work[j + 3 * per];
      if (NS > 1) {
        const int m = 2 * k * (N / (NS * 4));  // table resolution is 2N
        b = CMul(b, Twiddle<N>(tw, m));
        c = CMul(c, Twiddle<N>(tw, 2 * m));
        d = CMul(d, Twiddle<N>(tw, 3 * m));
      }
      const float2 s0 = CAdd(a, c), s1 = CSub(a, c), s2 = CAdd(b, d), df = CSub(b, d);
      const float2 s3 = make_float2(df.y, -df.x);  // -i (b - d)
      v[u][0] = CAdd(s0, s2);
      v[u][1] = CAdd(s1, s3);
      v[u][2] = CSub(s0, s2);
      v[u][3] = CSub(s1, s3);
    }
  }
  SpecWaveSync();
#pragma unroll
  for (int u = 0; u < UB; u++) {
    const int j = lane + 64 * u;
    if (j < per) {
      const int k = j & (NS - 1);
      float2 *o = work + ((j - k) << 2) + k;
#pragma unroll
      for (int r = 0; r < 4; r++) o[r * NS] = v[u][r];
    }
  }
  SpecWaveSync();
}

template <int N, int NS>
__device__ __forceinline__ void Radix2Step(float2 *work, const float2 *tw, int lane) {
  constexpr int per = N / 2, UB = (per + 63) / 64;
  float2 v[UB][2];
#pragma unroll
  for (int u = 0; u < UB; u++) {
    const int j = lane + 64 * u;
    if (j < per) {
      const int k = j & (NS - 1);
      float2 a = work[j], b = work[j + per];
      if (NS > 1) b = CMul(b, Twiddle<N>(tw, 2 * k * (N / (NS * 2))));
      v[u][0] = CAdd(a, b);
      v[u][1] = CSub(a, b);
    }
  }
  SpecWaveSync();
#pragma unroll
  for (int u = 0; u < UB; u++) {
    const int j = lane + 64 * u;
    if (j < per) {
      const int k = j & (NS - 1);
      float2 *o = work + ((j - k) << 1) + k;
      o[0] = v[u][0];
      o[NS] = v[u][1];
    }
  }
  SpecWaveSync();
}

template <int N, int LOG2N, int S>
struct FftSteps {
  static __device__ __forceinline__ void Run(float2 *work, const float2 *tw, int lane) {
    if constexpr (LOG2N - S >= 2) {
      Radix4Step<N, (1 << S)>(work, tw, lane);
      FftSteps<N, LOG2N, S + 2>::Run(work, tw, lane);
    } else if constexpr (LOG2N - S == 1) {
      Radix2Step<N, (1 << S)>(work, tw, lane);
    }
  }
};

template <int LOG2N>  // N = nfft / 2 = 1 << LOG2N complex points per frame
__global__ __launch_bounds__(kSpecThreads) void SpectrogramKernel(const daliamdSpectrogramDesc *__restrict__ descs, int ndesc,
                                                                  int total_wg, daliamdSpectrogramParams p,
                                                                  const float *__restrict__ window) {
  constexpr int N = 1 << LOG2N, nfft = 2 * N;
  constexpr int U = (N + 63) / 64;  // complex points per lane
  constexpr int FPW = SpecFramesPerWave(nfft), FPG = FPW * kSpecWaves, TS = FPG + 1;
  extern __shared__ __attribute__((aligned(16))) float2 spec_lds[];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdSpectrogramDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2 *tw = spec_lds;                              // [N]
  float2 *work = spec_lds + N + wave * N;             // [N] per wave
  float *tile = reinterpret_cast<float *>(spec_lds + N + kSpecWaves * N);  // [N + 1][TS]
  for (int k = tid; k < N; k += kSpecThreads) {
    float sn, cs;
    sincospif(-(float)k / (float)N, &sn, &cs);
    tw[k] = make_float2(cs, sn);
  }
  // the window, centred inside nfft (fft_cpu_impl_ffts.cc:108-110), lives in registers for all the frames of the wave
  const int pad0 = (nfft - p.window_length) / 2;
  float2 wr[U];
  unsigned inwin[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int i = lane + 64 * u, w0 = 2 * i - pad0, w1 = w0 + 1;
    const bool in0 = i < N && w0 >= 0 && w0 < p.window_length, in1 = i < N && w1 >= 0 && w1 < p.window_length;
    wr[u] = make_float2(in0 ? window[w0] : 0.0f, in1 ? window[w1] : 0.0f);
    inwin[u] = (in0 ? 1u : 0u) | (in1 ? 2u : 0u);
  }
  __syncthreads();
  const int T = d.num_windows;
  const int t0 = (wg - d.wg_start) * FPG;
  GFloat *in = (GFloat *)d.in;
  const bool pcm16 = p.input_pcm16 != 0;
  for (int f = 0; f < FPW; f++) {
    const int fi = f * kSpecWaves + wave;  // frame slot inside the tile
    const int frame = t0 + fi;
    if (frame >= T) break;
    // buffer position j holds window[j - pad0] * in[base + j]
    const long long base = (long long)frame * p.window_step - (p.center_windows ? p.window_length / 2 : 0) - pad0;
    if (base >= 0 && base + nfft <= d.length) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = lane + 64 * u;
        if (i < N) {
          const float x0 = SampleAt(in, pcm16, base + 2 * i), x1 = SampleAt(in, pcm16, base + 2 * i + 1);
          work[i] = make_float2((inwin[u] & 1) ? wr[u].x * x0 : 0.0f, (inwin[u] & 2) ? wr[u].y * x1 : 0.0f);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = lane + 64 * u;
        if (i < N) {
          float x[2] = {0.0f, 0.0f};
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (!((inwin[u] >> h) & 1)) continue;
            const long long idx = base + 2 * i + h;
            const float w = h ? wr[u].y : wr[u].x;
            if (p.reflect_padding) x[h] = w * SampleAt(in, pcm16, Reflect101L(idx, d.length));
            else if (idx >= 0 && idx < d.length) x[h] = w * SampleAt(in, pcm16, idx);
          }
          work[i] = make_float2(x[0], x[1]);
        }
      }
    }
    SpecWaveSync();
    FftSteps<N, LOG2N, 0>::Run(work, tw, lane);
    // real-signal spectrum from the half-length transform, power / magnitude into the tile
    constexpr int UP = (N / 2 + 1 + 63) / 64;
#pragma unroll
    for (int u = 0; u < UP; u++) {
      const int k = lane + 64 * u;
      if (k <= N / 2) {
        const float2 zk = work[k & (N - 1)], zn = work[(N - k) & (N - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const float2 t = CMul(tw[k], o);
        const float2 xa = CAdd(e, t), xb = CSub(e, t);
        float pa = xa.x * xa.x + xa.y * xa.y, pb = xb.x * xb.x + xb.y * xb.y;
        if (p.power != 2) {
          pa = sqrtf(pa);
          pb = sqrtf(pb);
        }
        tile[k * TS + fi] = pa;
        tile[(N - k) * TS + fi] = pb;
      }
    }
    SpecWaveSync();
  }
  __syncthreads();
  // frequency-major output: FPG consecutive frames per bin
  for (int idx = tid; idx < (N + 1) * FPG; idx += kSpecThreads) {
    const int b = idx / FPG, f = idx % FPG;
    if (t0 + f < T) d.out[(size_t)b * T + t0 + f] = tile[b * TS + f];
  }
}

// ---------------------------------------------------------------------------------------------
// nfft = 512 / 1024: the same transform with everything that does not depend on the frame in registers (window,
// the twiddles of every step - read once from a host-made table) and the FC frames of a wave in flight at the same
// time, which is what hides the LDS round trip of a step: a single frame per wave leaves the SIMDs 70 % idle.  The
// first radix-4 step works on the samples as loaded (lane l holds points l + 64 u, exactly the operands of its own
// butterflies); work buffers are padded by 2 points per 16 so that the strided writes of the early steps spread over
// the banks; the power tile reuses the work buffers once every wave is done with them.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int PadIdx(int e) { return e + ((e >> 4) << 1); }

// 8-point DFT (forward): three radix-2 stages, the only products are the two rotations by 45 degrees.
__device__ __forceinline__ float2 MulNegI(float2 v) { return make_float2(v.y, -v.x); }   // v * (-i)
__device__ __forceinline__ void Dft8(const float2 x[8], float2 X[8]) {
  constexpr float kS = 0.70710678118654752f;
  const float2 a0 = CAdd(x[0], x[4]), a1 = CSub(x[0], x[4]), a2 = CAdd(x[2], x[6]), a3 = MulNegI(CSub(x[2], x[6]));
  const float2 a4 = CAdd(x[1], x[5]), a5 = CSub(x[1], x[5]), a6 = CAdd(x[3], x[7]), a7 = MulNegI(CSub(x[3], x[7]));
  const float2 b0 = CAdd(a0, a2), b1 = CSub(a0, a2), b2 = CAdd(a1, a3), b3 = CSub(a1, a3);
  const float2 b4 = CAdd(a4, a6), b5 = MulNegI(CSub(a4, a6));
  const float2 s6 = CAdd(a5, a7), s7 = CSub(a5, a7);
  const float2 b6 = make_float2(kS * (s6.x + s6.y), kS * (s6.y - s6.x));      // * (1 - i) / sqrt 2
  const float2 b7 = make_float2(kS * (s7.y - s7.x), -kS * (s7.x + s7.y));     // * (-1 - i) / sqrt 2
  X[0] = CAdd(b0, b4); X[4] = CSub(b0, b4);
  X[2] = CAdd(b1, b5); X[6] = CSub(b1, b5);
  X[1] = CAdd(b2, b6); X[5] = CSub(b2, b6);
  X[3] = CAdd(b3, b7); X[7] = CSub(b3, b7);
}

// MEL != 0: the spectrogram never leaves the workgroup.  Its 16-frame power tile is multiplied by the mel filter bank
// where it sits in LDS and only the [nfilter][frames] result (mel energies, or their decibels) is written - see MelFromTile.
struct MelFuse {
  const float *tiles;       // MFMA variant: the filter bank as 16 x 4 tiles in A-operand lane order
  const int32_t *row_blocks;
  const float *weights;     // VALU variant: dense [nfilter][nbins] + bands
  const int32_t *bands;
  int32_t num_row_blocks, nfilter, nbins, decibels;
  float mul_log2, inv_ref, min_ratio;
  uint32_t *max_bits;
  int32_t max_stride;
};
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[m][t0 + f] = sum_k W[m][k] * tile[k][f] for the FPG = 16 frames of the tile.
//   MEL == 1  v_mfma_f32_16x16x4_f32: D[16 filters][16 frames] += A[16 filters][4 bins] . B[4 bins][16 frames].  The 16 frames
//             of the tile ARE the N of the instruction; B comes straight from LDS (lane l: tile[k0 + l / 16][l % 16]), A from
//             a host-made table of 16 x 4 tiles in lane order (one coalesced 256-byte load per instruction, L2-resident:
//             the table is the same for every workgroup).  The filters are triangles that overlap only their neighbours,
//             so a row block of 16 filters touches a contiguous range of bins: only those tiles exist (about 150 of the
//             5 x 129 of the dense product for 80 filters x 513 bins).  f32 in, f32 accumulate: the result is the ascending-k
//             fmaf chain of the banded product (zero weights add exact zeros).
//   MEL == 2  the banded product on the VALU, MelKernel's loop: thread = (filter, frame), 2 FMAs per tile element.
template <int MEL, int N, int TS>
__device__ __forceinline__ void MelFromTile(const float *tile, const MelFuse &mel, float *out, int T, int t0, int di, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  GOutFloat *gout = (GOutFloat *)out;
  float vmax = 0.0f;
  auto emit = [&](int m, int f, float acc) {
    if (m >= mel.nfilter || t0 + f >= T) return;
    vmax = fmaxf(vmax, acc);
    if (mel.decibels) acc = mel.mul_log2 * log2f(fmaxf(mel.min_ratio, acc * mel.inv_ref));
    gout[(size_t)m * T + t0 + f] = acc;
  };
  if constexpr (MEL == 1) {
    const int kk = lane >> 4, j = lane & 15;
    for (int mb = 0; mb < mel.num_row_blocks; mb++) {
      const int32_t __attribute__((address_space(1))) *rb = (const int32_t __attribute__((address_space(1))) *)mel.row_blocks + 4 * mb;
      if (rb[3] != wave) continue;   // the host dealt the row blocks to the waves by their tile counts
      const int first = rb[0], count = rb[1], k0 = rb[2];
      GFloat *a = (GFloat *)mel.tiles + (size_t)first * 64 + lane;
      const float *b = tile + (k0 + kk) * TS + j;
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      int i = 0;
      // eight operand pairs in flight in front of the eight dependent instructions: the A tiles come from L2 (a few
      // hundred cycles), and ONE accumulator keeps the sum the ascending-k chain of the banded product
      for (; i + 8 <= count; i += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; q++) av[q] = a[(i + q) * 64];
#pragma unroll
        for (int q = 0; q < 8; q++) bv[q] = b[(i + q) * 4 * TS];
#pragma unroll
        for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
      }
      for (; i < count; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * 64], b[i * 4 * TS], acc, 0, 0, 0);
      // D: column = lane % 16 (frame), rows 4 * (lane / 16) + r (filter inside the row block)
#pragma unroll
      for (int r = 0; r < 4; r++) emit(mb * 16 + kk * 4 + r, j, acc[r]);
    }
  } else {
    const int f = tid & 15;
    for (int m = tid >> 4; m < mel.nfilter; m += kSpecThreads / 16) {
      const int kb = ((const int32_t __attribute__((address_space(1))) *)mel.bands)[2 * m];
      const int ke = ((const int32_t __attribute__((address_space(1))) *)mel.bands)[2 * m + 1];
      GFloat *w = (GFloat *)mel.weights + (size_t)m * mel.nbins;
      float acc = 0.0f;
      for (int k = kb; k < ke; k++) acc = fmaf(w[k], tile[k * TS + f], acc);
      emit(m, f, acc);
    }
  }
  if (mel.max_bits) {   // to_decibels(reference = the sample's maximum): fold this tile's maximum in
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0 && vmax > 0.0f)
      atomicMax((uint32_t *)((uint8_t *)mel.max_bits + (size_t)di * mel.max_stride), __float_as_uint(vmax));
  }
}

template <int LOG2N, int FC, int MEL = 0>
__global__ __launch_bounds__(kSpecThreads) void SpectrogramFastKernel(const daliamdSpectrogramDesc *__restrict__ descs,
                                                                      int ndesc, int total_wg, daliamdSpectrogramParams p,
                                                                      const float *__restrict__ window,
                                                                      const float2 *__restrict__ twg, MelFuse mel) {
  constexpr int N = 1 << LOG2N, nfft = 2 * N;
  constexpr int U = N / 64;            // points per lane and frame
  constexpr int UB = U / 4;            // radix-4 butterflies per lane and frame
  constexpr int FPW = 4, FPG = FPW * kSpecWaves, TS = FPG + 1;
  constexpr int WS = N + N / 8;        // padded work buffer (points)
  constexpr int R4 = LOG2N / 2;        // radix-4 steps; one radix-2 step follows when LOG2N is odd
  constexpr int UP = (N / 2 + 1 + 63) / 64;
  static_assert(U >= 4 && R4 >= 2 && R4 <= 4 && FPW % FC == 0, "supported sizes: nfft 512, 1024");
  extern __shared__ __attribute__((aligned(16))) float2 spec_lds[];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2 *wbase = spec_lds + wave * (FC * WS);
  float *tile = reinterpret_cast<float *>(spec_lds);  // [N + 1][TS], after the transforms
  auto TwG = [&](int idx) {
    float2 w = twg[idx & (N - 1)];
    return idx >= N ? make_float2(-w.x, -w.y) : w;
  };
  // ---- per-lane constants ----
  const int pad0 = (nfft - p.window_length) / 2;
  float2 wr[U];
  unsigned inwin = 0;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int i = lane + 64 * u, w0 = 2 * i - pad0, w1 = w0 + 1;
    const bool in0 = w0 >= 0 && w0 < p.window_length, in1 = w1 >= 0 && w1 < p.window_length;
    wr[u] = make_float2(in0 ? window[w0] : 0.0f, in1 ? window[w1] : 0.0f);
    inwin |= (in0 ? 1u : 0u) << (2 * u) | (in1 ? 2u : 0u) << (2 * u);
  }
  float2 tw4[R4 - 1][3];  // steps NS = 4, 16, 64: k = j mod NS is the same for all the butterflies of a lane
#pragma unroll
  for (int s = 1; s < R4; s++) {
    const int NS = 1 << (2 * s);
    const int m = 2 * (lane & (NS - 1)) * (N / (NS * 4));
    tw4[s - 1][0] = TwG(m);
    tw4[s - 1][1] = TwG(2 * m);
    tw4[s - 1][2] = TwG(3 * m);
  }
  float2 tw2[U / 2];      // the radix-2 step (NS = N / 2): k = j
  if constexpr (LOG2N & 1) {
#pragma unroll
    for (int u = 0; u < U / 2; u++) tw2[u] = twg[2 * (lane + 64 * u)];
  }
  // N = 512 (nfft 1024) runs as three radix-8 Stockham steps - 512 = 8 x 8 x 8, one butterfly per lane and step, THREE
  // LDS round trips where four radix-4 steps and a radix-2 step take five: exp(-2 pi i r k / (8 NS)), k = lane mod NS
  constexpr bool kRadix8 = LOG2N == 9;
  float2 tw8b[7], tw8c[7];
  if constexpr (kRadix8) {
#pragma unroll
    for (int r = 1; r < 8; r++) {
      tw8b[r - 1] = TwG(r * (lane & 7) * (2 * N / 64));
      tw8c[r - 1] = TwG(r * lane * (2 * N / 512));
    }
  }
  float2 twp[UP];
#pragma unroll
  for (int u = 0; u < UP; u++) twp[u] = twg[min(lane + 64 * u, N - 1)];
  // the descriptor of this workgroup: one parallel look over the table instead of a chain of dependent loads
  int di = 0;
  for (int b0 = 0; b0 < ndesc; b0 += 64) {
    const int i = b0 + lane;
    const bool le = i < ndesc && descs[i].wg_start <= wg;
    di += __popcll(__ballot(le));
  }
  di = __builtin_amdgcn_readfirstlane(di - 1);
  const daliamdSpectrogramDesc &d = descs[di];
  const int T = d.num_windows;
  const int t0 = (wg - d.wg_start) * FPG;
  GFloat *in = (GFloat *)d.in;
  const bool pcm16 = p.input_pcm16 != 0;
  float pw[FPW][UP][2];
#pragma unroll
  for (int rd = 0; rd < FPW / FC; rd++) {
    float2 v[FC][UB][4];
    // ---- load and window the FC frames (frames past the end repeat the last one: no divergence, results unused) ----
    float2 z[FC][U];
    long long base[FC];
    bool interior = true, pairs_aligned = true;
#pragma unroll
    for (int c = 0; c < FC; c++) {
      const int frame = min(t0 + wave * FPW + rd * FC + c, T - 1);
      base[c] = (long long)frame * p.window_step - (p.center_windows ? p.window_length / 2 : 0) - pad0;
      interior = interior && base[c] >= 0 && base[c] + nfft <= d.length;
      pairs_aligned = pairs_aligned && (pcm16 ? ((uintptr_t)((GInt16 *)in + base[c]) & 3) == 0 : ((uintptr_t)(in + base[c]) & 7) == 0);
    }
    if (interior && pairs_aligned && pcm16) {  // one 4-byte load per point: two 16-bit samples
#pragma unroll
      for (int c = 0; c < FC; c++) {
        GWord *src = (GWord *)((GInt16 *)in + base[c]);
#pragma unroll
        for (int u = 0; u < U; u++) {
          const uint32_t w = src[lane + 64 * u];
          const float x0 = (float)(int16_t)(w & 0xffffu) * kPcm16Scale, x1 = (float)(int16_t)(w >> 16) * kPcm16Scale;
          z[c][u] = make_float2((inwin >> (2 * u)) & 1 ? wr[u].x * x0 : 0.0f, (inwin >> (2 * u)) & 2 ? wr[u].y * x1 : 0.0f);
        }
      }
    } else if (interior && pairs_aligned) {  // one 8-byte load per point
#pragma unroll
      for (int c = 0; c < FC; c++) {
        GFloat2 *src = (GFloat2 *)(in + base[c]);
#pragma unroll
        for (int u = 0; u < U; u++) {
          const floatx2 x = src[lane + 64 * u];
          z[c][u] = make_float2((inwin >> (2 * u)) & 1 ? wr[u].x * x.x : 0.0f, (inwin >> (2 * u)) & 2 ? wr[u].y * x.y : 0.0f);
        }
      }
    } else if (interior) {
#pragma unroll
      for (int c = 0; c < FC; c++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int i = lane + 64 * u;
          const float x0 = SampleAt(in, pcm16, base[c] + 2 * i), x1 = SampleAt(in, pcm16, base[c] + 2 * i + 1);
          z[c][u] = make_float2((inwin >> (2 * u)) & 1 ? wr[u].x * x0 : 0.0f, (inwin >> (2 * u)) & 2 ? wr[u].y * x1 : 0.0f);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < FC; c++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          float x[2] = {0.0f, 0.0f};
#pragma unroll
          for (int h = 0; h < 2; h++) {
            if (!((inwin >> (2 * u + h)) & 1)) continue;
            const long long idx = base[c] + 2 * (lane + 64 * u) + h;
            const float w = h ? wr[u].y : wr[u].x;
            if (p.reflect_padding) x[h] = w * SampleAt(in, pcm16, Reflect101L(idx, d.length));
            else if (idx >= 0 && idx < d.length) x[h] = w * SampleAt(in, pcm16, idx);
          }
          z[c][u] = make_float2(x[0], x[1]);
        }
      }
    }
    if constexpr (kRadix8) {
      float2 X[FC][8];
      // ---- step 1 (NS = 1, no twiddles) straight from the registers: lane l holds the points l + 64 r of its butterfly;
      // its 8 results are consecutive points ----
#pragma unroll
      for (int c = 0; c < FC; c++) {
        Dft8(z[c], X[c]);
        float4 *o = reinterpret_cast<float4 *>(wbase + c * WS + PadIdx(8 * lane));   // 64 bytes, 16-byte aligned
#pragma unroll
        for (int q = 0; q < 4; q++) o[q] = make_float4(X[c][2 * q].x, X[c][2 * q].y, X[c][2 * q + 1].x, X[c][2 * q + 1].y);
      }
      SpecWaveSync();
      // ---- step 2 (NS = 8) ----
#pragma unroll
      for (int c = 0; c < FC; c++) {
        const float2 *work = wbase + c * WS;
        float2 v8[8];
        v8[0] = work[PadIdx(lane)];
#pragma unroll
        for (int r = 1; r < 8; r++) v8[r] = CMul(work[PadIdx(lane + 64 * r)], tw8b[r - 1]);
        Dft8(v8, X[c]);
      }
      SpecWaveSync();
#pragma unroll
      for (int c = 0; c < FC; c++) {
        float2 *work = wbase + c * WS;
        const int k = lane & 7, o = ((lane - k) << 3) + k;
#pragma unroll
        for (int r = 0; r < 8; r++) work[PadIdx(o + 8 * r)] = X[c][r];
      }
      SpecWaveSync();
      // ---- step 3 (NS = 64): butterfly l reads and writes the points l + 64 r - in place, no hazard between lanes ----
#pragma unroll
      for (int c = 0; c < FC; c++) {
        float2 *work = wbase + c * WS;
        float2 v8[8];
        v8[0] = work[PadIdx(lane)];
#pragma unroll
        for (int r = 1; r < 8; r++) v8[r] = CMul(work[PadIdx(lane + 64 * r)], tw8c[r - 1]);
        Dft8(v8, X[c]);
#pragma unroll
        for (int r = 0; r < 8; r++) work[PadIdx(lane + 64 * r)] = X[c][r];
      }
      SpecWaveSync();
    } else {
    // ---- first radix-4 step (no twiddles) straight from the registers ----
#pragma unroll
    for (int c = 0; c < FC; c++) {
      float2 *work = wbase + c * WS;
#pragma unroll
      for (int ub = 0; ub < UB; ub++) {
        const float2 a = z[c][ub], b = z[c][ub + UB], cc = z[c][ub + 2 * UB], dd = z[c][ub + 3 * UB];
        const float2 s0 = CAdd(a, cc), s1 = CSub(a, cc), s2 = CAdd(b, dd), df = CSub(b, dd);
        const float2 s3 = make_float2(df.y, -df.x);
        const float2 o0 = CAdd(s0, s2), o1 = CAdd(s1, s3), o2 = CSub(s0, s2), o3 = CSub(s1, s3);
        float4 *o = reinterpret_cast<float4 *>(work + PadIdx(4 * (lane + 64 * ub)));  // 4 consecutive points, 32-byte aligned
        o[0] = make_float4(o0.x, o0.y, o1.x, o1.y);
        o[1] = make_float4(o2.x, o2.y, o3.x, o3.y);
      }
    }
    SpecWaveSync();
    // ---- the remaining radix-4 steps ----
#pragma unroll
    for (int s = 1; s < R4; s++) {
      const int NS = 1 << (2 * s);
      constexpr int per = N / 4;
#pragma unroll
      for (int c = 0; c < FC; c++) {
        const float2 *work = wbase + c * WS;
#pragma unroll
        for (int ub = 0; ub < UB; ub++) {
          const int j = lane + 64 * ub;
          const float2 a = work[PadIdx(j)];
          const float2 b = CMul(work[PadIdx(j + per)], tw4[s - 1][0]);
          const float2 cc = CMul(work[PadIdx(j + 2 * per)], tw4[s - 1][1]);
          const float2 dd = CMul(work[PadIdx(j + 3 * per)], tw4[s - 1][2]);
          const float2 s0 = CAdd(a, cc), s1 = CSub(a, cc), s2 = CAdd(b, dd), df = CSub(b, dd);
          const float2 s3 = make_float2(df.y, -df.x);
          v[c][ub][0] = CAdd(s0, s2);
          v[c][ub][1] = CAdd(s1, s3);
          v[c][ub][2] = CSub(s0, s2);
          v[c][ub][3] = CSub(s1, s3);
        }
      }
      SpecWaveSync();
#pragma unroll
      for (int c = 0; c < FC; c++) {
        float2 *work = wbase + c * WS;
#pragma unroll
        for (int ub = 0; ub < UB; ub++) {
          const int j = lane + 64 * ub, k = j & (NS - 1);
          const int o = ((j - k) << 2) + k;
#pragma unroll
          for (int r = 0; r < 4; r++) work[PadIdx(o + r * NS)] = v[c][ub][r];
        }
      }
      SpecWaveSync();
    }
    // ---- radix-2 step when log2 N is odd: NS = N / 2, so butterfly j writes j and j + NS ----
    if constexpr (LOG2N & 1) {
      constexpr int per = N / 2;
      float2 v2[FC][U / 2][2];
#pragma unroll
      for (int c = 0; c < FC; c++) {
        const float2 *work = wbase + c * WS;
#pragma unroll
        for (int u = 0; u < U / 2; u++) {
          const int j = lane + 64 * u;
          const float2 a = work[PadIdx(j)], b = CMul(work[PadIdx(j + per)], tw2[u]);
          v2[c][u][0] = CAdd(a, b);
          v2[c][u][1] = CSub(a, b);
        }
      }
      SpecWaveSync();
#pragma unroll
      for (int c = 0; c < FC; c++) {
        float2 *work = wbase + c * WS;
#pragma unroll
        for (int u = 0; u < U / 2; u++) {
          const int j = lane + 64 * u;
          work[PadIdx(j)] = v2[c][u][0];
          work[PadIdx(j + per)] = v2[c][u][1];
        }
      }
      SpecWaveSync();
    }
    }   // radix-4 path
    // ---- spectrum of the real signal, power / magnitude ----
#pragma unroll
    for (int c = 0; c < FC; c++) {
      const float2 *work = wbase + c * WS;
#pragma unroll
      for (int u = 0; u < UP; u++) {
        const int k = min(lane + 64 * u, N / 2);
        const float2 zk = work[PadIdx(k)], zn = work[PadIdx((N - k) & (N - 1))];
        const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const float2 t = CMul(twp[u], o);
        const float2 xa = CAdd(e, t), xb = CSub(e, t);
        float pa = xa.x * xa.x + xa.y * xa.y, pb = xb.x * xb.x + xb.y * xb.y;
        if (p.power != 2) {
          pa = sqrtf(pa);
          pb = sqrtf(pb);
        }
        pw[rd * FC + c][u][0] = pa;
        pw[rd * FC + c][u][1] = pb;
      }
    }
    SpecWaveSync();
  }
  __syncthreads();  // every wave is done with its work buffers: the tile may overwrite them
#pragma unroll
  for (int f = 0; f < FPW; f++) {
    if (t0 + wave * FPW + f >= T) continue;
#pragma unroll
    for (int u = 0; u < UP; u++) {
      const int k = lane + 64 * u;
      if (k <= N / 2) {
        tile[k * TS + wave * FPW + f] = pw[f][u][0];
        tile[(N - k) * TS + wave * FPW + f] = pw[f][u][1];
      }
    }
  }
  if constexpr (MEL == 1) {   // the last tile of a row block may reach three bins past the spectrum: they must read as zero
    if (tid < 3 * TS) tile[(N + 1) * TS + tid] = 0.0f;
  }
  __syncthreads();
  if constexpr (MEL != 0) {
    MelFromTile<MEL, N, TS>(tile, mel, d.out, T, t0, di, tid);
  } else {
    GOutFloat *gout = (GOutFloat *)d.out;
    for (int idx = tid; idx < (N + 1) * FPG; idx += kSpecThreads) {
      const int b = idx / FPG, f = idx % FPG;
      if (t0 + f < T) gout[(size_t)b * T + t0 + f] = tile[b * TS + f];
    }
  }
}
constexpr bool SpecFastPath(int nfft) { return nfft == 512 || nfft == 1024; }
constexpr int kSpecFastConcurrent = 4;
inline int SpecFastLds(int nfft) {
  const int N = nfft / 2;
  const int work = kSpecWaves * kSpecFastConcurrent * (N + N / 8) * (int)sizeof(float2);
  const int tile = (N + 1 + 3) * (4 * kSpecWaves + 1) * (int)sizeof(float);   // (+ 3 zero rows for the fused mel product)
  return work > tile ? work : tile;
}

// =============================================================================================
// mel filter bank: out[m][t] = sum over the filter's band of W[m][k] * S[k][t].  The triangular filters overlap only
// their neighbours, so the product is a banded one: 2 multiply-adds per spectrogram element, bound by reading S.  A
// workgroup owns 64 frames (one 256-byte row segment per load); its waves take the filters round-robin, so the second
// read of a bin (by the neighbouring filter) comes from the CU's cache.
// =============================================================================================
constexpr int kMelWaves = 8;
constexpr int kMelThreads = 64 * kMelWaves;

__global__ __launch_bounds__(kMelThreads) void MelKernel(const daliamdMelDesc *__restrict__ descs, int ndesc, int total_wg,
                                                         const float *__restrict__ W, const int32_t *__restrict__ bands,
                                                         const float *__restrict__ row_scale, int nfilter, int K) {
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdMelDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = d.frames;
  const int col = (wg - d.wg_start) * 64 + lane;
  const bool ok = col < T;
  GFloat *S = (GFloat *)d.in + (ok ? col : 0);
  GOutFloat *out = (GOutFloat *)d.out;
  for (int m = wave; m < nfilter; m += kMelWaves) {
    const int kb = bands ? ((const int32_t __attribute__((address_space(1))) *)bands)[2 * m] : 0;
    const int ke = bands ? ((const int32_t __attribute__((address_space(1))) *)bands)[2 * m + 1] : K;
    GFloat *w = (GFloat *)W + (size_t)m * K;
    float acc = 0.0f;
    int k = kb;
    for (; k + 4 <= ke; k += 4) {
      const float s0 = S[(size_t)k * T], s1 = S[(size_t)(k + 1) * T], s2 = S[(size_t)(k + 2) * T], s3 = S[(size_t)(k + 3) * T];
      acc = fmaf(w[k], s0, acc);
      acc = fmaf(w[k + 1], s1, acc);
      acc = fmaf(w[k + 2], s2, acc);
      acc = fmaf(w[k + 3], s3, acc);
    }
    for (; k < ke; k++) acc = fmaf(w[k], S[(size_t)k * T], acc);
    if (row_scale) acc = ((GFloat *)row_scale)[m] * acc;   // DCT liftering: coefficient * sum, like ApplyLifter
    if (ok) out[(size_t)m * T + col] = acc;
  }
}

// =============================================================================================
// to_decibels
// =============================================================================================
// Two launches when the reference is the per-sample maximum: chunk maxima folded into the descriptor's `max_bits`
// slot with an integer atomic max (the values are non-negative, so their bit patterns order like the floats), then
// the element-wise pass.  A sample is cut into chunks of kDbChunk elements so that a batch of short utterances still
// fills the chip (one workgroup per sample left it 3/4 idle).
constexpr int kDbThreads = 256;
constexpr int kDbChunk = kDbThreads * 16;

__global__ __launch_bounds__(kDbThreads) void DecibelMaxKernel(daliamdDecibelDesc *__restrict__ descs, int ndesc, int total_wg) {
  __shared__ float red[kDbThreads / 64];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const int di = FindDesc(descs, ndesc, wg);
  const daliamdDecibelDesc &d = descs[di];
  const int tid = threadIdx.x;
  GFloat *in = (GFloat *)d.in;
  const int64_t i0 = (int64_t)(wg - d.wg_start) * kDbChunk;
  const int64_t i1 = min(i0 + kDbChunk, d.size);
  float m = 0.0f;
  for (int64_t i = i0 + tid; i < i1; i += kDbThreads) m = fmaxf(m, in[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));  // wave64 shuffle reduction
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kDbThreads / 64; w++) m = fmaxf(m, red[w]);
    if (m > 0.0f) atomicMax(&descs[di].max_bits, __float_as_uint(m));
  }
}

__global__ __launch_bounds__(kDbThreads) void DecibelKernel(const daliamdDecibelDesc *__restrict__ descs, int ndesc, int total_wg,
                                                            float mul_log2, float reference, float min_ratio) {
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdDecibelDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int tid = threadIdx.x;
  float s_ref = reference;
  if (!(reference > 0.0f)) {
    s_ref = __uint_as_float(d.max_bits);
    if (s_ref == 0.0f) s_ref = 1.0f;
  }
  const float inv = s_ref == 1.0f ? 1.0f : 1.0f / s_ref;
  GFloat *in = (GFloat *)d.in;
  GOutFloat *out = (GOutFloat *)d.out;
  const int64_t i0 = (int64_t)(wg - d.wg_start) * kDbChunk;
  const int64_t i1 = min(i0 + kDbChunk, d.size);
  for (int64_t i = i0 + tid; i < i1; i += kDbThreads) out[i] = mul_log2 * log2f(fmaxf(min_ratio, in[i] * inv));
}

// =============================================================================================
// audio resampling: windowed sinc (Hann envelope), window coefficients from a table with linear interpolation
// (dali/kernels/signal/resampling.h:33-106, resampling_cpu.cc:129-172).  A workgroup is one of the reference's blocks of
// 256 outputs: the block's start is computed in double, the position inside it by repeated float additions of the
// step - every thread replays that chain up to its own output, like the reference's loop does.
// =============================================================================================
constexpr int kRsThreads = 256;

__global__ __launch_bounds__(kRsThreads) void AudioResampleKernel(const daliamdAudioResampleDesc *__restrict__ descs, int ndesc,
                                                                  int total_wg, const float *__restrict__ lookup_g,
                                                                  int lookup_size, float wscale, float wcenter, int lobes) {
  extern __shared__ float rs_lookup[];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdAudioResampleDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int tid = threadIdx.x;
  for (int i = tid; i < lookup_size; i += kRsThreads) rs_lookup[i] = ((GFloat *)lookup_g)[i];
  __syncthreads();
  const int64_t out_block = (int64_t)(wg - d.wg_start) * kRsThreads;
  const int64_t out_pos = out_block + tid;
  if (out_pos >= d.out_length) return;
  const double scale = d.in_rate / d.out_rate;
  const float fscale = (float)scale;
  const double in_block_f = (double)out_block * scale;
  const int64_t in_block_i = (int64_t)floor(in_block_f);
  float in_pos = (float)(in_block_f - (double)in_block_i);
  for (int j = 0; j < tid; j++) in_pos += fscale;
  const int xc = (int)ceilf(in_pos);
  int i0 = xc - lobes, i1 = xc + lobes;
  if (i0 + in_block_i < 0) i0 = (int)-in_block_i;
  if (i1 + in_block_i > d.in_length) i1 = (int)(d.in_length - in_block_i);
  const int C = d.channels;
  GFloat *in = (GFloat *)d.in + in_block_i * C;
  GOutFloat *out = (GOutFloat *)d.out + out_pos * C;
  for (int c = 0; c < C; c++) {
    float f = 0.0f;
    float x = (float)i0 - in_pos;
    for (int i = i0; i < i1; i++, x += 1.0f) {
      const float fi = x * wscale + wcenter;
      const float fl = floorf(fi);
      const float di = fi - fl;
      const int li = (int)fl;
      const float w = rs_lookup[li] + di * (rs_lookup[li + 1] - rs_lookup[li]);
      f += in[(int64_t)i * C + c] * w;
    }
    out[c] = f;
  }
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

// =============================================================================================
// Normalised sample-type conversion (ConvertSatNorm): see the header.  1024 elements per workgroup.
// =============================================================================================
constexpr int kCvtThreads = 256, kCvtPerWg = 1024;
__device__ __forceinline__ float LoadNorm(const void *p, int dtype, int64_t i) {
  using G = __attribute__((address_space(1))) const uint8_t;
  G *b = (G *)p;
  switch (dtype) {
    case DALIAMD_INT8: return (float)((const int8_t __attribute__((address_space(1))) *)b)[i] * (1.0f / 127.0f);
    case DALIAMD_UINT8: return (float)b[i] * (1.0f / 255.0f);
    case DALIAMD_INT16: return (float)((const int16_t __attribute__((address_space(1))) *)b)[i] * (1.0f / 32767.0f);
    case DALIAMD_UINT16: return (float)((const uint16_t __attribute__((address_space(1))) *)b)[i] * (1.0f / 65535.0f);
    case DALIAMD_INT32: return (float)((const int32_t __attribute__((address_space(1))) *)b)[i] * (1.0f / 2147483648.0f);
    case DALIAMD_UINT32: return (float)((const uint32_t __attribute__((address_space(1))) *)b)[i] * (1.0f / 4294967296.0f);
    default: return ((const float __attribute__((address_space(1))) *)b)[i];
  }
}
// clamp<Out>(std::round(v * max)): the bounds compare as floats (the int32 / uint32 maxima round up to 2^31 / 2^32)
__device__ __forceinline__ void StoreNorm(void *p, int dtype, int64_t i, float v) {
  using GB = __attribute__((address_space(1))) uint8_t;
  GB *b = (GB *)p;
  switch (dtype) {
    case DALIAMD_INT8: {
      const float r = roundf(v * 127.0f);
      ((int8_t __attribute__((address_space(1))) *)b)[i] = (int8_t)(r <= -128.0f ? -128 : r >= 127.0f ? 127 : (int)r);
      break;
    }
    case DALIAMD_UINT8: {
      const float r = roundf(v * 255.0f);
      b[i] = (uint8_t)(r <= 0.0f ? 0 : r >= 255.0f ? 255 : (int)r);
      break;
    }
    case DALIAMD_INT16: {
      const float r = roundf(v * 32767.0f);
      ((int16_t __attribute__((address_space(1))) *)b)[i] = (int16_t)(r <= -32768.0f ? -32768 : r >= 32767.0f ? 32767 : (int)r);
      break;
    }
    case DALIAMD_UINT16: {
      const float r = roundf(v * 65535.0f);
      ((uint16_t __attribute__((address_space(1))) *)b)[i] = (uint16_t)(r <= 0.0f ? 0 : r >= 65535.0f ? 65535 : (int)r);
      break;
    }
    case DALIAMD_INT32: {
      const float r = roundf(v * 2147483648.0f);
      ((int32_t __attribute__((address_space(1))) *)b)[i] =
          r <= -2147483648.0f ? (int32_t)0x80000000 : r >= 2147483648.0f ? 2147483647 : (int32_t)r;
      break;
    }
    case DALIAMD_UINT32: {
      const float r = roundf(v * 4294967296.0f);
      ((uint32_t __attribute__((address_space(1))) *)b)[i] = r <= 0.0f ? 0u : r >= 4294967296.0f ? 4294967295u : (uint32_t)r;
      break;
    }
    default: ((float __attribute__((address_space(1))) *)b)[i] = v;
  }
}
__global__ __launch_bounds__(kCvtThreads) void ConvertNormKernel(const daliamdConvertNormDesc *__restrict__ descs, int ndesc,
                                                                 int total_wg, int in_dtype, int out_dtype, int mode) {
  const int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const int di = FindDesc(descs, ndesc, wg);
  const daliamdConvertNormDesc &d = descs[di];
  const int64_t base = (int64_t)(wg - d.wg_start) * kCvtPerWg;
  for (int j = threadIdx.x; j < kCvtPerWg; j += kCvtThreads) {
    const int64_t i = base + j;
    if (i >= d.count) break;
    float f = LoadNorm(d.in, in_dtype, i);
    if (mode == 1) f = (f + 1.0f) * 0.5f;
    else if (mode == 2) f = f * 2.0f - 1.0f;
    StoreNorm(d.out, out_dtype, i, f);
  }
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
  const int fpg = SpecFramesPerWg(p->nfft);
  for (int i = 0; i < n; i++) {
    auto &d = descs[i];
    int64_t len = d.length;
    if (!p->center_windows) len -= p->window_length;
    DALIAMD_REQUIRE(d.length > 0 && len >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdSpectrogramSetup: sample %d is shorter than the window", i);
    d.num_windows = (int32_t)(len / p->window_step + 1);  // extract_windows_args.h:38-43
    d.wg_start = wg;
    wg += (d.num_windows + fpg - 1) / fpg;
  }
  *nwg = wg;
  const int N = p->nfft / 2;
  *lds_bytes = (N + kSpecWaves * N) * (int)sizeof(float2) + (N + 1) * (fpg + 1) * (int)sizeof(float);
  if (SpecFastPath(p->nfft) && SpecFastLds(p->nfft) > *lds_bytes) *lds_bytes = SpecFastLds(p->nfft);
  return DALIAMD_SUCCESS;
}

void daliamdSpectrogramTwiddles(int nfft, float *twiddles) {  // exp(-2 pi i k / nfft), k < nfft / 2, in double
  for (int k = 0; k < nfft / 2; k++) {
    const double a = -2.0 * M_PI * k / nfft;
    twiddles[2 * k] = (float)std::cos(a);
    twiddles[2 * k + 1] = (float)std::sin(a);
  }
}

daliamdResult_t daliamdSpectrogramRun(daliamdStream_t stream, const daliamdSpectrogramDesc *descs_dev, int n,
                                      const daliamdSpectrogramParams *p, const float *window_dev,
                                      const float *twiddles_dev, int nwg, int lds_bytes) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && p && window_dev && n > 0 && nwg > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdSpectrogramRun: invalid argument");
  int log2n = 0;
  while ((2 << log2n) < p->nfft) log2n++;
  dim3 grid(XcdGrid(nwg)), block(kSpecThreads);
  hipStream_t s = (hipStream_t)stream;
  KernelTimer timer("SpectrogramKernel", s);
  if (twiddles_dev && SpecFastPath(p->nfft)) {
    const int lds = SpecFastLds(p->nfft);
    const float2 *tw = reinterpret_cast<const float2 *>(twiddles_dev);
#define SPEC_FAST(L)                                                                                                     \
  {                                                                                                                      \
    auto kern = SpectrogramFastKernel<L, kSpecFastConcurrent>;                                                           \
    DALIAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(kern, grid, block, lds, s, descs_dev, n, nwg, *p, window_dev, tw, MelFuse{});                     \
  }
    if (p->nfft == 512) SPEC_FAST(8) else SPEC_FAST(9)
#undef SPEC_FAST
    DALIAMD_HIP_CHECK(hipGetLastError());
    return DALIAMD_SUCCESS;
  }
  {
    const int N = p->nfft / 2, fpg = SpecFramesPerWg(p->nfft);
    lds_bytes = (N + kSpecWaves * N) * (int)sizeof(float2) + (N + 1) * (fpg + 1) * (int)sizeof(float);
  }
#define SPEC_CASE(L)                                                                                                     \
  case L:                                                                                                                \
    if (lds_bytes > 64 * 1024)                                                                                           \
      DALIAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(SpectrogramKernel<L>),                        \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));                     \
    hipLaunchKernelGGL(SpectrogramKernel<L>, grid, block, lds_bytes, s, descs_dev, n, nwg, *p, window_dev);              \
    break;
  switch (log2n) {
    SPEC_CASE(0) SPEC_CASE(1) SPEC_CASE(2) SPEC_CASE(3) SPEC_CASE(4) SPEC_CASE(5) SPEC_CASE(6) SPEC_CASE(7) SPEC_CASE(8)
    SPEC_CASE(9) SPEC_CASE(10) SPEC_CASE(11)
    default: DALIAMD_REQUIRE(false, DALIAMD_ERROR_UNSUPPORTED, "daliamdSpectrogramRun: unsupported nfft %d", p->nfft);
  }
#undef SPEC_CASE
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

daliamdResult_t daliamdMelFilterBankBands(const float *weights, int nfilter, int nbins, int32_t *bands) {
  DALIAMD_REQUIRE(weights && bands && nfilter > 0 && nbins > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdMelFilterBankBands: invalid argument");
  for (int m = 0; m < nfilter; m++) {
    int lo = nbins, hi = 0;
    for (int k = 0; k < nbins; k++)
      if (weights[(size_t)m * nbins + k] != 0.0f) {
        if (k < lo) lo = k;
        hi = k + 1;
      }
    if (lo > hi) lo = hi = 0;
    bands[2 * m] = lo;
    bands[2 * m + 1] = hi;
  }
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdMelFilterBankRun(daliamdStream_t stream, const daliamdMelDesc *descs_dev, int n, int nwg,
                                        const float *W, const int32_t *bands_dev, int nfilter, int nbins) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && W && nfilter > 0 && nbins > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdMelFilterBankRun: invalid argument");
  {
    daliamd::KernelTimer timer("MelKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(MelKernel, dim3(XcdGrid(nwg)), dim3(kMelThreads), 0, (hipStream_t)stream, descs_dev, n, nwg, W, bands_dev,
                       (const float *)nullptr, nfilter, nbins);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdMelFilterBankMfmaLayout(const float *weights, int nfilter, int nbins, float *tiles, int32_t *row_blocks,
                                               int *num_tiles) {
  DALIAMD_REQUIRE(weights && num_tiles && nfilter > 0 && nbins > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdMelFilterBankMfmaLayout: invalid argument");
  const int nrb = (nfilter + 15) / 16;
  int total = 0;
  for (int mb = 0; mb < nrb; mb++) {
    int lo = nbins, hi = 0;   // bins any of the block's filters weighs
    for (int m = mb * 16; m < std::min(nfilter, mb * 16 + 16); m++)
      for (int k = 0; k < nbins; k++)
        if (weights[(size_t)m * nbins + k] != 0.0f) {
          lo = std::min(lo, k);
          hi = std::max(hi, k + 1);
        }
    if (lo >= hi) lo = hi = 0;
    const int k0 = lo & ~3, count = (hi - k0 + 3) / 4;
    if (row_blocks) {
      row_blocks[4 * mb] = total; row_blocks[4 * mb + 1] = count; row_blocks[4 * mb + 2] = k0; row_blocks[4 * mb + 3] = 0;
    }
    if (tiles)
      for (int t = 0; t < count; t++)
        for (int l = 0; l < 64; l++) {   // A operand of v_mfma_f32_16x16x4_f32: lane l holds A[l % 16][l / 16]
          const int m = mb * 16 + (l & 15), k = k0 + 4 * t + (l >> 4);
          tiles[(size_t)(total + t) * 64 + l] = m < nfilter && k < nbins ? weights[(size_t)m * nbins + k] : 0.0f;
        }
    total += count;
  }
  if (row_blocks) {   // deal the row blocks to the 4 waves of a workgroup: largest first, to the least loaded wave
    std::vector<int> order(nrb);
    for (int i = 0; i < nrb; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return row_blocks[4 * x + 1] > row_blocks[4 * y + 1]; });
    int load[kSpecWaves] = {0, 0, 0, 0};
    for (int mb : order) {
      int w = 0;
      for (int q = 1; q < kSpecWaves; q++)
        if (load[q] < load[w]) w = q;
      row_blocks[4 * mb + 3] = w;
      load[w] += row_blocks[4 * mb + 1];
    }
  }
  *num_tiles = total;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdSpectrogramMelRun(daliamdStream_t stream, const daliamdSpectrogramDesc *descs_dev, int n,
                                         const daliamdSpectrogramParams *p, const float *window_dev, const float *twiddles_dev,
                                         const daliamdSpecMelParams *m, int nwg) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && p && window_dev && twiddles_dev && m && n > 0 && nwg > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdSpectrogramMelRun: invalid argument");
  DALIAMD_REQUIRE(SpecFastPath(p->nfft), DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdSpectrogramMelRun: the fused kernel exists for nfft 512 and 1024, got %d", p->nfft);
  DALIAMD_REQUIRE(m->nbins == p->nfft / 2 + 1 && m->nfilter > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdSpectrogramMelRun: the filter bank is made for %d bins, the transform has %d", m->nbins, p->nfft / 2 + 1);
  const bool mfma = m->mfma_tiles != nullptr;
  DALIAMD_REQUIRE(mfma ? m->row_blocks != nullptr : (m->weights && m->bands), DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdSpectrogramMelRun: filter bank tables missing");
  MelFuse f{};
  f.tiles = m->mfma_tiles; f.row_blocks = m->row_blocks; f.weights = m->weights; f.bands = m->bands;
  f.num_row_blocks = (m->nfilter + 15) / 16; f.nfilter = m->nfilter; f.nbins = m->nbins;
  f.decibels = m->decibels;
  if (m->decibels) {
    DALIAMD_REQUIRE(m->reference > 0.0f, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdSpectrogramMelRun: fused decibels need an explicit reference (the per-sample maximum is only known "
                    "after the launch: pass max_bits and run daliamdToDecibelsRun on the result)");
    f.min_ratio = std::pow(10.0f, m->cutoff_db / m->multiplier);
    if (f.min_ratio == 0.0f) f.min_ratio = std::nextafter(0.0f, 1.0f);
    f.mul_log2 = m->multiplier * 0.3010299956639812f;
    f.inv_ref = m->reference == 1.0f ? 1.0f : 1.0f / m->reference;
  }
  f.max_bits = m->max_bits; f.max_stride = m->max_stride;
  dim3 grid(XcdGrid(nwg)), block(kSpecThreads);
  hipStream_t s = (hipStream_t)stream;
  const int lds = SpecFastLds(p->nfft);
  const float2 *tw = reinterpret_cast<const float2 *>(twiddles_dev);
  KernelTimer timer(mfma ? "SpectrogramMelMfmaKernel" : "SpectrogramMelKernel", s);
#define SPEC_MEL(L, MODE)                                                                                                \
  {                                                                                                                      \
    auto kern = SpectrogramFastKernel<L, kSpecFastConcurrent, MODE>;                                                     \
    DALIAMD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(kern, grid, block, lds, s, descs_dev, n, nwg, *p, window_dev, tw, f);                             \
  }
  if (p->nfft == 512) { if (mfma) SPEC_MEL(8, 1) else SPEC_MEL(8, 2) }
  else { if (mfma) SPEC_MEL(9, 1) else SPEC_MEL(9, 2) }
#undef SPEC_MEL
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdToDecibelsSetup(daliamdDecibelDesc *descs, int n, int *nwg) {
  DALIAMD_REQUIRE(descs && nwg && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdToDecibelsSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    DALIAMD_REQUIRE(descs[i].size >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdToDecibelsSetup: negative size");
    descs[i].max_bits = 0;
    descs[i].wg_start = wg;
    wg += (int)((descs[i].size + kDbChunk - 1) / kDbChunk);
  }
  *nwg = wg;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdToDecibelsRun(daliamdStream_t stream, daliamdDecibelDesc *descs_dev, int n, int nwg, float multiplier,
                                     float reference, float cutoff_db) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && nwg > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdToDecibelsRun: invalid argument");
  float min_ratio = std::pow(10.0f, cutoff_db / multiplier);        // to_decibels_op.h:41-49
  if (min_ratio == 0.0f) min_ratio = std::nextafter(0.0f, 1.0f);
  float mul_log2 = multiplier * 0.3010299956639812f;
  // reference == 0: the per-sample maximum, found here; reference < 0: the maximum is ALREADY in descs_dev[i].max_bits
  // (daliamdSpectrogramMelRun left it there)
  if (reference == 0.0f)
    {
      daliamd::KernelTimer timer("DecibelMaxKernel", (hipStream_t)stream);
      hipLaunchKernelGGL(DecibelMaxKernel, dim3(XcdGrid(nwg)), dim3(kDbThreads), 0, (hipStream_t)stream, descs_dev, n, nwg);
    }
  {
    daliamd::KernelTimer timer("DecibelKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(DecibelKernel, dim3(XcdGrid(nwg)), dim3(kDbThreads), 0, (hipStream_t)stream, descs_dev, n, nwg, mul_log2,
                       reference, min_ratio);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

// ---- DCT (MFCC): out[k][t] = lifter[k] * sum_n table[k][n] * in[n][t] ----
static double DctEntry(int type, bool normalize, int64_t n_in, int64_t k, int64_t n) {  // dct/table.h:26-96
  switch (type) {
    case 1: {
      if (n == 0) return 0.5;
      if (n == n_in - 1) return k % 2 == 0 ? 0.5 : -0.5;
      return std::cos(M_PI / (n_in - 1) * k * n);
    }
    case 2: {
      double f = 1;
      if (normalize) f = k == 0 ? 1.0 / std::sqrt((double)n_in) : std::sqrt(2.0 / n_in);
      return f * std::cos(M_PI / n_in * (n + 0.5) * k);
    }
    case 3: {
      double f0 = 0.5, fi = 1;
      if (normalize) { fi = std::sqrt(2.0 / n_in); f0 = 1.0 / std::sqrt((double)n_in); }
      return n == 0 ? f0 : fi * std::cos(M_PI / n_in * n * (k + 0.5));
    }
    default: {
      double f = normalize ? std::sqrt(2.0 / n_in) : 1.0;
      return f * std::cos(M_PI / n_in * (n + 0.5) * (k + 0.5));
    }
  }
}

daliamdResult_t daliamdDctTable(int dct_type, int normalize, int n_in, int ndct, float *table) {
  DALIAMD_REQUIRE(table && n_in > 0 && ndct > 0 && ndct <= n_in, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdDctTable: invalid argument");
  DALIAMD_REQUIRE(dct_type >= 1 && dct_type <= 4, DALIAMD_ERROR_INVALID_ARGUMENT, "Unsupported DCT type: %d. Supported types are: 1, 2, 3, 4", dct_type);
  DALIAMD_REQUIRE(dct_type != 1 || n_in > 1, DALIAMD_ERROR_INVALID_ARGUMENT, "DCT type I requires an input length > 1");
  if (dct_type == 1) normalize = 0;  // not defined for type I: ignored, like the reference (dct_cpu.cc:48-54)
  for (int k = 0; k < ndct; k++)
    for (int n = 0; n < n_in; n++) table[(size_t)k * n_in + n] = (float)DctEntry(dct_type, normalize != 0, n_in, k, n);
  return DALIAMD_SUCCESS;
}

void daliamdLifterCoeffs(float lifter, int n, float *coeffs) {  // mfcc.h:43-48
  const float ampl_mult = lifter / 2;
  const float phase_mult = static_cast<float>(M_PI) / lifter;
  for (int i = 0; i < n; i++) coeffs[i] = lifter == 0.0f ? 1.0f : 1.f + ampl_mult * std::sin(phase_mult * (i + 1));
}

daliamdResult_t daliamdDctRun(daliamdStream_t stream, const daliamdMelDesc *descs_dev, int n, int nwg, const float *table_dev,
                              const float *lifter_dev, int ndct, int n_in) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && table_dev && ndct > 0 && n_in > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdDctRun: invalid argument");
  {
    daliamd::KernelTimer timer("DctKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(MelKernel, dim3(XcdGrid(nwg)), dim3(kMelThreads), 0, (hipStream_t)stream, descs_dev, n, nwg, table_dev,
                       (const int32_t *)nullptr, lifter_dev, ndct, n_in);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

// ---- audio resampling ----
static double SincD(double x) {  // include/dali/core/math_util.h:179-185
  x *= M_PI;
  if (std::abs(x) < 1e-8) return 1.0 - x * x * (1.0 / 6);
  return std::sin(x) / x;
}
static float SincF(float x) {   // math_util.h:188-194 (the float overload is the one windowed_sinc calls)
  x = (float)(x * M_PI);       // `x *= M_PI`: the product is formed in double
  if (std::abs(x) < 1e-5f) return 1.0f - x * x * (1.0f / 6);
  return std::sin(x) / x;
}

int daliamdAudioResampleLobes(float quality) {  // resampling_params.h:27-30
  return (int)std::round(0.007 * quality * quality - 0.09 * quality + 3);
}

daliamdResult_t daliamdAudioResampleWindow(int lobes, float *lookup, int lookup_capacity, int *lookup_size, float *scale,
                                           float *center) {
  DALIAMD_REQUIRE(lookup && lookup_size && scale && center && lobes > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdAudioResampleWindow: invalid argument");
  const int coeffs = lobes * 64 + 1;  // ResamplingParams::FromQuality
  DALIAMD_REQUIRE(lookup_capacity >= coeffs + 5, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdAudioResampleWindow: %d floats needed",
                  coeffs + 5);
  (void)SincD;
  // windowed_sinc (resampling.h:79-99)
  const float wscale = 2.0f * lobes / (coeffs - 1);
  const float scale_envelope = 2.0f / coeffs;
  for (int i = 0; i < coeffs + 5; i++) lookup[i] = 0.0f;
  const int c = (int)((coeffs - 1) * 0.5f);
  for (int i = 0; i < coeffs; i++) {
    float x = (i - c) * wscale;
    float y = (i - c) * scale_envelope;
    float w = (float)(SincF(x) * (0.5 * (1 + std::cos((double)y * M_PI))));
    lookup[i + 1] = w;
  }
  *lookup_size = coeffs + 5;
  *center = (float)(c + 1);
  *scale = 1 / wscale;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdAudioResampleSetup(daliamdAudioResampleDesc *descs, int n, int *nwg) {
  DALIAMD_REQUIRE(descs && nwg && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdAudioResampleSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    auto &d = descs[i];
    DALIAMD_REQUIRE(d.in_length >= 0 && d.out_length >= 0 && d.channels >= 1, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdAudioResampleSetup: sample %d has an invalid shape", i);
    DALIAMD_REQUIRE(d.in_rate > 0 && d.out_rate > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "Sampling rate must be positive");
    d.wg_start = wg;
    wg += (int)((d.out_length + kRsThreads - 1) / kRsThreads);
  }
  *nwg = wg;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdAudioResampleRun(daliamdStream_t stream, const daliamdAudioResampleDesc *descs_dev, int n, int nwg,
                                        const float *lookup_dev, int lookup_size, float scale, float center, int lobes) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && lookup_dev && lookup_size > 0 && lobes > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdAudioResampleRun: invalid argument");
  const int lds = lookup_size * (int)sizeof(float);
  DALIAMD_REQUIRE(lds <= 64 * 1024, DALIAMD_ERROR_UNSUPPORTED, "daliamdAudioResampleRun: the window table does not fit in LDS");
  {
    daliamd::KernelTimer timer("AudioResampleKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(AudioResampleKernel, dim3(XcdGrid(nwg)), dim3(kRsThreads), lds, (hipStream_t)stream, descs_dev, n, nwg,
                       lookup_dev, lookup_size, scale, center, lobes);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdConvertNormSetup(daliamdConvertNormDesc *descs, int n, int *nwg) {
  DALIAMD_REQUIRE(descs && nwg && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdConvertNormSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    DALIAMD_REQUIRE(descs[i].count >= 0 && (descs[i].count == 0 || (descs[i].in && descs[i].out)), DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdConvertNormSetup: sample %d: NULL buffer or negative count", i);
    descs[i].wg_start = wg;
    wg += (int)((descs[i].count + kCvtPerWg - 1) / kCvtPerWg);
  }
  *nwg = wg;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdConvertNormRun(daliamdStream_t stream, const daliamdConvertNormDesc *descs_dev, int n, int nwg, int in_dtype,
                                      int out_dtype, int mode) {
  if (n == 0 || nwg == 0) return DALIAMD_SUCCESS;
  auto known = [](int t) {
    return t == DALIAMD_INT8 || t == DALIAMD_UINT8 || t == DALIAMD_INT16 || t == DALIAMD_UINT16 || t == DALIAMD_INT32 ||
           t == DALIAMD_UINT32 || t == DALIAMD_FLOAT;
  };
  DALIAMD_REQUIRE(descs_dev && known(in_dtype) && known(out_dtype) && mode >= 0 && mode <= 2, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdConvertNormRun: invalid argument");
  {
    daliamd::KernelTimer timer("ConvertNormKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(ConvertNormKernel, dim3(XcdGrid(nwg)), dim3(kCvtThreads), 0, (hipStream_t)stream, descs_dev, n, nwg, in_dtype,
                       out_dtype, mode);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
