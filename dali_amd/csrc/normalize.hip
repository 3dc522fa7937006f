// fn.normalize on gfx950: mean / standard deviation over a contiguous group of axes + (x - mean) * scale / stddev + shift.
//
// Reference: dali/operators/math/normalize/{normalize.h,normalize.cc,normalize_gpu.cu,normalize_utils.h},
// kernels dali/kernels/reduce/{mean_stddev_gpu_impl.cuh,reduce_axes_gpu_impl.cuh},
// dali/kernels/normalize/{normalize_cpu.h:38-70,normalize_gpu_impl.cuh:180-191}.
//
// A sample is viewed as [outer][reduced][inner]: the statistics are taken over `reduced` for every (outer, inner)
// pair - per-channel statistics of an HWC image are (1, H*W, C), per-row statistics (outer, W, 1), a full reduction
// (1, N, 1).  With batch normalisation all samples share one set of accumulators.
//
//   NormalizeStatsKernel<VAR>  partial sums (VAR: of squared deviations from the mean) in fp64, one fp64 atomic per
//                              workgroup and bin.  inner == 1 (full reductions, rows): wave64 __shfl_xor butterfly;
//                              narrow inner (channels): the lane stride is a multiple of `inner`, so a lane always
//                              meets the same channel and keeps ONE accumulator; wide inner (statistics per
//                              frequency bin over time): one lane per column, coalesced rows.  u8 samples with at
//                              most 4 channels (decoded images) take 16 bytes per lane and step: the lane stride is a
//                              multiple of 16 * inner, so byte j of a lane's group always belongs to the same channel;
//                              the mean pass sums in integers, the per-channel totals leave a wave by __shfl_xor
//   NormalizeFinalizeKernel    sum -> mean, or sum of squares -> scale / sqrt(var + eps) (0 where the variance is 0,
//                              like ScaleRSqrtKeepZero, normalize_utils.h:133-192, with an exact square root)
//   NormalizeApplyKernel       4 elements per lane and step (one 4-byte load, one 16-byte store for u8 -> float),
//                              kApplySteps steps 1024 elements apart; (outer, inner) indices advance by additions -
//                              one division per lane, not per element: ConvertSat((x - mean) * inv_stddev + shift)
// u8 sums are exact in fp64; fp32 inputs are accumulated in fp64 too, so the result does not depend on the launch
// geometry beyond the last bit of the final fp32 rounding.
#include "common.h"

namespace daliamd {

constexpr int kNormThreads = 256;
constexpr int kStatRowsPerWg = 64;       // wide-inner path: rows of `reduced` per workgroup
constexpr int kStatElemsPerWg = 1 << 16; // narrow-inner path: elements per workgroup
constexpr int kApplySteps = 8;           // apply pass: steps of 4 * kNormThreads elements per workgroup
constexpr int kApplyElemsPerWg = kApplySteps * 4 * kNormThreads;

__device__ __forceinline__ float LoadAsFloat(const void *p, int64_t i, int dtype) {
  return dtype == DALIAMD_UINT8 ? (float)static_cast<const uint8_t *>(p)[i] : static_cast<const float *>(p)[i];
}

// u8 input, inner <= 4, 16-byte aligned: [e0, e1) of one (sample, outer) slab.  kNormThreads / INNER * INNER lanes, 16
// bytes per lane and step; lanes * 16 is a multiple of INNER, so the channel of byte j of a lane's group never changes:
// rel[k] collects the bytes with j % INNER == k, and the lane's bin for rel[k] is (p + k) % INNER, p = the channel of the
// lane's first byte.  Mean: exact integer sums (a workgroup adds at most 65536 * 255).  Variance: fp64, as everywhere.
template <bool VAR, int INNER>
__device__ __forceinline__ void StatsU8Narrow(const uint8_t *__restrict__ in, int64_t e0, int64_t e1, const float *mean,
                                              float scalar_mean, double *sums, double (*wave_bins)[4]) {
  constexpr int lanes = (kNormThreads / INNER) * INNER;
  const int tid = threadIdx.x;
  const int64_t ngroups = (e1 - e0) >> 4;
  const int p = (int)((e0 + (int64_t)tid * 16) % INNER);
  float mrel[INNER];
  uint32_t isum[INNER];
  double dsum[INNER];
#pragma unroll
  for (int k = 0; k < INNER; k++) {
    mrel[k] = VAR ? (mean ? mean[(p + k) % INNER] : scalar_mean) : 0.0f;
    isum[k] = 0;
    dsum[k] = 0;
  }
  if (tid < lanes) {
    const uint4 *src = reinterpret_cast<const uint4 *>(in + e0);
    uint4 next = tid < ngroups ? src[tid] : make_uint4(0, 0, 0, 0);
    for (int64_t g = tid; g < ngroups; g += lanes) {
      const uint4 v = next;
      if (g + lanes < ngroups) next = src[g + lanes];  // the next group is in flight while this one is summed
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t x = (w[j >> 2] >> (8 * (j & 3))) & 255u;
        if (VAR) {
          const float dx = (float)x - mrel[j % INNER];
          dsum[j % INNER] += (double)dx * (double)dx;
        } else {
          isum[j % INNER] += x;
        }
      }
    }
  }
  // the last (e1 - e0) % 16 elements: one per lane, into the accumulator whose bin it is
  const int ntail = (int)((e1 - e0) & 15);
  if (tid < ntail) {
    const int64_t e = e0 + (ngroups << 4) + tid;
    const uint32_t x = in[e];
    const int k = ((int)(e % INNER) - p + INNER) % INNER;
#pragma unroll
    for (int kk = 0; kk < INNER; kk++) {
      if (kk == k) {
        if (VAR) { const float dx = (float)x - mrel[kk]; dsum[kk] += (double)dx * (double)dx; }
        else isum[kk] += x;
      }
    }
  }
#pragma unroll
  for (int b = 0; b < INNER; b++) {
    const int k = (b - p + INNER) % INNER;  // the lane's accumulator of bin b
    if (VAR) {
      double a = dsum[0];
#pragma unroll
      for (int kk = 1; kk < INNER; kk++) a = kk == k ? dsum[kk] : a;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
      if ((tid & 63) == 0) wave_bins[tid >> 6][b] = a;
    } else {
      uint32_t a = isum[0];
#pragma unroll
      for (int kk = 1; kk < INNER; kk++) a = kk == k ? isum[kk] : a;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a += (uint32_t)__shfl_xor((int)a, off, 64);
      if ((tid & 63) == 0) wave_bins[tid >> 6][b] = (double)a;
    }
  }
  __syncthreads();
  if (tid < INNER) {
    double t = 0;
    for (int w = 0; w < kNormThreads / 64; w++) t += wave_bins[w][tid];
    if (t != 0.0) atomicAdd(&sums[tid], t);
  }
}

// grid: descriptors sorted by stat_wg_start; workgroup -> (sample, outer index, chunk of the reduced extent)
template <bool VAR>
__global__ __launch_bounds__(kNormThreads) void NormalizeStatsKernel(const daliamdNormalizeDesc *__restrict__ descs, int n) {
  __shared__ double part[kNormThreads];
  __shared__ double wave_part[kNormThreads / 64];
  __shared__ double wave_bins[kNormThreads / 64][4];
  int lo = 0, hi = n - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].stat_wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  const daliamdNormalizeDesc &d = descs[lo];
  const int local = wg - d.stat_wg_start;
  const int o = local / d.stat_chunks, chunk = local - o * d.stat_chunks;
  const int tid = threadIdx.x;
  const int64_t inner = d.inner, reduced = d.reduced;
  const uint8_t *base = static_cast<const uint8_t *>(d.in);
  const int esz = d.in_dtype == DALIAMD_UINT8 ? 1 : 4;
  const void *in = base + (size_t)o * reduced * inner * esz;
  const float *mean = d.use_scalar_mean ? nullptr : d.mean + (size_t)o * inner;
  double *sums = (VAR ? d.sum_var : d.sum_mean) + (size_t)o * inner;
  if (d.in_dtype == DALIAMD_UINT8 && inner <= 4 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
    // decoded images (and u8 rows): 16 bytes per lane and step
    const int64_t total = reduced * inner;
    const int64_t e0 = (int64_t)chunk * kStatElemsPerWg, e1 = min(e0 + kStatElemsPerWg, total);
    const uint8_t *src = static_cast<const uint8_t *>(in);
    switch ((int)inner) {
      case 1: StatsU8Narrow<VAR, 1>(src, e0, e1, mean, d.scalar_mean, sums, wave_bins); break;
      case 2: StatsU8Narrow<VAR, 2>(src, e0, e1, mean, d.scalar_mean, sums, wave_bins); break;
      case 3: StatsU8Narrow<VAR, 3>(src, e0, e1, mean, d.scalar_mean, sums, wave_bins); break;
      default: StatsU8Narrow<VAR, 4>(src, e0, e1, mean, d.scalar_mean, sums, wave_bins); break;
    }
  } else if (inner == 1) {
    // full reductions and rows: all lanes of the workgroup share ONE bin -> wave64 shuffle (butterfly) reduction
    const int64_t e0 = (int64_t)chunk * kStatElemsPerWg, e1 = min(e0 + kStatElemsPerWg, reduced);
    const float m = VAR ? (d.use_scalar_mean ? d.scalar_mean : d.mean[o]) : 0.0f;
    double acc = 0;
    for (int64_t e = e0 + tid; e < e1; e += kNormThreads) {
      float x = LoadAsFloat(in, e, d.in_dtype);
      if (VAR) { float dx = x - m; acc += (double)dx * (double)dx; } else acc += (double)x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((tid & 63) == 0) wave_part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double s = 0;
      for (int w = 0; w < kNormThreads / 64; w++) s += wave_part[w];
      if (s != 0.0) atomicAdd(&sums[0], s);
    }
  } else if (inner <= kNormThreads) {
    // narrow inner extent (channels): lanes stride by a multiple of `inner`, so lane t only ever sees bin t % inner
    const int lanes = (kNormThreads / (int)inner) * (int)inner;
    const int64_t total = reduced * inner;
    const int64_t e0 = (int64_t)chunk * kStatElemsPerWg, e1 = min(e0 + kStatElemsPerWg, total);
    double acc = 0;
    if (tid < lanes) {
      const int bin = (int)((e0 + tid) % inner);
      const float m = VAR ? (d.use_scalar_mean ? d.scalar_mean : mean[bin]) : 0.0f;
      for (int64_t e = e0 + tid; e < e1; e += lanes) {
        float x = LoadAsFloat(in, e, d.in_dtype);
        if (VAR) { float dx = x - m; acc += (double)dx * (double)dx; } else acc += (double)x;
      }
    }
    part[tid] = tid < lanes ? acc : 0.0;
    __syncthreads();
    // lanes t, t + inner, t + 2*inner, ... hold the same bin; the chunk start e0 need not be a multiple of inner, so
    // the bin of lane t is (e0 + t) % inner: find the first lane of bin `tid`, then gather with stride inner
    if (tid < inner) {
      double s = 0;
      const int first = (int)(((int64_t)tid - e0 % inner + inner) % inner);  // first lane whose bin is `tid`
      for (int t = first; t < lanes; t += (int)inner) s += part[t];
      if (s != 0.0) atomicAdd(&sums[tid], s);
    }
  } else {
    // wide inner extent (e.g. statistics per frequency bin over time): one lane per column, coalesced across lanes
    const int col_tiles = (int)((inner + kNormThreads - 1) / kNormThreads);
    const int row_chunk = chunk / col_tiles, tile = chunk - row_chunk * col_tiles;
    const int64_t i = (int64_t)tile * kNormThreads + tid;
    const int64_t r0 = (int64_t)row_chunk * kStatRowsPerWg, r1 = min(r0 + kStatRowsPerWg, reduced);
    if (i < inner) {
      const float m = VAR ? (d.use_scalar_mean ? d.scalar_mean : mean[i]) : 0.0f;
      double acc = 0;
      for (int64_t r = r0; r < r1; r++) {
        float x = LoadAsFloat(in, r * inner + i, d.in_dtype);
        if (VAR) { float dx = x - m; acc += (double)dx * (double)dx; } else acc += (double)x;
      }
      if (acc != 0.0) atomicAdd(&sums[i], acc);
    }
  }
}

template <bool VAR>
__global__ void NormalizeFinalizeKernel(const daliamdNormalizeDesc *__restrict__ descs, int n, float epsilon, float scale,
                                        int ddof) {
  const daliamdNormalizeDesc &d = descs[blockIdx.y];
  if (!d.owns_stats) return;  // batch normalisation: the first sample finalises the shared accumulators
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.outer * d.inner) return;
  if (!VAR) {
    d.mean[p] = (float)(d.sum_mean[p] * (d.stat_count > 0 ? 1.0 / d.stat_count : 0.0));
  } else {
    // FoldStdDev / SumSquare2InvStdDev (normalize.cc:223-244, normalize_utils.h:199-220)
    float rdiv = 0, mul = scale;
    if (d.stat_count > ddof) rdiv = (float)(1.0 / (d.stat_count - ddof));
    else if (epsilon == 0) { rdiv = 1; mul = 0; }
    float x = (float)d.sum_var[p] * rdiv + epsilon;
    d.inv_std[p] = x != 0 ? mul / sqrtf(x) : 0.0f;
  }
}

__device__ __forceinline__ float SatRound(float v, float lo, float hi) { return fminf(fmaxf(rintf(v), lo), hi); }

__device__ __forceinline__ float Sel4(const float (&a)[4], int i) {
  float r = a[0];
  r = i == 1 ? a[1] : r;
  r = i == 2 ? a[2] : r;
  r = i == 3 ? a[3] : r;
  return r;
}

// One workgroup: kApplyElemsPerWg consecutive elements; step k of lane t covers the 4 elements from k * 1024 + 4 * t.
// Element e = (o * reduced + r') * inner + i has the statistics of p = o * inner + i: (o, e % plane, i) of the lane's
// first element come from one division, every later one from additions (a step advances e by 1024, an element by 1).
__global__ __launch_bounds__(kNormThreads) void NormalizeApplyKernel(const daliamdNormalizeDesc *__restrict__ descs, int n,
                                                                     float shift) {
  int lo = 0, hi = n - 1;
  const int wg = blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].apply_wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  const daliamdNormalizeDesc &d = descs[lo];
  const int64_t total = d.outer * d.reduced * d.inner;
  const int64_t plane = d.reduced * d.inner, inner = d.inner;
  constexpr int kStep = 4 * kNormThreads;
  int64_t e = (int64_t)(wg - d.apply_wg_start) * kApplyElemsPerWg + threadIdx.x * 4;
  if (e >= total) return;
  // lane state: outer index, offset inside the plane, inner index - and what one step adds to each
  int64_t o = e / plane, r = e - o * plane, i = r % inner;
  const int64_t step_o = kStep / plane, step_r = kStep - step_o * plane, step_i = kStep % inner;
  const bool few = d.outer == 1 && inner <= 4;  // decoded images: the statistics live in registers
  float fm[4], fs[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int cc = c < inner ? c : 0;
    fm[c] = few ? (d.use_scalar_mean ? d.scalar_mean : d.mean[cc]) : 0.0f;
    fs[c] = few ? (d.use_scalar_inv_std ? d.scalar_inv_std : d.inv_std[cc]) : 0.0f;
  }
  const bool in_u8 = d.in_dtype == DALIAMD_UINT8;
  const bool wide_in = (reinterpret_cast<uintptr_t>(d.in) & (in_u8 ? 3 : 15)) == 0;
  const int osz = d.out_dtype == DALIAMD_FLOAT ? 4 : d.out_dtype == DALIAMD_FLOAT16 ? 2 : 1;
  const bool wide_out = (reinterpret_cast<uintptr_t>(d.out) & (4 * osz - 1)) == 0;
  for (int k = 0; k < kApplySteps; k++) {
    if (e >= total) return;
    const int cnt = total - e >= 4 ? 4 : (int)(total - e);
    float x[4] = {0, 0, 0, 0};
    if (cnt == 4 && wide_in) {
      if (in_u8) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(d.in) + e);
        x[0] = (float)(w & 255u); x[1] = (float)((w >> 8) & 255u); x[2] = (float)((w >> 16) & 255u); x[3] = (float)(w >> 24);
      } else {
        const float4 w = *reinterpret_cast<const float4 *>(static_cast<const float *>(d.in) + e);
        x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
      }
    } else {
      for (int j = 0; j < cnt; j++) x[j] = LoadAsFloat(d.in, e + j, d.in_dtype);
    }
    float v[4];
    int64_t oj = o, rj = r, ij = i;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float m, sc;
      if (few) {
        m = Sel4(fm, (int)ij);
        sc = Sel4(fs, (int)ij);
      } else {
        const int64_t p = oj * inner + ij;
        const bool live = j < cnt;
        m = d.use_scalar_mean ? d.scalar_mean : (live ? d.mean[p] : 0.0f);
        sc = d.use_scalar_inv_std ? d.scalar_inv_std : (live ? d.inv_std[p] : 0.0f);
      }
      v[j] = (x[j] - m) * sc + shift;
      if (++ij == inner) ij = 0;
      if (++rj == plane) { rj = 0; oj++; }
    }
    if (cnt == 4 && wide_out) {
      switch (d.out_dtype) {
        case DALIAMD_FLOAT:
          *reinterpret_cast<float4 *>(static_cast<float *>(d.out) + e) = make_float4(v[0], v[1], v[2], v[3]);
          break;
        case DALIAMD_FLOAT16: {
          typedef _Float16 half4 __attribute__((ext_vector_type(4)));
          half4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          *reinterpret_cast<half4 *>(static_cast<_Float16 *>(d.out) + e) = h;
          break;
        }
        case DALIAMD_UINT8: {
          uint32_t w = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) w |= (uint32_t)(uint8_t)SatRound(v[j], 0.0f, 255.0f) << (8 * j);
          *reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d.out) + e) = w;
          break;
        }
        default: {
          uint32_t w = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) w |= (uint32_t)(uint8_t)(int8_t)SatRound(v[j], -128.0f, 127.0f) << (8 * j);
          *reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d.out) + e) = w;
          break;
        }
      }
    } else {
      for (int j = 0; j < cnt; j++) {
        switch (d.out_dtype) {
          case DALIAMD_FLOAT: static_cast<float *>(d.out)[e + j] = v[j]; break;
          case DALIAMD_FLOAT16: static_cast<_Float16 *>(d.out)[e + j] = (_Float16)v[j]; break;
          case DALIAMD_UINT8: static_cast<uint8_t *>(d.out)[e + j] = (uint8_t)SatRound(v[j], 0.0f, 255.0f); break;
          default: static_cast<int8_t *>(d.out)[e + j] = (int8_t)SatRound(v[j], -128.0f, 127.0f); break;
        }
      }
    }
    e += kStep;
    o += step_o;
    r += step_r;
    if (r >= plane) { r -= plane; o++; }
    i += step_i;
    if (i >= inner) i -= inner;
  }
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdNormalizeSetup(daliamdNormalizeDesc *descs, int n, int *stat_workgroups, int *apply_workgroups,
                                      int64_t *max_bins) {
  DALIAMD_REQUIRE(n >= 0 && (n == 0 || descs) && stat_workgroups && apply_workgroups && max_bins,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdNormalizeSetup: NULL argument");
  int sw = 0, aw = 0;
  int64_t bins = 0;
  for (int i = 0; i < n; i++) {
    daliamdNormalizeDesc &d = descs[i];
    DALIAMD_REQUIRE(d.in && d.out && d.outer >= 1 && d.reduced >= 1 && d.inner >= 1, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdNormalizeSetup: sample %d: NULL buffer or empty extent", i);
    DALIAMD_REQUIRE(d.in_dtype == DALIAMD_UINT8 || d.in_dtype == DALIAMD_FLOAT, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdNormalizeSetup: sample %d: input must be uint8 or float", i);
    DALIAMD_REQUIRE(d.out_dtype >= DALIAMD_UINT8 && d.out_dtype <= DALIAMD_INT8, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdNormalizeSetup: sample %d: unsupported output type", i);
    DALIAMD_REQUIRE(d.outer * d.inner < ((int64_t)1 << 31) && d.outer * d.reduced * d.inner < ((int64_t)1 << 40),
                    DALIAMD_ERROR_OUT_OF_RANGE, "daliamdNormalizeSetup: sample %d is too large", i);
    int64_t chunks;
    if (d.inner == 1) chunks = (d.reduced + daliamd::kStatElemsPerWg - 1) / daliamd::kStatElemsPerWg;
    else if (d.inner <= daliamd::kNormThreads)
      chunks = (d.reduced * d.inner + daliamd::kStatElemsPerWg - 1) / daliamd::kStatElemsPerWg;
    else
      chunks = ((d.reduced + daliamd::kStatRowsPerWg - 1) / daliamd::kStatRowsPerWg) *
               ((d.inner + daliamd::kNormThreads - 1) / daliamd::kNormThreads);
    DALIAMD_REQUIRE(chunks * d.outer < ((int64_t)1 << 30), DALIAMD_ERROR_OUT_OF_RANGE, "daliamdNormalizeSetup: grid too large");
    d.stat_chunks = (int32_t)chunks;
    d.stat_wg_start = sw;
    d.apply_wg_start = aw;
    sw += (int)(chunks * d.outer);
    aw += (int)((d.outer * d.reduced * d.inner + daliamd::kApplyElemsPerWg - 1) / daliamd::kApplyElemsPerWg);
    bins = bins > d.outer * d.inner ? bins : d.outer * d.inner;
  }
  *stat_workgroups = sw;
  *apply_workgroups = aw;
  *max_bins = bins;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdNormalizeRun(daliamdStream_t stream, const daliamdNormalizeDesc *descs_dev, int n, int stat_workgroups,
                                    int apply_workgroups, int64_t max_bins, int calc_mean, int calc_stddev, int ddof,
                                    float epsilon, float scale, float shift) {
  if (n == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && apply_workgroups > 0 && max_bins > 0 &&
                      (stat_workgroups > 0 || (!calc_mean && !calc_stddev)),
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdNormalizeRun: invalid argument");
  using namespace daliamd;
  hipStream_t s = (hipStream_t)stream;
  const dim3 fin((unsigned)((max_bins + 255) / 256), (unsigned)n);
  if (calc_mean) {
    {
      daliamd::KernelTimer timer("NormalizeStatsKernel", s);
      hipLaunchKernelGGL(NormalizeStatsKernel<false>, dim3(stat_workgroups), dim3(kNormThreads), 0, s, descs_dev, n);
    }
    {
      daliamd::KernelTimer timer("NormalizeFinalizeKernel", s);
      hipLaunchKernelGGL(NormalizeFinalizeKernel<false>, fin, dim3(256), 0, s, descs_dev, n, epsilon, scale, ddof);
    }
  }
  if (calc_stddev) {
    {
      daliamd::KernelTimer timer("NormalizeStatsKernel", s);
      hipLaunchKernelGGL(NormalizeStatsKernel<true>, dim3(stat_workgroups), dim3(kNormThreads), 0, s, descs_dev, n);
    }
    {
      daliamd::KernelTimer timer("NormalizeFinalizeKernel", s);
      hipLaunchKernelGGL(NormalizeFinalizeKernel<true>, fin, dim3(256), 0, s, descs_dev, n, epsilon, scale, ddof);
    }
  }
  {
    daliamd::KernelTimer timer("NormalizeApplyKernel", s);
    hipLaunchKernelGGL(NormalizeApplyKernel, dim3(apply_workgroups), dim3(kNormThreads), 0, s, descs_dev, n, shift);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
