// Stand-alone batched CropMirrorNormalize for gfx950: u8 HWC -> {fp16, fp32, u8, i8} HWC / CHW
// with crop, horizontal mirror, per-channel normalisation, channel padding and out-of-bounds
// fill, in one launch per batch.
//
// Arithmetic = the reference's CPU kernel (dali/kernels/slice/slice_flip_normalize_permute_pad_cpu.h:37-64):
//   out = ConvertSat<Out>((float(in) - mean[c]) * inv_std[c])     (sub, then mul: built with
//   -ffp-contract=off), out-of-bounds / padded-channel elements = fill[c] un-normalised (:100-145),
//   fp16 stores round to nearest with ties away from zero (include/dali/util/half.hpp:231-243),
//   integer stores std::round + clamp (include/dali/core/convert.h:306-321).
// This makes the GPU result bit-identical to the CPU backend for every mean/std, which the
// reference's own GPU fast path (fma(x, inv_std, -mean*inv_std), slice_hwc2chw_normalize_gpu.cu:408)
// only is when std == 1.
//
// Mapping: a workgroup owns 1024 consecutive crop pixels (row-major); each thread handles 4
// consecutive pixels of one row segment: 12 source bytes in (all loads issued before the first
// use), 4 values per output channel plane out (8-byte fp16 stores for CHW).  HBM traffic: C bytes read + Cout*sizeof(Out) written per pixel.
#include "common.h"

namespace daliamd {

constexpr int kCmnThreads = 256;
constexpr int kCmnPxPerThread = 4;
constexpr int kCmnPxPerWg = kCmnThreads * kCmnPxPerThread;

// half_float round-to-nearest, ties away from zero (half.hpp:464-536), every exponent range
__device__ __noinline__ uint16_t F2HAwaySlow(float f) {
  uint32_t bits = __float_as_uint(f);
  uint32_t e = (bits >> 23) & 0xff;
  uint32_t sign = (bits >> 16) & 0x8000;
  uint32_t mant = bits & 0x7FFFFF;
  uint32_t base;
  int shift;
  if (e < 103) { base = 0; shift = 24; }
  else if (e < 113) { base = 0x0400u >> (113 - e); shift = 126 - (int)e; }
  else if (e < 143) { base = (e - 112) << 10; shift = 13; }
  else if (e < 255) { base = 0x7C00; shift = 24; }
  else { base = 0x7C00; shift = 13; }
  uint32_t h = (base | sign) + (mant >> shift);
  uint32_t rnd = ((mant >> (shift - 1)) | (e == 102 ? 1u : 0u)) & ((h & 0x7C00) != 0x7C00 ? 1u : 0u);
  return (uint16_t)(h + rnd);
}

// In the normal half range the ties-away result is the hardware round-to-nearest-even result, plus one unit in the
// last place exactly when the dropped bits are a tie and RNE rounded down (the fused epilogue of resample.hip does the same)
__device__ __forceinline__ uint16_t F2HAway(float f) {
  const uint32_t bits = __float_as_uint(f);
  const uint32_t e = (bits >> 23) & 0xff;
  if (e >= 113 && e < 142) {
    const _Float16 hf = (_Float16)f;  // v_cvt_f16_f32
    const uint16_t h = __builtin_bit_cast(uint16_t, hf);
    const bool tie_down = ((bits & 0x1FFF) == 0x1000) && ((bits & 0x2000) == 0);
    return tie_down ? (uint16_t)(h + 1) : h;
  }
  if ((bits & 0x7fffffff) == 0) return (uint16_t)(bits >> 16);
  return F2HAwaySlow(f);
}

__device__ __forceinline__ float RoundAway(float v) {  // std::round
  float r = truncf(v);
  float d = v - r;
  if (d >= 0.5f) r += 1.0f;
  else if (d <= -0.5f) r -= 1.0f;
  return r;
}

__device__ __forceinline__ void StoreElem(void *out, size_t o, int dtype, float v) {
  switch (dtype) {
    case DALIAMD_FLOAT: reinterpret_cast<float *>(out)[o] = v; break;
    case DALIAMD_FLOAT16: reinterpret_cast<uint16_t *>(out)[o] = F2HAway(v); break;
    case DALIAMD_UINT8: {
      float r = RoundAway(v);
      reinterpret_cast<uint8_t *>(out)[o] = (uint8_t)(!(r > 0.0f) ? 0.0f : fminf(r, 255.0f));
    } break;
    default: {
      float r = RoundAway(v);
      r = r != r ? 0.0f : fminf(fmaxf(r, -128.0f), 127.0f);
      reinterpret_cast<int8_t *>(out)[o] = (int8_t)r;
    } break;
  }
}

__global__ __launch_bounds__(kCmnThreads) void CmnKernel(const daliamdCmnDesc *__restrict__ descs, int ndesc,
                                                         int total_wg) {
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  const daliamdCmnDesc &d = descs[FindDesc(descs, ndesc, wg)];
  const int cw = d.crop_w, ch = d.crop_h, C = d.channels, Co = d.out_channels;
  const int groups_per_row = (cw + kCmnPxPerThread - 1) / kCmnPxPerThread;
  int g = (wg - d.wg_start) * kCmnThreads + threadIdx.x;
  if (g >= groups_per_row * ch) return;
  int y = g / groups_per_row;
  int x0 = (g - y * groups_per_row) * kCmnPxPerThread;
  int npx = min(kCmnPxPerThread, cw - x0);
  int sy = d.anchor_y + y;
  bool row_in = sy >= 0 && sy < d.in_h;
  const uint8_t *row = d.in + (size_t)(row_in ? sy : 0) * d.in_pitch;

  // all the source bytes of the thread first (16 predicated loads in flight, not a load and a wait per element),
  // then the arithmetic
  using GByte = const uint8_t __attribute__((address_space(1)));
  GByte *grow = (GByte *)row;
  uint32_t raw[kCmnPxPerThread][4];
  bool inside[kCmnPxPerThread];
#pragma unroll
  for (int p = 0; p < kCmnPxPerThread; p++) {
    const int x = x0 + p;
    const int sx = d.mirror ? d.anchor_x + (cw - 1 - x) : d.anchor_x + x;
    inside[p] = p < npx && row_in && sx >= 0 && sx < d.in_w;
    GByte *px = grow + (size_t)(inside[p] ? sx : 0) * C;
#pragma unroll
    for (int c = 0; c < 4; c++) raw[p][c] = (inside[p] && c < C) ? px[c] : 0u;
  }
  float mean[4], inv_std[4], fill[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { mean[c] = d.mean[c]; inv_std[c] = d.inv_std[c]; fill[c] = d.fill[c]; }
  const bool norm = d.normalize != 0;
  float vals[4][kCmnPxPerThread];
#pragma unroll
  for (int p = 0; p < kCmnPxPerThread; p++) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float v = (float)raw[p][c];
      if (norm) v = (v - mean[c]) * inv_std[c];
      vals[c][p] = (c < C && inside[p]) ? v : fill[c];
    }
  }
  bool chw = d.out_layout == DALIAMD_LAYOUT_CHW;
  if (chw && d.out_dtype == DALIAMD_FLOAT16 && npx == 4 && ((cw & 3) == 0) &&
      ((reinterpret_cast<uintptr_t>(d.out) & 7) == 0)) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c >= Co) break;
      uint32_t lo = F2HAway(vals[c][0]) | ((uint32_t)F2HAway(vals[c][1]) << 16);
      uint32_t hi = F2HAway(vals[c][2]) | ((uint32_t)F2HAway(vals[c][3]) << 16);
      size_t o = ((size_t)c * ch + y) * cw + x0;
      *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(d.out) + o) = make_uint2(lo, hi);
    }
  } else {
#pragma unroll
    for (int p = 0; p < kCmnPxPerThread; p++) {
      if (p >= npx) break;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if (c >= Co) break;
        size_t o = chw ? ((size_t)c * ch + y) * cw + (x0 + p) : ((size_t)y * cw + (x0 + p)) * Co + c;
        StoreElem(d.out, o, d.out_dtype, vals[c][p]);
      }
    }
  }
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdCmnSetup(daliamdCmnDesc *descs, int n, int *num_workgroups) {
  DALIAMD_REQUIRE(descs && num_workgroups && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdCmnSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    auto &d = descs[i];
    DALIAMD_REQUIRE(d.channels >= 1 && d.channels <= 4 && d.out_channels >= d.channels && d.out_channels <= 4,
                    DALIAMD_ERROR_UNSUPPORTED, "daliamdCmnSetup: sample %d: unsupported channel count %d -> %d",
                    i, d.channels, d.out_channels);
    DALIAMD_REQUIRE(d.crop_h >= 0 && d.crop_w >= 0 && d.in_h >= 0 && d.in_w >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdCmnSetup: sample %d: negative extent", i);
    DALIAMD_REQUIRE(d.out_dtype >= DALIAMD_UINT8 && d.out_dtype <= DALIAMD_INT8, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdCmnSetup: sample %d: unsupported output type %d", i, d.out_dtype);
    d.wg_start = wg;
    int64_t groups = (int64_t)((d.crop_w + daliamd::kCmnPxPerThread - 1) / daliamd::kCmnPxPerThread) * d.crop_h;
    wg += (int)((groups + daliamd::kCmnThreads - 1) / daliamd::kCmnThreads);
  }
  *num_workgroups = wg;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdCmnRun(daliamdStream_t stream, const daliamdCmnDesc *descs_dev, int n, int num_workgroups) {
  if (n == 0 || num_workgroups == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_workgroups > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdCmnRun: invalid argument");
  {
    daliamd::KernelTimer timer("CmnKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(daliamd::CmnKernel, dim3(daliamd::XcdGrid(num_workgroups)), dim3(daliamd::kCmnThreads), 0,
                       (hipStream_t)stream, descs_dev, n, num_workgroups);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
