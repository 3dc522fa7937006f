// Issue rate of the f32 vector instructions the image kernels are made of (gfx950): cycles per wave instruction with
// 1 / 2 / 4 waves per SIMD, 8 independent accumulators per wave (no dependent-issue stalls).  Development tool:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/microbench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float floatx2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2048, kAcc = 8;

#define CHAIN(NAME, TYPE, BODY)                                                                     \
  __global__ void NAME(float *out, long long *cycles, float wf) {                                 \
    TYPE a[kAcc];                                                                                  \
    for (int i = 0; i < kAcc; i++) a[i] = TYPE(threadIdx.x * 0.001f + i);                          \
    TYPE w = TYPE(wf), c = TYPE(wf * 0.5f);                                                       \
    long long t0 = clock64();                                                                      \
    for (int it = 0; it < kIters; it++) {                                                          \
      _Pragma("unroll") for (int i = 0; i < kAcc; i++) { BODY; }                                   \
    }                                                                                              \
    long long t1 = clock64();                                                                      \
    float s = 0;                                                                                   \
    for (int i = 0; i < kAcc; i++) s += Sum(a[i]);                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;                                            \
  }
__device__ inline float Sum(float x) { return x; }
__device__ inline float Sum(floatx2 x) { return x.x + x.y; }

CHAIN(k_mul, float, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w)))
CHAIN(k_add, float, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)))
CHAIN(k_fma, float, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(c)))
CHAIN(k_pk_mul, floatx2, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w)))
CHAIN(k_pk_add, floatx2, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)))
CHAIN(k_pk_fma, floatx2, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(c)))
CHAIN(k_cvt_ubyte, float, asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i])))

template <typename K>
static void Run(const char *name, K kernel, int flops_per_lane_instr) {
  float *out;
  long long *cyc;
  hipMalloc(&out, 1024 * 1024 * 4);
  hipMalloc(&cyc, 4096 * 8);
  printf("%-18s", name);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 64 * 4 * waves_per_simd;     // one workgroup per CU, 4 SIMDs
    const int blocks = 256;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    // clock64 = s_memtime ticks at a constant 100 MHz on this part; the event time gives the wall rate
    const double instr_per_simd = (double)kIters * kAcc * waves_per_simd;
    const double tflops = (double)blocks * 4 * instr_per_simd * 64 * flops_per_lane_instr / (ms * 1e-3) / 1e12;
    printf("  %d w/SIMD: %7.3f ms %8.1f ns/instr/SIMD %6.1f TFLOP/s", waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, tflops);
    (void)mean;
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  Run("v_mul_f32", k_mul, 1);
  Run("v_add_f32", k_add, 1);
  Run("v_fma_f32", k_fma, 2);
  Run("v_pk_mul_f32", k_pk_mul, 2);
  Run("v_pk_add_f32", k_pk_add, 2);
  Run("v_pk_fma_f32", k_pk_fma, 4);
  Run("v_cvt_f32_ubyte0", k_cvt_ubyte, 1);
  return 0;
}
