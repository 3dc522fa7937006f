// Self-test of the hipemu execution model: barriers, LDS, wave collectives with and without divergence, atomics, MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void ScanKernel(const int *in, int *out, int n) {
  __shared__ int wave_sums[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gi = blockIdx.x * blockDim.x + tid;
  int v = gi < n ? in[gi] : 0;
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += wave_sums[w];
  __syncthreads();
  if (gi < n) out[gi] = base + incl - v;
}

__global__ void DivergentKernel(uint64_t *ballots, int *sums, int *firsts) {
  const int tid = threadIdx.x, lane = tid & 63;
  // lanes leave a loop at different times; the shuffles inside see the lanes still in it
  int acc = 0;
  for (int i = 0; i < (lane & 7) + 1; i++) {
    uint64_t m = __ballot(1);
    acc += __popcll(m);
  }
  sums[blockIdx.x * blockDim.x + tid] = acc;
  if (lane & 1) {
    ballots[blockIdx.x * blockDim.x + tid] = __ballot(lane & 2);
    firsts[blockIdx.x * blockDim.x + tid] = (int)__builtin_amdgcn_readfirstlane((uint32_t)lane);
  } else {
    ballots[blockIdx.x * blockDim.x + tid] = __ballot(lane & 4);
    firsts[blockIdx.x * blockDim.x + tid] = (int)__builtin_amdgcn_readfirstlane((uint32_t)lane + 100);
  }
  if (tid >= 70) return;   // early exit before a barrier
  __syncthreads();
}

__global__ void XorReduceKernel(const float *in, float *out, unsigned *counter, unsigned *maxv) {
  float a = in[blockIdx.x * blockDim.x + threadIdx.x];
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = a;
  atomicAdd(counter, 1u);
  atomicMax(maxv, blockIdx.x * 1000u + threadIdx.x);
}

__global__ void DynLdsKernel(int *out, int n) {
  HIP_DYNAMIC_SHARED(int, lds)
  lds[threadIdx.x] = threadIdx.x * 3 + blockIdx.x;
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = lds[blockDim.x - 1 - threadIdx.x];
  (void)n;
}

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void MfmaKernel(const float *A, const float *B, float *D) {   // A 16x4 row-major, B 4x16 row-major
  const int l = threadIdx.x;
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int v = 0; v < 4; v++) D[(4 * (l >> 4) + v) * 16 + (l & 15)] = c[v];
}

__global__ void Dim3Kernel(int *out) {
  const int t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  out[b * (blockDim.x * blockDim.y * blockDim.z) + t] = (int)__lane_id() + 64 * b;
}

#define CHECK(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  {  // scan
    const int n = 5000, threads = 512, blocks = (n + threads - 1) / threads;
    std::vector<int> in(n), out(n);
    for (int i = 0; i < n; i++) in[i] = (i * 7919) % 13;
    hipLaunchKernelGGL(ScanKernel, dim3(blocks), dim3(threads), 0, nullptr, in.data(), out.data(), n);
    for (int b = 0; b < blocks; b++) {
      int s = 0;
      for (int i = b * threads; i < std::min(n, (b + 1) * threads); i++) { CHECK(out[i] == s); s += in[i]; }
    }
  }
  {  // divergence
    const int threads = 128, blocks = 3;
    std::vector<uint64_t> ballots(threads * blocks);
    std::vector<int> sums(threads * blocks), firsts(threads * blocks);
    hipLaunchKernelGGL(DivergentKernel, dim3(blocks), dim3(threads), 0, nullptr, ballots.data(), sums.data(), firsts.data());
    for (int t = 0; t < threads * blocks; t++) {
      const int lane = t & 63;
      int expect = 0;
      for (int i = 0; i < (lane & 7) + 1; i++) expect += 8 * (8 - i);  // lanes with (lane & 7) >= i are still looping
      CHECK(sums[t] == expect);
      uint64_t eb = 0;
      for (int l = 0; l < 64; l++)
        if ((l & 1) == (lane & 1) && (l & ((lane & 1) ? 2 : 4))) eb |= 1ull << l;
      CHECK(ballots[t] == eb);
      CHECK(firsts[t] == ((lane & 1) ? 1 : 100));
    }
  }
  {  // butterfly + atomics over several OS threads
    const int threads = 256, blocks = 40;
    std::vector<float> in(threads * blocks), out(blocks * 4);
    for (size_t i = 0; i < in.size(); i++) in[i] = (float)(i % 17);
    unsigned counter = 0, maxv = 0;
    hipLaunchKernelGGL(XorReduceKernel, dim3(blocks), dim3(threads), 0, nullptr, in.data(), out.data(), &counter, &maxv);
    for (int w = 0; w < blocks * 4; w++) {
      float s = 0;
      for (int l = 0; l < 64; l++) s += in[w * 64 + l];
      CHECK(out[w] == s);
    }
    CHECK(counter == (unsigned)(threads * blocks));
    CHECK(maxv == (blocks - 1) * 1000u + threads - 1);
  }
  {  // dynamic LDS
    const int threads = 192, blocks = 5;
    std::vector<int> out(threads * blocks);
    hipLaunchKernelGGL(DynLdsKernel, dim3(blocks), dim3(threads), threads * sizeof(int), nullptr, out.data(), 0);
    for (int b = 0; b < blocks; b++)
      for (int t = 0; t < threads; t++) CHECK(out[b * threads + t] == (threads - 1 - t) * 3 + b);
  }
  {  // MFMA
    float A[64], B[64], D[256];
    for (int i = 0; i < 64; i++) { A[i] = (float)(i % 7) - 3; B[i] = (float)(i % 5) * 0.5f; }
    hipLaunchKernelGGL(MfmaKernel, dim3(1), dim3(64), 0, nullptr, A, B, D);
    for (int i = 0; i < 16; i++)
      for (int j = 0; j < 16; j++) {
        float s = 0;
        for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j];
        CHECK(D[i * 16 + j] == s);
      }
  }
  {  // 3-D launch
    std::vector<int> out(2 * 3 * 2 * 8 * 4 * 3);
    hipLaunchKernelGGL(Dim3Kernel, dim3(2, 3, 2), dim3(8, 4, 3), 0, nullptr, out.data());
    for (int b = 0; b < 12; b++)
      for (int t = 0; t < 96; t++) CHECK(out[b * 96 + t] == (t & 63) + 64 * b);
  }
  printf("hipemu selftest OK\n");
  return 0;
}
