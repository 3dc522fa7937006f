// Self-test of the racecheck build: the detector must stay silent on correctly synchronised kernels and must name the
// three kinds of missing barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void CleanKernel(int *out) {
  __shared__ int buf[256];
  const int t = threadIdx.x;
  buf[t] = t;
  __syncthreads();
  int v = buf[255 - t];
  __syncthreads();
  buf[t] = v * 2;                 // same lane as the read before the barrier? no: other lanes read buf[t] - the barrier orders it
  __builtin_amdgcn_wave_barrier();
  v += buf[t ^ 1];                // a neighbour of the same wave, behind a wave barrier
  out[blockIdx.x * 256 + t] = v;
}

__global__ void MissingBlockBarrier(int *out) {
  __shared__ int buf[256];
  const int t = threadIdx.x;
  buf[t] = t;
  out[blockIdx.x * 256 + t] = buf[255 - t];   // another wave's element, no barrier
}

__global__ void MissingWaveBarrier(int *out) {
  __shared__ int buf[256];
  const int t = threadIdx.x;
  buf[t] = t;
  out[blockIdx.x * 256 + t] = buf[t ^ 1];     // same wave, nothing wave-wide in between
}

__global__ void OverlappingStores(unsigned char *out) {
  const int t = threadIdx.x;
  *reinterpret_cast<unsigned *>(out + 3 * t) = 0x01010101u * (unsigned)t;   // 4-byte stores at a 3-byte stride
}

// initcheck: LDS holds what earlier workgroups left there - block 1 must not get away with reading block 0's values
__global__ void StaleLds(int *out) {
  __shared__ int buf[256];
  __shared__ unsigned counter;
  const int t = threadIdx.x;
  if (blockIdx.x == 0 || t < 128) buf[t] = t;
  if (blockIdx.x == 0 && t == 0) counter = 0;
  __syncthreads();
  if (t == 0) atomicAdd(&counter, 1u);          // block 1 never initialised it
  out[blockIdx.x * 256 + t] = buf[t];           // block 1, lanes 128..255: never written by this workgroup
}

int main() {
  std::vector<int> out(2 * 256);
  std::vector<unsigned char> bytes(3 * 256 + 8);
  hipLaunchKernelGGL(CleanKernel, dim3(2), dim3(256), 0, nullptr, out.data());
  if (hipemu::RaceCount() != 0 || hipemu::UninitCount() != 0) { printf("FAILED: false positive\n"); return 1; }
  hipLaunchKernelGGL(StaleLds, dim3(2), dim3(256), 0, nullptr, out.data());
  if (hipemu::RaceCount() != 0) { printf("FAILED: false positive (race) in StaleLds\n"); return 1; }
  if (hipemu::UninitCount() < 129 || hipemu::UninitCount() > 131) { printf("FAILED: %ld uninitialised LDS reads seen, 129 (+ the atomic's own load) expected\n", hipemu::UninitCount()); return 1; }
  hipLaunchKernelGGL(MissingBlockBarrier, dim3(1), dim3(256), 0, nullptr, out.data());
  const long a = hipemu::RaceCount();
  if (a == 0) { printf("FAILED: missing workgroup barrier not seen\n"); return 1; }
  hipLaunchKernelGGL(MissingWaveBarrier, dim3(1), dim3(256), 0, nullptr, out.data());
  const long b = hipemu::RaceCount();
  if (b == a) { printf("FAILED: missing wave barrier not seen\n"); return 1; }
  hipLaunchKernelGGL(OverlappingStores, dim3(1), dim3(256), 0, nullptr, bytes.data());
  if (hipemu::RaceCount() == b) { printf("FAILED: overlapping stores not seen\n"); return 1; }
  printf("hipemu racetest OK (%ld races reported, as intended)\n", hipemu::RaceCount());
  return 0;
}
