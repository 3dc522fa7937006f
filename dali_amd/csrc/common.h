// Shared helpers for the gfx950 kernel library (internal; not part of the C ABI).
#ifndef DALI_AMD_CSRC_COMMON_H_
#define DALI_AMD_CSRC_COMMON_H_

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "dali_amd_kernels.h"

namespace daliamd {

void SetLastError(const char *fmt, ...);

#define DALIAMD_HIP_CHECK(expr)                                                         \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ::daliamd::SetLastError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                              __FILE__, __LINE__);                                      \
      return DALIAMD_ERROR_HIP;                                                         \
    }                                                                                   \
  } while (0)

#define DALIAMD_REQUIRE(cond, code, ...)      \
  do {                                        \
    if (!(cond)) {                            \
      ::daliamd::SetLastError(__VA_ARGS__);   \
      return code;                            \
    }                                         \
  } while (0)

// Workgroup -> descriptor lookup: descs sorted by wg_start; returns the last index with
// wg_start <= wg.  Scalar (wave-uniform) binary search.
template <typename Desc>
__device__ __forceinline__ int FindDesc(const Desc *descs, int n, int wg) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// XCD-aware remap (block b is dispatched to XCD b % 8): consecutive logical workgroups, which
// belong to the same sample, land on the same XCD and share its L2.  Returns -1 for padding.
__device__ __forceinline__ int XcdRemap(int block, int logical_total) {
  int per_xcd = (logical_total + 7) >> 3;
  int l = (block & 7) * per_xcd + (block >> 3);
  return l < logical_total ? l : -1;
}
inline int XcdGrid(int logical_total) { return ((logical_total + 7) / 8) * 8; }

// Kernel timing for benchmarks (daliamdKernelTimingEnable): while enabled, a KernelTimer around a launch records a pair
// of timing events on the launch's stream; daliamdKernelTimingReport reads them back per kernel name.  Off: a relaxed
// load and nothing else.
class KernelTimer {
 public:
  KernelTimer(const char *name, hipStream_t stream);
  ~KernelTimer();
  KernelTimer(const KernelTimer &) = delete;
  KernelTimer &operator=(const KernelTimer &) = delete;

 private:
  const char *name_;
  hipStream_t stream_;
  hipEvent_t start_ = nullptr;
};

}  // namespace daliamd
#endif  // DALI_AMD_CSRC_COMMON_H_
