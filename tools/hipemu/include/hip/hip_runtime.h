// hipemu: a CPU execution model of the HIP constructs dali_amd/csrc uses - TEST INFRASTRUCTURE ONLY.
//
// The product (dali_amd/lib/libdali_amd_kernels.so) is built by hipcc for gfx950 and never sees this header.
// tools/hipemu/Makefile compiles the SAME kernel sources as plain C++ (clang++ -x c++) with this directory first on
// the include path; the result (tools/hipemu/_build/lib/) is loaded only by the `-m "not gpu"` tests that ask for it
// (tests/hipemu_env.py) - a way to run the device code's arithmetic, indexing and LDS choreography under
// AddressSanitizer and against the oracle in a container without a GPU.  It is not a fallback: nothing under dali_amd/
// knows it exists, and dali_amd._capi fails loudly without the gfx950 library exactly as before.
//
// Execution model: a launch runs synchronously, workgroup by workgroup (several OS threads take workgroups off a
// counter); the threads of a workgroup are fibers on one OS thread, so `__shared__` is `static thread_local`; a fiber
// runs until it reaches a workgroup barrier or a wave collective (shuffle / ballot / readfirstlane / wave barrier /
// MFMA); a collective resolves when every live lane of the 64-wide wave is blocked, over the lanes that wait at the
// same call site (the lowest call-site address first when lanes diverged).  Streams and events are ordering-free:
// every call completes before it returns.
#ifndef HIPEMU_HIP_RUNTIME_H_
#define HIPEMU_HIP_RUNTIME_H_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

#define __HIPEMU__ 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline, convergent))
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...) unused
// pointers are host pointers: __attribute__((address_space(1))) becomes the empty attribute
#define address_space(n)
#define amdgpu_flat_work_group_size(...) unused
#define __shared__ static thread_local
#define __constant__ static const
#define HIP_DYNAMIC_SHARED(type, name) type *name = reinterpret_cast<type *>(::hipemu::DynamicShared());

// ------------------------------------------------------------------------------------------------ vector types
struct dim3 {
  uint32_t x, y, z;
  constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define HIPEMU_VEC2(name, T, al) \
  struct alignas(al) name { T x, y; }; \
  static inline name make_##name(T x, T y) { return name{x, y}; }
#define HIPEMU_VEC3(name, T) \
  struct name { T x, y, z; }; \
  static inline name make_##name(T x, T y, T z) { return name{x, y, z}; }
#define HIPEMU_VEC4(name, T, al) \
  struct alignas(al) name { T x, y, z, w; }; \
  static inline name make_##name(T x, T y, T z, T w) { return name{x, y, z, w}; }
HIPEMU_VEC2(char2, signed char, 2) HIPEMU_VEC2(uchar2, unsigned char, 2) HIPEMU_VEC2(short2, short, 4)
HIPEMU_VEC2(ushort2, unsigned short, 4) HIPEMU_VEC2(int2, int, 8) HIPEMU_VEC2(uint2, unsigned, 8)
HIPEMU_VEC2(float2, float, 8) HIPEMU_VEC2(double2, double, 16) HIPEMU_VEC2(longlong2, long long, 16)
HIPEMU_VEC2(ulonglong2, unsigned long long, 16)
HIPEMU_VEC3(uchar3, unsigned char) HIPEMU_VEC3(int3, int) HIPEMU_VEC3(uint3, unsigned) HIPEMU_VEC3(float3, float)
HIPEMU_VEC4(char4, signed char, 4) HIPEMU_VEC4(uchar4, unsigned char, 4) HIPEMU_VEC4(short4, short, 8)
HIPEMU_VEC4(ushort4, unsigned short, 8) HIPEMU_VEC4(int4, int, 16) HIPEMU_VEC4(uint4, unsigned, 16)
HIPEMU_VEC4(float4, float, 16)
#undef HIPEMU_VEC2
#undef HIPEMU_VEC3
#undef HIPEMU_VEC4

namespace hipemu {
// 12 bytes, stored and loaded as 12 bytes: the x86 front end widens a 3-element ext_vector_type to 16 (the amdgcn one
// keeps it: global_load / store_dwordx3).  The build's sed puts this in place of the kernels' typedef.
template <typename T> struct Vec3 { T x, y, z; };
}  // namespace hipemu

// ------------------------------------------------------------------------------------------------ runtime types
typedef enum hipError_t {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNotReady = 600,
  hipErrorUnknown = 999
} hipError_t;
typedef struct hipemuStream *hipStream_t;
typedef struct hipemuEvent *hipEvent_t;
typedef enum hipMemcpyKind {
  hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3,
  hipMemcpyDefault = 4
} hipMemcpyKind;
typedef enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
  char name[256];
  size_t totalGlobalMem;
  size_t sharedMemPerBlock;
  int multiProcessorCount;
  int warpSize;
  int clockRate;
  int memoryClockRate;
  int memoryBusWidth;
  int major, minor;
  int pciBusID, pciDeviceID, pciDomainID;
  size_t l2CacheSize;
  int maxThreadsPerBlock;
  char gcnArchName[256];
};

namespace hipemu {

// ------------------------------------------------------------------------------------------------ execution model
enum Op : int { kShflIdx, kShflUp, kShflDown, kShflXor, kBallot, kFirstLane, kWaveBarrier, kMfma16x16x4F32 };

struct Lane;  // a fiber
struct ThreadState {
  dim3 thread_idx, block_idx, block_dim, grid_dim;
  char *dyn_shared;
  Lane *lane;
};
// the running fiber's coordinates (one copy per OS thread, rewritten at every fiber switch)
extern thread_local ThreadState tls;

void Launch(dim3 grid, dim3 block, size_t dyn_shared_bytes, const std::function<void()> &body);
__attribute__((convergent)) void BlockBarrier();
__attribute__((convergent)) int BlockBarrierOr(int pred);  // __syncthreads_or
// blocks until the wave's live lanes are blocked; `in`/`out`: up to 32 / 16 bytes; returns the mask of the lanes that
// took part
__attribute__((convergent)) uint64_t Collective(Op op, const void *in, int in_bytes, int arg, int width, void *out, int out_bytes, int site, const void *address);
inline char *DynamicShared() { return tls.dyn_shared; }
void RaceAccess(const void *addr, size_t size, bool is_write, const void *pc);   // racecheck build
long RaceCount();
long UninitCount();
void RaceAtomic(const void *addr, size_t size, bool reads, bool writes, const void *pc);
extern const void *launch_kernel;   // ldsprof: the kernel function of the launch being issued
inline int LaneId() {
  const ThreadState &t = tls;
  return (int)((t.thread_idx.x + t.block_dim.x * (t.thread_idx.y + t.block_dim.y * t.thread_idx.z)) & 63);
}

// runtime calls (hipemu_rt.cpp)
hipError_t Malloc(void **p, size_t n);
hipError_t Free(void *p);
hipError_t HostMalloc(void **p, size_t n);
hipError_t HostFree(void *p);
hipError_t EventCreate(hipEvent_t *e);
hipError_t EventDestroy(hipEvent_t e);
hipError_t EventRecord(hipEvent_t e);
hipError_t EventElapsed(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t StreamCreate(hipStream_t *s);
hipError_t StreamDestroy(hipStream_t s);

}  // namespace hipemu

#define threadIdx (::hipemu::tls.thread_idx)
#define blockIdx (::hipemu::tls.block_idx)
#define blockDim (::hipemu::tls.block_dim)
#define gridDim (::hipemu::tls.grid_dim)
static constexpr int warpSize = 64;

// ------------------------------------------------------------------------------------------------ device functions
#define HIPEMU_SITE() __builtin_extract_return_addr(__builtin_return_address(0))

static inline __attribute__((convergent)) void __syncthreads() { ::hipemu::BlockBarrier(); }
static inline __attribute__((convergent)) int __syncthreads_or(int pred) { return ::hipemu::BlockBarrierOr(pred); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// Every textual call site of a collective is its own function (template parameter = __COUNTER__): two calls in the arms
// of a branch must stay two calls (an optimiser that merges them into one call behind the join - clang does, `convergent`
// or not - would hand the lanes of both arms to one collective).  The site's identity is that number; its address
// orders the groups of a diverged wave.
namespace hipemu {
template <int Site, typename T>
__attribute__((noinline, convergent)) T Shuffle(Op op, T v, int arg, int width) {
  static_assert(sizeof(T) <= 8 && std::is_trivially_copyable<T>::value, "shuffle of a wide type");
  T out;
  Collective(op, &v, sizeof(T), arg, width, &out, sizeof(T), Site, HIPEMU_SITE());
  return out;
}
template <int Site>
__attribute__((noinline, convergent)) uint64_t Ballot(int pred) {
  uint64_t out;
  Collective(kBallot, &pred, sizeof(pred), 0, 64, &out, sizeof(out), Site, HIPEMU_SITE());
  return out;
}
template <int Site>
__attribute__((noinline, convergent)) uint32_t FirstLane(uint32_t v) {
  uint32_t out;
  Collective(kFirstLane, &v, sizeof(v), 0, 64, &out, sizeof(out), Site, HIPEMU_SITE());
  return out;
}
template <int Site>
__attribute__((noinline, convergent)) void WaveBarrier() {
  Collective(kWaveBarrier, nullptr, 0, 0, 64, nullptr, 0, Site, HIPEMU_SITE());
}
template <int Site, typename T> T ShflIdx(T v, int src, int width = 64) { return Shuffle<Site>(kShflIdx, v, src, width); }
template <int Site, typename T> T ShflUp(T v, unsigned d, int width = 64) { return Shuffle<Site>(kShflUp, v, (int)d, width); }
template <int Site, typename T> T ShflDown(T v, unsigned d, int width = 64) { return Shuffle<Site>(kShflDown, v, (int)d, width); }
template <int Site, typename T> T ShflXor(T v, int m, int width = 64) { return Shuffle<Site>(kShflXor, v, m, width); }
}  // namespace hipemu

#define __shfl(...) ::hipemu::ShflIdx<__COUNTER__>(__VA_ARGS__)
#define __shfl_up(...) ::hipemu::ShflUp<__COUNTER__>(__VA_ARGS__)
#define __shfl_down(...) ::hipemu::ShflDown<__COUNTER__>(__VA_ARGS__)
#define __shfl_xor(...) ::hipemu::ShflXor<__COUNTER__>(__VA_ARGS__)
#define __ballot(...) ::hipemu::Ballot<__COUNTER__>(__VA_ARGS__)
#define __any(...) (::hipemu::Ballot<__COUNTER__>(__VA_ARGS__) != 0)
#define __builtin_amdgcn_readfirstlane(...) ::hipemu::FirstLane<__COUNTER__>((uint32_t)(__VA_ARGS__))
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_wave_barrier() ::hipemu::WaveBarrier<__COUNTER__>()
static inline unsigned __lane_id() { return (unsigned)::hipemu::LaneId(); }

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __mul24(int a, int b) { return (int)((int64_t)((a << 8) >> 8) * ((b << 8) >> 8)); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffff) * (b & 0xffffff); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline void sincospif(float x, float *s, float *c) {
  // exact argument reduction first (sin / cos of pi x, x finite), then the host's libm on [-pi/4, pi/4]-sized arguments
  double r = (double)x - 2.0 * std::floor((double)x * 0.5);
  *s = (float)std::sin(M_PI * r);
  *c = (float)std::cos(M_PI * r);
}
using std::max;
using std::min;
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __saturatef(float x) { return x != x ? 0.f : std::min(std::max(x, 0.f), 1.f); }

// amdgcn builtins of the kernels (clang knows them for the amdgcn target only)
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) {
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (shift & 3)));
}
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
  return std::max(std::min(a, b), std::min(std::max(a, b), c));
}
static inline __attribute__((convergent)) void __builtin_amdgcn_s_barrier() { ::hipemu::BlockBarrier(); }
#define __builtin_amdgcn_fence(...) __atomic_thread_fence(__ATOMIC_SEQ_CST)
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_s_waitcnt(int) {}   // (the model has no counters to wait for: memory operations complete in order)

typedef float hipemu_float4_vec __attribute__((ext_vector_type(4)));
namespace hipemu {
template <int Site>
__attribute__((noinline, convergent)) hipemu_float4_vec Mfma16x16x4F32(float a, float b, hipemu_float4_vec c, int, int, int) {
  float in[6] = {a, b, c[0], c[1], c[2], c[3]};
  float out[4];
  Collective(kMfma16x16x4F32, in, sizeof(in), 0, 64, out, sizeof(out), Site, HIPEMU_SITE());
  return hipemu_float4_vec{out[0], out[1], out[2], out[3]};
}
}  // namespace hipemu
#define __builtin_amdgcn_mfma_f32_16x16x4f32(...) ::hipemu::Mfma16x16x4F32<__COUNTER__>(__VA_ARGS__)

// atomics (workgroups run on several OS threads).  The build caps the alignment assumed for pointers at 4 bytes (the
// device's vector loads need no more), so the 8-byte ones go through a type that states its alignment.
typedef uint32_t hipemu_a32 __attribute__((aligned(4)));
typedef uint64_t hipemu_a64 __attribute__((aligned(8)));
// (the typedefs are used directly: an alignment attribute does not survive a trip through a template parameter)
static inline uint32_t hipemu_load(void *p, uint32_t) { return __atomic_load_n((hipemu_a32 *)p, __ATOMIC_SEQ_CST); }
static inline uint64_t hipemu_load(void *p, uint64_t) { return __atomic_load_n((hipemu_a64 *)p, __ATOMIC_SEQ_CST); }
static inline bool hipemu_cas(void *p, uint32_t *expected, uint32_t v) {
  return __atomic_compare_exchange_n((hipemu_a32 *)p, expected, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
}
static inline bool hipemu_cas(void *p, uint64_t *expected, uint64_t v) {
  return __atomic_compare_exchange_n((hipemu_a64 *)p, expected, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
}
template <size_t N> struct hipemu_rep;
template <> struct hipemu_rep<4> { typedef uint32_t plain; };
template <> struct hipemu_rep<8> { typedef uint64_t plain; };
// *p = f(*p, v) atomically; returns the old value
template <typename T, typename F> static inline T hipemu_atomic_rmw(T *p, T v, F f) {
  typedef typename hipemu_rep<sizeof(T)>::plain Plain;
  Plain old = hipemu_load(p, Plain()), neu;
  T cur;
  do {
    memcpy(&cur, &old, sizeof(T));
    const T res = f(cur, v);
    memcpy(&neu, &res, sizeof(T));
  } while (!hipemu_cas(p, &old, neu));
  return cur;
}
#define HIPEMU_ATOMIC(name, T, expr) \
  static inline T name(T *p, T v) { return hipemu_atomic_rmw(p, v, [](T a, T b) { return (T)(expr); }); }
HIPEMU_ATOMIC(atomicAdd, int, a + b) HIPEMU_ATOMIC(atomicAdd, unsigned, a + b)
HIPEMU_ATOMIC(atomicAdd, unsigned long long, a + b) HIPEMU_ATOMIC(atomicAdd, float, a + b)
HIPEMU_ATOMIC(atomicAdd, double, a + b)
HIPEMU_ATOMIC(atomicSub, int, a - b) HIPEMU_ATOMIC(atomicSub, unsigned, a - b)
HIPEMU_ATOMIC(atomicMax, int, a > b ? a : b) HIPEMU_ATOMIC(atomicMax, unsigned, a > b ? a : b)
HIPEMU_ATOMIC(atomicMax, unsigned long long, a > b ? a : b)
HIPEMU_ATOMIC(atomicMin, int, a < b ? a : b) HIPEMU_ATOMIC(atomicMin, unsigned, a < b ? a : b)
HIPEMU_ATOMIC(atomicMin, unsigned long long, a < b ? a : b)
HIPEMU_ATOMIC(atomicOr, int, a | b) HIPEMU_ATOMIC(atomicOr, unsigned, a | b)
HIPEMU_ATOMIC(atomicOr, unsigned long long, a | b)
HIPEMU_ATOMIC(atomicAnd, int, a & b) HIPEMU_ATOMIC(atomicAnd, unsigned, a & b)
HIPEMU_ATOMIC(atomicExch, int, b) HIPEMU_ATOMIC(atomicExch, unsigned, b) HIPEMU_ATOMIC(atomicExch, float, b)
#undef HIPEMU_ATOMIC
template <typename T> static inline T hipemu_atomic_cas(T *p, T cmp, T v) {
  typedef typename hipemu_rep<sizeof(T)>::plain Plain;
  Plain c, n;
  memcpy(&c, &cmp, sizeof(T));
  memcpy(&n, &v, sizeof(T));
  hipemu_cas(p, &c, n);
  T old;
  memcpy(&old, &c, sizeof(T));
  return old;
}
static inline int atomicCAS(int *p, int cmp, int v) { return hipemu_atomic_cas(p, cmp, v); }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v) { return hipemu_atomic_cas(p, cmp, v); }
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long v) {
  return hipemu_atomic_cas(p, cmp, v);
}

// ------------------------------------------------------------------------------------------------ host API
static inline const char *hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "hipSuccess";
    case hipErrorInvalidValue: return "hipErrorInvalidValue";
    case hipErrorOutOfMemory: return "hipErrorOutOfMemory";
    case hipErrorNotReady: return "hipErrorNotReady";
    default: return "hipErrorUnknown";
  }
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "hipemu (CPU model of gfx950)");
  strcpy(p->gcnArchName, "gfx950");
  p->totalGlobalMem = (size_t)288 << 30;
  p->sharedMemPerBlock = 160 << 10;
  p->multiProcessorCount = 256;
  p->warpSize = 64;
  p->clockRate = 2400000;
  p->memoryClockRate = 2000000;
  p->memoryBusWidth = 8192;
  p->major = 9; p->minor = 5;
  p->l2CacheSize = 4 << 20;
  p->maxThreadsPerBlock = 1024;
  return hipSuccess;
}
static inline hipError_t hipDeviceGetPCIBusId(char *buf, int len, int) {
  snprintf(buf, (size_t)len, "0000:00:00.0");
  return hipSuccess;
}
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) {
  *least = 0; *greatest = -1;
  return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { return ::hipemu::Malloc(p, n); }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return ::hipemu::Malloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { return ::hipemu::Free(p); }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return ::hipemu::HostMalloc(p, n); }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned = 0) {
  return ::hipemu::HostMalloc((void **)p, n);
}
static inline hipError_t hipHostFree(void *p) { return ::hipemu::HostFree(p); }
// host memory registered for device access: the model's "device" reads host memory anyway
enum { hipHostRegisterDefault = 0 };
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  if (n) memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
  if (n) memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width,
                                          size_t height, hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t y = 0; y < height; y++) memmove((char *)d + y * dpitch, (const char *)s + y * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) {
  if (n) memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) {
  if (n) memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return ::hipemu::StreamCreate(s); }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return ::hipemu::StreamCreate(s); }
static inline hipError_t hipStreamCreate(hipStream_t *s) { return ::hipemu::StreamCreate(s); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { return ::hipemu::StreamDestroy(s); }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { return ::hipemu::EventCreate(e); }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return ::hipemu::EventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { return ::hipemu::EventDestroy(e); }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { return ::hipemu::EventRecord(e); }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  return ::hipemu::EventElapsed(ms, a, b);
}
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

// hipLaunchKernelGGL(kernel, grid, block, dynamic LDS bytes, stream, args...): arguments are converted to the kernel's
// parameter types once (as a launch does), every thread of the grid calls the kernel with copies of them
template <typename... Params, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(Params...), dim3 grid, dim3 block, size_t dyn_shared,
                                      hipStream_t, Args &&...args) {
  std::tuple<std::decay_t<Params>...> params(std::forward<Args>(args)...);
  ::hipemu::launch_kernel = reinterpret_cast<const void *>(kernel);
  ::hipemu::Launch(grid, block, dyn_shared, [&]() { std::apply(kernel, params); });
}

#endif  // HIPEMU_HIP_RUNTIME_H_
