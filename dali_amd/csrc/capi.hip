// Library plumbing of the C ABI: last-error, device / stream / event / memory helpers.
// These give the C++ host executor what the reference takes from CUDAStreamPool,
// CUDAEventPool and mm:: resources (include/dali/core/cuda_stream_pool.h,
// cuda_event_pool.h, mm/) without the host ever including HIP headers.
#include <dlfcn.h>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"

namespace daliamd {
static thread_local char g_last_error[1024] = "";

void SetLastError(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

// ---- kernel timing (benchmarks) ----
namespace {
struct TimedLaunch { const char *name; hipEvent_t start, stop; };
std::atomic<int> g_timing_on{0};
std::mutex g_timing_mu;
std::vector<TimedLaunch> g_timed;
std::vector<hipEvent_t> g_event_pool;   // events handed back by the report: a timed launch costs two records, no creation
hipEvent_t TakeEvent() {
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (!g_event_pool.empty()) {
      hipEvent_t e = g_event_pool.back();
      g_event_pool.pop_back();
      return e;
    }
  }
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
}  // namespace

KernelTimer::KernelTimer(const char *name, hipStream_t stream) : name_(name), stream_(stream) {
  if (!g_timing_on.load(std::memory_order_relaxed)) return;
  start_ = TakeEvent();
  if (start_) (void)hipEventRecord(start_, stream_);
}
KernelTimer::~KernelTimer() {
  if (!start_) return;
  hipEvent_t stop = TakeEvent();
  if (stop) (void)hipEventRecord(stop, stream_);
  std::lock_guard<std::mutex> lk(g_timing_mu);
  if (stop && g_timed.size() < (1u << 20)) {
    g_timed.push_back({name_, start_, stop});
  } else {
    g_event_pool.push_back(start_);
    if (stop) g_event_pool.push_back(stop);
  }
}
// events for `launches` timed launches, created ahead of a measurement so that none is created inside it
void ReserveTimingEvents(int launches) {
  std::vector<hipEvent_t> fresh;
  for (int i = 0; i < 2 * launches; i++) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) break;
    fresh.push_back(e);
  }
  std::lock_guard<std::mutex> lk(g_timing_mu);
  g_event_pool.insert(g_event_pool.end(), fresh.begin(), fresh.end());
}
}  // namespace daliamd

// Profiler ranges (the reference's DomainTimeRange / nvtx ranges, include/dali/core/nvtx.h:53-82) through roctx,
// resolved at first use so that the library does not depend on the tracer being installed.
namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
Roctx &GetRoctx() {
  static Roctx r;
  return r;
}
}  // namespace
extern "C" {

const char *daliamdGetLastErrorMessage(void) { return daliamd::g_last_error; }
void daliamdClearLastError(void) { daliamd::g_last_error[0] = 0; }
int daliamdVersion(void) { return 100; }

daliamdResult_t daliamdDeviceCount(int *count) {
  DALIAMD_REQUIRE(count, DALIAMD_ERROR_INVALID_ARGUMENT, "count is NULL");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); }
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdSetDevice(int device_id) {
  DALIAMD_HIP_CHECK(hipSetDevice(device_id));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdDeviceInfo(int device_id, char *arch_name, int arch_name_len, int *num_cus,
                                  size_t *total_mem) {
  hipDeviceProp_t p;
  DALIAMD_HIP_CHECK(hipGetDeviceProperties(&p, device_id));
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, p.gcnArchName, arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  if (num_cus) *num_cus = p.multiProcessorCount;
  if (total_mem) *total_mem = p.totalGlobalMem;
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdDevicePciBusId(int device_id, char *bus_id, int len) {
  DALIAMD_REQUIRE(bus_id && len >= 16, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdDevicePciBusId: buffer of >= 16 bytes needed");
  DALIAMD_HIP_CHECK(hipDeviceGetPCIBusId(bus_id, len, device_id));
  return DALIAMD_SUCCESS;
}

void daliamdKernelTimingEnable(int on) {
  // on > 1: also set aside the events of `on` timed launches now (none is created while the measurement runs)
  if (on > 1) daliamd::ReserveTimingEvents(on);
  daliamd::g_timing_on.store(on ? 1 : 0);
}
// "name\tlaunches\tavg_ms\n" per kernel for the launches recorded since the last report (which it consumes); waits for
// them to finish.  Returns the length needed (excluding the terminator), writes at most len - 1 characters.
int daliamdKernelTimingReport(char *buf, int len) {
  static std::map<std::string, std::pair<int, double>> acc;   // launches read back so far, not yet handed out
  static std::mutex acc_mu;
  std::vector<daliamd::TimedLaunch> take;
  {
    std::lock_guard<std::mutex> lk(daliamd::g_timing_mu);
    take.swap(daliamd::g_timed);
  }
  std::lock_guard<std::mutex> lk(acc_mu);
  for (auto &t : take) {
    float ms = 0;
    if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      auto &a = acc[t.name];
      a.first++;
      a.second += ms;
    }
  }
  {
    std::lock_guard<std::mutex> lk2(daliamd::g_timing_mu);
    for (auto &t : take) {
      daliamd::g_event_pool.push_back(t.start);
      daliamd::g_event_pool.push_back(t.stop);
    }
  }
  std::string out;
  for (auto &kv : acc)
    out += kv.first + "\t" + std::to_string(kv.second.first) + "\t" + std::to_string(kv.second.second / kv.second.first) + "\n";
  if (buf && len > 0) {  // a call with a buffer hands the statistics out and starts over
    int n = (int)out.size() < len - 1 ? (int)out.size() : len - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
    acc.clear();
  }
  return (int)out.size();
}

void daliamdRangePush(const char *name) {
  Roctx &r = GetRoctx();
  if (r.push) r.push(name ? name : "");
}
void daliamdRangePop(void) {
  Roctx &r = GetRoctx();
  if (r.pop) r.pop();
}

daliamdResult_t daliamdStreamCreate(daliamdStream_t *stream, int non_blocking) {
  DALIAMD_REQUIRE(stream, DALIAMD_ERROR_INVALID_ARGUMENT, "stream is NULL");
  hipStream_t s;
  DALIAMD_HIP_CHECK(hipStreamCreateWithFlags(&s, non_blocking ? hipStreamNonBlocking : hipStreamDefault));
  *stream = s;
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdStreamCreateWithPriority(daliamdStream_t *stream, int non_blocking, int priority) {
  DALIAMD_REQUIRE(stream, DALIAMD_ERROR_INVALID_ARGUMENT, "stream is NULL");
  int least = 0, greatest = 0;   // numerically: greatest priority = the lowest number
  DALIAMD_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  const int p = priority < 0 ? greatest : priority > 0 ? least : (least + greatest) / 2;
  hipStream_t s;
  DALIAMD_HIP_CHECK(hipStreamCreateWithPriority(&s, non_blocking ? hipStreamNonBlocking : hipStreamDefault, p));
  *stream = s;
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdStreamDestroy(daliamdStream_t stream) {
  DALIAMD_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdStreamSynchronize(daliamdStream_t stream) {
  DALIAMD_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdStreamWaitEvent(daliamdStream_t stream, daliamdEvent_t event) {
  DALIAMD_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdEventCreate(daliamdEvent_t *event, int enable_timing) {
  DALIAMD_REQUIRE(event, DALIAMD_ERROR_INVALID_ARGUMENT, "event is NULL");
  hipEvent_t e;
  // enable_timing: bit 0 = timing, bit 1 = blocking synchronisation (the waiting thread sleeps instead of polling)
  unsigned flags = (enable_timing & 1) ? hipEventDefault : hipEventDisableTiming;
  if (enable_timing & 2) flags |= hipEventBlockingSync;
  DALIAMD_HIP_CHECK(hipEventCreateWithFlags(&e, flags));
  *event = e;
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdEventDestroy(daliamdEvent_t event) {
  DALIAMD_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdEventRecord(daliamdEvent_t event, daliamdStream_t stream) {
  DALIAMD_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdEventSynchronize(daliamdEvent_t event) {
  DALIAMD_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdEventQuery(daliamdEvent_t event, int *done) {
  DALIAMD_REQUIRE(done, DALIAMD_ERROR_INVALID_ARGUMENT, "done is NULL");
  hipError_t e = hipEventQuery((hipEvent_t)event);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();  // "not ready" is an answer, not an error
    *done = 0;
    return DALIAMD_SUCCESS;
  }
  DALIAMD_HIP_CHECK(e);
  *done = 1;
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdEventElapsedMs(daliamdEvent_t start, daliamdEvent_t stop, float *ms) {
  DALIAMD_REQUIRE(ms, DALIAMD_ERROR_INVALID_ARGUMENT, "ms is NULL");
  DALIAMD_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdMalloc(void **ptr, size_t bytes) {
  DALIAMD_REQUIRE(ptr, DALIAMD_ERROR_INVALID_ARGUMENT, "ptr is NULL");
  DALIAMD_HIP_CHECK(hipMalloc(ptr, bytes));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdFree(void *ptr) {
  DALIAMD_HIP_CHECK(hipFree(ptr));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdHostRegister(void *ptr, size_t bytes, int *same_address) {
  DALIAMD_REQUIRE(ptr && bytes > 0 && same_address, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdHostRegister: empty range or NULL result");
  *same_address = 0;
  if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();   // "cannot be page-locked" is an answer: the caller keeps copying
    return DALIAMD_ERROR_HIP;
  }
  void *dev = nullptr;
  if (hipHostGetDevicePointer(&dev, ptr, 0) == hipSuccess) *same_address = dev == ptr ? 1 : 0;
  else (void)hipGetLastError();
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdHostUnregister(void *ptr) {
  DALIAMD_HIP_CHECK(hipHostUnregister(ptr));
  return DALIAMD_SUCCESS;
}

namespace daliamd {
// Workgroup (x = chunk, y = record): kGatherChunk bytes of one record, 16 bytes per lane and load; the loads of a lane
// are all issued before its stores (a read out of host memory takes microseconds: the bus is kept busy by the bytes
// in flight, 4 KB per wave here).  Source and destination share no alignment: the 16-byte units are laid on the
// SOURCE's grid (every load is one aligned bus read inside the record - never a byte in front of it or behind it, the
// neighbouring page of a file mapping may not exist), the stores go out as they fall.
#ifndef DALIAMD_GATHER_UNITS
#define DALIAMD_GATHER_UNITS 4
#endif
#ifndef DALIAMD_GATHER_THREADS
#define DALIAMD_GATHER_THREADS 256
#endif
constexpr int kGatherThreads = DALIAMD_GATHER_THREADS, kGatherUnits = DALIAMD_GATHER_UNITS, kGatherChunk = kGatherThreads * kGatherUnits * 16;
// A FEW workgroups walk all the chunks of all the records (measured inside the pipeline, round 5: one workgroup per chunk -
// 2 300 of them, every CU's vector-memory queue full of reads that take microseconds - doubled the duration of every other
// kernel on the device; the bus needs about 200 KB in flight, not 37 MB).
__device__ __forceinline__ void GatherChunk(const daliamdGatherDesc &d, uint64_t first) {
  const uint8_t *src = static_cast<const uint8_t *>(d.src);
  uint8_t *dst = static_cast<uint8_t *>(d.dst);
  // unit k covers the source bytes [base + 16 k, base + 16 k + 16), base = the source rounded down to 16
  const uint64_t lead = (uint64_t)(reinterpret_cast<uintptr_t>(src) & 15);
  const uint64_t total = lead + d.bytes;                     // bytes from `base` to the record's end
  if (first >= total) return;
  const uint64_t u0 = first / 16, u1 = (first + kGatherChunk < total ? first + kGatherChunk : total + 15) / 16;
  const uint8_t *base = src - lead;
  uint4 v[kGatherUnits];
  bool whole[kGatherUnits];
#pragma unroll
  for (int q = 0; q < kGatherUnits; q++) {
    const uint64_t u = u0 + (uint64_t)q * kGatherThreads + threadIdx.x;
    whole[q] = u < u1 && u * 16 >= lead && u * 16 + 16 <= total;
    if (whole[q]) v[q] = *reinterpret_cast<const uint4 *>(base + u * 16);
  }
#pragma unroll
  for (int q = 0; q < kGatherUnits; q++) {
    const uint64_t u = u0 + (uint64_t)q * kGatherThreads + threadIdx.x;
    if (u >= u1) continue;
    if (whole[q]) {
      uint8_t *o = dst + (u * 16 - lead);
      if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
        *reinterpret_cast<uint4 *>(o) = v[q];
      } else {
        const uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
        for (int b = 0; b < 16; b++) o[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
      }
    } else {   // the record's first or last unit: the bytes that belong to it, one by one
      const uint64_t lo = u * 16 < lead ? lead : u * 16, hi = u * 16 + 16 < total ? u * 16 + 16 : total;
      for (uint64_t b = lo; b < hi; b++) dst[b - lead] = base[b];
    }
  }
}
__global__ __launch_bounds__(kGatherThreads) void GatherCopyKernel(const daliamdGatherDesc *__restrict__ descs, int n, unsigned chunks) {
  // (raised wave priority: these waves have a handful of instructions between loads that take microseconds.  Measured
  // inside the pipeline it changes nothing either way - 0.62 against 0.62 ms per launch, gpurun_out/r05_v - and stays.)
  __builtin_amdgcn_s_setprio(3);
  for (unsigned long long work = blockIdx.x; work < (unsigned long long)n * chunks; work += gridDim.x)
    GatherChunk(descs[work / chunks], (uint64_t)(work % chunks) * kGatherChunk);
}
}  // namespace daliamd

daliamdResult_t daliamdGatherCopy(const daliamdGatherDesc *descs, int n, size_t max_bytes, daliamdStream_t s) {
  DALIAMD_REQUIRE(n >= 0 && (n == 0 || descs), DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdGatherCopy: NULL table");
  if (n == 0 || max_bytes == 0) return DALIAMD_SUCCESS;
  const size_t chunks = (max_bytes + 15 + daliamd::kGatherChunk - 1) / daliamd::kGatherChunk;
  DALIAMD_REQUIRE(chunks <= 0x7fffffffu, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdGatherCopy: record too long");
  {
    daliamd::KernelTimer timer("GatherCopyKernel", (hipStream_t)s);
    static const int wgs = [] { const char *e = getenv("DALI_AMD_GATHER_WGS"); return e && atoi(e) > 0 ? atoi(e) : 64; }();
    const unsigned long long work = (unsigned long long)n * chunks;
    hipLaunchKernelGGL(daliamd::GatherCopyKernel, dim3((unsigned)(work < (unsigned long long)wgs ? work : wgs)), dim3(daliamd::kGatherThreads), 0,
                       (hipStream_t)s, descs, n, (unsigned)chunks);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdHostAlloc(void **ptr, size_t bytes) {
  DALIAMD_REQUIRE(ptr, DALIAMD_ERROR_INVALID_ARGUMENT, "ptr is NULL");
  DALIAMD_HIP_CHECK(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdHostFree(void *ptr) {
  DALIAMD_HIP_CHECK(hipHostFree(ptr));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdMemcpyH2DAsync(void *dst, const void *src, size_t bytes, daliamdStream_t s) {
  DALIAMD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdMemcpyD2HAsync(void *dst, const void *src, size_t bytes, daliamdStream_t s) {
  DALIAMD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdMemcpyD2DAsync(void *dst, const void *src, size_t bytes, daliamdStream_t s) {
  DALIAMD_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdMemsetAsync(void *dst, int value, size_t bytes, daliamdStream_t s) {
  DALIAMD_HIP_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)s));
  return DALIAMD_SUCCESS;
}
daliamdResult_t daliamdMemcpy2DD2DAsync(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes,
                                        size_t height, daliamdStream_t s) {
  DALIAMD_HIP_CHECK(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, hipMemcpyDeviceToDevice,
                                     (hipStream_t)s));
  return DALIAMD_SUCCESS;
}

}  // extern "C"
