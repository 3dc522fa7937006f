// The calls that -fsanitize=thread plants in front of every memory access of the kernel sources, answered by hipemu's
// own lane-level race detector (hipemu_rt.cpp) instead of the ThreadSanitizer runtime: lanes are fibers of one OS
// thread, and what orders their accesses are the workgroup's barriers and the wave-wide operations, which that
// runtime knows nothing about.  TEST INFRASTRUCTURE (make RACE=1).
#include <hip/hip_runtime.h>

#define PC __builtin_extract_return_addr(__builtin_return_address(0))
#define HOOK(n)                                                                                              \
  extern "C" __attribute__((visibility("default"))) void __tsan_read##n(void *a) { ::hipemu::RaceAccess(a, n, false, PC); } \
  extern "C" __attribute__((visibility("default"))) void __tsan_write##n(void *a) { ::hipemu::RaceAccess(a, n, true, PC); } \
  extern "C" __attribute__((visibility("default"))) void __tsan_unaligned_read##n(void *a) { ::hipemu::RaceAccess(a, n, false, PC); } \
  extern "C" __attribute__((visibility("default"))) void __tsan_unaligned_write##n(void *a) { ::hipemu::RaceAccess(a, n, true, PC); }
HOOK(1) HOOK(2) HOOK(4) HOOK(8) HOOK(16)

#define API extern "C" __attribute__((visibility("default")))
API void __tsan_init() {}
API void __tsan_func_entry(void *) {}
API void __tsan_func_exit() {}
API void __tsan_vptr_update(void **, void *) {}
API void __tsan_vptr_read(void **) {}
API void *__tsan_memcpy(void *d, const void *s, size_t n) {
  if (n) { ::hipemu::RaceAccess(s, n, false, PC); ::hipemu::RaceAccess(d, n, true, PC); }
  return memcpy(d, s, n);
}
API void *__tsan_memmove(void *d, const void *s, size_t n) {
  if (n) { ::hipemu::RaceAccess(s, n, false, PC); ::hipemu::RaceAccess(d, n, true, PC); }
  return memmove(d, s, n);
}
API void *__tsan_memset(void *d, int v, size_t n) {
  if (n) ::hipemu::RaceAccess(d, n, true, PC);
  return memset(d, v, n);
}
// atomics are not data races; they keep their meaning (the memory-order arguments are the __ATOMIC_* values)
API void __tsan_atomic_thread_fence(int) { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
API void __tsan_atomic_signal_fence(int) {}
#define RMW(p, r, w) ::hipemu::RaceAtomic(const_cast<const void *>(reinterpret_cast<const volatile void *>(p)), sizeof(*(p)), r, w, PC)
#define ATOMIC(bits, T)                                                                                          \
  API T __tsan_atomic##bits##_load(const volatile T *p, int) { RMW(p, true, false); return __atomic_load_n(p, __ATOMIC_SEQ_CST); }     \
  API void __tsan_atomic##bits##_store(volatile T *p, T v, int) { RMW(p, false, true); __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }     \
  API T __tsan_atomic##bits##_exchange(volatile T *p, T v, int) { RMW(p, false, true); return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); } \
  API T __tsan_atomic##bits##_fetch_add(volatile T *p, T v, int) { RMW(p, true, true); return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); } \
  API T __tsan_atomic##bits##_fetch_sub(volatile T *p, T v, int) { RMW(p, true, true); return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); } \
  API T __tsan_atomic##bits##_fetch_and(volatile T *p, T v, int) { RMW(p, true, true); return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); } \
  API T __tsan_atomic##bits##_fetch_or(volatile T *p, T v, int) { RMW(p, true, true); return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }  \
  API T __tsan_atomic##bits##_fetch_xor(volatile T *p, T v, int) { RMW(p, true, true); return __atomic_fetch_xor(p, v, __ATOMIC_SEQ_CST); } \
  API T __tsan_atomic##bits##_compare_exchange_val(volatile T *p, T c, T v, int, int) {                          \
    RMW(p, true, true);                                                                                           \
    __atomic_compare_exchange_n(p, &c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);                             \
    return c;                                                                                                     \
  }                                                                                                               \
  API int __tsan_atomic##bits##_compare_exchange_strong(volatile T *p, T *c, T v, int, int) {                     \
    RMW(p, true, true);                                                                                           \
    return __atomic_compare_exchange_n(p, c, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);                       \
  }                                                                                                               \
  API int __tsan_atomic##bits##_compare_exchange_weak(volatile T *p, T *c, T v, int, int) {                       \
    RMW(p, true, true);                                                                                           \
    return __atomic_compare_exchange_n(p, c, v, true, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);                        \
  }
ATOMIC(8, uint8_t) ATOMIC(16, uint16_t) ATOMIC(32, uint32_t) ATOMIC(64, uint64_t)
