// Internal helpers of libdali_amd_host.so (not part of the C ABI).
#ifndef DALI_AMD_HOST_COMMON_H_
#define DALI_AMD_HOST_COMMON_H_
namespace daliamd_host {
// Records a thread-local error message and returns a non-zero status.
int Fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
}  // namespace daliamd_host
#endif
