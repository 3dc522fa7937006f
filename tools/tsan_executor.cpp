// ThreadSanitizer driver for the host executor (host-stage worker, thread pool, ring slots, shutdown): builds a
// reader-only pipeline through the flat C API three times and runs 60 iterations each.  Development tool:
//   g++ -O1 -g -std=c++17 -fsanitize=thread -fPIC -Iinclude -Idali_amd/host -pthread tools/tsan_executor.cpp \
//       dali_amd/host/*.cpp -Ldali_amd/lib -ldali_amd_kernels -lz -Wl,-rpath,$PWD/dali_amd/lib -o /tmp/tsan_executor
//   /tmp/tsan_executor DIR_WITH_CLASS_SUBDIRECTORIES      (clean at the time of writing)
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
extern "C" {
void *daliamdOpSpecCreate(const char *);
void daliamdOpSpecDestroy(void *);
void daliamdOpSpecAddArgInt(void *, const char *, int64_t);
void daliamdOpSpecAddArgBool(void *, const char *, int);
void daliamdOpSpecAddArgStr(void *, const char *, const char *);
void daliamdOpSpecAddOutput(void *, const char *, int);
void *daliamdPipelineCreate(int, int, int, int64_t, int, int);
void daliamdPipelineDestroy(void *);
int daliamdPipelineAddOperator(void *, void *, const char *);
int daliamdPipelineBuild(void *, const char *const *, const int *, int);
int daliamdPipelineRun(void *);
int daliamdPipelineOutputs(void *, int *);
int daliamdPipelineOutputSamples(void *, int, void **, int64_t *, int *, int64_t *);
const char *daliamdHostGetLastErrorMessage(void);
}
int main(int argc, char **argv) {
  for (int rep = 0; rep < 3; rep++) {
    void *p = daliamdPipelineCreate(32, 4, -1, 1234, 2, 1);
    void *s = daliamdOpSpecCreate("readers__File");
    daliamdOpSpecAddArgStr(s, "file_root", argv[1]);
    daliamdOpSpecAddArgBool(s, "random_shuffle", 1);
    daliamdOpSpecAddOutput(s, "jpegs", 0);
    daliamdOpSpecAddOutput(s, "labels", 0);
    if (daliamdPipelineAddOperator(p, s, "Reader")) { printf("add failed\n"); return 1; }
    const char *names[2] = {"jpegs", "labels"};
    int gpu[2] = {0, 0};
    if (daliamdPipelineBuild(p, names, gpu, 2)) { printf("build failed\n"); return 1; }
    long bytes = 0;
    for (int it = 0; it < 60; it++) {
      if (daliamdPipelineRun(p)) { printf("run failed\n"); return 1; }
      int n = 0;
      if (daliamdPipelineOutputs(p, &n)) { printf("outputs failed\n"); return 1; }
      void *ptrs[32]; int64_t shapes[32 * 8]; int nd[32]; int64_t pitch[32];
      daliamdPipelineOutputSamples(p, 0, ptrs, shapes, nd, pitch);
      for (int i = 0; i < 32; i++) bytes += ((const unsigned char *)ptrs[i])[0] + shapes[i * 8];
    }
    daliamdPipelineDestroy(p);
    daliamdOpSpecDestroy(s);
    printf("rep %d ok %ld\n", rep, bytes);
  }
  return 0;
}
