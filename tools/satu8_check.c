// Exhaustive proof for SatU8 (dali_amd/csrc/augment.hip): ConvertSat<uint8_t>(float) - round half away from zero, then
// clamp (the reference: include/dali/core/convert.h:306-321) - equals trunc(clamp(v + pred(0.5f), 0, 255)) with ONE fp32
// addition in round-to-nearest-even, for every float in [-1000, 1000] (outside both sides are 0 / 255).  Three
// instructions on the GPU (v_add_f32, v_med3_f32, v_cvt_u32_f32) instead of eight.
//   gcc -O2 -ffp-contract=off -o satu8_check tools/satu8_check.c -lm && ./satu8_check      (7 s)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t ref(float v) { if (!(v > 0.0f)) return 0; float r = floorf(v); r += (v - r >= 0.5f) ? 1.0f : 0.0f; return (uint32_t)fminf(r, 255.0f); }
static inline uint32_t fast(float v) { volatile float t = v + 0.49999997f; float u = t; u = u < 0.0f ? 0.0f : u; u = u > 255.0f ? 255.0f : u; return (uint32_t)u; }
int main() {
  uint64_t bad = 0, n = 0;
  float hi = 1000.0f; uint32_t hib; memcpy(&hib, &hi, 4);
  for (uint32_t b = 0; b <= hib; b++) { float v; memcpy(&v, &b, 4); n++; if (ref(v) != fast(v)) { if (bad < 5) printf("bad %.9g ref %u fast %u\n", v, ref(v), fast(v)); bad++; } }
  for (uint32_t b = 0x80000000u; b <= 0x80000000u + hib; b++) { float v; memcpy(&v, &b, 4); n++; if (ref(v) != fast(v)) { if (bad < 5) printf("bad %.9g\n", v); bad++; } }
  printf("checked %llu values, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
  return 0;
}
