// Integer 8-point inverse DCT butterfly of libjpeg-turbo's accurate ("islow") IDCT, shared by the stand-alone IDCT
// kernel (jpeg_idct.hip) and the entropy decoder's fused block output (jpeg_huffman.hip: ExpandKernel).
#ifndef DALI_AMD_CSRC_JPEG_IDCT_MATH_H_
#define DALI_AMD_CSRC_JPEG_IDCT_MATH_H_
#include <cstdint>
#include <hip/hip_runtime.h>

namespace daliamd {

#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172

__device__ __forceinline__ int32_t Descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// Product with a 13/15-bit constant.  v_mul_lo_u32 runs at a quarter of the rate of v_mul_i32_i24; every operand of
// the butterfly fits in 24 bits for any stream a baseline 8-bit JPEG can hold (dequantised coefficients < 2^20, first
// pass results < 2^21), and the low 32 bits of the product are the same then.
__device__ __forceinline__ int32_t MulC(int32_t v, int32_t c) { return __mul24(v, c); }

// One 8-point pass of the islow butterfly.  in[0..7] -> out[0..7] (not yet descaled).
__device__ __forceinline__ void Butterfly8(const int32_t in[8], int32_t out[8]) {
  int32_t z2 = in[2], z3 = in[6];
  int32_t z1 = MulC(z2 + z3, FIX_0_541196100);
  int32_t tmp2 = z1 + MulC(z3, -FIX_1_847759065);
  int32_t tmp3 = z1 + MulC(z2, FIX_0_765366865);
  int32_t tmp0 = (in[0] + in[4]) * (1 << CONST_BITS);
  int32_t tmp1 = (in[0] - in[4]) * (1 << CONST_BITS);
  int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int32_t z4 = tmp1 + tmp3;
  int32_t z5 = MulC(z3 + z4, FIX_1_175875602);
  tmp0 = MulC(tmp0, FIX_0_298631336); tmp1 = MulC(tmp1, FIX_2_053119869);
  tmp2 = MulC(tmp2, FIX_3_072711026); tmp3 = MulC(tmp3, FIX_1_501321110);
  z1 = MulC(z1, -FIX_0_899976223); z2 = MulC(z2, -FIX_2_562915447);
  z3 = MulC(z3, -FIX_1_961570560); z4 = MulC(z4, -FIX_0_390180644);
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = tmp10 + tmp3; out[7] = tmp10 - tmp3;
  out[1] = tmp11 + tmp2; out[6] = tmp11 - tmp2;
  out[2] = tmp12 + tmp1; out[5] = tmp12 - tmp1;
  out[3] = tmp13 + tmp0; out[4] = tmp13 - tmp0;
}

// range_limit[x & RANGE_MASK] of libjpeg (table centred on 128): 10-bit signed wrap, +128, clamp
__device__ __forceinline__ uint32_t RangeLimit(int32_t x) {
  int32_t v = ((x & 1023) ^ 512) - 512 + 128;
  return (uint32_t)min(max(v, 0), 255);
}

}  // namespace daliamd

#endif  // DALI_AMD_CSRC_JPEG_IDCT_MATH_H_
