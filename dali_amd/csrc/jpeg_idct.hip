// Batched dequantise + 8x8 inverse DCT for gfx950 (wave64).
//
// Arithmetic: libjpeg-turbo's accurate integer IDCT ("islow", CONST_BITS 13 / PASS1_BITS 2),
// i.e. what the reference's CPU decoder computes through nvImageCodec
// (dali/operators/imgcodec/image_decoder.h:289-305,810-815).  Integer math => bit-exact.
//
// Mapping: 8 lanes per 8x8 block, 8 blocks per wave, 32 blocks per 256-thread workgroup.
//   pass 1: lane (b, c) loads column c of block b with ONE 16-byte load (blocks are stored
//           column-major exactly for this), dequantises and runs the column butterfly in
//           registers;
//   LDS transpose (int32, block stride padded to 72 dwords => conflict-free ds_write_b32);
//   pass 2: lane (b, r) reads row r with two ds_read_b128, runs the row butterfly,
//           range-limits and stores 8 output bytes with one 8-byte store.
// HBM traffic per block: 128 B read + 64 B written; no other global traffic.
#include "common.h"
#include "jpeg_idct_math.h"

namespace daliamd {

constexpr int kIdctThreads = 256;
constexpr int kBlocksPerWg = kIdctThreads / 8;
constexpr int kLdsBlockStride = 72;  // dwords; 64 + 8 padding

typedef int16_t short8 __attribute__((ext_vector_type(8)));
typedef uint16_t ushort8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(kIdctThreads) void JpegIdctKernel(const daliamdJpegIdctDesc *__restrict__ descs,
                                                               int ndesc, int total_wg) {
  __shared__ int32_t lds[kBlocksPerWg * kLdsBlockStride];
  int wg = XcdRemap(blockIdx.x, total_wg);
  if (wg < 0) return;
  int di = FindDesc(descs, ndesc, wg);
  const daliamdJpegIdctDesc &d = descs[di];
  int tid = threadIdx.x;
  int lb = tid >> 3;  // local block
  int c = tid & 7;    // column in pass 1, row in pass 2
  int blk = (wg - d.wg_start) * kBlocksPerWg + lb;
  bool active = blk < d.nblocks;
  // position of the block in the component: raster order over the whole component or over the requested rectangle
  int by, bx;
  if (d.rect_w > 0) {
    by = blk / d.rect_w;
    bx = d.rect_x0 + (blk - by * d.rect_w);
    by += d.rect_y0;
  } else {
    by = blk / d.blocks_x;
    bx = blk - by * d.blocks_x;
  }
  if (active) {
    short8 v = *reinterpret_cast<const short8 *>(d.coef + ((size_t)by * d.blocks_x + bx) * 64 + c * 8);
    ushort8 q = *reinterpret_cast<const ushort8 *>(d.quant + c * 8);
    int32_t in[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; r++) in[r] = (int32_t)v[r] * (int32_t)q[r];
    Butterfly8(in, o);
    int32_t *w = lds + lb * kLdsBlockStride + c;
#pragma unroll
    for (int r = 0; r < 8; r++) w[r * 8] = Descale(o[r], CONST_BITS - PASS1_BITS);
  }
  __syncthreads();
  if (active) {
    const int4 *rp = reinterpret_cast<const int4 *>(lds + lb * kLdsBlockStride + c * 8);
    int4 a = rp[0], b = rp[1];
    int32_t in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int32_t o[8];
    Butterfly8(in, o);
    const int S = CONST_BITS + PASS1_BITS + 3;
    uint32_t lo = RangeLimit(Descale(o[0], S)) | (RangeLimit(Descale(o[1], S)) << 8) |
                  (RangeLimit(Descale(o[2], S)) << 16) | (RangeLimit(Descale(o[3], S)) << 24);
    uint32_t hi = RangeLimit(Descale(o[4], S)) | (RangeLimit(Descale(o[5], S)) << 8) |
                  (RangeLimit(Descale(o[6], S)) << 16) | (RangeLimit(Descale(o[7], S)) << 24);
    uint2 *dst = reinterpret_cast<uint2 *>(d.plane + (size_t)(by * 8 + c) * d.pitch + bx * 8);
    *dst = make_uint2(lo, hi);
  }
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdJpegIdctSetup(daliamdJpegIdctDesc *descs, int n, int *num_workgroups) {
  DALIAMD_REQUIRE(descs && num_workgroups && n >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegIdctSetup: NULL argument");
  int wg = 0;
  for (int i = 0; i < n; i++) {
    DALIAMD_REQUIRE(descs[i].nblocks >= 0 && descs[i].blocks_x > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegIdctSetup: desc %d has invalid block counts", i);
    DALIAMD_REQUIRE(descs[i].rect_w >= 0 && descs[i].rect_x0 >= 0 && descs[i].rect_y0 >= 0 &&
                        descs[i].rect_x0 + descs[i].rect_w <= descs[i].blocks_x &&
                        (descs[i].rect_w == 0 || descs[i].nblocks % descs[i].rect_w == 0),
                    DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegIdctSetup: desc %d: bad block rectangle", i);
    DALIAMD_REQUIRE((descs[i].pitch & 7) == 0 && descs[i].pitch >= descs[i].blocks_x * 8,
                    DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegIdctSetup: desc %d: pitch %d must be a multiple of 8 and >= %d", i,
                    descs[i].pitch, descs[i].blocks_x * 8);
    descs[i].wg_start = wg;
    wg += (descs[i].nblocks + daliamd::kBlocksPerWg - 1) / daliamd::kBlocksPerWg;
  }
  *num_workgroups = wg;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegIdctRun(daliamdStream_t stream, const daliamdJpegIdctDesc *descs_dev, int n,
                                   int num_workgroups) {
  if (n == 0 || num_workgroups == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_workgroups > 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegIdctRun: invalid argument");
  {
    daliamd::KernelTimer timer("JpegIdctKernel", (hipStream_t)stream);
    hipLaunchKernelGGL(daliamd::JpegIdctKernel, dim3(daliamd::XcdGrid(num_workgroups)),
                       dim3(daliamd::kIdctThreads), 0, (hipStream_t)stream, descs_dev, n, num_workgroups);
  }
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
