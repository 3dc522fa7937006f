// GPU Huffman entropy decoder for baseline JPEG (one interleaved scan, no restart markers) on gfx950.
//
// Reference counterpart: the GPU Huffman stage of nvJPEG inside nvImageCodec, reached from
// ImageDecoder::RunImplImpl (dali/operators/imgcodec/image_decoder.h:810-815).  The output is the same
// column-major coefficient layout the host decoder (dali_amd/host/jpeg_entropy.cpp) produces, so results are
// bit-identical by construction and the IDCT kernel does not care who decoded the stream.
//
// Algorithm: self-synchronising parallel decode (the entropy-coded segment is one long serial bit stream):
//   kernel 1  UnstuffKernel   one workgroup per image removes the 0xFF00 byte stuffing (count / scan / scatter)
//   kernel 2  HuffmanDecodeKernel    one workgroup (1024 lanes) per image:
//     a) lane i takes the i-th slice of the clean stream and decodes the symbols that START in its slice from a
//        guessed state (bit position = slice start, block-in-MCU 0, zig-zag index 0); lane 0 has the true state;
//     b) relaxation: lane i publishes the state it reached to lane i+1; lanes whose input changed decode again.
//        Huffman streams re-synchronise after a few symbols, so this converges in a handful of rounds; since
//        lane 0 is right from the start, round r fixes at least lane r, so the loop is bounded by the lane count
//        and needs no failure path;
//     c) an exclusive scan of the completed-block counts gives every lane its first block ordinal;
//     d) write pass: lanes decode once more and scatter the non-zero coefficients (the buffer is pre-zeroed);
//        DC differences are written as lane-local running sums per component;
//     e) an exclusive scan per component of the lanes' DC sums turns them into absolute DC values.
// HBM traffic: the stream is read a few times (L2 resident: <= 0.5 MB per image) + sparse 2-byte coefficient stores.
#include "common.h"

namespace daliamd {

constexpr int kHuffThreads = 1024;
constexpr int kFastBits = 9;
constexpr int kMinSliceBytes = 32;

struct HuffLds {
  uint16_t fast[4][1 << kFastBits];  // [0],[1] = DC tables 0,1; [2],[3] = AC tables 0,1; entry = (len << 8) | symbol
  int32_t maxcode[4][18];
  int32_t valoff[4][18];
  uint8_t vals[4][256];
  uint8_t zz[64];                    // zig-zag index -> column-major position
  uint8_t blk_comp[12], blk_dc[12], blk_ac[12];  // per block of the MCU: component, DC slot, AC slot
};

// zig-zag scan order expressed in column-major block positions (= the transposed zig-zag)
__device__ __constant__ uint8_t kZigZagColMajor[64] = {
    0, 8, 1, 2, 9, 16, 24, 17, 10, 3, 4, 11, 18, 25, 32, 40, 33, 26, 19, 12, 5, 6, 13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
    28, 21, 14, 7, 15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62,
    55, 63};

// ------------------------------------------------------------------------------------------------ unstuff
__global__ __launch_bounds__(kHuffThreads) void UnstuffKernel(const daliamdJpegHuffDesc *__restrict__ descs) {
  __shared__ int scan[kHuffThreads];
  const daliamdJpegHuffDesc &d = descs[blockIdx.x];
  const int tid = threadIdx.x;
  const int len = d.ecs_len;
  const int chunk = (len + kHuffThreads - 1) / kHuffThreads;
  const int b0 = min(tid * chunk, len), b1 = min(b0 + chunk, len);
  const uint8_t *src = d.ecs;
  int stuffed = 0;
  {
    uint8_t prev = b0 > 0 && b0 < b1 ? src[b0 - 1] : 0;
    for (int i = b0; i < b1; i++) {
      uint8_t b = src[i];
      stuffed += (b == 0 && prev == 0xFF);
      prev = b;
    }
  }
  scan[tid] = stuffed;
  __syncthreads();
  for (int off = 1; off < kHuffThreads; off <<= 1) {  // Hillis-Steele inclusive scan
    int v = tid >= off ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  uint8_t *dst = d.clean;
  int o = b0 - (scan[tid] - stuffed);
  {
    uint8_t prev = b0 > 0 && b0 < b1 ? src[b0 - 1] : 0;
    for (int i = b0; i < b1; i++) {
      uint8_t b = src[i];
      if (!(b == 0 && prev == 0xFF)) dst[o++] = b;
      prev = b;
    }
  }
  if (tid == kHuffThreads - 1) {
    int clean_len = len - scan[kHuffThreads - 1];
    *d.clean_len = clean_len;
    for (int k = 0; k < 24; k++) dst[clean_len + k] = 0xFF;  // all-ones padding: never a valid code
  }
}

// ------------------------------------------------------------------------------------------------ decode
struct BitWindow {
  const uint32_t *words;  // clean stream, dword aligned
  uint64_t w;             // bits [32*k, 32*k + 64) of the stream, MSB first
  int k;
  __device__ __forceinline__ void Seek(uint32_t pos) {
    k = (int)(pos >> 5);
    uint32_t a = __builtin_bswap32(words[k]), b = __builtin_bswap32(words[k + 1]);
    w = ((uint64_t)a << 32) | b;
  }
  // the 32 bits starting at bit `pos` (pos >= 32*k)
  __device__ __forceinline__ uint32_t Peek32(uint32_t pos) {
    uint32_t off = pos - ((uint32_t)k << 5);
    while (off > 32) {
      k++;
      w = (w << 32) | __builtin_bswap32(words[k + 1]);
      off -= 32;
    }
    return (uint32_t)((w << off) >> 32);
  }
};

struct DecodeState {
  uint32_t pos;  // bit position of the next symbol
  int c;         // block index inside the MCU
  int z;         // zig-zag index of the next coefficient (0 = DC)
};
__device__ __forceinline__ uint64_t Pack(const DecodeState &s) {
  return ((uint64_t)s.pos << 16) | ((uint64_t)s.c << 8) | (uint64_t)s.z;
}
__device__ __forceinline__ DecodeState Unpack(uint64_t v) {
  return DecodeState{(uint32_t)(v >> 16), (int)((v >> 8) & 255), (int)(v & 255)};
}

struct DcAcc {
  int sum0 = 0, sum1 = 0, sum2 = 0;  // running sums of the DC differences this lane decoded, per component
  int count = 0;                     // number of DC symbols this lane decoded
};

__device__ __forceinline__ int16_t *BlockPtr(const daliamdJpegHuffDesc &d, const HuffLds &L, int ordinal) {
  if (ordinal >= d.total_blocks) return nullptr;
  const int bpm = d.blocks_per_mcu;
  int mcu = ordinal / bpm, k = ordinal - mcu * bpm;
  int cc = L.blk_comp[k];
  int my = mcu / d.mcus_x, mx = mcu - my * d.mcus_x;
  int bx = mx * d.h_samp[cc] + d.h_of_block[k], by = my * d.v_samp[cc] + d.v_of_block[k];
  return d.coef[cc] + ((size_t)by * d.blocks_x[cc] + bx) * 64;
}

// Decodes the symbols that start in [st.pos, end_bits); returns the number of blocks completed.
// WRITE: also scatters the coefficients, the block in progress at st being block ordinal `ord`.
template <bool WRITE>
__device__ __forceinline__ int DecodeRange(const daliamdJpegHuffDesc &d, const HuffLds &L, BitWindow &bw, DecodeState &st,
                                           uint32_t end_bits, int ord, DcAcc &dc) {
  int nblk = 0;
  const int bpm = d.blocks_per_mcu;
  uint32_t pos = st.pos;
  int c = st.c, z = st.z;
  bw.Seek(pos);
  int16_t *blk = nullptr;
  if (WRITE) blk = BlockPtr(d, L, ord);
  while (pos < end_bits) {
    uint32_t peek = bw.Peek32(pos);
    int slot = z == 0 ? L.blk_dc[c] : L.blk_ac[c];
    uint32_t e = L.fast[slot][peek >> (32 - kFastBits)];
    int len, sym;
    if (e) {
      len = e >> 8;
      sym = e & 255;
    } else {
      uint32_t code16 = peek >> 16;
      len = 16;
      sym = 0;  // invalid code (garbage start state or the padding): consume 16 bits, decode nothing
      for (int l = kFastBits + 1; l <= 16; l++) {
        int cd = (int)(code16 >> (16 - l));
        if (cd <= L.maxcode[slot][l]) {
          len = l;
          sym = L.vals[slot][(cd + L.valoff[slot][l]) & 255];
          break;
        }
      }
    }
    int s = sym & 15;
    int val = 0;
    if (s) {
      uint32_t m = (peek << len) >> (32 - s);
      val = m < (1u << (s - 1)) ? (int)m - (1 << s) + 1 : (int)m;
    }
    pos += len + s;
    if (z == 0) {
      if (WRITE) {
        int comp = L.blk_comp[c];
        int cur;
        if (comp == 0) cur = (dc.sum0 += val);
        else if (comp == 1) cur = (dc.sum1 += val);
        else cur = (dc.sum2 += val);
        if (blk) blk[0] = (int16_t)cur;
        dc.count++;
      }
      z = 1;
    } else {
      int r = sym >> 4;
      if (s == 0) {
        z = r == 15 ? z + 16 : 64;
      } else {
        z += r;
        if (WRITE && blk && z < 64) blk[L.zz[z]] = (int16_t)val;
        z++;
      }
    }
    if (z >= 64) {
      z = 0;
      c = c + 1 == bpm ? 0 : c + 1;
      nblk++;
      if (WRITE) blk = BlockPtr(d, L, ord + nblk);
    }
  }
  st.pos = pos;
  st.c = c;
  st.z = z;
  return nblk;
}

__global__ __launch_bounds__(kHuffThreads) void HuffmanDecodeKernel(const daliamdJpegHuffDesc *__restrict__ descs) {
  __shared__ HuffLds L;
  __shared__ uint64_t state[kHuffThreads + 1];
  __shared__ int iscan[kHuffThreads];
  __shared__ int dscan[3][kHuffThreads];
  const daliamdJpegHuffDesc &d = descs[blockIdx.x];
  const int tid = threadIdx.x;
  // ---- tables ----
  for (int t = tid; t < 4 * (1 << kFastBits); t += kHuffThreads) L.fast[t >> kFastBits][t & ((1 << kFastBits) - 1)] = 0;
  if (tid < 64) L.zz[tid] = kZigZagColMajor[tid];
  if (tid < 12) {
    int comp = tid < d.blocks_per_mcu ? d.comp_of_block[tid] : 0;
    L.blk_comp[tid] = (uint8_t)comp;
    L.blk_dc[tid] = d.dc_sel[comp] & 1;
    L.blk_ac[tid] = 2 + (d.ac_sel[comp] & 1);
  }
  for (int t = tid; t < 4 * 256; t += kHuffThreads) L.vals[t >> 8][t & 255] = d.vals[t >> 8][t & 255];
  __syncthreads();
  if (tid < 4) {
    // canonical code assignment (ITU-T T.81 Annex C), one lane per table
    int code = 0, p = 0;
    for (int l = 1; l <= 16; l++) {
      int n = d.bits[tid][l - 1];
      L.valoff[tid][l] = p - code;
      for (int i = 0; i < n; i++, p++, code++) {
        if (l <= kFastBits) {
          int first = code << (kFastBits - l);
          uint16_t e = (uint16_t)((l << 8) | L.vals[tid][p & 255]);
          for (int j = 0; j < (1 << (kFastBits - l)); j++) L.fast[tid][(first + j) & ((1 << kFastBits) - 1)] = e;
        }
      }
      L.maxcode[tid][l] = n ? code - 1 : -1;
      code <<= 1;
    }
  }
  __syncthreads();

  const int clean_len = *d.clean_len;
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  // slice size in bytes (multiple of 4) such that the lanes cover the stream
  const int slice = max(kMinSliceBytes, ((clean_len + kHuffThreads - 1) / kHuffThreads + 3) & ~3);
  const uint32_t my_begin = min((uint32_t)tid * (uint32_t)slice * 8u, total_bits);
  const uint32_t my_end = min((uint32_t)(tid + 1) * (uint32_t)slice * 8u, total_bits);
  BitWindow bw;
  bw.words = reinterpret_cast<const uint32_t *>(d.clean);

  // ---- (a) speculative decode ----
  DecodeState in{my_begin, 0, 0}, out;
  DcAcc unused;
  state[tid] = Pack(in);
  out = in;
  int nblk = 0;
  if (in.pos < my_end) nblk = DecodeRange<false>(d, L, bw, out, my_end, 0, unused);
  // ---- (b) relaxation ----
  for (int round = 0; round <= kHuffThreads; round++) {
    __syncthreads();
    int changed = 0;
    uint64_t o = Pack(out);
    if (tid + 1 < kHuffThreads && my_begin < total_bits && state[tid + 1] != o) {
      state[tid + 1] = o;
      changed = 1;
    }
    if (!__syncthreads_or(changed)) break;
    uint64_t ni = state[tid];
    if (ni != Pack(in)) {
      in = Unpack(ni);
      out = in;
      nblk = 0;
      if (in.pos < my_end) nblk = DecodeRange<false>(d, L, bw, out, my_end, 0, unused);
    }
  }
  // ---- (c) first block ordinal of every lane ----
  iscan[tid] = nblk;
  __syncthreads();
  for (int off = 1; off < kHuffThreads; off <<= 1) {
    int v = tid >= off ? iscan[tid - off] : 0;
    __syncthreads();
    iscan[tid] += v;
    __syncthreads();
  }
  const int ord = iscan[tid] - nblk;
  // ---- (d) write pass ----
  DcAcc dc;
  DecodeState ws = in;
  if (in.pos < my_end) DecodeRange<true>(d, L, bw, ws, my_end, ord, dc);
  // ---- (e) absolute DC values: exclusive scan of the lane sums, per component ----
  dscan[0][tid] = dc.sum0;
  dscan[1][tid] = dc.sum1;
  dscan[2][tid] = dc.sum2;
  __syncthreads();
  for (int off = 1; off < kHuffThreads; off <<= 1) {
    int v0 = 0, v1 = 0, v2 = 0;
    if (tid >= off) {
      v0 = dscan[0][tid - off];
      v1 = dscan[1][tid - off];
      v2 = dscan[2][tid - off];
    }
    __syncthreads();
    dscan[0][tid] += v0;
    dscan[1][tid] += v1;
    dscan[2][tid] += v2;
    __syncthreads();
  }
  const int base0 = dscan[0][tid] - dc.sum0, base1 = dscan[1][tid] - dc.sum1, base2 = dscan[2][tid] - dc.sum2;
  if (dc.count > 0 && (base0 | base1 | base2)) {
    // blocks whose DC this lane decoded: ordinals [first, first + count)
    const int first = ord + (in.z != 0 ? 1 : 0);
    for (int k = 0; k < dc.count; k++) {
      int16_t *p = BlockPtr(d, L, first + k);
      if (!p) break;
      int bi = (first + k) % d.blocks_per_mcu;
      int cc = L.blk_comp[bi];
      int add = cc == 0 ? base0 : cc == 1 ? base1 : base2;
      p[0] = (int16_t)(p[0] + add);
    }
  }
  // the segment must hold every block the frame header promises (the padding may add garbage after them)
  if (tid == 0 && iscan[kHuffThreads - 1] < d.total_blocks) *d.status = 2;
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdJpegHuffmanRun(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n) {
  if (n == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanRun: invalid argument");
  hipLaunchKernelGGL(daliamd::UnstuffKernel, dim3(n), dim3(daliamd::kHuffThreads), 0, (hipStream_t)stream, descs_dev);
  hipLaunchKernelGGL(daliamd::HuffmanDecodeKernel, dim3(n), dim3(daliamd::kHuffThreads), 0, (hipStream_t)stream, descs_dev);
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

}  // extern "C"
