// GPU Huffman entropy decoder for baseline JPEG (one interleaved scan, no restart markers) on gfx950.
//
// Reference counterpart: the GPU Huffman stage of nvJPEG inside nvImageCodec, reached from
// ImageDecoder::RunImplImpl (dali/operators/imgcodec/image_decoder.h:810-815).  The output is the same
// column-major coefficient layout the host decoder (dali_amd/host/jpeg_entropy.cpp) produces, so results are
// bit-identical by construction and the IDCT kernel does not care who decoded the stream.
//
// The entropy-coded segment of an image is ONE serial bit stream; the decoder state is (bit position, block index
// inside the MCU, zig-zag index).  Parallelism comes from self-synchronisation: a decoder started from a guessed
// state at an arbitrary byte falls into step with the true decoder after a while.  Work is cut into pieces whose
// size does not depend on the image (so a batch with one 500 KB stream and many 50 KB streams still fills the chip):
//
//   tile     16 KB of the stuffed stream    (un-stuffing, 1024 lanes x 16 bytes)
//   slice    256 bytes of the clean stream  (one decoder lane)
//   segment  116 slices = 29 KB             (one 128-lane workgroup: 12 warm-up lanes + 116 slices)
//
//   1 PrepareKernel         per tile: number of bytes that survive the removal of the 0xFF00 stuffing; and, in extra
//                           workgroups of the same launch (they need nothing but the descriptor), per image: two-level
//                           code tables (11-bit first level, direct second level for the long codes), the symbol-group
//                           tables of the position-only passes and the MCU geometry -> global scratch, copied into
//                           LDS by the decoding workgroups
//   2 UnstuffScatterKernel  per tile: compaction through LDS to its final place in the clean stream
//   3 SyncKernel            per segment: every lane decodes its slice from a guessed state, then the relaxation
//                           "publish the state you reached to the next lane, decode again if your input changed"
//                           runs until nothing changes.  The 12 warm-up lanes replay the end of the previous
//                           segment so that the first slice of the segment starts from the true state with
//                           overwhelming probability.  No values are extracted in this pass, so one table look-up
//                           steps over up to three symbols; every decode also notes the state at the slice's midpoint
//   4 PropagateKernel       per image, serial over its segments: checks that every segment started from the state
//                           its predecessor ended in (if not - pathological streams - repairs it with the same
//                           relaxation, so correctness never depends on luck), assigns block ordinals
//   5 WriteKernel           per segment, one lane per HALF slice (232 lanes in 256 threads): decodes once more from
//                           the now-known states, extracting the values, and appends one 32-bit record per symbol to
//                           the image's record stream (sequential per lane); DC values as lane-local running sums of
//                           the differences; notes where every block starts
//   6 DcScanKernel          per segment: prefix sums of the per-lane DC sums -> DC level at the start of every lane
//   7 ExpandKernel          per block: builds the 8x8 block from its records in LDS, adds the DC level and either
//                           dequantises + inverse-transforms it on the spot and stores the 8x8 samples to the component
//                           plane (fused output, the default of the callers) or stores the coefficients as ONE full
//                           128-byte line (no zero-fill, no partial writes, no read-modify-write)
//
// What bounds the decode loops (measured): the position-only pass is a latency chain - a wave's step is
// ~55 dependent instructions and the kernel lasts as long as the workgroup with the longest chain of slices that do
// not self-synchronise - so its steps were made fewer (symbol groups); the write pass is VALU-issue bound (a wave64
// instruction occupies a SIMD16 for 4 cycles and divergent branches execute the union of their bodies), hence short
// lanes, 4 waves per SIMD, branch-free DC / refill code, a two-dword bit window fed from an LDS ring, and no
// vector-memory instruction between two wave-uniform points.
#include <cstring>
#include "common.h"
#include "jpeg_idct_math.h"

namespace daliamd {

constexpr int kTileThreads = 1024;
constexpr int kTileBytes = kTileThreads * 16;
constexpr int kSliceBytes = 256;
constexpr int kSegThreads = 128;
constexpr int kWarmLanes = 12;
constexpr int kSegLanes = kSegThreads - kWarmLanes;
constexpr int kSegBytes = kSegLanes * kSliceBytes;
constexpr int kFastBits = 11;
constexpr int kL2Entries = 512;
constexpr int kCleanPadBytes = 40;

// Explicit global address space: a generic pointer would make these `flat` accesses, which count against the LDS
// counter as well and would serialise the table look-ups behind the stream prefetch.
using GlobalWords = const uint32_t __attribute__((address_space(1)));
using GlobalCoef = int16_t __attribute__((address_space(1)));
using GlobalBytes = uint8_t __attribute__((address_space(1)));
using GlobalU32 = uint32_t __attribute__((address_space(1)));

// ------------------------------------------------------------------------------------------------ scratch layout
// The synchronisation passes work on slices of kSliceBytes; the write pass splits every slice at its midpoint into
// two WRITE LANES (the time of that pass is the length of its longest lane, and the machine has room for twice the
// lanes).  The state at the midpoint is a by-product of the final synchronisation decode of the slice.
struct LaneRec {      // one per slice
  uint64_t in, out;   // packed decoder state at the start / end of the slice
  uint64_t mid;       // ... at the first symbol that starts at or behind the midpoint (= in when in is behind it)
  int32_t nblk;       // blocks completed inside the slice
  int32_t nsym;       // symbols that start inside the slice = records the write pass emits for it
  int32_t nblk_a;     // ... of them before the midpoint
  int32_t nsym_a;
  int32_t dc[2][3];   // write pass: sum of the DC differences each half holds, per component
  int32_t base[2][3]; // DcScanKernel: DC level at the start of each half, per component
};
static_assert(sizeof(LaneRec) == 88, "ExpandKernel addresses the base[] words directly");
struct SegRec {  // one per segment
  uint64_t out;  // state at the end of the segment
  int32_t nblk_total, block_base;
  int32_t nsym_total, rec_base;
  int32_t dc_total[3];
  int32_t reserved;
};
// The write pass does not scatter 2-byte coefficients into the (185 MB per batch) coefficient arrays - that costs
// 3.6x the algorithmic HBM traffic in partial-line writes plus a zero-fill plus a read-modify-write pass for the DC
// prediction.  It appends one 32-bit RECORD per symbol to a per-image stream (sequential per lane, so the lines
// fill up in L2) and notes where every block starts; ExpandKernel then builds each 8x8 block in LDS and stores it
// as one full 128-byte line, adding the DC level on the way.
//   record: bits 0-15 value (AC coefficient, or the lane-local running sum of the DC differences),
//           bits 16-21 zig-zag index of the coefficient, bit 22 = first record of a block (DC),
//           bit 23 = carries a coefficient (clear for ZRL / end-of-block symbols)
constexpr uint32_t kRecDc = 1u << 22, kRecValid = 1u << 23;
struct BlockIndex {
  uint32_t first_record;  // index of the block's DC record in the image's record stream
  uint32_t lane;          // write lane (image-wide: 2 * slice + half) that decoded the DC: its LaneRec holds the DC level
};

struct HuffTables {
  uint16_t fast[4][1 << kFastBits];  // [0],[1] = DC tables 0,1; [2],[3] = AC tables 0,1
  uint16_t l2[4][kL2Entries];        // codes longer than kFastBits, indexed by (16-bit code window) - l2_first
  int32_t l2_first[4];
  int32_t l2_size[4];                // entries in use; -1: the long codes span more than kL2Entries -> search
  int32_t maxcode[4][18];            // canonical-code search tables (T.81 F.2.2.3), the fallback
  int32_t valoff[4][18];
  uint8_t vals[4][256];
  uint8_t zz[64];                    // zig-zag index -> column-major position
  uint8_t blk_comp[16];              // component of the k-th block of the MCU
  int32_t blk_sx[12], blk_sy[12];    // coefficient-array stride (elements) per MCU column / MCU row
  GlobalCoef *blk_base[12];          // address of that block in MCU (0, 0)
  uint32_t dc_mask, ac_mask;         // bit k: table selector of the k-th block of the MCU
  int32_t bpm, mcus_x, total_blocks;
  // region-of-interest decode: blocks outside their component's rectangle are parsed but not stored
  int32_t use_rect;
  int32_t last_ordinal;              // first block ordinal behind the last MCU row that is needed
  int32_t reserved;
  uint8_t blk_hs[16], blk_vs[16], blk_ho[16], blk_vo[16];  // block position = (mx*hs + ho, my*vs + vo)
  int32_t blk_rect[12][4];           // {x0, y0, x1, y1} of the block's component
  // fused output (dequantise + IDCT inside ExpandKernel): plane of the block's component, or all null
  GlobalBytes *blk_plane[12];
  int32_t blk_pitch[12];
};
static_assert(sizeof(HuffTables) % 16 == 0, "copied with 16-byte accesses");

// Tables of the position-only passes (SyncKernel / PropagateKernel).  They do not extract values, so one look-up may
// step over a GROUP of up to three symbols of one block: a 32-bit entry holds, for the kFastBits-bit window,
//   bits  0-13  the whole group:  z advance (7 bits) | bits used << 7 (5 bits) | symbol count << 12 (1..3)
//   bits 14-23  what precedes the group's LAST symbol: z advance (6 bits) | bits used << 6 (4 bits); zero for a
//               single symbol.  The group may be taken when these symbols leave the block open and the last one
//               still starts inside the range - decided per step
//   bits 24-31  groups of three only: the first symbol alone, bits used (4 bits) | (z advance - 1) << 4, for the
//               (rare) step that cannot take the group; with two symbols the fields above already describe it
// A group continues behind a symbol when that one is not an end-of-block and the CODE of the next lies inside the
// window behind it (its magnitude bits need not).  DC entries continue into the AC table of their block when all the
// blocks that use the DC table use the same AC table.
// On the ImageNet-like bench set (tools/sync_sim.cpp): pairs 0.61x the steps, this 0.54x.
struct SyncTables {
  uint32_t t32[4][1 << kFastBits];   // [0],[1] = DC tables 0,1; [2],[3] = AC tables 0,1; 0 = code longer than the window
  uint16_t l2[4][kL2Entries];
  int32_t l2_first[4];
  int32_t l2_size[4];
  int32_t maxcode[4][18];
  int32_t valoff[4][18];
  uint8_t vals[4][256];
  uint32_t dc_mask, ac_mask;
  int32_t bpm, reserved;
};
static_assert(sizeof(SyncTables) % 16 == 0, "copied with 16-byte accesses");
__host__ __device__ __forceinline__ uint32_t SyncGroup(uint32_t z, uint32_t used, uint32_t count) {
  return z | (used << 7) | (count << 12);
}
constexpr int kSyncGroup = 3;

struct ScratchLayout {
  size_t tile_kept, clean, tables, sync_tables, lanes, segs, records, blocks, total;
};
__host__ __device__ inline size_t AlignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }
__host__ __device__ inline ScratchLayout MakeLayout(int ecs_len, int num_tiles, int num_segments, int total_blocks) {
  ScratchLayout l;
  size_t o = 16;  // [0]: int32 clean_len
  l.tile_kept = o;
  o += AlignUp(sizeof(int32_t) * (size_t)num_tiles, 16);
  l.clean = o;
  o += AlignUp((size_t)ecs_len + 64, 16);
  l.tables = o;
  o += sizeof(HuffTables);
  l.sync_tables = o;
  o += sizeof(SyncTables);
  l.lanes = o;
  o += sizeof(LaneRec) * (size_t)num_segments * kSegLanes;
  l.segs = o;
  o += AlignUp(sizeof(SegRec) * (size_t)num_segments, 16);
  // every symbol consumes at least 2 bits (streams with a 1-bit code are not eligible): <= 4 records per byte
  l.records = o;
  o += sizeof(uint32_t) * (4 * (size_t)ecs_len + 64);
  l.blocks = o;
  o += sizeof(BlockIndex) * ((size_t)total_blocks + 1);
  l.total = AlignUp(o, 256);
  return l;
}
inline int NumTiles(int head, int len) {
  int t = (head + len + kTileBytes - 1) / kTileBytes;
  return t > 0 ? t : 1;
}
inline int NumSegments(int len) { return len > kSegBytes ? (len + kSegBytes - 1) / kSegBytes : 1; }

struct ImageRef {
  const daliamdJpegHuffDesc *d;
  int local;  // tile / segment index inside the image
};
// workgroup -> (image, tile) / (image, segment); descriptors are sorted by their start indices
template <bool TILES>
__device__ __forceinline__ ImageRef FindImage(const daliamdJpegHuffDesc *descs, int n, int wg) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    int start = TILES ? descs[mid].tile_start : descs[mid].seg_start;
    if (start <= wg) lo = mid; else hi = mid - 1;
  }
  return ImageRef{descs + lo, wg - (TILES ? descs[lo].tile_start : descs[lo].seg_start)};
}

// Exclusive scan of one int per lane over the workgroup (NW waves); returns the exclusive prefix, sets `total`.
template <int NW>
__device__ __forceinline__ int WorkgroupExclusiveScan(int v, int *wave_sums, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    int s = wave_sums[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();  // wave_sums may be reused by the caller's next scan
  total = tot;
  return base + incl - v;
}

// ------------------------------------------------------------------------------------------------ un-stuffing
// 16 bytes of the stuffed stream per lane -> mask of the bytes that stay (bit j = byte j).
struct TileChunk {
  uint32_t w[4];
  uint32_t keep;
};
__device__ __forceinline__ TileChunk LoadChunk(const daliamdJpegHuffDesc &d, int tile) {
  const int head = (int)(reinterpret_cast<uintptr_t>(d.ecs) & 15);  // bytes between the 16-byte boundary and the segment
  const int end = head + d.ecs_len;
  const int g = tile * kTileBytes + (int)threadIdx.x * 16;
  TileChunk c{{0, 0, 0, 0}, 0};
  uint32_t prev = 0;
  if (g < end) {
    GlobalWords *p = (GlobalWords *)__builtin_assume_aligned((const void *)(d.ecs - head + g), 16);
    c.w[0] = p[0]; c.w[1] = p[1]; c.w[2] = p[2]; c.w[3] = p[3];
    if (g > head) prev = ((const GlobalBytes *)(d.ecs - head))[g - 1];
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    uint32_t b = (c.w[j >> 2] >> (8 * (j & 3))) & 255u;
    bool valid = g + j >= head && g + j < end;
    bool stuffed = b == 0 && prev == 0xFF && g + j > head;  // the first byte of the segment has no predecessor
    if (valid && !stuffed) c.keep |= 1u << j;
    prev = b;
  }
  return c;
}

// First pass of the un-stuffing: bytes each tile keeps (PrepareKernel).
__device__ __forceinline__ void CountTile(const daliamdJpegHuffDesc *__restrict__ descs, int n, int tile, int *wave_sums) {
  const ImageRef r = FindImage<true>(descs, n, tile);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  TileChunk c = LoadChunk(d, r.local);
  int total;
  WorkgroupExclusiveScan<kTileThreads / 64>(__popc(c.keep), wave_sums, total);
  if (threadIdx.x == 0) reinterpret_cast<int32_t *>(d.scratch + lay.tile_kept)[r.local] = total;
}

__global__ __launch_bounds__(kTileThreads) void UnstuffScatterKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n) {
  __shared__ uint32_t stage[kTileBytes / 4 + 4];
  __shared__ int wave_sums[kTileThreads / 64];
  const ImageRef r = FindImage<true>(descs, n, blockIdx.x);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const int tid = threadIdx.x;
  // clean-stream position of this tile = bytes kept by the tiles before it
  const int32_t *tile_kept = reinterpret_cast<const int32_t *>(d.scratch + lay.tile_kept);
  int before = 0, base;
  for (int t = tid; t < r.local; t += kTileThreads) before += tile_kept[t];
  WorkgroupExclusiveScan<kTileThreads / 64>(before, wave_sums, base);
  const int shift = base & 3;  // stage byte i <-> clean byte (base - shift) + i
  TileChunk c = LoadChunk(d, r.local);
  int total;
  int o = shift + WorkgroupExclusiveScan<kTileThreads / 64>(__popc(c.keep), wave_sums, total);
  uint8_t *stage_b = reinterpret_cast<uint8_t *>(stage);
  if (c.keep == 0xFFFFu && (o & 3) == 0) {
    uint32_t *p = stage + (o >> 2);  // common case: nothing to drop, dword aligned
    p[0] = c.w[0]; p[1] = c.w[1]; p[2] = c.w[2]; p[3] = c.w[3];
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (c.keep & (1u << j)) stage_b[o++] = (uint8_t)(c.w[j >> 2] >> (8 * (j & 3)));
  }
  __syncthreads();
  GlobalBytes *dst_b = (GlobalBytes *)(d.scratch + lay.clean) + (base - shift);
  GlobalU32 *dst_w = (GlobalU32 *)dst_b;
  const int lo = shift, hi = shift + total;  // stage bytes [lo, hi) belong to this tile
  for (int w = tid; w * 4 < hi; w += kTileThreads) {
    if (w * 4 >= lo && w * 4 + 4 <= hi) {
      dst_w[w] = stage[w];
    } else {  // first / last dword, shared with the neighbouring tiles: byte stores
      for (int b = 0; b < 4; b++)
        if (w * 4 + b >= lo && w * 4 + b < hi) dst_b[w * 4 + b] = stage_b[w * 4 + b];
    }
  }
  if (r.local == d.num_tiles - 1) {
    const int clean_len = base + total;
    if (tid == 0) *reinterpret_cast<int32_t *>(d.scratch) = clean_len;
    // all-ones padding: never a valid code, lets the bit window run past the end
    if (tid < kCleanPadBytes) ((GlobalBytes *)(d.scratch + lay.clean))[clean_len + tid] = 0xFF;
  }
}

// ------------------------------------------------------------------------------------------------ tables
// Table entry: bits 0-6 zig-zag advance (1..64), 7-11 bits consumed (code length + magnitude bits s), 12-15 s.
//   DC symbol (category s):  advance 1
//   AC symbol (run r, size s): s != 0: r + 1;  ZRL (0xF0): 16;  any other s == 0 (EOB): 64 = "to the end of the block"
__host__ __device__ __forceinline__ uint32_t MakeEntry(int len, int sym, bool is_dc) {
  int s = sym & 15, r = sym >> 4;
  int adv = is_dc ? 1 : (s ? r + 1 : (r == 15 ? 16 : 64));
  return (uint32_t)((s << 12) | ((len + s) << 7) | adv);
}

// zig-zag scan order expressed in column-major block positions (= the transposed zig-zag)
__device__ __constant__ uint8_t kZigZagColMajor[64] = {
    0, 8, 1, 2, 9, 16, 24, 17, 10, 3, 4, 11, 18, 25, 32, 40, 33, 26, 19, 12, 5, 6, 13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
    28, 21, 14, 7, 15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62,
    55, 63};

template <int THREADS, typename Tables>
__device__ __forceinline__ void CopyTables(Tables &dst, const Tables *src) {
  const uint4 *s = reinterpret_cast<const uint4 *>(src);
  uint4 *t = reinterpret_cast<uint4 *>(&dst);
  for (int i = threadIdx.x; i < (int)(sizeof(Tables) / 16); i += THREADS) t[i] = s[i];
}

// Code tables of one stream (all threads of a PrepareKernel workgroup).
__device__ __forceinline__ void BuildTables(const daliamdJpegHuffDesc &d, HuffTables &L) {
  constexpr int NT = kTileThreads;
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const int tid = threadIdx.x;
  {
    uint4 *z = reinterpret_cast<uint4 *>(&L);
    for (int i = tid; i < (int)(sizeof(HuffTables) / 16); i += NT) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  if (tid < 64) L.zz[tid] = kZigZagColMajor[tid];
  for (int t = tid; t < 4 * 256; t += NT) L.vals[t >> 8][t & 255] = d.vals[t >> 8][t & 255];
  if (tid < 12 && tid < d.blocks_per_mcu) {
    const int comp = d.comp_of_block[tid];
    L.blk_comp[tid] = (uint8_t)comp;
    L.blk_sx[tid] = d.h_samp[comp] * 64;
    L.blk_sy[tid] = d.v_samp[comp] * d.blocks_x[comp] * 64;
    L.blk_base[tid] = (GlobalCoef *)d.coef[comp] + ((size_t)d.v_of_block[tid] * d.blocks_x[comp] + d.h_of_block[tid]) * 64;
    L.blk_hs[tid] = (uint8_t)d.h_samp[comp];
    L.blk_vs[tid] = (uint8_t)d.v_samp[comp];
    L.blk_ho[tid] = d.h_of_block[tid];
    L.blk_vo[tid] = d.v_of_block[tid];
    for (int j = 0; j < 4; j++) L.blk_rect[tid][j] = d.rect[comp][j];
    L.blk_plane[tid] = (GlobalBytes *)d.plane[comp];
    L.blk_pitch[tid] = d.plane_pitch[comp];
  }
  if (tid == 0) {
    uint32_t dc_mask = 0, ac_mask = 0;
    for (int k = 0; k < d.blocks_per_mcu; k++) {
      int comp = d.comp_of_block[k];
      dc_mask |= (uint32_t)(d.dc_sel[comp] & 1) << k;
      ac_mask |= (uint32_t)(d.ac_sel[comp] & 1) << k;
    }
    L.dc_mask = dc_mask;
    L.ac_mask = ac_mask;
    L.bpm = d.blocks_per_mcu;
    L.mcus_x = d.mcus_x;
    L.total_blocks = d.total_blocks;
    // region of interest: any non-empty rectangle switches the filter on
    int use_rect = 0, last_mcu_row = 0;
    for (int k = 0; k < d.blocks_per_mcu; k++) {
      int comp = d.comp_of_block[k];
      if (d.rect[comp][2] > d.rect[comp][0] && d.rect[comp][3] > d.rect[comp][1]) {
        use_rect = 1;
        int rows = (d.rect[comp][3] + d.v_samp[comp] - 1) / d.v_samp[comp];  // MCU rows up to the rectangle's bottom
        last_mcu_row = rows > last_mcu_row ? rows : last_mcu_row;
      }
    }
    L.use_rect = use_rect;
    int last = last_mcu_row * d.mcus_x * d.blocks_per_mcu;
    L.last_ordinal = use_rect && last < d.total_blocks ? last : d.total_blocks;
  }
  __syncthreads();
  if (tid < 4) {
    // canonical code assignment (ITU-T T.81 Annex C): per code length the largest code and the symbol offset
    int code = 0, p = 0;
    int l2_first = 1 << 16, l2_end = 0;
    for (int l = 1; l <= 16; l++) {
      const int n = d.bits[tid][l - 1];
      L.valoff[tid][l] = p - code;
      if (n && l > kFastBits) {
        if (l2_end == 0) l2_first = (code << (16 - l)) & 0xFFFF;
        l2_end = ((code + n) << (16 - l));  // one past the last 16-bit window of the codes seen so far
      }
      p += n;
      code += n;
      L.maxcode[tid][l] = n ? code - 1 : -1;
      code <<= 1;
    }
    const int size = l2_end ? l2_end - l2_first : 0;
    L.l2_first[tid] = l2_first;
    L.l2_size[tid] = size <= kL2Entries ? size : -1;  // -1: too spread out for the direct table, LongCode searches
  }
  __syncthreads();
  // every table entry is found independently: the shortest length whose code range contains the window's prefix
  for (int idx = tid; idx < 4 * (1 << kFastBits); idx += NT) {
    const int t = idx >> kFastBits, w = idx & ((1 << kFastBits) - 1);
    uint16_t e = 0;
    for (int l = 1; l <= kFastBits; l++) {
      const int cd = w >> (kFastBits - l);
      if (cd <= L.maxcode[t][l]) {
        e = (uint16_t)MakeEntry(l, L.vals[t][(cd + L.valoff[t][l]) & 255], t < 2);
        break;
      }
    }
    L.fast[t][w] = e;
  }
  for (int idx = tid; idx < 4 * kL2Entries; idx += NT) {
    const int t = idx / kL2Entries, j = idx % kL2Entries;
    uint16_t e = 0;
    if (j < L.l2_size[t]) {
      const int w = L.l2_first[t] + j;
      for (int l = kFastBits + 1; l <= 16; l++) {
        const int cd = w >> (16 - l);
        if (cd <= L.maxcode[t][l]) {
          e = (uint16_t)MakeEntry(l, L.vals[t][(cd + L.valoff[t][l]) & 255], t < 2);
          break;
        }
      }
    }
    L.l2[t][j] = e;
  }
  __syncthreads();
  const uint4 *s = reinterpret_cast<const uint4 *>(&L);
  uint4 *t = reinterpret_cast<uint4 *>(d.scratch + lay.tables);
  for (int i = tid; i < (int)(sizeof(HuffTables) / 16); i += NT) t[i] = s[i];
  // ---- tables of the position-only passes, straight to global memory ----
  SyncTables *S = reinterpret_cast<SyncTables *>(d.scratch + lay.sync_tables);
  for (int idx = tid; idx < 4 * (1 << kFastBits); idx += NT) {
    const int tb = idx >> kFastBits, w = idx & ((1 << kFastBits) - 1);
    const uint32_t e1 = L.fast[tb][w];
    uint32_t e = 0;
    if (e1) {
      // table the block continues with: an AC table itself, or the one AC table every block of this DC table uses
      int ac = tb;
      if (tb < 2) {
        ac = -1;
        for (int k = 0; k < L.bpm; k++) {
          if ((int)((L.dc_mask >> k) & 1u) != tb) continue;
          const int a = 2 + (int)((L.ac_mask >> k) & 1u);
          ac = ac == -1 || ac == a ? a : -2;
        }
      }
      uint32_t z = e1 & 127, used = (e1 >> 7) & 31, count = 1, zprev = 0, uprev = 0;
      const uint32_t z1 = z, u1 = used;
      while (ac >= 2 && count < (uint32_t)kSyncGroup && z < 64 && used < (uint32_t)kFastBits) {
        const uint32_t e2 = L.fast[ac][(w << used) & ((1 << kFastBits) - 1)];
        const uint32_t z2 = e2 & 127, u2 = (e2 >> 7) & 31, len2 = u2 - (e2 >> 12);
        if (!e2 || used + len2 > (uint32_t)kFastBits) break;  // the next code is not determined by the window
        zprev = z; uprev = used;
        z += z2; used += u2; count++;
      }
      e = SyncGroup(z, used, count) | (zprev << 14) | (uprev << 20);
      if (count == 3) e |= (u1 << 24) | ((z1 - 1) << 28);
    }
    S->t32[tb][w] = e;
  }
  for (int i = tid; i < 4 * kL2Entries; i += NT) S->l2[i / kL2Entries][i % kL2Entries] = L.l2[i / kL2Entries][i % kL2Entries];
  for (int i = tid; i < 4 * 256; i += NT) S->vals[i >> 8][i & 255] = L.vals[i >> 8][i & 255];
  if (tid < 4 * 18) {
    S->maxcode[tid / 18][tid % 18] = L.maxcode[tid / 18][tid % 18];
    S->valoff[tid / 18][tid % 18] = L.valoff[tid / 18][tid % 18];
  }
  if (tid < 4) {
    S->l2_first[tid] = L.l2_first[tid];
    S->l2_size[tid] = L.l2_size[tid];
  }
  if (tid == 0) {
    S->dc_mask = L.dc_mask;
    S->ac_mask = L.ac_mask;
    S->bpm = L.bpm;
    S->reserved = 0;
  }
}

// One launch for the two jobs that only need the descriptors: workgroups [0, n) build the code tables of one stream
// each (a long chain of short phases: they go first so that they run next to the tile workgroups instead of behind
// them), workgroups [n, n + num_tiles) count the bytes their tile of the stuffed stream keeps.
__global__ __launch_bounds__(kTileThreads) void PrepareKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n,
                                                              int num_tiles) {
  __shared__ __attribute__((aligned(16))) HuffTables L;
  __shared__ int wave_sums[kTileThreads / 64];
  if ((int)blockIdx.x < n) BuildTables(descs[blockIdx.x], L);
  else CountTile(descs, n, (int)blockIdx.x - n, wave_sums);
}

// ------------------------------------------------------------------------------------------------ decode
struct DecodeState {
  uint32_t pos;  // bit position of the next symbol
  uint32_t c;    // block index inside the MCU
  uint32_t z;    // zig-zag index of the next coefficient (0 = DC)
};
__device__ __forceinline__ uint64_t Pack(const DecodeState &s) {
  return ((uint64_t)s.pos << 16) | ((uint64_t)s.c << 8) | (uint64_t)s.z;
}
__device__ __forceinline__ DecodeState Unpack(uint64_t v) {
  return DecodeState{(uint32_t)(v >> 16), (uint32_t)((v >> 8) & 255), (uint32_t)(v & 255)};
}
constexpr uint64_t kNoState = ~0ull;  // unpacks to a position past any stream

struct DcAcc {
  int sum0 = 0, sum1 = 0, sum2 = 0;  // running sums of the DC differences this lane decoded, per component
};

// Rare path: the code is longer than kFastBits bits (or is not a code at all).
template <typename Tables>
__device__ __noinline__ uint32_t LongCode(const Tables &L, uint32_t slot, uint32_t peek, bool is_dc) {
  const uint32_t code16 = peek >> 16;
  uint32_t e = 0;
  const int size = L.l2_size[slot];
  if (size >= 0) {
    const int idx = (int)code16 - L.l2_first[slot];
    if (idx >= 0 && idx < size) e = L.l2[slot][idx];
  } else {
    for (int l = kFastBits + 1; l <= 16; l++) {
      int cd = (int)(code16 >> (16 - l));
      if (cd <= L.maxcode[slot][l]) {
        e = MakeEntry(l, L.vals[slot][(cd + L.valoff[slot][l]) & 255], is_dc);
        break;
      }
    }
  }
  // not a code (garbage start state, or the padding behind the stream): consume 16 bits, decode nothing
  return e ? e : MakeEntry(16, 0, is_dc);
}

// Decodes the symbols that start in [st.pos, end_bits); returns the number of blocks completed.  Positions only:
// no value is extracted (synchronisation passes).
struct HalfCount { uint64_t mid; int nblk, nsym; };  // state at the midpoint, blocks / symbols before it
__device__ __forceinline__ int DecodeRange(const SyncTables &L, GlobalWords *__restrict__ words, DecodeState &st,
                                           uint32_t mid_bits, uint32_t end_bits, int &nsym_out, HalfCount &half) {
  int nblk = 0, nsym = 0;
  uint32_t c = st.c, z = st.z;
  int rem = (int)(end_bits - st.pos);  // bits left before the end of the slice (<= 0: done)
  // bit window: hi:lo = stream bits [32k, 32k+64), `off` of hi's bits already consumed; the following dword is in flight
  int k = (int)(st.pos >> 5);
  uint32_t off = st.pos & 31;
  uint32_t hi = __builtin_bswap32(words[k]), lo = __builtin_bswap32(words[k + 1]), nxt = words[k + 2];
  const uint32_t *t32 = &L.t32[0][0];
  const uint32_t dc_mask = L.dc_mask, ac_mask = L.ac_mask, bpm = (uint32_t)L.bpm;
  // symbols that start in [pos, end - limit); called for the two halves of the slice in turn (no per-symbol cost
  // for the midpoint: the first loop simply stops there)
  auto run = [&](int limit) {
    while (rem > limit) {
      const uint32_t peek = (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
      const bool is_dc = z == 0;
      const uint32_t slot = (((is_dc ? dc_mask : ac_mask) >> c) & 1u) + (is_dc ? 0u : 2u);
      uint32_t e = t32[(slot << kFastBits) + (peek >> (32 - kFastBits))];
      if (__builtin_expect(e == 0, 0)) {
        const uint32_t e16 = LongCode(L, slot, peek, is_dc);
        e = SyncGroup(e16 & 127, (e16 >> 7) & 31, 1);
      }
      // the group may be taken when the symbols before its last one leave the block open and the last one starts
      // inside the range: both differences negative <=> the sign bit of their AND is set
      const int zprev = (int)((e >> 14) & 63), uprev = (int)((e >> 20) & 15);
      const int ok = ((int)z + zprev - 64) & (uprev - (rem - limit));
      uint32_t used = (e >> 7) & 31, zinc = e & 127, count = (e >> 12) & 3;
      if (__builtin_expect(ok >= 0, 0)) {  // rare: take the first symbol only
        const bool three = count == 3;
        used = three ? (e >> 24) & 15 : (uint32_t)uprev;
        zinc = three ? ((e >> 28) & 15) + 1 : (uint32_t)zprev;
        count = 1;
      }
      rem -= (int)used;
      off += used;
      z += zinc;
      nsym += (int)count;
      if (off >= 32) {
        hi = lo;
        lo = __builtin_bswap32(nxt);
        k++;
        nxt = words[k + 2];
        off -= 32;
      }
      const bool end_of_block = z >= 64;
      const uint32_t c1 = c + 1 == bpm ? 0 : c + 1;
      z = end_of_block ? 0 : z;
      c = end_of_block ? c1 : c;
      nblk += end_of_block ? 1 : 0;
    }
  };
  run(mid_bits < end_bits ? (int)(end_bits - mid_bits) : 0);
  half.mid = Pack(DecodeState{end_bits - (uint32_t)rem, c, z});
  half.nblk = nblk;
  half.nsym = nsym;
  run(0);
  st.pos = end_bits - (uint32_t)rem;
  st.c = c;
  st.z = z;
  nsym_out = nsym;
  return nblk;
}

// 16 bytes at a 4-byte aligned address in one instruction (gfx9 global accesses need dword alignment only).  Scattered
// accesses cost the texture-address unit one cycle or more PER LANE, so a lane moving its 32 bytes with two of
// these instead of eight dword accesses is the difference between a TA-bound and a VALU-bound write pass.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
using GlobalQuadA4 = u32x4_a4 __attribute__((address_space(1)));

constexpr int kGroupSteps = 8;   // symbols between two wave-uniform "points" of the write pass
constexpr int kRingWords = 16;   // dwords of the clean stream buffered in LDS per lane (ring)
constexpr int kRingNeed = 10;    // a group reads up to dword k + 2 + 7 (eight symbols of at most 27 bits)
constexpr int kWriteThreads = 256;
constexpr int kWriteLanes = 2 * kSegLanes;  // two write lanes per slice of the segment

// Write pass of one slice: decodes the symbols that start in [st.pos, end_bits) once more, now extracting the values,
// and appends one record per symbol to `rec` (this slice's part of the image's record stream; exactly the `nsym`
// records the synchronisation pass counted).  The block in progress at `st` is block ordinal `ord`; every DC record
// registers its block in the block index.  DC records carry the lane-local running sum of the differences of their
// component (ExpandKernel adds the level at the start of the slice).
//
// gfx9 has ONE in-order counter for vector-memory loads and stores: waiting for a load also waits for every store
// issued before it.  So the loop is software-pipelined around wave-uniform points, kGroupSteps symbols apart:
// at a point the lane (1) takes delivery of the 8 stream dwords it requested at the previous point (long arrived:
// the wait costs nothing) and appends them to its LDS ring, (2) stores the records and block-index entries of the
// last 8 symbols, kept in registers until now, (3) requests the next 8 stream dwords.  Between two points there
// is no vector-memory instruction at all: the bit window is refilled from the LDS ring.
__device__ __forceinline__ void WriteRange(const HuffTables &L, GlobalWords *__restrict__ words, DecodeState st,
                                           uint32_t end_bits, int ord, bool live, uint32_t *ring,
                                           GlobalU32 *__restrict__ rec, uint32_t rec_index, uint32_t lane_id,
                                           BlockIndex *__restrict__ blocks, DcAcc &dc) {
  uint32_t c = st.c, z = st.z;
  int rem = (int)(end_bits - st.pos);
  int k = (int)(st.pos >> 5);  // dword index of `hi`
  int kl = k;                  // the ring holds dwords [.., kl)
  uint32_t off = st.pos & 31;
  uint32_t hi = 0, lo = 0, nxt = 0;
  const uint16_t *fast = &L.fast[0][0];
  const uint32_t dc_mask = L.dc_mask, ac_mask = L.ac_mask, bpm = (uint32_t)L.bpm;
  // component of the k-th block of the MCU, two bits each (no LDS look-up on the DC path)
  uint32_t comp_bits = 0;
  for (uint32_t kk = 0; kk < bpm; kk++) comp_bits |= (uint32_t)L.blk_comp[kk] << (2 * kk);
  int ordinal = ord;  // block the next symbol belongs to
  if (live) {         // prologue: 12 dwords straight into the ring (the only exposed memory latency of the pass)
    const GlobalQuadA4 *src = (const GlobalQuadA4 *)(words + k);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const u32x4_a4 v = src[q];
      ring[(k + 4 * q) & (kRingWords - 1)] = v.x;
      ring[(k + 4 * q + 1) & (kRingWords - 1)] = v.y;
      ring[(k + 4 * q + 2) & (kRingWords - 1)] = v.z;
      ring[(k + 4 * q + 3) & (kRingWords - 1)] = v.w;
    }
    kl = k + 12;
    hi = __builtin_bswap32(ring[k & (kRingWords - 1)]);
    lo = __builtin_bswap32(ring[(k + 1) & (kRingWords - 1)]);
    nxt = ring[(k + 2) & (kRingWords - 1)];
  }
  uint32_t pre[4];        // stream dwords [kl, kl + 4) in flight
  bool have_pre = false;
  uint32_t r[kGroupSteps];  // records of the current group
  int nrec = 0;
  int group_ord = ordinal;  // `ordinal` and "inside a block" at the start of the group whose records are in r[]
  bool group_mid_block = z != 0;
  while (__ballot(live || nrec > 0) != 0) {  // wave-uniform trip count
    // ---------------- point ----------------
    if (have_pre) {
#pragma unroll
      for (int q = 0; q < 4; q++) ring[(kl + q) & (kRingWords - 1)] = pre[q];
      kl += 4;
      have_pre = false;
    }
    // Safety net: the usual 4 dwords per point cover 16 bits per symbol.  A lane that could run dry inside the next
    // group (eight symbols in a row longer than that) fetches synchronously; real streams never get here.
    while (__ballot(live && kl - k < kRingNeed) != 0) {
      if (live && kl - k < kRingNeed) {
        const u32x4_a4 v = *(const GlobalQuadA4 *)(words + kl);
        ring[kl & (kRingWords - 1)] = v.x;
        ring[(kl + 1) & (kRingWords - 1)] = v.y;
        ring[(kl + 2) & (kRingWords - 1)] = v.z;
        ring[(kl + 3) & (kRingWords - 1)] = v.w;
        kl += 4;
      }
    }
    if (nrec > 0) {
      if (nrec == kGroupSteps) {  // the common case: the lane's 8 records as two 16-byte stores
        GlobalQuadA4 *dst = (GlobalQuadA4 *)(rec + rec_index);
        dst[0] = u32x4_a4{r[0], r[1], r[2], r[3]};
        dst[1] = u32x4_a4{r[4], r[5], r[6], r[7]};
      } else {
#pragma unroll
        for (int q = 0; q < kGroupSteps; q++)
          if (q < nrec) rec[rec_index + q] = r[q];
      }
      int dc_ord = group_ord + (group_mid_block ? 1 : 0);  // ordinal of the first DC record of the group
#pragma unroll
      for (int q = 0; q < kGroupSteps; q++) {
        if (q < nrec && (r[q] & kRecDc)) {
          if (dc_ord < L.total_blocks) blocks[dc_ord] = BlockIndex{rec_index + q, lane_id};
          dc_ord++;
        }
      }
      rec_index += (uint32_t)nrec;
      nrec = 0;
    }
    if (live && kl - k <= kRingWords - 4) {  // room in the ring: request the next 4 dwords (delivered at the next point)
      const u32x4_a4 a = *(const GlobalQuadA4 *)(words + kl);
      pre[0] = a.x; pre[1] = a.y; pre[2] = a.z; pre[3] = a.w;
      have_pre = true;
    }
    group_ord = ordinal;
    group_mid_block = z != 0;
    // ---------------- 8 symbols, no vector-memory instruction ----------------
#pragma unroll
    for (int j = 0; j < kGroupSteps; j++) {
      if (live) {
        const uint32_t peek = (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
        const bool is_dc = z == 0;
        const uint32_t slot = (((is_dc ? dc_mask : ac_mask) >> c) & 1u) + (is_dc ? 0u : 2u);
        uint32_t e = fast[(slot << kFastBits) + (peek >> (32 - kFastBits))];
        if (__builtin_expect(e == 0, 0)) e = LongCode(L, slot, peek, is_dc);
        const uint32_t used = (e >> 7) & 31, s = e >> 12, adv = e & 127;
        // magnitude bits -> value (T.81 F.2.2.1 EXTEND): the s bits behind the code; s == 0 gives 0
        const uint32_t m = __builtin_amdgcn_ubfe(peek, 32u - used, s);
        const uint32_t full = (1u << s) - 1u;          // 2^s - 1
        int val = (int)m - (int)(m <= (full >> 1) ? full : 0u);
        uint32_t zt = z + adv - 1;  // zig-zag index of the coefficient this symbol carries (AC)
        uint32_t flags = (s && zt < 64) ? kRecValid : 0;
        // DC: the record carries the lane-local running sum of its component's differences (no branches: the three
        // sums live in registers and every step offers them a masked increment)
        const uint32_t comp = (comp_bits >> (2 * c)) & 3u;
        const int dval = is_dc ? val : 0;
        dc.sum0 += comp == 0 ? dval : 0;
        dc.sum1 += comp == 1 ? dval : 0;
        dc.sum2 += comp == 2 ? dval : 0;
        const int sum = comp == 0 ? dc.sum0 : comp == 1 ? dc.sum1 : dc.sum2;
        val = is_dc ? sum : val;
        zt = is_dc ? 0u : zt;
        flags = is_dc ? (kRecDc | kRecValid) : flags;
        r[j] = ((uint32_t)val & 0xFFFFu) | ((zt & 63u) << 16) | flags;
        nrec = j + 1;
        rem -= (int)used;
        off += used;
        z += adv;
        // window refill without a branch: the ring read is unconditional (it re-reads the same dword until k moves)
        const bool refill = off >= 32;
        hi = refill ? lo : hi;
        lo = refill ? __builtin_bswap32(nxt) : lo;
        k += refill ? 1 : 0;
        off &= 31;
        nxt = ring[(k + 2) & (kRingWords - 1)];
        const bool end_of_block = z >= 64;
        const uint32_t c1 = c + 1 == bpm ? 0 : c + 1;
        z = end_of_block ? 0 : z;
        c = end_of_block ? c1 : c;
        ordinal += end_of_block ? 1 : 0;
        live = rem > 0;
      }
    }
  }
}

// One decoder lane of the relaxation.
struct Lane {
  uint32_t begin, end;  // bit range of the slice (clipped to the stream)
  bool active;          // the slice holds data
  uint64_t in = kNoState, out = kNoState, mid = kNoState;
  int nblk = 0, nsym = 0, nblk_a = 0, nsym_a = 0;
};

// "Publish the state you reached to the next lane, decode again if your input changed", until nothing changes.
// state[t] is the published input of lane t; lanes whose `in` already equals it do not decode.  Lane 0's input is
// never written here, so whatever the caller put there is taken as the truth; each round fixes at least one more
// lane, which bounds the loop by the lane count.
template <int THREADS>
__device__ __forceinline__ void Relax(const SyncTables &L, GlobalWords *words, uint64_t *state, Lane &ln) {
  const int tid = threadIdx.x;
  for (int round = 0; round <= THREADS; round++) {
    const uint64_t ni = state[tid];
    if (ln.active && ni != ln.in) {
      ln.in = ni;
      DecodeState st = Unpack(ni);
      ln.nblk = ln.nsym = 0;
      HalfCount half{ni, 0, 0};
      if (st.pos < ln.end) ln.nblk = DecodeRange(L, words, st, ln.begin + kSliceBytes * 4u, ln.end, ln.nsym, half);
      ln.out = Pack(st);
      ln.mid = half.mid;
      ln.nblk_a = half.nblk;
      ln.nsym_a = half.nsym;
    }
    __syncthreads();
    int changed = 0;
    if (ln.active && tid + 1 < THREADS && state[tid + 1] != ln.out) {
      state[tid + 1] = ln.out;
      changed = 1;
    }
    if (!__syncthreads_or(changed)) break;
  }
}

__device__ __forceinline__ Lane MakeLane(long long slice_index, uint32_t total_bits) {
  Lane ln;
  if (slice_index < 0) {
    ln.begin = ln.end = 0;
    ln.active = false;
    return ln;
  }
  const unsigned long long b = (unsigned long long)slice_index * (kSliceBytes * 8ull);
  ln.begin = (uint32_t)(b < total_bits ? b : total_bits);
  ln.end = (uint32_t)(b + kSliceBytes * 8ull < total_bits ? b + kSliceBytes * 8ull : total_bits);
  ln.active = ln.begin < total_bits;
  return ln;
}

__global__ __launch_bounds__(kSegThreads) void SyncKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ __attribute__((aligned(16))) SyncTables L;
  __shared__ uint64_t state[kSegThreads];
  __shared__ int wave_sums[kSegThreads / 64];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const int tid = threadIdx.x, seg = r.local;
  const int clean_len = *reinterpret_cast<const int32_t *>(d.scratch);
  LaneRec *recs = reinterpret_cast<LaneRec *>(d.scratch + lay.lanes) + (size_t)seg * kSegLanes;
  SegRec *segrec = reinterpret_cast<SegRec *>(d.scratch + lay.segs) + seg;
  if (seg > 0 && (long long)seg * kSegBytes >= clean_len) {  // segment behind the end of the stream
    if (tid >= kWarmLanes) recs[tid - kWarmLanes] = LaneRec{kNoState, kNoState, kNoState, 0, 0, 0, 0, {}, {}};
    if (tid == 0) *segrec = SegRec{kNoState, 0, 0, 0, 0, {0, 0, 0}, 0};
    return;
  }
  CopyTables<kSegThreads>(L, reinterpret_cast<const SyncTables *>(d.scratch + lay.sync_tables));
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  // lanes [0, kWarmLanes) replay the last slices of the previous segment, lanes [kWarmLanes, ..) are this segment's
  Lane ln = MakeLane((long long)seg * kSegLanes + tid - kWarmLanes, total_bits);
  state[tid] = Pack(DecodeState{ln.begin, 0, 0});  // the guess; exact for the very first slice of the image
  __syncthreads();
  Relax<kSegThreads>(L, (GlobalWords *)(d.scratch + lay.clean), state, ln);
  int total, total_sym;
  WorkgroupExclusiveScan<kSegThreads / 64>(tid >= kWarmLanes ? ln.nblk : 0, wave_sums, total);
  WorkgroupExclusiveScan<kSegThreads / 64>(tid >= kWarmLanes ? ln.nsym : 0, wave_sums, total_sym);
  if (tid >= kWarmLanes) {
    recs[tid - kWarmLanes] = LaneRec{ln.in, ln.out, ln.mid, ln.nblk, ln.nsym, ln.nblk_a, ln.nsym_a, {}, {}};
    // the last slice with data ends the segment (an empty stream: the first lane passes its input on)
    const bool next_has_data = tid + 1 < kSegThreads && ln.end < total_bits;
    if (ln.active && !next_has_data) *segrec = SegRec{ln.out, total, 0, total_sym, 0, {0, 0, 0}, 0};
    if (total_bits == 0 && tid == kWarmLanes) *segrec = SegRec{Pack(DecodeState{0, 0, 0}), 0, 0, 0, 0, {0, 0, 0}, 0};
  }
}

__global__ __launch_bounds__(kSegThreads) void PropagateKernel(const daliamdJpegHuffDesc *__restrict__ descs) {
  __shared__ __attribute__((aligned(16))) SyncTables L;
  __shared__ uint64_t state[kSegThreads];
  __shared__ int wave_sums[kSegThreads / 64];
  const daliamdJpegHuffDesc &d = descs[blockIdx.x];
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const int tid = threadIdx.x;
  const int clean_len = *reinterpret_cast<const int32_t *>(d.scratch);
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  LaneRec *all_recs = reinterpret_cast<LaneRec *>(d.scratch + lay.lanes);
  SegRec *segs = reinterpret_cast<SegRec *>(d.scratch + lay.segs);
  uint64_t truth = Pack(DecodeState{0, 0, 0});
  int block_base = 0, rec_base = 0;
  bool tables_loaded = false;
  for (int seg = 0; seg < d.num_segments; seg++) {
    if (seg > 0 && (long long)seg * kSegBytes >= clean_len) {
      if (tid == 0) {
        segs[seg].out = truth;
        segs[seg].block_base = block_base;
        segs[seg].rec_base = rec_base;
      }
      continue;
    }
    LaneRec *recs = all_recs + (size_t)seg * kSegLanes;
    if (total_bits != 0 && recs[0].in != truth) {
      // The warm-up lanes did not synchronise before this segment (long flat or periodic content): repair it.
      if (!tables_loaded) {
        CopyTables<kSegThreads>(L, reinterpret_cast<const SyncTables *>(d.scratch + lay.sync_tables));
        tables_loaded = true;
      }
      Lane ln = MakeLane(tid < kSegLanes ? (long long)seg * kSegLanes + tid : -1, total_bits);
      if (tid < kSegLanes) {
        ln.in = recs[tid].in;
        ln.out = recs[tid].out;
        ln.mid = recs[tid].mid;
        ln.nblk = recs[tid].nblk;
        ln.nsym = recs[tid].nsym;
        ln.nblk_a = recs[tid].nblk_a;
        ln.nsym_a = recs[tid].nsym_a;
      }
      state[tid] = tid == 0 ? truth : ln.in;
      __syncthreads();
      Relax<kSegThreads>(L, (GlobalWords *)(d.scratch + lay.clean), state, ln);
      int total, total_sym;
      WorkgroupExclusiveScan<kSegThreads / 64>(ln.nblk, wave_sums, total);
      WorkgroupExclusiveScan<kSegThreads / 64>(ln.nsym, wave_sums, total_sym);
      if (tid < kSegLanes) {
        recs[tid].in = ln.in;
        recs[tid].out = ln.out;
        recs[tid].mid = ln.mid;
        recs[tid].nblk = ln.nblk;
        recs[tid].nsym = ln.nsym;
        recs[tid].nblk_a = ln.nblk_a;
        recs[tid].nsym_a = ln.nsym_a;
        const bool next_has_data = tid + 1 < kSegLanes && ln.end < total_bits;
        if (ln.active && !next_has_data) {
          segs[seg].out = ln.out;
          segs[seg].nblk_total = total;
          segs[seg].nsym_total = total_sym;
        }
      }
      __threadfence();
      __syncthreads();  // the records written above are read below (same workgroup)
    }
    if (tid == 0) {
      segs[seg].block_base = block_base;
      segs[seg].rec_base = rec_base;
    }
    block_base += segs[seg].nblk_total;
    rec_base += segs[seg].nsym_total;
    truth = segs[seg].out;
  }
  // the segment must hold every block the frame header promises (the padding may add garbage after them)
  if (tid == 0 && block_base < d.total_blocks) *d.status = 2;
  if (tid == 0) {
    reinterpret_cast<int32_t *>(d.scratch)[1] = rec_base;    // records of the whole stream
    reinterpret_cast<int32_t *>(d.scratch)[2] = block_base;  // blocks the stream really holds (truncated streams: fewer)
  }
}

__global__ __launch_bounds__(kWriteThreads) void WriteKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ __attribute__((aligned(16))) HuffTables L;
  __shared__ __attribute__((aligned(16))) uint32_t ring[kWriteThreads * kRingWords];
  __shared__ int wave_sums[kWriteThreads / 64];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const int tid = threadIdx.x, seg = r.local;
  const int clean_len = *reinterpret_cast<const int32_t *>(d.scratch);
  if (seg > 0 && (long long)seg * kSegBytes >= clean_len) return;
  CopyTables<kWriteThreads>(L, reinterpret_cast<const HuffTables *>(d.scratch + lay.tables));
  LaneRec *recs = reinterpret_cast<LaneRec *>(d.scratch + lay.lanes) + (size_t)seg * kSegLanes;
  SegRec *segrec = reinterpret_cast<SegRec *>(d.scratch + lay.segs) + seg;
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  // write lane = (slice, half): the first half ends at the slice's midpoint, the second starts from the state the
  // synchronisation pass noted there
  const int slice = tid >> 1, half = tid & 1;
  const bool has = tid < kWriteLanes;
  const Lane ln = MakeLane(has ? (long long)seg * kSegLanes + slice : -1, total_bits);
  const uint32_t mid_bits = min(ln.begin + kSliceBytes * 4u, ln.end);
  const uint32_t end_bits = half ? ln.end : mid_bits;
  uint64_t in = kNoState;
  int nblk = 0, nsym = 0;
  if (has) {
    const LaneRec &rc = recs[slice];
    in = half ? rc.mid : rc.in;
    nblk = half ? rc.nblk - rc.nblk_a : rc.nblk_a;
    nsym = half ? rc.nsym - rc.nsym_a : rc.nsym_a;
  }
  int total;
  const int ord = segrec->block_base + WorkgroupExclusiveScan<kWriteThreads / 64>(nblk, wave_sums, total);
  const int rec_off = segrec->rec_base + WorkgroupExclusiveScan<kWriteThreads / 64>(nsym, wave_sums, total);
  DcAcc dc;
  DecodeState st = Unpack(in);
  // region of interest: lanes behind the last needed block are not decoded again; the lane that STARTS at that
  // block still is, so that the DC record ending the last needed block exists
  const bool live = ln.active && st.pos < end_bits && ord <= L.last_ordinal;
  WriteRange(L, (GlobalWords *)(d.scratch + lay.clean), st, end_bits, ord, live, ring + tid * kRingWords,
             (GlobalU32 *)(d.scratch + lay.records), (uint32_t)rec_off, (uint32_t)((seg * kSegLanes + slice) * 2 + half),
             reinterpret_cast<BlockIndex *>(d.scratch + lay.blocks), dc);
  if (has) {
    recs[slice].dc[half][0] = dc.sum0;
    recs[slice].dc[half][1] = dc.sum1;
    recs[slice].dc[half][2] = dc.sum2;
  }
  int t0, t1, t2;
  WorkgroupExclusiveScan<kWriteThreads / 64>(dc.sum0, wave_sums, t0);
  WorkgroupExclusiveScan<kWriteThreads / 64>(dc.sum1, wave_sums, t1);
  WorkgroupExclusiveScan<kWriteThreads / 64>(dc.sum2, wave_sums, t2);
  if (tid == 0) {
    segrec->dc_total[0] = t0;
    segrec->dc_total[1] = t1;
    segrec->dc_total[2] = t2;
  }
}

// DC prediction: the level at the start of every write lane = sum of the differences of all lanes before it.
__global__ __launch_bounds__(kWriteThreads) void DcScanKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ int wave_sums[kWriteThreads / 64];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const int tid = threadIdx.x, seg = r.local;
  const int clean_len = *reinterpret_cast<const int32_t *>(d.scratch);
  if (seg > 0 && (long long)seg * kSegBytes >= clean_len) return;
  LaneRec *recs = reinterpret_cast<LaneRec *>(d.scratch + lay.lanes) + (size_t)seg * kSegLanes;
  const SegRec *segs = reinterpret_cast<const SegRec *>(d.scratch + lay.segs);
  int p[3] = {0, 0, 0};
  for (int j = tid; j < seg; j += kWriteThreads)
    for (int c = 0; c < 3; c++) p[c] += segs[j].dc_total[c];
  for (int c = 0; c < 3; c++) {
    int seg_base, unused;
    WorkgroupExclusiveScan<kWriteThreads / 64>(p[c], wave_sums, seg_base);
    const int mine = tid < kWriteLanes ? recs[tid >> 1].dc[tid & 1][c] : 0;
    const int base = seg_base + WorkgroupExclusiveScan<kWriteThreads / 64>(mine, wave_sums, unused);
    if (tid < kWriteLanes) recs[tid >> 1].base[tid & 1][c] = base;
  }
}

// Builds every needed 8x8 block from its records and stores it as one full 128-byte line: 8 lanes per block
// (coalesced record loads, 16 bytes of the line each), 32 blocks at a time, 256 blocks per workgroup (the per-image
// constants are fetched once per workgroup).
constexpr int kExpandThreads = 256;
constexpr int kExpandBlocks = kExpandThreads / 8;  // blocks in flight
constexpr int kExpandPerWg = 256;                  // blocks per workgroup
struct ExpandGeom {
  int32_t bpm, mcus_x, last_ordinal, use_rect, decoded_blocks, total_blocks;
  uint32_t total_records;
  uint8_t comp[16], hs[16], vs[16], ho[16], vo[16], zz[64];
  int32_t sx[12], sy[12], rect[12][4];
  GlobalCoef *base[12];
  GlobalBytes *plane[12];
  int32_t pitch[12];
  int32_t fused;
};
__global__ __launch_bounds__(kExpandThreads) void ExpandKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n,
                                                               int nwg) {
  __shared__ __attribute__((aligned(16))) int16_t stage[kExpandBlocks][72];  // 64 + padding
  __shared__ __attribute__((aligned(16))) int32_t trans[kExpandBlocks][72];  // fused output: IDCT transpose buffer
  __shared__ __attribute__((aligned(16))) uint16_t quant[3][64];
  __shared__ ExpandGeom G;
  const int wg = XcdRemap(blockIdx.x, nwg);
  if (wg < 0) return;
  // workgroup -> image (descriptors sorted by blk_wg_start)
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk_wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  const daliamdJpegHuffDesc &d = descs[lo];
  const ScratchLayout lay = MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks);
  const HuffTables *T = reinterpret_cast<const HuffTables *>(d.scratch + lay.tables);
  const int tid = threadIdx.x;
  if (tid < 12) {
    G.comp[tid] = T->blk_comp[tid]; G.hs[tid] = T->blk_hs[tid]; G.vs[tid] = T->blk_vs[tid];
    G.ho[tid] = T->blk_ho[tid]; G.vo[tid] = T->blk_vo[tid];
    G.sx[tid] = T->blk_sx[tid]; G.sy[tid] = T->blk_sy[tid];
    G.base[tid] = T->blk_base[tid];
    for (int j = 0; j < 4; j++) G.rect[tid][j] = T->blk_rect[tid][j];
    G.plane[tid] = T->blk_plane[tid];
    G.pitch[tid] = T->blk_pitch[tid];
  }
  if (tid >= 192 && tid < 192 + 48) {  // 3 x 64 quantisation values, four per thread
    const int c = (tid - 192) >> 4, j = ((tid - 192) & 15) * 4;
    for (int q = 0; q < 4; q++) quant[c][j + q] = d.quant[c][j + q];
  }
  if (tid >= 64 && tid < 128) G.zz[tid - 64] = T->zz[tid - 64];
  if (tid == 128) {
    G.bpm = T->bpm; G.mcus_x = T->mcus_x; G.last_ordinal = T->last_ordinal; G.use_rect = T->use_rect;
    G.decoded_blocks = reinterpret_cast<const int32_t *>(d.scratch)[2];  // < total_blocks: corrupt stream (status 2)
    G.total_blocks = d.total_blocks;
    G.total_records = (uint32_t)reinterpret_cast<const int32_t *>(d.scratch)[1];
    G.fused = d.plane[d.comp_of_block[0]] != nullptr;
  }
  __syncthreads();
  // {first_record, lane} pairs as dwords; everything below is read through the global address space so that the
  // loads may stay in flight across the LDS traffic and the barriers
  const GlobalWords *index = (const GlobalWords *)(d.scratch + lay.blocks);
  const GlobalWords *rec = (const GlobalWords *)(d.scratch + lay.records);
  const GlobalWords *lane_words = (const GlobalWords *)(d.scratch + lay.lanes);
  constexpr int kLaneWords = (int)(sizeof(LaneRec) / 4), kBaseWord = (int)(offsetof(LaneRec, base) / 4);
  constexpr int kIters = kExpandPerWg / kExpandBlocks;
  const int lb = tid >> 3, part = tid & 7;
  uint4 *blk_v = reinterpret_cast<uint4 *>(&stage[lb][0]);
  const int first_ordinal = (wg - d.blk_wg_start) * kExpandPerWg;

  // ---- geometry and index entries of all the blocks this thread touches: issued up front (one round trip) ----
  uint32_t first_rec[kIters], end_rec[kIters], src_lane[kIters];
  int kk[kIters], mxs[kIters], mys[kIters];
  uint32_t need_mask = 0;
#pragma unroll
  for (int it = 0; it < kIters; it++) {
    const int ordinal = first_ordinal + it * kExpandBlocks + lb;
    bool needed = ordinal < G.last_ordinal && ordinal < G.decoded_blocks;
    int k = 0, mx = 0, my = 0;
    if (needed) {
      const int mcu = ordinal / G.bpm;
      k = ordinal - mcu * G.bpm;
      my = mcu / G.mcus_x;
      mx = mcu - my * G.mcus_x;
      if (G.use_rect) {
        const int bx = mx * G.hs[k] + G.ho[k], by = my * G.vs[k] + G.vo[k];
        needed = bx >= G.rect[k][0] && by >= G.rect[k][1] && bx < G.rect[k][2] && by < G.rect[k][3];
      }
    }
    kk[it] = k; mxs[it] = mx; mys[it] = my;
    first_rec[it] = end_rec[it] = src_lane[it] = 0;
    if (needed) {
      need_mask |= 1u << it;
      first_rec[it] = index[2 * (size_t)ordinal];
      src_lane[it] = index[2 * (size_t)ordinal + 1];
      // the block's records end where the next block's begin (the write pass registers the block behind the last
      // needed one as well); the last block of the image ends with the stream
      end_rec[it] = ordinal + 1 < G.decoded_blocks && ordinal + 1 < G.total_blocks ? index[2 * (size_t)ordinal + 2]
                                                                                   : G.total_records;
    }
  }
  // records of one block: three per lane cover 24, a typical block; longer ones finish in a loop
  uint32_t w_next[3] = {0, 0, 0}, count_next = 0;
  int dc_base_next = 0;
  auto fetch = [&](int it) {
    w_next[0] = w_next[1] = w_next[2] = 0;
    count_next = 0;
    dc_base_next = 0;
    if ((need_mask >> it) & 1) {
      const uint32_t end = min(end_rec[it], G.total_records);
      count_next = first_rec[it] < end ? min(end - first_rec[it], 72u) : 0u;  // 1 DC + 63 AC + 3 ZRL + EOB
#pragma unroll
      for (int j = 0; j < 3; j++)
        if ((uint32_t)(part + 8 * j) < count_next) w_next[j] = rec[first_rec[it] + part + 8 * j];
      // DC: lane-local sum + level at the start of the lane that decoded it
      if (part == 0)  // base[half][component] of the slice's record
        dc_base_next = (int)lane_words[(size_t)(src_lane[it] >> 1) * kLaneWords + kBaseWord + (src_lane[it] & 1) * 3 +
                                       G.comp[kk[it]]];
    }
  };
  fetch(0);
#pragma unroll
  for (int it = 0; it < kIters; it++) {
    if (first_ordinal + it * kExpandBlocks >= G.last_ordinal) break;  // uniform
    const bool needed = (need_mask >> it) & 1;
    const uint32_t w0 = w_next[0], w1 = w_next[1], w2 = w_next[2], count = count_next;
    const int dc_base = dc_base_next;
    if (it + 1 < kIters) fetch(it + 1);  // in flight while this block is assembled
    blk_v[part] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (needed) {
      const uint32_t ws[3] = {w0, w1, w2};
      bool open = true;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const uint32_t i = (uint32_t)(part + 8 * j), w = ws[j];
        if (i >= count) open = false;
        if (open && i > 0 && (w & kRecDc)) open = false;  // defensive: never run into the next block
        if (open && (w & kRecValid)) {
          int v = (int)(int16_t)(w & 0xFFFF);
          if (i == 0) v += dc_base;
          stage[lb][G.zz[(w >> 16) & 63]] = (int16_t)v;
        }
      }
      for (uint32_t i = (uint32_t)part + 24; open && i < count; i += 8) {
        const uint32_t w = rec[first_rec[it] + i];
        if (w & kRecDc) break;
        if (w & kRecValid) stage[lb][G.zz[(w >> 16) & 63]] = (int16_t)(w & 0xFFFF);
      }
    }
    __syncthreads();
    if (!G.fused) {
      if (needed) {
        const int k = kk[it];
        uint4 *dst = reinterpret_cast<uint4 *>((int16_t *)(G.base[k] + ((size_t)mys[it] * (size_t)G.sy[k] + (size_t)(mxs[it] * G.sx[k]))));
        dst[part] = blk_v[part];
      }
      continue;
    }
    // ---- fused output: dequantise + inverse DCT (JpegIdctKernel's two passes on the block sitting in LDS: lane
    // `part` owns column `part` in pass 1 and row `part` in pass 2) and store the 8x8 samples to the plane ----
    if (needed) {
      const int comp = G.comp[kk[it]];
      const uint4 raw = blk_v[part];
      const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
      int32_t in[8], o[8];
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++) {
        const int16_t cv = (int16_t)(rw[r8 >> 1] >> (16 * (r8 & 1)));
        in[r8] = (int32_t)cv * (int32_t)quant[comp][part * 8 + r8];
      }
      Butterfly8(in, o);
      int32_t *w = &trans[lb][part];
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++) w[r8 * 8] = Descale(o[r8], CONST_BITS - PASS1_BITS);
    }
    __syncthreads();
    if (needed) {
      const int k = kk[it];
      const int4 *rp = reinterpret_cast<const int4 *>(&trans[lb][part * 8]);
      const int4 a = rp[0], b = rp[1];
      int32_t in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      int32_t o[8];
      Butterfly8(in, o);
      const int S = CONST_BITS + PASS1_BITS + 3;
      const uint32_t lo = RangeLimit(Descale(o[0], S)) | (RangeLimit(Descale(o[1], S)) << 8) |
                          (RangeLimit(Descale(o[2], S)) << 16) | (RangeLimit(Descale(o[3], S)) << 24);
      const uint32_t hi = RangeLimit(Descale(o[4], S)) | (RangeLimit(Descale(o[5], S)) << 8) |
                          (RangeLimit(Descale(o[6], S)) << 16) | (RangeLimit(Descale(o[7], S)) << 24);
      const int bx = mxs[it] * G.hs[k] + G.ho[k], by = mys[it] * G.vs[k] + G.vo[k];
      typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
      using GlobalPair = u32x2_t __attribute__((address_space(1)));
      GlobalPair *dst = (GlobalPair *)(G.plane[k] + (size_t)(by * 8 + part) * G.pitch[k] + (size_t)bx * 8);
      *dst = u32x2_t{lo, hi};
    }
  }
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdJpegHuffmanScratchBytes(int ecs_len, int total_blocks, size_t *bytes) {
  DALIAMD_REQUIRE(ecs_len >= 0 && total_blocks >= 0 && bytes, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegHuffmanScratchBytes: invalid argument");
  *bytes = daliamd::MakeLayout(ecs_len, daliamd::NumTiles(15, ecs_len), daliamd::NumSegments(ecs_len), total_blocks).total;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegHuffmanSetup(daliamdJpegHuffDesc *descs_host, int n, int *num_tiles, int *num_segments,
                                        int *num_block_workgroups) {
  DALIAMD_REQUIRE(n >= 0 && (n == 0 || descs_host) && num_tiles && num_segments && num_block_workgroups,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanSetup: invalid argument");
  int tiles = 0, segs = 0, bwgs = 0;
  for (int i = 0; i < n; i++) {
    daliamdJpegHuffDesc &d = descs_host[i];
    DALIAMD_REQUIRE(d.ecs && d.scratch && d.status && d.ecs_len >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegHuffmanSetup: sample %d: NULL buffer or negative length", i);
    DALIAMD_REQUIRE((reinterpret_cast<uintptr_t>(d.scratch) & 15) == 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegHuffmanSetup: sample %d: scratch must be 16-byte aligned", i);
    DALIAMD_REQUIRE(d.blocks_per_mcu >= 1 && d.blocks_per_mcu <= DALIAMD_JPEG_MAX_BLOCKS_PER_MCU && d.mcus_x >= 1 &&
                        d.total_blocks >= 1 && d.total_blocks % d.blocks_per_mcu == 0,
                    DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanSetup: sample %d: bad MCU geometry", i);
    const bool fused = d.comp_of_block[0] < 3 && d.plane[d.comp_of_block[0]] != nullptr;
    for (int k = 0; k < d.blocks_per_mcu; k++) {
      const int comp = d.comp_of_block[k];
      DALIAMD_REQUIRE(comp < 3 && (fused ? d.plane[comp] != nullptr : d.coef[comp] != nullptr),
                      DALIAMD_ERROR_INVALID_ARGUMENT,
                      "daliamdJpegHuffmanSetup: sample %d: block %d refers to a missing component", i, k);
      if (fused) {
        DALIAMD_REQUIRE((reinterpret_cast<uintptr_t>(d.plane[comp]) & 7) == 0 && (d.plane_pitch[comp] & 7) == 0 &&
                            d.plane_pitch[comp] >= d.blocks_x[comp] * 8,
                        DALIAMD_ERROR_INVALID_ARGUMENT,
                        "daliamdJpegHuffmanSetup: sample %d: planes must be 8-byte aligned with a pitch that is a "
                        "multiple of 8 and covers blocks_x * 8 samples", i);
      } else {
        DALIAMD_REQUIRE((reinterpret_cast<uintptr_t>(d.coef[comp]) & 15) == 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                        "daliamdJpegHuffmanSetup: sample %d: coefficient arrays must be 16-byte aligned", i);
      }
    }
    for (int t = 0; t < 4; t++)  // record capacity: every symbol consumes at least two bits
      DALIAMD_REQUIRE(d.bits[t][0] == 0, DALIAMD_ERROR_UNSUPPORTED,
                      "daliamdJpegHuffmanSetup: sample %d: Huffman table %d has a 1-bit code (decode it on the host)", i, t);
    d.tile_start = tiles;
    d.num_tiles = daliamd::NumTiles((int)(reinterpret_cast<uintptr_t>(d.ecs) & 15), d.ecs_len);
    d.seg_start = segs;
    d.num_segments = daliamd::NumSegments(d.ecs_len);
    d.blk_wg_start = bwgs;
    tiles += d.num_tiles;
    segs += d.num_segments;
    bwgs += (d.total_blocks + daliamd::kExpandPerWg - 1) / daliamd::kExpandPerWg;
  }
  *num_tiles = tiles;
  *num_segments = segs;
  *num_block_workgroups = bwgs;
  return DALIAMD_SUCCESS;
}

static daliamdResult_t LaunchHuffman(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                     int num_segments, int num_block_workgroups, daliamdEvent_t *events) {
  if (n == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_tiles >= n && num_segments >= n && num_block_workgroups >= n,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanRun: invalid argument");
  using namespace daliamd;
  hipStream_t s = (hipStream_t)stream;
  const int seg_grid = XcdGrid(num_segments);
  int e = 0;
  auto mark = [&]() -> hipError_t { return events ? hipEventRecord((hipEvent_t)events[e++], s) : hipSuccess; };
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(PrepareKernel, dim3(num_tiles + n), dim3(kTileThreads), 0, s, descs_dev, n, num_tiles);
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(UnstuffScatterKernel, dim3(num_tiles), dim3(kTileThreads), 0, s, descs_dev, n);
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(SyncKernel, dim3(seg_grid), dim3(kSegThreads), 0, s, descs_dev, n, num_segments);
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(PropagateKernel, dim3(n), dim3(kSegThreads), 0, s, descs_dev);
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(WriteKernel, dim3(seg_grid), dim3(kWriteThreads), 0, s, descs_dev, n, num_segments);
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(DcScanKernel, dim3(seg_grid), dim3(kWriteThreads), 0, s, descs_dev, n, num_segments);
  DALIAMD_HIP_CHECK(mark());
  hipLaunchKernelGGL(ExpandKernel, dim3(XcdGrid(num_block_workgroups)), dim3(kExpandThreads), 0, s, descs_dev, n,
                     num_block_workgroups);
  DALIAMD_HIP_CHECK(mark());
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegHuffmanRun(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                      int num_segments, int num_block_workgroups) {
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, nullptr);
}

daliamdResult_t daliamdJpegHuffmanRunProfiled(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                              int num_tiles, int num_segments, int num_block_workgroups,
                                              daliamdEvent_t *events) {
  DALIAMD_REQUIRE(events, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanRunProfiled: events is NULL");
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, events);
}

}  // extern "C"
