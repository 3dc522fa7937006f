// GPU Huffman entropy decoder for baseline JPEG (one interleaved scan, with or without restart intervals) on gfx950.
//
// Reference counterpart: the GPU Huffman stage of nvJPEG inside nvImageCodec, reached from
// ImageDecoder::RunImplImpl (dali/operators/imgcodec/image_decoder.h:810-815).  The output is what the host decoder
// (dali_amd/host/jpeg_entropy.cpp) + the IDCT kernel produce, bit for bit.
//
// The entropy-coded segment of an image is ONE serial bit stream; the decoder state is (bit position, block index
// inside the MCU, zig-zag index).  Parallelism comes from self-synchronisation: a decoder started from a guessed
// state at an arbitrary byte falls into step with the true decoder after a while.  Work is cut into pieces whose
// size does not depend on the image (so a batch with one 500 KB stream and many 50 KB streams still fills the chip):
//
//   tile     8 KB of the stuffed stream     (un-stuffing, 512 lanes x 16 bytes)
//   slice    256 bytes of the clean stream  (one decoder lane of the synchronisation)
//   segment  244 slices = 61 KB             (one 256-lane workgroup: 12 warm-up lanes + 244 slices)
//   block    one 8x8 block                  (one lane of the value pass)
//
//   1 PrepareKernel         per tile: number of bytes that survive the removal of the 0xFF00 stuffing and of the RSTn
//                           markers, and WHERE THE SEGMENT ENDS (the first marker that is not RSTn: the host only
//                           parses the headers and hands over "everything behind SOS"); and, in extra
//                           workgroups of the same launch (they need nothing but the descriptor), per image: two-level
//                           code tables (11-bit first level, direct second level for the long codes) and the
//                           symbol-group tables of the position-only passes -> global scratch
//   2 UnstuffScatterKernel  per tile: compaction through LDS to its final place in the clean stream
//   3 SyncKernel            per segment: every lane decodes its slice from a guessed state - positions only, up to three
//                           symbols per table look-up -, then the relaxation "publish the state you reached to the next
//                           lane, decode again if your input changed" runs until nothing changes.  The 12 warm-up lanes
//                           replay the end of the previous segment so that the first slice of the segment starts from
//                           the true state with overwhelming probability.  Every decode notes WHERE EACH BLOCK STARTS
//                           (a short list per lane in LDS, one store per step, no branch); the converged lists are
//                           written out densely per segment
//   4 PropagateKernel       per image, serial over its segments: checks that every segment started from the state
//                           its predecessor ended in (if not - pathological streams - repairs it with the same
//                           relaxation, so correctness never depends on luck), assigns block ordinals
//   5 DcKernel              per segment, one lane per block: decodes the block's DC difference (one look-up), prefix
//                           sums per component inside the segment; notes the bit position behind the DC symbol
//   6 BlockKernel           one lane per block: AC symbols -> coefficients of the lane's block in LDS (the loop knows
//                           nothing but one code table: no DC / block bookkeeping, ~26 instructions per symbol); then
//                           every lane dequantises + inverse-transforms ITS block in its registers and stores the 8x8
//                           samples to the component plane (or the coefficients, for callers that want them)
//
// Round 1's decoder ran a second full sequential decode per half slice that appended one 32-bit record per symbol to
// a stream in HBM (WriteKernel, 80 instructions per symbol), a DC scan and an expand kernel that gathered the records
// of each block again: 2 x 137 MB of scratch traffic per 256-image batch and 0.36 ms.  Knowing the block starts makes
// the value pass embarrassingly parallel, its inner loop three times shorter, and the record stream disappears.
#include <cstring>
#include <vector>
#include "common.h"
#include "huff_core.h"
#include "jpeg_idct_math.h"
#include "jpeg_color_math.h"

namespace daliamd {

#ifndef DALIAMD_TILE_THREADS
#define DALIAMD_TILE_THREADS 512   // 8 KB tiles (MI355X, 4 batches in flight: 1024 threads 358k img/s, 512: 378k, 256: 351k - a 16-wave workgroup waits for a whole free CU)
#endif
constexpr int kTileThreads = DALIAMD_TILE_THREADS;
constexpr int kTileBytes = kTileThreads * 16;
#ifndef DALIAMD_SLICE_BYTES
#define DALIAMD_SLICE_BYTES 256
#endif
constexpr int kSliceBytes = DALIAMD_SLICE_BYTES;
#ifndef DALIAMD_SEG_THREADS
#define DALIAMD_SEG_THREADS 256
#endif
constexpr int kSegThreads = DALIAMD_SEG_THREADS;
constexpr int kWarmLanes = 12;
constexpr int kSegLanes = kSegThreads - kWarmLanes;
constexpr int kSegBytes = kSegLanes * kSliceBytes;
constexpr int kCleanPadBytes = 40;
// Block-start lists of the synchronisation pass: kListCap entries per lane in LDS + one slot that swallows the
// overflow; the stride (in 16-bit entries) is an odd number of dwords so that lanes at the same index hit different banks.
// A slice with more starts is decoded once more after the relaxation, writing its starts directly.  Measured on the
// bench set (sync pass, us): cap 65: 371, 33: 390, 17: 311, 9: 326, 5: 332, 1: 322 - the store inside the relaxation
// loop costs more than one extra decode of the slices that overflow, and a 50 KB workgroup leaves LDS for the other
// stream's kernels.
#ifndef DALIAMD_LIST_CAP
#define DALIAMD_LIST_CAP 17
#endif
constexpr int kListCap = DALIAMD_LIST_CAP;       // odd, so that the stride below is an odd number of dwords
constexpr int kListStride = kListCap + 1;
// A block takes at least 2 bits (a 1-bit DC code + a 1-bit end-of-block), a group may overshoot the slice by 3 symbols.
constexpr int kMaxStartsPerSlice = kSliceBytes * 8 / 2 + 32;

// Explicit global address space: a generic pointer would make these `flat` accesses, which count against the LDS
// counter as well and would serialise the table look-ups behind the stream prefetch.
using GlobalWords = const uint32_t __attribute__((address_space(1)));
using GlobalCoef = int16_t __attribute__((address_space(1)));
using GlobalBytes = uint8_t __attribute__((address_space(1)));
using GlobalU32 = uint32_t __attribute__((address_space(1)));
using GlobalI32 = int32_t __attribute__((address_space(1)));
using GlobalU16 = uint16_t __attribute__((address_space(1)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
using GlobalQuad = u32x4 __attribute__((address_space(1)));

// ------------------------------------------------------------------------------------------------ scratch layout
struct LaneRec {      // one per slice
  uint64_t in, out;   // packed decoder state at the start / end of the slice
  int32_t nstart;     // blocks that start in the slice (= entries it contributed to the segment's start list)
  int32_t reserved;
};
static_assert(sizeof(LaneRec) == 24, "layout");
struct SegRec {  // one per segment
  uint64_t out;  // state at the end of the segment
  int32_t nstart_total;  // block starts found in the segment
  int32_t block_base;    // ... in the segments before it = ordinal of the segment's first start (PropagateKernel)
  int32_t dc_total[3];   // DcKernel: sum of the DC differences of the segment's blocks, per component
  int32_t crossed;       // restart intervals: a lane's final decode ran over a boundary (Lane::crossed)
};
static_assert(sizeof(SegRec) == 32, "layout");
constexpr int kSegRecInts = 8, kSegRecNstart = 2, kSegRecBlockBase = 3, kSegRecDcTotal = 4;  // int32 view of a SegRec

struct TileRec {   // one per tile of the stuffed stream (PrepareKernel)
  int32_t kept;    // bytes of the tile that go into the clean stream (up to the end of the segment, if it ends here)
  int32_t ended;   // the segment ends in this tile: a marker other than RSTn (or a fill byte's 0xFF) was found
  int32_t nrst;    // RSTn markers in front of that
  int32_t reserved;
};
static_assert(sizeof(TileRec) == 16, "layout");
struct ScratchLayout {
  size_t tile_recs, clean, tables, sync_tables, lanes, segs, seg_starts, blk_pos, blk_dc, blk_seg, seams, rst_pos, total;
  int seg_cap;  // entries per segment in seg_starts
  int rst_cap;  // entries in rst_pos
};
__host__ __device__ inline size_t AlignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }
// num_intervals: restart intervals of the frame (0: the stream has none) - their array comes LAST, so every other offset
// is the same whatever it is.
__host__ __device__ inline ScratchLayout MakeLayout(int ecs_len, int num_tiles, int num_segments, int total_blocks,
                                                    int num_intervals) {
  ScratchLayout l;
  size_t o = 16;  // int32 [0] clean_len, [1] restart boundaries found, [2] block starts found in the whole stream
  l.tile_recs = o;
  o += sizeof(TileRec) * (size_t)num_tiles;
  l.clean = o;
  o += AlignUp((size_t)ecs_len + 256, 16);
  l.tables = o;
  o += sizeof(HuffTables);
  l.sync_tables = o;
  o += sizeof(SyncTables);
  l.lanes = o;
  o += AlignUp(sizeof(LaneRec) * (size_t)num_segments * kSegLanes, 16);
  l.segs = o;
  o += AlignUp(sizeof(SegRec) * (size_t)num_segments, 16);
  // block starts per segment, densely: a stream holds total_blocks real blocks (+ a few phantom ones parsed from the
  // padding behind it), so no segment can own more than that - nor more than its slices can hold
  const long long by_slices = (long long)kSegLanes * kMaxStartsPerSlice, by_blocks = (long long)total_blocks + 128;
  l.seg_cap = (int)(by_slices < by_blocks ? by_slices : by_blocks);
  l.seg_starts = o;
  o += AlignUp(sizeof(uint32_t) * (size_t)num_segments * (size_t)l.seg_cap, 16);
  l.blk_pos = o;   // per block: bit position behind its DC symbol
  o += AlignUp(sizeof(uint32_t) * (size_t)total_blocks, 16);
  l.blk_dc = o;    // per block: its DC level relative to the start of its segment
  o += AlignUp(sizeof(int32_t) * (size_t)total_blocks, 16);
  l.blk_seg = o;   // per block: the segment it starts in
  o += AlignUp(sizeof(uint16_t) * (size_t)total_blocks, 16);
  // fused colour output: per band of MCU rows two strips (its first and its last pixel row: 16 luma + 2 x 8 chroma
  // bytes per MCU column) for the seam launch; bands <= MCU rows, so 64 bytes per MCU = 10.7 per block of a 4:2:0 frame
  l.seams = o;
  o += AlignUp((size_t)total_blocks * 11 + 64, 16);
  l.rst_pos = o;   // per restart boundary: clean byte offset at which the next interval starts
  l.rst_cap = num_intervals;
  o += AlignUp(sizeof(uint32_t) * (size_t)num_intervals, 16);
  l.total = AlignUp(o, 256);
  return l;
}
__host__ __device__ inline int NumIntervals(int total_blocks, int blocks_per_mcu, int restart_interval) {
  return restart_interval > 0 ? (total_blocks / blocks_per_mcu + restart_interval - 1) / restart_interval : 0;
}
__host__ __device__ inline ScratchLayout LayoutOf(const daliamdJpegHuffDesc &d) {
  return MakeLayout(d.ecs_len, d.num_tiles, d.num_segments, d.total_blocks,
                    NumIntervals(d.total_blocks, d.blocks_per_mcu, d.restart_interval));
}
inline int NumTiles(int head, int len) {
  int t = (head + len + kTileBytes - 1) / kTileBytes;
  return t > 0 ? t : 1;
}
inline int NumSegments(int len) { return len > kSegBytes ? (len + kSegBytes - 1) / kSegBytes : 1; }

// ------------------------------------------------------------------------------------------------ resident index
// Index entry of a resident stream (daliamdJpegHuffDesc.index / index_out; huff_core.h: SliceIndex): a 64-byte header,
// the CLEAN stream (un-stuffed, markers removed, all-ones padding behind it) and one SliceIndex per slice + a sentinel.
// Sized by the upper bound ecs_len (the clean stream is never longer than the stuffed one).
struct IndexHeader {
  int32_t clean_len;      // bytes of the clean stream
  int32_t total_starts;   // block starts the stream holds (blocks + 1 for a complete stream)
  int32_t num_slices;     // entries that describe data
  int32_t reserved[13];
};
static_assert(sizeof(IndexHeader) == 64, "layout");
__host__ __device__ inline int IndexSliceCap(int ecs_len) { return (ecs_len + kSliceBytes - 1) / kSliceBytes; }
__host__ __device__ inline size_t IndexCleanOffset() { return sizeof(IndexHeader); }
__host__ __device__ inline size_t IndexEntriesOffset(int ecs_len) { return sizeof(IndexHeader) + AlignUp((size_t)ecs_len + 256, 64); }
__host__ __device__ inline size_t IndexBytes(int ecs_len) {
  return AlignUp(IndexEntriesOffset(ecs_len) + sizeof(SliceIndex) * (size_t)(IndexSliceCap(ecs_len) + 1), 64);
}
// the clean stream a decode of `d` reads: the resident one, or the one this batch's un-stuffing wrote
__device__ __forceinline__ const uint8_t *CleanStream(const daliamdJpegHuffDesc &d, const ScratchLayout &lay) {
  return d.index ? d.index + IndexCleanOffset() : d.scratch + lay.clean;
}

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

// Code tables are built once per DISTINCT set of Huffman tables of the batch (most JPEG files carry the standard
// tables of Annex K: a batch of 256 images is usually one or two sets): `table_owner` (Setup) names the stream in whose
// scratch the tables of this stream live.  PrepareKernel builds 1-3 sets instead of 256, and every workgroup of the
// decode passes copies its 36 KB of tables from the same few L2-resident lines.
// (round 5) A stream whose caller brings finished tables (daliamdJpegHuffDesc.tables: built on the host once per distinct
// DHT contents, daliamdJpegHuffmanTablesBuild) reads those; nobody builds anything for it.
__device__ __forceinline__ const uint8_t *TablesBase(const daliamdJpegHuffDesc *descs, const daliamdJpegHuffDesc &d, bool sync_tables) {
  if (d.tables) return d.tables + (sync_tables ? sizeof(HuffTables) : 0);
  const daliamdJpegHuffDesc &o = descs[d.table_owner];
  const ScratchLayout lay = LayoutOf(o);
  return o.scratch + (sync_tables ? lay.sync_tables : lay.tables);
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
// 16 bytes of the stuffed stream per lane -> mask of the bytes that stay (bit j = byte j).  What goes: the zero behind
// a 0xFF (byte stuffing), both bytes of an RSTn marker, fill bytes (a 0xFF in front of a 0xFF), and everything from the
// first other marker on - that is where the entropy-coded segment ENDS, wherever the caller said it does (ecs_len may
// be "the rest of the file": the host does not walk the scan, jpeg_entropy.cpp: daliamdJpegAnalyzeHeader).
struct TileChunk {
  uint32_t w[4];
  uint32_t keep;
  uint32_t rst;   // bit j: byte j is the 0xFF of an RSTn marker
  int end_j;      // first byte of this lane that starts an ending marker (16: none)
};
__device__ __forceinline__ TileChunk LoadChunk(const daliamdJpegHuffDesc &d, int tile) {
  const int head = (int)(reinterpret_cast<uintptr_t>(d.ecs) & 15);  // bytes between the 16-byte boundary and the segment
  const int end = head + d.ecs_len;
  const int g = tile * kTileBytes + (int)threadIdx.x * 16;
  TileChunk c{{0, 0, 0, 0}, 0, 0, 16};
  uint32_t prev = 0, behind = 0;
  const int lane = threadIdx.x & 63;
  if (g < end) {
    GlobalWords *p = (GlobalWords *)__builtin_assume_aligned((const void *)(d.ecs - head + g), 16);
    c.w[0] = p[0]; c.w[1] = p[1]; c.w[2] = p[2]; c.w[3] = p[3];
    // the bytes next to the chunk: the neighbouring lanes hold them (below) - only the ends of a wave load theirs
    if (lane == 0 && g > head) prev = ((const GlobalBytes *)(d.ecs - head))[g - 1];
    if (lane == 63 && g + 16 < end) behind = ((const GlobalBytes *)(d.ecs - head))[g + 16];
  }
  {
    const uint32_t from_below = __shfl_up(c.w[3] >> 24, 1, 64), from_above = __shfl_down(c.w[0] & 255u, 1, 64);
    if (lane != 0) prev = from_below;              // (a lane past the end of the segment holds zeros: "no byte behind")
    if (lane != 63) behind = g + 16 < end ? from_above : 0u;
  }
  // Sixteen bytes at once (a byte loop costs 20 instructions per byte): per dword the bytes that are 0xFF, 0x00 or RSTn's
  // second byte as 0x80 flags (exact zero-byte test, no borrow between bytes), gathered into 16-bit masks, bit j = byte j.
  auto zero_bytes = [](uint32_t x) { return ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) | 0x7F7F7F7Fu); };
  auto gather = [](uint32_t y) { return ((y >> 7) * 0x01020408u) >> 24; };   // flags of bytes 0..3 -> bits 0..3
  uint32_t F = 0, Z = 0, D = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    F |= gather(zero_bytes(~c.w[i])) << (4 * i);
    Z |= gather(zero_bytes(c.w[i])) << (4 * i);
    D |= gather(zero_bytes((c.w[i] & 0xF8F8F8F8u) ^ 0xD0D0D0D0u)) << (4 * i);
  }
  auto below = [](int k) { return k >= 16 ? 0xFFFFu : (k <= 0 ? 0u : (1u << k) - 1u); };   // bits [0, k)
  const uint32_t valid = below(end - g) & ~below(head - g);      // bytes of the segment
  const uint32_t next_valid = below(end - g - 1);                 // ... whose successor is one, too
  F &= valid;
  // what the byte BEHIND byte j is (byte 15: the neighbour's first; zero past the end: a last 0xFF stays what it would be
  // in front of a stuffed zero)
  const uint32_t nZ = (Z >> 1) | ((behind == 0u) << 15), nD = (D >> 1) | (((behind & 0xF8u) == 0xD0u) << 15),
                 nF = (F >> 1) | ((behind == 0xFFu) << 15);
  const uint32_t marker = F & next_valid & ~nZ;                   // RSTn, fill byte or the end
  const uint32_t is_rst = marker & nD, is_end = marker & ~nD & ~nF;
  // a byte behind a 0xFF: the stuffed zero, or the second byte of RSTn (the first byte of the segment has no predecessor)
  const uint32_t after_ff = ((F << 1) | (prev == 0xFFu && g > head ? 1u : 0u)) & 0xFFFFu;
  c.keep = valid & ~(marker | (after_ff & (Z | D)));
  c.rst = is_rst;
  c.end_j = is_end ? __ffs(is_end) - 1 : 16;
  return c;
}

// Exclusive scan over the workgroup of the bytes each lane keeps (`c.keep` as loaded, not yet cut) together with the
// search for the end of the segment (the lanes' end_j) - ONE pair of barriers for both: the prefix of a lane in front
// of the end does not depend on what the lanes behind it hold.  Returns the lane's exclusive prefix (meaningless behind
// the end); `tile_end` = index of the first ending byte in the tile (kTileBytes: none), `total` = bytes the tile keeps in
// front of it.  Cuts c.keep / c.rst at the end.  `wave_sums`: kTileThreads / 64 + 1 ints, `wave_ends`: kTileThreads / 64.
__device__ __forceinline__ int ScanKeepAndEnd(TileChunk &c, int *wave_sums, int *wave_ends, int &tile_end, int &total) {
  constexpr int NW = kTileThreads / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int v = __popc(c.keep);
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  // first lane of the wave that holds an end (ballot: nearly always zero, no shuffles then)
  const unsigned long long ends = __ballot(c.end_j < 16);
  int wave_end = kTileBytes;
  if (ends) {
    const int first = __ffsll((long long)ends) - 1;
    wave_end = (wave * 64 + first) * 16 + __shfl(c.end_j, first, 64);
  }
  if (lane == 63) wave_sums[wave] = incl;
  if (lane == 0) wave_ends[wave] = wave_end;
  __syncthreads();
  int base = 0, tot = 0, te = kTileBytes;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    const int sum = wave_sums[w];
    if (w < wave) base += sum;
    tot += sum;
    te = min(te, wave_ends[w]);
  }
  const int excl = base + incl - v;
  tile_end = te;
  // cut at the end; the lane that holds it knows how many bytes the tile keeps in front of it
  const int mine = te - (int)threadIdx.x * 16;
  const uint32_t m = mine >= 16 ? 0xFFFFu : (mine <= 0 ? 0u : (1u << mine) - 1u);
  c.keep &= m;
  c.rst &= m;
  if (te < kTileBytes && mine >= 0 && mine < 16) wave_sums[NW] = excl + __popc(c.keep);
  __syncthreads();  // (wave_sums may be reused by the caller's next scan)
  total = te < kTileBytes ? wave_sums[NW] : tot;
  return excl;
}

// First pass of the un-stuffing: what each tile keeps (PrepareKernel).
__device__ __forceinline__ void CountTile(const daliamdJpegHuffDesc *__restrict__ descs, int n, int tile, int *wave_sums, int *wave_ends) {
  const ImageRef r = FindImage<true>(descs, n, tile);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = LayoutOf(d);
  TileChunk c = LoadChunk(d, r.local);
  int tile_end, total, nrst = 0;
  ScanKeepAndEnd(c, wave_sums, wave_ends, tile_end, total);
  if (!d.restart_interval && c.rst) *d.status = 3;   // RSTn markers in a stream without DRI
  if (d.restart_interval) {   // (uniform)
    __syncthreads();          // wave_sums[NW] was read above
    WorkgroupExclusiveScan<kTileThreads / 64>(__popc(c.rst), wave_sums, nrst);
  }
  if (threadIdx.x == 0)
    reinterpret_cast<TileRec *>(d.scratch + lay.tile_recs)[r.local] = TileRec{total, tile_end < kTileBytes ? 1 : 0, nrst, 0};
}

__global__ __launch_bounds__(kTileThreads) void UnstuffScatterKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n) {
  __shared__ uint32_t stage[kTileBytes / 4 + 4];
  __shared__ int wave_sums[kTileThreads / 64 + 1];
  __shared__ int wave_ends[kTileThreads / 64];
  __shared__ int wave_pre[kTileThreads / 64][3];
  constexpr int NW = kTileThreads / 64;
  const ImageRef r = FindImage<true>(descs, n, blockIdx.x);
  const daliamdJpegHuffDesc &d = *r.d;
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // clean-stream position of this tile = bytes kept by the tiles before it, up to (and including) the tile the segment
  // ends in; a tile behind that one is dead.  One barrier pair: per wave the sums over its lanes up to its first tile
  // with the end flag, and whether it holds one.
  const TileRec *recs = reinterpret_cast<const TileRec *>(d.scratch + lay.tile_recs);
  int base = 0, rst_base = 0;
  bool dead = false;
  for (int t0 = 0; t0 < r.local && !dead; t0 += kTileThreads) {   // (one round unless the stream has > 4 MB in front of the tile)
    const int t = t0 + tid;
    TileRec rec{0, 0, 0, 0};
    if (t < r.local) rec = recs[t];
    const unsigned long long ended = __ballot(rec.ended != 0);
    const int first = ended ? __ffsll((long long)ended) - 1 : 64;
    int kept = lane <= first ? rec.kept : 0, nrst = lane <= first ? rec.nrst : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      kept += __shfl_xor(kept, off, 64);
      nrst += __shfl_xor(nrst, off, 64);
    }
    if (lane == 0) { wave_pre[wave][0] = kept; wave_pre[wave][1] = nrst; wave_pre[wave][2] = ended != 0; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; w++) {
      if (dead) break;
      base += wave_pre[w][0];
      rst_base += wave_pre[w][1];
      dead = wave_pre[w][2] != 0;
    }
    __syncthreads();
  }
  const int shift = base & 3;  // stage byte i <-> clean byte (base - shift) + i
  TileChunk c = LoadChunk(d, r.local);
  int tile_end, total;
  int o = shift + ScanKeepAndEnd(c, wave_sums, wave_ends, tile_end, total);
  if (dead) { c.keep = 0; c.rst = 0; total = 0; }
  if (d.restart_interval) {   // (uniform) where the intervals start in the clean stream
    __syncthreads();          // wave_sums[NW] was read above
    int nrst;
    int k = rst_base + WorkgroupExclusiveScan<kTileThreads / 64>(__popc(c.rst), wave_sums, nrst);
    GlobalU32 *rst_pos = (GlobalU32 *)(d.scratch + lay.rst_pos);
    for (uint32_t m = c.rst; m; m &= m - 1) {
      const int j = __ffs(m) - 1;
      if (k < lay.rst_cap) rst_pos[k] = (uint32_t)(base - shift + o + __popc(c.keep & ((1u << j) - 1u)));
      k++;
    }
    if (r.local == d.num_tiles - 1 && tid == 0) {
      const int found = rst_base + nrst;
      reinterpret_cast<int32_t *>(d.scratch)[1] = found < lay.rst_cap ? found : lay.rst_cap;
    }
  }
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
    if (tid == 0) {
      *reinterpret_cast<int32_t *>(d.scratch) = clean_len;
      if (!d.restart_interval) reinterpret_cast<int32_t *>(d.scratch)[1] = 0;
    }
    // all-ones padding: never a valid code, lets the bit window run past the end
    if (tid < kCleanPadBytes) ((GlobalBytes *)(d.scratch + lay.clean))[clean_len + tid] = 0xFF;
  }
}

// ------------------------------------------------------------------------------------------------ tables
// `count` 16-byte chunks from global memory to LDS.  The loads of a thread are issued TOGETHER (up to 9 in flight), then
// stored: the loop "load, wait, store" this replaces paid one memory round trip per chunk - nine in a row for the 36 KB of
// the position passes' tables, 4-5 us at the start of every one of their 500 workgroups (round 5, from the ISA listing).
template <int THREADS>
__device__ __forceinline__ void CopyChunks(uint4 *t, const GlobalQuad *s, int count) {
  constexpr int kInFlight = 9;
  for (int i0 = threadIdx.x; i0 < count; i0 += THREADS * kInFlight) {
    u32x4 v[kInFlight];
#pragma unroll
    for (int k = 0; k < kInFlight; k++)
      if (i0 + k * THREADS < count) v[k] = s[i0 + k * THREADS];
#pragma unroll
    for (int k = 0; k < kInFlight; k++)
      if (i0 + k * THREADS < count) t[i0 + k * THREADS] = make_uint4(v[k].x, v[k].y, v[k].z, v[k].w);
  }
}
template <int THREADS, typename Tables>
__device__ __forceinline__ void CopyTables(Tables &dst, const Tables *src) {
  CopyChunks<THREADS>(reinterpret_cast<uint4 *>(&dst), (const GlobalQuad *)src, (int)(sizeof(Tables) / 16));
}

// Code tables of one stream (all threads of a PrepareKernel workgroup); every entry is found independently
// (huff_core.h: FastEntry / L2Entry / SyncEntry).
__device__ __forceinline__ void BuildTables(const daliamdJpegHuffDesc &d, HuffTables &L) {
  constexpr int NT = kTileThreads;
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x;
  {
    uint4 *z = reinterpret_cast<uint4 *>(&L);
    for (int i = tid; i < (int)(sizeof(HuffTables) / 16); i += NT) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  for (int t = tid; t < 4 * 256; t += NT) L.vals[t >> 8][t & 255] = d.vals[t >> 8][t & 255];
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
  }
  if (tid >= 64 && tid < 68) {
    const int t = tid - 64;
    int32_t first, size;
    CodeRanges(d.bits[t], L.maxcode[t], L.valoff[t], &first, &size);
    L.l2_first[t] = first;
    L.l2_size[t] = size;
  }
  __syncthreads();
  for (int idx = tid; idx < 4 * (1 << kFastBits); idx += NT)
    L.fast[idx >> kFastBits][idx & ((1 << kFastBits) - 1)] = FastEntry(L, idx >> kFastBits, idx & ((1 << kFastBits) - 1));
  for (int idx = tid; idx < 4 * kL2Entries; idx += NT) L.l2[idx / kL2Entries][idx % kL2Entries] = L2Entry(L, idx / kL2Entries, idx % kL2Entries);
  __syncthreads();
  const uint4 *s = reinterpret_cast<const uint4 *>(&L);
  uint4 *t = reinterpret_cast<uint4 *>(d.scratch + lay.tables);
  for (int i = tid; i < (int)(sizeof(HuffTables) / 16); i += NT) t[i] = s[i];
  // ---- tables of the position-only passes, straight to global memory ----
  SyncTables *S = reinterpret_cast<SyncTables *>(d.scratch + lay.sync_tables);
  for (int idx = tid; idx < 4 * (1 << kFastBits); idx += NT)
    S->t32[idx >> kFastBits][idx & ((1 << kFastBits) - 1)] = SyncEntry(L, idx >> kFastBits, idx & ((1 << kFastBits) - 1));
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
// (ntab = n when some stream of the table needs its code tables built here, 0 when every stream brought them)
__global__ __launch_bounds__(kTileThreads) void PrepareKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n,
                                                              int num_tiles, int ntab) {
  __shared__ __attribute__((aligned(16))) HuffTables L;
  __shared__ int wave_sums[kTileThreads / 64 + 1];
  __shared__ int wave_ends[kTileThreads / 64];
  if ((int)blockIdx.x < ntab) {
    const daliamdJpegHuffDesc &d = descs[blockIdx.x];
    if (d.table_owner == (int)blockIdx.x && !d.tables) BuildTables(d, L);   // (uniform per workgroup)
  } else {
    CountTile(descs, n, (int)blockIdx.x - ntab, wave_sums, wave_ends);
  }
}

// ------------------------------------------------------------------------------------------------ synchronisation
// One decoder lane of the relaxation.
struct Lane {
  uint32_t begin, end;  // bit range of the slice (clipped to the stream)
  bool active;          // the slice holds data
  uint64_t in = kNoState, out = kNoState;
  int nstart = 0;       // blocks that start in the slice, by the lane's latest decode
  bool crossed = false; // that decode ran over a restart boundary it did not see coming (huff_core.h)
};

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

// "Publish the state you reached to the next lane, decode again if your input changed", until nothing changes.
// state[t] is the published input of lane t; lanes whose `in` already equals it do not decode.  Lane 0's input is
// never written here, so whatever the caller put there is taken as the truth; each round fixes at least one more
// lane, which bounds the loop by the lane count.  `list` = this lane's block-start list in LDS.
using RstView = RestartView<GlobalWords *>;
// The stream's restart boundaries as the lane at bit `begin` needs them: hint = the first one at or behind its slice.
__device__ __forceinline__ RstView MakeRstView(const daliamdJpegHuffDesc &d, const ScratchLayout &lay, uint32_t begin_bits) {
  RstView v{(GlobalWords *)(d.scratch + lay.rst_pos), 0, (uint32_t)d.restart_interval, 0};
  if (d.restart_interval) {
    v.n = ((const GlobalI32 *)d.scratch)[1];
    const uint32_t want = begin_bits >> 3;
    int lo = 0, hi = v.n;   // first k with pos[k] >= want
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (v.pos[mid] < want) lo = mid + 1; else hi = mid;
    }
    v.hint = lo;
  }
  return v;
}

#ifdef DALIAMD_EXP_STAMPS
// Development probe (tools/stamp_probe.py; never in the shipped library): wall-clock stamps (100 MHz) per workgroup of the
// two position kernels - where a launch's time goes, workgroup by workgroup.
__device__ unsigned long long g_stamps[3][8192 * 16];
__device__ __forceinline__ void Stamp(int which, int k) {
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[which][blockIdx.x * 16 + k] = wall_clock64();
}
__device__ __forceinline__ void StampMax(int which, int k) {
  if (blockIdx.x < 8192) atomicMax(&g_stamps[which][blockIdx.x * 16 + k], (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void StampIds(int which, int wg) {
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    g_stamps[which][blockIdx.x * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    g_stamps[which][blockIdx.x * 16 + 14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    g_stamps[which][blockIdx.x * 16 + 15] = (unsigned long long)wg;
  }
}
extern "C" DALIAMD_API int daliamdDebugReadStamps(int which, void *dst, size_t bytes, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, (size_t)which * sizeof(g_stamps[0])) != hipSuccess) return 2;
  if (clear) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess) return 3;
    if (hipMemset((char *)p + (size_t)which * sizeof(g_stamps[0]), 0, sizeof(g_stamps[0])) != hipSuccess) return 4;
  }
  return 0;
}
#define STAMP(which, k) Stamp(which, k)
#define STAMP_MAX(which, k) StampMax(which, k)
#define STAMP_IDS(which, wg) StampIds(which, wg)
#else
#define STAMP(which, k)
#define STAMP_MAX(which, k)
#define STAMP_IDS(which, wg)
#endif
// Round 4: the lanes that have to decode in a round are PACKED into the first waves (a work list in LDS).  Rounds 1 and 2
// keep nearly every lane busy; from round 3 on a handful of slices re-decode, and with lane = slice each of them kept its
// whole wave issuing for the length of a slice.  Host model on the bench's first batch (tools/sync_sim.cpp): 1 089 136
// wave-steps become 863 814 (0.79) - the critical path is the same, the instructions are not.  The lane states live
// in LDS, indexed by slice; `ln` receives this thread's own slice at the end.
#ifndef DALIAMD_SYNC_COMPACT
#define DALIAMD_SYNC_COMPACT 1
#endif
struct RelaxShared {
  uint64_t state[kSegThreads + 1];   // state[t]: the published input of lane t; state[t + 1] = what lane t reached
  uint64_t last_in[kSegThreads];     // the input lane t last decoded from (kNoState: never)
  uint16_t nstart[kSegThreads];      // blocks that start in slice t, by its latest decode
  uint8_t work[kSegThreads];         // this round's work list: slices whose input changed
  uint8_t crossed[kSegThreads];
  int wave_count[kSegThreads / 64];
};
static_assert(kSegThreads <= 256, "the work list holds 8-bit lane numbers");

// slice_of(t): index of lane t's slice in the stream (negative: the lane has none)
template <typename SliceOf>
__device__ __forceinline__ void Relax(const SyncTables &L, GlobalWords *words, RelaxShared &R, uint16_t *lists, Lane &ln,
                                      uint32_t total_bits, const daliamdJpegHuffDesc &d, const ScratchLayout &lay,
                                      SliceOf slice_of) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ln = MakeLane(slice_of(tid), total_bits);
  R.last_in[tid] = kNoState;
  R.nstart[tid] = 0;
  R.crossed[tid] = 0;
  __syncthreads();   // (also: the caller's R.state is complete)
  for (int round = 0; round <= kSegThreads; round++) {
    const bool want = ln.active && R.state[tid] != R.last_in[tid];
#if DALIAMD_SYNC_COMPACT
    const unsigned long long m = __ballot(want);
    if (lane == 0) R.wave_count[wave] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSegThreads / 64; w++) {
      const int c = R.wave_count[w];
      base += w < wave ? c : 0;
      total += c;
    }
    if (want) R.work[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)tid;
    __syncthreads();
    if (total == 0) break;
    const int s = tid < total ? (int)R.work[tid] : -1;
#else
    if (!__syncthreads_or(want)) break;
    const int s = want ? tid : -1;
#endif
    uint64_t out = 0;
    if (s >= 0) {
      const Lane w = MakeLane(slice_of(s), total_bits);
      const uint64_t ni = R.state[s];
      R.last_in[s] = ni;
      DecodeState st = Unpack(ni);
      int n = 0;
      bool crossed = false;
      if (st.pos < w.end) {
        const RstView rst = MakeRstView(d, lay, w.begin);
        uint16_t *list = lists + s * kListStride;
        n = SyncDecodeRange(L, words, st, w.end, rst, [&](int nb, int rem, bool) { list[nb < kListCap ? nb : kListCap] = (uint16_t)rem; },
                            &crossed);
      }
      R.nstart[s] = (uint16_t)n;
      R.crossed[s] = crossed ? 1 : 0;
      out = Pack(st);
    }
    __syncthreads();   // every decode of the round has read its input
    if (s >= 0) R.state[s + 1] = out;
    __syncthreads();
    STAMP(0, 2 + (round < 8 ? round : 8));
  }
  ln.in = R.last_in[tid];
  ln.out = ln.in != kNoState ? R.state[tid + 1] : kNoState;
  ln.nstart = R.nstart[tid];
  ln.crossed = R.crossed[tid] != 0;
}

// After the relaxation: the segment's block starts, densely, as absolute bit positions (the lists of its lanes one
// after the other).  Returns the number of starts in the segment.
//  * The lists leave WAVE by wave, one lane's list per step: 64 consecutive dwords per store instruction where a lane
//    walking its own list touched 64 cache lines with every one of its stores.
//  * A lane whose list outgrew its LDS slots (kListCap; a quarter of the bench's slices) decodes once more, writing
//    straight to memory - these lanes are PACKED into the first waves (the work list of the relaxation rounds): measured
//    with per-workgroup time stamps (round 5), that extra decode kept all four waves of nearly every workgroup issuing
//    for the length of a slice, 45 us of the kernel's 305 and a third of its instructions.
struct WriteShared {
  uint32_t bases[kSegThreads], ends[kSegThreads], counts[kSegThreads];
  uint64_t in[kSegThreads];
  uint8_t work[kSegThreads];
  int wave_count[kSegThreads / 64];
};
union SegShared {   // the relaxation's state is dead when the lists leave (the scan's barriers lie in between)
  RelaxShared R;
  WriteShared W;
};
template <typename LaneOf>
__device__ __forceinline__ int WriteSegmentStarts(const SyncTables &L, GlobalWords *words, const Lane &ln, const uint16_t *lists,
                                                  bool mine, GlobalU32 *seg_starts, int seg_cap, int *wave_sums, WriteShared &W,
                                                  const daliamdJpegHuffDesc &d, const ScratchLayout &lay, LaneOf lane_of) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wave_first = tid & ~63;
  const int count = mine ? ln.nstart : 0;
  int total;
  const int base = WorkgroupExclusiveScan<kSegThreads / 64>(count, wave_sums, total);
  W.bases[tid] = (uint32_t)base;
  W.ends[tid] = ln.end;
  W.counts[tid] = (uint32_t)count;
  W.in[tid] = ln.in;
  const bool over = count > kListCap;
  const unsigned long long m = __ballot(over);
  if (lane == 0) W.wave_count[wave] = __popcll(m);
  __syncthreads();
  for (int l = 0; l < 64; l++) {
    const int src = wave_first + l;
    const int n = min((int)W.counts[src], kListCap);   // (uniform per wave: one address)
    if (n == 0) continue;
    const uint32_t b = W.bases[src], e = W.ends[src];
    const uint16_t *list = lists + src * kListStride;
    for (int j = lane; j < n; j += 64)
      if ((int)b + j < seg_cap) seg_starts[b + j] = e - (uint32_t)(int32_t)(int16_t)list[j];
  }
  int first = 0, nover = 0;
#pragma unroll
  for (int w = 0; w < kSegThreads / 64; w++) {
    const int c = W.wave_count[w];
    first += w < wave ? c : 0;
    nover += c;
  }
  if (nover == 0) return total;   // (uniform)
  if (over) W.work[first + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)tid;
  __syncthreads();
  if (tid < nover) {
    const int s = (int)W.work[tid];
    // ... from the LAST start its list holds (entry j = the end of the slice's (j + 1)-th block: zig-zag index 0, block
    // index inside the MCU advanced by j + 1), not from the slice's first bit: the entries in front left with the lists
    const Lane w = lane_of(s);
    const uint32_t end = w.end;
    // (the stream's first slice lists the START of its first block too: one block fewer has ended at the same entry;
    // restart intervals reset the block index at every boundary: those streams decode the slice from its first bit)
    DecodeState st = Unpack(W.in[s]);
    const int kept = d.restart_interval ? 0 : kListCap;   // entries the lists delivered
    if (kept) {
      const uint32_t ended = (uint32_t)kept - (st.pos == 0 ? 1u : 0u);
      st.pos = end - (uint32_t)(int32_t)(int16_t)lists[s * kListStride + kept - 1];
      st.c = (st.c + ended) % (uint32_t)d.blocks_per_mcu;
      st.z = 0;
    }
    const int cnt = (int)W.counts[s] - kept, b = (int)W.bases[s] + kept;
    const RstView rst = MakeRstView(d, lay, st.pos);
    bool crossed;   // the slot behind the last start is "in progress" for ever: it belongs to the next lane
    SyncDecodeRange(L, words, st, end, rst, [&](int nb, int rem, bool ended) {  // one store per block, not per step
      if (ended && nb < cnt && b + nb < seg_cap) seg_starts[b + nb] = end - (uint32_t)rem;
    }, &crossed);
  }
  return total;
}

__global__ __launch_bounds__(kSegThreads) void SyncKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ __attribute__((aligned(16))) SyncTables L;
  __shared__ SegShared S;
  RelaxShared &R = S.R;
  __shared__ uint16_t lists[kSegThreads * kListStride];
  __shared__ int wave_sums[kSegThreads / 64];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  if (d.index) return;   // a resident stream with its index: IndexedSyncKernel (uniform)
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x, seg = r.local;
  const int clean_len = *(const GlobalI32 *)d.scratch;
  LaneRec *recs = reinterpret_cast<LaneRec *>(d.scratch + lay.lanes) + (size_t)seg * kSegLanes;
  SegRec *segrec = reinterpret_cast<SegRec *>(d.scratch + lay.segs) + seg;
  if (seg > 0 && (long long)seg * kSegBytes >= clean_len) {  // segment behind the end of the stream
    if (tid >= kWarmLanes) recs[tid - kWarmLanes] = LaneRec{kNoState, kNoState, 0, 0};
    if (tid == 0) *segrec = SegRec{kNoState, 0, 0, {0, 0, 0}, 0};
    return;
  }
  STAMP(0, 0);
  STAMP_IDS(0, wg);
  CopyTables<kSegThreads>(L, reinterpret_cast<const SyncTables *>(TablesBase(descs, d, true)));
  STAMP(0, 1);
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  // lanes [0, kWarmLanes) replay the last slices of the previous segment, lanes [kWarmLanes, ..) are this segment's
  auto slice_of = [&](int t) { return (long long)seg * kSegLanes + t - kWarmLanes; };
  Lane ln = MakeLane(slice_of(tid), total_bits);
  R.state[tid] = Pack(DecodeState{ln.begin, 0, 0});  // the guess; exact for the very first slice of the image
  GlobalWords *words = (GlobalWords *)(d.scratch + lay.clean);
  Relax(L, words, R, lists, ln, total_bits, d, lay, slice_of);
  STAMP(0, 11);
  const bool mine = tid >= kWarmLanes;
  const int total = WriteSegmentStarts(L, words, ln, lists, mine, (GlobalU32 *)(d.scratch + lay.seg_starts) + (size_t)seg * lay.seg_cap,
                                       lay.seg_cap, wave_sums, S.W, d, lay, [&](int t) { return MakeLane(slice_of(t), total_bits); });
  // restart intervals: did a lane's last decode run over a boundary?  (Wrong when the lanes started from the truth -
  // PropagateKernel knows whether they did.)
  const int crossed = d.restart_interval ? __syncthreads_or(mine && ln.active && ln.crossed) : 0;
  if (mine) {
    recs[tid - kWarmLanes] = LaneRec{ln.in, ln.out, ln.nstart, 0};
    // the last slice with data ends the segment (an empty stream: the first lane passes its input on)
    const bool next_has_data = tid + 1 < kSegThreads && ln.end < total_bits;
    if (ln.active && !next_has_data) *segrec = SegRec{ln.out, total, 0, {0, 0, 0}, crossed};
    if (total_bits == 0 && tid == kWarmLanes) *segrec = SegRec{Pack(DecodeState{0, 0, 0}), 0, 0, {0, 0, 0}, 0};
  }
  STAMP(0, 12);
}

// (The code tables of the rare repair stay in global memory: a 50 KB workgroup - the tables are 36 KB of it - waits for a
// CU with that much LDS free, which inside the five-batch schedule made this 10 us kernel last 50 us.)
__global__ __launch_bounds__(kSegThreads) void PropagateKernel(const daliamdJpegHuffDesc *__restrict__ descs) {
  __shared__ SegShared S;
  RelaxShared &R = S.R;
  __shared__ uint16_t lists[kSegThreads * kListStride];
  __shared__ int wave_sums[kSegThreads / 64];
  const daliamdJpegHuffDesc &d = descs[blockIdx.x];
  if (d.index) return;
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x;
  const int clean_len = *(const GlobalI32 *)d.scratch;
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  LaneRec *all_recs = reinterpret_cast<LaneRec *>(d.scratch + lay.lanes);
  SegRec *segs = reinterpret_cast<SegRec *>(d.scratch + lay.segs);
  GlobalWords *words = (GlobalWords *)(d.scratch + lay.clean);
  uint64_t truth = Pack(DecodeState{0, 0, 0});
  int block_base = 0;
  const SyncTables &L = *reinterpret_cast<const SyncTables *>(TablesBase(descs, d, true));
  for (int seg = 0; seg < d.num_segments; seg++) {
    if (seg > 0 && (long long)seg * kSegBytes >= clean_len) {
      if (tid == 0) {
        segs[seg].out = truth;
        segs[seg].block_base = block_base;
      }
      continue;
    }
    LaneRec *recs = all_recs + (size_t)seg * kSegLanes;
    if (total_bits != 0 && recs[0].in != truth) {
      // The warm-up lanes did not synchronise before this segment (long flat or periodic content): repair it.
      // every lane decodes again (its start list lives in LDS only while the kernel that decoded it runs)
      const bool mine = tid < kSegLanes;
      auto slice_of = [&](int t) { return t < kSegLanes ? (long long)seg * kSegLanes + t : -1ll; };
      Lane ln;
      R.state[tid] = tid == 0 ? truth : (mine ? recs[tid].in : kNoState);
      Relax(L, words, R, lists, ln, total_bits, d, lay, slice_of);
      const int total = WriteSegmentStarts(L, words, ln, lists, mine,
                                           (GlobalU32 *)(d.scratch + lay.seg_starts) + (size_t)seg * lay.seg_cap, lay.seg_cap,
                                           wave_sums, S.W, d, lay, [&](int t) { return MakeLane(slice_of(t), total_bits); });
      const int crossed = d.restart_interval ? __syncthreads_or(mine && ln.active && ln.crossed) : 0;
      if (mine) {
        recs[tid] = LaneRec{ln.in, ln.out, ln.nstart, 0};
        const bool next_has_data = tid + 1 < kSegLanes && ln.end < total_bits;
        if (ln.active && !next_has_data) {
          segs[seg].out = ln.out;
          segs[seg].nstart_total = total;
          segs[seg].crossed = crossed;
        }
      }
      __threadfence();
      __syncthreads();  // the records written above are read below (same workgroup)
    }
    // every lane of the segment has decoded from the true state by now: a restart boundary met in the middle of an MCU
    // means the intervals are not padded with one-bits (or hold the wrong number of MCUs) - not for this decoder
    if (tid == 0 && segs[seg].crossed) *d.status = 4;
    if (tid == 0) segs[seg].block_base = block_base;
    block_base += segs[seg].nstart_total;
    truth = segs[seg].out;
  }
  // every block of the frame must have started AND ended inside the segment (the end of the last one is the start
  // of a block that does not exist; the padding may add garbage after it)
  if (tid == 0 && block_base - 1 < d.total_blocks) *d.status = 2;
  if (tid == 0) reinterpret_cast<int32_t *>(d.scratch)[2] = block_base;  // block starts the stream really holds
}

// ------------------------------------------------------------------------------------------------ value passes
// First block ordinal behind the last MCU row a region-of-interest decode needs (total_blocks: everything).
__device__ __forceinline__ int LastOrdinal(const daliamdJpegHuffDesc &d) {
  int use_rect = 0, last_mcu_row = 0;
  for (int k = 0; k < d.blocks_per_mcu; k++) {
    const int comp = d.comp_of_block[k];
    if (d.rect[comp][2] > d.rect[comp][0] && d.rect[comp][3] > d.rect[comp][1]) {
      use_rect = 1;
      const int rows = (d.rect[comp][3] + d.v_samp[comp] - 1) / d.v_samp[comp];  // MCU rows up to the rectangle's bottom
      last_mcu_row = rows > last_mcu_row ? rows : last_mcu_row;
    }
  }
  const long long last = (long long)last_mcu_row * d.mcus_x * d.blocks_per_mcu;
  return use_rect && last < d.total_blocks ? (int)last : d.total_blocks;
}

// Region-of-interest decode: the rectangle of MCUs that holds every block some component's rect asks for
// ({x0, y0, cols, rows} in MCUs; false: no rect, every MCU).  The block kernel's workgroups walk THIS rectangle in raster
// order - the MCUs outside it are parsed by the position passes (the stream is serial) but get no lane of the value pass.
__host__ __device__ inline bool RectMcus(const daliamdJpegHuffDesc &d, int out[4]) {
  int x0 = 1 << 30, y0 = 1 << 30, x1 = 0, y1 = 0, any = 0;
  for (int k = 0; k < d.blocks_per_mcu; k++) {
    const int comp = d.comp_of_block[k];
    const int32_t *r = d.rect[comp];
    if (!(r[2] > r[0] && r[3] > r[1])) continue;
    any = 1;
    const int hs = d.h_samp[comp], vs = d.v_samp[comp];
    x0 = r[0] / hs < x0 ? r[0] / hs : x0;
    y0 = r[1] / vs < y0 ? r[1] / vs : y0;
    x1 = (r[2] + hs - 1) / hs > x1 ? (r[2] + hs - 1) / hs : x1;
    y1 = (r[3] + vs - 1) / vs > y1 ? (r[3] + vs - 1) / vs : y1;
  }
  if (!any) return false;
  const int mcus_y = d.total_blocks / d.blocks_per_mcu / d.mcus_x;
  x1 = x1 < d.mcus_x ? x1 : d.mcus_x;
  y1 = y1 < mcus_y ? y1 : mcus_y;
  out[0] = x0; out[1] = y0; out[2] = x1 > x0 ? x1 - x0 : 0; out[3] = y1 > y0 ? y1 - y0 : 0;
  return true;
}

// The DC or the AC half of HuffTables in LDS, indexed by table selector (LongCode / DecodeDc / DecodeBlockAc address
// the members by name).
struct HalfTables {
  uint16_t fast[2][1 << kFastBits];
  uint16_t l2[2][kL2Entries];
  int32_t l2_first[2], l2_size[2];
  int32_t maxcode[2][18], valoff[2][18];
  uint8_t vals[2][256];
};
// copies tables [first, first + 2) of H (fast[] and l2[] are contiguous there)
template <int THREADS>
__device__ __forceinline__ void CopyHalfTables(HalfTables &T, const HuffTables *H, int first) {
  const int tid = threadIdx.x;
  CopyChunks<THREADS>(reinterpret_cast<uint4 *>(&T.fast[0][0]), (const GlobalQuad *)&H->fast[first][0], (int)(sizeof(T.fast) / 16));
  CopyChunks<THREADS>(reinterpret_cast<uint4 *>(&T.l2[0][0]), (const GlobalQuad *)&H->l2[first][0], (int)(sizeof(T.l2) / 16));
  // (vals[first .. first + 1] are 512 contiguous bytes in both structures)
  CopyChunks<THREADS>(reinterpret_cast<uint4 *>(&T.vals[0][0]), (const GlobalQuad *)&H->vals[first][0], (int)(sizeof(T.vals) / 16));
  if (tid < 36) {
    T.maxcode[tid / 18][tid % 18] = H->maxcode[first + tid / 18][tid % 18];
    T.valoff[tid / 18][tid % 18] = H->valoff[first + tid / 18][tid % 18];
  }
  if (tid < 2) {
    T.l2_first[tid] = H->l2_first[first + tid];
    T.l2_size[tid] = H->l2_size[first + tid];
  }
}

// Exclusive scan of three ints per lane over the workgroup with ONE pair of barriers.
template <int NW>
__device__ __forceinline__ void WorkgroupExclusiveScan3(const int v[3], int excl[3], int total[3], int (*wave_sums)[3]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl[3] = {v[0], v[1], v[2]};
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int t = __shfl_up(incl[c], off, 64);
      if (lane >= off) incl[c] += t;
    }
  }
  if (lane == 63)
    for (int c = 0; c < 3; c++) wave_sums[wave][c] = incl[c];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; c++) {
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const int sum = wave_sums[w][c];
      if (w < wave) base += sum;
      tot += sum;
    }
    excl[c] = base + incl[c] - v[c];
    total[c] = tot;
  }
  __syncthreads();  // wave_sums is reused by the next call
}

// DC pass, one workgroup per segment, one lane per four consecutive blocks that start in it: the DC difference of the
// block (one look-up, DC tables in LDS), inclusive prefix sums per component over the segment (the levels relative to
// the segment's start; BlockKernel adds the totals of the segments before), the bit position behind the DC symbol.
constexpr int kDcThreads = 256;
constexpr int kDcPerThread = 4;
__global__ __launch_bounds__(kDcThreads) void DcKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ __attribute__((aligned(16))) HalfTables T;
  __shared__ int wave_sums[kDcThreads / 64][3];
  __shared__ uint8_t comp_of[16], dcsel_of[16];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  if (d.index) return;   // (the indexed pass knows the DC predictors)
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x, seg = r.local;
  const int clean_len = *(const GlobalI32 *)d.scratch;
  SegRec *segrec = reinterpret_cast<SegRec *>(d.scratch + lay.segs) + seg;
  if (seg > 0 && (long long)seg * kSegBytes >= clean_len) return;  // dc_total is already zero
  CopyHalfTables<kDcThreads>(T, reinterpret_cast<const HuffTables *>(TablesBase(descs, d, false)), 0);
  if (tid < d.blocks_per_mcu) {
    comp_of[tid] = d.comp_of_block[tid];
    dcsel_of[tid] = d.dc_sel[d.comp_of_block[tid]] & 1;
  }
  __syncthreads();
  const int bpm = d.blocks_per_mcu;
  const int total_starts = ((const GlobalI32 *)d.scratch)[2];
  // (a decode that also builds the stream's index needs the DC level in front of EVERY slice, not only up to the window)
  const int last_ordinal = d.index_out ? d.total_blocks : LastOrdinal(d);
  GlobalI32 *segrec_i = (GlobalI32 *)segrec;
  const int block_base = segrec_i[kSegRecBlockBase];
  int nstart = segrec_i[kSegRecNstart];
  nstart = nstart < lay.seg_cap ? nstart : lay.seg_cap;
  GlobalWords *words = (GlobalWords *)(d.scratch + lay.clean);
  GlobalWords *starts = (GlobalWords *)(d.scratch + lay.seg_starts) + (size_t)seg * lay.seg_cap;
  GlobalU32 *blk_pos = (GlobalU32 *)(d.scratch + lay.blk_pos);
  GlobalI32 *blk_dc = (GlobalI32 *)(d.scratch + lay.blk_dc);
  GlobalU16 *blk_seg = (GlobalU16 *)(d.scratch + lay.blk_seg);
  int carry[3] = {0, 0, 0};
  for (int j0 = 0; j0 < nstart; j0 += kDcThreads * kDcPerThread) {
    const int jt = j0 + tid * kDcPerThread;
    bool valid[kDcPerThread];
    uint32_t pos[kDcPerThread], comp[kDcPerThread], used[kDcPerThread];
    int diff[kDcPerThread];
#pragma unroll
    for (int q = 0; q < kDcPerThread; q++) {
      const int j = jt + q, ordinal = block_base + j;
      // a block counts when it starts AND ends inside the stream (its end is the next start) and the decode needs it
      valid[q] = j < nstart && ordinal < last_ordinal && ordinal + 1 < total_starts;
      pos[q] = valid[q] ? starts[j] : 0u;
    }
    int sum[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < kDcPerThread; q++) {
      const int k = (block_base + jt + q) % bpm;
      comp[q] = comp_of[k];
      used[q] = 0;
      diff[q] = valid[q] ? DecodeDc(T, words, pos[q], (uint32_t)dcsel_of[k], &used[q]) : 0;
#pragma unroll
      for (int c = 0; c < 3; c++) sum[c] += comp[q] == (uint32_t)c ? diff[q] : 0;
    }
    int excl[3], total[3];
    WorkgroupExclusiveScan3<kDcThreads / 64>(sum, excl, total, wave_sums);
    int run[3] = {excl[0] + carry[0], excl[1] + carry[1], excl[2] + carry[2]};
#pragma unroll
    for (int c = 0; c < 3; c++) carry[c] += total[c];
#pragma unroll
    for (int q = 0; q < kDcPerThread; q++) {
      int mine = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        run[c] += comp[q] == (uint32_t)c ? diff[q] : 0;
        mine = comp[q] == (uint32_t)c ? run[c] : mine;
      }
      if (valid[q]) {
        const int ordinal = block_base + jt + q;
        blk_pos[ordinal] = pos[q] + used[q];
        blk_dc[ordinal] = mine;
        blk_seg[ordinal] = (uint16_t)seg;
      }
    }
  }
  if (tid < 3) segrec_i[kSegRecDcTotal + tid] = carry[tid];
}

// ------------------------------------------------------------------------------------------------ resident streams
// Per block of an indexed stream: {bit position behind its DC symbol, DC level (mod 2^16)} - one 8-byte record where the
// other path keeps blk_pos / blk_dc / blk_seg (the two arrays are adjacent in the scratch: the records lie over them).
typedef uint32_t u32x2p __attribute__((ext_vector_type(2)));
using GlobalPosDc = u32x2p __attribute__((address_space(1)));
using GlobalIndex = const SliceIndex __attribute__((address_space(1)));
__device__ __forceinline__ SliceIndex LoadIndex(GlobalIndex *p) { return SliceIndex{p->w0, p->w1, p->w2}; }

// Position pass of the streams that bring their index (daliamdJpegHuffDesc.index): workgroup = a segment's 244 slices as
// in SyncKernel (same grid), ONE decode per slice from its recorded entry state, the DC symbol of every block start on
// the way (huff_core.h: IndexedDecodeSlice), results straight into the per-block records.  With a region of interest
// the slices whose blocks it does not need are not decoded (their ordinals are in the index), and the slices that are
// get packed into the first waves; a segment without any leaves before it copies its tables.
__global__ __launch_bounds__(kSegThreads) void IndexedSyncKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ __attribute__((aligned(16))) SyncTables L;
  __shared__ __attribute__((aligned(16))) HalfTables D;
  __shared__ uint8_t work[kSegThreads];
  __shared__ int wave_count[kSegThreads / 64];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  if (!d.index) return;   // (uniform)
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, seg = r.local;
  const IndexHeader *hdr = reinterpret_cast<const IndexHeader *>(d.index);
  const int clean_len = ((const GlobalI32 *)hdr)[0];
  if ((long long)seg * kSegBytes >= clean_len) return;
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  GlobalIndex *entries = (GlobalIndex *)(d.index + IndexEntriesOffset(d.ecs_len));
  // what the decode needs of this stream: blocks below `last` inside the MCU rectangle `rm` (no rectangle: all of them)
  const int bpm = d.blocks_per_mcu, mcus_x = d.mcus_x;
  const int last = min(LastOrdinal(d), d.total_blocks);
  int rm[4] = {0, 0, 0, 0};
  const bool use_rect = RectMcus(d, rm);
  auto needed = [&](int b0, int b1) {   // any block of [b0, b1) inside the rectangle?
    b1 = min(b1, last);
    if (b0 >= b1) return false;
    if (!use_rect) return true;
    const int m0 = b0 / bpm, m1 = (b1 - 1) / bpm;
    const int r0 = m0 / mcus_x, r1 = m1 / mcus_x;
    const int c0 = m0 - r0 * mcus_x, c1 = m1 - r1 * mcus_x;
    const int x0 = rm[0], x1 = rm[0] + rm[2], y0 = rm[1], y1 = rm[1] + rm[3];
    if (x1 <= x0) return false;
    auto row = [&](int ry, int lo, int hi) { return ry >= y0 && ry < y1 && lo < x1 && hi >= x0; };   // columns [lo, hi]
    if (r0 == r1) return row(r0, c0, c1);
    return row(r0, c0, mcus_x - 1) || row(r1, 0, c1) || (max(r0 + 1, y0) < min(r1, y1));
  };
  const long long slice = (long long)seg * kSegLanes + tid;
  STAMP(1, 0);
  STAMP_IDS(1, wg);
  const bool has = tid < kSegLanes && slice * (kSliceBytes * 8ll) < (long long)total_bits;
  bool want = false;
  if (has) {
    const uint32_t b0 = entries[slice].w0 & kIndexOrdinalMask, b1 = entries[slice + 1].w0 & kIndexOrdinalMask;
    want = needed((int)b0, (int)b1);
  }
  // the slices to decode, packed into the first lanes
  const unsigned long long m = __ballot(want);
  if (lane == 0) wave_count[wave] = __popcll(m);
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kSegThreads / 64; w++) {
    const int c = wave_count[w];
    base += w < wave ? c : 0;
    total += c;
  }
  if (total == 0) return;   // (uniform: nothing of this segment is needed)
  if (want) work[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)tid;
  CopyTables<kSegThreads>(L, reinterpret_cast<const SyncTables *>(TablesBase(descs, d, true)));
  CopyHalfTables<kSegThreads>(D, reinterpret_cast<const HuffTables *>(TablesBase(descs, d, false)), 0);
  __syncthreads();
  STAMP(1, 1);
  if (tid >= total) return;
  const long long s = (long long)seg * kSegLanes + work[tid];
  const SliceIndex e = LoadIndex(entries + s);
  const uint32_t begin = (uint32_t)(s * (kSliceBytes * 8ll));
  const uint32_t end = begin + kSliceBytes * 8u < total_bits ? begin + kSliceBytes * 8u : total_bits;
  DecodeState st = IndexEntryState(e);
  st.pos += begin;
  if (st.pos >= end) return;
  uint32_t comp_of = 0;   // two bits per block of the MCU
  for (int k = 0; k < bpm; k++) comp_of |= (uint32_t)(d.comp_of_block[k] & 3) << (2 * k);
  comp_of = HUFF_UNIFORM(comp_of);
  int dc0 = (int)(e.w1 >> 16), dc1 = (int)(e.w2 & 0xFFFFu), dc2 = (int)(e.w2 >> 16);
  uint32_t b = IndexFirstBlock(e);
  const uint32_t total_blocks = (uint32_t)d.total_blocks;
  GlobalPosDc *blk = (GlobalPosDc *)(d.scratch + lay.blk_pos);
  GlobalWords *words = (GlobalWords *)(d.index + IndexCleanOffset());
  IndexedDecodeSlice(L, D, words, st, end, [&](uint32_t c, uint32_t pos, int diff) {
    const uint32_t comp = (comp_of >> (2 * c)) & 3u;
    dc0 += comp == 0 ? diff : 0;
    dc1 += comp == 1 ? diff : 0;
    dc2 += comp == 2 ? diff : 0;
    const int level = comp == 0 ? dc0 : comp == 1 ? dc1 : dc2;
    if (b < total_blocks) blk[b] = u32x2p{pos, (uint32_t)level};
    b++;
  });
  STAMP_MAX(1, 2 + wave);
}

// Builds the index entry of a stream (daliamdJpegHuffDesc.index_out) from what this batch's position passes left in the
// scratch - behind PropagateKernel (every slice's input state is the truth by then) and a DcKernel that covered every
// block.  Workgroup = segment: copies its part of the clean stream, writes the entries of its slices.
__global__ __launch_bounds__(kSegThreads) void IndexBuildKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nseg) {
  __shared__ int wave_sums[kSegThreads / 64];
  const int wg = XcdRemap(blockIdx.x, nseg);
  if (wg < 0) return;
  const ImageRef r = FindImage<false>(descs, n, wg);
  const daliamdJpegHuffDesc &d = *r.d;
  if (!d.index_out || d.index) return;   // (uniform)
  const ScratchLayout lay = LayoutOf(d);
  const int tid = threadIdx.x, seg = r.local;
  const int clean_len = *(const GlobalI32 *)d.scratch;
  const int total_starts = ((const GlobalI32 *)d.scratch)[2];
  const int cap = IndexSliceCap(d.ecs_len);
  GlobalU32 *out_entries = (GlobalU32 *)(d.index_out + IndexEntriesOffset(d.ecs_len));
  auto put = [&](long long slice, const SliceIndex &e) {
    out_entries[3 * slice] = e.w0; out_entries[3 * slice + 1] = e.w1; out_entries[3 * slice + 2] = e.w2;
  };
  const uint32_t zero3[3] = {0, 0, 0};
  const SliceIndex none = PackSliceIndex((uint32_t)total_starts, DecodeState{0, 0, 0}, zero3);
  if (seg == 0 && tid == kSegThreads - 1) {
    GlobalI32 *h = (GlobalI32 *)d.index_out;
    h[0] = clean_len; h[1] = total_starts; h[2] = (clean_len + kSliceBytes - 1) / kSliceBytes;
    for (int i = 3; i < 16; i++) h[i] = 0;
    put(cap, none);   // the sentinel behind the last slice
  }
  // ---- the clean bytes of the segment (the last one: with the all-ones padding behind the stream)
  {
    const long long first = (long long)seg * kSegBytes;
    const long long stream_end = (long long)AlignUp((size_t)clean_len + kCleanPadBytes, 16);
    const long long stop = min(first + kSegBytes, stream_end);
    const GlobalQuad *src = (const GlobalQuad *)(d.scratch + lay.clean);
    GlobalQuad *dst = (GlobalQuad *)(d.index_out + IndexCleanOffset());
    for (long long q = first / 16 + tid; q * 16 < stop; q += kSegThreads) dst[q] = src[q];
  }
  // ---- the entries of its slices
  const long long slice = (long long)seg * kSegLanes + tid;
  const bool mine = tid < kSegLanes && slice < cap;
  const bool active = mine && slice * (long long)kSliceBytes < clean_len;
  const LaneRec *recs = reinterpret_cast<const LaneRec *>(d.scratch + lay.lanes) + (size_t)seg * kSegLanes;
  const GlobalI32 *segs_i = (const GlobalI32 *)(d.scratch + lay.segs);
  uint64_t in = kNoState;
  int nstart = 0;
  if (active) {
    in = recs[tid].in;
    nstart = recs[tid].nstart;
  }
  int total;
  const int excl = WorkgroupExclusiveScan<kSegThreads / 64>(nstart, wave_sums, total);
  if (!mine) return;
  if (!active || in == kNoState) {
    put(slice, none);
    return;
  }
  const int ord = segs_i[seg * kSegRecInts + kSegRecBlockBase] + excl;   // first start this slice's decode FOUND
  DecodeState st = Unpack(in);
  // ... the first block whose DC symbol it DECODES: a block that starts exactly at the entry position was found by the
  // slice before (the step that ended its predecessor) and is decoded here - except the stream's very first block
  const int first_block = ord - (st.z == 0 && st.pos != 0 ? 1 : 0);
  // DC level of every component in front of that block: its last block before, relative to its segment + the segments before
  const GlobalI32 *blk_dc = (const GlobalI32 *)(d.scratch + lay.blk_dc);
  const GlobalU16 *blk_seg = (const GlobalU16 *)(d.scratch + lay.blk_seg);
  uint32_t dc[3] = {0, 0, 0};
  uint32_t found = 0;
  for (int j = 1; j <= d.blocks_per_mcu && found != 7u; j++) {
    const int q = first_block - j;
    if (q < 0) break;
    const int comp = d.comp_of_block[q % d.blocks_per_mcu];
    if (comp > 2 || (found >> comp) & 1u) continue;
    found |= 1u << comp;
    if (q >= d.total_blocks || q + 1 >= total_starts) continue;   // (a block the stream does not really hold)
    int level = blk_dc[q];
    const int qs = blk_seg[q];
    for (int sg = 0; sg < qs; sg++) level += segs_i[sg * kSegRecInts + kSegRecDcTotal + comp];
    dc[comp] = (uint32_t)level & 0xFFFFu;
  }
  st.pos -= (uint32_t)(slice * (long long)(kSliceBytes * 8));
  put(slice, PackSliceIndex((uint32_t)first_block, st, dc));
}

// Value pass.  A workgroup owns a run of MCUs of one image; its waves take TASKS of 64 blocks that all use the same
// AC table (luma / chroma blocks differ 2-3x in their number of symbols: a wave's loop lasts as long as its longest
// block), one block per lane: the lane zero-fills nothing but decodes its AC symbols straight into its 64-coefficient
// slot in LDS (natural zig-zag order, plus the DC level; coefficient-major, see kCoefRows), then transforms that block in
// its own registers - the two passes of JpegIdctKernel, sixteen 8-point butterflies - and stores its eight rows of samples.
// Waves never wait for each other.
// Workgroup shape (MI355X, headline batch, waves x tasks per wave; round 3, four batches in flight through the product
// pipeline): 4x3 (45 KB of LDS: three workgroups per CU) 150 us alone / 384k images per second, 3x4 377k, 6x2 356k.
// (Round 2, block-major LDS with the 8-lanes-per-block IDCT: 2x3 216 us, 3x2 201, 4x3 204, 6x2 258, 12x1 196.)
#ifndef DALIAMD_BLOCK_WAVES
#define DALIAMD_BLOCK_WAVES 4
#endif
constexpr int kBlockWaves = DALIAMD_BLOCK_WAVES;  // per wave 8.3 KB of coefficients, + 11 KB of tables per workgroup
constexpr int kBlockThreads = kBlockWaves * 64;
// Coefficients of a wave's 64 blocks in LDS, COEFFICIENT-major: element k of lane l's block is the int16 at k * 64 + l
// (row 64 swallows the stores that carry nothing).  Whatever positions the lanes of a wave store to in one step of the
// decode loop, lane l only ever touches bank l / 2 (+ 32 for odd k): at most two lanes meet in a bank, where the
// block-major layout this replaces (33 dwords per block) spread 64 unrelated positions over the banks at random and
// paid 3-4 conflict cycles per store (SQ_LDS_BANK_CONFLICT 12.6 M against 8.6 M busy LDS cycles in round 2's profile).
// Reading a fixed coefficient for all lanes (the IDCT below) is one contiguous 128-byte row: conflict-free.
constexpr int kCoefRows = 65;
struct LaneCoef {   // what DecodeBlockAc stores through
  int16_t *base;
  __device__ __forceinline__ int16_t &operator[](uint32_t k) const { return base[k * 64]; }
};
#ifndef DALIAMD_BLOCK_TASKS
#define DALIAMD_BLOCK_TASKS 3
#endif
constexpr int kBlocksPerWg = 64 * DALIAMD_BLOCK_TASKS * kBlockWaves;  // tasks per wave
__host__ __device__ inline int McusPerWg(int bpm) {
  const int m = (kBlocksPerWg / bpm) / 32 * 32;
  return m > 32 ? m : 32;
}
struct BlockGeom {
  int32_t bpm, mcus_x, total_mcus, last_ordinal, use_rect, total_starts, fused, n0;
  uint32_t total_bits;   // of the clean stream
  int32_t interval_blocks;  // blocks per restart interval (0: none)
  uint8_t klast[4];   // per component: its last block inside the MCU
  uint8_t klist[12];  // block indices of the MCU, the ones with AC table 0 first
  uint8_t comp[12], acs[12], hs[12], vs[12], ho[12], vo[12];
  int32_t sx[12], sy[12], rect[12][4], pitch[12], comp_pitch[4];
  GlobalCoef *base[12];
  GlobalBytes *plane[12];
  // fused colour output (BlockKernel<true>)
  int32_t roi_x0, roi_y0, roi_cols, roi_mcus;  // region-of-interest decode: the MCU rectangle the workgroups walk (RectMcus)
  int32_t band_rows, band_c0, band_c1, bands;  // MCU rows per band; chroma rows [c0, c1) of this workgroup's band
  int32_t width, height, dw, dh, rgb_pitch;
  GlobalBytes *rgb, *seams;
};
// Fused colour output: a workgroup owns a band of whole MCU rows, as many as fit kColorBandMcus MCUs (the chroma tile
// in LDS: 2 x 64 bytes per MCU).
constexpr int kColorBandMcus = 128;
__host__ __device__ inline int ColorBandRows(int mcus_x) {
  const int r = kColorBandMcus / mcus_x;
  return r > 1 ? r : 1;
}
// seam strip `side` (0: first pixel row, 1: last pixel row) of band b: [0, 16 mx) luma, [16 mx, 24 mx) Cb, [24 mx, 32 mx) Cr
__host__ __device__ inline size_t SeamStrip(int band, int side, int mcus_x) { return (size_t)(band * 2 + side) * 32u * (size_t)mcus_x; }
// Per block of a task, where its output goes: bits 0-47 the address (fused: top-left sample of the block in its
// plane; else the block's 64 coefficients), bits 48-49 the component, bit 50 "needed".
constexpr uint64_t kInfoNeeded = 1ull << 50;
// position in the scan (zig-zag index) of the coefficient at column-major block position p = column * 8 + row,
// a compile-time function (LDS offsets of the IDCT's reads become immediates)
__host__ __device__ constexpr int kScanIndexOfColMajorHost(int p) {
  constexpr uint8_t t[64] = {0, 2, 3, 9, 10, 20, 21, 35, 1, 4, 8, 11, 19, 22, 34, 36, 5, 7, 12, 18, 23, 33, 37, 48, 6, 13, 17, 24,
                             32, 38, 47, 49, 14, 16, 25, 31, 39, 46, 50, 57, 15, 26, 30, 40, 45, 51, 56, 58, 27, 29, 41, 44, 52, 55,
                             59, 62, 28, 42, 43, 53, 54, 60, 61, 63};
  return t[p];
}

// LDS accesses of one wave are executed in order; this only keeps the compiler from moving them across the point
__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool kColor>
__global__ __launch_bounds__(kBlockThreads) void BlockKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nwg) {
  __shared__ __attribute__((aligned(16))) HalfTables T;
  __shared__ __attribute__((aligned(16))) int16_t coef[kBlockWaves][kCoefRows * 64];
  __shared__ __attribute__((aligned(16))) uint16_t quant[3][64];
  __shared__ BlockGeom G;
  // kColor: the chroma samples of the band, [Cb | Cr][chroma row of the band][8 * mcus_x] - 64 bytes per MCU and component
  constexpr int kChromaTile = kColorBandMcus * 64;
  __shared__ __attribute__((aligned(16))) uint8_t ctile[kColor ? 2 * kChromaTile : 16];
  const int wg = XcdRemap(blockIdx.x, nwg);
  if (wg < 0) return;
  // workgroup -> image (descriptors sorted by blk_wg_start)
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk_wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  const daliamdJpegHuffDesc &d = descs[lo];
  if ((d.rgb != nullptr) != kColor) return;   // the other instance's stream (uniform)
  STAMP(2, 0);
  STAMP_IDS(2, wg);
  const ScratchLayout lay = LayoutOf(d);
  const HuffTables *H = reinterpret_cast<const HuffTables *>(TablesBase(descs, d, false));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  CopyHalfTables<kBlockThreads>(T, H, 2);  // the AC tables
  if (tid < 12 && tid < d.blocks_per_mcu) {
    const int comp = d.comp_of_block[tid];
    G.comp[tid] = (uint8_t)comp;
    G.acs[tid] = d.ac_sel[comp] & 1;
    G.hs[tid] = (uint8_t)d.h_samp[comp];
    G.vs[tid] = (uint8_t)d.v_samp[comp];
    G.ho[tid] = d.h_of_block[tid];
    G.vo[tid] = d.v_of_block[tid];
    G.sx[tid] = d.h_samp[comp] * 64;
    G.sy[tid] = d.v_samp[comp] * d.blocks_x[comp] * 64;
    G.base[tid] = (GlobalCoef *)d.coef[comp] + ((size_t)d.v_of_block[tid] * d.blocks_x[comp] + d.h_of_block[tid]) * 64;
    for (int j = 0; j < 4; j++) G.rect[tid][j] = d.rect[comp][j];
    G.plane[tid] = (GlobalBytes *)d.plane[comp];
    G.pitch[tid] = d.plane_pitch[comp];
    G.comp_pitch[comp] = d.plane_pitch[comp];
  }
  for (int i = tid; i < 3 * 64; i += kBlockThreads) quant[i >> 6][i & 63] = d.quant[i >> 6][i & 63];
  if (tid == kBlockThreads - 1) {
    G.bpm = d.blocks_per_mcu;
    G.mcus_x = d.mcus_x;
    G.total_mcus = d.total_blocks / d.blocks_per_mcu;
    G.last_ordinal = LastOrdinal(d);
    int use_rect = 0;
    for (int k = 0; k < d.blocks_per_mcu; k++) {
      const int comp = d.comp_of_block[k];
      use_rect |= d.rect[comp][2] > d.rect[comp][0] && d.rect[comp][3] > d.rect[comp][1];
    }
    G.use_rect = use_rect;
    int rm[4] = {0, 0, 0, 0};
    const bool roi = !kColor && RectMcus(d, rm);
    G.roi_x0 = rm[0]; G.roi_y0 = rm[1]; G.roi_cols = roi ? rm[2] : 0; G.roi_mcus = rm[2] * rm[3];
    // (a resident stream with its index: the header of the index knows, and the per-block records hold absolute levels)
    G.total_starts = d.index ? ((const GlobalI32 *)d.index)[1] : ((const GlobalI32 *)d.scratch)[2];
    G.total_bits = (uint32_t)(d.index ? ((const GlobalI32 *)d.index)[0] : ((const GlobalI32 *)d.scratch)[0]) * 8u;
    G.fused = kColor || d.plane[d.comp_of_block[0]] != nullptr;
    G.interval_blocks = d.index ? 0 : d.restart_interval * d.blocks_per_mcu;
    for (int k = 0; k < d.blocks_per_mcu; k++) G.klast[d.comp_of_block[k]] = (uint8_t)k;
    // the two task classes: by AC table - or, with the fused colour output, luma / chroma (usually the same split)
    int n0 = 0;
    for (int k = 0; k < d.blocks_per_mcu; k++)
      if (kColor ? d.comp_of_block[k] == 0 : (d.ac_sel[d.comp_of_block[k]] & 1) == 0) G.klist[n0++] = (uint8_t)k;
    G.n0 = n0;
    for (int k = 0; k < d.blocks_per_mcu; k++)
      if (kColor ? d.comp_of_block[k] != 0 : (d.ac_sel[d.comp_of_block[k]] & 1) != 0) G.klist[n0++] = (uint8_t)k;
    if (kColor) {
      const int rows = ColorBandRows(d.mcus_x), band = wg - d.blk_wg_start, mcus_y = d.total_blocks / d.blocks_per_mcu / d.mcus_x;
      G.band_rows = rows;
      G.bands = (mcus_y + rows - 1) / rows;
      G.band_c0 = band * rows * 8;
      G.band_c1 = min((band + 1) * rows, mcus_y) * 8;
      G.width = d.width; G.height = d.height;
      G.dw = (d.width + 1) >> 1; G.dh = (d.height + 1) >> 1;
      G.rgb_pitch = d.rgb_pitch;
      G.rgb = (GlobalBytes *)d.rgb;
      G.seams = (GlobalBytes *)(d.scratch + lay.seams);
    }
  }
  __syncthreads();
  STAMP(2, 1);
  const int bpm = G.bpm, mpw = kColor ? G.band_rows * G.mcus_x : McusPerWg(bpm);
  const int band = wg - d.blk_wg_start;
  const int m0 = band * mpw;
  // fused colour output: the band's geometry in scalar registers (G is LDS: the byte stores into the chroma tile would make
  // the compiler read it again and again)
  const int cmx = (int)HUFF_UNIFORM(G.mcus_x), cpitch = cmx * 8;   // cpitch: bytes per chroma row of the band's tile
  const int cmode = (int)HUFF_UNIFORM(G.bpm) == 6 ? 0 : (int)HUFF_UNIFORM(G.bpm) == 3 ? 1 : 2;   // 4:2:0 / 4:4:4 / grayscale
  const int band_c0 = kColor ? (int)HUFF_UNIFORM(G.band_c0) : 0, band_c1 = kColor ? (int)HUFF_UNIFORM(G.band_c1) : 0;
  const int nbands = kColor ? (int)HUFF_UNIFORM(G.bands) : 0;
  const int cwidth = kColor ? (int)HUFF_UNIFORM(G.width) : 0, cheight = kColor ? (int)HUFF_UNIFORM(G.height) : 0;
  const int cdw = (cwidth + 1) >> 1, cdh = (cheight + 1) >> 1;
  const int crgb_pitch = kColor ? (int)HUFF_UNIFORM(G.rgb_pitch) : 0;
  GlobalBytes *rgb = kColor ? (GlobalBytes *)d.rgb : nullptr;
  GlobalBytes *seams = kColor ? (GlobalBytes *)(d.scratch + lay.seams) : nullptr;
  const int roi_cols = (int)HUFF_UNIFORM(G.roi_cols);   // > 0: m0 / M count MCUs of the region's rectangle, not of the frame
  if (roi_cols ? m0 >= G.roi_mcus : (long long)m0 * bpm >= G.last_ordinal) return;  // nothing needed here (uniform)
  const int M = min(mpw, (roi_cols ? G.roi_mcus : G.total_mcus) - m0);
  const int n0 = G.n0, n1 = bpm - n0;
  const int tasks0 = (M * n0 + 63) >> 6, tasks1 = (M * n1 + 63) >> 6;
  GlobalWords *words = (GlobalWords *)CleanStream(d, lay);
  const bool indexed = d.index != nullptr;   // (uniform)
  const GlobalU32 *blk_pos = (const GlobalU32 *)(d.scratch + lay.blk_pos);
  const GlobalPosDc *blk_pd = (const GlobalPosDc *)(d.scratch + lay.blk_pos);
  const GlobalI32 *blk_dc = (const GlobalI32 *)(d.scratch + lay.blk_dc);
  const GlobalU16 *blk_seg = (const GlobalU16 *)(d.scratch + lay.blk_seg);
  const GlobalI32 *segs_i = (const GlobalI32 *)(d.scratch + lay.segs);  // SegRec fields as ints (global loads, not flat)
  int16_t *wcoef = &coef[wave][0];
  const LaneCoef mycoef{wcoef + lane};
  // the waves share the tasks of each class evenly (a luma task takes 2-3 times as long as a chroma task)
  const int my0 = wave < tasks0 ? (tasks0 - wave + kBlockWaves - 1) / kBlockWaves : 0;
  const int rwave = kBlockWaves - 1 - wave;  // the chroma tasks are dealt from the other end
  const int my1 = rwave < tasks1 ? (tasks1 - rwave + kBlockWaves - 1) / kBlockWaves : 0;
  // One task ahead: what a lane needs to know about its block of the NEXT task - indices and output address, then
  // (stage A) its entries in the per-block arrays, then (stage B) the first three dwords of its bit stream - is
  // fetched while the wave transforms the blocks of the current task, so that a decode starts without a memory
  // round trip (three dependent ones otherwise: block arrays, segment totals, stream).
  struct Prepared {
    uint64_t info;
    int ordinal, comp, acs, dc, seg, seg0;
    int bx, by;   // the block's position in its component, in blocks
    uint32_t pos;
    bool needed;
    BitWindow win;
  };
  // task t of this wave -> (class, index inside the class).  Fused colour output: the chroma tasks come first - the luma
  // lanes read the band's chroma samples from LDS behind one workgroup barrier
  const int nfirst = kColor ? my1 : my0;
  auto stage_a = [&](int t) -> Prepared {
    Prepared p;
    p.needed = false; p.info = 0; p.ordinal = 0; p.comp = 0; p.acs = 0; p.dc = 0; p.seg = 0; p.seg0 = 0; p.pos = 0;
    p.win = BitWindow{0, 0, 0, 0, 0};
    p.bx = 0; p.by = 0;
    if (t >= my0 + my1) return p;
    const bool cls = kColor ? t < nfirst : t >= nfirst;
    const int tt = t < nfirst ? t : t - nfirst;
    const int ncls = cls ? n1 : n0;
    const int j = (cls ? tt * kBlockWaves + rwave : tt * kBlockWaves + wave) * 64 + lane;
    int mi = j / ncls, k = G.klist[(cls ? n0 : 0) + (j - mi * ncls)];
    if (kColor && !cls) {
      // luma blocks of the band in raster order of BLOCKS: the lanes of a wave own x-adjacent blocks of one block row, so
      // their 24-byte RGB segments of an output row are one contiguous run (MCU order would spread a store instruction
      // over two pixel rows and leave gaps between the pairs)
      // (luma sampling 2 x 2 with 4:2:0 - the MCU's luma blocks are k = 2 v + h -, 1 x 1 otherwise: k = 0)
      const int ls = cmode == 0 ? 1 : 0;   // log2 of the luma sampling factor (both directions)
      const int bw = cmx << ls, brow = j / bw, bcol = j - brow * bw;
      mi = (brow >> ls) * cmx + (bcol >> ls);
      k = ls ? ((brow & 1) << 1) | (bcol & 1) : 0;
    }
    int my, mx;
    if (roi_cols) {   // the (m0 + mi)-th MCU of the region's rectangle
      const int ry = (m0 + mi) / roi_cols;
      my = G.roi_y0 + ry;
      mx = G.roi_x0 + (m0 + mi) - ry * roi_cols;
    } else {
      my = (m0 + mi) / G.mcus_x;
      mx = (m0 + mi) - my * G.mcus_x;
    }
    const int mcu = my * G.mcus_x + mx, ordinal = mcu * bpm + k;
    const int bx = mx * G.hs[k] + G.ho[k], by = my * G.vs[k] + G.vo[k];
    bool needed = mi < M && ordinal < G.last_ordinal && ordinal + 1 < G.total_starts;
    if (G.use_rect) needed = needed && bx >= G.rect[k][0] && by >= G.rect[k][1] && bx < G.rect[k][2] && by < G.rect[k][3];
    const int comp = G.comp[k];
    const uint64_t dst = G.fused ? (uint64_t)(uintptr_t)(G.plane[k] + (size_t)(by * 8) * G.pitch[k] + (size_t)bx * 8)
                                 : (uint64_t)(uintptr_t)(G.base[k] + ((size_t)my * (size_t)G.sy[k] + (size_t)(mx * G.sx[k])));
    p.info = (dst & ((1ull << 48) - 1)) | ((uint64_t)comp << 48) | (needed ? kInfoNeeded : 0ull);
    p.needed = needed; p.ordinal = ordinal; p.comp = comp; p.acs = G.acs[k];
    p.bx = bx; p.by = by;
    if (needed && indexed) {   // IndexedSyncKernel's record: position and absolute level (mod 2^16) in one load
      const u32x2p pd = blk_pd[ordinal];
      // (an index entry may come from a FILE - an indexed container - and say anything: a block its slices never reached
      // keeps whatever the scratch held; no position leaves the stream)
      p.pos = min(pd.x, G.total_bits);
      p.dc = (int)pd.y;
    } else if (needed) {
      p.pos = blk_pos[ordinal];
      p.dc = blk_dc[ordinal];
      p.seg = blk_seg[ordinal];
      if (G.interval_blocks) {
        // restart intervals: the DC prediction starts at zero in every interval.  DcKernel's sums run through the whole
        // stream; what lies in front of the interval is the sum at the component's last block of the MCU before it
        const int first = ordinal / G.interval_blocks * G.interval_blocks;
        if (first > 0) {
          const int q = first - bpm + G.klast[comp];
          p.dc -= blk_dc[q];
          p.seg0 = blk_seg[q];
        }
      }
    }
    return p;
  };
  auto stage_b = [&](Prepared &p) {
    if (!p.needed) return;
    p.win = OpenWindow(words, p.pos);
    // DC level: the block's level inside its segment + the differences of all the segments before (an image is a
    // handful of segments; only a stream of many megabytes makes this loop long)
    for (int s = p.seg0; s < p.seg; s++) p.dc += segs_i[s * kSegRecInts + kSegRecDcTotal + p.comp];
  };
  // The lane's own block, start to end in its registers: dequantisation, the eight column passes, the eight row passes
  // of the islow IDCT, eight 8-byte row stores.  (Round 2 spread a block over 8 lanes with an int32 transpose through
  // LDS between the passes: the same number of butterflies per wave, plus 2 LDS round trips and 16 wave-wide fences.)
  // Two halves so that the next task's stream words can be requested in between.
  int32_t ws[8][8];
  auto idct_columns = [&](const Prepared &p) {
    if (!p.needed || !G.fused) return;
    const uint16_t *q = &quant[p.comp][0];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      int32_t in[8], o[8];
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++)
        in[r8] = __mul24((int32_t)wcoef[kScanIndexOfColMajorHost(c * 8 + r8) * 64 + lane], (int32_t)q[c * 8 + r8]);
      Butterfly8(in, o);
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++) ws[r8][c] = Descale(o[r8], CONST_BITS - PASS1_BITS);
    }
  };
  auto idct_rows = [&](const Prepared &p) {
    if (!p.needed) return;
    const uint64_t dst = p.info & ((1ull << 48) - 1);
    if (!G.fused) {   // the block as one 128-byte line of column-major coefficients
#pragma unroll
      for (int c = 0; c < 8; c++) {
        uint32_t w[4];
#pragma unroll
        for (int qd = 0; qd < 4; qd++)
          w[qd] = (uint32_t)(uint16_t)wcoef[kScanIndexOfColMajorHost(c * 8 + 2 * qd) * 64 + lane] |
                  ((uint32_t)(uint16_t)wcoef[kScanIndexOfColMajorHost(c * 8 + 2 * qd + 1) * 64 + lane] << 16);
        ((GlobalQuad *)(uintptr_t)dst)[c] = u32x4{w[0], w[1], w[2], w[3]};
      }
      return;
    }
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    using GlobalPair = u32x2_t __attribute__((address_space(1)));
    constexpr int S = CONST_BITS + PASS1_BITS + 3;
    if (!kColor) {
      const int pitch = G.comp_pitch[p.comp];
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++) {
        int32_t o[8];
        Butterfly8(ws[r8], o);
        const uint32_t lo32 = RangeLimit(Descale(o[0], S)) | (RangeLimit(Descale(o[1], S)) << 8) |
                              (RangeLimit(Descale(o[2], S)) << 16) | (RangeLimit(Descale(o[3], S)) << 24);
        const uint32_t hi32 = RangeLimit(Descale(o[4], S)) | (RangeLimit(Descale(o[5], S)) << 8) |
                              (RangeLimit(Descale(o[6], S)) << 16) | (RangeLimit(Descale(o[7], S)) << 24);
        *(GlobalPair *)((GlobalBytes *)(uintptr_t)dst + (size_t)r8 * pitch) = u32x2_t{lo32, hi32};
      }
      return;
    }
    if (p.comp != 0) {
      // ---- a chroma block: its 8 x 8 samples go to the band's tile in LDS; the band's first / last chroma row also to
      // the seam strips (the neighbouring bands' edge rows interpolate towards them)
      uint8_t *tile = ctile + (p.comp - 1) * kChromaTile + (p.by * 8 - band_c0) * cpitch + p.bx * 8;
      const bool top = cmode == 0 && p.by * 8 == band_c0 && band > 0;
      const bool bottom = cmode == 0 && p.by * 8 + 8 == band_c1 && band + 1 < nbands;
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++) {
        int32_t o[8];
        Butterfly8(ws[r8], o);
        const uint32_t lo32 = RangeLimit(Descale(o[0], S)) | (RangeLimit(Descale(o[1], S)) << 8) |
                              (RangeLimit(Descale(o[2], S)) << 16) | (RangeLimit(Descale(o[3], S)) << 24);
        const uint32_t hi32 = RangeLimit(Descale(o[4], S)) | (RangeLimit(Descale(o[5], S)) << 8) |
                              (RangeLimit(Descale(o[6], S)) << 16) | (RangeLimit(Descale(o[7], S)) << 24);
        *reinterpret_cast<uint2 *>(tile + r8 * cpitch) = make_uint2(lo32, hi32);
        if ((r8 == 0 && top) || (r8 == 7 && bottom))
          *(GlobalPair *)(seams + SeamStrip(band, r8 == 7, cmx) + (size_t)(8 + 8 * p.comp) * cmx + p.bx * 8) = u32x2_t{lo32, hi32};
      }
      return;
    }
    if (cmode != 0) {
      // ---- a luma block of a 4:4:4 stream (its chroma samples are the tile's block at the same place) or of a grayscale
      // stream (R = G = B = the sample)
      const int x0 = p.bx * 8, npx = min(8, cwidth - x0);
      const uint8_t *tile = ctile + (p.by * 8 - band_c0) * cpitch + p.bx * 8;
#pragma unroll
      for (int r8 = 0; r8 < 8; r8++) {
        const int y = p.by * 8 + r8;
        int32_t o[8];
        Butterfly8(ws[r8], o);
        int yy[8];
#pragma unroll
        for (int i = 0; i < 8; i++) yy[i] = (int)RangeLimit(Descale(o[i], S));
        if (y >= cheight || npx <= 0) continue;
        uint32_t px[24];
        if (cmode == 1) {
          int up[2][8];
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const uint2 v = *reinterpret_cast<const uint2 *>(tile + c * kChromaTile + r8 * cpitch);
#pragma unroll
            for (int i = 0; i < 4; i++) { up[c][i] = (int)((v.x >> (8 * i)) & 255); up[c][4 + i] = (int)((v.y >> (8 * i)) & 255); }
          }
          YccToRgb8(yy, up, px);
        } else {
#pragma unroll
          for (int i = 0; i < 8; i++) px[3 * i] = px[3 * i + 1] = px[3 * i + 2] = (uint32_t)yy[i];
        }
        StoreRgb8((GOutBytes *)(rgb + (size_t)y * crgb_pitch + (size_t)x0 * 3), px, npx);
      }
      return;
    }
    // ---- a luma block of a 4:2:0 stream: fancy h2v2 upsampling of the chroma around it (rows 4 by - 1 .. 4 by + 4 of the
    // tile, samples 4 bx - 1 .. 4 bx + 4) and YCbCr -> RGB, the arithmetic of jpeg_color.hip's ColorRows420; eight rows of
    // 24 bytes
    const int k0 = p.bx * 4;
    const int ka = max(k0 - 4, 0), kc = min(k0 + 4, cpitch - 4);
    uint32_t ca[2][6], cb[2][6], cc[2][6];   // [component][chroma row 4 by - 1 + j]: dwords left of / at / right of k0
    const int r_above = ClampI(p.by * 4 - 1, 0, cdh - 1), r_below = ClampI(p.by * 4 + 4, 0, cdh - 1);
    const bool seam_top = r_above < band_c0, seam_bottom = r_below >= band_c1;   // rows of a neighbouring band
#pragma unroll
    for (int j = 0; j < 6; j++) {
      int r = ClampI(p.by * 4 - 1 + j, 0, cdh - 1) - band_c0;
      r = ClampI(r, 0, band_c1 - band_c0 - 1);   // (seam rows: read something valid, the row is not produced here)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const uint8_t *row = ctile + c * kChromaTile + r * cpitch;
        ca[c][j] = *reinterpret_cast<const uint32_t *>(row + ka);
        cb[c][j] = *reinterpret_cast<const uint32_t *>(row + k0);
        cc[c][j] = *reinterpret_cast<const uint32_t *>(row + kc);
      }
    }
    const int x0 = p.bx * 8, npx = min(8, cwidth - x0);
#pragma unroll
    for (int r8 = 0; r8 < 8; r8++) {
      const int y = p.by * 8 + r8;
      int32_t o[8];
      Butterfly8(ws[r8], o);
      int yy[8];
#pragma unroll
      for (int i = 0; i < 8; i++) yy[i] = (int)RangeLimit(Descale(o[i], S));
      if (y >= cheight || npx <= 0) continue;
      if ((r8 == 0 && seam_top) || (r8 == 7 && seam_bottom)) {   // the seam launch finishes this row
        const uint32_t lo32 = yy[0] | (yy[1] << 8) | (yy[2] << 16) | (yy[3] << 24);
        const uint32_t hi32 = yy[4] | (yy[5] << 8) | (yy[6] << 16) | (yy[7] << 24);
        *(GlobalPair *)(seams + SeamStrip(band, r8 == 7, cmx) + p.bx * 8) = u32x2_t{lo32, hi32};
        continue;
      }
      // output row y: the nearer chroma row is 4 by + (r8 >> 1) (index 1 + (r8 >> 1) above), the further one the row
      // above it (r8 even) or below it (r8 odd)
      const int near = 1 + (r8 >> 1), far = (r8 & 1) ? near + 1 : near - 1;
      int up[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        int n7[7], f7[7], sv[7];
        Chroma7(ca[c][near], cb[c][near], cc[c][near], k0 == 0, n7);
        Chroma7(ca[c][far], cb[c][far], cc[c][far], k0 == 0, f7);
#pragma unroll
        for (int i = 0; i < 7; i++) sv[i] = n7[i] * 3 + f7[i];
        ClampRight7(sv, k0, cdw);
        TriangleX8<4, 8, 7>(sv, false, up[c]);
      }
      uint32_t px[24];
      YccToRgb8(yy, up, px);
      StoreRgb8((GOutBytes *)(rgb + (size_t)y * crgb_pitch + (size_t)x0 * 3), px, npx);
    }
  };
  Prepared cur = stage_a(0);
  stage_b(cur);
  for (int t = 0; t < my0 + my1; t++) {
    {
      uint4 *z = reinterpret_cast<uint4 *>(wcoef);   // the 64 coefficient rows (the 65th only swallows)
      for (int i = lane; i < 64 * 64 * 2 / 16; i += 64) z[i] = make_uint4(0, 0, 0, 0);
    }
    WaveSync();
    if (cur.needed) {
      mycoef[0] = (int16_t)cur.dc;
      DecodeBlockAc(T, words, cur.win, (uint32_t)cur.acs, mycoef);
    }
    WaveSync();
    Prepared nxt = stage_a(t + 1);
    idct_columns(cur);
    stage_b(nxt);
    // fused colour output: the wave's first luma task reads the chroma tile every wave has written by now (each wave
    // passes exactly one barrier: here, or behind the loop when it has no luma task)
    if (kColor && t == nfirst && my0 > 0) __syncthreads();
    idct_rows(cur);
    WaveSync();   // the next round's zero-fill must not overtake this round's reads
    cur = nxt;
  }
  if (kColor && my0 == 0) __syncthreads();
  STAMP_MAX(2, 2);
#ifdef DALIAMD_EXP_STAMPS
  if (tid == 0 && blockIdx.x < 8192) g_stamps[2][blockIdx.x * 16 + 12] = (unsigned long long)(tasks0 * 1000 + tasks1);
#endif
}

// Fused colour output, the seams: output rows 16 R b - 1 and 16 R b (R MCU rows per band) interpolate between the last
// chroma row of band b - 1 and the first one of band b.  One workgroup per band (the first has no seam above it), a
// thread per 8 pixels of both rows; luma and chroma come from the strips the block kernel left in the scratch.
constexpr int kSeamThreads = 256;
__global__ __launch_bounds__(kSeamThreads) void SeamKernel(const daliamdJpegHuffDesc *__restrict__ descs, int n, int nwg) {
  const int wg = XcdRemap(blockIdx.x, nwg);
  if (wg < 0) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk_wg_start <= wg) lo = mid; else hi = mid - 1;
  }
  const daliamdJpegHuffDesc &d = descs[lo];
  const int band = wg - d.blk_wg_start;
  if (d.rgb == nullptr || band == 0 || d.blocks_per_mcu != 6) return;   // (4:4:4 / grayscale: nothing is interpolated)
  const int rows = ColorBandRows(d.mcus_x);
  const int y_up = 16 * rows * band - 1;   // last pixel row of the band above; y_up + 1: first row of this band
  if (y_up + 1 >= d.height) return;
  const ScratchLayout lay = LayoutOf(d);
  GBytes *seams = (GBytes *)(d.scratch + lay.seams);
  const int mx = d.mcus_x, cpitch = mx * 8, dw = (d.width + 1) >> 1;
  GBytes *up = seams + SeamStrip(band - 1, 1, mx), *dn = seams + SeamStrip(band, 0, mx);
  GOutBytes *rgb = (GOutBytes *)d.rgb;
  for (int x0 = threadIdx.x * 8; x0 < d.width; x0 += kSeamThreads * 8) {
    const int k0 = x0 >> 1, npx = min(8, d.width - x0);
    int cu[2][7], cd[2][7];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      LoadSamples7(up + (16 + 8 * c) * mx, cpitch, k0, cu[c]);
      LoadSamples7(dn + (16 + 8 * c) * mx, cpitch, k0, cd[c]);
    }
    const u32x2 yu = *reinterpret_cast<GPair *>(up + x0), yd = *reinterpret_cast<GPair *>(dn + x0);
#pragma unroll
    for (int side = 0; side < 2; side++) {   // 0: the row above the seam (near = up, far = down); 1: the row below it
      int uv[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        int sv[7];
#pragma unroll
        for (int i = 0; i < 7; i++) sv[i] = side == 0 ? cu[c][i] * 3 + cd[c][i] : cd[c][i] * 3 + cu[c][i];
        ClampRight7(sv, k0, dw);
        TriangleX8<4, 8, 7>(sv, false, uv[c]);
      }
      const u32x2 yv = side == 0 ? yu : yd;
      int yy[8];
#pragma unroll
      for (int i = 0; i < 8; i++) yy[i] = (int)(((i < 4 ? yv.x : yv.y) >> (8 * (i & 3))) & 255);
      uint32_t px[24];
      YccToRgb8(yy, uv, px);
      StoreRgb8(rgb + (size_t)(y_up + side) * d.rgb_pitch + (size_t)x0 * 3, px, npx);
    }
  }
}

}  // namespace daliamd

extern "C" {

daliamdResult_t daliamdJpegHuffmanScratchBytes(int ecs_len, int total_blocks, size_t *bytes) {
  return daliamdJpegHuffmanScratchBytesRestart(ecs_len, total_blocks, 0, bytes);
}

daliamdResult_t daliamdJpegHuffmanScratchBytesRestart(int ecs_len, int total_blocks, int num_intervals, size_t *bytes) {
  DALIAMD_REQUIRE(ecs_len >= 0 && total_blocks >= 0 && num_intervals >= 0 && bytes, DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegHuffmanScratchBytes: invalid argument");
  *bytes = daliamd::MakeLayout(ecs_len, daliamd::NumTiles(15, ecs_len), daliamd::NumSegments(ecs_len), total_blocks,
                               num_intervals).total;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegHuffmanTablesBytes(size_t *bytes) {
  DALIAMD_REQUIRE(bytes, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanTablesBytes: NULL argument");
  *bytes = sizeof(daliamd::HuffTables) + sizeof(daliamd::SyncTables);
  return DALIAMD_SUCCESS;
}

// The code tables of one stream on the HOST: the functions BuildTables runs on the device (huff_core.h compiles for both),
// entry by entry - the same bytes.
daliamdResult_t daliamdJpegHuffmanTablesBuild(const daliamdJpegHuffDesc *d, void *out_host) {
  using namespace daliamd;
  DALIAMD_REQUIRE(d && out_host && d->blocks_per_mcu >= 1 && d->blocks_per_mcu <= DALIAMD_JPEG_MAX_BLOCKS_PER_MCU,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanTablesBuild: invalid argument");
  memset(out_host, 0, sizeof(HuffTables) + sizeof(SyncTables));
  HuffTables &L = *static_cast<HuffTables *>(out_host);
  SyncTables &S = *reinterpret_cast<SyncTables *>(static_cast<uint8_t *>(out_host) + sizeof(HuffTables));
  memcpy(L.vals, d->vals, sizeof(L.vals));
  for (int k = 0; k < d->blocks_per_mcu; k++) {
    const int comp = d->comp_of_block[k];
    DALIAMD_REQUIRE(comp < 3, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanTablesBuild: block %d refers to component %d", k, comp);
    L.dc_mask |= (uint32_t)(d->dc_sel[comp] & 1) << k;
    L.ac_mask |= (uint32_t)(d->ac_sel[comp] & 1) << k;
  }
  L.bpm = d->blocks_per_mcu;
  for (int t = 0; t < 4; t++) CodeRanges(d->bits[t], L.maxcode[t], L.valoff[t], &L.l2_first[t], &L.l2_size[t]);
  for (int t = 0; t < 4; t++) {
    for (int w = 0; w < (1 << kFastBits); w++) L.fast[t][w] = FastEntry(L, t, w);
    for (int j = 0; j < kL2Entries; j++) L.l2[t][j] = L2Entry(L, t, j);
  }
  for (int t = 0; t < 4; t++)
    for (int w = 0; w < (1 << kFastBits); w++) S.t32[t][w] = SyncEntry(L, t, w);
  memcpy(S.l2, L.l2, sizeof(S.l2));
  memcpy(S.l2_first, L.l2_first, sizeof(S.l2_first));
  memcpy(S.l2_size, L.l2_size, sizeof(S.l2_size));
  memcpy(S.maxcode, L.maxcode, sizeof(S.maxcode));
  memcpy(S.valoff, L.valoff, sizeof(S.valoff));
  memcpy(S.vals, L.vals, sizeof(S.vals));
  S.dc_mask = L.dc_mask; S.ac_mask = L.ac_mask; S.bpm = L.bpm; S.reserved = 0;
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegHuffmanIndexBytes(int ecs_len, size_t *bytes) {
  DALIAMD_REQUIRE(ecs_len >= 0 && bytes, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanIndexBytes: invalid argument");
  *bytes = daliamd::IndexBytes(ecs_len);
  return DALIAMD_SUCCESS;
}

// The index entry of a stream, built on the HOST (round 6: tools/jpeg2idx.py writes it next to the file, the mixed decoder
// uploads it with - instead of - the entropy-coded segment, so that a COLD process and epoch 1 from files decode from the
// index too).  The same per-lane functions the device runs (huff_core.h compiles for both), in sequence: where the kernels'
// relaxation CONVERGES to the state in front of every slice, a sequential walk simply has it.  The bytes the device's
// IndexBuildKernel defines - header, clean stream with its all-ones padding, entries and sentinel - are the same bytes
// (tests/test_gpu_jpeg_index.py holds the two to each other); what lies between them is zero here.
// `d`: bits / vals / comp_of_block / dc_sel / ac_sel / blocks_per_mcu / total_blocks / ecs_len as for a decode, `ecs` a HOST
// pointer to the byte-stuffed segment (ecs_len may run past its end marker).  *status: 0, or 2 / 3 like a decode.
daliamdResult_t daliamdJpegHuffmanIndexBuildHost(const daliamdJpegHuffDesc *d, void *index_out_host, int32_t *status) {
  using namespace daliamd;
  DALIAMD_REQUIRE(d && index_out_host && status && d->ecs && d->ecs_len >= 0 && d->restart_interval == 0 &&
                      d->blocks_per_mcu >= 1 && d->blocks_per_mcu <= DALIAMD_JPEG_MAX_BLOCKS_PER_MCU && d->total_blocks >= 0,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanIndexBuildHost: invalid argument (streams with restart intervals have no index)");
  DALIAMD_REQUIRE((long long)d->total_blocks + 128 < (1ll << 26), DALIAMD_ERROR_UNSUPPORTED,
                  "daliamdJpegHuffmanIndexBuildHost: %d blocks do not fit the 26-bit ordinals of an index entry", d->total_blocks);
  *status = 0;
  const int ecs_len = d->ecs_len;
  uint8_t *out = static_cast<uint8_t *>(index_out_host);
  memset(out, 0, IndexBytes(ecs_len));
  // ---- un-stuffing (LoadChunk's rules, byte by byte): the zero behind a 0xFF goes, fill bytes go, the segment ends at the
  // first marker; a last 0xFF without a successor stays
  uint8_t *clean = out + IndexCleanOffset();
  int clean_len = 0;
  for (int i = 0; i < ecs_len;) {
    const uint8_t b = d->ecs[i];
    if (b == 0xFF && i + 1 < ecs_len) {
      const uint8_t nx = d->ecs[i + 1];
      if (nx == 0x00) { clean[clean_len++] = 0xFF; i += 2; continue; }
      if (nx == 0xFF) { i += 1; continue; }
      if (nx >= 0xD0 && nx <= 0xD7) { *status = 3; return DALIAMD_SUCCESS; }   // RSTn in a stream that announces none
      break;
    }
    clean[clean_len++] = b;
    i++;
  }
  memset(clean + clean_len, 0xFF, kCleanPadBytes);   // (what UnstuffScatterKernel leaves behind the stream)
  // ---- code tables
  std::vector<uint8_t> tab(sizeof(HuffTables) + sizeof(SyncTables));
  daliamdResult_t rc = daliamdJpegHuffmanTablesBuild(d, tab.data());
  if (rc != DALIAMD_SUCCESS) return rc;
  const HuffTables &D = *reinterpret_cast<const HuffTables *>(tab.data());
  const SyncTables &L = *reinterpret_cast<const SyncTables *>(tab.data() + sizeof(HuffTables));
  const uint32_t *words = reinterpret_cast<const uint32_t *>(clean);
  // ---- every slice from the state its predecessor ended in: entry states, block starts found, DC level of every block
  const uint32_t total_bits = (uint32_t)clean_len * 8u;
  const int num_slices = (clean_len + kSliceBytes - 1) / kSliceBytes, cap = IndexSliceCap(ecs_len);
  std::vector<DecodeState> in((size_t)num_slices);
  std::vector<int> ord((size_t)num_slices + 1, 0);
  std::vector<int32_t> level_of;           // absolute DC level of block q (blk_dc + the segments' totals on the device)
  level_of.reserve((size_t)d->total_blocks + 64);
  int level[3] = {0, 0, 0};
  DecodeState st{0, 0, 0};
  for (int s = 0; s < num_slices; s++) {
    const uint32_t begin = (uint32_t)s * (kSliceBytes * 8u);
    const uint32_t end = begin + kSliceBytes * 8u < total_bits ? begin + kSliceBytes * 8u : total_bits;
    in[s] = st;
    if (st.pos < end)
      IndexedDecodeSlice(L, D, words, st, end, [&](uint32_t c, uint32_t, int diff) {
        const int comp = d->comp_of_block[c] < 3 ? d->comp_of_block[c] : 0;
        level[comp] += diff;
        level_of.push_back(level[comp]);
      });
    ord[s + 1] = ord[s] + SyncDecodeRange(L, words, st, end, [](int, int, bool) {});
  }
  const int total_starts = ord[num_slices];
  if (total_starts - 1 < d->total_blocks) *status = 2;   // (PropagateKernel: every block must start AND end inside the segment)
  int32_t *h = reinterpret_cast<int32_t *>(out);
  h[0] = clean_len; h[1] = total_starts; h[2] = num_slices;
  uint32_t *entries = reinterpret_cast<uint32_t *>(out + IndexEntriesOffset(ecs_len));
  auto put = [&](long long slice, const SliceIndex &e) {
    entries[3 * slice] = e.w0; entries[3 * slice + 1] = e.w1; entries[3 * slice + 2] = e.w2;
  };
  const uint32_t zero3[3] = {0, 0, 0};
  const SliceIndex none = PackSliceIndex((uint32_t)total_starts, DecodeState{0, 0, 0}, zero3);
  for (int s = num_slices; s <= cap; s++) put(s, none);
  for (int s = 0; s < num_slices; s++) {   // (IndexBuildKernel, lane by lane)
    DecodeState e = in[s];
    const int first_block = ord[s] - (e.z == 0 && e.pos != 0 ? 1 : 0);
    uint32_t dc[3] = {0, 0, 0};
    uint32_t found = 0;
    for (int j = 1; j <= d->blocks_per_mcu && found != 7u; j++) {
      const int q = first_block - j;
      if (q < 0) break;
      const int comp = d->comp_of_block[q % d->blocks_per_mcu];
      if (comp > 2 || (found >> comp) & 1u) continue;
      found |= 1u << comp;
      if (q >= d->total_blocks || q + 1 >= total_starts || q >= (int)level_of.size()) continue;
      dc[comp] = (uint32_t)level_of[q] & 0xFFFFu;
    }
    e.pos -= (uint32_t)s * (kSliceBytes * 8u);
    put(s, PackSliceIndex((uint32_t)first_block, e, dc));
  }
  return DALIAMD_SUCCESS;
}

int daliamdJpegHuffmanColorFusable(const daliamdJpegHuffDesc *d) {
  if (!d || d->mcus_x < 1 || d->mcus_x > daliamd::kColorBandMcus) return 0;
  for (int c = 0; c < 3; c++)
    for (int j = 0; j < 4; j++)
      if (d->rect[c][j]) return 0;
  if (d->blocks_per_mcu == 6) {          // YCbCr 4:2:0
    static const uint8_t comp[6] = {0, 0, 0, 0, 1, 2}, ho[6] = {0, 1, 0, 1, 0, 0}, vo[6] = {0, 0, 1, 1, 0, 0};
    if (memcmp(d->comp_of_block, comp, 6) || memcmp(d->h_of_block, ho, 6) || memcmp(d->v_of_block, vo, 6)) return 0;
    return d->h_samp[0] == 2 && d->v_samp[0] == 2 && d->h_samp[1] == 1 && d->v_samp[1] == 1 && d->h_samp[2] == 1 && d->v_samp[2] == 1;
  }
  if (d->blocks_per_mcu == 3) {          // YCbCr 4:4:4
    static const uint8_t comp[3] = {0, 1, 2}, zero[3] = {0, 0, 0};
    if (memcmp(d->comp_of_block, comp, 3) || memcmp(d->h_of_block, zero, 3) || memcmp(d->v_of_block, zero, 3)) return 0;
    return d->h_samp[0] == d->h_samp[1] && d->h_samp[1] == d->h_samp[2] && d->v_samp[0] == d->v_samp[1] && d->v_samp[1] == d->v_samp[2];
  }
  if (d->blocks_per_mcu == 1)            // grayscale
    return d->comp_of_block[0] == 0 && d->h_of_block[0] == 0 && d->v_of_block[0] == 0;
  return 0;
}

daliamdResult_t daliamdJpegHuffmanSetup(daliamdJpegHuffDesc *descs_host, int n, int *num_tiles, int *num_segments,
                                        int *num_block_workgroups) {
  int kinds = 0;
  daliamdResult_t r = daliamdJpegHuffmanSetupColor(descs_host, n, num_tiles, num_segments, num_block_workgroups, &kinds);
  if (r != DALIAMD_SUCCESS) return r;
  DALIAMD_REQUIRE(!(kinds & 2), DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegHuffmanSetup: a stream asks for the fused colour output (rgb != NULL): use "
                  "daliamdJpegHuffmanSetupColor / daliamdJpegHuffmanRunColor");
  DALIAMD_REQUIRE(!(kinds & (DALIAMD_JPEG_HUFFMAN_INDEXED | DALIAMD_JPEG_HUFFMAN_BUILD_INDEX)), DALIAMD_ERROR_INVALID_ARGUMENT,
                  "daliamdJpegHuffmanSetup: a stream brings or asks for an index: use daliamdJpegHuffmanSetupColor / "
                  "daliamdJpegHuffmanRunColor");
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegHuffmanSetupColor(daliamdJpegHuffDesc *descs_host, int n, int *num_tiles, int *num_segments,
                                             int *num_block_workgroups, int *block_kernels) {
  DALIAMD_REQUIRE(n >= 0 && (n == 0 || descs_host) && num_tiles && num_segments && num_block_workgroups && block_kernels,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanSetup: invalid argument");
  int tiles = 0, segs = 0, bwgs = 0, kinds = 0;
  std::vector<int> owners;   // one stream per distinct table set seen so far
  auto same_tables = [](const daliamdJpegHuffDesc &a, const daliamdJpegHuffDesc &b) {
    return a.blocks_per_mcu == b.blocks_per_mcu && !memcmp(a.bits, b.bits, sizeof(a.bits)) && !memcmp(a.vals, b.vals, sizeof(a.vals)) &&
           !memcmp(a.comp_of_block, b.comp_of_block, sizeof(a.comp_of_block)) && !memcmp(a.dc_sel, b.dc_sel, sizeof(a.dc_sel)) &&
           !memcmp(a.ac_sel, b.ac_sel, sizeof(a.ac_sel));
  };
  for (int i = 0; i < n; i++) {
    daliamdJpegHuffDesc &d = descs_host[i];
    d.table_owner = i;
    if (!d.tables) {   // (a stream that brings its tables owns nothing and needs no owner)
      for (int o : owners)
        if (same_tables(descs_host[o], d)) { d.table_owner = o; break; }
      if (d.table_owner == i && owners.size() < 64) owners.push_back(i);
      kinds |= DALIAMD_JPEG_HUFFMAN_BUILD_TABLES;
    } else {
      DALIAMD_REQUIRE((reinterpret_cast<uintptr_t>(d.tables) & 15) == 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                      "daliamdJpegHuffmanSetup: sample %d: tables must be 16-byte aligned", i);
    }
    DALIAMD_REQUIRE((d.ecs || d.index) && d.scratch && d.status && d.ecs_len >= 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegHuffmanSetup: sample %d: NULL buffer or negative length", i);
    if (d.index || d.index_out) {
      DALIAMD_REQUIRE(!(d.index && d.index_out), DALIAMD_ERROR_INVALID_ARGUMENT,
                      "daliamdJpegHuffmanSetup: sample %d: index and index_out are exclusive", i);
      DALIAMD_REQUIRE(d.restart_interval == 0 && d.total_blocks + 128 < (1 << 26), DALIAMD_ERROR_UNSUPPORTED,
                      "daliamdJpegHuffmanSetup: sample %d: no index for streams with restart intervals or %d blocks and more", i,
                      (1 << 26) - 128);
      DALIAMD_REQUIRE(((reinterpret_cast<uintptr_t>(d.index) | reinterpret_cast<uintptr_t>(d.index_out)) & 63) == 0,
                      DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanSetup: sample %d: index entries must be 64-byte aligned", i);
    }
    kinds |= d.index ? DALIAMD_JPEG_HUFFMAN_INDEXED : DALIAMD_JPEG_HUFFMAN_PARSED;
    if (d.index_out) kinds |= DALIAMD_JPEG_HUFFMAN_BUILD_INDEX;
    DALIAMD_REQUIRE((reinterpret_cast<uintptr_t>(d.scratch) & 15) == 0, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegHuffmanSetup: sample %d: scratch must be 16-byte aligned", i);
    DALIAMD_REQUIRE(d.blocks_per_mcu >= 1 && d.blocks_per_mcu <= DALIAMD_JPEG_MAX_BLOCKS_PER_MCU && d.mcus_x >= 1 &&
                        d.total_blocks >= 1 && d.total_blocks % d.blocks_per_mcu == 0,
                    DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanSetup: sample %d: bad MCU geometry", i);
    const bool color = d.rgb != nullptr;
    kinds |= color ? 2 : 1;
    if (color) {
      DALIAMD_REQUIRE(daliamdJpegHuffmanColorFusable(&d), DALIAMD_ERROR_UNSUPPORTED,
                      "daliamdJpegHuffmanSetup: sample %d: the fused colour output needs a 4:2:0 / 4:4:4 / one-component stream "
                      "in the usual block order, at most %d MCUs wide, without a block rectangle", i, daliamd::kColorBandMcus);
      const int mcu_px = d.blocks_per_mcu == 6 ? 16 : 8;
      DALIAMD_REQUIRE((reinterpret_cast<uintptr_t>(d.rgb) & 7) == 0 && (d.rgb_pitch & 7) == 0 && d.height > 0 &&
                          d.width > (d.blocks_per_mcu == 6 ? 4 : 0) && d.rgb_pitch >= 3 * d.width &&
                          (d.width + mcu_px - 1) / mcu_px == d.mcus_x &&
                          (d.height + mcu_px - 1) / mcu_px == d.total_blocks / d.blocks_per_mcu / d.mcus_x,
                      DALIAMD_ERROR_INVALID_ARGUMENT,
                      "daliamdJpegHuffmanSetup: sample %d: rgb must be 8-byte aligned with a pitch that is a multiple of 8 "
                      "and covers 3 * width bytes; width (4:2:0: > 4) and height must match the MCU geometry", i);
    }
    const bool fused = !color && d.comp_of_block[0] < 3 && d.plane[d.comp_of_block[0]] != nullptr;
    for (int k = 0; k < d.blocks_per_mcu && !color; k++) {
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
    DALIAMD_REQUIRE(d.restart_interval >= 0 && d.restart_interval <= 65535, DALIAMD_ERROR_INVALID_ARGUMENT,
                    "daliamdJpegHuffmanSetup: sample %d: restart interval %d", i, d.restart_interval);
    d.tile_start = tiles;
    // (a stream that brings its index is not un-stuffed again: no tiles)
    d.num_tiles = d.index ? 0 : daliamd::NumTiles((int)(reinterpret_cast<uintptr_t>(d.ecs) & 15), d.ecs_len);
    d.seg_start = segs;
    d.num_segments = daliamd::NumSegments(d.ecs_len);
    DALIAMD_REQUIRE(d.num_segments <= 65535, DALIAMD_ERROR_UNSUPPORTED,
                    "daliamdJpegHuffmanSetup: sample %d: entropy-coded segment too long (decode it on the host)", i);
    d.blk_wg_start = bwgs;
    tiles += d.num_tiles;
    segs += d.num_segments;
    int mcus = d.total_blocks / d.blocks_per_mcu, rm[4];
    if (!color && daliamd::RectMcus(d, rm)) mcus = rm[2] * rm[3] > 0 ? rm[2] * rm[3] : 1;   // region of interest: its MCU rectangle
    const int mpw = color ? daliamd::ColorBandRows(d.mcus_x) * d.mcus_x : daliamd::McusPerWg(d.blocks_per_mcu);
    bwgs += (mcus + mpw - 1) / mpw;
  }
  *num_tiles = tiles;
  *num_segments = segs;
  *num_block_workgroups = bwgs;
  *block_kernels = kinds;
  return DALIAMD_SUCCESS;
}

// parts: bit 0 - the front (code tables + un-stuffing: needs the descriptors and the streams' bytes, nothing else), bit 1 - the
// rest.  A caller may run the front on a side stream as soon as the bytes are on the device and order the rest behind it.
static daliamdResult_t LaunchHuffman(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                     int num_segments, int num_block_workgroups, daliamdEvent_t *events, int block_kernels,
                                     int parts = 3) {
  if (n == 0) return DALIAMD_SUCCESS;
  DALIAMD_REQUIRE(descs_dev && n > 0 && num_tiles >= 0 && num_segments >= n && num_block_workgroups >= n,
                  DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanRun: invalid argument");
  using namespace daliamd;
  hipStream_t s = (hipStream_t)stream;
  const int seg_grid = XcdGrid(num_segments);
  int e = 0;
  auto mark = [&]() -> hipError_t { return events ? hipEventRecord((hipEvent_t)events[e++], s) : hipSuccess; };
  // (each launch also sits in a KernelTimer scope: daliamdKernelTimingEnable times the product path's launches
  // by name; `events` is the explicit variant of the same thing for callers that bring their own events)
  // which streams the table holds (Setup): PARSED - un-stuffing, relaxation, hand-over check, DC pass; INDEXED - resident
  // streams that bring their index: one launch instead of those four (+ the table build, which every stream needs)
  const bool parsed = (block_kernels & DALIAMD_JPEG_HUFFMAN_PARSED) != 0 ||
                      !(block_kernels & (DALIAMD_JPEG_HUFFMAN_PARSED | DALIAMD_JPEG_HUFFMAN_INDEXED));
  const bool indexed = (block_kernels & DALIAMD_JPEG_HUFFMAN_INDEXED) != 0;
  // (callers of the plain entry points pass no kinds: build)
  const bool build_tables = (block_kernels & DALIAMD_JPEG_HUFFMAN_BUILD_TABLES) != 0 ||
                            !(block_kernels & (DALIAMD_JPEG_HUFFMAN_PARSED | DALIAMD_JPEG_HUFFMAN_INDEXED));
  const int ntab = build_tables ? n : 0;
  DALIAMD_HIP_CHECK(mark());
  if ((parts & 1) && num_tiles + ntab > 0) {
    KernelTimer timer("PrepareKernel", s);
    hipLaunchKernelGGL(PrepareKernel, dim3(num_tiles + ntab), dim3(kTileThreads), 0, s, descs_dev, n, num_tiles, ntab);
  }
  DALIAMD_HIP_CHECK(mark());
  if ((parts & 1) && parsed && num_tiles > 0) {
    KernelTimer timer("UnstuffScatterKernel", s);
    hipLaunchKernelGGL(UnstuffScatterKernel, dim3(num_tiles), dim3(kTileThreads), 0, s, descs_dev, n);
  }
  DALIAMD_HIP_CHECK(mark());
  if (!(parts & 2)) {
    DALIAMD_HIP_CHECK(hipGetLastError());
    return DALIAMD_SUCCESS;
  }
  if (parsed) {
    KernelTimer timer("SyncKernel", s);
    hipLaunchKernelGGL(SyncKernel, dim3(seg_grid), dim3(kSegThreads), 0, s, descs_dev, n, num_segments);
  }
  if (indexed) {
    KernelTimer timer("IndexedSyncKernel", s);
    hipLaunchKernelGGL(IndexedSyncKernel, dim3(seg_grid), dim3(kSegThreads), 0, s, descs_dev, n, num_segments);
  }
  DALIAMD_HIP_CHECK(mark());
  if (parsed) {
    KernelTimer timer("PropagateKernel", s);
    hipLaunchKernelGGL(PropagateKernel, dim3(n), dim3(kSegThreads), 0, s, descs_dev);
  }
  DALIAMD_HIP_CHECK(mark());
  if (parsed) {
    KernelTimer timer("DcKernel", s);
    hipLaunchKernelGGL(DcKernel, dim3(seg_grid), dim3(kDcThreads), 0, s, descs_dev, n, num_segments);
  }
  DALIAMD_HIP_CHECK(mark());
  // bit 0: streams with plane / coefficient output, bit 1: streams with the fused colour output (both instances walk the
  // same grid; a workgroup of the other instance's stream leaves at once)
  if (block_kernels & 1) {
    KernelTimer timer("BlockKernel", s);
    hipLaunchKernelGGL(BlockKernel<false>, dim3(XcdGrid(num_block_workgroups)), dim3(kBlockThreads), 0, s, descs_dev, n,
                       num_block_workgroups);
  }
  if (block_kernels & 2) {
    {
      KernelTimer timer("BlockColorKernel", s);
      hipLaunchKernelGGL(BlockKernel<true>, dim3(XcdGrid(num_block_workgroups)), dim3(kBlockThreads), 0, s, descs_dev, n,
                         num_block_workgroups);
    }
    KernelTimer timer("SeamKernel", s);
    hipLaunchKernelGGL(SeamKernel, dim3(XcdGrid(num_block_workgroups)), dim3(kSeamThreads), 0, s, descs_dev, n,
                       num_block_workgroups);
  }
  if (block_kernels & DALIAMD_JPEG_HUFFMAN_BUILD_INDEX) {   // (behind the value pass: off this batch's critical path)
    KernelTimer timer("IndexBuildKernel", s);
    hipLaunchKernelGGL(IndexBuildKernel, dim3(seg_grid), dim3(kSegThreads), 0, s, descs_dev, n, num_segments);
  }
  DALIAMD_HIP_CHECK(mark());
  DALIAMD_HIP_CHECK(hipGetLastError());
  return DALIAMD_SUCCESS;
}

daliamdResult_t daliamdJpegHuffmanRun(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                      int num_segments, int num_block_workgroups) {
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, nullptr, 1);
}

daliamdResult_t daliamdJpegHuffmanRunColor(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                           int num_segments, int num_block_workgroups, int block_kernels) {
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, nullptr, block_kernels & 63);
}

daliamdResult_t daliamdJpegHuffmanRunFront(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                           int num_segments, int num_block_workgroups, int block_kernels) {
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, nullptr, block_kernels & 63, 1);
}
daliamdResult_t daliamdJpegHuffmanRunBack(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n, int num_tiles,
                                          int num_segments, int num_block_workgroups, int block_kernels) {
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, nullptr, block_kernels & 63, 2);
}

daliamdResult_t daliamdJpegHuffmanRunProfiled(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                              int num_tiles, int num_segments, int num_block_workgroups,
                                              daliamdEvent_t *events) {
  DALIAMD_REQUIRE(events, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanRunProfiled: events is NULL");
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, events, 1);
}

daliamdResult_t daliamdJpegHuffmanRunProfiledColor(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                                   int num_tiles, int num_segments, int num_block_workgroups,
                                                   int block_kernels, daliamdEvent_t *events) {
  DALIAMD_REQUIRE(events, DALIAMD_ERROR_INVALID_ARGUMENT, "daliamdJpegHuffmanRunProfiledColor: events is NULL");
  return LaunchHuffman(stream, descs_dev, n, num_tiles, num_segments, num_block_workgroups, events, block_kernels & 63);
}

}  // extern "C"
