/*
 * dali_amd_kernels.h -- C ABI of libdali_amd_kernels.so (hand-written gfx950 HIP kernels).
 *
 * This is the drop-in boundary for the JPEG -> RandomResizedCrop -> CropMirrorNormalize hot path
 * (and its sibling per-sample kernels).  It replaces the reference's *kernel-level* contract
 *     Kernel::Setup(KernelContext&, in_shapes, args) -> KernelRequirements
 *     Kernel::Run  (KernelContext&, OutListGPU, InListGPU, args)    with ctx.gpu.stream
 * (dali/kernels/context.h:34-189, dali/kernels/kernel_manager.h) for the kernels listed below.
 *
 * Conventions (modelled on include/dali/dali.h:140-164):
 *   - every entry point returns daliamdResult_t; on failure a thread-local message is available
 *     through daliamdGetLastErrorMessage();
 *   - no exceptions cross the ABI, no torch/HIP types appear in signatures: streams and events are
 *     opaque `void*` (hipStream_t / hipEvent_t values), buffers are plain pointers + sizes;
 *   - "*Setup" functions are pure host code: they turn per-sample arguments into POD descriptor
 *     tables in caller-owned HOST memory; the caller copies the tables to the device;
 *   - "*Run" functions take DEVICE pointers to those tables and only ENQUEUE work on the given
 *     stream; they never allocate, never synchronise and are hipGraph-capturable;
 *   - the caller owns every buffer.
 */
#ifndef DALI_AMD_KERNELS_H_
#define DALI_AMD_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DALIAMD_API __attribute__((visibility("default")))

typedef enum {
  DALIAMD_SUCCESS = 0,
  DALIAMD_ERROR_INVALID_ARGUMENT = 1,
  DALIAMD_ERROR_UNSUPPORTED = 2,
  DALIAMD_ERROR_OUT_OF_RANGE = 3,
  DALIAMD_ERROR_HIP = 4,
  DALIAMD_ERROR_INTERNAL = 5,
  DALIAMD_ERROR_CORRUPT_STREAM = 6
} daliamdResult_t;

typedef void *daliamdStream_t; /* hipStream_t */
typedef void *daliamdEvent_t;  /* hipEvent_t  */

/* ----------------------------------------------------------------------------------------------
 * Library / device plumbing (what the reference gets from CUDAStreamPool / CUDAEventPool /
 * mm:: resources: include/dali/core/cuda_stream_pool.h, cuda_event_pool.h, mm/).
 * -------------------------------------------------------------------------------------------- */
DALIAMD_API const char *daliamdGetLastErrorMessage(void);
DALIAMD_API void daliamdClearLastError(void);
DALIAMD_API int daliamdVersion(void);
DALIAMD_API daliamdResult_t daliamdDeviceCount(int *count);
DALIAMD_API daliamdResult_t daliamdSetDevice(int device_id);
DALIAMD_API daliamdResult_t daliamdDeviceInfo(int device_id, char *arch_name, int arch_name_len,
                                              int *num_cus, size_t *total_mem);
/* PCI bus id "dddd:bb:dd.f" of the device (>= 16 bytes): the host executor derives the device's NUMA-local CPUs from
 * /sys/bus/pci/devices/<id>/local_cpulist for Pipeline(set_affinity=True) (the reference asks NVML:
 * dali/util/nvml.h, pipeline.py:164). */
DALIAMD_API daliamdResult_t daliamdDevicePciBusId(int device_id, char *bus_id, int len);
/* Profiler ranges around the executor's stages and operators (reference: include/dali/core/nvtx.h:53-82); roctx is
 * resolved at first use, both calls do nothing when it is not installed. */
/* Benchmarks: while enabled every kernel launch of this library is bracketed by a pair of timing events on the stream it
 * is launched on.  Report: "name\tlaunches\tavg_ms\n" per kernel for the launches since the previous report (waits for
 * them); returns the length needed, writes at most len - 1 characters + terminator.  A call without a buffer only
 * sizes; the call that receives the text also resets the statistics.  on > 1 additionally creates the events of `on`
 * launches up front, so that no event is created inside the measured region. */
DALIAMD_API void daliamdKernelTimingEnable(int on);
DALIAMD_API int daliamdKernelTimingReport(char *buf, int len);
DALIAMD_API void daliamdRangePush(const char *name);
DALIAMD_API void daliamdRangePop(void);
DALIAMD_API daliamdResult_t daliamdStreamCreate(daliamdStream_t *stream, int non_blocking);
/* priority < 0: the device's highest stream priority, 0: the default, > 0: the lowest.  The ROCm runtime keeps one set
 * of hardware queues PER PRIORITY: a stream of another priority never shares an in-order hardware queue with the
 * default-priority streams of the process (torch's, the null stream). */
DALIAMD_API daliamdResult_t daliamdStreamCreateWithPriority(daliamdStream_t *stream, int non_blocking, int priority);
DALIAMD_API daliamdResult_t daliamdStreamDestroy(daliamdStream_t stream);
DALIAMD_API daliamdResult_t daliamdStreamSynchronize(daliamdStream_t stream);
DALIAMD_API daliamdResult_t daliamdStreamWaitEvent(daliamdStream_t stream, daliamdEvent_t event);
/* enable_timing: bit 0 = the event records time stamps, bit 1 = daliamdEventSynchronize on it blocks (sleeps) instead of polling */
DALIAMD_API daliamdResult_t daliamdEventCreate(daliamdEvent_t *event, int enable_timing);
DALIAMD_API daliamdResult_t daliamdEventDestroy(daliamdEvent_t event);
DALIAMD_API daliamdResult_t daliamdEventRecord(daliamdEvent_t event, daliamdStream_t stream);
DALIAMD_API daliamdResult_t daliamdEventSynchronize(daliamdEvent_t event);
/* *done = 1 when all work captured by the last record of `event` has finished, 0 otherwise (never blocks) */
DALIAMD_API daliamdResult_t daliamdEventQuery(daliamdEvent_t event, int *done);
DALIAMD_API daliamdResult_t daliamdEventElapsedMs(daliamdEvent_t start, daliamdEvent_t stop, float *ms);
DALIAMD_API daliamdResult_t daliamdMalloc(void **ptr, size_t bytes);
DALIAMD_API daliamdResult_t daliamdFree(void *ptr);
DALIAMD_API daliamdResult_t daliamdHostAlloc(void **ptr, size_t bytes); /* pinned */
DALIAMD_API daliamdResult_t daliamdHostFree(void *ptr);
DALIAMD_API daliamdResult_t daliamdMemcpyH2DAsync(void *dst, const void *src, size_t bytes, daliamdStream_t s);
DALIAMD_API daliamdResult_t daliamdMemcpyD2HAsync(void *dst, const void *src, size_t bytes, daliamdStream_t s);
DALIAMD_API daliamdResult_t daliamdMemcpyD2DAsync(void *dst, const void *src, size_t bytes, daliamdStream_t s);
DALIAMD_API daliamdResult_t daliamdMemsetAsync(void *dst, int value, size_t bytes, daliamdStream_t s);
/* Host memory the device reads IN PLACE (round 5; no reference counterpart: the reference's readers copy every file out of
 * its mapping, dali/util/mmaped_file.cc / operators/reader/loader/file_label_loader.cc, and the mixed decoder copies it
 * again to the device): page-locks [ptr, ptr + bytes) - e.g. a read-only file mapping, i.e. the page cache's own pages -
 * and reports in *same_address whether the device addresses the range at the same addresses (it does on MI355X; a caller
 * that gets 0 keeps copying).  Measured on the bench box: a device-side copy out of such a mapping runs at the bus rate
 * (57 GB/s), the registration of 64 MiB takes 0.2 ms (tools/probes/hostreg_probe.py).
 * CAUTION (round 6): pages registered this way must stay what they are while the registration lives.  Round 5's readers.file
 * registered its file mappings; truncating such a file made the driver evict the process's queues for minutes.  The reader
 * now hands out page-locked COPIES (daliamdHostAlloc) for the device-side fetch and no longer calls this; the entry stays
 * for callers that own the memory they register. */
DALIAMD_API daliamdResult_t daliamdHostRegister(void *ptr, size_t bytes, int *same_address);
DALIAMD_API daliamdResult_t daliamdHostUnregister(void *ptr);
/* Batched device-side copy: record i moves `bytes` bytes from src (device-visible memory: device, page-locked or
 * registered host memory) to dst (device), any alignment; one launch for the whole table (`descs`: device-visible,
 * `max_bytes` >= every record's bytes).  The mixed decoders fetch the encoded files of a batch with it straight out of
 * the reader's page-locked resident copies (DALI_AMD_READER_ZERO_COPY) - the files cross the bus once and no host core touches
 * their bytes again. */
typedef struct {
  const void *src;
  void *dst;
  uint64_t bytes;
  uint64_t reserved;
} daliamdGatherDesc;
DALIAMD_API daliamdResult_t daliamdGatherCopy(const daliamdGatherDesc *descs, int n, size_t max_bytes, daliamdStream_t s);
/* rows of `width_bytes` between two pitched device buffers (e.g. a row-padded image -> dense) */
DALIAMD_API daliamdResult_t daliamdMemcpy2DD2DAsync(void *dst, size_t dst_pitch, const void *src, size_t src_pitch,
                                                    size_t width_bytes, size_t height, daliamdStream_t s);

/* Element types of kernel outputs (subset of DALIDataType, include/dali/core/dali_data_type.h) */
typedef enum {
  DALIAMD_UINT8 = 0, DALIAMD_FLOAT16 = 1, DALIAMD_FLOAT = 2, DALIAMD_INT8 = 3, DALIAMD_INT16 = 4, DALIAMD_UINT16 = 5,
  DALIAMD_INT32 = 6, DALIAMD_UINT32 = 7
} daliamdDType_t;
typedef enum { DALIAMD_LAYOUT_HWC = 0, DALIAMD_LAYOUT_CHW = 1 } daliamdLayout_t;

/* ----------------------------------------------------------------------------------------------
 * JPEG: dequantise + 8x8 inverse DCT + chroma upsampling + YCbCr->RGB.
 * Replaces the arithmetic nvImageCodec performs for ImageDecoder
 * (dali/operators/imgcodec/image_decoder.h:810-815); libjpeg-turbo semantics: integer "islow"
 * IDCT, fancy (triangle) upsampling, BT.601 full range -- bit-exact with the CPU decoder.
 *
 * Coefficient layout (produced by the host or GPU entropy decoder): per component a dense array
 * of 8x8 blocks in raster order [blocks_y][blocks_x][64] of int16, each block stored COLUMN-MAJOR
 * (element index = col*8 + row) so one 16-byte load yields one column for IDCT pass 1.
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  const int16_t *coef; /* device: [nblocks][64], column-major blocks                      */
  uint8_t *plane;      /* device: component plane, [blocks_y*8][pitch]                    */
  int32_t blocks_x;    /* blocks per row (already padded to the MCU)                       */
  int32_t nblocks;     /* blocks_x * blocks_y                                              */
  int32_t pitch;       /* plane row pitch in bytes, multiple of 8                          */
  int32_t wg_start;    /* filled by Setup: first workgroup of this component               */
  uint16_t quant[64];  /* quantisation table, same column-major element order as the blocks */
  /* Region-of-interest decode: only the blocks [rect_y0, ..) x [rect_x0, rect_x0 + rect_w) are transformed;
   * nblocks = rect_w * rect_h then.  rect_w == 0: the whole component. */
  int32_t rect_x0, rect_y0, rect_w, reserved;
} daliamdJpegIdctDesc;

/* Fills wg_start of descs[0..n) and returns the grid size. */
DALIAMD_API daliamdResult_t daliamdJpegIdctSetup(daliamdJpegIdctDesc *descs_host, int n, int *num_workgroups);
DALIAMD_API daliamdResult_t daliamdJpegIdctRun(daliamdStream_t stream, const daliamdJpegIdctDesc *descs_dev,
                                               int n, int num_workgroups);

/* ----------------------------------------------------------------------------------------------
 * JPEG Huffman entropy decoding on the GPU (baseline, one interleaved scan, with or without restart intervals).
 * Replaces the GPU Huffman stage nvJPEG runs for device="mixed" when the stream is larger than
 * hybrid_huffman_threshold (dali/operators/imgcodec/image_decoder.h:810-815,
 * dali/operators/imgcodec/decoder_schema.cc "hybrid_huffman_threshold").  Output layout and values
 * are those of daliamdJpegDecodeCoefficients (dali_amd_host.h), i.e. the input of daliamdJpegIdctRun.
 * The caller fills the descriptor from the scan analysis (daliamdJpegAnalyzeScan / daliamdJpegAnalyzeHeader) and
 *   - uploads the entropy-coded segment to `ecs`; the bytes up to the next 16-byte boundary behind it must be
 *     readable.  The segment ENDS at the first marker that is not RSTn (0xFF followed by anything but 0x00, 0xFF,
 *     0xD0-0xD7): the un-stuffing pass finds it, so `ecs_len` may include the EOI marker and whatever follows it -
 *     "the rest of the file" - and the host never has to walk the scan,
 *   - zero-fills `status` before the launch (the coefficient arrays need no initialisation: every block that is
 *     decoded is written exactly once, as a full 128-byte line),
 *   - provides `scratch`: daliamdJpegHuffmanScratchBytes(ecs_len, total_blocks) bytes (streams with restart
 *     intervals: daliamdJpegHuffmanScratchBytesRestart), 16-byte aligned (clean stream, code tables, per-slice
 *     decoder states, the block starts of every segment, 10 bytes per block, 4 bytes per restart interval),
 *   - calls daliamdJpegHuffmanSetup on the host table, copies it to the device, calls daliamdJpegHuffmanRun.
 * The work is cut into image-independent pieces (8 KB tiles for the byte un-stuffing, 61 KB segments of 256-byte
 * slices for the self-synchronising parallel decode, one lane per 8x8 block for the values), so a batch of very differently sized streams still fills
 * the device.  Restart intervals: the RSTn markers are removed with the byte stuffing and every boundary is a point
 * where the parallel decode knows its state; the DC predictions restart per interval.
 * After the launch *status is 0 on success, 2 when the segment holds fewer blocks than the frame
 * header promises (truncated / corrupt stream: decode it with the host decoder to get the diagnosis), 3 when RSTn
 * markers turn up in a stream whose header announces none, 4 when a restart interval does not end the way T.81
 * F.1.2.3 prescribes (wrong MCU count or a padding that is not one-bits).
 * -------------------------------------------------------------------------------------------- */
#define DALIAMD_JPEG_MAX_BLOCKS_PER_MCU 10
typedef struct {
  const uint8_t *ecs;      /* device: entropy-coded segment (still byte-stuffed)                  */
  uint8_t *scratch;        /* device scratch, see above                                            */
  int32_t *status;         /* device: one int, pre-zeroed                                          */
  int16_t *coef[3];        /* device: per component [blocks_y][blocks_x][64], 16-byte aligned (or see plane) */
  int32_t ecs_len;
  int32_t blocks_per_mcu;  /* sum of h*v over the components                                       */
  int32_t mcus_x;          /* MCUs per row                                                         */
  int32_t total_blocks;    /* mcus_x * mcus_y * blocks_per_mcu                                     */
  int32_t blocks_x[3];     /* allocated blocks per row of each component                           */
  int32_t h_samp[3], v_samp[3];
  int32_t tile_start, num_tiles;   /* filled by Setup: un-stuffing workgroups of this stream       */
  int32_t seg_start, num_segments; /* filled by Setup: decoding workgroups of this stream          */
  int32_t blk_wg_start;            /* filled by Setup: block-decoding workgroups of this stream    */
  int32_t table_owner;             /* filled by Setup: index of the stream that builds (and holds) the code tables this
                                      stream uses - streams with identical DHT contents and MCU structure share them */
  uint8_t comp_of_block[12];  /* component of the k-th block of an MCU                             */
  uint8_t h_of_block[12], v_of_block[12]; /* its position inside the component's MCU footprint     */
  uint8_t dc_sel[4], ac_sel[4];  /* per component: table selector, 0 or 1                           */
  uint8_t bits[4][16];     /* DHT code-length counts: [0],[1] = DC tables 0,1; [2],[3] = AC 0,1    */
  uint8_t vals[4][256];    /* DHT symbol lists, same order                                         */
  /* Region-of-interest decode: per component the block rectangle {x0, y0, x1, y1} (x1/y1 exclusive) whose
   * coefficients are needed; blocks outside are parsed (the stream is serial) but not stored, and the parse stops
   * after the last MCU row that intersects a rectangle.  All zero: every block. */
  int32_t rect[3][4];
  /* Optional fused output.  With plane[c] != NULL for the components of the stream the decoder dequantises and
   * inverse-transforms every block it assembles (the arithmetic of daliamdJpegIdctRun) and writes the 8x8 samples
   * into the component planes - [blocks_y*8][plane_pitch[c]], the input of daliamdJpegColorRun - instead of storing
   * the coefficients; coef[] is not used then and the 2 x 2 bytes per coefficient of traffic between the two kernels
   * disappear.  plane[c] must be 8-byte aligned, plane_pitch[c] a multiple of 8 (>= blocks_x[c] * 8). */
  uint8_t *plane[3];
  int32_t plane_pitch[3];
  int32_t restart_interval; /* MCUs per restart interval (DRI), 0: the stream has no RSTn markers  */
  uint16_t quant[3][64];   /* per component, column-major element order as daliamdJpegIdctDesc.quant */
  /* Optional fused colour output (round 4).  With rgb != NULL the decoder also upsamples the chroma (libjpeg-turbo's
   * fancy h2v2 triangle filter) and converts to RGB where the samples are - the chroma blocks of a band of MCU rows in
   * LDS, the luma block of a lane in its registers - and writes interleaved RGB: no component planes, no
   * daliamdJpegColorRun for this stream (the arithmetic is that kernel's: same bits).  The first / last pixel row
   * of a band needs the chroma row of the neighbouring band: the bands leave those rows in `scratch` and one small
   * launch behind the block kernel finishes the seams.  Only for streams daliamdJpegHuffmanColorFusable accepts
   * (YCbCr 4:2:0 or 4:4:4 in the usual block order, or one component - its samples become R = G = B -, at most 128 MCUs
   * wide = 2048 / 1024 pixels, no rect); the caller vouches for the rest: a YCbCr or grayscale stream (not RGB-coded),
   * upright 3-channel RGB output wanted, 4:2:0: width > 4.  rgb 8-byte aligned, rgb_pitch a multiple of 8
   * (>= 3 * width); plane[] / coef[] are not used. */
  uint8_t *rgb;
  int32_t rgb_pitch;
  int32_t width, height;   /* image size in pixels */
  int32_t reserved;
  /* Side information of a RESIDENT stream (round 5; the reference indexes what it reads again and again - tools/tfrecord2idx,
   * tools/wds2idx.py, tools/rec2idx.py -, and its decoder cache keeps what epoch 2 would otherwise recompute,
   * dali/operators/decoder/cache/cached_decoder_impl.cc:124-141).  A stream that stays in device memory is parsed identically in
   * every epoch: where its stuffing bytes are and in which state a decoder reaches each 256-byte slice does not change.
   * index_out != NULL: this decode also leaves an index entry there - a 64-byte header, the un-stuffed stream and 12 bytes per
   * slice (entry bit position, block index inside the MCU, zig-zag index, ordinal of the first block, DC predictors):
   * daliamdJpegHuffmanIndexBytes(ecs_len) bytes, 64-byte aligned, valid once the launch has finished with *status == 0.
   * index != NULL: decode from such an entry - `ecs` is not looked at (may be NULL; ecs_len must be the value the entry was
   * built with): no un-stuffing, one decode per slice instead of the relaxation, no hand-over check, no DC pass, and under a
   * region of interest (rect) only the slices that hold blocks of it.  The output is the same bits.  Streams with restart
   * intervals take neither (Setup refuses). */
  const uint8_t *index;
  uint8_t *index_out;
  /* Finished code tables in device memory (round 5): daliamdJpegHuffmanTablesBytes() bytes as daliamdJpegHuffmanTablesBuild makes
   * them on the host from this descriptor's bits / vals / comp_of_block / dc_sel / ac_sel / blocks_per_mcu, 16-byte aligned.
   * They depend on nothing else, and a data set holds a handful of distinct sets (most files carry the tables of T.81 Annex K):
   * a caller that keeps them per set spares every launch the table-building workgroups - and an all-indexed launch its whole
   * first kernel.  NULL: built in the launch, once per distinct set of the table, as before. */
  const uint8_t *tables;
} daliamdJpegHuffDesc;

/* kinds of streams in a table, OR-ed into *block_kernels by daliamdJpegHuffmanSetupColor next to bits 0 / 1 */
#define DALIAMD_JPEG_HUFFMAN_PARSED 4       /* streams that are un-stuffed and synchronised in this launch */
#define DALIAMD_JPEG_HUFFMAN_INDEXED 8      /* streams that bring their index                              */
#define DALIAMD_JPEG_HUFFMAN_BUILD_INDEX 16 /* streams whose index entry is built behind the decode        */
#define DALIAMD_JPEG_HUFFMAN_BUILD_TABLES 32 /* streams that do not bring their code tables                 */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanTablesBytes(size_t *bytes);
/* host only: out_host receives daliamdJpegHuffmanTablesBytes() bytes */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanTablesBuild(const daliamdJpegHuffDesc *desc, void *out_host);
DALIAMD_API daliamdResult_t daliamdJpegHuffmanIndexBytes(int ecs_len, size_t *bytes);
/* host only: the index entry (daliamdJpegHuffmanIndexBytes(desc->ecs_len) bytes at index_out_host) of the stream whose
 * byte-stuffed segment `desc->ecs` points to IN HOST MEMORY - what a decode with index_out leaves behind on the device, made
 * offline (the reference indexes its containers offline: tools/tfrecord2idx, tools/wds2idx.py, tools/rec2idx.py).  Needs the
 * descriptor's code-table and MCU fields, total_blocks, ecs_len; restart_interval must be 0.  *status: 0, or 2 / 3 as after
 * a decode (truncated stream / restart markers) - such a stream gets no index. */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanIndexBuildHost(const daliamdJpegHuffDesc *desc, void *index_out_host, int32_t *status);

/* 1 when the stream's geometry (blocks_per_mcu, comp_of_block, h/v_of_block, h/v_samp, mcus_x, rect) allows the fused
 * colour output, else 0.  Host helper, looks at nothing else. */
DALIAMD_API int daliamdJpegHuffmanColorFusable(const daliamdJpegHuffDesc *desc);

DALIAMD_API daliamdResult_t daliamdJpegHuffmanScratchBytes(int ecs_len, int total_blocks, size_t *bytes);
/* num_intervals = ceil(MCUs of the frame / restart_interval), 0 without restart intervals */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanScratchBytesRestart(int ecs_len, int total_blocks, int num_intervals,
                                                                  size_t *bytes);
/* Validates the table, fills tile_start/num_tiles/seg_start/num_segments/blk_wg_start, returns the three grid sizes. */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanSetup(daliamdJpegHuffDesc *descs_host, int n, int *num_tiles,
                                                    int *num_segments, int *num_block_workgroups);
/* Six launches: prepare (un-stuff count + code tables), un-stuff scatter, synchronise (positions + block starts),
 * propagate (segment hand-over, block ordinals), DC (one lane per block), block decode + dequantisation + IDCT
 * (+ the seam launch when a stream of the table has a fused colour output). */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanRun(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                                  int num_tiles, int num_segments, int num_block_workgroups);
/* Same launches with events[0..6] (created with timing enabled) recorded before each kernel and after the last:
 * events[i] .. events[i+1] brackets kernel i of {prepare, un-stuff scatter, synchronise, propagate, DC, block}.
 * For benchmarks. */
/* The same three calls for tables in which streams ask for the fused colour output (rgb != NULL; the plain Setup
 * refuses those).  block_kernels (out of Setup, into Run): bit 0 - streams with plane / coefficient output, bit 1 -
 * streams with the fused colour output; Run launches the block kernel instance(s) that have streams, and the seam
 * launch behind the colour instance. */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanSetupColor(daliamdJpegHuffDesc *descs_host, int n, int *num_tiles,
                                                         int *num_segments, int *num_block_workgroups, int *block_kernels);
DALIAMD_API daliamdResult_t daliamdJpegHuffmanRunColor(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                                       int num_tiles, int num_segments, int num_block_workgroups,
                                                       int block_kernels);
/* daliamdJpegHuffmanRunColor in two halves.  The front (code tables of the streams that do not bring them, un-stuffing) needs
 * the descriptor table and the streams' bytes on the device and nothing else: it may run on a side stream while `stream` is
 * still busy with the previous batch, the back (everything else) then waits for it through an event of the caller's. */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanRunFront(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                                       int num_tiles, int num_segments, int num_block_workgroups,
                                                       int block_kernels);
DALIAMD_API daliamdResult_t daliamdJpegHuffmanRunBack(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev, int n,
                                                      int num_tiles, int num_segments, int num_block_workgroups,
                                                      int block_kernels);
#define DALIAMD_JPEG_HUFFMAN_KERNELS 6
DALIAMD_API daliamdResult_t daliamdJpegHuffmanRunProfiled(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev,
                                                          int n, int num_tiles, int num_segments,
                                                          int num_block_workgroups, daliamdEvent_t *events);
/* (the last bracket holds the block kernel instance(s) and the seam launch) */
DALIAMD_API daliamdResult_t daliamdJpegHuffmanRunProfiledColor(daliamdStream_t stream, const daliamdJpegHuffDesc *descs_dev,
                                                               int n, int num_tiles, int num_segments,
                                                               int num_block_workgroups, int block_kernels,
                                                               daliamdEvent_t *events);

typedef enum {
  DALIAMD_JPEG_GRAY = 0,   /* 1 component                                   */
  DALIAMD_JPEG_YCC = 1,    /* 3 components, YCbCr -> RGB                    */
  DALIAMD_JPEG_RGB = 2     /* 3 components stored as RGB (Adobe transform 0) */
} daliamdJpegColor_t;

/* Output format of the colour stage = the decoder's `output_type` (values of DALIImageType): RGB; BGR; GRAY - one
 * byte per pixel, the luma plane of a gray / YCbCr stream (libjpeg-turbo's JCS_GRAYSCALE output) or
 * 0.299 R + 0.587 G + 0.114 B; YCbCr - ITU-R BT.601 with head room computed from the RGB result in float
 * (dali/operators/imgcodec/util/convert.h:140-192, kernels/imgproc/color_manipulation/color_space_conversion_impl.h). */
typedef enum {
  DALIAMD_JPEG_OUT_RGB = 0, DALIAMD_JPEG_OUT_BGR = 1, DALIAMD_JPEG_OUT_GRAY = 2, DALIAMD_JPEG_OUT_YCBCR = 3
} daliamdJpegOutFormat_t;

typedef struct {
  const uint8_t *plane[3]; /* device component planes (output of the IDCT)                  */
  int32_t pitch[3];
  int32_t h_samp[3], v_samp[3]; /* sampling factors; hmax/vmax = max over components         */
  int32_t down_w[3], down_h[3]; /* downsampled_width/height of each component (samples)      */
  int32_t width, height;        /* image size                                                */
  int32_t color;                /* daliamdJpegColor_t                                        */
  uint8_t *out;                 /* device: u8 HWC, 3 channels (1 for DALIAMD_JPEG_OUT_GRAY)   */
  int32_t out_pitch;            /* bytes per output row (>= channels * output width)         */
  int32_t wg_start;             /* filled by Setup                                           */
  int32_t orientation;          /* EXIF orientation to undo while writing: 0/1 none, 2..8;   */
                                /* for 5..8 the output is height x width transposed          */
  int32_t out_format;           /* daliamdJpegOutFormat_t (decoders.image output_type)        */
  /* Region-of-interest decode: only the source pixels [roi_y0, roi_y0 + roi_h) x [roi_x0, roi_x0 + roi_w) of the
   * (un-rotated) image are produced; `out` then receives the window whose upright-image origin is
   * (out_y0, out_x0), i.e. upright pixel (oy, ox) lands at out[(oy - out_y0) * out_pitch + 3 * (ox - out_x0)].
   * roi_w == 0: the whole image (out_x0 = out_y0 = 0). */
  int32_t roi_x0, roi_y0, roi_w, roi_h, out_x0, out_y0;
} daliamdJpegColorDesc;

/* kernel_mask (out): which colour kernels have samples in this table - bits 0 / 1: the RGB kernel (fast paths: YCbCr
 * 4:2:0, 4:4:4, grayscale in planes-aligned windows / every other sampling and orientation), bit 2: the kernel with a BGR / YCbCr / gray conversion behind it (100 registers more, kept out of
 * the common kernel), bit 3: the 4:2:0 fast path for windows that start anywhere (region-of-interest decode); hand it to
 * Run, which launches only those (all walk the same grid, a workgroup of another kernel's sample leaves at once). */
DALIAMD_API daliamdResult_t daliamdJpegColorSetup(daliamdJpegColorDesc *descs_host, int n, int *num_workgroups,
                                                  int *kernel_mask);

/* Region-of-interest decode (the reference's decoders.image_crop / image_slice / image_random_crop hand a region to
 * nvImageCodec: dali/operators/imgcodec/image_decoder.h:683-716, roi_image_decoder.h:36-91).  Host helper: turns a
 * window of the UPRIGHT (orientation-adjusted) image into what the three decode stages need: the window in the
 * un-rotated source image + output origin (daliamdJpegColorDesc.roi_*, out_*), and per component the rectangle of
 * 8x8 blocks whose samples the colour stage will read (daliamdJpegHuffDesc.rect, daliamdJpegIdctDesc.rect_*). */
typedef struct {
  int32_t roi_x0, roi_y0, roi_w, roi_h; /* source (un-rotated) pixels                      */
  int32_t out_x0, out_y0;               /* upright-image origin of the output window       */
  int32_t rect[3][4];                   /* per component: block x0, y0, x1, y1 (exclusive) */
} daliamdJpegRoiPlan;
DALIAMD_API daliamdResult_t daliamdJpegPlanRoi(int width, int height, int num_components, const int32_t *h_samp,
                                               const int32_t *v_samp, int orientation, int up_y0, int up_x0, int up_h,
                                               int up_w, daliamdJpegRoiPlan *plan);
DALIAMD_API daliamdResult_t daliamdJpegColorRun(daliamdStream_t stream, const daliamdJpegColorDesc *descs_dev,
                                                int n, int num_workgroups, int kernel_mask);

/* ----------------------------------------------------------------------------------------------
 * Separable resampling (RandomResizedCrop / Resize) with an optional fused
 * CropMirrorNormalize epilogue.
 * Replaces SeparableResamplingGPUImpl::{Setup,Run} (dali/kernels/imgproc/resample/
 * separable_impl.h:90-190, resampling_setup.cc:271-469, resampling_batch.cu:25-109) and, when
 * fused, SliceHwc2HwcChwNormalizeGPU (dali/kernels/slice/slice_hwc2chw_normalize_gpu.cu:631-990).
 * Arithmetic follows the CPU backend (pre-normalised coefficients, mul+add, fp32 intermediate,
 * reference pass order): dali/kernels/imgproc/resample/{resampling_impl_cpu.cc:22-47,
 * resampling_impl_cpu.h:50-390, separable_cpu.h:152-241}.
 * -------------------------------------------------------------------------------------------- */
/* CUBIC / LANCZOS3 / GAUSSIAN: the reference's tabulated windows (resampling_filters.cu:38-142); NN on both axes only */
typedef enum {
  DALIAMD_INTERP_NN = 0, DALIAMD_INTERP_LINEAR = 1, DALIAMD_INTERP_TRIANGULAR = 2, DALIAMD_INTERP_CUBIC = 3,
  DALIAMD_INTERP_LANCZOS3 = 4, DALIAMD_INTERP_GAUSSIAN = 5
} daliamdInterp_t;

typedef struct {
  /* input image, u8 HWC */
  const uint8_t *in;
  int32_t in_h, in_w, channels, in_pitch;
  /* region of interest in source pixels: [roi_y0, roi_y1) x [roi_x0, roi_x1); use_roi = 0: whole */
  int32_t use_roi;
  float roi_y0, roi_x0, roi_y1, roi_x1;
  int32_t out_h, out_w;
  int32_t min_filter, mag_filter, antialias; /* daliamdInterp_t; defaults LINEAR, LINEAR, 1 */
  /* output */
  void *out;
  int32_t out_dtype;  /* daliamdDType_t: UINT8 (plain resize) or FLOAT16 / FLOAT (fused CMN) */
  int32_t out_layout; /* daliamdLayout_t */
  int32_t normalize;  /* 1: out = (u8 - mean[c]) * inv_std[c] after rounding to u8 */
  int32_t mirror;     /* 1: horizontal flip of the output */
  float mean[4], inv_std[4];
  /* Other element types (the reference resamples u8 / i16 / u16 / f32, resampling_batch.cu:125-152): in_dtype = UINT8
   * (0, default), INT16, UINT16 or FLOAT; out_dtype then = in_dtype, or FLOAT with `unrounded` = 1 for the float result
   * of the second pass as it is (fn.resize(dtype=FLOAT)).  These take a plain two-launch path (fp32 intermediate in the
   * workspace), not the fused tile kernel; `normalize` must be 0. */
  int32_t in_dtype, unrounded;
  /* full_h > 0: `in` holds only the window [org_y, org_y + in_h) x [org_x, org_x + in_w) of a full_h x full_w image, and
   * the region of interest (required) is given in THAT image's coordinates.  All coordinate arithmetic - filter centres,
   * coefficients, pass order, border clamping - is the full image's, so the result equals resampling the full image bit
   * for bit; the window must contain every source pixel the filters touch (the set-up refuses it otherwise).  This is
   * what lets a decoder that feeds only a RandomResizedCrop decode the crop window instead of the image. */
  int32_t full_h, full_w, org_y, org_x;
} daliamdResampleArgs;

typedef struct {
  const uint8_t *in;
  void *out;
  int32_t in_h, in_w, channels, in_pitch;
  int32_t out_h, out_w;
  int32_t first_axis;        /* 0: horizontal pass first; 1: vertical pass first (cost model)   */
  float origin[2], scale[2]; /* [0] = x, [1] = y; origin of the second axis is ROI-relative     */
  float fscale[2], fanchor[2];
  int32_t support[2];
  int32_t lo[2], ext[2];     /* clamp window of each axis (absolute start, extent)              */
  int32_t tile_w, tile_h, tiles_x, tiles_y;
  int32_t wg_start;
  int32_t out_dtype, out_layout, normalize, mirror;
  float mean[4], inv_std[4];
  /* H-last pass: output column x rounds half to even (the SIMD body of the reference's row loop) iff
   * round_lo[r] <= x < round_hi[r] for one of the four regions r, half away from zero otherwise (its scalar tails);
   * any output width (kernels/common/simd.h:53-56 vs core/convert.h:306-321) */
  int32_t round_lo[4], round_hi[4];
  int32_t lds_bytes;
  int32_t staged;            /* 1: the tile's source window is staged in LDS; 0: read from global memory */
  /* per-sample tables in the workspace (first-tap indices and normalised coefficients of every output column and
   * row, and - fused fp16 normalisation - the 256-entry result table per channel), filled by the first kernel of Run */
  int64_t table_off;         /* byte offset of the sample's tables inside the workspace */
  int32_t tab_start;         /* first table entry of the sample (work index of the tables kernel) */
  int32_t use_lut;
  int32_t filter_kind[2];    /* per axis: 0 nearest, 1 triangular, 2 Gaussian, 3 Lanczos3, 4 cubic (dali_amd_resample_filters.h) */
  /* the two-launch path: fp32 intermediate [tmp_h][tmp_w][channels] at tmp_off in the workspace, gen_start[p] = first work
   * item of the sample in pass p.  generic = 1: the other element types; generic = 2: a u8 sample whose down-scaling is too
   * extreme for the tile kernel (its fused epilogue - normalize, mirror, layout, fp16 - runs in the second launch) */
  int32_t in_dtype, unrounded, generic, round_lanes;
  int32_t tmp_w, tmp_h;
  int64_t tmp_off, gen_start[2];
} daliamdResampleDesc;

/* Fills descs_host[0..n); returns the grid size and the dynamic LDS bytes the launch needs. */
/* What Setup hands to Run besides the descriptor table. */
typedef struct {
  int32_t num_tiles;          /* tiles of the batch (the resampling kernel takes a few consecutive tiles per workgroup) */
  int32_t lds_bytes;          /* dynamic LDS of the launch */
  int32_t table_entries;      /* per-sample table entries the first kernel computes */
  int32_t reserved;
  size_t workspace_bytes;     /* device scratch Run needs: per-sample tables, fp32 intermediates of the generic path and one
                                 128-byte record per tile; written and read by Run's own kernels, so one buffer per stream
                                 (or per iteration in flight) is enough */
  int64_t generic_items[2];   /* elements of the two passes of the generic (non-u8 / unrounded) path */
} daliamdResamplePlan;
DALIAMD_API daliamdResult_t daliamdResampleSetup(const daliamdResampleArgs *args, int n, daliamdResampleDesc *descs_host,
                                                daliamdResamplePlan *plan);
DALIAMD_API daliamdResult_t daliamdResampleRun(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                              const daliamdResamplePlan *plan, void *workspace_dev);
/* daliamdResampleRun = RunTables + RunPasses on one stream.  RunTables (per-sample coefficient tables and per-tile records into
 * the workspace) reads nothing but the descriptor table: it may run on a side stream while `stream` still executes the kernels
 * that produce the source images; the caller then orders RunPasses behind it with an event. */
DALIAMD_API daliamdResult_t daliamdResampleRunTables(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                                     const daliamdResamplePlan *plan, void *workspace_dev);
DALIAMD_API daliamdResult_t daliamdResampleRunPasses(daliamdStream_t stream, const daliamdResampleDesc *descs_dev, int n,
                                                     const daliamdResamplePlan *plan, void *workspace_dev);

/* ----------------------------------------------------------------------------------------------
 * Stand-alone CropMirrorNormalize: u8 HWC -> {fp16, fp32, u8, i8} HWC/CHW with crop, horizontal
 * mirror, per-channel normalisation, channel padding and out-of-bounds fill.
 * Replaces SliceFlipNormalizePermutePad (CPU arithmetic:
 * dali/kernels/slice/slice_flip_normalize_permute_pad_cpu.h:37-64) and the GPU fast path
 * dali/kernels/slice/slice_hwc2chw_normalize_gpu.cu:631-990.
 * fp16 stores round to nearest, ties away from zero, like the CPU backend
 * (include/dali/util/half.hpp:231-243).
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *in;
  int32_t in_h, in_w, channels, in_pitch;
  int32_t anchor_y, anchor_x, crop_h, crop_w;
  int32_t mirror;
  int32_t normalize; /* 0: plain conversion */
  float mean[4], inv_std[4];
  float fill[4];
  int32_t out_channels; /* channels or next_pow2(channels) when pad_output */
  int32_t out_dtype, out_layout;
  void *out;
  int32_t wg_start; /* filled by Setup */
} daliamdCmnDesc;

DALIAMD_API daliamdResult_t daliamdCmnSetup(daliamdCmnDesc *descs_host, int n, int *num_workgroups);
DALIAMD_API daliamdResult_t daliamdCmnRun(daliamdStream_t stream, const daliamdCmnDesc *descs_dev, int n,
                                          int num_workgroups);

/* ----------------------------------------------------------------------------------------------
 * Heavy-augmentation kernels (BASELINE.json configs[2]): warp_affine, gaussian_blur, color_twist, erase.
 * u8 HWC in, u8 HWC out; arithmetic follows the reference's CPU kernels operation by operation:
 *   warp    dali/kernels/imgproc/warp_cpu.h:143-178 + sampler.h:258-338 (incremental source coordinates
 *           re-anchored every 256 px, bilinear s0 + (s1 - s0) * qy, constant or clamp border)
 *   blur    dali/kernels/imgproc/convolution/convolution_cpu.h:241-340 (W pass then H pass, float intermediate,
 *           reflect-101 border, taps accumulated in order)
 *   twist   dali/kernels/imgproc/pointwise/linear_transformation_cpu.h:57-77 (M * px + offset)
 *   erase   dali/kernels/erase/erase_cpu.h (copy + fill of clipped regions)
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *in;
  uint8_t *out;
  int32_t in_h, in_w, channels, in_pitch;
  int32_t out_h, out_w, out_pitch;
  float matrix[6];      /* row-major 2x3, maps DESTINATION (x, y) to SOURCE (x, y) */
  int32_t interp;       /* DALIAMD_INTERP_NN or DALIAMD_INTERP_LINEAR */
  int32_t border_clamp; /* 1: clamp to edge; 0: constant `fill` */
  float fill[4];
  int32_t wg_start;     /* filled by Setup */
  int32_t reserved;
} daliamdWarpAffineDesc;
DALIAMD_API daliamdResult_t daliamdWarpAffineSetup(daliamdWarpAffineDesc *descs_host, int n, int *num_workgroups);
DALIAMD_API daliamdResult_t daliamdWarpAffineRun(daliamdStream_t stream, const daliamdWarpAffineDesc *descs_dev, int n,
                                                 int num_workgroups);

#define DALIAMD_MAX_BLUR_WINDOW 63
typedef struct {
  const uint8_t *in;
  uint8_t *out;
  int32_t h, w, channels, in_pitch, out_pitch;
  int32_t size_x, size_y;                 /* odd window diameters (<= DALIAMD_MAX_BLUR_WINDOW) */
  float window_x[DALIAMD_MAX_BLUR_WINDOW + 1], window_y[DALIAMD_MAX_BLUR_WINDOW + 1];
  int32_t tile_w, tile_h, tiles_x, wg_start, lds_bytes;  /* filled by Setup */
} daliamdGaussianBlurDesc;
/* host helper: FillGaussian (dali/operators/image/convolution/gaussian_blur_params.h:60-83); returns the diameter
 * (2*ceil(3*sigma)+1 when window_size == 0; sigma derived from the window when sigma == 0) or < 0 on error */
DALIAMD_API int daliamdGaussianWindow(float sigma, int window_size, float *window, float *sigma_used);
DALIAMD_API daliamdResult_t daliamdGaussianBlurSetup(daliamdGaussianBlurDesc *descs_host, int n, int *num_workgroups,
                                                    int *lds_bytes);
DALIAMD_API daliamdResult_t daliamdGaussianBlurRun(daliamdStream_t stream, const daliamdGaussianBlurDesc *descs_dev,
                                                  int n, int num_workgroups, int lds_bytes);

#define DALIAMD_MAX_ERASE_REGIONS 8
typedef struct {
  const uint8_t *in;
  uint8_t *out;
  int32_t h, w, channels, in_pitch, out_pitch;
  int32_t transform;    /* 1: out = sat(M * px + offset) (3 channels); 0: copy */
  float matrix[9], offset[3];
  int32_t num_regions;  /* rectangles filled after the transform (erase) */
  int32_t region[DALIAMD_MAX_ERASE_REGIONS][4]; /* y0, x0, y1, x1 (already clipped) */
  float fill[4];
  int32_t wg_start;     /* filled by Setup */
} daliamdPointwiseDesc;
/* host helper: colour-twist matrix/offset (dali/operators/image/color/color_twist.h:50-83,156-170) */
DALIAMD_API void daliamdColorTwistMatrix(float hue, float saturation, float value, float brightness, float contrast,
                                         float *matrix9, float *offset);
DALIAMD_API daliamdResult_t daliamdPointwiseSetup(daliamdPointwiseDesc *descs_host, int n, int *num_workgroups);
DALIAMD_API daliamdResult_t daliamdPointwiseRun(daliamdStream_t stream, const daliamdPointwiseDesc *descs_dev, int n,
                                                int num_workgroups);
/* The blur with a pointwise operator in its write-out: pointwise_dev[i] (transform / regions / fill of sample i; its
 * in / out / pitch fields are not used) is applied to the rounded pixels of sample i before they leave the workgroup -
 * color_twist / erase behind a gaussian_blur cost no launch and no pass over the image.  NULL: the plain blur. */
DALIAMD_API daliamdResult_t daliamdGaussianBlurPointwiseRun(daliamdStream_t stream, const daliamdGaussianBlurDesc *descs_dev,
                                                           int n, int num_workgroups, int lds_bytes,
                                                           const daliamdPointwiseDesc *pointwise_dev);

/* ----------------------------------------------------------------------------------------------
 * Audio features (BASELINE.json configs[3]): spectrogram -> mel filter bank -> decibels, f32.
 *   spectrogram  window extraction (centred, reflect-101 / zero padding) fused with a half-length complex radix-4
 *                Stockham FFT in LDS (one frame per wave, wave-local synchronisation only) and the power / magnitude
 *                spectrum; replaces ExtractWindows* + cuFFT R2C + fft_postprocess
 *                (dali/kernels/signal/window/extract_windows_gpu.cuh:153-302, signal/fft/stft_gpu_impl.cu:116-265)
 *   mel          banded (nfilter x nbins) . (nbins x frames) product: each filter only visits its own bins (2 FMAs
 *                per spectrogram element, bound by reading the spectrogram); weights = the reference's triangular filters
 *                (dali/kernels/audio/mel_scale/mel_scale.h:79-130, mel_filter_bank_cpu.cc:77-111)
 *   decibels     mul * log10(max(min_ratio, x / ref)), ref given or the per-sample maximum (chunk maxima by wave64
 *                shuffles, folded with an atomic max; then the element-wise pass) (dali/kernels/signal/decibel/decibel_calculator.h:25-52, to_decibels_cpu.cc:54-66)
 * Layout: "ft" (frequency-major): spectrogram [nfft/2+1][frames], mel [nfilter][frames].
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  const float *in;      /* device: signal, `length` samples */
  float *out;           /* device: [nfft/2+1][num_windows] */
  int64_t length;
  int32_t num_windows;  /* filled by Setup */
  int32_t wg_start;     /* filled by Setup */
} daliamdSpectrogramDesc;
typedef struct {
  int32_t nfft, window_length, window_step;
  int32_t center_windows, reflect_padding, power; /* power: 1 magnitude, 2 power */
  /* 0: the signals are float32.  1: daliamdSpectrogramDesc.in points to int16 samples that stand for sample / 32768 -
   * 16-bit PCM as decoders.audio finds it in the file (audio_decoder_impl.cc:49-120: libsndfile's float read of PCM16
   * divides by 32768): the conversion happens in the kernel's load, the signal crosses the bus at 2 bytes per sample
   * (round 4; same bits as converting on the host). */
  int32_t input_pcm16;
} daliamdSpectrogramParams;
DALIAMD_API void daliamdHannWindow(int n, float *window);
DALIAMD_API daliamdResult_t daliamdSpectrogramSetup(daliamdSpectrogramDesc *descs_host, int n,
                                                   const daliamdSpectrogramParams *params, int *num_workgroups,
                                                   int *lds_bytes);
/* host helper: nfft floats = nfft/2 complex twiddles exp(-2 pi i k / nfft) */
DALIAMD_API void daliamdSpectrogramTwiddles(int nfft, float *twiddles);
/* twiddles_dev: device copy of that table; with it nfft = 512 / 1024 take the register-resident kernel (4 frames of
 * a wave in flight).  NULL: every size runs the generic kernel, which derives its own twiddles.  lds_bytes: the value
 * Setup returned (an upper bound; Run sizes the launch itself). */
DALIAMD_API daliamdResult_t daliamdSpectrogramRun(daliamdStream_t stream, const daliamdSpectrogramDesc *descs_dev, int n,
                                                 const daliamdSpectrogramParams *params, const float *window_dev,
                                                 const float *twiddles_dev, int num_workgroups, int lds_bytes);

typedef struct {
  const float *in;      /* device: [nbins][frames] */
  float *out;           /* device: [nfilter][frames] */
  int32_t frames;
  int32_t wg_start;     /* filled by Setup */
} daliamdMelDesc;
/* host helper: dense weights [nfilter][nfft/2+1]; mel_formula 0 = slaney, 1 = htk; freq_high <= 0 -> sample_rate/2 */
DALIAMD_API daliamdResult_t daliamdMelFilterBankWeights(int nfilter, int nfft, float sample_rate, float freq_low,
                                                       float freq_high, int normalize, int mel_formula, float *weights);
/* host helper: bands[2m], bands[2m+1] = [first, last+1) bin with a non-zero weight in filter m */
DALIAMD_API daliamdResult_t daliamdMelFilterBankBands(const float *weights_host, int nfilter, int nbins, int32_t *bands);
DALIAMD_API daliamdResult_t daliamdMelFilterBankSetup(daliamdMelDesc *descs_host, int n, int *num_workgroups);
/* bands_dev: device copy of the band table, or NULL to visit every bin of every filter */
DALIAMD_API daliamdResult_t daliamdMelFilterBankRun(daliamdStream_t stream, const daliamdMelDesc *descs_dev, int n,
                                                   int num_workgroups, const float *weights_dev, const int32_t *bands_dev,
                                                   int nfilter, int nbins);

/* The same product for the f32 matrix cores: the filter bank cut into 16-filter x 4-bin tiles in the A-operand lane
 * order of v_mfma_f32_16x16x4_f32 (lane l: W[16 b + l % 16][k0 + 4 t + l / 16]); a row block of 16 filters only owns the
 * tiles between its first and last weighted bin.  tiles = NULL: *num_tiles is the count to allocate (64 floats each);
 * row_blocks: 4 ints per row block {first tile, tile count, first bin, wave of the workgroup that takes it}. */
DALIAMD_API daliamdResult_t daliamdMelFilterBankMfmaLayout(const float *weights_host, int nfilter, int nbins, float *tiles,
                                                          int32_t *row_blocks, int *num_tiles);
/* Spectrogram -> mel filter bank (-> decibels) in ONE launch: the power spectrum of a 16-frame tile is multiplied by the
 * filter bank where it sits in LDS; what reaches HBM is [nfilter][frames] only.  descs as for daliamdSpectrogramRun
 * (Setup included) with out = [nfilter][num_windows]; nfft 512 / 1024.  mfma_tiles != NULL: the product runs on the matrix
 * cores (reference: the filterbank x frames GEMM of dali/kernels/audio/mel_scale/mel_filter_bank_gpu.cu:104-128), else as
 * the banded VALU product of daliamdMelFilterBankRun.  decibels: also apply daliamdToDecibelsRun's formula with the given
 * reference (> 0).  max_bits: optional, per sample i at byte offset i * max_stride: receives the bit pattern of the
 * maximum mel value (zero it first) - for to_decibels(reference = maximum): pass the max_bits field of the uploaded
 * daliamdDecibelDesc table and call daliamdToDecibelsRun with reference = -1 afterwards. */
typedef struct {
  const float *mfma_tiles;
  const int32_t *row_blocks;
  const float *weights;       /* dense [nfilter][nbins] and bands: the VALU variant */
  const int32_t *bands;
  int32_t nfilter, nbins;
  int32_t decibels;
  float multiplier, reference, cutoff_db;
  uint32_t *max_bits;
  int32_t max_stride;
} daliamdSpecMelParams;
DALIAMD_API daliamdResult_t daliamdSpectrogramMelRun(daliamdStream_t stream, const daliamdSpectrogramDesc *descs_dev, int n,
                                                    const daliamdSpectrogramParams *params, const float *window_dev,
                                                    const float *twiddles_dev, const daliamdSpecMelParams *mel,
                                                    int num_workgroups);

typedef struct {
  const float *in;
  float *out;
  int64_t size;
  uint32_t max_bits;    /* device scratch: bit pattern of the sample's maximum (Setup zeroes it; upload the table per run) */
  int32_t wg_start;     /* filled by Setup */
} daliamdDecibelDesc;
DALIAMD_API daliamdResult_t daliamdToDecibelsSetup(daliamdDecibelDesc *descs_host, int n, int *num_workgroups);
/* reference == 0: use the per-sample maximum (1 when that maximum is 0), found by a first kernel that writes
 * descs_dev[i].max_bits; reference < 0: the maximum is already there (daliamdSpectrogramMelRun); `in` may equal `out` */
DALIAMD_API daliamdResult_t daliamdToDecibelsRun(daliamdStream_t stream, daliamdDecibelDesc *descs_dev, int n,
                                                int num_workgroups, float multiplier, float reference, float cutoff_db);

/* MFCC = DCT of the mel spectrogram along the frequency axis (types I-IV as en.wikipedia.org/wiki/Discrete_cosine_transform,
 * optional ortho-normal basis) followed by liftering: out[k][t] = lifter[k] * sum_n table[k][n] * in[n][t]
 * (dali/operators/audio/mfcc/mfcc.cc:24-182, mfcc.h:43-48, dali/kernels/signal/dct/{table.h:26-96,dct_cpu.cc:75-110}).
 * Descriptors / Setup: daliamdMelDesc / daliamdMelFilterBankSetup (in [n_in][frames], out [ndct][frames]). */
DALIAMD_API daliamdResult_t daliamdDctTable(int dct_type, int normalize, int n_in, int ndct, float *table /* [ndct][n_in] */);
DALIAMD_API void daliamdLifterCoeffs(float lifter, int n, float *coeffs);   /* 1 everywhere when lifter == 0 */
DALIAMD_API daliamdResult_t daliamdDctRun(daliamdStream_t stream, const daliamdMelDesc *descs_dev, int n, int num_workgroups,
                                          const float *table_dev, const float *lifter_dev /* or NULL */, int ndct, int n_in);

/* Audio resampling: windowed sinc with a Hann envelope, `lobes` zero crossings on each side, coefficients looked up in a
 * table of lobes * 64 + 1 entries with linear interpolation; output i sits at input position i * in_rate / out_rate
 * (dali/kernels/signal/resampling.h:33-106, resampling_cpu.cc:129-172, dali/operators/audio/resample.h:60-140,
 * resampling_params.h:27-30).  f32 time series, optionally with an innermost channel dimension. */
typedef struct {
  const float *in;      /* device: [in_length][channels] */
  float *out;           /* device: [out_length][channels] */
  int64_t in_length, out_length;
  double in_rate, out_rate;
  int32_t channels;
  int32_t wg_start;     /* filled by Setup */
} daliamdAudioResampleDesc;
DALIAMD_API int daliamdAudioResampleLobes(float quality);   /* 0 -> 3, 50 -> 16, 100 -> 64 */
DALIAMD_API daliamdResult_t daliamdAudioResampleWindow(int lobes, float *lookup_host, int lookup_capacity, int *lookup_size,
                                                      float *scale, float *center);
DALIAMD_API daliamdResult_t daliamdAudioResampleSetup(daliamdAudioResampleDesc *descs_host, int n, int *num_workgroups);
DALIAMD_API daliamdResult_t daliamdAudioResampleRun(daliamdStream_t stream, const daliamdAudioResampleDesc *descs_dev, int n,
                                                   int num_workgroups, const float *lookup_dev, int lookup_size, float scale,
                                                   float center, int lobes);

/* Normalised sample-type conversion around the audio resampler (ConvertSatNorm, include/dali/core/convert.h:262-350;
 * the input rules of dali/operators/audio/resample.cc:160-192): integers <-> floats in [-1, 1] (signed) or [0, 1]
 * (unsigned): f = in / max(in type); mode 1: f = (f + 1) * 0.5 (signed source to an unsigned result), mode 2:
 * f = f * 2 - 1 (unsigned source to a signed result); out = clamp(round_half_away(f * max(out type))) or f itself.
 * Types: int8 / uint8 / int16 / uint16 / int32 / uint32 / float. */
typedef struct {
  const void *in;
  void *out;
  int64_t count;
  int32_t wg_start;     /* filled by Setup */
  int32_t reserved;
} daliamdConvertNormDesc;
DALIAMD_API daliamdResult_t daliamdConvertNormSetup(daliamdConvertNormDesc *descs_host, int n, int *num_workgroups);
DALIAMD_API daliamdResult_t daliamdConvertNormRun(daliamdStream_t stream, const daliamdConvertNormDesc *descs_dev, int n,
                                                 int num_workgroups, int in_dtype, int out_dtype, int mode);

/* ----------------------------------------------------------------------------------------------
 * fn.normalize: out = (in - mean) * scale / stddev + shift with mean / stddev given or computed over a contiguous group
 * of axes (dali/operators/math/normalize/normalize.cc:24-123, normalize_utils.h:133-220,
 * dali/kernels/normalize/normalize_cpu.h:38-70, dali/kernels/reduce/mean_stddev_gpu_impl.cuh).
 * A sample is viewed as [outer][reduced][inner]; one statistic per (outer, inner) pair.  Sums are accumulated in
 * fp64 (exact for uint8 input).  Batch normalisation: every descriptor points at the SAME sum/mean/inv_std arrays,
 * stat_count is the total over the batch and only one descriptor has owns_stats = 1.
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  const void *in;               /* device, dense; uint8 or float                                        */
  void *out;                    /* device, dense, same shape                                            */
  int64_t outer, reduced, inner;
  double *sum_mean, *sum_var;   /* device accumulators [outer*inner], zero-filled by the caller         */
  float *mean, *inv_std;        /* device [outer*inner]: results of the reductions (unused with scalars) */
  double stat_count;            /* elements each statistic is taken over                                */
  float scalar_mean, scalar_inv_std; /* used when use_scalar_* is set: inv_std = scale / stddev          */
  int32_t use_scalar_mean, use_scalar_inv_std;
  int32_t in_dtype, out_dtype;  /* daliamdDType_t                                                       */
  int32_t owns_stats;           /* 1: this descriptor finalises the accumulators it points at           */
  int32_t stat_wg_start, stat_chunks, apply_wg_start; /* filled by Setup                                */
} daliamdNormalizeDesc;

DALIAMD_API daliamdResult_t daliamdNormalizeSetup(daliamdNormalizeDesc *descs_host, int n, int *stat_workgroups,
                                                  int *apply_workgroups, int64_t *max_bins);
/* calc_mean / calc_stddev: run the reductions (sum -> mean; sum of squared deviations -> scale / sqrt(var / (N - ddof)
 * + epsilon), 0 where that is not finite, like the reference); then the element-wise pass. */
DALIAMD_API daliamdResult_t daliamdNormalizeRun(daliamdStream_t stream, const daliamdNormalizeDesc *descs_dev, int n,
                                                int stat_workgroups, int apply_workgroups, int64_t max_bins,
                                                int calc_mean, int calc_stddev, int ddof, float epsilon, float scale,
                                                float shift);

#ifdef __cplusplus
}
#endif
#endif /* DALI_AMD_KERNELS_H_ */
