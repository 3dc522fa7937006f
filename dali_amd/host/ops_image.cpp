// Image operators of the hot path, host side.  All arithmetic runs in libdali_amd_kernels.so; these
// classes do what the reference's operator shells do: argument handling, shape inference, random
// parameter generation, and launching the batched kernels.
//   decoders.image (mixed)   dali/operators/imgcodec/image_decoder.h:131-933, decoder_schema.cc:22-151
//   RandomResizedCrop        dali/operators/image/resize/random_resized_crop.{h,cc,cu}
//   CropMirrorNormalize      dali/operators/image/crop/crop_mirror_normalize.{h,cc}, new_crop_mirror_normalize.cu
//   CropAttr                 dali/operators/image/crop/crop_attr.cc:21-240
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>

#include "dali_amd_host.h"
#include "image_cache.h"
#include "ops.h"
#include "pipeline.h"

namespace daliamd_host {

static constexpr int kImagePitchAlign = 16;  // row pitch of device images handed between our own operators

// DALI enum values (include/dali/core/common.h:144-165)
enum { DALI_INTERP_NN = 0, DALI_INTERP_LINEAR = 1, DALI_INTERP_CUBIC = 2, DALI_INTERP_LANCZOS3 = 3,
       DALI_INTERP_TRIANGULAR = 4, DALI_INTERP_GAUSSIAN = 5 };
enum { DALI_RGB = 0, DALI_BGR = 1, DALI_GRAY = 2, DALI_YCbCr = 3, DALI_ANY_DATA = 4 };

static int ToKernelInterp(int64_t dali_interp) {
  switch (dali_interp) {  // DALIInterpType (include/dali/core/common.h): NN 0, LINEAR 1, CUBIC 2, LANCZOS3 3, TRIANGULAR 4, GAUSSIAN 5
    case 0: return DALIAMD_INTERP_NN;
    case DALI_INTERP_LINEAR: return DALIAMD_INTERP_LINEAR;
    case 2: return DALIAMD_INTERP_CUBIC;
    case 3: return DALIAMD_INTERP_LANCZOS3;
    case DALI_INTERP_TRIANGULAR: return DALIAMD_INTERP_TRIANGULAR;
    case 5: return DALIAMD_INTERP_GAUSSIAN;
    default: DALI_FAIL("Unknown interpolation type ", dali_interp);
  }
}

// =============================================================================================
// decoders.image, device="mixed": host parse + Huffman (thread pool, pinned staging) -> H2D ->
// dequant/IDCT -> upsample + colour on the GPU.
// =============================================================================================
DALI_SCHEMA(decoders__Image)
    .DocStr("Decodes images.\n\nSupported format in this MI355X-native build: JPEG (baseline and progressive, 8-bit, "
            "grayscale / YCbCr / RGB).  Baseline single-scan streams without restart markers are decoded entirely on "
            "the GPU (Huffman included); the others are entropy-decoded on the host thread pool.  Dequantisation, "
            "inverse DCT, chroma upsampling and colour conversion always run on the GPU and are bit-exact with "
            "libjpeg-turbo's accurate integer path, i.e. with DALI's CPU backend.\n\nThe output is in HWC layout.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("output_type", "The color space of the output image: RGB (default), BGR, GRAY (one channel: the luma plane "
                    "of a YCbCr / grayscale stream, 0.299 R + 0.587 G + 0.114 B otherwise), YCbCr (ITU-R BT.601 with head room) or "
                    "ANY_DATA (grayscale streams stay one channel, everything else RGB); CMYK / YCCK streams are converted to RGB "
                    "first.", ArgValue::Int(DALI_RGB))
    .AddOptionalArg("dtype", "Output data type.", ArgValue::Int(DALI_UINT8))
    .AddOptionalArg("adjust_orientation", "Use EXIF orientation metadata to rectify the images.", ArgValue::Bool(true))
    .AddOptionalArg("use_fast_idct", "Ignored: the accurate integer IDCT is always used.", ArgValue::Bool(false))
    .AddOptionalArg("jpeg_fancy_upsampling", "Fancy (triangle) chroma upsampling is always used, like DALI's CPU backend.",
                    ArgValue::Bool(false))
    .AddOptionalArg("hybrid_huffman_threshold", "When given explicitly: images with fewer pixels (H*W) are "
                    "Huffman-decoded on the host; by default every supported stream is entropy-decoded on the GPU.",
                    ArgValue::Int(1000000))
    .AddOptionalArg("device_memory_padding", "Ignored.", ArgValue::Int(16 * 1024 * 1024))
    .AddOptionalArg("host_memory_padding", "Ignored.", ArgValue::Int(8 * 1024 * 1024))
    .AddOptionalArg("device_memory_padding_jpeg2k", "Ignored.", ArgValue::Int(0))
    .AddOptionalArg("host_memory_padding_jpeg2k", "Ignored.", ArgValue::Int(0))
    .AddOptionalArg("hw_decoder_load", "Ignored (no fixed-function JPEG engine is used).", ArgValue::Float(0.65))
    .AddOptionalArg("preallocate_width_hint", "Ignored.", ArgValue::Int(0))
    .AddOptionalArg("preallocate_height_hint", "Ignored.", ArgValue::Int(0))
    .AddOptionalArg("affine", "Ignored.", ArgValue::Bool(true))
    .AddOptionalArg("split_stages", "Deprecated, ignored.", ArgValue::Bool(false))
    .AddOptionalArg("use_chunk_allocator", "Deprecated, ignored.", ArgValue::Bool(false))
    .AddOptionalArg("memory_stats", "Deprecated, ignored.", ArgValue::Bool(false))
    .AddOptionalArg("cache_size", "Total size of the decoder cache in megabytes. When provided, the decoded images that "
                    "are larger than `cache_threshold` will be cached in GPU memory.", ArgValue::Int(0))
    .AddOptionalArg("cache_type", "``threshold``: caches every image with a size that is larger than `cache_threshold` "
                    "until the cache is full (warm-up: 1 epoch). ``largest``: stores the largest images that can fit in "
                    "the cache (warm-up: 2 epochs). ``encoded`` (MI355X extension): keeps the entropy-coded segment and "
                    "the parse results of every JPEG the GPU entropy decoder takes resident in GPU memory instead of "
                    "the decoded pixels - from the second epoch on a sample needs no file read (with the reader's "
                    "``skip_cached_images``), no header parse and no host-to-device transfer, and is still decoded anew "
                    "(warm-up: 1 epoch).  ``indexed`` (MI355X extension): like ``encoded``, and the stream is kept UN-STUFFED "
                    "together with the decoder state in front of each of its 256-byte slices (12 bytes per slice, 5 % of the "
                    "stream) as its first decode found them: later decodes skip the un-stuffing, the parallel "
                    "synchronisation and the DC pass, and a window decode skips the slices outside the window.  Same pixels.",
                    ArgValue::Str(""))
    .AddOptionalArg("cache_threshold", "The size threshold, in bytes, for decoded images to be cached.", ArgValue::Int(0))
    .AddOptionalArg("cache_debug", "Prints the debug information about the decoder cache.", ArgValue::Bool(false))
    .AddOptionalArg("cache_batch_copy", "Accepted for compatibility: cached images are handed out in place, there is no "
                    "copy to batch.", ArgValue::Bool(true))
    .InputLayout(0, {""});
DALI_SCHEMA(ImageDecoder).DocStr("Legacy alias of decoders.image").NumInput(1).NumOutput(1).AddParent("decoders__Image");
DALI_SCHEMA(experimental__decoders__Image).DocStr("Alias of decoders.image").NumInput(1).NumOutput(1).AddParent("decoders__Image");

class ImageDecoderMixed : public OperatorBase {
 public:
  explicit ImageDecoderMixed(const OpSpec &spec, bool allow_cache = true) : OperatorBase(spec) {
    int64_t ot = spec.GetInt("output_type");
    DALI_ENFORCE(ot == DALI_RGB || ot == DALI_BGR || ot == DALI_GRAY || ot == DALI_YCbCr || ot == DALI_ANY_DATA,
                 "decoders.image: unsupported output_type ", ot);
    // ANY_DATA: colour streams come out as RGB (a grayscale stream as well here: one channel count per batch)
    out_type_ = ot == DALI_ANY_DATA ? (int)DALI_RGB : (int)ot;
    oc_ = out_type_ == DALI_GRAY ? 1 : 3;
    DALI_ENFORCE(spec.GetInt("dtype") == DALI_UINT8, "decoders.image: only dtype=UINT8 is supported");
    adjust_orientation_ = spec.GetBool("adjust_orientation");
    device_id_ = (int)spec.GetInt("device_id");
    // Entropy decoding runs on the GPU for every stream the kernel supports.  An EXPLICIT hybrid_huffman_threshold
    // keeps the reference's meaning: streams with fewer pixels than that are Huffman-decoded on the host.
    if (spec.Args().count("hybrid_huffman_threshold")) huffman_threshold_ = spec.GetInt("hybrid_huffman_threshold");
    if (const char *env = getenv("DALI_AMD_HOST_HUFFMAN")) host_huffman_only_ = atoi(env) != 0;
    // DALI_AMD_FUSE_COLOR=1: upsampling + colour conversion inside the entropy decoder's block kernel (no component planes,
    // no colour launch; bit-identical).  Off by default: measured on MI355X it moves 25 % fewer bytes and is 10 % SLOWER
    // (397 000 against 443 000 images/s, gpurun_out/r04c3) - the step is bound by instruction issue, not by HBM, the fused
    // kernel issues the same colour arithmetic and holds 62 KB of LDS + 191 registers per workgroup (HISTORY.md section 9).
    if (const char *env = getenv("DALI_AMD_FUSE_COLOR")) fuse_color_ = atoi(env) != 0;
    ring_ = (int)spec.GetInt("gpu_prefetch_queue_depth") + 1;
    for (int i = 0; i < ring_; i++) {
      staging_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
      coef_dev_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      planes_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      ecs_stage_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
      ecs_dev_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      scratch_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      status_host_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
      raster_stage_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
    }
    h2d_done_.assign(ring_, nullptr);
    front_done_.assign(ring_, nullptr);
    // DALI_AMD_DECODER_FRONT_ON_SIDE_STREAM=1: un-stuffing + table build on the executor's side stream, behind the transfer.
    // Measured on MI355X (gpurun_out/r05_g, resident headline over 200 steps): 402-403 k images/s against 500-505 k with
    // the front on the compute stream - the side stream already carries the resampling tables of the same iteration, the
    // two queue up behind each other there and the compute stream waits for both.  Off by default.
    if (const char *env = getenv("DALI_AMD_DECODER_FRONT_ON_SIDE_STREAM")) keep_front_on_compute_ = atoi(env) == 0;
    // decoded-image cache (cached_decoder_impl.cc:24-48); the fused crop decoders have no cache options
    if (spec.Args().count("cache_size") && (spec.GetString("cache_type") == "encoded" || spec.GetString("cache_type") == "indexed")) {
      // the encoded-stream cache (image_cache.h); also for the region-of-interest decoders: they decode from it
      const size_t bytes = (size_t)spec.GetInt("cache_size") * 1024 * 1024;
      if (bytes > 0) stream_cache_ = StreamCache::Get((int)spec.GetInt("device_id"), bytes, spec.GetBool("cache_debug"));
      // "indexed": a stream becomes resident together with the side information its first decode found (un-stuffed bytes,
      // the decoder state in front of every 256-byte slice) and is decoded from that afterwards
      index_streams_ = spec.GetString("cache_type") == "indexed";
    } else if (allow_cache && spec.Args().count("cache_size")) {
      const size_t bytes = (size_t)spec.GetInt("cache_size") * 1024 * 1024;
      const size_t threshold = (size_t)spec.GetInt("cache_threshold");
      if (bytes > 0 && bytes >= threshold)
        cache_ = ImageCache::Get((int)spec.GetInt("device_id"),
                                 ImageCache::Params{spec.GetString("cache_type"), bytes, threshold, spec.GetBool("cache_debug")});
    }
  }
  ~ImageDecoderMixed() override {
    for (auto e : h2d_done_)
      if (e) daliamdEventDestroy(e);
    for (auto e : front_done_)
      if (e) daliamdEventDestroy(e);
    if (trace_ && trace_runs_ > 0) {
      static const char *names[] = {"parse + scan analysis + staging copy (thread pool)", "layout: plans, scratch sizes",
                                    "host entropy decode of the other streams", "descriptor tables",
                                    "transfer + kernel launches", "layout: windows (the consumer's draw)",
                                    "layout: buffers, output resize"};
      for (int i = 0; i < 7; i++)
        fprintf(stderr, "[dali_amd trace]     decoder: %-52s %8.3f ms\n", names[i], 1e3 * trace_s_[i] / trace_runs_);
    }
  }
  int OutputPitchAlign(int) const override { return kImagePitchAlign; }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }

  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    const int n = in.num_samples();
    DALI_ENFORCE(in.type() == DALI_UINT8, "decoders.image expects encoded streams as 1-D uint8 tensors");
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](int phase) {  // DALI_AMD_TRACE=1: host time of the phases of this operator
      if (!trace_) return;
      auto t = std::chrono::steady_clock::now();
      trace_s_[phase] += std::chrono::duration<double>(t - t_last).count();
      t_last = t;
    };
    // DALI_AMD_TRACE_SKIP=N: the first N runs (set-up epochs: file reads, first decodes, buffer growth) are not counted
    static const int64_t trace_skip = getenv("DALI_AMD_TRACE_SKIP") ? atoll(getenv("DALI_AMD_TRACE_SKIP")) : 0;
    if (trace_ && ++trace_seen_ == trace_skip) { for (double &t : trace_s_) t = 0; trace_runs_ = 0; }
    trace_runs_++;
    infos_.resize(n);
    scans_.resize(n);
    tscans_.resize(n);
    if (transcoded_.size() != (size_t)ring_) transcoded_.resize(ring_);
    auto src = [&](int i) { return i < (int)in.source_info.size() && !in.source_info[i].empty() ? in.source_info[i]
                                                                                               : make_string("sample #", i); };
    // ---- one thread-pool pass per sample: header parse, scan analysis and (GPU path) the copy of the entropy-coded
    // bytes into the pinned staging buffer.  Staging offsets come from the FILE sizes, which are known up front, so
    // no second pass is needed once the segment lengths are known (cost: the transfer is longer by the few hundred
    // header bytes of each file).
    const int slot = (int)(ws.iteration % ring_);
    transcoded_[slot].resize(n);
    for (auto &t : transcoded_[slot]) t.clear();
    Buffer &ecs_stage = *ecs_stage_[slot];
    // ---- decoded-image cache: a hit needs no work at all, its output sample IS the cache entry ----
    hit_.assign(n, 0);
    cached_.assign(n, ImageCache::Entry{});
    int nact = n;
    if (cache_) {
      for (int i = 0; i < n; i++) {
        if (i < (int)in.source_info.size() && cache_->Lookup(in.source_info[i], &cached_[i], ws.stream)) {
          hit_[i] = 1;
          nact--;
        }
      }
    }
    // ---- encoded-stream cache: a resident sample brings its parse results and its entropy-coded segment (in HBM);
    // the input bytes are not looked at (the reader may have skipped the file: they are empty then) ----
    structural_.assign(n, 0);
    boxed_.assign(n, 0);
    erec_.assign(n, nullptr);
    int nehit = 0;
    if (stream_cache_) nehit = stream_cache_->Lookup(in.source_info, hit_, &erec_, ws.stream, ws.aux_stream);
    // Samples that are not JPEG (PNG, BMP, PNM): decoded on the host thread pool further down and uploaded; for the JPEG
    // machinery they do not exist, like cache hits.
    raster_.assign(n, 0);
    int nraster = 0;
    for (int i = 0; i < n; i++) {
      const uint8_t *b = static_cast<const uint8_t *>(in.raw(i));
      if (!hit_[i] && !erec_[i] && !(in.nbytes(i) >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF) &&
          !daliamdJpegIndexedIs(b, in.nbytes(i))) {
        raster_[i] = 1;
        hit_[i] = 2;  // skipped by every JPEG loop
        nraster++;
        nact--;
      }
    }
    raster_hw_.assign(2 * (size_t)n, 0);
    // "raster residents" (round 6): samples of formats the device cannot decode, kept in the encoded-stream cache as their
    // decoded upright image - handed out in place (a window of one: a view), no host decode, no upload, no file read
    rres_.assign(n, nullptr);
    rkeep_.assign(n, nullptr);
    for (int i = 0; i < n && nehit > 0; i++) {
      if (!erec_[i] || !erec_[i]->pixels) continue;
      DALI_ENFORCE(erec_[i]->image_type == out_type_, "Failed to decode ", src(i), ": it is resident as a decoded image of another "
                   "output_type (decoders that share a device's encoded cache must agree on output_type for non-JPEG samples)");
      rres_[i] = std::move(erec_[i]);
      erec_[i] = nullptr;
      raster_[i] = 3;
      raster_hw_[2 * i] = rres_[i]->h; raster_hw_[2 * i + 1] = rres_[i]->w;
      hit_[i] = 2;
      nehit--;
      nact--;
    }
    // The reader's output already sits in page-locked memory (Buffer::Reserve): the whole block is transferred as it
    // is and the entropy-coded segments are addressed inside it - no staging copy of the JPEG bytes.  (With cache
    // hits in the batch, or input from elsewhere, the segments of the active samples are packed into the staging
    // buffer as before.)
    const bool direct = nact == n && nehit == 0 && n > 0 && in.device() == StorageDevice::CPU && in.pinned() &&
                        !in.has_external();
    // Round 5: the reader handed out its file mappings, registered with the device (TensorList::ext_device_visible) - the
    // device fetches the entropy-coded segments itself (daliamdGatherCopy on the copy stream, where the transfer was): the
    // bytes cross the bus once and no host core touches them; only the descriptor tables are still uploaded.  All or
    // nothing: a batch with a sample that is not in such a mapping yet (its first sighting) is staged as before.
    bool gather = !direct && nact > nehit && in.device() == StorageDevice::CPU && in.ext_device_visible();
    for (int i = 0; i < n && gather; i++)
      if (!hit_[i] && !erec_[i] && !in.is_external(i)) gather = false;
    ecs_off_.assign(n, 0);
    size_t ecs_bytes = 0;
    for (int i = 0; i < n; i++) {
      ecs_off_[i] = direct ? (size_t)(static_cast<const uint8_t *>(in.raw(i)) - static_cast<const uint8_t *>(in.base()))
                           : ecs_bytes;
      if (!hit_[i] && !erec_[i]) ecs_bytes += ((size_t)in.nbytes(i) + 63) & ~(size_t)63;   // (64: an indexed container's entry)
    }
    if (direct) ecs_bytes = (in.total_bytes() + 15) & ~(size_t)15;
    // the three descriptor tables of the iteration live behind the JPEG bytes in the same staging buffer, so that
    // ONE host->device copy (on the copy stream) carries everything the kernels need
    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t table_bytes = align16((size_t)n * sizeof(daliamdJpegHuffDesc)) + align16((size_t)n * 3 * sizeof(daliamdJpegIdctDesc)) +
                               align16((size_t)n * sizeof(daliamdJpegColorDesc)) + align16((size_t)n * sizeof(daliamdGatherDesc));
    ecs_stage.Reserve((direct || gather ? 0 : ecs_bytes) + table_bytes + 256);
    for (int i = 0; i < n; i++) {
      if (hit_[i]) {
        infos_[i] = daliamdJpegInfo{};  // no components: every per-component loop below skips the sample
        scans_[i].eligible = 0;
        if (raster_[i] == 1) {
          daliamdImageFormat fmt;
          if (daliamdImageProbe(static_cast<const uint8_t *>(in.raw(i)), in.nbytes(i), &fmt, &raster_hw_[2 * i + 1],
                                &raster_hw_[2 * i]) != 0)
            DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
        }
        continue;
      }
      if (erec_[i]) {
        infos_[i] = erec_[i]->info;
        continue;
      }
      ws.GetThreadPool().AddWork([&, i](int) {
        const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
        // a file this process has parsed before (epoch >= 2 of a shard that is not resident): what its headers said
        const bool named = i < (int)in.source_info.size() && !in.source_info[i].empty();
        if (daliamdJpegIndexedIs(data, in.nbytes(i))) {
          // Indexed container (tools/jpeg2idx.py, host/jpeg_indexed.cpp): the headers of the JPEG + the index entry of its
          // entropy-coded segment, made offline.  The entry goes where the segment would go and the stream is decoded FROM it
          // (daliamdJpegHuffDesc.index) - in the first epoch, in a cold process: the position passes never run for it.
          daliamdJpegIndexedView view;
          if (daliamdJpegIndexedParse(data, in.nbytes(i), &view) != 0 ||
              daliamdJpegAnalyzeHeader(view.header, (size_t)view.header_len, &infos_[i], &scans_[i]) != 0)
            DALI_FAIL("Failed to parse ", src(i), ": ", daliamdHostGetLastErrorMessage());
          DALI_ENFORCE(scans_[i].eligible && scans_[i].restart_interval == 0 && !host_huffman_only_, "Failed to decode ", src(i),
                       ": an indexed JPEG container holds a stream for the GPU entropy decoder"
                       " (not usable with device=\"cpu\" semantics / hybrid_huffman_threshold forcing the host decoder)");
          if (daliamdJpegIndexedValidate(data, in.nbytes(i), &view, scans_[i].blocks_per_mcu,
                                         scans_[i].mcus_x * scans_[i].mcus_y * scans_[i].blocks_per_mcu) != 0)
            DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
          scans_[i].ecs_offset = view.index_offset;
          scans_[i].ecs_length = view.ecs_len;
          scans_[i].length_is_upper_bound = 0;
          boxed_[i] = view.index_bytes;
          if (!direct && !gather)
            memcpy(static_cast<uint8_t *>(ecs_stage.data()) + ecs_off_[i], data + view.index_offset, (size_t)view.index_bytes);
          return;
        }
        const bool known = named && !host_huffman_only_ && HeaderCache::Find(in.source_info[i], data, in.nbytes(i), &infos_[i], &scans_[i]);
        // ONE pass over the headers, up to SOS: frame geometry + what the GPU entropy decoder needs.  The scan itself is
        // not walked here - its end (the first marker that is not RSTn) is found by the un-stuffing kernel, which
        // looks at every byte anyway (the memchr walk was two thirds of this operator's host time per sample).
        if (known) {
        } else if (host_huffman_only_ || daliamdJpegAnalyzeHeader(data, in.nbytes(i), &infos_[i], &scans_[i]) != 0) {
          scans_[i].eligible = 0;  // (a broken table or SOS header: the host decoder will produce the diagnosis)
          if (daliamdJpegParse(data, in.nbytes(i), &infos_[i]) != 0)
            DALI_FAIL("Failed to parse ", src(i), ": ", daliamdHostGetLastErrorMessage());
        } else if (named) {
          HeaderCache::Put(in.source_info[i], data, in.nbytes(i), infos_[i], scans_[i]);
        }
        if (infos_[i].num_components == 4) { scans_[i].eligible = 0; return; }  // CMYK / YCCK: the host decodes these (below)
        DALI_ENFORCE(infos_[i].num_components == 1 || infos_[i].num_components == 3, "Failed to decode ", src(i),
                     ": JPEG with ", infos_[i].num_components, " components is not supported");
        // (a stream the kernels do not take because of its STRUCTURE - progressive, several scans ... - can be kept resident
        // re-encoded, below; one that stays on the host because of the caller's threshold cannot)
        structural_[i] = !scans_[i].eligible && !host_huffman_only_;
        if ((int64_t)infos_[i].width * infos_[i].height < huffman_threshold_) scans_[i].eligible = 0;
        if (scans_[i].eligible && !direct && !gather)
          memcpy(static_cast<uint8_t *>(ecs_stage.data()) + ecs_off_[i], data + scans_[i].ecs_offset,
                 (size_t)scans_[i].ecs_length);
      }, (int64_t)in.nbytes(i));
    }
    if (nact > nehit) ws.GetThreadPool().RunAll();
    // the scan analysis of sample i: its own, or the resident one
    auto scan = [&](int i) -> const daliamdJpegScan & { return erec_[i] ? erec_[i]->scan : scans_[i]; };
    // Four-component streams (CMYK / YCCK; real ImageNet holds 22): not for the kernels - they join the samples the
    // host decodes and uploads (daliamdJpegDecodeHost), and stop existing for the JPEG machinery
    jpeg4_.assign(n, daliamdJpegInfo{});
    for (int i = 0; i < n; i++) {
      if (hit_[i] || infos_[i].num_components != 4) continue;
      jpeg4_[i] = infos_[i];
      const bool swap = adjust_orientation_ && infos_[i].orientation >= 5;
      raster_hw_[2 * i] = swap ? infos_[i].width : infos_[i].height;
      raster_hw_[2 * i + 1] = swap ? infos_[i].height : infos_[i].width;
      infos_[i] = daliamdJpegInfo{};
      raster_[i] = 2;
      hit_[i] = 2;
      nraster++;
      nact--;
    }
    lap(0);
    // ---- layout ----
    std::vector<TensorShape> shapes(n);
    coef_off_.assign(n * 3, 0);
    scratch_off_.assign(n, 0);
    gpu_samples_.clear();
    int64_t elems = 0;
    size_t scratch_bytes = 0;
    int ncomp_total = 0;
    // region-of-interest decode (decoders.image_crop / image_random_crop): windows of the upright images
    upright_hw_.resize(2 * n);
    for (int i = 0; i < n; i++) {
      bool swap = adjust_orientation_ && infos_[i].orientation >= 5;
      upright_hw_[2 * i] = swap ? infos_[i].width : infos_[i].height;
      upright_hw_[2 * i + 1] = swap ? infos_[i].height : infos_[i].width;
      if (raster_[i]) { upright_hw_[2 * i] = raster_hw_[2 * i]; upright_hw_[2 * i + 1] = raster_hw_[2 * i + 1]; }
    }
    rois_.assign(4 * n, 0);
    lap(1);
    ComputeRois(ws, n);
    lap(5);
    // Windows drawn for this iteration belong to it: when the decode fails on the way out (a corrupt stream met by a host
    // decoder, an oversized segment, a kernel-library error) the consumer never runs for this iteration, so the draw is
    // taken back - the next iteration then gets the windows it would have got without the fusion
    struct RoiDraw {
      std::function<void(int64_t)> undo;
      int64_t iteration;
      bool armed;
      ~RoiDraw() { if (armed && undo) undo(iteration); }
    } roi_draw{roi_undo_, (int64_t)ws.iteration, roi_source_ != nullptr};
    plans_.assign(n, daliamdJpegRoiPlan{});
    std::vector<void *> ext_ptr(cache_ || stream_cache_ ? n : 0, nullptr);
    std::vector<int64_t> ext_pitch(cache_ || stream_cache_ ? n : 0, 0);
    struct ReservedRasters {   // raster slots reserved in this run; handed back unless their uploads get enqueued and committed
      StreamCache *cache;
      std::vector<std::string> keys;
      ~ReservedRasters() { if (cache && !keys.empty()) cache->Abandon(keys); }
    } reserved_rasters{stream_cache_.get(), {}};
    // slots reserved in this run; dropped again unless the decodes get enqueued (an exception on the way out)
    struct Reserved {
      ImageCache *cache;
      std::vector<std::string> keys;
      ~Reserved() { for (auto &k : keys) cache->Invalidate(k); }
    } reserved{cache_.get(), {}};
    for (int i = 0; i < n; i++) {
      const auto &inf = infos_[i];
      if (raster_[i]) {
        const bool window = rois_[4 * i + 2] > 0;
        shapes[i] = {window ? rois_[4 * i + 2] : upright_hw_[2 * i], window ? rois_[4 * i + 3] : upright_hw_[2 * i + 1], oc_};
        const int wy = window ? rois_[4 * i] : 0, wx = window ? rois_[4 * i + 1] : 0;
        if (raster_[i] == 3) {   // resident: the output sample is (the window of) the resident image
          DALI_ENFORCE(rres_[i]->c == oc_, "Failed to decode ", src(i), ": resident decoded image has ", rres_[i]->c, " channels");
          ext_ptr[i] = const_cast<uint8_t *>(rres_[i]->pixels) + (size_t)wy * rres_[i]->pitch + (size_t)wx * oc_;
          ext_pitch[i] = rres_[i]->pitch;
        } else if (stream_cache_ && raster_residents_ && i < (int)in.source_info.size() && !in.source_info[i].empty()) {
          // becomes resident now: the WHOLE upright image is decoded into the cache slot, this iteration's sample is a view
          const int64_t pitch = ((int64_t)upright_hw_[2 * i + 1] * oc_ + kImagePitchAlign - 1) / kImagePitchAlign * kImagePitchAlign;
          if (uint8_t *slot_ptr = stream_cache_->Reserve(in.source_info[i], (size_t)upright_hw_[2 * i] * (size_t)pitch)) {
            reserved_rasters.keys.push_back(in.source_info[i]);
            rkeep_[i] = slot_ptr;
            ext_ptr[i] = slot_ptr + (size_t)wy * pitch + (size_t)wx * oc_;
            ext_pitch[i] = pitch;
          }
        }
        continue;
      }
      if (hit_[i]) {
        shapes[i] = {cached_[i].h, cached_[i].w, cached_[i].c};
        ext_ptr[i] = cached_[i].data;
        ext_pitch[i] = cached_[i].pitch;
        continue;
      }
      shapes[i] = {upright_hw_[2 * i], upright_hw_[2 * i + 1], oc_};
      if (cache_ && i < (int)in.source_info.size()) {
        // a miss the policy keeps is decoded straight into its cache slot
        const int64_t pitch = (shapes[i][1] * oc_ + kImagePitchAlign - 1) / kImagePitchAlign * kImagePitchAlign;
        if (uint8_t *slot_ptr = cache_->Reserve(in.source_info[i], (int)shapes[i][0], (int)shapes[i][1], oc_, pitch)) {
          ext_ptr[i] = slot_ptr;
          ext_pitch[i] = pitch;
          reserved.keys.push_back(in.source_info[i]);
        }
      }
      if (rois_[4 * i + 2] > 0) {
        if (daliamdJpegPlanRoi(inf.width, inf.height, inf.num_components, inf.h_samp, inf.v_samp,
                               adjust_orientation_ ? inf.orientation : 1, rois_[4 * i], rois_[4 * i + 1], rois_[4 * i + 2],
                               rois_[4 * i + 3], &plans_[i]) != DALIAMD_SUCCESS)
          DALI_FAIL("Failed to decode ", src(i), ": ", daliamdGetLastErrorMessage());
        shapes[i] = {rois_[4 * i + 2], rois_[4 * i + 3], oc_};
      }
      for (int c = 0; c < inf.num_components; c++) {
        coef_off_[i * 3 + c] = elems;
        elems += inf.coef_elems[c];
        ncomp_total++;
      }
      if (scan(i).eligible) {
        DALI_ENFORCE(scan(i).ecs_length < (int64_t)1 << 30, "Failed to decode ", src(i), ": entropy-coded segment too long");
        size_t need = 0;
        const int mcus = scan(i).mcus_x * scan(i).mcus_y, ri = scan(i).restart_interval;
        KCHECK(daliamdJpegHuffmanScratchBytesRestart((int)scan(i).ecs_length, mcus * scan(i).blocks_per_mcu,
                                                     ri > 0 ? (mcus + ri - 1) / ri : 0, &need));
        scratch_off_[i] = scratch_bytes;
        scratch_bytes += need;
        gpu_samples_.push_back(i);
      }
    }
    const int ngpu = (int)gpu_samples_.size();
    lap(1);
    Buffer &stage = *staging_[slot], &cdev = *coef_dev_[slot], &planes = *planes_[slot];
    Buffer &ecs_dev = *ecs_dev_[slot], &scratch = *scratch_[slot];
    Buffer &status_host = *status_host_[slot];
    if (ngpu < nact) stage.Reserve((size_t)elems * 2 + 256);
    cdev.Reserve((size_t)elems * 2 + 256);
    planes.Reserve((size_t)elems + 256);
    ecs_dev.Reserve(ecs_bytes + table_bytes + 256);
    scratch.Reserve(scratch_bytes + 256);
    status_host.Reserve(sizeof(int32_t) * (size_t)std::max(n, 1));
    {
      // windows change from iteration to iteration: the output is sized for the whole images once, so that a batch with a
      // new record of window bytes does not re-allocate device memory (a synchronising call) in the middle of an epoch
      size_t whole = 0;
      bool windows = false;
      for (int i = 0; i < n; i++) {
        windows = windows || rois_[4 * i + 2] > 0;
        const size_t row = ((size_t)upright_hw_[2 * i + 1] * oc_ + kImagePitchAlign - 1) / kImagePitchAlign * kImagePitchAlign;
        whole += (row * (size_t)upright_hw_[2 * i] + 255) & ~(size_t)255;
      }
      if (windows && !cache_) out.SetMinReserve(whole);
    }
    out.Resize(shapes, DALI_UINT8, kImagePitchAlign, ext_ptr, ext_pitch,
               cache_ ? std::shared_ptr<void>(cache_) : std::shared_ptr<void>(stream_cache_));
    out.SetLayout("HWC");
    out.source_info = in.source_info;
    quant_.assign((size_t)n * 3 * 64, 0);
    lap(6);
    // ---- host entropy decode of the streams the GPU kernel does not take (thread pool) ----
    int16_t *coef_host = static_cast<int16_t *>(stage.data());
    for (int i = 0; i < n; i++) {
      if (hit_[i]) continue;
      if (scan(i).eligible) continue;   // (its quantisation tables go from the scan analysis into the descriptor)
      ws.GetThreadPool().AddWork([&, i](int) {
        const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
        int16_t *ptrs[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int c = 0; c < infos_[i].num_components; c++) ptrs[c] = coef_host + coef_off_[i * 3 + c];
        if (daliamdJpegDecodeCoefficients(data, in.nbytes(i), &infos_[i], ptrs, &quant_[(size_t)i * 192]) != 0)
          DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
        // Round 6: the encoded-stream cache keeps such a stream too - as the baseline re-encoding of the coefficients just
        // decoded (lossless: daliamdJpegEncodeBaselineScan), which the device decodes from the next epoch on.  A progressive
        // ImageNet-sized file costs 4-5 ms of host decode; 5 % of them in every batch held the resident pipeline at 24 000
        // images/s (bench.py value_mixed).  A stream whose coefficients do not fit the baseline code stays a host decode.
        if (stream_cache_ && structural_[i] && i < (int)in.source_info.size() && !in.source_info[i].empty() &&
            (infos_[i].num_components == 1 || infos_[i].num_components == 3) && transcode_) {
          auto &t = transcoded_[slot][i];
          t.resize(2 * in.nbytes(i) + 4096);
          size_t len = 0;
          const int16_t *cptrs[4] = {ptrs[0], ptrs[1], ptrs[2], ptrs[3]};
          if (daliamdJpegEncodeBaselineScan(&infos_[i], cptrs, &quant_[(size_t)i * 192], t.data(), t.size(), &len, &tscans_[i]) == 0)
            t.resize(len);
          else
            t.clear();
        }
      }, (int64_t)in.nbytes(i));
    }
    // ---- the other container formats: host decode into page-locked memory laid out like the output, one upload each
    if (nraster) {
      Buffer &rs = *raster_stage_[slot];
      std::vector<size_t> roff(n, 0);
      size_t rbytes = 0;
      for (int i = 0; i < n; i++) {
        if (raster_[i] != 1 && raster_[i] != 2) continue;   // (3: resident, nothing to decode)
        roff[i] = rbytes;
        // a sample that becomes resident is decoded WHOLE (its slot holds the upright image); any other only its window
        const size_t rows = rkeep_[i] ? (size_t)upright_hw_[2 * i] : (size_t)out.shape(i)[0];
        rbytes += (rows * (size_t)out.row_pitch(i) + 255) & ~(size_t)255;
      }
      rs.Reserve(rbytes + 256);
      for (int i = 0; i < n; i++) {
        if (raster_[i] != 1 && raster_[i] != 2) continue;
        ws.GetThreadPool().AddWork([&, i](int) {
          const bool window = rois_[4 * i + 2] > 0 && !rkeep_[i];
          const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
          uint8_t *dst = static_cast<uint8_t *>(rs.data()) + roff[i];
          const int64_t pitch = out.row_pitch(i);
          const int wy = window ? rois_[4 * i] : 0, wx = window ? rois_[4 * i + 1] : 0;
          const int wh = window ? (int)out.shape(i)[0] : upright_hw_[2 * i], ww = window ? (int)out.shape(i)[1] : upright_hw_[2 * i + 1];
          int rc;
          if (raster_[i] == 2) {
            // CMYK / YCCK JPEG: the whole upright image on the host, then the window
            const int H = raster_hw_[2 * i], W = raster_hw_[2 * i + 1];
            const int orient = adjust_orientation_ ? jpeg4_[i].orientation : 1;
            if (!window) {
              rc = daliamdJpegDecodeHost(data, in.nbytes(i), &jpeg4_[i], orient, out_type_, dst, pitch);
            } else {
              std::vector<uint8_t> full((size_t)H * W * oc_);
              rc = daliamdJpegDecodeHost(data, in.nbytes(i), &jpeg4_[i], orient, out_type_, full.data(), (int64_t)W * oc_);
              for (int y = 0; rc == 0 && y < wh; y++)
                memcpy(dst + (size_t)y * pitch, full.data() + ((size_t)(wy + y) * W + wx) * oc_, (size_t)ww * oc_);
            }
          } else if (out_type_ == DALI_RGB) {
            rc = daliamdImageDecodeRgb(data, in.nbytes(i), dst, pitch, wy, wx, window ? wh : 0, window ? ww : 0);
          } else {  // RGB rows first, then the requested format
            std::vector<uint8_t> rgb((size_t)wh * ww * 3);
            rc = daliamdImageDecodeRgb(data, in.nbytes(i), rgb.data(), (int64_t)ww * 3, wy, wx, window ? wh : 0, window ? ww : 0);
            if (rc == 0) rc = daliamdConvertRgbRows(rgb.data(), (int64_t)ww * 3, ww, wh, out_type_, dst, pitch);
          }
          if (rc != 0) DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
        }, (int64_t)in.nbytes(i));
      }
      ws.GetThreadPool().RunAll();
      for (int i = 0; i < n; i++) {
        if (raster_[i] != 1 && raster_[i] != 2) continue;
        if (rkeep_[i]) {   // the whole image into its cache slot (this iteration's output sample is a view of it), resident from here on
          KCHECK(daliamdMemcpyH2DAsync(rkeep_[i], static_cast<uint8_t *>(rs.data()) + roff[i],
                                       (size_t)upright_hw_[2 * i] * (size_t)out.row_pitch(i), ws.stream));
          stream_cache_->CommitRaster(in.source_info[i], upright_hw_[2 * i], upright_hw_[2 * i + 1], oc_, out.row_pitch(i), out_type_,
                                      ws.stream);
          auto &pending = reserved_rasters.keys;
          pending.erase(std::remove(pending.begin(), pending.end(), in.source_info[i]), pending.end());
        } else {
          KCHECK(daliamdMemcpyH2DAsync(out.raw(i), static_cast<uint8_t *>(rs.data()) + roff[i],
                                       (size_t)out.shape(i)[0] * (size_t)out.row_pitch(i), ws.stream));
        }
      }
    } else if (ngpu < nact) {
      ws.GetThreadPool().RunAll();
    }
    lap(2);
    if (nact == 0) { roi_draw.armed = false; return; }  // empty batch, or every sample came from the cache / was not a JPEG
    // ---- entropy decoding on the device ----
    int16_t *coef = static_cast<int16_t *>(cdev.data());
    // descriptor tables: built in the pinned staging buffer, addressed on the device at the same offsets
    uint8_t *stage_base = static_cast<uint8_t *>(ecs_stage.data());
    const uint8_t *dev_base = static_cast<const uint8_t *>(ecs_dev.data());
    const size_t huff_off = align16(ecs_bytes), idct_off = huff_off + align16((size_t)ngpu * sizeof(daliamdJpegHuffDesc));
    const size_t color_off = idct_off + align16((size_t)ncomp_total * sizeof(daliamdJpegIdctDesc));
    const size_t gather_off = color_off + align16((size_t)nact * sizeof(daliamdJpegColorDesc));
    const size_t upload_bytes = gather_off + (gather ? align16((size_t)ngpu * sizeof(daliamdGatherDesc)) : 0);
    // host copy of the tables: behind the packed segments, or (direct transfer, device-side fetch) at the start of the staging buffer
    uint8_t *tab_host = stage_base + (direct || gather ? 0 : huff_off);
    daliamdGatherDesc *fetch = reinterpret_cast<daliamdGatherDesc *>(tab_host + (gather_off - huff_off));
    size_t fetch_max = 0;
    daliamdJpegHuffDesc *huff = reinterpret_cast<daliamdJpegHuffDesc *>(tab_host);
    daliamdJpegIdctDesc *idct = reinterpret_cast<daliamdJpegIdctDesc *>(tab_host + (idct_off - huff_off));
    daliamdJpegColorDesc *color = reinterpret_cast<daliamdJpegColorDesc *>(tab_host + (color_off - huff_off));
    // (the Huffman descriptors are cleared one by one below - around the 1 088 bytes of DHT contents a stream with finished
    // code tables neither fills nor reads -, the other two tables here)
    memset(tab_host + (idct_off - huff_off), 0, upload_bytes - idct_off);
    int ntiles = 0, nsegs = 0, nbwg = 0, block_kernels = 0;
    fused_color_.assign(n, 0);
    // encoded-stream cache: the streams of this batch that the cache has room for become resident - as they are (one
    // device-to-device copy each, behind the transfer, out of this iteration's buffer, which is reused ring_ iterations from
    // now) or, cache_type="indexed", as the index entry the entropy decoder leaves behind this decode
    struct Kept { int j; uint8_t *slot; bool indexed, boxed; };
    std::vector<Kept> keep;
    struct ReservedStreams {   // reservations of this run; handed back unless the copies get enqueued and committed
      StreamCache *cache;
      std::vector<std::string> keys;
      ~ReservedStreams() { if (cache && !keys.empty()) cache->Abandon(keys); }
    } reserved_streams{stream_cache_.get(), {}};
    if (stream_cache_ && ngpu > nehit)
      for (int j = 0; j < ngpu; j++) {
        const int i = gpu_samples_[j];
        if (erec_[i] || i >= (int)in.source_info.size()) continue;
        // (a sample that arrived as an indexed container stays what it is: resident WITH its entry, whatever cache_type says)
        // Round 6: a stream of nearly empty blocks - large flat areas, a frame saturated to black or white: fewer than 64
        // bits per block on average, i.e. dozens of block starts in every 256-byte slice - is kept with its index under
        // cache_type="encoded" as well.  Such content never re-synchronises inside a slice, the relaxation of the position
        // pass then walks a segment lane by lane (5.8 ms for five 12-megapixel frames of 0.5 bits per pixel where 0.3 ms is
        // the norm); from the index every later epoch decodes it at the price of any other stream.
        const int64_t nblocks = (int64_t)scans_[i].mcus_x * scans_[i].mcus_y * scans_[i].blocks_per_mcu;
        const bool flat = index_flat_streams_ && nblocks > 0 && scans_[i].ecs_length * 8 < 64 * nblocks;
        const bool indexed = boxed_[i] > 0 || ((index_streams_ || flat) && scans_[i].restart_interval == 0 &&
                                               nblocks + 128 < ((int64_t)1 << 26));
        size_t bytes = (size_t)scans_[i].ecs_length;
        if (indexed) KCHECK(daliamdJpegHuffmanIndexBytes((int)scans_[i].ecs_length, &bytes));
        if (uint8_t *slot_ptr = stream_cache_->Reserve(in.source_info[i], bytes)) {
          keep.push_back({j, slot_ptr, indexed, boxed_[i] > 0});
          reserved_streams.keys.push_back(in.source_info[i]);
        }
      }
    if (ngpu) {
      // status words: pinned host memory the kernels write directly (no copy back); cleared here by the CPU
      int32_t *status = static_cast<int32_t *>(status_host.data());
      memset(status, 0, sizeof(int32_t) * (size_t)ngpu);
      for (int j = 0; j < ngpu; j++) {
        const int i = gpu_samples_[j];
        const auto &inf = infos_[i];
        const auto &sc = scan(i);
        auto &d = huff[j];
        // finished code tables of the stream's table set, when the store keeps them (else: built inside the launch)
        // (a resident record made before its table set was built - the set's first sighting - asks the store again)
        const uint8_t *tables = erec_[i] && erec_[i]->tables ? erec_[i]->tables : HuffTableStore::Get(device_id_, sc);
        {
          constexpr size_t dht0 = offsetof(daliamdJpegHuffDesc, bits), dht1 = offsetof(daliamdJpegHuffDesc, rect);
          uint8_t *raw = reinterpret_cast<uint8_t *>(&d);
          memset(raw, 0, tables ? dht0 : dht1);
          memset(raw + dht1, 0, sizeof(d) - dht1);
        }
        d.tables = tables;
        d.ecs = erec_[i] ? erec_[i]->ecs : dev_base + ecs_off_[i] + (direct ? (size_t)sc.ecs_offset : 0);
        if (gather) {
          // the segment where the file's mapping holds it -> its place in this iteration's buffer, on the source's 16-byte grid
          const uint8_t *src = static_cast<const uint8_t *>(in.raw(i)) + sc.ecs_offset;
          const size_t shift = reinterpret_cast<uintptr_t>(src) & 15;
          if (!erec_[i]) d.ecs = dev_base + ecs_off_[i] + shift;
          const size_t fetch_bytes = boxed_[i] ? (size_t)boxed_[i] : (size_t)sc.ecs_length;
          fetch[j] = daliamdGatherDesc{src, const_cast<uint8_t *>(d.ecs), erec_[i] ? 0u : (uint64_t)fetch_bytes, 0};
          if (!erec_[i]) fetch_max = std::max(fetch_max, fetch_bytes);
        }
        d.index = erec_[i] ? erec_[i]->index : nullptr;   // a resident stream with its side information: decoded from that
        if (boxed_[i] && !erec_[i]) {   // an indexed container: its entry sits where the segment of an ordinary file would
          d.index = d.ecs;
          d.ecs = nullptr;
        }
        d.scratch = static_cast<uint8_t *>(scratch.data()) + scratch_off_[i];
        d.status = status + j;
        d.ecs_len = (int32_t)sc.ecs_length;
        d.blocks_per_mcu = sc.blocks_per_mcu;
        d.mcus_x = sc.mcus_x;
        d.total_blocks = sc.mcus_x * sc.mcus_y * sc.blocks_per_mcu;
        d.restart_interval = sc.restart_interval;
        for (int c = 0; c < inf.num_components; c++) {
          d.coef[c] = coef + coef_off_[i * 3 + c];
          // fused output: the decoder dequantises + inverse-transforms its blocks and writes the planes itself
          d.plane[c] = static_cast<uint8_t *>(planes.data()) + coef_off_[i * 3 + c];
          d.plane_pitch[c] = inf.blocks_x[c] * 8;
          memcpy(d.quant[c], sc.quant[c], 128);
          d.blocks_x[c] = inf.blocks_x[c];
          d.h_samp[c] = inf.h_samp[c];
          d.v_samp[c] = inf.v_samp[c];
          d.dc_sel[c] = sc.dc_sel[c];
          d.ac_sel[c] = sc.ac_sel[c];
        }
        memcpy(d.comp_of_block, sc.comp_of_block, 10);
        memcpy(d.h_of_block, sc.h_of_block, 10);
        memcpy(d.v_of_block, sc.v_of_block, 10);
        if (plans_[i].roi_w > 0) memcpy(d.rect, plans_[i].rect, sizeof(d.rect));
        // fused colour output: an upright RGB image of a YCbCr 4:2:0 stream leaves the entropy decoder as RGB - no
        // component planes, no colour launch for this sample
        fused_color_[i] = 0;
        if (fuse_color_ && (inf.color == DALIAMD_JPEG_YCC || inf.color == DALIAMD_JPEG_GRAY) && out_type_ == DALI_RGB &&
            plans_[i].roi_w == 0 && (inf.width > 4 || sc.blocks_per_mcu != 6) &&
            (adjust_orientation_ ? inf.orientation : 1) <= 1 && (out.row_pitch(i) & 7) == 0 &&
            (reinterpret_cast<uintptr_t>(out.raw(i)) & 7) == 0 && daliamdJpegHuffmanColorFusable(&d)) {
          d.rgb = static_cast<uint8_t *>(out.raw(i));
          d.rgb_pitch = (int32_t)out.row_pitch(i);
          d.width = inf.width;
          d.height = inf.height;
          fused_color_[i] = 1;
        }
        for (int t = 0; t < 2 && !tables; t++) {   // (nobody reads the DHT contents of a stream that brings its tables)
          memcpy(d.bits[t], sc.dc_bits[t], 16);
          memcpy(d.bits[2 + t], sc.ac_bits[t], 16);
          memcpy(d.vals[t], sc.dc_vals[t], 256);
          memcpy(d.vals[2 + t], sc.ac_vals[t], 256);
        }
      }
      for (auto &k : keep)
        if (k.indexed && !k.boxed) huff[k.j].index_out = k.slot;
      KCHECK(daliamdJpegHuffmanSetupColor(huff, ngpu, &ntiles, &nsegs, &nbwg, &block_kernels));
      // the status words are valid once the iteration has finished: checked when its outputs are handed over.  (The
      // samples' names are looked up only when a word is set: the input of this iteration - a ring slot that is not written
      // again before the iteration has been handed out - is kept instead of 256 copied strings per batch.)
      const int32_t *st = status;
      std::shared_ptr<ImageCache> cache = cache_;
      std::shared_ptr<StreamCache> scache = stream_cache_;
      // The names of the decoded samples, kept by THIS operator per ring slot (the check runs when the iteration
      // is handed out; the input list it used to read belongs to the host stage, which may be rewriting that slot for
      // iteration it + ring by then - wrong names in the message, the wrong cache entry dropped).  This slot's copy is next
      // written by this operator's run for iteration it + ring, i.e. after the outputs of `it` have been handed over.  The
      // strings keep their capacity: no allocation per batch.
      if (check_names_.size() != (size_t)ring_) {
        check_names_.resize(ring_);
        for (auto &p : check_names_) p = std::make_shared<std::vector<std::string>>();
      }
      std::shared_ptr<std::vector<std::string>> names_ptr = check_names_[slot];
      names_ptr->resize(ngpu);
      for (int j = 0; j < ngpu; j++) {
        const int i = gpu_samples_[j];
        if (i < (int)in.source_info.size() && !in.source_info[i].empty()) (*names_ptr)[j].assign(in.source_info[i]);
        else (*names_ptr)[j] = make_string("sample #", i);
      }
      const size_t nsamples = (size_t)ngpu;
      ws.AddCompletionCheck([st, names_ptr, nsamples, cache, scache] {
        bool any = false;
        for (size_t j = 0; j < nsamples; j++) any = any || st[j] != 0;
        if (!any) return;
        const std::vector<std::string> names(names_ptr->begin(), names_ptr->begin() + nsamples);
        for (size_t j = 0; j < names.size(); j++)
          if (st[j] != 0 && cache) cache->Invalidate(names[j]);  // a slot may hold the broken image
        for (size_t j = 0; j < names.size(); j++)
          if (st[j] != 0 && scache) scache->Invalidate(names[j]);
        for (size_t j = 0; j < names.size(); j++)
          if (st[j] != 0) HeaderCache::Invalidate(names[j]);
        for (size_t j = 0; j < names.size(); j++)
          if (st[j] != 0)
            DALI_FAIL("Failed to decode ", names[j], ": corrupt JPEG data: ",
                      st[j] == 3 ? "restart markers in a stream without a restart interval"
                      : st[j] == 4 ? "a restart interval does not end where its marker is"
                                   : "the entropy-coded segment ends before the last MCU",
                      " (GPU Huffman status ", st[j], ")");
      });
    }
    // ---- dequantisation + IDCT, upsampling + colour conversion: descriptors ----
    int k = 0, kc = 0;
    for (int i = 0; i < n; i++) {
      if (hit_[i] || fused_color_[i]) continue;
      const auto &inf = infos_[i];
      auto &cd = color[kc++];
      for (int c = 0; c < inf.num_components; c++) {
        cd.plane[c] = static_cast<uint8_t *>(planes.data()) + coef_off_[i * 3 + c];
        cd.pitch[c] = inf.blocks_x[c] * 8;
        if (!scan(i).eligible) {  // host-decoded coefficients: the stand-alone IDCT kernel
          auto &d = idct[k++];
          d.coef = coef + coef_off_[i * 3 + c];
          d.plane = static_cast<uint8_t *>(planes.data()) + coef_off_[i * 3 + c];
          d.blocks_x = inf.blocks_x[c];
          d.nblocks = inf.blocks_x[c] * inf.blocks_y[c];
          d.pitch = cd.pitch[c];
          if (plans_[i].roi_w > 0) {
            const int32_t *r = plans_[i].rect[c];
            d.rect_x0 = r[0]; d.rect_y0 = r[1]; d.rect_w = r[2] - r[0];
            d.nblocks = (r[2] - r[0]) * (r[3] - r[1]);
          }
          memcpy(d.quant, &quant_[(size_t)i * 192 + c * 64], 128);
        }
        cd.h_samp[c] = inf.h_samp[c]; cd.v_samp[c] = inf.v_samp[c];
        cd.down_w[c] = inf.down_w[c]; cd.down_h[c] = inf.down_h[c];
      }
      for (int c = inf.num_components; c < 3; c++) { cd.h_samp[c] = cd.v_samp[c] = 1; }
      cd.width = inf.width; cd.height = inf.height; cd.color = inf.color;
      cd.out = static_cast<uint8_t *>(out.raw(i));
      cd.out_pitch = (int32_t)out.row_pitch(i);
      cd.orientation = adjust_orientation_ ? inf.orientation : 1;
      cd.out_format = out_type_;  // DALIImageType values = daliamdJpegOutFormat_t
      if (plans_[i].roi_w > 0) {
        cd.roi_x0 = plans_[i].roi_x0; cd.roi_y0 = plans_[i].roi_y0; cd.roi_w = plans_[i].roi_w; cd.roi_h = plans_[i].roi_h;
        cd.out_x0 = plans_[i].out_x0; cd.out_y0 = plans_[i].out_y0;
      }
    }
    int wg_idct = 0, wg_color = 0, color_kernels = 0;
    const int nidct = k;  // components of the host-decoded streams only
    KCHECK(daliamdJpegIdctSetup(idct, nidct, &wg_idct));
    const int ncolor = kc;   // samples that go through the colour launch (the others left the entropy decoder as RGB)
    KCHECK(daliamdJpegColorSetup(color, ncolor, &wg_color, &color_kernels));
    lap(3);
    // ---- ONE transfer (JPEG bytes + the three tables) on the copy stream: it overlaps the kernels of the previous
    // iteration; the compute stream waits for it through an event ----
    daliamdStream_t cs = ws.copy_stream ? ws.copy_stream : ws.stream;
    if (direct) {
      KCHECK(daliamdMemcpyH2DAsync(ecs_dev.data(), in.base(), in.total_bytes(), cs));
      KCHECK(daliamdMemcpyH2DAsync(static_cast<uint8_t *>(ecs_dev.data()) + huff_off, tab_host, upload_bytes - huff_off, cs));
    } else if (gather) {
      KCHECK(daliamdMemcpyH2DAsync(static_cast<uint8_t *>(ecs_dev.data()) + huff_off, tab_host, upload_bytes - huff_off, cs));
      KCHECK(daliamdGatherCopy(reinterpret_cast<const daliamdGatherDesc *>(dev_base + gather_off), ngpu, fetch_max, cs));
      NoteLaunch(ws, "gather_encoded");
    } else {
      KCHECK(daliamdMemcpyH2DAsync(ecs_dev.data(), ecs_stage.data(), upload_bytes, cs));
    }
    // The front of the entropy decoder (un-stuffing, code tables of streams that do not bring them) needs the transfer and
    // nothing else: on the side stream it runs while ws.stream is still with the previous iteration's kernels.
    daliamdStream_t side = ws.aux_stream && cs != ws.stream && ngpu && !keep_front_on_compute_ ? ws.aux_stream : nullptr;
    if (cs != ws.stream) {
      if (!h2d_done_[slot]) KCHECK(daliamdEventCreate(&h2d_done_[slot], 0));
      KCHECK(daliamdEventRecord(h2d_done_[slot], cs));
      if (side) {
        KCHECK(daliamdStreamWaitEvent(side, h2d_done_[slot]));
        KCHECK(daliamdJpegHuffmanRunFront(side, reinterpret_cast<const daliamdJpegHuffDesc *>(dev_base + huff_off), ngpu, ntiles,
                                          nsegs, nbwg, block_kernels));
        if (!front_done_[slot]) KCHECK(daliamdEventCreate(&front_done_[slot], 0));
        KCHECK(daliamdEventRecord(front_done_[slot], side));
        KCHECK(daliamdStreamWaitEvent(ws.stream, front_done_[slot]));
      } else {
        KCHECK(daliamdStreamWaitEvent(ws.stream, h2d_done_[slot]));
      }
    }
    // resident from here on (visible to later iterations, the reader's skip_cached_images and other pipelines): the streams
    // kept as they are, once their copies are on the stream; the ones kept with their index, behind the decode that builds it
    auto commit = [&](bool indexed) {
      std::vector<std::string> keys;
      std::vector<const daliamdJpegInfo *> kinfos;
      std::vector<const daliamdJpegScan *> kscans;
      for (auto &k : keep) {
        if (k.indexed != indexed) continue;
        const int i = gpu_samples_[k.j];
        if (!indexed) KCHECK(daliamdMemcpyD2DAsync(k.slot, huff[k.j].ecs, (size_t)scans_[i].ecs_length, ws.stream));
        else if (k.boxed) KCHECK(daliamdMemcpyD2DAsync(k.slot, huff[k.j].index, (size_t)boxed_[i], ws.stream));   // the entry it came with
        keys.push_back(in.source_info[i]);
        kinfos.push_back(&infos_[i]);
        kscans.push_back(&scans_[i]);
      }
      if (keys.empty()) return;
      stream_cache_->Commit(keys, kinfos, kscans, ws.stream, std::vector<uint8_t>(keys.size(), indexed ? 1 : 0), device_id_);
      auto &pending = reserved_streams.keys;
      for (auto &key : keys) pending.erase(std::remove(pending.begin(), pending.end(), key), pending.end());
    };
    commit(false);
    if (stream_cache_ && ngpu < nact) {   // the re-encoded streams of this batch's host decodes become resident
      std::vector<std::string> keys;
      std::vector<const daliamdJpegInfo *> kinfos;
      std::vector<const daliamdJpegScan *> kscans;
      for (int i = 0; i < n; i++) {
        if (hit_[i] || erec_[i] || scan(i).eligible || !structural_[i] || transcoded_[slot][i].empty()) continue;
        auto &t = transcoded_[slot][i];
        if (uint8_t *slot_ptr = stream_cache_->Reserve(in.source_info[i], t.size())) {
          reserved_streams.keys.push_back(in.source_info[i]);
          KCHECK(daliamdMemcpyH2DAsync(slot_ptr, t.data(), t.size(), ws.stream));
          keys.push_back(in.source_info[i]);
          kinfos.push_back(&infos_[i]);
          kscans.push_back(&tscans_[i]);
        }
      }
      if (!keys.empty()) {
        stream_cache_->Commit(keys, kinfos, kscans, ws.stream, std::vector<uint8_t>(keys.size(), 0), device_id_);
        auto &pending = reserved_streams.keys;
        for (auto &key : keys) pending.erase(std::remove(pending.begin(), pending.end(), key), pending.end());
      }
    }
    if (ngpu) {
      if (side)
        KCHECK(daliamdJpegHuffmanRunBack(ws.stream, reinterpret_cast<const daliamdJpegHuffDesc *>(dev_base + huff_off), ngpu,
                                         ntiles, nsegs, nbwg, block_kernels));
      else
        KCHECK(daliamdJpegHuffmanRunColor(ws.stream, reinterpret_cast<const daliamdJpegHuffDesc *>(dev_base + huff_off), ngpu,
                                          ntiles, nsegs, nbwg, block_kernels));
      NoteLaunch(ws, "jpeg_huffman");
      if (block_kernels & DALIAMD_JPEG_HUFFMAN_INDEXED) NoteLaunch(ws, "jpeg_huffman_indexed");   // (streams decoded from their index)
    }
    commit(true);
    // host-decoded streams (progressive, restart markers, multi-scan, below the threshold): H2D of their coefficients
    for (int i = 0; i < n; i++) {
      if (scan(i).eligible || hit_[i]) continue;
      int64_t first = coef_off_[i * 3], count = 0;
      for (int c = 0; c < infos_[i].num_components; c++) count += infos_[i].coef_elems[c];
      KCHECK(daliamdMemcpyH2DAsync(coef + first, coef_host + first, (size_t)count * 2, ws.stream));
    }
    if (nidct)
      KCHECK(daliamdJpegIdctRun(ws.stream, reinterpret_cast<const daliamdJpegIdctDesc *>(dev_base + idct_off), nidct,
                                wg_idct));
    KCHECK(daliamdJpegColorRun(ws.stream, reinterpret_cast<const daliamdJpegColorDesc *>(dev_base + color_off), ncolor,
                               wg_color, color_kernels));
    if (nidct) NoteLaunch(ws, "jpeg_idct");
    if (block_kernels & 2) NoteLaunch(ws, "jpeg_huffman_rgb");
    if (ncolor) NoteLaunch(ws, "jpeg_color");
    if (roi_source_) NoteLaunch(ws, "windows_of_the_consumer");   // (not a kernel: the graph-level fusion was in effect)
    if (cache_) {
      cache_->Commit(reserved.keys, ws.stream);  // visible to later iterations (and other pipelines) from here on
      reserved.keys.clear();
    }
    roi_draw.armed = false;
    lap(4);
  }

 protected:
  // Fills rois_[4*i .. 4*i+3] = {y0, x0, h, w} (h == 0: whole image) from upright_hw_ = {H, W} per sample.
  virtual void ComputeRois(const Workspace &ws, int n) {
    if (roi_source_) roi_source_((int64_t)ws.iteration, n, upright_hw_.data(), rois_.data());
  }
  std::vector<int32_t> upright_hw_, rois_;

 public:
  // Graph-level fusion (pipeline.cpp, TryEnableRoiDecodeFusion): the ONLY consumer of this decoders.image is a
  // RandomResizedCrop.  That operator draws the windows of the batch here - from ITS generator, in iteration order, so they
  // are the windows it would have drawn in its own Setup - and only they are decoded: the entropy decoder stops at the
  // window's last MCU row, the block and colour kernels serve its MCU rectangle.  The pixels the consumer resamples are
  // the ones a full decode holds at those positions (tests/test_gpu_jpeg.py: a window decode equals decode-then-crop),
  // so the pipeline's output does not change by a bit; what disappears is the 58 % of every image the crop throws away.
  using RoiSource = std::function<void(int64_t iteration, int n, const int32_t *upright_hw, int32_t *rois)>;
  // (the decoded-image cache keeps whole images; the opt-in colour output of the block kernel works on whole frames)
  bool CanTakeRoiSource() const { return cache_ == nullptr && !fuse_color_; }
  // undo(iteration): the windows drawn for `iteration` will not be decoded (the decoder failed behind the draw)
  void SetRoiSource(RoiSource source, std::function<void(int64_t)> undo) { roi_source_ = std::move(source); roi_undo_ = std::move(undo); }

 protected:
  RoiSource roi_source_;
  std::function<void(int64_t)> roi_undo_;

 private:
  std::vector<daliamdJpegRoiPlan> plans_;
  bool trace_ = getenv("DALI_AMD_TRACE") && atoi(getenv("DALI_AMD_TRACE")) != 0;
  double trace_s_[7] = {0, 0, 0, 0, 0, 0, 0};
  int64_t trace_runs_ = 0, trace_seen_ = 0;
  std::shared_ptr<ImageCache> cache_;
  std::shared_ptr<StreamCache> stream_cache_;
  bool index_streams_ = false;                                     // cache_type="indexed"
  bool index_flat_streams_ = !(getenv("DALI_AMD_INDEX_FLAT_STREAMS") && atoi(getenv("DALI_AMD_INDEX_FLAT_STREAMS")) == 0);
  int device_id_ = 0;
  std::vector<std::shared_ptr<const StreamCache::Record>> erec_;   // resident samples of the batch
  std::vector<uint8_t> hit_, raster_;
  std::vector<int32_t> raster_hw_;
  std::vector<ImageCache::Entry> cached_;
  std::vector<daliamdEvent_t> h2d_done_, front_done_;
  bool keep_front_on_compute_ = true;    // (DALI_AMD_DECODER_FRONT_ON_SIDE_STREAM=1 moves it)
  bool adjust_orientation_;
  bool host_huffman_only_ = false;
  int64_t huffman_threshold_ = 0;
  int ring_;
  std::vector<std::unique_ptr<Buffer>> staging_, coef_dev_, planes_, ecs_stage_, ecs_dev_, scratch_, status_host_, raster_stage_;
  int out_type_ = 0, oc_ = 3;            // output_type (DALIImageType) and its channel count
  std::vector<daliamdJpegInfo> jpeg4_;   // four-component streams of the batch (decoded on the host)
  std::vector<daliamdJpegInfo> infos_;
  std::vector<daliamdJpegScan> scans_;
  // host-decoded streams kept resident as their baseline re-encoding (round 6): per ring slot and sample the bytes (they are
  // the source of an asynchronous upload), per sample the scan analysis that goes with them
  std::vector<std::vector<std::vector<uint8_t>>> transcoded_;
  std::vector<daliamdJpegScan> tscans_;
  std::vector<uint8_t> structural_;
  std::vector<std::shared_ptr<std::vector<std::string>>> check_names_;   // per ring slot: names of the device-decoded samples
  std::vector<int64_t> boxed_;           // per sample: bytes of the index entry it brings (an indexed JPEG container), else 0
  std::vector<std::shared_ptr<const StreamCache::Record>> rres_;   // raster residents of the batch (decoded images kept in the encoded cache)
  std::vector<uint8_t *> rkeep_;                                   // cache slots of the rasters that become resident in this iteration
  bool raster_residents_ = !(getenv("DALI_AMD_RASTER_RESIDENTS") && atoi(getenv("DALI_AMD_RASTER_RESIDENTS")) == 0);
  bool transcode_ = !(getenv("DALI_AMD_TRANSCODE_PROGRESSIVE") && atoi(getenv("DALI_AMD_TRANSCODE_PROGRESSIVE")) == 0);
  std::vector<int> gpu_samples_;
  std::vector<char> fused_color_;        // per sample: the entropy decoder writes its RGB image (no colour launch)
  bool fuse_color_ = false;
  std::vector<int64_t> coef_off_;
  std::vector<size_t> ecs_off_, scratch_off_;
  std::vector<uint16_t> quant_;
};
// ---- decoders.image_random_crop: RandomCropAttr window, only the window is decoded --------------------------
// (dali/operators/imgcodec/decoder_schema.cc:270-299, roi_image_decoder.h:79-91, operators/image/crop/random_crop_attr.h)
DALI_SCHEMA(decoders__ImageRandomCrop)
    .DocStr("Decodes images and randomly crops them.\n\nThe cropping window's area (relative to the entire image) and "
            "aspect ratio can be restricted to a range of values specified by ``random_area`` and "
            "``random_aspect_ratio``.  Only the blocks of the JPEG that the window touches are dequantised, "
            "transformed and colour-converted (region-of-interest decoding); the entropy-coded stream is parsed up to "
            "the last MCU row of the window.\n\nThe output is in HWC layout.")
    .NumInput(1)
    .NumOutput(1)
    .AddParent("decoders__Image")
    .AddParent("RandomCropAttr");
DALI_SCHEMA(ImageDecoderRandomCrop).DocStr("Legacy alias of decoders.image_random_crop").NumInput(1).NumOutput(1)
    .AddParent("decoders__ImageRandomCrop");
DALI_SCHEMA(experimental__decoders__ImageRandomCrop).DocStr("Alias of decoders.image_random_crop").NumInput(1).NumOutput(1)
    .AddParent("decoders__ImageRandomCrop");

class ImageDecoderRandomCropMixed : public ImageDecoderMixed {
 public:
  explicit ImageDecoderRandomCropMixed(const OpSpec &spec) : ImageDecoderMixed(spec, false) {
    auto ar = spec.GetFloatVec("random_aspect_ratio"), area = spec.GetFloatVec("random_area");
    if (ar.size() == 1) ar.push_back(ar[0]);
    if (area.size() == 1) area.push_back(area[0]);
    DALI_ENFORCE(ar.size() == 2 && ar[0] <= ar[1], "Provided empty range");
    DALI_ENFORCE(area.size() == 2 && area[0] <= area[1], "Provided empty range");
    ar_lo_ = (float)ar[0]; ar_hi_ = (float)ar[1]; area_lo_ = (float)area[0]; area_hi_ = (float)area[1];
    num_attempts_ = (int)spec.GetInt("num_attempts");
    master_.key = (uint64_t)spec.GetInt("seed");
    master_.ctr[0] = master_.ctr[1] = 0;
    master_.phase = 0;
  }
  std::string SaveState() const override {
    char buf[96];
    daliamdPhiloxStateToString(&master_, buf, sizeof(buf));
    return buf;
  }
  void RestoreState(const std::string &s) override {
    DALI_ENFORCE(daliamdPhiloxStateFromString(&master_, s.c_str()) == 0, daliamdHostGetLastErrorMessage());
  }

 protected:
  void ComputeRois(const Workspace &, int n) override {
    anchors_.resize(2 * n); crops_.resize(2 * n);
    if (daliamdRandomCropBatch(&master_, n, upright_hw_.data(), ar_lo_, ar_hi_, area_lo_, area_hi_, num_attempts_,
                               anchors_.data(), crops_.data()) != 0)
      DALI_FAIL(daliamdHostGetLastErrorMessage());
    for (int i = 0; i < n; i++) {
      rois_[4 * i] = anchors_[2 * i]; rois_[4 * i + 1] = anchors_[2 * i + 1];
      rois_[4 * i + 2] = crops_[2 * i]; rois_[4 * i + 3] = crops_[2 * i + 1];
    }
    daliamdPhiloxAdvanceSequence(&master_, (uint64_t)n);  // OperatorWithRng::Advance(batch)
  }

 private:
  int num_attempts_;
  float ar_lo_, ar_hi_, area_lo_, area_hi_;
  daliamdPhiloxState master_;
  std::vector<int32_t> anchors_, crops_;
};

// ---- decoders.image_crop: CropAttr window (fixed size, normalised anchor) ---------------------------------------
// (decoder_schema.cc:168-196, roi_image_decoder.h:47-61, operators/image/crop/crop_attr.cc:90-240)
DALI_SCHEMA(decoders__ImageCrop)
    .DocStr("Decodes images and extracts regions-of-interest (ROI) that are specified by fixed window dimensions and "
            "variable anchors.  Only the window is decoded.\n\nThe output is in HWC layout.")
    .NumInput(1)
    .NumOutput(1)
    .AddParent("decoders__Image")
    .AddParent("CropAttr");
DALI_SCHEMA(ImageDecoderCrop).DocStr("Legacy alias of decoders.image_crop").NumInput(1).NumOutput(1)
    .AddParent("decoders__ImageCrop");
DALI_SCHEMA(experimental__decoders__ImageCrop).DocStr("Alias of decoders.image_crop").NumInput(1).NumOutput(1)
    .AddParent("decoders__ImageCrop");

class ImageDecoderCropMixed : public ImageDecoderMixed {
 public:
  explicit ImageDecoderCropMixed(const OpSpec &spec) : ImageDecoderMixed(spec, false) {
    std::string r = spec.GetString("rounding");
    DALI_ENFORCE(r == "round" || r == "truncate", "Unsupported rounding \"", r, "\"");
    round_ = r == "round";
    DALI_ENFORCE(spec.ArgumentDefined("crop") || (spec.ArgumentDefined("crop_w") && spec.ArgumentDefined("crop_h")),
                 "decoders.image_crop needs `crop` or both `crop_w` and `crop_h`");
  }

 protected:
  void ComputeRois(const Workspace &ws, int n) override {
    auto pos_x = GetPerSampleFloat(spec_, ws, "crop_pos_x", n), pos_y = GetPerSampleFloat(spec_, ws, "crop_pos_y", n);
    std::vector<float> crop_h, crop_w;
    if (spec_.ArgumentDefined("crop_h")) crop_h = GetPerSampleFloat(spec_, ws, "crop_h", n);
    if (spec_.ArgumentDefined("crop_w")) crop_w = GetPerSampleFloat(spec_, ws, "crop_w", n);
    for (int i = 0; i < n; i++) {
      int64_t H = upright_hw_[2 * i], W = upright_hw_[2 * i + 1], ch = H, cw = W;
      if (spec_.ArgumentDefined("crop")) {
        DALI_ENFORCE(!spec_.HasTensorArgument("crop"), "Per-sample `crop` tensors are not supported yet");
        auto c = spec_.GetFloatVec("crop");
        DALI_ENFORCE(c.size() == 2, "`crop` must hold (crop_H, crop_W)");
        ch = (int64_t)c[0]; cw = (int64_t)c[1];
      }
      if (!crop_h.empty()) ch = (int64_t)crop_h[i];
      if (!crop_w.empty()) cw = (int64_t)crop_w[i];
      DALI_ENFORCE(ch > 0 && cw > 0, "Crop window must have a positive size");
      DALI_ENFORCE(pos_x[i] >= 0.0f && pos_x[i] <= 1.0f && pos_y[i] >= 0.0f && pos_y[i] <= 1.0f,
                   "Anchor for dimension ", 0, " is out of range [0.0, 1.0]");
      int64_t ay = daliamdCropAnchor(pos_y[i], ch, H, round_), ax = daliamdCropAnchor(pos_x[i], cw, W, round_);
      // CropWindow::EnforceInRange (roi_image_decoder.h:36-45)
      DALI_ENFORCE(ay >= 0 && ax >= 0 && ay + ch <= H && ax + cw <= W, "Cropping window [", ay, ":", ay + ch, ", ", ax,
                   ":", ax + cw, "] is out of the bounds of the ", H, "x", W, " image");
      rois_[4 * i] = (int32_t)ay; rois_[4 * i + 1] = (int32_t)ax; rois_[4 * i + 2] = (int32_t)ch; rois_[4 * i + 3] = (int32_t)cw;
    }
  }

 private:
  bool round_;
};

// ---- decoders.image, device="cpu": the whole decode on the host thread pool ------------------------------------
// ImageDecoder<CPUBackend> (dali/operators/imgcodec/image_decoder.h:613-880, registered host_decoder.cc:35-48): header
// parse -> output shapes (EXIF orientation swaps height and width, :677-681) -> one task per sample
// (:724-749) decoding with libjpeg-turbo semantics (accurate IDCT, fancy upsampling: :289-305).  Product code:
// daliamdJpegDecodeRgbHost (host/jpeg_pixels.cpp) - the same bytes as the device path.
class ImageDecoderCpu : public OperatorBase {
 public:
  explicit ImageDecoderCpu(const OpSpec &spec) : OperatorBase(spec) {
    const int64_t ot = spec.GetInt("output_type");
    DALI_ENFORCE(ot == DALI_RGB || ot == DALI_BGR || ot == DALI_GRAY || ot == DALI_YCbCr || ot == DALI_ANY_DATA,
                 "decoders.image: unsupported output_type ", ot);
    out_type_ = (int)ot;
    DALI_ENFORCE(spec.GetInt("dtype") == DALI_UINT8, "decoders.image: only dtype=UINT8 is supported");
    adjust_orientation_ = spec.GetBool("adjust_orientation");
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    const int n = in.num_samples();
    DALI_ENFORCE(in.type() == DALI_UINT8, "decoders.image expects encoded streams as 1-D uint8 tensors");
    infos_.resize(n);
    raster_.assign(n, 0);
    orient_.assign(n, 1);
    desc[0].type = DALI_UINT8;
    desc[0].shape.resize(n);
    auto src = [&](int i) { return i < (int)in.source_info.size() && !in.source_info[i].empty() ? in.source_info[i]
                                                                                                  : make_string("sample ", i); };
    for (int i = 0; i < n; i++) {
      const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
      const size_t size = (size_t)in.nbytes(i);
      daliamdImageFormat fmt = DALIAMD_IMAGE_UNKNOWN;
      int32_t w = 0, h = 0;
      if (daliamdImageProbe(data, size, &fmt, &w, &h) != 0)
        DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
      if (fmt == DALIAMD_IMAGE_JPEG) {
        if (daliamdJpegParse(data, size, &infos_[i]) != 0)
          DALI_FAIL("Failed to parse ", src(i), ": ", daliamdHostGetLastErrorMessage());
        orient_[i] = adjust_orientation_ ? infos_[i].orientation : 1;
        w = infos_[i].width; h = infos_[i].height;
        if (orient_[i] >= 5 && orient_[i] <= 8) std::swap(w, h);
        desc[0].shape[i] = TensorShape{h, w, daliamdJpegOutputChannels(infos_[i].num_components, out_type_)};
      } else {
        raster_[i] = 1;
        desc[0].shape[i] = TensorShape{h, w, out_type_ == DALI_GRAY ? 1 : 3};
      }
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    out.SetLayout("HWC");
    out.source_info = in.source_info;
    const int n = in.num_samples();
    for (int i = 0; i < n; i++) {
      ws.GetThreadPool().AddWork([this, &in, &out, i](int) {
        const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
        const size_t size = (size_t)in.nbytes(i);
        uint8_t *dst = static_cast<uint8_t *>(out.raw(i));
        const int oc = (int)out.shape(i)[2];
        const int64_t pitch = out.row_pitch(i) ? out.row_pitch(i) : out.shape(i)[1] * oc;
        int rc;
        if (!raster_[i]) {
          rc = daliamdJpegDecodeHost(data, size, &infos_[i], orient_[i], out_type_, dst, pitch);
        } else if (out_type_ == DALI_RGB || out_type_ == DALI_ANY_DATA) {
          rc = daliamdImageDecodeRgb(data, size, dst, pitch, 0, 0, 0, 0);
        } else {
          const int h = (int)out.shape(i)[0], w = (int)out.shape(i)[1];
          std::vector<uint8_t> rgb((size_t)h * w * 3);
          rc = daliamdImageDecodeRgb(data, size, rgb.data(), (int64_t)w * 3, 0, 0, 0, 0);
          if (rc == 0) rc = daliamdConvertRgbRows(rgb.data(), (int64_t)w * 3, w, h, out_type_, dst, pitch);
        }
        if (rc != 0)
          DALI_FAIL("Failed to decode ", i < (int)in.source_info.size() ? in.source_info[i] : make_string("sample ", i), ": ",
                    daliamdHostGetLastErrorMessage());
      }, in.nbytes(i));
    }
    ws.GetThreadPool().RunAll();
    NoteLaunch(ws, "host_jpeg_decode");
  }

 private:
  bool adjust_orientation_;
  int out_type_ = 0;
  std::vector<daliamdJpegInfo> infos_;
  std::vector<int> raster_, orient_;
};
DALI_REGISTER_OPERATOR(decoders__Image, ImageDecoderCpu, CPU);
DALI_REGISTER_OPERATOR(ImageDecoder, ImageDecoderCpu, CPU);
DALI_REGISTER_OPERATOR(experimental__decoders__Image, ImageDecoderCpu, CPU);

// ---- decoders.image_slice: SliceAttr window (named start / end / shape arguments or anchor + shape inputs) --------
// (decoder_schema.cc:200-252, generic/slice/slice_attr.h:40-352, slice_attr.cc:21-110)
DALI_SCHEMA(SliceAttr)
    .DocStr("Slice attributes placeholder")
    .MakeInternal()
    .AddOptionalArg("axes", "Order of dimensions used for the anchor and shape slice inputs as dimension indices.",
                    ArgValue::IntVec({1, 0}))
    .AddOptionalArg("axis_names", "Order of the dimensions used for the anchor and shape slice inputs, as described in "
                    "layout.  If a value is provided, it has a higher priority than `axes`.", ArgValue::Str("WH"))
    .AddOptionalTypeArg("start", "Start coordinates of the slice.", ArgType::INT_VEC, true)
    .AddOptionalTypeArg("rel_start", "Start relative coordinates of the slice (range [0.0 - 1.0]).", ArgType::FLOAT_VEC, true)
    .AddOptionalTypeArg("end", "End coordinates of the slice.", ArgType::INT_VEC, true)
    .AddOptionalTypeArg("rel_end", "End relative coordinates of the slice (range [0.0 - 1.0]).", ArgType::FLOAT_VEC, true)
    .AddOptionalTypeArg("shape", "Shape of the slice.", ArgType::INT_VEC, true)
    .AddOptionalTypeArg("rel_shape", "Relative shape of the slice (range [0.0 - 1.0]).", ArgType::FLOAT_VEC, true)
    .AddOptionalArg("normalized_anchor", "Determines whether the anchor positional input should be interpreted as "
                    "normalized (range [0.0, 1.0]) or as absolute coordinates (float anchors only).", ArgValue::Bool(true))
    .AddOptionalArg("normalized_shape", "Determines whether the shape positional input should be interpreted as "
                    "normalized (range [0.0, 1.0]) or as absolute coordinates (float shapes only).", ArgValue::Bool(true));

DALI_SCHEMA(decoders__ImageSlice)
    .DocStr("Decodes images and extracts regions of interest.  The slice is given by start and end coordinates or start "
            "coordinates and shape (`start` / `rel_start`, `end` / `rel_end`, `shape` / `rel_shape`), or by two extra "
            "positional inputs `anchor` and `shape` (normalised by default, \"WH\" order).  Only the window is "
            "decoded.\n\nThe output is in HWC layout.")
    .NumInput(1, 3)
    .NumOutput(1)
    .AddParent("decoders__Image")
    .AddParent("SliceAttr");
DALI_SCHEMA(ImageDecoderSlice).DocStr("Legacy alias of decoders.image_slice").NumInput(1, 3).NumOutput(1)
    .AddParent("decoders__ImageSlice");
DALI_SCHEMA(experimental__decoders__ImageSlice).DocStr("Alias of decoders.image_slice").NumInput(1, 3).NumOutput(1)
    .AddParent("decoders__ImageSlice");

class ImageDecoderSliceMixed : public ImageDecoderMixed {
 public:
  explicit ImageDecoderSliceMixed(const OpSpec &spec) : ImageDecoderMixed(spec, false) {
    const bool start = spec.ArgumentDefined("start"), rel_start = spec.ArgumentDefined("rel_start");
    DALI_ENFORCE(!(start && rel_start), "\"start\" and \"rel_start\" arguments are mutually exclusive");
    const int ends = spec.ArgumentDefined("end") + spec.ArgumentDefined("rel_end") + spec.ArgumentDefined("shape") +
                     spec.ArgumentDefined("rel_shape");
    DALI_ENFORCE(ends <= 1, "\"end\", \"rel_end\", \"shape\", and \"rel_shape\" arguments are mutually exclusive");
    named_ = start || rel_start || ends > 0;
    // axes of the slice arguments in the HWC image: axis_names wins over axes (AxisArgs, operators/util/axis_args.h)
    std::string names = spec.GetString("axis_names");
    if (spec.ArgumentDefined("axis_names") || !spec.ArgumentDefined("axes")) {
      for (char c : names) {
        DALI_ENFORCE(c == 'H' || c == 'W', "decoders.image_slice: axis \"", std::string(1, c), "\" is not a spatial axis "
                     "of an HWC image");
        axes_.push_back(c == 'H' ? 0 : 1);
      }
    } else {
      for (int64_t a : spec.GetIntVec("axes")) {
        if (a < 0) a += 3;
        DALI_ENFORCE(a == 0 || a == 1, "decoders.image_slice: only the spatial axes (0, 1) of an HWC image can be sliced");
        axes_.push_back((int)a);
      }
    }
    DALI_ENFORCE(axes_.size() <= 2 && (axes_.size() < 2 || axes_[0] != axes_[1]), "Axis indices must be unique");
  }

 protected:
  void ComputeRois(const Workspace &ws, int n) override {
    const bool positional = ws.NumInput() == 3;
    DALI_ENFORCE(ws.NumInput() == 1 || positional, "decoders.image_slice takes the encoded images and, optionally, the "
                 "anchor AND the shape of the slice");
    DALI_ENFORCE(!(positional && named_), "Named slice arguments (start / end / shape ...) are incompatible with the "
                 "positional anchor and shape inputs");
    std::vector<std::vector<float>> a, b;   // per sample: start-like and end-like values per listed axis
    enum { kNone, kAbs, kRel } a_kind = kNone;
    enum { kEndNone, kEnd, kRelEnd, kShape, kRelShape } b_kind = kEndNone;
    bool norm_anchor = false, norm_shape = false;
    if (positional) {
      a = InputAsFloat(ws.Input(1), n, &norm_anchor);
      b = InputAsFloat(ws.Input(2), n, &norm_shape);
      DALI_ENFORCE(ws.Input(1).type() == ws.Input(2).type(), "Anchor and shape should have the same type. Got: ",
                   TypeName(ws.Input(1).type()), " and ", TypeName(ws.Input(2).type()));
      norm_anchor = norm_anchor && spec_.GetBool("normalized_anchor");
      norm_shape = norm_shape && spec_.GetBool("normalized_shape");
    } else {
      if (spec_.ArgumentDefined("start")) { a = GetPerSampleFloatVec(spec_, ws, "start", n); a_kind = kAbs; }
      else if (spec_.ArgumentDefined("rel_start")) { a = GetPerSampleFloatVec(spec_, ws, "rel_start", n); a_kind = kRel; }
      if (spec_.ArgumentDefined("end")) { b = GetPerSampleFloatVec(spec_, ws, "end", n); b_kind = kEnd; }
      else if (spec_.ArgumentDefined("rel_end")) { b = GetPerSampleFloatVec(spec_, ws, "rel_end", n); b_kind = kRelEnd; }
      else if (spec_.ArgumentDefined("shape")) { b = GetPerSampleFloatVec(spec_, ws, "shape", n); b_kind = kShape; }
      else if (spec_.ArgumentDefined("rel_shape")) { b = GetPerSampleFloatVec(spec_, ws, "rel_shape", n); b_kind = kRelShape; }
    }
    for (int i = 0; i < n; i++) {
      const int64_t extent[2] = {upright_hw_[2 * i], upright_hw_[2 * i + 1]};
      int64_t anchor[2] = {0, 0}, shape[2] = {extent[0], extent[1]};
      for (size_t k = 0; k < axes_.size(); k++) {
        const int dim = axes_[k];
        const double ext = (double)extent[dim];
        double anchor_val = 0, end_val = ext;
        if (positional) {
          DALI_ENFORCE(a[i].size() == axes_.size() && b[i].size() == axes_.size(), "Expected ", axes_.size(),
                       " elements for slice arguments (start/shape). Got ", a[i].size());
          anchor_val = a[i][k];
          double shape_val = b[i][k];
          if (norm_anchor && norm_shape) {  // one multiplication after the sum (slice_attr.h:312-315)
            end_val = (anchor_val + shape_val) * ext;
            anchor_val *= ext;
          } else {
            if (norm_anchor) anchor_val *= ext;
            if (norm_shape) shape_val *= ext;
            end_val = anchor_val + shape_val;
          }
        } else {
          if (a_kind != kNone && !a[i].empty()) {
            DALI_ENFORCE(a[i].size() == axes_.size(), "Expected ", axes_.size(), " elements for the slice start");
            anchor_val = a_kind == kAbs ? (double)a[i][k] : (double)a[i][k] * ext;
          }
          if (b_kind != kEndNone && !b[i].empty()) {
            DALI_ENFORCE(b[i].size() == axes_.size(), "Expected ", axes_.size(), " elements for the slice end / shape");
            const double v = b[i][k];
            if (b_kind == kEnd) end_val = v;
            else if (b_kind == kRelEnd) end_val = v * ext;
            else if (b_kind == kShape) { DALI_ENFORCE(v >= 0, "shape value out of range. Got: ", v); end_val = anchor_val + v; }
            else if (a_kind == kRel && !a[i].empty()) {  // rel_start + rel_shape: multiply once after the sum
              DALI_ENFORCE(v >= 0, "negative shapes are not allowed. Got: ", v);
              end_val = ((double)a[i][k] + v) * ext;
            } else { DALI_ENFORCE(v >= 0, "negative shapes are not allowed. Got: ", v); end_val = anchor_val + v * ext; }
          }
        }
        DALI_ENFORCE(end_val >= anchor_val, "end coordinates can't be before start coordinates. Got: start=", anchor_val,
                     " end=", end_val);
        anchor[dim] = std::llround(anchor_val);
        shape[dim] = std::llround(end_val) - anchor[dim];
      }
      // CropWindow::EnforceInRange (roi_image_decoder.h:36-45)
      DALI_ENFORCE(anchor[0] >= 0 && anchor[1] >= 0 && shape[0] > 0 && shape[1] > 0 && anchor[0] + shape[0] <= extent[0] &&
                       anchor[1] + shape[1] <= extent[1],
                   "Cropping window [", anchor[0], ":", anchor[0] + shape[0], ", ", anchor[1], ":", anchor[1] + shape[1],
                   "] is out of the bounds of the ", extent[0], "x", extent[1], " image");
      rois_[4 * i] = (int32_t)anchor[0]; rois_[4 * i + 1] = (int32_t)anchor[1];
      rois_[4 * i + 2] = (int32_t)shape[0]; rois_[4 * i + 3] = (int32_t)shape[1];
    }
  }

 private:
  // a CPU input of 1-D samples as floats; *is_float: the element type is a floating-point type
  static std::vector<std::vector<float>> InputAsFloat(const TensorList &t, int n, bool *is_float) {
    DALI_ENFORCE(t.device() == StorageDevice::CPU, "decoders.image_slice: the anchor and shape inputs must be CPU tensors");
    DALI_ENFORCE(t.num_samples() == n, "The anchor / shape inputs must have one sample per image");
    *is_float = t.type() == DALI_FLOAT || t.type() == DALI_FLOAT64;
    std::vector<std::vector<float>> out(n);
    for (int i = 0; i < n; i++) {
      const int64_t cnt = volume(t.shape(i));
      out[i].resize(cnt);
      for (int64_t k = 0; k < cnt; k++) {
        switch (t.type()) {
          case DALI_FLOAT: out[i][k] = static_cast<const float *>(t.raw(i))[k]; break;
          case DALI_FLOAT64: out[i][k] = (float)static_cast<const double *>(t.raw(i))[k]; break;
          case DALI_INT32: out[i][k] = (float)static_cast<const int32_t *>(t.raw(i))[k]; break;
          case DALI_INT64: out[i][k] = (float)static_cast<const int64_t *>(t.raw(i))[k]; break;
          default: DALI_FAIL("Unsupported type of anchor and shape arguments: ", TypeName(t.type()));
        }
      }
    }
    return out;
  }
  bool named_;
  std::vector<int> axes_;
};
// ---- the region-of-interest decoders with device="cpu" ----------------------------------------------------------------
// ImageDecoder<CPUBackend> with a CropWindowGenerator (roi_image_decoder.h:47-96, host_decoder.cc:35-48): the window
// arithmetic IS the mixed operator's (its ComputeRois, its random-crop generator state and checkpoint); the decode is the
// host decoder of decoders.image(device="cpu"), the window copied out of the upright image.  Nothing of the device path
// of the base class runs: its buffers are allocated on first use only.
template <class MixedRoiDecoder>
class RoiDecoderCpu : public MixedRoiDecoder {
 public:
  explicit RoiDecoderCpu(const OpSpec &spec) : MixedRoiDecoder(WithQueueDepth(spec)) {
    const int64_t ot = spec.GetInt("output_type");
    cpu_out_type_ = (int)ot;
    cpu_adjust_orientation_ = spec.GetBool("adjust_orientation");
  }
  int OutputPitchAlign(int) const override { return 1; }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }
  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    const int n = in.num_samples();
    DALI_ENFORCE(in.type() == DALI_UINT8, "decoders.image expects encoded streams as 1-D uint8 tensors");
    auto src = [&](int i) { return i < (int)in.source_info.size() && !in.source_info[i].empty() ? in.source_info[i]
                                                                                               : make_string("sample ", i); };
    cpu_infos_.resize(n);
    cpu_raster_.assign(n, 0);
    cpu_orient_.assign(n, 1);
    std::vector<int> channels(n, 3);
    this->upright_hw_.assign(2 * (size_t)n, 0);
    for (int i = 0; i < n; i++) {
      const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
      const size_t size = (size_t)in.nbytes(i);
      daliamdImageFormat fmt = DALIAMD_IMAGE_UNKNOWN;
      int32_t w = 0, h = 0;
      if (daliamdImageProbe(data, size, &fmt, &w, &h) != 0) DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
      if (fmt == DALIAMD_IMAGE_JPEG) {
        if (daliamdJpegParse(data, size, &cpu_infos_[i]) != 0) DALI_FAIL("Failed to parse ", src(i), ": ", daliamdHostGetLastErrorMessage());
        cpu_orient_[i] = cpu_adjust_orientation_ ? cpu_infos_[i].orientation : 1;
        w = cpu_infos_[i].width; h = cpu_infos_[i].height;
        if (cpu_orient_[i] >= 5 && cpu_orient_[i] <= 8) std::swap(w, h);
        channels[i] = daliamdJpegOutputChannels(cpu_infos_[i].num_components, cpu_out_type_);
      } else {
        cpu_raster_[i] = 1;
        channels[i] = cpu_out_type_ == DALI_GRAY ? 1 : 3;
      }
      this->upright_hw_[2 * i] = h;
      this->upright_hw_[2 * i + 1] = w;
    }
    this->rois_.assign(4 * (size_t)n, 0);
    this->ComputeRois(ws, n);
    std::vector<TensorShape> shapes(n);
    for (int i = 0; i < n; i++) {
      const bool window = this->rois_[4 * i + 2] > 0;
      shapes[i] = {window ? this->rois_[4 * i + 2] : this->upright_hw_[2 * i],
                   window ? this->rois_[4 * i + 3] : this->upright_hw_[2 * i + 1], channels[i]};
    }
    out.Resize(shapes, DALI_UINT8);
    out.SetLayout("HWC");
    out.source_info = in.source_info;
    for (int i = 0; i < n; i++) {
      ws.GetThreadPool().AddWork([this, &in, &out, &channels, i, src](int) {
        const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
        const size_t size = (size_t)in.nbytes(i);
        const int H = this->upright_hw_[2 * i], W = this->upright_hw_[2 * i + 1], oc = channels[i];
        std::vector<uint8_t> full((size_t)H * W * oc);
        const int64_t fpitch = (int64_t)W * oc;
        int rc;
        if (!cpu_raster_[i]) {
          rc = daliamdJpegDecodeHost(data, size, &cpu_infos_[i], cpu_orient_[i], cpu_out_type_, full.data(), fpitch);
        } else if (cpu_out_type_ == DALI_RGB || cpu_out_type_ == DALI_ANY_DATA) {
          rc = daliamdImageDecodeRgb(data, size, full.data(), fpitch, 0, 0, 0, 0);
        } else {
          std::vector<uint8_t> rgb((size_t)H * W * 3);
          rc = daliamdImageDecodeRgb(data, size, rgb.data(), (int64_t)W * 3, 0, 0, 0, 0);
          if (rc == 0) rc = daliamdConvertRgbRows(rgb.data(), (int64_t)W * 3, W, H, cpu_out_type_, full.data(), fpitch);
        }
        if (rc != 0) DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
        const bool window = this->rois_[4 * i + 2] > 0;
        const int y0 = window ? this->rois_[4 * i] : 0, x0 = window ? this->rois_[4 * i + 1] : 0;
        const int h = (int)out.shape(i)[0], w = (int)out.shape(i)[1];
        uint8_t *dst = static_cast<uint8_t *>(out.raw(i));
        const int64_t pitch = out.row_pitch(i) ? out.row_pitch(i) : (int64_t)w * oc;
        for (int y = 0; y < h; y++)
          memcpy(dst + y * pitch, full.data() + (size_t)(y0 + y) * fpitch + (size_t)x0 * oc, (size_t)w * oc);
      }, in.nbytes(i));
    }
    ws.GetThreadPool().RunAll();
    NoteLaunch(ws, "host_jpeg_decode_roi");
  }

 private:
  static OpSpec WithQueueDepth(OpSpec spec) {  // the executor hands this argument to device operators only
    if (!spec.Args().count("gpu_prefetch_queue_depth")) spec.AddArg("gpu_prefetch_queue_depth", ArgValue::Int(1));
    return spec;
  }
  int cpu_out_type_ = 0;
  bool cpu_adjust_orientation_ = true;
  std::vector<daliamdJpegInfo> cpu_infos_;
  std::vector<int> cpu_raster_, cpu_orient_;
};
using ImageDecoderRandomCropCpu = RoiDecoderCpu<ImageDecoderRandomCropMixed>;
using ImageDecoderCropCpu = RoiDecoderCpu<ImageDecoderCropMixed>;
using ImageDecoderSliceCpu = RoiDecoderCpu<ImageDecoderSliceMixed>;
DALI_REGISTER_OPERATOR(decoders__ImageRandomCrop, ImageDecoderRandomCropCpu, CPU);
DALI_REGISTER_OPERATOR(ImageDecoderRandomCrop, ImageDecoderRandomCropCpu, CPU);
DALI_REGISTER_OPERATOR(experimental__decoders__ImageRandomCrop, ImageDecoderRandomCropCpu, CPU);
DALI_REGISTER_OPERATOR(decoders__ImageCrop, ImageDecoderCropCpu, CPU);
DALI_REGISTER_OPERATOR(ImageDecoderCrop, ImageDecoderCropCpu, CPU);
DALI_REGISTER_OPERATOR(experimental__decoders__ImageCrop, ImageDecoderCropCpu, CPU);
DALI_REGISTER_OPERATOR(decoders__ImageSlice, ImageDecoderSliceCpu, CPU);
DALI_REGISTER_OPERATOR(ImageDecoderSlice, ImageDecoderSliceCpu, CPU);
DALI_REGISTER_OPERATOR(experimental__decoders__ImageSlice, ImageDecoderSliceCpu, CPU);

DALI_REGISTER_OPERATOR(decoders__ImageSlice, ImageDecoderSliceMixed, MIXED);
DALI_REGISTER_OPERATOR(ImageDecoderSlice, ImageDecoderSliceMixed, MIXED);
DALI_REGISTER_OPERATOR(experimental__decoders__ImageSlice, ImageDecoderSliceMixed, MIXED);

DALI_REGISTER_OPERATOR(decoders__ImageRandomCrop, ImageDecoderRandomCropMixed, MIXED);
DALI_REGISTER_OPERATOR(ImageDecoderRandomCrop, ImageDecoderRandomCropMixed, MIXED);
DALI_REGISTER_OPERATOR(experimental__decoders__ImageRandomCrop, ImageDecoderRandomCropMixed, MIXED);
DALI_REGISTER_OPERATOR(decoders__ImageCrop, ImageDecoderCropMixed, MIXED);
DALI_REGISTER_OPERATOR(ImageDecoderCrop, ImageDecoderCropMixed, MIXED);
DALI_REGISTER_OPERATOR(experimental__decoders__ImageCrop, ImageDecoderCropMixed, MIXED);
DALI_REGISTER_OPERATOR(decoders__Image, ImageDecoderMixed, MIXED);
DALI_REGISTER_OPERATOR(ImageDecoder, ImageDecoderMixed, MIXED);
DALI_REGISTER_OPERATOR(experimental__decoders__Image, ImageDecoderMixed, MIXED);

// =============================================================================================
// Resampling operators
// =============================================================================================
DALI_SCHEMA(ResamplingFilterAttr)
    .DocStr("Resampling filter attribute placeholder")
    .MakeInternal()
    .AddOptionalArg("interp_type", "Type of interpolation to be used.", ArgValue::Int(DALI_INTERP_LINEAR), true)
    .AddOptionalArg("mag_filter", "Filter used when scaling up.", ArgValue::Int(DALI_INTERP_LINEAR), true)
    .AddOptionalArg("min_filter", "Filter used when scaling down.", ArgValue::Int(DALI_INTERP_LINEAR), true)
    .AddOptionalArg("antialias", "If enabled, it applies an antialiasing filter when scaling down.", ArgValue::Bool(true))
    .AddOptionalTypeArg("dtype", "Output data type: the input type (uint8, int16, uint16 or float) or FLOAT.", ArgType::INT)
    .AddOptionalArg("temp_buffer_hint", "Ignored (the intermediate lives in LDS).", ArgValue::Int(0))
    .AddOptionalArg("minibatch_size", "Ignored (one launch per batch).", ArgValue::Int(32));

DALI_SCHEMA(RandomCropAttr)
    .DocStr("Random Crop attributes placeholder.")
    .MakeInternal()
    .AddOptionalArg("random_aspect_ratio", "Range from which to choose random aspect ratio (width/height).",
                    ArgValue::FloatVec({3.0 / 4, 4.0 / 3}))
    .AddOptionalArg("random_area", "Range from which to choose random area fraction A.", ArgValue::FloatVec({0.08, 1.0}))
    .AddOptionalArg("num_attempts", "Maximum number of attempts used to choose random area and aspect ratio.",
                    ArgValue::Int(10))
    .AddRandomSeedArg();

DALI_SCHEMA(RandomResizedCrop)
    .DocStr("Performs a crop with a randomly selected area and aspect ratio and resizes it to the specified size.\n\n"
            "Expects an input with samples in height, width, channels (HWC) layout, or any layout in which W follows H "
            "(FHWC video, CHW, FCHW, CFHW): the dimensions in front of H are frames that share the sample's crop window, the "
            "ones behind W channels (1 to 4).")
    .NumInput(1)
    .NumOutput(1)
    .AddArg("size", "Size of the resized image.", ArgType::INT_VEC)
    .AddParent("RandomCropAttr")
    .AddParent("ResamplingFilterAttr")
    .AllowSequences()
    .InputLayout(0, {"HWC", "CHW", "FHWC", "FCHW", "CFHW"});

// per-operator filter arguments -> kernel enums (ResamplingFilterAttr::PrepareFilterParams, resampling_attr.cc:76-121)
struct FilterArgs {
  int min_filter = DALIAMD_INTERP_LINEAR, mag_filter = DALIAMD_INTERP_LINEAR, antialias = 1;
  // `interp_type` / `min_filter` / `mag_filter` may be tensor arguments, one value per sample (resampling_attr.cc:25-37: the
  // trailing `true` of their schema entries; GetPerSampleArgument in PrepareFilterParams): Resolve() reads them per batch
  bool per_sample = false;
  explicit FilterArgs(const OpSpec &spec) {
    t_interp_ = spec.HasTensorArgument("interp_type"); t_min_ = spec.HasTensorArgument("min_filter");
    t_mag_ = spec.HasTensorArgument("mag_filter");
    per_sample = t_interp_ || t_min_ || t_mag_;
    antialias = spec.GetBool("antialias");
    bool has_interp = spec.Args().count("interp_type"), has_min = spec.Args().count("min_filter"),
         has_mag = spec.Args().count("mag_filter");
    if (has_min) min_filter = ToKernelInterp(spec.GetInt("min_filter"));
    else if (has_interp) min_filter = ToKernelInterp(spec.GetInt("interp_type"));
    if (has_mag) mag_filter = ToKernelInterp(spec.GetInt("mag_filter"));
    else if (has_interp) mag_filter = ToKernelInterp(spec.GetInt("interp_type"));
    if (const ArgValue *d = spec.TryArg("dtype")) dtype = (int)d->i;
  }
  // the filters of the batch's samples: min_filter / mag_filter win over interp_type, tensor or not, as in the reference
  void Resolve(const OpSpec &spec, const Workspace &ws, int n) {
    if (!per_sample) return;
    min_v_.assign(n, min_filter);
    mag_v_.assign(n, mag_filter);
    auto read = [&](const char *name) {
      std::vector<int> v = GetPerSampleInt(spec, ws, name, n), out(n);
      for (int i = 0; i < n; i++) out[i] = ToKernelInterp(v[i]);
      return out;
    };
    if (t_interp_) {
      const std::vector<int> v = read("interp_type");
      if (!t_min_ && !spec.Args().count("min_filter")) min_v_ = v;
      if (!t_mag_ && !spec.Args().count("mag_filter")) mag_v_ = v;
    }
    if (t_min_) min_v_ = read("min_filter");
    if (t_mag_) mag_v_ = read("mag_filter");
  }
  int Min(int i) const { return per_sample ? min_v_[i] : min_filter; }
  int Mag(int i) const { return per_sample ? mag_v_[i] : mag_filter; }

 private:
  bool t_interp_ = false, t_min_ = false, t_mag_ = false;
  std::vector<int> min_v_, mag_v_;

 public:
  int dtype = -1;  // `dtype` argument: absent = the input's type

  // element types of one sample: u8 / i16 / u16 / f32 in (resize_base.cc:41,49), out = the input's type or FLOAT (the
  // unrounded result).  Returns the DALI type of the output.
  DALIDataType ApplyTypes(daliamdResampleArgs &a, DALIDataType in_type, const char *op) const {
    switch (in_type) {
      case DALI_UINT8: a.in_dtype = DALIAMD_UINT8; break;
      case DALI_INT16: a.in_dtype = DALIAMD_INT16; break;
      case DALI_UINT16: a.in_dtype = DALIAMD_UINT16; break;
      case DALI_FLOAT: a.in_dtype = DALIAMD_FLOAT; break;
      default: DALI_FAIL(op, ": unsupported input type ", TypeName(in_type), " (supported: uint8, int16, uint16, float)");
    }
    DALI_ENFORCE(dtype < 0 || dtype == (int)in_type || dtype == DALI_FLOAT, op,
                 ": the output type must be the input type or FLOAT, got ", TypeName((DALIDataType)dtype));
    if (dtype == DALI_FLOAT && in_type != DALI_FLOAT) {
      a.unrounded = 1;
      a.out_dtype = DALIAMD_FLOAT;
      return DALI_FLOAT;
    }
    a.unrounded = 0;
    a.out_dtype = a.in_dtype;
    return in_type;
  }
};

static void LaunchResample(Workspace &ws, DescUploader &up, std::vector<daliamdResampleArgs> &args,
                           std::vector<daliamdResampleDesc> &descs, const char *what) {
  int n = (int)args.size();
  if (!n) return;
  descs.resize(n);
  daliamdResamplePlan plan{};
  KCHECK(daliamdResampleSetup(args.data(), n, descs.data(), &plan));
  if (ws.backend == OpType::CPU) {
    // CPU backend: the same descriptors, one thread-pool task per sample (resize_op_impl_cpu.h:84-107)
    for (int i = 0; i < n; i++)
      ws.GetThreadPool().AddWork([&descs, i](int) {
        if (daliamdResampleRunHost(&descs[i]) != 0) DALI_FAIL(daliamdHostGetLastErrorMessage());
      }, (int64_t)descs[i].in_h * descs[i].in_w);
    ws.GetThreadPool().RunAll();
    NoteLaunch(ws, std::string("host_") + what);
    return;
  }
  // The descriptor upload and the table launch need host data only: on the side stream they run while ws.stream is still
  // with the decoder's kernels of this iteration, and the passes wait for them through the slot's event.
  daliamdStream_t side = ws.aux_stream ? ws.aux_stream : ws.stream;
  auto *dev = static_cast<const daliamdResampleDesc *>(
      up.Upload(descs.data(), descs.size() * sizeof(descs[0]), side, ws.ring + 1, plan.workspace_bytes));
  KCHECK(daliamdResampleRunTables(side, dev, n, &plan, up.Scratch()));
  if (side != ws.stream) {
    KCHECK(daliamdEventRecord(up.LastEvent(), side));
    KCHECK(daliamdStreamWaitEvent(ws.stream, up.LastEvent()));
  }
  KCHECK(daliamdResampleRunPasses(ws.stream, dev, n, &plan, up.Scratch()));
  NoteLaunch(ws, what);
}

// A sample of the resizing operators as FRAMES of H x W x C images, the way the reference's resize implementation sees any
// layout in which W follows H (ResizeBase::SetupResize(.., first_spatial_dim), resize_base.cc:33-52, resize_op_impl_*:
// the dimensions in front of H are flattened into frames, the ones behind W into channels): "HWC" one frame, "FHWC" video,
// "CHW" C single-channel frames, "FCHW" / "CFHW" F x C of them.  Frames of a sample share its arguments (region of
// interest, filters - one crop window per video, random_resized_crop.h:62-85).
struct FrameView {
  int64_t frames = 1;
  int h = 0, w = 0, c = 1, hi = 0;   // hi: index of H in the shape
  bool plain_hwc = true;             // three-dimensional HWC: the case the fused paths are written for
};
static FrameView SplitFrames(const TensorList &in, int i, const char *op) {
  const TensorShape &s = in.shape(i);
  std::string layout = in.layout();
  if (layout.empty()) layout = s.size() == 3 ? "HWC" : s.size() == 2 ? "HW" : "";
  const size_t hi = layout.find('H');
  DALI_ENFORCE(layout.size() == s.size() && hi != std::string::npos && hi + 1 < layout.size() && layout[hi + 1] == 'W', op,
               ": expected a layout in which W follows H (HWC, FHWC, CHW, FCHW, CFHW, HW) that matches the ", s.size(),
               "-dimensional input, got \"", in.layout(), "\"");
  FrameView v;
  v.hi = (int)hi;
  for (size_t k = 0; k < hi; k++) v.frames *= s[k];
  v.h = (int)s[hi]; v.w = (int)s[hi + 1];
  int64_t c = 1;
  for (size_t k = hi + 2; k < s.size(); k++) c *= s[k];
  DALI_ENFORCE(c >= 1 && c <= 4, op, ": ", c, " channels behind the width (supported: 1..4; put the channels in front - CHW - for more)");
  v.c = (int)c;
  v.plain_hwc = layout == "HWC";
  const int64_t dense = (int64_t)v.w * v.c * (int64_t)TypeSize(in.type());
  DALI_ENFORCE(v.plain_hwc || !in.row_pitch(i) || in.row_pitch(i) == dense, op, ": frames need densely packed rows");
  return v;
}
static void FillSourceArgs(daliamdResampleArgs &a, const TensorList &in, int i, const FrameView &v, int64_t frame) {
  const int64_t row = (int64_t)v.w * v.c * (int64_t)TypeSize(in.type());
  a.in_h = v.h; a.in_w = v.w; a.channels = v.c;
  a.in_pitch = (int32_t)(v.plain_hwc && in.row_pitch(i) ? in.row_pitch(i) : row);   // bytes
  a.in = static_cast<const uint8_t *>(in.raw(i)) + frame * (int64_t)v.h * a.in_pitch;
}
// the output shape of a sample: its input shape with H and W replaced
static TensorShape ResizedShape(const TensorList &in, int i, const FrameView &v, int out_h, int out_w) {
  TensorShape s = in.shape(i);
  s[v.hi] = out_h; s[v.hi + 1] = out_w;
  return s;
}

class RandomResizedCropGpu : public OperatorBase {
 public:
  explicit RandomResizedCropGpu(const OpSpec &spec) : OperatorBase(spec), filters_(spec) {
    auto sz = spec.GetIntVec("size");
    if (sz.size() == 1) sz.push_back(sz[0]);
    DALI_ENFORCE(sz.size() == 2 && sz[0] > 0 && sz[1] > 0, "`size` must hold two positive values (H, W)");
    out_h_ = (int)sz[0]; out_w_ = (int)sz[1];
    auto ar = spec.GetFloatVec("random_aspect_ratio"), area = spec.GetFloatVec("random_area");
    if (ar.size() == 1) ar.push_back(ar[0]);
    if (area.size() == 1) area.push_back(area[0]);
    DALI_ENFORCE(ar.size() == 2 && ar[0] <= ar[1], "Provided empty range");
    DALI_ENFORCE(area.size() == 2 && area[0] <= area[1], "Provided empty range");
    ar_lo_ = (float)ar[0]; ar_hi_ = (float)ar[1]; area_lo_ = (float)area[0]; area_hi_ = (float)area[1];
    num_attempts_ = (int)spec.GetInt("num_attempts");
    master_.key = (uint64_t)spec.GetInt("seed");
    master_.ctr[0] = master_.ctr[1] = 0;
    master_.phase = 0;
  }
  void EnableFusion() { fused_ = true; }

  // The producer decodes only the windows (ImageDecoderMixed::SetRoiSource): it asks for them when ITS stage runs, an
  // iteration or more ahead of this operator's Setup, which then finds its input already cropped.  Same generator, same
  // order of draws, same advance per batch as without the fusion.
  // (the Gaussian window's reach is not a multiple of the scale that is written down anywhere: no fusion for it)
  bool CanTakeCroppedInput() const {
    // (per-sample filters: the reach of a window would differ from sample to sample and is only known with the batch)
    return !filters_.per_sample && filters_.min_filter != DALIAMD_INTERP_GAUSSIAN && filters_.mag_filter != DALIAMD_INTERP_GAUSSIAN &&
           filters_.dtype < 0;
  }
  void ExpectCroppedInput() { cropped_input_ = true; }
  // rois[4 i ..] = the window of image i to DECODE: the crop window plus the reach of the resampling filter on every side
  // (the filters' taps at the window's edge are pixels of the image, not repetitions of the edge), cut to the image.  The
  // resampling set-up is told where that window sits (daliamdResampleArgs.full_h ...) and refuses one that is too small.
  void DrawWindows(int64_t iteration, int n, const int32_t *shapes_hw, int32_t *rois) {
    Drawn d;
    d.iteration = iteration;
    d.anchors.resize(2 * n); d.crops.resize(2 * n); d.full_hw.assign(shapes_hw, shapes_hw + 2 * n); d.window.resize(4 * n);
    std::lock_guard<std::mutex> g(drawn_mu_);
    d.state_before = master_;
    if (daliamdRandomCropBatch(&master_, n, shapes_hw, ar_lo_, ar_hi_, area_lo_, area_hi_, num_attempts_, d.anchors.data(),
                               d.crops.data()) != 0)
      DALI_FAIL(daliamdHostGetLastErrorMessage());
    daliamdPhiloxAdvanceSequence(&master_, (uint64_t)n);  // OperatorWithRng::Advance(batch)
    const int out[2] = {out_h_, out_w_};
    for (int i = 0; i < n; i++)
      for (int a = 0; a < 2; a++) {   // a = 0: rows, 1: columns
        const int crop = d.crops[2 * i + a], anchor = d.anchors[2 * i + a], size = shapes_hw[2 * i + a];
        const int type = out[a] < crop ? filters_.min_filter : filters_.mag_filter;
        const float per_unit = type == DALIAMD_INTERP_LANCZOS3 ? 3.0f : type == DALIAMD_INTERP_CUBIC ? 2.0f : 1.0f;
        const float ratio = filters_.antialias && crop > out[a] ? (float)crop / (float)out[a] : 1.0f;
        const int reach = (int)std::ceil(per_unit * ratio) + 2;
        // The window starts on the colour kernel's grid - a multiple of 8 columns, an even row -: every upright sample then
        // takes its planes-aligned fast path, ONE launch per batch where windows that start anywhere need up to three
        // (row-by-row instance for 4:4:4 / 4:2:2 / gray samples, any-origin 4:2:0 instance, aligned instance: 0.18 ms of
        // mostly idle launches inside the schedule, profiles/r05_kernel_stats.csv).  At most 7 more columns are converted;
        // the value passes work on whole MCUs anyway.
        const int grid = a == 1 ? 8 : 2;
        const int lo = std::max(0, anchor - reach) / grid * grid, hi = std::min(size, anchor + crop + reach);
        d.window[4 * i + a] = lo; d.window[4 * i + 2 + a] = hi - lo;
        rois[4 * i + a] = lo; rois[4 * i + 2 + a] = hi - lo;
      }
    drawn_.push_back(std::move(d));
  }
  // The producer failed behind its draw for `iteration`: that iteration never reaches this operator.  The generator goes
  // back to where it stood, so the next iteration draws what it would have drawn without the fusion (there the operator
  // simply does not run in a failed iteration).  The producer draws in iteration order: the entry is the newest one.
  void UndoDraw(int64_t iteration) {
    std::lock_guard<std::mutex> g(drawn_mu_);
    if (drawn_.empty() || drawn_.back().iteration != iteration) return;
    master_ = drawn_.back().state_before;
    drawn_.pop_back();
  }

  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    int n = in.num_samples();
    shapes_hw_.resize(2 * n); anchors_.resize(2 * n); crops_.resize(2 * n);
    filters_.Resolve(spec_, ws, n);
    // one resampling job per FRAME (SplitFrames); a sample's frames share its window
    views_.resize(n);
    arg_sample_.clear(); arg_frame_.clear();
    bool plain = true;
    for (int i = 0; i < n; i++) {
      views_[i] = SplitFrames(in, i, "RandomResizedCrop");
      plain = plain && views_[i].plain_hwc;
      for (int64_t f = 0; f < views_[i].frames; f++) { arg_sample_.push_back(i); arg_frame_.push_back(f); }
    }
    const int na = (int)arg_sample_.size();
    args_.assign(na, daliamdResampleArgs{});
    int ch = 3;
    for (int k = 0; k < na; k++) FillSourceArgs(args_[k], in, arg_sample_[k], views_[arg_sample_[k]], arg_frame_[k]);
    for (int i = 0; i < n; i++) {
      shapes_hw_[2 * i] = views_[i].h; shapes_hw_[2 * i + 1] = views_[i].w;
      ch = views_[i].c;
    }
    if (cropped_input_) {
      // the windows of this iteration were drawn when the producer ran: the input IS the window
      std::lock_guard<std::mutex> g(drawn_mu_);
      // (windows of an iteration that never got here - belt and braces next to UndoDraw - are not this iteration's)
      while (!drawn_.empty() && drawn_.front().iteration < (int64_t)ws.iteration) drawn_.pop_front();
      DALI_ENFORCE(!drawn_.empty() && drawn_.front().iteration == (int64_t)ws.iteration &&
                       (int)drawn_.front().crops.size() == 2 * n,
                   "internal: RandomResizedCrop expected the windows its producer decoded");
      Drawn d = std::move(drawn_.front());
      drawn_.pop_front();
      crops_ = std::move(d.crops);
      anchors_ = std::move(d.anchors);
      DALI_ENFORCE(plain && na == n, "internal: RandomResizedCrop expected HWC images from its producer");
      for (int i = 0; i < n; i++) {
        DALI_ENFORCE(shapes_hw_[2 * i] == d.window[4 * i + 2] && shapes_hw_[2 * i + 1] == d.window[4 * i + 3],
                     "internal: RandomResizedCrop expected a ", d.window[4 * i + 2], " x ", d.window[4 * i + 3], " window, got ",
                     shapes_hw_[2 * i], " x ", shapes_hw_[2 * i + 1]);
        args_[i].full_h = d.full_hw[2 * i]; args_[i].full_w = d.full_hw[2 * i + 1];
        args_[i].org_y = d.window[4 * i]; args_[i].org_x = d.window[4 * i + 1];
      }
    } else if (daliamdRandomCropBatch(&master_, n, shapes_hw_.data(), ar_lo_, ar_hi_, area_lo_, area_hi_, num_attempts_,
                                      anchors_.data(), crops_.data()) != 0) {
      DALI_FAIL(daliamdHostGetLastErrorMessage());
    }
    for (int k = 0; k < na; k++) {
      const int i = arg_sample_[k];
      auto &a = args_[k];
      a.use_roi = 1;
      a.roi_y0 = (float)anchors_[2 * i]; a.roi_x0 = (float)anchors_[2 * i + 1];
      a.roi_y1 = (float)(anchors_[2 * i] + crops_[2 * i]); a.roi_x1 = (float)(anchors_[2 * i + 1] + crops_[2 * i + 1]);
      a.out_h = out_h_; a.out_w = out_w_;
      a.min_filter = filters_.Min(i); a.mag_filter = filters_.Mag(i); a.antialias = filters_.antialias;
      a.out_layout = DALIAMD_LAYOUT_HWC;
      out_type_ = filters_.ApplyTypes(a, in.type(), "RandomResizedCrop");
    }
    n_ = n; ch_ = ch;
    plain_u8_ = plain && in.type() == DALI_UINT8 && out_type_ == DALI_UINT8;
    layout_ = in.layout().empty() ? std::string("HWC") : in.layout();
    if (fused_ && plain_u8_) return false;  // no buffer: the consumer launches the fused kernel
    desc[0].type = n ? out_type_ : in.type();
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) desc[0].shape[i] = ResizedShape(in, i, views_[i], out_h_, out_w_);
    return true;
  }

  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    if (fused_ && plain_u8_) {
      auto d = std::make_shared<DeferredResample>();
      d->source = ws.inputs[0];
      d->args = args_;
      d->out_h = out_h_; d->out_w = out_w_; d->channels = ch_;
      out.Resize({}, DALI_UINT8);
      out.deferred = d;
      out.SetLayout("HWC");
    } else {
      out.deferred.reset();
      out.SetLayout(layout_);
      const int64_t frame_bytes = (int64_t)out_h_ * out_w_ * ch_ * (int64_t)TypeSize(out_type_);
      for (size_t k = 0; k < args_.size(); k++) {
        // the kernel writes dense HWC rows; request that from the executor by a dense pitch
        args_[k].out = static_cast<uint8_t *>(out.raw(arg_sample_[k])) + arg_frame_[k] * frame_bytes;
      }
      DALI_ENFORCE(out.is_dense() || n_ == 0, "internal: resample output must be dense");
      LaunchResample(ws, uploader_, args_, descs_, "resample");
    }
    if (!cropped_input_) daliamdPhiloxAdvanceSequence(&master_, (uint64_t)n_);  // OperatorWithRng::Advance(batch)
  }

  std::string SaveState() const override {
    char buf[96];
    std::lock_guard<std::mutex> g(drawn_mu_);
    // (windows drawn for iterations this operator has not run yet are not part of what has been handed out)
    daliamdPhiloxStateToString(drawn_.empty() ? &master_ : &drawn_.front().state_before, buf, sizeof(buf));
    return buf;
  }
  void RestoreState(const std::string &s) override {
    std::lock_guard<std::mutex> g(drawn_mu_);
    drawn_.clear();
    DALI_ENFORCE(daliamdPhiloxStateFromString(&master_, s.c_str()) == 0, daliamdHostGetLastErrorMessage());
  }

 private:
  struct Drawn {
    int64_t iteration = 0;
    daliamdPhiloxState state_before;
    std::vector<int32_t> anchors, crops;   // the crop windows, in image coordinates
    std::vector<int32_t> full_hw, window;  // the images' sizes; what the producer decodes: {y0, x0, h, w} per sample
  };
  bool cropped_input_ = false;
  mutable std::mutex drawn_mu_;
  std::deque<Drawn> drawn_;
  FilterArgs filters_;
  int out_h_, out_w_, num_attempts_, n_ = 0, ch_ = 3;
  float ar_lo_, ar_hi_, area_lo_, area_hi_;
  daliamdPhiloxState master_;
  bool fused_ = false, plain_u8_ = true;
  DALIDataType out_type_ = DALI_UINT8;
  std::vector<int32_t> shapes_hw_, anchors_, crops_;
  std::vector<FrameView> views_;                 // per sample
  std::vector<int> arg_sample_;                  // per resampling job (frame): its sample ...
  std::vector<int64_t> arg_frame_;               // ... and its frame inside it
  std::string layout_ = "HWC";
  std::vector<daliamdResampleArgs> args_;
  std::vector<daliamdResampleDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(RandomResizedCrop, RandomResizedCropGpu, GPU);
DALI_REGISTER_OPERATOR(RandomResizedCrop, RandomResizedCropGpu, CPU);

// =============================================================================================
// Resize (fn.resize): ResizeAttr size arithmetic on the host + the same resampling kernel
//   schema / flags      dali/operators/image/resize/resize_attr.cc:24-113, resize_attr_base.cc:22-86
//   AdjustOutputSize    dali/operators/image/resize/resize_attr_base.cc:88-188
//   CalculateSampleParams  dali/operators/image/resize/resize_attr_base.h:50-116
//   operator            dali/operators/image/resize/resize.cc:21-96
// =============================================================================================
DALI_SCHEMA(ResizeAttrBase)
    .DocStr("Resize attributes placeholder")
    .MakeInternal()
    .AddOptionalArg("mode", "Resize mode: \"default\" (missing extents keep the aspect ratio), \"stretch\" (missing "
                    "extents are not scaled), \"not_larger\", \"not_smaller\" (keep the aspect ratio so that no extent "
                    "exceeds / falls below the requested size).", ArgValue::Str("default"))
    .AddOptionalArg("subpixel_scale", "If True, fractional sizes, directly specified or calculated, will cause the input "
                    "ROI to be adjusted to keep the scale factor.", ArgValue::Bool(true))
    .AddOptionalTypeArg("roi_start", "Origin of the input region of interest (ROI), in the order of `size` (y, x).",
                        ArgType::FLOAT_VEC, true)
    .AddOptionalTypeArg("roi_end", "End of the input region of interest (ROI).", ArgType::FLOAT_VEC, true)
    .AddOptionalArg("roi_relative", "If true, ROI coordinates are relative to the input size.", ArgValue::Bool(false))
    .AddOptionalTypeArg("max_size", "Limit of the output size (one value, or one per axis).", ArgType::FLOAT_VEC);

DALI_SCHEMA(ResizeAttr)
    .DocStr("Resize attributes placeholder")
    .MakeInternal()
    .AddOptionalArg("resize_x", "The length of the X dimension of the resized image.", ArgValue::Float(0.0), true)
    .AddOptionalArg("resize_y", "The length of the Y dimension of the resized image.", ArgValue::Float(0.0), true)
    .AddOptionalTypeArg("size", "The desired output size (H, W); 0 = derive from the other extent and `mode`.",
                        ArgType::FLOAT_VEC, true)
    .AddOptionalArg("resize_shorter", "The length of the shorter dimension of the resized image.", ArgValue::Float(0.0), true)
    .AddOptionalArg("resize_longer", "The length of the longer dimension of the resized image.", ArgValue::Float(0.0), true)
    .AddParent("ResizeAttrBase");

DALI_SCHEMA(Resize)
    .DocStr("Resize images.\n\nExpects an input in HWC layout, or any layout in which W follows H (FHWC video, CHW, FCHW, "
            "CFHW): the dimensions in front of H are frames that share the sample's arguments, the ones behind W channels (1 to 4).")
    .NumInput(1)
    .NumOutput(1)
    .AddParent("ResizeAttr")
    .AddParent("ResamplingFilterAttr")
    .AllowSequences()
    .InputLayout(0, {"HWC", "CHW", "FHWC", "FCHW", "CFHW"});

enum class ResizeMode { Default, Stretch, NotLarger, NotSmaller };

// out_size: requested (0 = unspecified) -> final fractional size per dimension
static void AdjustOutputSize(float *out_size, const float *in_size, int ndim, ResizeMode mode, const float *max_size) {
  double scale[3] = {1, 1, 1};
  bool given[3] = {false, false, false};
  int ngiven = 0;
  for (int d = 0; d < ndim; d++) {
    given[d] = out_size[d] != 0 && in_size[d] != 0;
    scale[d] = in_size[d] ? out_size[d] / in_size[d] : 1;
    ngiven += given[d];
  }
  if (ngiven == 0) {  // nothing to go by: keep the size
    for (int d = 0; d < ndim; d++) out_size[d] = in_size[d];
    return;
  }
  auto apply_limit = [&] {
    if (!max_size) return;
    for (int d = 0; d < ndim; d++)
      if (max_size[d] > 0 && std::fabs(out_size[d]) > max_size[d]) {
        out_size[d] = std::copysign(max_size[d], out_size[d]);
        scale[d] = out_size[d] / in_size[d];
      }
  };
  if (mode == ResizeMode::Default) {
    if (ngiven < ndim) {  // the missing extents follow the (geometric) mean scale of the given ones
      double mean_scale = 1;
      for (int d = 0; d < ndim; d++)
        if (given[d]) mean_scale *= std::fabs(scale[d]);
      if (ngiven > 1) mean_scale = std::pow(mean_scale, 1.0 / ngiven);
      for (int d = 0; d < ndim; d++)
        if (!given[d]) {
          scale[d] = mean_scale;
          out_size[d] = (float)(in_size[d] * scale[d]);
        }
    }
    apply_limit();
  } else if (mode == ResizeMode::Stretch) {
    for (int d = 0; d < ndim; d++)
      if (!given[d]) { scale[d] = 1; out_size[d] = in_size[d]; }
    apply_limit();
  } else {
    double final_scale = 0;
    bool first = true;
    for (int d = 0; d < ndim; d++) {
      if (!given[d]) continue;
      float sc = (float)std::fabs(scale[d]);
      if (first || (mode == ResizeMode::NotSmaller && sc > final_scale) || (mode == ResizeMode::NotLarger && sc < final_scale))
        final_scale = sc;
      first = false;
    }
    if (max_size)
      for (int d = 0; d < ndim; d++)
        if (max_size[d] > 0) final_scale = std::min(final_scale, (double)max_size[d] / in_size[d]);
    for (int d = 0; d < ndim; d++)
      if (!given[d] || std::fabs(scale[d]) != final_scale) {
        scale[d] = std::copysign(final_scale, scale[d]);
        out_size[d] = (float)(in_size[d] * scale[d]);
      }
  }
}

class ResizeGpu : public OperatorBase {
 public:
  explicit ResizeGpu(const OpSpec &spec) : OperatorBase(spec), filters_(spec) {
    has_shorter_ = spec.ArgumentDefined("resize_shorter");
    has_longer_ = spec.ArgumentDefined("resize_longer");
    has_x_ = spec.ArgumentDefined("resize_x");
    has_y_ = spec.ArgumentDefined("resize_y");
    has_size_ = spec.ArgumentDefined("size");
    has_max_size_ = spec.ArgumentDefined("max_size");
    bool has_mode = spec.ArgumentDefined("mode");
    subpixel_scale_ = spec.GetBool("subpixel_scale");
    DALI_ENFORCE((int)(has_x_ || has_y_) + has_size_ + has_shorter_ + has_longer_ == 1,
                 "Exactly one method of specifying size must be used. The available methods:\n"
                 "    - separate resize_x, resize_y, resize_z arguments\n    - size argument\n    - resize_longer\n"
                 "    - resize_shorter");
    DALI_ENFORCE(has_shorter_ + has_longer_ + has_mode <= 1,
                 "`resize_shorter`, ``resize_longer`` and ``mode`` arguments are mutually exclusive");
    bool roi_s = spec.ArgumentDefined("roi_start"), roi_e = spec.ArgumentDefined("roi_end");
    DALI_ENFORCE(roi_s == roi_e, "``roi_start`` and ``roi_end`` must be specified together");
    has_roi_ = roi_s;
    roi_relative_ = spec.GetBool("roi_relative");
    if (has_shorter_) mode_ = ResizeMode::NotSmaller;
    else if (has_longer_) mode_ = ResizeMode::NotLarger;
    else if (has_mode) {
      std::string m = spec.GetString("mode");
      if (m == "default") mode_ = ResizeMode::Default;
      else if (m == "stretch") mode_ = ResizeMode::Stretch;
      else if (m == "not_larger") mode_ = ResizeMode::NotLarger;
      else if (m == "not_smaller") mode_ = ResizeMode::NotSmaller;
      else DALI_FAIL("Invalid resize mode: \"", m, "\"");
    }
    if (has_max_size_) {
      auto ms = spec.GetFloatVec("max_size");
      DALI_ENFORCE(ms.size() == 1 || ms.size() == 2, "`max_size` must hold one value or one per spatial dimension");
      max_size_[0] = (float)ms[0];
      max_size_[1] = (float)(ms.size() == 2 ? ms[1] : ms[0]);
    }
  }
  void EnableFusion() { fused_ = true; }

  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    const int n = in.num_samples();
    std::vector<float> rx, ry, rs;
    std::vector<std::vector<float>> size, roi_start, roi_end;
    if (has_x_) rx = GetPerSampleFloat(spec_, ws, "resize_x", n);
    if (has_y_) ry = GetPerSampleFloat(spec_, ws, "resize_y", n);
    if (has_shorter_) rs = GetPerSampleFloat(spec_, ws, "resize_shorter", n);
    if (has_longer_) rs = GetPerSampleFloat(spec_, ws, "resize_longer", n);
    if (has_size_) size = GetPerSampleFloatVec(spec_, ws, "size", n);
    if (has_roi_) {
      roi_start = GetPerSampleFloatVec(spec_, ws, "roi_start", n);
      roi_end = GetPerSampleFloatVec(spec_, ws, "roi_end", n);
    }
    filters_.Resolve(spec_, ws, n);
    // one resampling job per FRAME (SplitFrames): a sample's arguments are worked out once, on its first frame, and copied
    views_.resize(n);
    arg_sample_.clear(); arg_frame_.clear();
    first_arg_.assign(n, 0);
    bool plain = true;
    for (int i = 0; i < n; i++) {
      views_[i] = SplitFrames(in, i, "Resize");
      plain = plain && views_[i].plain_hwc;
      first_arg_[i] = (int)arg_sample_.size();
      for (int64_t f = 0; f < std::max<int64_t>(views_[i].frames, 0); f++) { arg_sample_.push_back(i); arg_frame_.push_back(f); }
    }
    args_.assign(arg_sample_.size(), daliamdResampleArgs{});
    desc[0].type = in.type();
    desc[0].shape.resize(n);
    int ch = 3;
    for (int i = 0; i < n; i++) {
      daliamdResampleArgs first{};
      auto &a = views_[i].frames > 0 ? args_[first_arg_[i]] : first;
      FillSourceArgs(a, in, i, views_[i], 0);
      desc[0].type = filters_.ApplyTypes(a, in.type(), "Resize");
      ch = a.channels;
      const int in_hw[2] = {a.in_h, a.in_w};
      float req[2] = {0, 0};  // H, W
      if (has_x_ || has_y_) {
        req[0] = has_y_ ? ry[i] : 0;
        req[1] = has_x_ ? rx[i] : 0;
      } else if (has_shorter_ || has_longer_) {
        req[0] = req[1] = rs[i];
      } else {
        DALI_ENFORCE(size[i].size() == 2, "`size` must have one entry per spatial dimension (2), got ", size[i].size());
        req[0] = size[i][0]; req[1] = size[i][1];
      }
      // CalculateInputRoI
      float lo[2], hi[2], in_size[2];
      for (int d = 0; d < 2; d++) {
        if (has_roi_ && in_hw[d] > 0) {
          DALI_ENFORCE(roi_start[i].size() == 2 && roi_end[i].size() == 2, "`roi_start`/`roi_end` must have 2 entries");
          double l = roi_start[i][d], h = roi_end[i][d];
          if (roi_relative_) { l *= in_hw[d]; h *= in_hw[d]; }
          const float min_size = 1e-3f;  // a degenerate region is widened instead of dividing by zero
          if (std::fabs(h - l) < min_size) {
            float off = l <= h ? 0.5f * min_size : -0.5f * min_size;
            l -= off; h += off;
          }
          lo[d] = (float)l; hi[d] = (float)h;
        } else {
          lo[d] = 0; hi[d] = (float)in_hw[d];
        }
      }
      // CalculateSampleParams
      for (int d = 0; d < 2; d++) {
        float sz = hi[d] - lo[d];
        if (sz < 0) { std::swap(hi[d], lo[d]); req[d] = -req[d]; sz = -sz; }
        in_size[d] = sz;
      }
      AdjustOutputSize(req, in_size, 2, mode_, has_max_size_ ? max_size_ : nullptr);
      int out_hw[2];
      for (int d = 0; d < 2; d++) {
        DALI_ENFORCE(lo[d] != hi[d] || req[d] == 0, "Cannot produce non-empty output from empty input");
        out_hw[d] = std::max(1, (int)std::round(std::fabs(req[d])));
        if (req[d] < 0) std::swap(lo[d], hi[d]);  // a negative size flips: the region is traversed backwards
        if (subpixel_scale_ && (float)out_hw[d] != std::fabs(req[d])) {
          // the rounded size differs from the fractional one: shrink/grow the region around its centre
          double adj = std::min(10.0, std::max(-10.0, (double)out_hw[d] / std::fabs(req[d])));
          double center = 0.5 * lo[d] + 0.5 * hi[d];
          double nlo = std::min(1e9, std::max(-1e9, center + (lo[d] - center) * adj));
          double nhi = std::min(1e9, std::max(-1e9, center + (hi[d] - center) * adj));
          lo[d] = (float)nlo; hi[d] = (float)nhi;
        }
      }
      AdjustSample(ws, i, n, lo, hi, out_hw);
      a.use_roi = 1;
      a.roi_y0 = lo[0]; a.roi_x0 = lo[1]; a.roi_y1 = hi[0]; a.roi_x1 = hi[1];
      a.out_h = out_hw[0]; a.out_w = out_hw[1];
      a.min_filter = filters_.Min(i); a.mag_filter = filters_.Mag(i); a.antialias = filters_.antialias;
      a.out_layout = DALIAMD_LAYOUT_HWC;
      desc[0].shape[i] = ResizedShape(in, i, views_[i], out_hw[0], out_hw[1]);
      for (int64_t f = 1; f < views_[i].frames; f++) {   // the sample's other frames: the same job on the next image
        args_[first_arg_[i] + f] = a;
        FillSourceArgs(args_[first_arg_[i] + f], in, i, views_[i], f);
      }
    }
    n_ = n; ch_ = ch;
    layout_ = in.layout().empty() ? std::string("HWC") : in.layout();
    out_elem_ = (int64_t)TypeSize(desc[0].type);
    // only u8 -> u8 HWC images of one size can be deferred to a consumer's fused kernel
    uniform_ = n > 0 && plain && in.type() == DALI_UINT8 && desc[0].type == DALI_UINT8;
    for (int i = 1; i < n && uniform_; i++) uniform_ &= args_[i].out_h == args_[0].out_h && args_[i].out_w == args_[0].out_w;
    if (fused_ && uniform_) return false;  // the consumer launches the fused kernel
    return true;
  }

  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout(layout_);
    if (fused_ && uniform_) {
      auto d = std::make_shared<DeferredResample>();
      d->source = ws.inputs[0];
      d->args = args_;
      d->out_h = args_[0].out_h; d->out_w = args_[0].out_w; d->channels = ch_;
      out.Resize({}, DALI_UINT8);
      out.deferred = d;
      return;
    }
    out.deferred.reset();
    for (size_t k = 0; k < args_.size(); k++)
      args_[k].out = static_cast<uint8_t *>(out.raw(arg_sample_[k])) +
                     arg_frame_[k] * (int64_t)args_[k].out_h * args_[k].out_w * args_[k].channels * out_elem_;
    LaunchResample(ws, uploader_, args_, descs_, "resample");
  }

 protected:
  // hook for operators that derive their region from the resized image (ResizeCropMirror): lo/hi = source region
  // (y, x), out_hw = output size of sample i
  virtual void AdjustSample(const Workspace &, int, int, float *, float *, int *) {}

 private:
  FilterArgs filters_;
  bool has_shorter_, has_longer_, has_x_, has_y_, has_size_, has_max_size_, has_roi_, roi_relative_, subpixel_scale_;
  ResizeMode mode_ = ResizeMode::Default;
  float max_size_[2] = {0, 0};
  bool fused_ = false, uniform_ = false;
  int n_ = 0, ch_ = 3;
  std::vector<FrameView> views_;                 // per sample
  std::vector<int> arg_sample_, first_arg_;      // per resampling job (frame): its sample; per sample: its first job
  std::vector<int64_t> arg_frame_;
  std::string layout_ = "HWC";
  int64_t out_elem_ = 1;
  std::vector<daliamdResampleArgs> args_;
  std::vector<daliamdResampleDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(Resize, ResizeGpu, GPU);
DALI_REGISTER_OPERATOR(Resize, ResizeGpu, CPU);

// ---- ResizeCropMirror: resize, then a CropAttr window of the resized image, then flips - as ONE resampling of the
// back-projected window (resize_crop_mirror.cc:85-118) ----
DALI_SCHEMA(ResizeCropMirrorAttr)
    .DocStr("ResizeCropMirror attributes placeholder")
    .MakeInternal()
    .AddOptionalArg("mirror", "Mask for flipping: 0 - no flip, 1 - horizontal flip, 2 - vertical flip (bitwise combination).",
                    ArgValue::Int(0), true)
    .AddParent("ResizeAttr")
    .AddParent("CropAttr");

DALI_SCHEMA(ResizeCropMirror)
    .DocStr("Performs a fused resize, crop, mirror operation.\n\nThe result of the operation is equivalent to applying "
            "``resize``, followed by ``crop`` and ``flip``. Internally, the operator calculates the relevant region of "
            "interest and performs a single resizing operation on that region.")
    .NumInput(1)
    .NumOutput(1)
    .AddParent("ResizeCropMirrorAttr")
    .AddParent("ResamplingFilterAttr")
    .InputLayout(0, {"HWC"});

DALI_SCHEMA(FastResizeCropMirror)
    .DocStr("Legacy alias for ResizedCropMirror, with antialiasing disabled by default.")
    .NumInput(1)
    .NumOutput(1)
    .AddParent("ResizeCropMirror")
    .AddOptionalArg("antialias", "If enabled, it applies an antialiasing filter when scaling down.", ArgValue::Bool(false))
    .InputLayout(0, {"HWC"});

class ResizeCropMirrorGpu : public ResizeGpu {
 public:
  explicit ResizeCropMirrorGpu(const OpSpec &spec) : ResizeGpu(spec) {
    std::string r = spec.GetString("rounding");
    DALI_ENFORCE(r == "round" || r == "truncate", "``rounding`` value ", r,
                 " is not supported. Supported values are \"round\", or \"truncate\".");
    round_ = r == "round";
    has_crop_ = spec.ArgumentDefined("crop");
    has_wh_ = spec.ArgumentDefined("crop_w") || spec.ArgumentDefined("crop_h");
    DALI_ENFORCE(spec.ArgumentDefined("crop_w") == spec.ArgumentDefined("crop_h"),
                 "`crop_w` and `crop_h` arguments must be provided together");
    DALI_ENFORCE(!(has_crop_ && has_wh_), "`crop` argument is not compatible with `crop_h`, `crop_w`, `crop_d`");
    if (has_crop_) {
      DALI_ENFORCE(!spec.HasTensorArgument("crop"), "Per-sample `crop` tensors are not supported yet");
      auto c = spec.GetFloatVec("crop");
      DALI_ENFORCE(c.size() == 2, "`crop` argument should have 2 or 3 elements depending on the input data shape");
      crop_[0] = (int)c[0]; crop_[1] = (int)c[1];
    }
  }

 protected:
  void AdjustSample(const Workspace &ws, int i, int n, float *lo, float *hi, int *out_hw) override {
    if (i == 0) {  // per-batch argument fetch
      mirror_ = GetPerSampleInt(spec_, ws, "mirror", n);
      pos_[1] = GetPerSampleFloat(spec_, ws, "crop_pos_x", n);
      pos_[0] = GetPerSampleFloat(spec_, ws, "crop_pos_y", n);
      if (has_wh_) {
        wh_[0] = GetPerSampleFloat(spec_, ws, "crop_h", n);
        wh_[1] = GetPerSampleFloat(spec_, ws, "crop_w", n);
      }
    }
    for (int d = 0; d < 2; d++) {
      // the crop window in the resized image (crop_attr.cc:168-240); a non-positive extent means "whole axis"
      int64_t crop = has_crop_ ? crop_[d] : has_wh_ ? (int64_t)(int)wh_[d][i] : 0;
      float norm = pos_[d][i];
      if (crop <= 0) { crop = out_hw[d]; norm = 0.5f; }
      DALI_ENFORCE(norm >= 0.0f && norm <= 1.0f, "Anchor for dimension ", d, " (", norm, ") is out of range [0.0, 1.0]");
      int64_t anchor = daliamdCropAnchor(norm, crop, out_hw[d], round_);
      // back-projection to source coordinates
      double ratio = ((double)hi[d] - (double)lo[d]) / out_hw[d], offset = lo[d];
      lo[d] = (float)((double)anchor * ratio + offset);
      hi[d] = (float)((double)(anchor + crop) * ratio + offset);
      if (mirror_[i] & (1 << (1 - d))) std::swap(lo[d], hi[d]);   // 1: horizontal (W), 2: vertical (H)
      out_hw[d] = (int)crop;
    }
  }

 private:
  bool round_, has_crop_, has_wh_;
  int crop_[2] = {0, 0};
  std::vector<int> mirror_;
  std::vector<float> pos_[2], wh_[2];
};
DALI_REGISTER_OPERATOR(ResizeCropMirror, ResizeCropMirrorGpu, GPU);
DALI_REGISTER_OPERATOR(ResizeCropMirror, ResizeCropMirrorGpu, CPU);
DALI_REGISTER_OPERATOR(FastResizeCropMirror, ResizeCropMirrorGpu, GPU);
DALI_REGISTER_OPERATOR(FastResizeCropMirror, ResizeCropMirrorGpu, CPU);

// =============================================================================================
// CropMirrorNormalize
// =============================================================================================
DALI_SCHEMA(CropAttr)
    .DocStr("Crops attributes placeholder")
    .MakeInternal()
    .AddOptionalTypeArg("crop", "Shape of the cropped image, specified as a list of values (for example, (crop_H, crop_W)).",
                        ArgType::FLOAT_VEC, true)
    .AddOptionalArg("crop_pos_x", "Normalized (0.0 - 1.0) horizontal position of the cropping window (upper left corner).",
                    ArgValue::Float(0.5), true)
    .AddOptionalArg("crop_pos_y", "Normalized (0.0 - 1.0) vertical position of the cropping window.", ArgValue::Float(0.5), true)
    .AddOptionalArg("crop_pos_z", "Unused (2-D images only).", ArgValue::Float(0.5), true)
    .AddOptionalTypeArg("crop_w", "Cropping window width (in pixels).", ArgType::FLOAT, true)
    .AddOptionalTypeArg("crop_h", "Cropping window height (in pixels).", ArgType::FLOAT, true)
    .AddOptionalTypeArg("crop_d", "Unused (2-D images only).", ArgType::FLOAT, true)
    .AddOptionalArg("rounding", "Determines the rounding function used to convert the starting coordinate of the window "
                    "to an integral value: \"round\" (half away from zero) or \"truncate\".", ArgValue::Str("round"));

DALI_SCHEMA(OutOfBoundsAttr)
    .DocStr("Out-of-bounds slicing attributes placeholder")
    .MakeInternal()
    .AddOptionalArg("out_of_bounds_policy", "Determines the policy when slicing out of bounds of the input: \"error\", "
                    "\"pad\" or \"trim_to_shape\".", ArgValue::Str("error"))
    .AddOptionalArg("fill_values", "Determines padding values and is only relevant if ``out_of_bounds_policy`` is \"pad\".",
                    ArgValue::FloatVec({0.0}));

DALI_SCHEMA(CropMirrorNormalize)
    .DocStr("Performs fused cropping, normalization, format conversion (NHWC to NCHW) if desired, and type casting.\n\n"
            "Normalization takes the input images and produces the output by using the following formula::\n\n"
            "  output = scale * (input - mean) / std + shift")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("dtype", "Output data type. Supported types: FLOAT, FLOAT16, INT8, UINT8.", ArgValue::Int(DALI_FLOAT))
    .AddOptionalArg("output_layout", "Tensor data layout for the output (\"CHW\" or \"HWC\").", ArgValue::Str("CHW"))
    .AddOptionalArg("pad_output", "Determines whether to pad the output so that the number of channels is a power of 2.",
                    ArgValue::Bool(false))
    .AddOptionalArg("mirror", "If nonzero, the image will be flipped (mirrored) horizontally.", ArgValue::Int(0), true)
    .AddOptionalArg("mean", "Mean pixel values for image normalization.", ArgValue::FloatVec({0.0}), true)
    .AddOptionalArg("std", "Standard deviation values for image normalization.", ArgValue::FloatVec({1.0}), true)
    .AddOptionalArg("scale", "The value by which the result is multiplied.", ArgValue::Float(1.0))
    .AddOptionalArg("shift", "The value added to the (scaled) result.", ArgValue::Float(0.0))
    .AddParent("CropAttr")
    .AddParent("OutOfBoundsAttr")
    .InputLayout(0, {"HWC"});

class CropMirrorNormalizeGpu : public OperatorBase {
 public:
  explicit CropMirrorNormalizeGpu(const OpSpec &spec) : OperatorBase(spec) {
    out_type_ = (DALIDataType)spec.GetInt("dtype");
    ToKernelDType(out_type_);  // validates: FLOAT, FLOAT16, INT8, UINT8
    std::string layout = spec.GetString("output_layout");
    DALI_ENFORCE(layout == "CHW" || layout == "HWC" || layout.empty(), "Unsupported output_layout \"", layout,
                 "\": expected \"CHW\" or \"HWC\"");
    chw_ = layout == "CHW";
    pad_output_ = spec.GetBool("pad_output");
    scale_ = (float)spec.GetFloat("scale");
    shift_ = (float)spec.GetFloat("shift");
    std::string policy = spec.GetString("out_of_bounds_policy");
    DALI_ENFORCE(policy == "error" || policy == "pad" || policy == "trim_to_shape", "Unsupported out_of_bounds_policy \"",
                 policy, "\"");
    pad_oob_ = policy == "pad";
    trim_ = policy == "trim_to_shape";
    for (double f : spec.GetFloatVec("fill_values")) fill_.push_back((float)f);
    std::string r = spec.GetString("rounding");
    DALI_ENFORCE(r == "round" || r == "truncate", "Unsupported rounding \"", r, "\"");
    round_ = r == "round";
    DALI_ENFORCE(!spec.HasTensorArgument("mean") && !spec.HasTensorArgument("std"),
                 "Per-sample `mean`/`std` tensors are not supported yet");
    std::vector<float> mean, stdv;
    for (double f : spec.GetFloatVec("mean")) mean.push_back((float)f);
    for (double f : spec.GetFloatVec("std")) stdv.push_back((float)f);
    mean_.resize(std::max(mean.size(), stdv.size()));
    inv_std_.resize(mean_.size());
    int k = daliamdCmnNormArgs(mean.data(), (int)mean.size(), stdv.data(), (int)stdv.size(), scale_, shift_, mean_.data(),
                               inv_std_.data());
    DALI_ENFORCE(k >= 0, daliamdHostGetLastErrorMessage());
    mean_.resize(k);
    inv_std_.resize(k);
    has_crop_ = spec.ArgumentDefined("crop") || spec.ArgumentDefined("crop_w") || spec.ArgumentDefined("crop_h");
  }
  bool CanFuse() const { return !has_crop_ && !pad_output_ && (out_type_ == DALI_FLOAT || out_type_ == DALI_FLOAT16); }
  void ExpectFusedInput() { fused_input_ = true; }

  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    const DeferredResample *def = in.deferred.get();
    // a fused producer may still hand over a real tensor list (fn.resize with per-sample output sizes)
    int n = def ? (int)def->args.size() : in.num_samples();
    if (!def) DALI_ENFORCE(in.type() == DALI_UINT8, "CropMirrorNormalize (gpu): only uint8 input is supported, got ",
                           TypeName(in.type()));
    mirror_ = GetPerSampleInt(spec_, ws, "mirror", n);
    windows_.assign(4 * n, 0);
    desc[0].type = out_type_;
    desc[0].shape.resize(n);
    auto pos_x = GetPerSampleFloat(spec_, ws, "crop_pos_x", n), pos_y = GetPerSampleFloat(spec_, ws, "crop_pos_y", n);
    for (int i = 0; i < n; i++) {
      int64_t H = def ? def->out_h : in.shape(i)[0], W = def ? def->out_w : in.shape(i)[1];
      int64_t C = def ? def->channels : in.shape(i)[2];
      DALI_ENFORCE(C >= 1 && C <= 4, "CropMirrorNormalize (gpu) supports 1..4 channels, got ", C);
      DALI_ENFORCE(mean_.size() <= 1 || (int64_t)mean_.size() == C, "The number of per-channel arguments should match the "
                   "number of channels in the output slice");
      int64_t ch = H, cw = W;
      if (has_crop_) {
        if (spec_.ArgumentDefined("crop")) {
          DALI_ENFORCE(!spec_.HasTensorArgument("crop"), "Per-sample `crop` tensors are not supported yet");
          auto c = spec_.GetFloatVec("crop");
          DALI_ENFORCE(c.size() == 2, "`crop` must hold (crop_H, crop_W)");
          ch = (int64_t)c[0]; cw = (int64_t)c[1];
        }
        if (spec_.ArgumentDefined("crop_h")) ch = (int64_t)GetPerSampleFloat(spec_, ws, "crop_h", n)[i];
        if (spec_.ArgumentDefined("crop_w")) cw = (int64_t)GetPerSampleFloat(spec_, ws, "crop_w", n)[i];
      }
      DALI_ENFORCE(ch > 0 && cw > 0, "Crop window must have a positive size");
      DALI_ENFORCE(pos_x[i] >= 0.0f && pos_x[i] <= 1.0f && pos_y[i] >= 0.0f && pos_y[i] <= 1.0f,
                   "Anchor for dimension ", 0, " is out of range [0.0, 1.0]");
      int64_t ay = daliamdCropAnchor(pos_y[i], ch, H, round_), ax = daliamdCropAnchor(pos_x[i], cw, W, round_);
      if (ay < 0 || ax < 0 || ay + ch > H || ax + cw > W) {
        if (trim_) {
          int64_t y0 = std::max<int64_t>(ay, 0), x0 = std::max<int64_t>(ax, 0);
          ch = std::min(ay + ch, H) - y0; cw = std::min(ax + cw, W) - x0;
          ay = y0; ax = x0;
        } else if (!pad_oob_) {
          DALI_FAIL("Slice can't be place out of bounds with current policy. Got: input_shape={", H, ", ", W,
                    "}, slice_shape={", ch, ", ", cw, "}, slice_anchor={", ay, ", ", ax, "}");
        }
      }
      windows_[4 * i] = (int)ay; windows_[4 * i + 1] = (int)ax; windows_[4 * i + 2] = (int)ch; windows_[4 * i + 3] = (int)cw;
      int64_t co = C;
      if (pad_output_) { co = 1; while (co < C) co <<= 1; }
      desc[0].shape[i] = chw_ ? TensorShape{co, ch, cw} : TensorShape{ch, cw, co};
    }
    return true;
  }

  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    out.SetLayout(chw_ ? "CHW" : "HWC");
    const DeferredResample *def = in.deferred.get();
    int n = out.num_samples();
    if (!n) return;
    if (def) {
      // ---- fused RandomResizedCrop/Resize + CMN: one kernel, the u8 intermediate never exists ----
      rargs_ = def->args;
      for (int i = 0; i < n; i++) {
        auto &a = rargs_[i];
        a.out = out.raw(i);
        a.out_dtype = ToKernelDType(out_type_);
        a.out_layout = chw_ ? DALIAMD_LAYOUT_CHW : DALIAMD_LAYOUT_HWC;
        a.normalize = !mean_.empty();
        a.mirror = mirror_[i] != 0;
        FillNorm(a.mean, a.inv_std);
      }
      LaunchResample(ws, uploader_, rargs_, rdescs_, "fused_resample_cmn");
      return;
    }
    descs_.assign(n, daliamdCmnDesc{});
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      const TensorShape &s = in.shape(i);
      d.in = static_cast<const uint8_t *>(in.raw(i));
      d.in_h = (int32_t)s[0]; d.in_w = (int32_t)s[1]; d.channels = (int32_t)s[2];
      d.in_pitch = (int32_t)(in.row_pitch(i) ? in.row_pitch(i) : s[1] * s[2]);
      d.anchor_y = windows_[4 * i]; d.anchor_x = windows_[4 * i + 1];
      d.crop_h = windows_[4 * i + 2]; d.crop_w = windows_[4 * i + 3];
      d.mirror = mirror_[i] != 0;
      d.normalize = !mean_.empty();
      FillNorm(d.mean, d.inv_std);
      for (int c = 0; c < 4; c++)
        d.fill[c] = fill_.empty() ? 0.0f : fill_.size() == 1 ? fill_[0] : (c < (int)fill_.size() ? fill_[c] : 0.0f);
      const TensorShape &os = out.shape(i);
      d.out_channels = (int32_t)(chw_ ? os[0] : os[2]);
      d.out_dtype = ToKernelDType(out_type_);
      d.out_layout = chw_ ? DALIAMD_LAYOUT_CHW : DALIAMD_LAYOUT_HWC;
      d.out = out.raw(i);
    }
    int nwg = 0;
    KCHECK(daliamdCmnSetup(descs_.data(), n, &nwg));
    if (ws.backend == OpType::CPU) {  // crop_mirror_normalize.cc:116-144: one task per sample
      for (int i = 0; i < n; i++)
        ws.GetThreadPool().AddWork([this, i](int) {
          if (daliamdCmnRunHost(&descs_[i]) != 0) DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, (int64_t)descs_[i].crop_h * descs_[i].crop_w);
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_cmn");
      return;
    }
    auto *dev = static_cast<const daliamdCmnDesc *>(uploader_.Upload(descs_.data(), descs_.size() * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdCmnRun(ws.stream, dev, n, nwg));
    NoteLaunch(ws, "cmn");
  }

 private:
  void FillNorm(float *mean, float *inv) const {
    for (int c = 0; c < 4; c++) {
      mean[c] = mean_.empty() ? 0.0f : mean_.size() == 1 ? mean_[0] : (c < (int)mean_.size() ? mean_[c] : 0.0f);
      inv[c] = inv_std_.empty() ? 1.0f : inv_std_.size() == 1 ? inv_std_[0] : (c < (int)inv_std_.size() ? inv_std_[c] : 0.0f);
    }
  }
  DALIDataType out_type_;
  bool chw_, pad_output_, pad_oob_ = false, trim_ = false, round_ = true, has_crop_ = false, fused_input_ = false;
  float scale_, shift_;
  std::vector<float> mean_, inv_std_, fill_;
  std::vector<int> mirror_, windows_;
  std::vector<daliamdCmnDesc> descs_;
  std::vector<daliamdResampleArgs> rargs_;
  std::vector<daliamdResampleDesc> rdescs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(CropMirrorNormalize, CropMirrorNormalizeGpu, GPU);
DALI_REGISTER_OPERATOR(CropMirrorNormalize, CropMirrorNormalizeGpu, CPU);  // same class: host kernels when run on the CPU

bool TryEnableRoiDecodeFusion(OperatorBase *decoder, OperatorBase *consumer) {
  auto *dec = dynamic_cast<ImageDecoderMixed *>(decoder);
  auto *rrc = dynamic_cast<RandomResizedCropGpu *>(consumer);
  // (the crop decoders are ImageDecoderMixed too: they have their own windows)
  if (!dec || !rrc || !dec->CanTakeRoiSource() || !rrc->CanTakeCroppedInput() || dynamic_cast<ImageDecoderRandomCropMixed *>(decoder) ||
      dynamic_cast<ImageDecoderCropMixed *>(decoder) || dynamic_cast<ImageDecoderSliceMixed *>(decoder))
    return false;
  rrc->ExpectCroppedInput();
  dec->SetRoiSource([rrc](int64_t it, int n, const int32_t *hw, int32_t *rois) { rrc->DrawWindows(it, n, hw, rois); },
                    [rrc](int64_t it) { rrc->UndoDraw(it); });
  return true;
}

void TryEnableFusion(OperatorBase *producer, OperatorBase *consumer) {
  auto *rrc = dynamic_cast<RandomResizedCropGpu *>(producer);
  auto *rsz = dynamic_cast<ResizeGpu *>(producer);
  auto *cmn = dynamic_cast<CropMirrorNormalizeGpu *>(consumer);
  if ((!rrc && !rsz) || !cmn || !cmn->CanFuse()) return;
  if (rrc) rrc->EnableFusion();
  if (rsz) rsz->EnableFusion();  // defers only the batches whose outputs all have the same size
  cmn->ExpectFusedInput();
}

}  // namespace daliamd_host
