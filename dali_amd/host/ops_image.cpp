// Image operators of the hot path, host side.  All arithmetic runs in libdali_amd_kernels.so; these
// classes do what the reference's operator shells do: argument handling, shape inference, random
// parameter generation, and launching the batched kernels.
//   decoders.image (mixed)   dali/operators/imgcodec/image_decoder.h:131-933, decoder_schema.cc:22-151
//   RandomResizedCrop        dali/operators/image/resize/random_resized_crop.{h,cc,cu}
//   CropMirrorNormalize      dali/operators/image/crop/crop_mirror_normalize.{h,cc}, new_crop_mirror_normalize.cu
//   CropAttr                 dali/operators/image/crop/crop_attr.cc:21-240
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "dali_amd_host.h"
#include "ops.h"
#include "pipeline.h"

namespace daliamd_host {

static constexpr int kImagePitchAlign = 16;  // row pitch of device images handed between our own operators

// DALI enum values (include/dali/core/common.h:144-165)
enum { DALI_INTERP_NN = 0, DALI_INTERP_LINEAR = 1, DALI_INTERP_CUBIC = 2, DALI_INTERP_LANCZOS3 = 3,
       DALI_INTERP_TRIANGULAR = 4, DALI_INTERP_GAUSSIAN = 5 };
enum { DALI_RGB = 0, DALI_BGR = 1, DALI_GRAY = 2, DALI_YCbCr = 3, DALI_ANY_DATA = 4 };

static int ToKernelInterp(int64_t dali_interp) {
  switch (dali_interp) {
    case DALI_INTERP_LINEAR: return DALIAMD_INTERP_LINEAR;
    case DALI_INTERP_TRIANGULAR: return DALIAMD_INTERP_TRIANGULAR;
    default:
      DALI_FAIL("Interpolation type ", dali_interp, " is not supported by the MI355X resampling kernel yet "
                "(supported: INTERP_LINEAR, INTERP_TRIANGULAR)");
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
    .AddOptionalArg("output_type", "The color space of the output image (RGB or GRAY-as-RGB inputs only).",
                    ArgValue::Int(DALI_RGB))
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
    .AddOptionalArg("cache_size", "Ignored (no decoder cache).", ArgValue::Int(0))
    .AddOptionalArg("cache_type", "Ignored.", ArgValue::Str(""))
    .AddOptionalArg("cache_threshold", "Ignored.", ArgValue::Int(0))
    .AddOptionalArg("cache_debug", "Ignored.", ArgValue::Bool(false))
    .AddOptionalArg("cache_batch_copy", "Ignored.", ArgValue::Bool(true))
    .InputLayout(0, {""});
DALI_SCHEMA(ImageDecoder).DocStr("Legacy alias of decoders.image").NumInput(1).NumOutput(1).AddParent("decoders__Image");
DALI_SCHEMA(experimental__decoders__Image).DocStr("Alias of decoders.image").NumInput(1).NumOutput(1).AddParent("decoders__Image");

class ImageDecoderMixed : public OperatorBase {
 public:
  explicit ImageDecoderMixed(const OpSpec &spec) : OperatorBase(spec) {
    int64_t ot = spec.GetInt("output_type");
    DALI_ENFORCE(ot == DALI_RGB || ot == DALI_ANY_DATA, "decoders.image: only output_type=RGB is supported, got ", ot);
    DALI_ENFORCE(spec.GetInt("dtype") == DALI_UINT8, "decoders.image: only dtype=UINT8 is supported");
    adjust_orientation_ = spec.GetBool("adjust_orientation");
    // Entropy decoding runs on the GPU for every stream the kernel supports.  An EXPLICIT hybrid_huffman_threshold
    // keeps the reference's meaning: streams with fewer pixels than that are Huffman-decoded on the host.
    if (spec.Args().count("hybrid_huffman_threshold")) huffman_threshold_ = spec.GetInt("hybrid_huffman_threshold");
    if (const char *env = getenv("DALI_AMD_HOST_HUFFMAN")) host_huffman_only_ = atoi(env) != 0;
    ring_ = (int)spec.GetInt("gpu_prefetch_queue_depth") + 1;
    for (int i = 0; i < ring_; i++) {
      staging_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
      coef_dev_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      planes_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      ecs_stage_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
      ecs_dev_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      scratch_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      status_dev_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      status_host_.emplace_back(std::make_unique<Buffer>(StorageDevice::CPU));
    }
  }
  int OutputPitchAlign(int) const override { return kImagePitchAlign; }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }

  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    const int n = in.num_samples();
    DALI_ENFORCE(in.type() == DALI_UINT8, "decoders.image expects encoded streams as 1-D uint8 tensors");
    infos_.resize(n);
    scans_.resize(n);
    auto src = [&](int i) { return i < (int)in.source_info.size() && !in.source_info[i].empty() ? in.source_info[i]
                                                                                               : make_string("sample #", i); };
    // ---- header parse + scan analysis (thread pool: the analysis walks the stream once to find its end) ----
    for (int i = 0; i < n; i++) {
      ws.GetThreadPool().AddWork([&, i](int) {
        const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
        if (daliamdJpegParse(data, in.nbytes(i), &infos_[i]) != 0)
          DALI_FAIL("Failed to parse ", src(i), ": ", daliamdHostGetLastErrorMessage());
        DALI_ENFORCE(infos_[i].num_components == 1 || infos_[i].num_components == 3, "Failed to decode ", src(i),
                     ": JPEG with ", infos_[i].num_components, " components (CMYK/YCCK) is not supported");
        scans_[i].eligible = 0;
        if (!host_huffman_only_ && (int64_t)infos_[i].width * infos_[i].height >= huffman_threshold_ &&
            daliamdJpegAnalyzeScan(data, in.nbytes(i), &infos_[i], &scans_[i]) != 0)
          scans_[i].eligible = 0;  // the host decoder will produce the diagnosis
      }, (int64_t)in.nbytes(i));
    }
    ws.GetThreadPool().RunAll();
    // ---- layout ----
    std::vector<TensorShape> shapes(n);
    coef_off_.assign(n * 3, 0);
    ecs_off_.assign(n, 0);
    scratch_off_.assign(n, 0);
    gpu_samples_.clear();
    int64_t elems = 0;
    size_t ecs_bytes = 0, scratch_bytes = 0;
    int ncomp_total = 0;
    for (int i = 0; i < n; i++) {
      const auto &inf = infos_[i];
      bool swap = adjust_orientation_ && inf.orientation >= 5;
      shapes[i] = {swap ? inf.width : inf.height, swap ? inf.height : inf.width, 3};
      for (int c = 0; c < inf.num_components; c++) {
        coef_off_[i * 3 + c] = elems;
        elems += inf.coef_elems[c];
        ncomp_total++;
      }
      if (scans_[i].eligible) {
        DALI_ENFORCE(scans_[i].ecs_length < (int64_t)1 << 30, "Failed to decode ", src(i), ": entropy-coded segment too long");
        size_t need = 0;
        KCHECK(daliamdJpegHuffmanScratchBytes((int)scans_[i].ecs_length, &need));
        ecs_off_[i] = ecs_bytes;
        scratch_off_[i] = scratch_bytes;
        ecs_bytes += ((size_t)scans_[i].ecs_length + 15) & ~(size_t)15;
        scratch_bytes += need;
        gpu_samples_.push_back(i);
      }
    }
    const int ngpu = (int)gpu_samples_.size();
    const int slot = (int)(ws.iteration % ring_);
    Buffer &stage = *staging_[slot], &cdev = *coef_dev_[slot], &planes = *planes_[slot];
    Buffer &ecs_stage = *ecs_stage_[slot], &ecs_dev = *ecs_dev_[slot], &scratch = *scratch_[slot];
    Buffer &status_dev = *status_dev_[slot], &status_host = *status_host_[slot];
    if (ngpu < n) stage.Reserve((size_t)elems * 2 + 256);
    cdev.Reserve((size_t)elems * 2 + 256);
    planes.Reserve((size_t)elems + 256);
    ecs_stage.Reserve(ecs_bytes + 256);
    ecs_dev.Reserve(ecs_bytes + 256);
    scratch.Reserve(scratch_bytes + 256);
    status_dev.Reserve(sizeof(int32_t) * (size_t)std::max(n, 1));
    status_host.Reserve(sizeof(int32_t) * (size_t)std::max(n, 1));
    out.Resize(shapes, DALI_UINT8, kImagePitchAlign);
    out.SetLayout("HWC");
    out.source_info = in.source_info;
    quant_.assign((size_t)n * 3 * 64, 0);
    // ---- thread pool: gather the entropy-coded segments (GPU path) / entropy decode (host path) ----
    int16_t *coef_host = static_cast<int16_t *>(stage.data());
    for (int i = 0; i < n; i++) {
      const bool gpu = scans_[i].eligible != 0;
      ws.GetThreadPool().AddWork([&, i, gpu](int) {
        const uint8_t *data = static_cast<const uint8_t *>(in.raw(i));
        if (gpu) {
          memcpy(static_cast<uint8_t *>(ecs_stage.data()) + ecs_off_[i], data + scans_[i].ecs_offset,
                 (size_t)scans_[i].ecs_length);
          for (int c = 0; c < infos_[i].num_components; c++) memcpy(&quant_[(size_t)i * 192 + c * 64], scans_[i].quant[c], 128);
          return;
        }
        int16_t *ptrs[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int c = 0; c < infos_[i].num_components; c++) ptrs[c] = coef_host + coef_off_[i * 3 + c];
        if (daliamdJpegDecodeCoefficients(data, in.nbytes(i), &infos_[i], ptrs, &quant_[(size_t)i * 192]) != 0)
          DALI_FAIL("Failed to decode ", src(i), ": ", daliamdHostGetLastErrorMessage());
      }, gpu ? (int64_t)in.nbytes(i) / 16 : (int64_t)in.nbytes(i));
    }
    ws.GetThreadPool().RunAll();
    if (n == 0) return;
    // ---- entropy decoding on the device ----
    int16_t *coef = static_cast<int16_t *>(cdev.data());
    if (ngpu) {
      KCHECK(daliamdMemcpyH2DAsync(ecs_dev.data(), ecs_stage.data(), ecs_bytes, ws.stream));
      KCHECK(daliamdMemsetAsync(cdev.data(), 0, (size_t)elems * 2, ws.stream));
      KCHECK(daliamdMemsetAsync(status_dev.data(), 0, sizeof(int32_t) * (size_t)ngpu, ws.stream));
      huff_.assign(ngpu, daliamdJpegHuffDesc{});
      for (int j = 0; j < ngpu; j++) {
        const int i = gpu_samples_[j];
        const auto &inf = infos_[i];
        const auto &sc = scans_[i];
        auto &d = huff_[j];
        d.ecs = static_cast<const uint8_t *>(ecs_dev.data()) + ecs_off_[i];
        d.scratch = static_cast<uint8_t *>(scratch.data()) + scratch_off_[i];
        d.status = static_cast<int32_t *>(status_dev.data()) + j;
        d.ecs_len = (int32_t)sc.ecs_length;
        d.blocks_per_mcu = sc.blocks_per_mcu;
        d.mcus_x = sc.mcus_x;
        d.total_blocks = sc.mcus_x * sc.mcus_y * sc.blocks_per_mcu;
        for (int c = 0; c < inf.num_components; c++) {
          d.coef[c] = coef + coef_off_[i * 3 + c];
          d.blocks_x[c] = inf.blocks_x[c];
          d.h_samp[c] = inf.h_samp[c];
          d.v_samp[c] = inf.v_samp[c];
          d.dc_sel[c] = sc.dc_sel[c];
          d.ac_sel[c] = sc.ac_sel[c];
        }
        memcpy(d.comp_of_block, sc.comp_of_block, 10);
        memcpy(d.h_of_block, sc.h_of_block, 10);
        memcpy(d.v_of_block, sc.v_of_block, 10);
        for (int t = 0; t < 2; t++) {
          memcpy(d.bits[t], sc.dc_bits[t], 16);
          memcpy(d.bits[2 + t], sc.ac_bits[t], 16);
          memcpy(d.vals[t], sc.dc_vals[t], 256);
          memcpy(d.vals[2 + t], sc.ac_vals[t], 256);
        }
      }
      int ntiles = 0, nsegs = 0;
      KCHECK(daliamdJpegHuffmanSetup(huff_.data(), ngpu, &ntiles, &nsegs));
      auto *huff_dev = static_cast<const daliamdJpegHuffDesc *>(
          up_huff_.Upload(huff_.data(), huff_.size() * sizeof(huff_[0]), ws.stream));
      KCHECK(daliamdJpegHuffmanRun(ws.stream, huff_dev, ngpu, ntiles, nsegs));
      KCHECK(daliamdMemcpyD2HAsync(status_host.data(), status_dev.data(), sizeof(int32_t) * (size_t)ngpu, ws.stream));
      NoteLaunch(ws, "jpeg_huffman");
      // the status words are valid once the iteration has finished: checked when its outputs are handed over
      std::vector<int> samples = gpu_samples_;
      std::vector<std::string> names(ngpu);
      for (int j = 0; j < ngpu; j++) names[j] = src(samples[j]);
      const int32_t *st = static_cast<const int32_t *>(status_host.data());
      ws.AddCompletionCheck([st, names] {
        for (size_t j = 0; j < names.size(); j++)
          if (st[j] != 0)
            DALI_FAIL("Failed to decode ", names[j], ": corrupt JPEG data: the entropy-coded segment ends before the "
                      "last MCU (GPU Huffman status ", st[j], ")");
      });
    }
    // host-decoded streams (progressive, restart markers, multi-scan, below the threshold): H2D of their coefficients
    for (int i = 0; i < n; i++) {
      if (scans_[i].eligible) continue;
      int64_t first = coef_off_[i * 3], count = 0;
      for (int c = 0; c < infos_[i].num_components; c++) count += infos_[i].coef_elems[c];
      KCHECK(daliamdMemcpyH2DAsync(coef + first, coef_host + first, (size_t)count * 2, ws.stream));
    }
    // ---- dequantisation + IDCT, upsampling + colour conversion ----
    idct_.assign(ncomp_total, daliamdJpegIdctDesc{});
    color_.assign(n, daliamdJpegColorDesc{});
    int k = 0;
    for (int i = 0; i < n; i++) {
      const auto &inf = infos_[i];
      auto &cd = color_[i];
      for (int c = 0; c < inf.num_components; c++) {
        auto &d = idct_[k++];
        d.coef = coef + coef_off_[i * 3 + c];
        d.plane = static_cast<uint8_t *>(planes.data()) + coef_off_[i * 3 + c];
        d.blocks_x = inf.blocks_x[c];
        d.nblocks = inf.blocks_x[c] * inf.blocks_y[c];
        d.pitch = inf.blocks_x[c] * 8;
        memcpy(d.quant, &quant_[(size_t)i * 192 + c * 64], 128);
        cd.plane[c] = d.plane;
        cd.pitch[c] = d.pitch;
        cd.h_samp[c] = inf.h_samp[c]; cd.v_samp[c] = inf.v_samp[c];
        cd.down_w[c] = inf.down_w[c]; cd.down_h[c] = inf.down_h[c];
      }
      for (int c = inf.num_components; c < 3; c++) { cd.h_samp[c] = cd.v_samp[c] = 1; }
      cd.width = inf.width; cd.height = inf.height; cd.color = inf.color;
      cd.out = static_cast<uint8_t *>(out.raw(i));
      cd.out_pitch = (int32_t)out.row_pitch(i);
      cd.orientation = adjust_orientation_ ? inf.orientation : 1;
    }
    int wg_idct = 0, wg_color = 0;
    KCHECK(daliamdJpegIdctSetup(idct_.data(), ncomp_total, &wg_idct));
    KCHECK(daliamdJpegColorSetup(color_.data(), n, &wg_color));
    auto *idct_dev = static_cast<const daliamdJpegIdctDesc *>(
        up_idct_.Upload(idct_.data(), idct_.size() * sizeof(idct_[0]), ws.stream));
    auto *color_dev = static_cast<const daliamdJpegColorDesc *>(
        up_color_.Upload(color_.data(), color_.size() * sizeof(color_[0]), ws.stream));
    KCHECK(daliamdJpegIdctRun(ws.stream, idct_dev, ncomp_total, wg_idct));
    KCHECK(daliamdJpegColorRun(ws.stream, color_dev, n, wg_color));
    NoteLaunch(ws, "jpeg_idct");
    NoteLaunch(ws, "jpeg_color");
  }

 private:
  bool adjust_orientation_;
  bool host_huffman_only_ = false;
  int64_t huffman_threshold_ = 0;
  int ring_;
  std::vector<std::unique_ptr<Buffer>> staging_, coef_dev_, planes_, ecs_stage_, ecs_dev_, scratch_, status_dev_, status_host_;
  std::vector<daliamdJpegInfo> infos_;
  std::vector<daliamdJpegScan> scans_;
  std::vector<int> gpu_samples_;
  std::vector<int64_t> coef_off_;
  std::vector<size_t> ecs_off_, scratch_off_;
  std::vector<uint16_t> quant_;
  std::vector<daliamdJpegHuffDesc> huff_;
  std::vector<daliamdJpegIdctDesc> idct_;
  std::vector<daliamdJpegColorDesc> color_;
  DescUploader up_huff_, up_idct_, up_color_;
};
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
    .AddOptionalTypeArg("dtype", "Output data type. Must be same as input type (uint8).", ArgType::INT)
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
            "Expects a three-dimensional input with samples in height, width, channels (HWC) layout.")
    .NumInput(1)
    .NumOutput(1)
    .AddArg("size", "Size of the resized image.", ArgType::INT_VEC)
    .AddParent("RandomCropAttr")
    .AddParent("ResamplingFilterAttr")
    .AllowSequences()
    .InputLayout(0, {"HWC"});

// per-operator filter arguments -> kernel enums (ResamplingFilterAttr::PrepareFilterParams, resampling_attr.cc:76-121)
struct FilterArgs {
  int min_filter = DALIAMD_INTERP_LINEAR, mag_filter = DALIAMD_INTERP_LINEAR, antialias = 1;
  explicit FilterArgs(const OpSpec &spec) {
    DALI_ENFORCE(!spec.HasTensorArgument("interp_type") && !spec.HasTensorArgument("min_filter") &&
                 !spec.HasTensorArgument("mag_filter"), "Per-sample interpolation types are not supported yet");
    antialias = spec.GetBool("antialias");
    bool has_interp = spec.Args().count("interp_type"), has_min = spec.Args().count("min_filter"),
         has_mag = spec.Args().count("mag_filter");
    if (has_min) min_filter = ToKernelInterp(spec.GetInt("min_filter"));
    else if (has_interp) min_filter = ToKernelInterp(spec.GetInt("interp_type"));
    if (has_mag) mag_filter = ToKernelInterp(spec.GetInt("mag_filter"));
    else if (has_interp) mag_filter = ToKernelInterp(spec.GetInt("interp_type"));
    if (const ArgValue *d = spec.TryArg("dtype"))
      DALI_ENFORCE(d->i == DALI_UINT8, "Resampling output dtype must be the input type (uint8)");
  }
};

static void LaunchResample(Workspace &ws, DescUploader &up, std::vector<daliamdResampleArgs> &args,
                           std::vector<daliamdResampleDesc> &descs, const char *what) {
  int n = (int)args.size();
  if (!n) return;
  descs.resize(n);
  int nwg = 0, lds = 0;
  KCHECK(daliamdResampleSetup(args.data(), n, descs.data(), &nwg, &lds));
  auto *dev = static_cast<const daliamdResampleDesc *>(up.Upload(descs.data(), descs.size() * sizeof(descs[0]), ws.stream));
  KCHECK(daliamdResampleRun(ws.stream, dev, n, nwg, lds));
  NoteLaunch(ws, what);
}

static void FillSourceArgs(daliamdResampleArgs &a, const TensorList &in, int i) {
  const TensorShape &s = in.shape(i);
  DALI_ENFORCE(s.size() == 3, "Expected a three-dimensional HWC input, got ", s.size(), " dimensions");
  a.in = static_cast<const uint8_t *>(in.raw(i));
  a.in_h = (int32_t)s[0]; a.in_w = (int32_t)s[1]; a.channels = (int32_t)s[2];
  a.in_pitch = (int32_t)(in.row_pitch(i) ? in.row_pitch(i) : s[1] * s[2]);
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

  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    int n = in.num_samples();
    DALI_ENFORCE(in.type() == DALI_UINT8, "RandomResizedCrop (gpu): only uint8 input is supported, got ", TypeName(in.type()));
    shapes_hw_.resize(2 * n); anchors_.resize(2 * n); crops_.resize(2 * n);
    args_.assign(n, daliamdResampleArgs{});
    int ch = 3;
    for (int i = 0; i < n; i++) {
      FillSourceArgs(args_[i], in, i);
      shapes_hw_[2 * i] = args_[i].in_h; shapes_hw_[2 * i + 1] = args_[i].in_w;
      ch = args_[i].channels;
    }
    if (daliamdRandomCropBatch(&master_, n, shapes_hw_.data(), ar_lo_, ar_hi_, area_lo_, area_hi_, num_attempts_,
                               anchors_.data(), crops_.data()) != 0)
      DALI_FAIL(daliamdHostGetLastErrorMessage());
    for (int i = 0; i < n; i++) {
      auto &a = args_[i];
      a.use_roi = 1;
      a.roi_y0 = (float)anchors_[2 * i]; a.roi_x0 = (float)anchors_[2 * i + 1];
      a.roi_y1 = (float)(anchors_[2 * i] + crops_[2 * i]); a.roi_x1 = (float)(anchors_[2 * i + 1] + crops_[2 * i + 1]);
      a.out_h = out_h_; a.out_w = out_w_;
      a.min_filter = filters_.min_filter; a.mag_filter = filters_.mag_filter; a.antialias = filters_.antialias;
      a.out_dtype = DALIAMD_UINT8; a.out_layout = DALIAMD_LAYOUT_HWC;
    }
    n_ = n; ch_ = ch;
    if (fused_) return false;  // no buffer: the consumer launches the fused kernel
    desc[0].type = DALI_UINT8;
    desc[0].shape.assign(n, TensorShape{out_h_, out_w_, ch});
    return true;
  }

  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    if (fused_) {
      auto d = std::make_shared<DeferredResample>();
      d->source = ws.inputs[0];
      d->args = args_;
      d->out_h = out_h_; d->out_w = out_w_; d->channels = ch_;
      out.Resize({}, DALI_UINT8);
      out.deferred = d;
      out.SetLayout("HWC");
    } else {
      out.SetLayout("HWC");
      for (int i = 0; i < n_; i++) {
        // the kernel writes dense HWC rows; request that from the executor by a dense pitch
        args_[i].out = out.raw(i);
      }
      DALI_ENFORCE(out.is_dense() || n_ == 0, "internal: resample output must be dense");
      LaunchResample(ws, uploader_, args_, descs_, "resample");
    }
    daliamdPhiloxAdvanceSequence(&master_, (uint64_t)n_);  // OperatorWithRng::Advance(batch)
  }

  std::string SaveState() const override {
    char buf[96];
    daliamdPhiloxStateToString(&master_, buf, sizeof(buf));
    return buf;
  }
  void RestoreState(const std::string &s) override {
    DALI_ENFORCE(daliamdPhiloxStateFromString(&master_, s.c_str()) == 0, daliamdHostGetLastErrorMessage());
  }

 private:
  FilterArgs filters_;
  int out_h_, out_w_, num_attempts_, n_ = 0, ch_ = 3;
  float ar_lo_, ar_hi_, area_lo_, area_hi_;
  daliamdPhiloxState master_;
  bool fused_ = false;
  std::vector<int32_t> shapes_hw_, anchors_, crops_;
  std::vector<daliamdResampleArgs> args_;
  std::vector<daliamdResampleDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(RandomResizedCrop, RandomResizedCropGpu, GPU);

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
    DALI_ENFORCE(!fused_input_ || def, "internal: the producer did not hand over its resampling arguments");
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
    auto *dev = static_cast<const daliamdCmnDesc *>(uploader_.Upload(descs_.data(), descs_.size() * sizeof(descs_[0]), ws.stream));
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

void TryEnableFusion(OperatorBase *producer, OperatorBase *consumer) {
  auto *rrc = dynamic_cast<RandomResizedCropGpu *>(producer);
  auto *cmn = dynamic_cast<CropMirrorNormalizeGpu *>(consumer);
  if (!rrc || !cmn || !cmn->CanFuse()) return;
  rrc->EnableFusion();
  cmn->ExpectFusedInput();
}

}  // namespace daliamd_host
