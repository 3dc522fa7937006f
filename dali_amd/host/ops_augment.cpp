// Heavy-augmentation operators (BASELINE.json configs[2]), host side; arithmetic in libdali_amd_kernels.so.
//   WarpAffine    dali/operators/image/remap/warp_affine.cc:19-57, warp_affine_params.h:44-78, warp_attr.cc:20-43
//   GaussianBlur  dali/operators/image/convolution/gaussian_blur.cc:33-73, gaussian_blur_params.h:27-83
//   ColorTwist    dali/operators/image/color/color_twist.{h,cc}
//   Erase         dali/operators/generic/erase/erase.cc:29-160, erase_utils.h:52-165
#include <cmath>

#include "ops.h"
#include "pipeline.h"
#include "dali_amd_host.h"

namespace daliamd_host {

enum { DALI_INTERP_NN = 0, DALI_INTERP_LINEAR = 1 };

static void CheckU8Hwc(const TensorList &in, const char *op) {
  DALI_ENFORCE(in.type() == DALI_UINT8, op, " (gpu): only uint8 input is supported, got ", TypeName(in.type()));
  for (int i = 0; i < in.num_samples(); i++)
    DALI_ENFORCE(in.shape(i).size() == 3, op, " expects three-dimensional HWC input");
}
static int InPitch(const TensorList &in, int i) {
  return (int)(in.row_pitch(i) ? in.row_pitch(i) : in.shape(i)[1] * in.shape(i)[2]);
}
static void CheckOutDtype(const OpSpec &spec, const char *op) {
  if (const ArgValue *d = spec.TryArg("dtype"))
    DALI_ENFORCE(d->i == DALI_UINT8, op, ": output dtype must equal the input type (uint8) in this build");
}

// =============================================================================================
DALI_SCHEMA(WarpAffine)
    .DocStr("Applies an affine transformation to the images.\n\nThe matrix maps destination to source coordinates "
            "(``inverse_map=True``, the default) in (x, y) order with pixel centres at half-integers.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalTypeArg("matrix", "Transform matrix: 6 values, rows of a 2x3 matrix.", ArgType::FLOAT_VEC, true)
    .AddOptionalArg("inverse_map", "Set to False if the given transform is a destination to source mapping... "
                    "(True: matrix maps destination to source).", ArgValue::Bool(true))
    .AddOptionalTypeArg("size", "Output size, in pixels (H, W).", ArgType::FLOAT_VEC, true)
    .AddOptionalTypeArg("fill_value", "Value used to fill areas that are outside the source image. If not specified, "
                        "the source coordinates are clamped (border pixels are repeated).", ArgType::FLOAT)
    .AddOptionalArg("interp_type", "Type of interpolation used (INTERP_NN or INTERP_LINEAR).", ArgValue::Int(DALI_INTERP_LINEAR))
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});

// Shared by WarpAffine and Rotate: the operators differ in where the destination->source matrix and the output size
// of a sample come from (WarpParamProvider subclasses in the reference: warp_affine_params.h, rotate_params.h).
class WarpGpuBase : public OperatorBase {
 public:
  WarpGpuBase(const OpSpec &spec, const char *name) : OperatorBase(spec), name_(name) {
    CheckOutDtype(spec, name);
    int64_t it = spec.GetInt("interp_type");
    DALI_ENFORCE(it == DALI_INTERP_NN || it == DALI_INTERP_LINEAR, name, " supports INTERP_NN and INTERP_LINEAR, got ", it);
    interp_ = it == DALI_INTERP_NN ? DALIAMD_INTERP_NN : DALIAMD_INTERP_LINEAR;
    has_fill_ = spec.TryArg("fill_value") != nullptr;
    if (has_fill_) fill_ = (float)spec.GetFloat("fill_value");
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    CheckU8Hwc(in, name_);
    int n = in.num_samples();
    std::vector<std::vector<float>> sizes;
    if (spec_.ArgumentDefined("size")) sizes = GetPerSampleFloatVec(spec_, ws, "size", n);
    descs_.assign(n, daliamdWarpAffineDesc{});
    desc[0].type = DALI_UINT8;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      const TensorShape &s = in.shape(i);
      d.in = static_cast<const uint8_t *>(in.raw(i));
      d.in_h = (int)s[0]; d.in_w = (int)s[1]; d.channels = (int)s[2]; d.in_pitch = InPitch(in, i);
      d.out_h = d.in_h; d.out_w = d.in_w;
      if (!sizes.empty()) {
        DALI_ENFORCE(sizes[i].size() == 2, "`size` must hold (H, W)");
        d.out_h = (int)sizes[i][0]; d.out_w = (int)sizes[i][1];
        DALI_ENFORCE(d.out_h > 0 && d.out_w > 0, "`size` must be positive");
      }
      d.interp = interp_;
      d.border_clamp = !has_fill_;
      for (int c = 0; c < 4; c++) d.fill[c] = fill_;
    }
    SampleParams(ws, n, !sizes.empty());
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      d.out_pitch = d.out_w * d.channels;
      desc[0].shape[i] = {d.out_h, d.out_w, d.channels};
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout("HWC");
    int n = (int)descs_.size();
    if (!n) return;
    for (int i = 0; i < n; i++) descs_[i].out = static_cast<uint8_t *>(out.raw(i));
    if (ws.backend == OpType::CPU) {  // one thread-pool task per sample on the host kernel
      for (int i = 0; i < n; i++)
        ws.GetThreadPool().AddWork([this, i](int) {
          if (daliamdWarpAffineHost(&descs_[i]) != 0) DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, (int64_t)descs_[i].out_h * descs_[i].out_w);
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_warp_affine");
      return;
    }
    int nwg = 0;
    KCHECK(daliamdWarpAffineSetup(descs_.data(), n, &nwg));
    auto *dev = static_cast<const daliamdWarpAffineDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdWarpAffineRun(ws.stream, dev, n, nwg));
    NoteLaunch(ws, "warp_affine");
  }

 protected:
  // fills descs_[i].matrix (destination -> source) and, unless `explicit_size`, may set descs_[i].out_h / out_w
  virtual void SampleParams(const Workspace &ws, int n, bool explicit_size) = 0;
  std::vector<daliamdWarpAffineDesc> descs_;

 private:
  const char *name_;
  int interp_;
  bool has_fill_;
  float fill_ = 0;
  DescUploader uploader_;
};

class WarpAffineGpu : public WarpGpuBase {
 public:
  explicit WarpAffineGpu(const OpSpec &spec) : WarpGpuBase(spec, "WarpAffine") {
    invert_ = !spec.GetBool("inverse_map");
    DALI_ENFORCE(spec.ArgumentDefined("matrix"), "`matrix` argument must be provided");
  }

 protected:
  void SampleParams(const Workspace &ws, int n, bool) override {
    auto mats = GetPerSampleFloatVec(spec_, ws, "matrix", n);
    for (int i = 0; i < n; i++) {
      DALI_ENFORCE(mats[i].size() == 6, "`matrix` parameter must have 6 elements, got ", mats[i].size());
      auto &d = descs_[i];
      const float *m = mats[i].data();
      if (invert_) {
        // affine_mat_inv (include/dali/core/geom/transform.h:166-174; 2x2 inverse mat.h:610-622)
        float det = m[0] * m[4] - m[1] * m[3];
        DALI_ENFORCE(det != 0, "Cannot calculate the inverse of a singular matrix.");
        float i00 = m[4] / det, i01 = -m[1] / det, i10 = -m[3] / det, i11 = m[0] / det;
        float t0 = (-i00) * m[2]; t0 += (-i01) * m[5];
        float t1 = (-i10) * m[2]; t1 += (-i11) * m[5];
        float inv[6] = {i00, i01, t0, i10, i11, t1};
        memcpy(d.matrix, inv, sizeof(inv));
      } else {
        memcpy(d.matrix, m, 6 * sizeof(float));
      }
    }
  }

 private:
  bool invert_;
};
DALI_REGISTER_OPERATOR(WarpAffine, WarpAffineGpu, GPU);
DALI_REGISTER_OPERATOR(WarpAffine, WarpAffineGpu, CPU);  // same class: the host kernel when run on the CPU

// ---- Rotate: the warp kernel with the matrix and canvas of rotate_params.h (dali/operators/image/remap/rotate.cc:19-44)
DALI_SCHEMA(Rotate)
    .DocStr("Rotates the images by the specified angle.\n\nThe rotation is counter-clockwise, assuming the top-left corner "
            "is at ``(0,0)``.  Unless ``size`` or ``keep_size`` is given, the canvas is adjusted to accommodate the rotated "
            "image with the least padding possible (its parity follows the input's to reduce blur).")
    .NumInput(1)
    .NumOutput(1)
    .AddArg("angle", "Angle, in degrees, by which the image is rotated.", ArgType::FLOAT, true)
    .AddOptionalArg("keep_size", "If True, original canvas size is kept.", ArgValue::Bool(false))
    .AddOptionalTypeArg("axis", "3D rotation only: not supported (two-dimensional HWC images).", ArgType::FLOAT_VEC, true)
    .AddOptionalTypeArg("size", "Output size, in pixels (H, W).", ArgType::FLOAT_VEC, true)
    .AddOptionalTypeArg("fill_value", "Value used to fill areas that are outside the source image. If not specified, "
                        "the source coordinates are clamped (border pixels are repeated).", ArgType::FLOAT)
    .AddOptionalArg("interp_type", "Type of interpolation used (INTERP_NN or INTERP_LINEAR).", ArgValue::Int(DALI_INTERP_LINEAR))
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});

class RotateGpu : public WarpGpuBase {
 public:
  explicit RotateGpu(const OpSpec &spec) : WarpGpuBase(spec, "Rotate"), keep_size_(spec.GetBool("keep_size")) {
    DALI_ENFORCE(!spec.ArgumentDefined("axis"), "Rotate: `axis` applies to volumetric input, which this build does not take");
  }

 protected:
  void SampleParams(const Workspace &ws, int n, bool explicit_size) override {
    auto angles = GetPerSampleFloat(spec_, ws, "angle", n);
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      const float a = -angles[i];                          // SetParams: positive = counter-clockwise with (0,0) top-left
      const float rad = a * (float)(M_PI / 180);           // deg2rad(float), math_util.h:153-157
      if (!explicit_size && !keep_size_) {
        // RotatedCanvasSize (rotate_params.h:33-52) + the parity correction of InferSize (:285) for one frame
        const double eps = 1e-2, abs_cos = std::abs(std::cos((double)rad)), abs_sin = std::abs(std::sin((double)rad));
        const int w = d.in_w, h = d.in_h;
        int w_out = (int)std::ceil(abs_cos * w + abs_sin * h - eps), h_out = (int)std::ceil(abs_cos * h + abs_sin * w - eps);
        const int par_w = abs_sin <= abs_cos ? w % 2 : h % 2, par_h = abs_sin <= abs_cos ? h % 2 : w % 2;
        w_out += (w_out % 2) ^ par_w;
        h_out += (h_out % 2) ^ par_h;
        d.out_w = w_out; d.out_h = h_out;
      }
      // M = translation(in_size / 2) * rotation2D(-a) * translation(-out_size / 2)   (rotate_params.h:225-226), float
      // products in the order of mat.h (s = a0 * b0; s += a1 * b1; s += a2 * b2)
      const float c = std::cos(-rad), sn = std::sin(-rad);
      const float tx = d.in_w * 0.5f, ty = d.in_h * 0.5f, ox = -(d.out_w * 0.5f), oy = -(d.out_h * 0.5f);
      float m02 = c * ox; m02 += (-sn) * oy; m02 += tx;
      float m12 = sn * ox; m12 += c * oy; m12 += ty;
      const float m[6] = {c, -sn, m02, sn, c, m12};
      memcpy(d.matrix, m, sizeof(m));
    }
  }

 private:
  bool keep_size_;
};
DALI_REGISTER_OPERATOR(Rotate, RotateGpu, GPU);
DALI_REGISTER_OPERATOR(Rotate, RotateGpu, CPU);

// =============================================================================================
DALI_SCHEMA(GaussianBlur)
    .DocStr("Applies a Gaussian Blur to the input.\n\nGaussian blur is calculated by applying a convolution with a Gaussian "
            "kernel, parameterized with ``window_size`` and ``sigma``. If only the sigma is specified, the window size is "
            "``2 * ceil(3 * sigma) + 1``; if only the window size is provided, "
            "``sigma = ((window_size - 1) / 2 - 1) * 0.3 + 0.8``. Values can be given per axis in (H, W) order.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("sigma", "Sigma value for the Gaussian Kernel.", ArgValue::FloatVec({0.0}), true)
    .AddOptionalArg("window_size", "The diameter of the kernel.", ArgValue::IntVec({0}), true)
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});

class GaussianBlurGpu : public OperatorBase {
 public:
  explicit GaussianBlurGpu(const OpSpec &spec) : OperatorBase(spec) { CheckOutDtype(spec, "GaussianBlur"); }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    CheckU8Hwc(in, "GaussianBlur");
    int n = in.num_samples();
    auto sig = GetPerSampleFloatVec(spec_, ws, "sigma", n);
    auto win = GetPerSampleFloatVec(spec_, ws, "window_size", n);
    descs_.assign(n, daliamdGaussianBlurDesc{});
    desc[0].type = DALI_UINT8;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      const TensorShape &s = in.shape(i);
      d.in = static_cast<const uint8_t *>(in.raw(i));
      d.h = (int)s[0]; d.w = (int)s[1]; d.channels = (int)s[2]; d.in_pitch = InPitch(in, i);
      d.out_pitch = d.w * d.channels;
      DALI_ENFORCE((sig[i].size() == 1 || sig[i].size() == 2) && (win[i].size() == 1 || win[i].size() == 2),
                   "`sigma` and `window_size` take one value or one value per axis (H, W)");
      // per-axis values are given in layout order (H, W); window_x is the W axis
      float sy = sig[i][0], sx = sig[i].back();
      int wy = (int)win[i][0], wx = (int)win[i].back();
      int dx = daliamdGaussianWindow(sx, wx, d.window_x, nullptr);
      DALI_ENFORCE(dx > 0, daliamdGetLastErrorMessage());
      int dy = daliamdGaussianWindow(sy, wy, d.window_y, nullptr);
      DALI_ENFORCE(dy > 0, daliamdGetLastErrorMessage());
      d.size_x = dx; d.size_y = dy;
      desc[0].shape[i] = s;
    }
    return !(fused_ && ws.backend != OpType::CPU);   // fused: no buffer, the pointwise operator behind launches for both
  }
  void EnableFusion() { fused_ = true; }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout("HWC");
    int n = (int)descs_.size();
    if (fused_ && ws.backend != OpType::CPU) {
      auto d = std::make_shared<DeferredBlur>();
      d->source = ws.inputs[0];
      d->descs = descs_;
      out.Resize({}, DALI_UINT8);
      out.deferred_blur = d;
      return;
    }
    if (!n) return;
    for (int i = 0; i < n; i++) descs_[i].out = static_cast<uint8_t *>(out.raw(i));
    if (ws.backend == OpType::CPU) {
      for (int i = 0; i < n; i++) {
        ws.GetThreadPool().AddWork([this, i](int) {
          if (daliamdGaussianBlurHost(&descs_[i]) != 0) DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, (int64_t)descs_[i].h * descs_[i].w);
      }
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_gaussian_blur");
      return;
    }
    int nwg = 0, lds = 0;
    KCHECK(daliamdGaussianBlurSetup(descs_.data(), n, &nwg, &lds));
    auto *dev = static_cast<const daliamdGaussianBlurDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdGaussianBlurRun(ws.stream, dev, n, nwg, lds));
    NoteLaunch(ws, "gaussian_blur");
  }

 private:
  bool fused_ = false;
  std::vector<daliamdGaussianBlurDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(GaussianBlur, GaussianBlurGpu, GPU);
DALI_REGISTER_OPERATOR(GaussianBlur, GaussianBlurGpu, CPU);

// =============================================================================================
// ColorTwist and Erase share the pointwise kernel
// =============================================================================================
static void FillPointwiseCommon(daliamdPointwiseDesc &d, const TensorList &in, int i) {
  const TensorShape &s = in.shape(i);
  d.in = static_cast<const uint8_t *>(in.raw(i));
  d.h = (int)s[0]; d.w = (int)s[1]; d.channels = (int)s[2]; d.in_pitch = InPitch(in, i);
  d.out_pitch = d.w * d.channels;
}
static void LaunchPointwise(Workspace &ws, DescUploader &up, std::vector<daliamdPointwiseDesc> &descs, const char *what,
                            const DeferredBlur *blur = nullptr, DescUploader *blur_up = nullptr) {
  TensorList &out = ws.Output(0);
  out.SetLayout("HWC");
  int n = (int)descs.size();
  if (!n) return;
  for (int i = 0; i < n; i++) descs[i].out = static_cast<uint8_t *>(out.raw(i));
  if (blur) {   // the blur in front runs here, with this operator's arithmetic in its write-out
    std::vector<daliamdGaussianBlurDesc> bd = blur->descs;
    for (int i = 0; i < n; i++) bd[i].out = descs[i].out;
    int nwg = 0, lds = 0;
    KCHECK(daliamdGaussianBlurSetup(bd.data(), n, &nwg, &lds));
    auto *pdev = static_cast<const daliamdPointwiseDesc *>(up.Upload(descs.data(), n * sizeof(descs[0]), ws.stream, ws.ring + 1));
    auto *bdev = static_cast<const daliamdGaussianBlurDesc *>(blur_up->Upload(bd.data(), n * sizeof(bd[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdGaussianBlurPointwiseRun(ws.stream, bdev, n, nwg, lds, pdev));
    NoteLaunch(ws, (std::string("gaussian_blur+") + what).c_str());
    return;
  }
  if (ws.backend == OpType::CPU) {
    for (int i = 0; i < n; i++)
      ws.GetThreadPool().AddWork([&descs, i](int) {
        if (daliamdPointwiseHost(&descs[i]) != 0) DALI_FAIL(daliamdHostGetLastErrorMessage());
      }, (int64_t)descs[i].h * descs[i].w);
    ws.GetThreadPool().RunAll();
    NoteLaunch(ws, (std::string("host_") + what).c_str());
    return;
  }
  int nwg = 0;
  KCHECK(daliamdPointwiseSetup(descs.data(), n, &nwg));
  auto *dev = static_cast<const daliamdPointwiseDesc *>(up.Upload(descs.data(), n * sizeof(descs[0]), ws.stream, ws.ring + 1));
  KCHECK(daliamdPointwiseRun(ws.stream, dev, n, nwg));
  NoteLaunch(ws, what);
}

DALI_SCHEMA(ColorTwist)
    .DocStr("Adjusts hue, saturation, value, brightness and contrast of the image.\n\n"
            "The transformation is the matrix ``B * C * Yiq2Rgb * Hue(h) * Sat(s) * V * Rgb2Yiq`` applied per pixel, plus the "
            "offset ``(128 - 128 * contrast) * brightness``.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("hue", "Hue delta, in degrees.", ArgValue::Float(0.0), true)
    .AddOptionalArg("saturation", "The saturation multiplier.", ArgValue::Float(1.0), true)
    .AddOptionalArg("value", "The value multiplier.", ArgValue::Float(1.0), true)
    .AddOptionalArg("brightness", "Brightness change factor.", ArgValue::Float(1.0), true)
    .AddOptionalArg("contrast", "Contrast change factor.", ArgValue::Float(1.0), true)
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});

// The siblings that are the same operator class under other schemas (color_twist.cc:26-103,142-144): arguments a
// schema does not define keep the neutral value.
DALI_SCHEMA(Hsv)
    .DocStr("Adjusts hue, saturation and value (brightness) of the images.\n\nThe operation is approximated by a linear "
            "transform in the RGB space: the color vector is projected along the neutral (gray) axis, rotated based on the "
            "hue delta, scaled based on the value and saturation multipliers, and restored to the original color space.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("hue", "Hue delta, in degrees.", ArgValue::Float(0.0), true)
    .AddOptionalArg("saturation", "The saturation multiplier.", ArgValue::Float(1.0), true)
    .AddOptionalArg("value", "The value multiplier.", ArgValue::Float(1.0), true)
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});
DALI_SCHEMA(Hue)
    .DocStr("Changes the hue level of the image.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("hue", "The hue change in degrees.", ArgValue::Float(0.0), true)
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});
DALI_SCHEMA(Saturation)
    .DocStr("Changes the saturation level of the image.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("saturation", "The saturation change factor (0: completely desaturated, 1: no change).",
                    ArgValue::Float(1.0), true)
    .AddOptionalTypeArg("dtype", "Output data type (same as input).", ArgType::INT)
    .InputLayout(0, {"HWC"});

class ColorTwistGpu : public OperatorBase {
 public:
  explicit ColorTwistGpu(const OpSpec &spec) : OperatorBase(spec) { CheckOutDtype(spec, "ColorTwist"); }
  void ExpectBlurInput() { blur_input_ = true; }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    blur_ = blur_input_ && ws.backend != OpType::CPU ? ws.Input(0).deferred_blur : nullptr;
    DALI_ENFORCE(!blur_input_ || ws.backend == OpType::CPU || blur_, "internal: expected the deferred arguments of the GaussianBlur in front");
    const TensorList &in = blur_ ? *blur_->source : ws.Input(0);
    CheckU8Hwc(in, "ColorTwist");
    int n = in.num_samples();
    // (AcquireArguments, color_twist.h:111-141: an argument the schema does not define keeps its neutral value)
    auto arg = [&](const char *name, float neutral) {
      return spec_.ArgumentDefined(name) ? GetPerSampleFloat(spec_, ws, name, n) : std::vector<float>(n, neutral);
    };
    auto hue = arg("hue", 0.f), sat = arg("saturation", 1.f), val = arg("value", 1.f), bri = arg("brightness", 1.f),
         con = arg("contrast", 1.f);
    descs_.assign(n, daliamdPointwiseDesc{});
    desc[0].type = DALI_UINT8;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      FillPointwiseCommon(d, in, i);
      DALI_ENFORCE(d.channels == 3, "ColorTwist expects 3-channel (RGB) input, got ", d.channels, " channels");
      d.transform = 1;
      float off;
      daliamdColorTwistMatrix(hue[i], sat[i], val[i], bri[i], con[i], d.matrix, &off);
      d.offset[0] = d.offset[1] = d.offset[2] = off;
      desc[0].shape[i] = in.shape(i);
    }
    return !fused_;  // fused: no buffer, the Erase behind this operator launches the kernel for both
  }
  void RunImpl(Workspace &ws) override {
    if (!fused_) {
      LaunchPointwise(ws, uploader_, descs_, "color_twist", blur_.get(), &blur_uploader_);
      return;
    }
    TensorList &out = ws.Output(0);
    auto d = std::make_shared<DeferredPointwise>();
    d->source = blur_ ? blur_->source : ws.inputs[0];
    d->blur = blur_;
    d->descs = descs_;
    out.Resize({}, DALI_UINT8);
    out.deferred_pointwise = d;
    out.SetLayout("HWC");
  }
  void EnableFusion() { fused_ = true; }

 private:
  bool fused_ = false, blur_input_ = false;
  std::shared_ptr<DeferredBlur> blur_;
  std::vector<daliamdPointwiseDesc> descs_;
  DescUploader uploader_, blur_uploader_;
};
DALI_REGISTER_OPERATOR(ColorTwist, ColorTwistGpu, GPU);
DALI_REGISTER_OPERATOR(ColorTwist, ColorTwistGpu, CPU);
DALI_REGISTER_OPERATOR(Hsv, ColorTwistGpu, GPU);
DALI_REGISTER_OPERATOR(Hsv, ColorTwistGpu, CPU);
DALI_REGISTER_OPERATOR(Hue, ColorTwistGpu, GPU);
DALI_REGISTER_OPERATOR(Hue, ColorTwistGpu, CPU);
DALI_REGISTER_OPERATOR(Saturation, ColorTwistGpu, GPU);
DALI_REGISTER_OPERATOR(Saturation, ColorTwistGpu, CPU);

DALI_SCHEMA(Erase)
    .DocStr("Erases one or more regions from the input tensors.\n\nThe region is specified by ``anchor`` (starting point) and "
            "``shape`` (dimensions) given for the (H, W) axes; several regions can be given as multiples of the number of axes.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("anchor", "Coordinates for the anchor or the starting point of the erase region.", ArgValue::FloatVec({}), true)
    .AddOptionalArg("shape", "Values for shape or dimensions of the erase region.", ArgValue::FloatVec({}), true)
    .AddOptionalTypeArg("axes", "Order of dimensions used for anchor and shape, as dimension indices.", ArgType::INT_VEC)
    .AddOptionalTypeArg("axis_names", "Order of dimensions used for anchor and shape, as layout names (default \"HW\").",
                        ArgType::STRING)
    .AddOptionalArg("fill_value", "Value to fill the erased region (one value or one per channel).", ArgValue::FloatVec({0.0}), true)
    .AddOptionalArg("normalized_anchor", "Whether the anchor is given in normalized coordinates.", ArgValue::Bool(false))
    .AddOptionalArg("normalized_shape", "Whether the shape is given in normalized coordinates.", ArgValue::Bool(false))
    .AddOptionalTypeArg("normalized", "Whether anchor and shape are given in normalized coordinates.", ArgType::BOOL)
    .AddOptionalArg("centered_anchor", "If True, the anchors refer to the center of the region.", ArgValue::Bool(false))
    .InputLayout(0, {"HWC"});

class EraseGpu : public OperatorBase {
 public:
  explicit EraseGpu(const OpSpec &spec) : OperatorBase(spec) {
    norm_anchor_ = spec.GetBool("normalized_anchor");
    norm_shape_ = spec.GetBool("normalized_shape");
    if (spec.TryArg("normalized")) {
      DALI_ENFORCE(!spec.Args().count("normalized_anchor") && !spec.Args().count("normalized_shape"),
                   "`normalized` argument is incompatible with providing a separate value for `normalized_anchor` and "
                   "`normalized_shape`");
      norm_anchor_ = norm_shape_ = spec.GetBool("normalized");
    }
    centered_ = spec.GetBool("centered_anchor");
    // axes: indices into HWC, or names; default = all but C, i.e. (H, W)
    axes_ = {0, 1};
    if (spec.TryArg("axis_names")) {
      std::string names = spec.GetString("axis_names");
      axes_.clear();
      for (char ch : names) {
        DALI_ENFORCE(ch == 'H' || ch == 'W', "Erase (gpu) supports the H and W axes, got '", std::string(1, ch), "'");
        axes_.push_back(ch == 'H' ? 0 : 1);
      }
    } else if (spec.TryArg("axes")) {
      axes_.clear();
      for (auto a : spec.GetIntVec("axes")) {
        DALI_ENFORCE(a == 0 || a == 1, "Erase (gpu) supports axes 0 (H) and 1 (W), got ", a);
        axes_.push_back((int)a);
      }
    }
    DALI_ENFORCE(!axes_.empty(), "At least one axis is required");
  }
  void ExpectFusedInput() { fused_input_ = true; }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    // a fused ColorTwist in front: work on ITS input and carry its transform along
    const DeferredPointwise *def = fused_input_ ? ws.Input(0).deferred_pointwise.get() : nullptr;
    DALI_ENFORCE(!fused_input_ || def, "internal: Erase expected the deferred arguments of the ColorTwist in front of it");
    // a blur fused in front: directly, or in front of the fused ColorTwist
    blur_ = def ? def->blur : (blur_input_ && ws.backend != OpType::CPU ? ws.Input(0).deferred_blur : nullptr);
    DALI_ENFORCE(def || !blur_input_ || ws.backend == OpType::CPU || blur_, "internal: expected the deferred arguments of the GaussianBlur in front");
    const TensorList &in = def ? *def->source : (blur_ ? *blur_->source : ws.Input(0));
    CheckU8Hwc(in, "Erase");
    int n = in.num_samples();
    auto anchors = GetPerSampleFloatVec(spec_, ws, "anchor", n), shapes = GetPerSampleFloatVec(spec_, ws, "shape", n);
    auto fills = GetPerSampleFloatVec(spec_, ws, "fill_value", n);
    descs_.assign(n, daliamdPointwiseDesc{});
    desc[0].type = DALI_UINT8;
    desc[0].shape.resize(n);
    int naxes = (int)axes_.size();
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      if (def) d = def->descs[i];
      else FillPointwiseCommon(d, in, i);
      auto a = anchors[i], s = shapes[i];
      if (a.empty() && !s.empty()) a.assign(s.size(), 0.0f);
      DALI_ENFORCE(a.size() == s.size(), "`anchor` and `shape` must have the same number of elements");
      DALI_ENFORCE(s.size() % naxes == 0, "The number of `anchor`/`shape` values must be a multiple of the number of axes");
      int nregions = (int)s.size() / naxes;
      DALI_ENFORCE(nregions <= DALIAMD_MAX_ERASE_REGIONS, "At most ", DALIAMD_MAX_ERASE_REGIONS, " regions are supported");
      d.num_regions = nregions;
      int dims[2] = {d.h, d.w};
      for (int r = 0, k = 0; r < nregions; r++) {
        int64_t lo[2] = {0, 0}, hi[2] = {d.h, d.w};
        for (int j = 0; j < naxes; j++, k++) {
          int axis = axes_[j];
          float anchor_val = norm_anchor_ ? a[k] * dims[axis] : a[k];
          float shape_val = norm_shape_ ? s[k] * dims[axis] : s[k];
          if (centered_) anchor_val -= shape_val / 2;
          lo[axis] = (int64_t)anchor_val;
          hi[axis] = (int64_t)(anchor_val + shape_val);
        }
        d.region[r][0] = (int)std::max<int64_t>(0, lo[0]); d.region[r][1] = (int)std::max<int64_t>(0, lo[1]);
        d.region[r][2] = (int)std::min<int64_t>(d.h, hi[0]); d.region[r][3] = (int)std::min<int64_t>(d.w, hi[1]);
      }
      const auto &f = fills[i];
      DALI_ENFORCE(f.size() == 1 || (int)f.size() == d.channels, "`fill_value` must hold one value or one per channel");
      for (int c = 0; c < 4; c++) d.fill[c] = f.size() == 1 ? f[0] : (c < (int)f.size() ? f[c] : 0.0f);
      desc[0].shape[i] = in.shape(i);
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    LaunchPointwise(ws, uploader_, descs_, fused_input_ ? "color_twist+erase" : "erase", blur_.get(), &blur_uploader_);
  }
  void ExpectBlurInput() { blur_input_ = true; }

 private:
  bool fused_input_ = false, blur_input_ = false;
  std::shared_ptr<DeferredBlur> blur_;
  DescUploader blur_uploader_;
  bool norm_anchor_, norm_shape_, centered_;
  std::vector<int> axes_;
  std::vector<daliamdPointwiseDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(Erase, EraseGpu, GPU);
DALI_REGISTER_OPERATOR(Erase, EraseGpu, CPU);

void TryEnableBlurFusion(OperatorBase *producer, OperatorBase *consumer) {
  // OPT-IN (DALI_AMD_BLUR_FUSION=1).  Measured on configs[2] (512 x 512, batch 128, four batches in flight): the fused
  // write-out needs the rounded tile in LDS and a workgroup barrier in front of it, which costs the blur more (930 ->
  // 1 244 us) than the pointwise launch it replaces (104 us) - the 201 MB it keeps out of HBM are not the bound of this
  // VALU-heavy kernel.  Kept for the parity tests and for a version of the H pass that owns whole pixels in registers.
  if (!(getenv("DALI_AMD_BLUR_FUSION") && atoi(getenv("DALI_AMD_BLUR_FUSION")) != 0)) return;
  auto *blur = dynamic_cast<GaussianBlurGpu *>(producer);
  if (!blur) return;
  if (auto *twist = dynamic_cast<ColorTwistGpu *>(consumer)) {
    blur->EnableFusion();
    twist->ExpectBlurInput();
  } else if (auto *erase = dynamic_cast<EraseGpu *>(consumer)) {
    blur->EnableFusion();
    erase->ExpectBlurInput();
  }
}

void TryEnablePointwiseFusion(OperatorBase *producer, OperatorBase *consumer) {
  auto *twist = dynamic_cast<ColorTwistGpu *>(producer);
  auto *erase = dynamic_cast<EraseGpu *>(consumer);
  if (!twist || !erase) return;
  twist->EnableFusion();
  erase->ExpectFusedInput();
}

}  // namespace daliamd_host
