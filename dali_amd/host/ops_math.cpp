// fn.normalize, host side.
//   schema / argument rules   dali/operators/math/normalize/normalize.cc:24-123, normalize.h:60-330
//   arithmetic                 normalize.cc:209-244 (FoldMeans / FoldStdDev), normalize_utils.h:133-220
// The reductions and the element-wise pass run in libdali_amd_kernels.so (csrc/normalize.hip).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ops.h"
#include "pipeline.h"
#include "dali_amd_host.h"

namespace daliamd_host {

DALI_SCHEMA(Normalize)
    .DocStr("Normalizes the input by removing the mean and dividing by the standard deviation.\n\n"
            "The mean and standard deviation can be calculated internally for the specified subset of axes or can be "
            "externally provided as the `mean` and `stddev` arguments (scalars in this build).\n\n"
            "The normalization is done following the formula::\n\n  out = scale * (in - mean) / stddev + shift\n\n"
            "In this MI355X-native build the reduced axes must form one contiguous group (for example \"HW\" of an HWC "
            "image, the last axis of a spectrogram, or all axes).")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("batch", "If set to True, the mean and standard deviation are calculated across tensors in the batch.",
                    ArgValue::Bool(false))
    .AddOptionalTypeArg("mean", "Mean value to be subtracted from the data (scalar). If not specified, the mean is "
                        "calculated from the input.", ArgType::FLOAT)
    .AddOptionalTypeArg("stddev", "Standard deviation value to scale the data (scalar). If not specified, the standard "
                        "deviation is calculated from the input.", ArgType::FLOAT)
    .AddOptionalTypeArg("axes", "Indices of dimensions along which the input is normalized. By default, all axes are used.",
                        ArgType::INT_VEC)
    .AddOptionalTypeArg("axis_names", "Names of the axes in the input. Axis indices are taken from the input layout, and "
                        "this argument cannot be used with `axes`.", ArgType::STRING)
    .AddOptionalArg("shift", "The value to which the mean will map in the output.", ArgValue::Float(0.0))
    .AddOptionalArg("scale", "The scaling factor applied to the output.", ArgValue::Float(1.0))
    .AddOptionalArg("epsilon", "A value that is added to the variance to avoid division by small numbers.", ArgValue::Float(0.0))
    .AddOptionalArg("ddof", "Delta Degrees of Freedom for Bessel's correction.", ArgValue::Int(0))
    .AddOptionalArg("dtype", "Output data type (FLOAT, FLOAT16, UINT8 or INT8).", ArgValue::Int(DALI_FLOAT));

class NormalizeGpu : public OperatorBase {
 public:
  explicit NormalizeGpu(const OpSpec &spec) : OperatorBase(spec) {
    batch_ = spec.GetBool("batch");
    has_mean_ = spec.ArgumentDefined("mean");
    has_stddev_ = spec.ArgumentDefined("stddev");
    DALI_ENFORCE(!spec.HasTensorArgument("mean") && !spec.HasTensorArgument("stddev"),
                 "Normalize (gpu): tensor `mean` / `stddev` arguments are not supported yet, only scalars");
    if (has_mean_) mean_value_ = (float)spec.GetFloat("mean");
    if (has_stddev_) stddev_value_ = (float)spec.GetFloat("stddev");
    shift_ = (float)spec.GetFloat("shift");
    scale_ = (float)spec.GetFloat("scale");
    epsilon_ = (float)spec.GetFloat("epsilon");
    ddof_ = (int)spec.GetInt("ddof");
    DALI_ENFORCE(ddof_ >= 0, "The parameter 'ddof' must be a non-negative integer.");
    DALI_ENFORCE(epsilon_ >= 0, "The value of 'epsilon' must be non-negative.");
    out_type_ = (DALIDataType)spec.GetInt("dtype");
    ToKernelDType(out_type_);
    has_axes_ = spec.ArgumentDefined("axes");
    has_axis_names_ = spec.ArgumentDefined("axis_names");
    DALI_ENFORCE(!(has_axes_ && has_axis_names_), "Normalize: Arguments `axes` and `axis_names` are mutually exclusive");
    ring_ = spec.TryArg("gpu_prefetch_queue_depth") ? (int)spec.GetInt("gpu_prefetch_queue_depth") + 1 : 1;  // (not given to CPU operators)
  }

  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    const int n = in.num_samples();
    DALI_ENFORCE(in.type() == DALI_UINT8 || in.type() == DALI_FLOAT, "Normalize (gpu): the input must be uint8 or float, got ",
                 TypeName(in.type()));
    desc[0].type = out_type_;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) desc[0].shape[i] = in.shape(i);
    return true;
  }

  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    out.SetLayout(in.layout());
    const int n = in.num_samples();
    if (!n) return;
    const int ndim = (int)in.shape(0).size();
    // ---- the reduced axes: one contiguous group [a0, a1] ----
    std::vector<int> axes;
    if (has_axes_) {
      for (int64_t a : spec_.GetIntVec("axes")) axes.push_back((int)a);
    } else if (has_axis_names_) {
      const std::string names = spec_.GetString("axis_names"), &layout = in.layout();
      for (char c : names) {
        size_t pos = layout.find(c);
        DALI_ENFORCE(pos != std::string::npos, "Axis \"", std::string(1, c), "\" is not present in the input layout \"", layout, "\"");
        axes.push_back((int)pos);
      }
    } else {
      for (int a = 0; a < ndim; a++) axes.push_back(a);
    }
    std::sort(axes.begin(), axes.end());
    for (size_t k = 0; k < axes.size(); k++) {
      DALI_ENFORCE(axes[k] >= 0 && axes[k] < ndim, "Axis index out of range: ", axes[k], " for a ", ndim, "-D input");
      DALI_ENFORCE(k == 0 || axes[k] != axes[k - 1], "Axis index ", axes[k], " occurs more than once");
      DALI_ENFORCE(k == 0 || axes[k] == axes[k - 1] + 1, "Normalize (gpu): the reduced axes must be adjacent (got a gap "
                   "between ", axes[k - 1], " and ", axes[k], "); other axis sets are not supported yet");
    }
    const bool calc_mean = !has_mean_, calc_std = !has_stddev_;
    const int slot = (int)(ws.iteration % ring_);
    // ---- descriptors ----
    descs_.assign(n, daliamdNormalizeDesc{});
    int64_t bins_total = 0;
    std::vector<int64_t> bin_off(n, 0);
    double batch_count = 0;
    size_t dense_bytes = 0;
    std::vector<size_t> dense_off(n, 0);
    for (int i = 0; i < n; i++) {
      const TensorShape &sh = in.shape(i);
      DALI_ENFORCE((int)sh.size() == ndim, "All samples must have the same number of dimensions");
      auto &d = descs_[i];
      d.outer = d.reduced = d.inner = 1;
      for (int a = 0; a < ndim; a++) {
        if (axes.empty() || a < axes.front()) d.outer *= sh[a];
        else if (a > axes.back()) d.inner *= sh[a];
        else d.reduced *= sh[a];
      }
      DALI_ENFORCE(d.outer * d.reduced * d.inner > 0, "Normalize (gpu): empty samples are not supported");
      if (batch_) {
        DALI_ENFORCE(d.outer == descs_[0].outer && d.inner == descs_[0].inner,
                     "Batch normalization requires that non-reduced dimensions have equal extent in all samples in the batch");
        batch_count += (double)d.reduced;
      } else {
        bin_off[i] = bins_total;
        bins_total += d.outer * d.inner;
      }
      if (in.row_pitch(i) && ndim == 3 && in.row_pitch(i) != sh[1] * sh[2] * TypeSize(in.type())) {
        dense_off[i] = dense_bytes;  // row-padded image (decoder output): densify first
        dense_bytes += ((size_t)volume(sh) * TypeSize(in.type()) + 255) & ~(size_t)255;
      } else {
        dense_off[i] = (size_t)-1;
      }
    }
    if (batch_) bins_total = descs_[0].outer * descs_[0].inner;
    if (ws.backend == OpType::CPU) {  // host kernel: one task per sample, or one for the batch when it shares the statistics
      DALI_ENFORCE(out_type_ != DALI_FLOAT16, "Normalize (cpu): float16 output is not supported");
      float scalar_inv_std_cpu = 1;
      if (has_stddev_)
        scalar_inv_std_cpu = epsilon_ ? scale_ / std::sqrt(stddev_value_ * stddev_value_ + epsilon_) : scale_ / stddev_value_;
      host_samples_.assign(n, daliamdNormalizeHostSample{});
      host_dense_.resize(n);
      for (int i = 0; i < n; i++) {
        const TensorShape &sh = in.shape(i);
        const void *src = in.raw(i);
        if (dense_off[i] != (size_t)-1) {  // row-padded image: densify
          const size_t row = (size_t)sh[1] * sh[2] * TypeSize(in.type());
          host_dense_[i].resize(row * (size_t)sh[0]);
          for (int64_t y = 0; y < sh[0]; y++)
            std::memcpy(host_dense_[i].data() + y * row, static_cast<const uint8_t *>(src) + y * (size_t)in.row_pitch(i), row);
          src = host_dense_[i].data();
        }
        host_samples_[i] = {src, out.raw(i), descs_[i].outer, descs_[i].reduced, descs_[i].inner};
      }
      const int kin = ToKernelDType(in.type()), kout = ToKernelDType(out_type_);
      auto run = [this, kin, kout, scalar_inv_std_cpu](int first, int count) {
        if (daliamdNormalizeHost(host_samples_.data() + first, count, kin, kout, has_mean_, mean_value_, has_stddev_, scalar_inv_std_cpu,
                                 ddof_, epsilon_, scale_, shift_) != 0)
          DALI_FAIL(daliamdHostGetLastErrorMessage());
      };
      if (batch_) {
        run(0, n);
      } else {
        for (int i = 0; i < n; i++) ws.GetThreadPool().AddWork([run, i](int) { run(i, 1); }, volume(in.shape(i)));
        ws.GetThreadPool().RunAll();
      }
      NoteLaunch(ws, "host_normalize");
      return;
    }
    if (scratch_.empty())
      for (int k = 0; k < ring_; k++) {
        scratch_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
        dense_.emplace_back(std::make_unique<Buffer>(StorageDevice::GPU));
      }
    Buffer &scr = *scratch_[slot], &dense = *dense_[slot];
    // [sum_mean f64][sum_var f64][mean f32][inv_std f32], each bins_total long
    const size_t sums_bytes = 2 * sizeof(double) * (size_t)bins_total;
    scr.Reserve(sums_bytes + 2 * sizeof(float) * (size_t)bins_total + 256);
    if (dense_bytes) dense.Reserve(dense_bytes);
    double *sum_mean = static_cast<double *>(scr.data()), *sum_var = sum_mean + bins_total;
    float *mean = reinterpret_cast<float *>(sum_var + bins_total), *inv_std = mean + bins_total;
    if (calc_mean || calc_std) KCHECK(daliamdMemsetAsync(scr.data(), 0, sums_bytes, ws.stream));
    float scalar_inv_std = 1;
    if (has_stddev_)  // normalize.cc:291-296
      scalar_inv_std = epsilon_ ? scale_ / std::sqrt(stddev_value_ * stddev_value_ + epsilon_) : scale_ / stddev_value_;
    for (int i = 0; i < n; i++) {
      auto &d = descs_[i];
      const TensorShape &sh = in.shape(i);
      const void *src = in.raw(i);
      if (dense_off[i] != (size_t)-1) {
        void *dst = static_cast<uint8_t *>(dense.data()) + dense_off[i];
        size_t row = (size_t)sh[1] * sh[2] * TypeSize(in.type());
        KCHECK(daliamdMemcpy2DD2DAsync(dst, row, src, (size_t)in.row_pitch(i), row, (size_t)sh[0], ws.stream));
        src = dst;
      }
      d.in = src;
      d.out = out.raw(i);
      const int64_t off = batch_ ? 0 : bin_off[i];
      d.sum_mean = sum_mean + off; d.sum_var = sum_var + off; d.mean = mean + off; d.inv_std = inv_std + off;
      d.stat_count = batch_ ? batch_count : (double)d.reduced;
      d.use_scalar_mean = has_mean_; d.scalar_mean = mean_value_;
      d.use_scalar_inv_std = has_stddev_; d.scalar_inv_std = scalar_inv_std;
      d.in_dtype = ToKernelDType(in.type());
      d.out_dtype = ToKernelDType(out_type_);
      d.owns_stats = !batch_ || i == 0;
    }
    int stat_wg = 0, apply_wg = 0;
    int64_t max_bins = 0;
    KCHECK(daliamdNormalizeSetup(descs_.data(), n, &stat_wg, &apply_wg, &max_bins));
    auto *dev = static_cast<const daliamdNormalizeDesc *>(
        uploader_.Upload(descs_.data(), descs_.size() * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdNormalizeRun(ws.stream, dev, n, stat_wg, apply_wg, max_bins, calc_mean, calc_std, ddof_, epsilon_, scale_,
                               shift_));
    NoteLaunch(ws, "normalize");
  }

 private:
  bool batch_, has_mean_, has_stddev_, has_axes_, has_axis_names_;
  float mean_value_ = 0, stddev_value_ = 1, shift_, scale_, epsilon_;
  int ddof_, ring_;
  DALIDataType out_type_;
  std::vector<daliamdNormalizeDesc> descs_;
  std::vector<daliamdNormalizeHostSample> host_samples_;
  std::vector<std::vector<uint8_t>> host_dense_;
  std::vector<std::unique_ptr<Buffer>> scratch_, dense_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(Normalize, NormalizeGpu, GPU);
DALI_REGISTER_OPERATOR(Normalize, NormalizeGpu, CPU);  // same class: the host kernel when run on the CPU

}  // namespace daliamd_host
