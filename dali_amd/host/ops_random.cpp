// random.coin_flip / random.uniform (CPU): per-sample Philox streams exactly like the reference
//   OperatorWithRng        dali/operators/random/rng_base.h:67-140 (master(seed,0,0); sample i = master advanced by
//                          i*65537 sequences; element p skips 257*p draws; master += batch per run)
//   bernoulli_dist         dali/operators/random/random_dist.h:293-312
//   uniform_real_dist      dali/operators/random/random_dist.h:168-205
//   checkpoint format      dali/core/random/philox.cc:75-108
#include <cmath>

#include "dali_amd_host.h"
#include "ops.h"

namespace daliamd_host {

DALI_SCHEMA(RNGAttr)
    .DocStr("Random number generator arguments.")
    .MakeInternal()
    .AddOptionalTypeArg("shape", "Shape of the output data (default: one value per sample).", ArgType::INT_VEC, true)
    .AddOptionalTypeArg("dtype", "Output data type.", ArgType::INT)
    .AddRandomSeedArg();

class RngOp : public OperatorBase {
 public:
  explicit RngOp(const OpSpec &spec) : OperatorBase(spec) {
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
  std::vector<int64_t> SampleShape() const {
    if (spec_.TryArg("shape")) return spec_.GetIntVec("shape");
    return {};
  }
  void Advance(int batch) { daliamdPhiloxAdvanceSequence(&master_, (uint64_t)batch); }
  // generator for element p of sample i
  daliamdPhiloxState ElementState(int i, int64_t p) const {
    daliamdPhiloxState s = master_;
    s.ctr[1] += (uint64_t)i * 65537ull;
    // skipahead(p * 257)
    uint64_t n = (uint64_t)p * 257ull;
    s.phase += (int)(n & 3);
    n >>= 2;
    if (s.phase > 3) { n++; s.phase -= 4; }
    uint64_t old = s.ctr[0];
    s.ctr[0] += n;
    if (s.ctr[0] < old) s.ctr[1]++;
    return s;
  }
  daliamdPhiloxState master_;
};

template <typename T>
static void StoreAs(void *dst, int64_t idx, DALIDataType t, T v) {
  switch (t) {
    case DALI_INT32: static_cast<int32_t *>(dst)[idx] = (int32_t)v; break;
    case DALI_INT64: static_cast<int64_t *>(dst)[idx] = (int64_t)v; break;
    case DALI_UINT8: case DALI_BOOL: static_cast<uint8_t *>(dst)[idx] = (uint8_t)v; break;
    case DALI_FLOAT: static_cast<float *>(dst)[idx] = (float)v; break;
    case DALI_FLOAT64: static_cast<double *>(dst)[idx] = (double)v; break;
    default: DALI_FAIL("Data type ", TypeName(t), " is currently not supported. Supported types are : uint8, bool, int32, "
                       "int64, float, double");
  }
}

// ---------------------------------------------------------------------------------------------
DALI_SCHEMA(random__CoinFlip)
    .DocStr("Generates random boolean values following a bernoulli distribution.\n\nThe probability of generating a "
            "value 1 (true) is determined by the ``probability`` argument.")
    .NumInput(0, 1)
    .NumOutput(1)
    .AddOptionalArg("probability", "Probability of value 1.", ArgValue::Float(0.5), true)
    .AddParent("RNGAttr");
DALI_SCHEMA(CoinFlip).DocStr("Legacy alias of random.coin_flip").NumInput(0, 1).NumOutput(1).AddParent("random__CoinFlip");

class CoinFlipOp : public RngOp {
 public:
  explicit CoinFlipOp(const OpSpec &spec) : RngOp(spec) {
    dtype_ = spec.TryArg("dtype") ? (DALIDataType)spec.GetInt("dtype") : DALI_INT32;
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    n_ = ws.NumInput() ? ws.GetInputBatchSize(0) : ws.batch_size;
    desc[0].type = dtype_;
    TensorShape shape = ws.NumInput() ? ws.Input(0).shape(0) : SampleShape();
    desc[0].shape.assign(n_, shape);
    if (ws.NumInput())
      for (int i = 0; i < n_; i++) desc[0].shape[i] = ws.Input(0).shape(i);
    return true;
  }
  void RunImpl(Workspace &ws) override {
    auto prob = GetPerSampleFloat(spec_, ws, "probability", n_);
    TensorList &out = ws.Output(0);
    for (int i = 0; i < n_; i++) {
      float th = prob[i] * 0x1p32f;
      uint32_t threshold = th >= 0x1p32f ? 0xffffffffu : (uint32_t)th;
      int64_t count = volume(out.shape(i));
      for (int64_t p = 0; p < count; p++) {
        daliamdPhiloxState s = ElementState(i, p);
        uint32_t r;
        daliamdPhiloxGenerate(&s, &r, 1);
        StoreAs(out.raw(i), p, dtype_, r <= threshold ? 1 : 0);
      }
    }
    Advance(n_);
  }

 private:
  DALIDataType dtype_;
  int n_ = 0;
};
DALI_REGISTER_OPERATOR(random__CoinFlip, CoinFlipOp, CPU);
DALI_REGISTER_OPERATOR(CoinFlip, CoinFlipOp, CPU);

// ---------------------------------------------------------------------------------------------
DALI_SCHEMA(random__Uniform)
    .DocStr("Generates random numbers following a uniform distribution in the ``range`` [min, max), or one of the "
            "discrete ``values`` with equal probability.")
    .NumInput(0, 1)
    .NumOutput(1)
    .AddOptionalArg("range", "Range ``[min, max)`` of a continuous uniform distribution.",
                    ArgValue::FloatVec({-1.0, 1.0}), true)
    .AddOptionalTypeArg("values", "The discrete values produced by a discrete uniform distribution.", ArgType::FLOAT_VEC,
                        true)
    .AddParent("RNGAttr");
DALI_SCHEMA(Uniform).DocStr("Legacy alias of random.uniform").NumInput(0, 1).NumOutput(1).AddParent("random__Uniform");

class UniformOp : public RngOp {
 public:
  explicit UniformOp(const OpSpec &spec) : RngOp(spec) {
    dtype_ = spec.TryArg("dtype") ? (DALIDataType)spec.GetInt("dtype") : DALI_FLOAT;
    DALI_ENFORCE(!(spec.TryArg("values") && spec.Args().count("range")),
                 "`values` and `range` are mutually exclusive");
    if (spec.TryArg("values")) values_ = spec.GetFloatVec("values");
    else {
      auto r = spec.GetFloatVec("range");
      DALI_ENFORCE(r.size() == 2, "`range` must have exactly two elements");
      lo_ = (float)r[0]; hi_ = (float)r[1];
    }
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    n_ = ws.NumInput() ? ws.GetInputBatchSize(0) : ws.batch_size;
    desc[0].type = dtype_;
    desc[0].shape.assign(n_, SampleShape());
    if (ws.NumInput())
      for (int i = 0; i < n_; i++) desc[0].shape[i] = ws.Input(0).shape(i);
    return true;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    // uniform_real_dist<float>(start, end): random_dist.h:175-205
    float min_v = lo_, max_v = std::nextafter(hi_, lo_);
    if (min_v > max_v) std::swap(min_v, max_v);
    float factor = (max_v - min_v) * 0x1p-32f;
    for (int i = 0; i < n_; i++) {
      int64_t count = volume(out.shape(i));
      for (int64_t p = 0; p < count; p++) {
        daliamdPhiloxState s = ElementState(i, p);
        uint32_t r;
        daliamdPhiloxGenerate(&s, &r, 1);
        if (!values_.empty()) {
          // uniform_discrete_dist: idx = (u * nvalues) >> 32   (random_dist.h:266-284)
          uint64_t idx = ((uint64_t)r * (uint64_t)values_.size()) >> 32;
          StoreAs(out.raw(i), p, dtype_, (float)values_[idx]);
        } else {
          float val = std::fma((float)r, factor, min_v);
          val = std::min(val, max_v);
          if (dtype_ == DALI_FLOAT || dtype_ == DALI_FLOAT64) StoreAs(out.raw(i), p, dtype_, val);
          else StoreAs(out.raw(i), p, dtype_, std::round(val));
        }
      }
    }
    Advance(n_);
  }

 private:
  DALIDataType dtype_;
  std::vector<double> values_;
  float lo_ = -1, hi_ = 1;
  int n_ = 0;
};
DALI_REGISTER_OPERATOR(random__Uniform, UniformOp, CPU);
DALI_REGISTER_OPERATOR(Uniform, UniformOp, CPU);

}  // namespace daliamd_host
