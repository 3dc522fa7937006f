// A tiny out-of-tree operator library for tests/test_plugin_manager.py: compiled against dali_amd/host/framework.h,
// loaded with dali_amd.plugin_manager.load_library - the way a user's custom operator reaches the registry
// (reference: docs/examples/custom_operations/custom_operator, dali/plugin/plugin_manager.cc).
#include "framework.h"

namespace custom {
using namespace daliamd_host;

DALI_SCHEMA(custom__AddConstant)
    .DocStr("Adds `value` to every element of a uint8 tensor (saturating).")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("value", "The constant to add.", ArgValue::Int(1));

class AddConstant : public OperatorBase {
 public:
  explicit AddConstant(const OpSpec &spec) : OperatorBase(spec), value_((int)spec.GetInt("value")) {}
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    DALI_ENFORCE(in.type() == DALI_UINT8, "custom.add_constant expects uint8 data");
    desc[0].type = in.type();
    desc[0].shape.clear();
    for (int i = 0; i < in.num_samples(); i++) desc[0].shape.push_back(in.shape(i));
    return true;
  }
  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    out.SetLayout(in.layout());
    for (int i = 0; i < in.num_samples(); i++) {
      ws.GetThreadPool().AddWork([&, i](int) {
        const uint8_t *src = static_cast<const uint8_t *>(in.raw(i));
        uint8_t *dst = static_cast<uint8_t *>(out.raw(i));
        const int64_t n = volume(in.shape(i));
        for (int64_t k = 0; k < n; k++) {
          const int v = src[k] + value_;
          dst[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
      });
    }
    ws.GetThreadPool().RunAll();
  }

 private:
  int value_;
};
DALI_REGISTER_OPERATOR(custom__AddConstant, AddConstant, CPU);

}  // namespace custom
