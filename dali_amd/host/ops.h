// Cross-file hooks between the pipeline/executor and operator implementations.
#ifndef DALI_AMD_HOST_OPS_H_
#define DALI_AMD_HOST_OPS_H_
#include "framework.h"

namespace daliamd_host {

// ExternalSource: queue one batch (copied) for the operator instance
void FeedExternalSource(OperatorBase *op, const std::vector<const void *> &data, const std::vector<TensorShape> &shapes,
                        DALIDataType type, const std::string &layout);

// If `producer` is a resampling operator (RandomResizedCrop / Resize) and `consumer` a compatible
// CropMirrorNormalize, the producer stops launching its own kernel and hands its per-sample resampling
// arguments to the consumer, which launches ONE fused resample + normalise kernel.
void TryEnableFusion(OperatorBase *producer, OperatorBase *consumer);

// Resampling arguments deferred from a producer to its CropMirrorNormalize consumer.
struct DeferredResample {
  std::shared_ptr<TensorList> source;                 // the producer's input (u8 HWC on the device)
  std::vector<daliamdResampleArgs> args;              // one per sample; `out*`/epilogue fields unset
  int out_h = 0, out_w = 0, channels = 0;
};

}  // namespace daliamd_host
#endif
