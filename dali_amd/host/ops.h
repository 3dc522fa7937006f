// Cross-file hooks between the pipeline/executor and operator implementations.
#ifndef DALI_AMD_HOST_OPS_H_
#define DALI_AMD_HOST_OPS_H_
#include <deque>
#include <random>

#include "framework.h"

namespace daliamd_host {

// ExternalSource: queue one batch (copied) for the operator instance
void FeedExternalSource(OperatorBase *op, const std::vector<const void *> &data, const std::vector<TensorShape> &shapes,
                        DALIDataType type, const std::string &layout);

// If `producer` is a resampling operator (RandomResizedCrop / Resize) and `consumer` a compatible
// CropMirrorNormalize, the producer stops launching its own kernel and hands its per-sample resampling
// arguments to the consumer, which launches ONE fused resample + normalise kernel.
void TryEnableFusion(OperatorBase *producer, OperatorBase *consumer);
// A decoders.image (mixed) whose ONLY consumer is a RandomResizedCrop decodes only the windows that operator draws (the
// operator draws them when the decoder's stage runs; same generator, same sequence): bit-identical output, no work for
// the part of every image the crop discards.  DALI_AMD_ROI_FUSION=0 switches it off.  Returns true when the pair was fused.
bool TryEnableRoiDecodeFusion(OperatorBase *decoder, OperatorBase *consumer);

// Resampling arguments deferred from a producer to its CropMirrorNormalize consumer.
struct DeferredResample {
  std::shared_ptr<TensorList> source;                 // the producer's input (u8 HWC on the device)
  std::vector<daliamdResampleArgs> args;              // one per sample; `out*`/epilogue fields unset
  int out_h = 0, out_w = 0, channels = 0;
};

// Colour-twist arguments deferred to an Erase consumer: both are the same streaming kernel (transform and / or erase per
// descriptor), so `erase(color_twist(x))` is one launch and the intermediate image never exists.
// A Gaussian blur whose only consumer is a pointwise operator (ColorTwist and its siblings, Erase, or the two fused):
// the blur kernel applies the pointwise arithmetic to the rounded pixels of a tile before they leave the workgroup
// (daliamdGaussianBlurPointwiseRun) - no launch and no pass over the image for the operators behind the blur.
struct DeferredBlur {
  std::shared_ptr<TensorList> source;                 // the blur's input (u8 HWC on the device)
  std::vector<daliamdGaussianBlurDesc> descs;         // one per sample: in / shape / windows set
};
struct DeferredPointwise {
  std::shared_ptr<TensorList> source;                 // the producer's input (u8 HWC on the device)
  std::vector<daliamdPointwiseDesc> descs;            // one per sample: in / shape / transform / matrix / offset set
  std::shared_ptr<DeferredBlur> blur;                 // a blur fused in front of the pair
};
void TryEnablePointwiseFusion(OperatorBase *producer, OperatorBase *consumer);
void TryEnableBlurFusion(OperatorBase *producer, OperatorBase *consumer);

// Spectrogram -> MelFilterBank (-> ToDecibels): the power spectrum is consumed where it is produced.  A 513-bin
// spectrogram is 6.4 times the size of its 80-filter mel reduction; written to HBM and read back it is the whole traffic
// of the audio path.  The fused kernel multiplies every 16-frame tile by the filter bank while it sits in LDS
// (daliamdSpectrogramMelRun), so the spectrogram operator only hands its arguments on and the LAST operator of the chain
// launches: MelFilterBank, or ToDecibels (inside the same launch when `reference` is given; as an in-place pass behind it
// when the reference is the sample's maximum, which the fused kernel collects).
struct DeferredAudio {
  std::shared_ptr<TensorList> source;                 // the signal
  daliamdSpectrogramParams params;
  const float *window_dev = nullptr, *twiddles_dev = nullptr;
  std::vector<daliamdSpectrogramDesc> descs;          // in / length / num_windows / wg_start set
  int nwg = 0;
  // set by a MelFilterBank that defers to a ToDecibels behind it
  bool has_mel = false;
  daliamdSpecMelParams mel{};
};
// true when the pair was fused (the pipeline then tries to extend the chain)
bool TryEnableAudioFusion(OperatorBase *producer, OperatorBase *consumer);
// decoders.audio -> copy to the device -> Spectrogram (gpu): 16-bit PCM travels as int16, the kernel's load converts
void TryEnablePcm16Fusion(OperatorBase *decoder, OperatorBase *spectrogram);

// The sample-index stream every reader draws from (Loader, dali/operators/reader/loader/loader.h:78-503,
// loader.cc:78-87): sequential over the data set starting at this shard (start = size * shard_id / num_shards),
// moving on to the next shard every epoch unless stick_to_shard, a shuffle reservoir of `initial_fill` samples,
// padding of the last batch, epoch bookkeeping for the iterators (ReaderMeta) and a complete checkpoint.
// A reader owns one, tells it the data set size once, and asks for one index per output sample.
class Loader {
 public:
  explicit Loader(const OpSpec &spec);
  void Init(int64_t size);          // after the data set has been discovered
  int64_t Size() const { return size_; }
  int64_t NextIndex(bool is_new_batch);
  ReaderMeta Meta() const;
  std::string Save() const;
  void Restore(const std::string &state);

 private:
  void Reset(bool wrap_to_shard);
  bool IsNextShard(int64_t idx) const;
  int64_t ReadSequential();
  bool shuffle_;
  int initial_fill_, num_shards_, shard_id_;
  bool stick_to_shard_, pad_last_batch_;
  int64_t size_ = 0;
  std::default_random_engine rng_;
  int virtual_shard_id_ = 0;
  int64_t current_index_ = 0, read_in_shard_ = 0, total_read_ = 0, consumed_ = 0, returned_ = 0, epoch_ = 0;
  int64_t last_pick_ = -1;
  bool filled_ = false;
  std::vector<std::pair<int64_t, int64_t>> buffer_;  // (sequence number, dataset index)
  std::deque<int64_t> shard_ends_;                   // sequence numbers at which an epoch (shard) ends
};

}  // namespace daliamd_host
#endif
