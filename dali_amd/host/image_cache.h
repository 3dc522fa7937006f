// Decoded-image cache of the mixed image decoder (`cache_size`, `cache_type`, `cache_threshold`, `cache_debug`).
//
// Behaviour follows the reference's decoder cache:
//   CachedDecoderImpl / CachedDecoderAttr   dali/operators/decoder/cache/cached_decoder_impl.cc:24-135
//   ImageCacheBlob  ("threshold")          dali/operators/decoder/cache/image_cache_blob.cc:27-135
//   ImageCacheLargest ("largest")          dali/operators/decoder/cache/image_cache_largest.cc:25-89
//   ImageCacheFactory (one per device)     dali/operators/decoder/cache/image_cache_factory.cc:22-70
//
// MI355X design: the blob is one HBM allocation (a shard of decoded ImageNet fits in 288 GB) and there is NO copy on
// either side of it.  On a miss the decoder's colour kernel writes the RGB image straight into its cache slot; on a
// hit the output TensorList of the decoder points at the slot (TensorList::Resize with external samples).  The
// bookkeeping (which key goes where) is host-only and split out as ImageCachePolicy so that it runs without a GPU.
#ifndef DALI_AMD_HOST_IMAGE_CACHE_H_
#define DALI_AMD_HOST_IMAGE_CACHE_H_

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "dali_amd_host.h"
#include "dali_amd_kernels.h"

namespace daliamd_host {

// Host-only bookkeeping: decides, per decode of `key`, whether (and at which offset of the blob) the image is kept.
class ImageCachePolicy {
 public:
  // type: "threshold" (keep every image of at least `threshold` bytes until the blob is full) or "largest" (first
  // pass over the data: find the largest images that fit together; they are stored from the second pass on)
  ImageCachePolicy(const std::string &type, size_t cache_size, size_t threshold);
  // One decode of `key`: `data_size` = H*W*C (compared with the threshold, image_cache_blob.cc:95), `stored_size` =
  // bytes the image takes in the blob.  Returns the blob offset it must be written to, or -1 when it is not kept.
  int64_t OnDecode(const std::string &key, size_t data_size, size_t stored_size);
  // offset of a stored image or -1
  int64_t Find(const std::string &key) const;
  void Erase(const std::string &key);
  size_t bytes_used() const { return tail_; }
  size_t cache_size() const { return cache_size_; }
  size_t threshold() const { return threshold_; }
  bool is_full() const { return full_; }

 private:
  int64_t Store(const std::string &key, size_t data_size, size_t stored_size, size_t threshold);
  bool largest_;
  size_t cache_size_, threshold_, tail_ = 0;
  bool full_ = false;
  std::unordered_map<std::string, int64_t> stored_;
  // "largest": candidates of the first pass, smallest on top
  using Candidate = std::pair<size_t, std::string>;
  std::priority_queue<Candidate, std::vector<Candidate>, std::greater<Candidate>> biggest_;
  size_t biggest_total_ = 0;
  std::set<std::string> images_;
  bool start_caching_ = false;
};

class ImageCache {
 public:
  // all work that wrote a group of entries; readers on other streams wait for it
  struct Fence {
    ~Fence();
    daliamdEvent_t event = nullptr;
    bool done = false;
  };
  struct Entry {
    uint8_t *data = nullptr;
    int32_t h = 0, w = 0, c = 0;
    int64_t pitch = 0;
    std::shared_ptr<Fence> fence;  // null once the write is known to have finished
  };
  struct Params {
    std::string type;
    size_t size, threshold;
    bool debug;
    bool operator==(const Params &o) const { return type == o.type && size == o.size && threshold == o.threshold && debug == o.debug; }
  };

  // the cache of `device_id`, shared by every decoder instance on that device (all must ask for the same parameters)
  static std::shared_ptr<ImageCache> Get(int device_id, const Params &params);
  // the live cache of the device, if any operator holds one (readers: `skip_cached_images`, loader.h:466-480)
  static std::shared_ptr<ImageCache> Find(int device_id);
  bool IsCached(const std::string &key) const;
  ~ImageCache();

  // A committed entry.  Makes `stream` wait for the entry's write when that may still be in flight.
  bool Lookup(const std::string &key, Entry *entry, daliamdStream_t stream);
  // A decode of `key` is about to be enqueued: returns the slot it should write to (rows of `pitch` bytes), or nullptr
  // when the policy does not keep it.  The entry stays invisible to Lookup until Commit.
  uint8_t *Reserve(const std::string &key, int h, int w, int c, int64_t pitch);
  // The decodes reserved since the last Commit by this caller are enqueued on `stream`.
  void Commit(const std::vector<std::string> &keys, daliamdStream_t stream);
  // The decode behind a reserved/committed entry failed
  void Invalidate(const std::string &key);

 private:
  explicit ImageCache(const Params &params);
  struct Stats { int64_t decodes = 0, reads = 0; bool cached = false; };
  void PrintStats() const;
  Params params_;
  ImageCachePolicy policy_;
  uint8_t *blob_ = nullptr;
  mutable std::mutex m_;
  std::unordered_map<std::string, Entry> entries_, pending_;
  std::map<std::string, Stats> stats_;
};

// Encoded-stream cache: `cache_type="encoded"` - an MI355X extension of the reference's decoder cache.
//
// What stays resident is not the decoded image but everything the GPU entropy decoder needs to decode it again: the
// entropy-coded segment of the JPEG (in one HBM blob) and the parse results (header fields, scan analysis with the
// Huffman / quantisation tables; host memory).  A whole ImageNet shard's JPEG bytes fit in a fraction of the 288 GB
// of one MI355X (the full training set is ~140 GB), seven times less than its decoded pixels.  From the second epoch
// on a sample costs no file read (readers: `skip_cached_images`), no header parse, no staging copy and no
// host->device transfer - the decode itself still runs, so region-of-interest decoding and everything behind the
// decoder see fresh work every epoch.  Only streams the GPU entropy decoder takes are kept.
// Same rules as the decoded cache: one per device, fill until full, nothing is evicted, entries become visible when
// the copy that fills them has been enqueued (readers on other streams wait for its event).
class StreamCache {
 public:
  struct Record {
    const uint8_t *ecs = nullptr;  // the entropy-coded segment in the blob (device), scan.ecs_length bytes
    // `cache_type="indexed"` (round 5): what is resident is the stream's index entry instead (daliamdJpegHuffDesc.index:
    // the un-stuffed stream + 12 bytes of decoder state per 256-byte slice, written by the decode that made it resident);
    // `ecs` is null then
    const uint8_t *index = nullptr;
    const uint8_t *tables = nullptr;   // HuffTableStore: the stream's finished code tables, when its set is kept
    // Round 6, "raster resident": a sample the device cannot decode at all (CMYK / YCCK JPEG, PNG, BMP, PNM - host pixel
    // decoders, 4-7 ms each, on the critical path of the batch they sit in) is kept as its DECODED upright image: h x w x c
    // bytes, rows `pitch` apart, in the decoder's output_type `image_type`.  `ecs` and `index` are null then.
    const uint8_t *pixels = nullptr;
    int32_t h = 0, w = 0, c = 0, image_type = 0;
    int64_t pitch = 0;
    daliamdJpegInfo info;
    daliamdJpegScan scan;
  };
  static std::shared_ptr<StreamCache> Get(int device_id, size_t bytes, bool debug);
  static std::shared_ptr<StreamCache> Find(int device_id);
  ~StreamCache();

  bool IsCached(const std::string &key) const;
  // out[i] = the record of keys[i] or null; samples with skip[i] != 0 are not looked up.  Makes `stream` wait for the
  // copies behind the returned records where they may still be in flight.  One lock for the whole batch.
  // `also`: a second stream that will read the records' device data (the decoder's side stream), made to wait the same way
  int Lookup(const std::vector<std::string> &keys, const std::vector<uint8_t> &skip,
             std::vector<std::shared_ptr<const Record>> *out, daliamdStream_t stream, daliamdStream_t also = nullptr);
  // room for the `bytes` of a new segment, or nullptr (known key, or the blob is full)
  uint8_t *Reserve(const std::string &key, size_t bytes);
  // the copies into the slots reserved for `keys` are enqueued on `stream`
  // indexed[k] != 0: the slot of keys[k] holds an index entry (built by the decode enqueued on `stream`), not the raw segment
  void Commit(const std::vector<std::string> &keys, const std::vector<const daliamdJpegInfo *> &infos,
              const std::vector<const daliamdJpegScan *> &scans, daliamdStream_t stream,
              const std::vector<uint8_t> &indexed = {}, int device_id = -1);
  // the slot reserved for `key` holds a decoded image (Record::pixels) once `stream` has passed this point
  void CommitRaster(const std::string &key, int h, int w, int c, int64_t pitch, int image_type, daliamdStream_t stream);
  void Invalidate(const std::string &key);
  // Reservations that will never be committed (an exception between Reserve and Commit): the keys become reservable
  // again and the space of those that still sit at the end of the blob is handed back.
  void Abandon(const std::vector<std::string> &keys);
  size_t bytes_used() const { return tail_; }
  void Stats(int64_t *out4) const {
    std::lock_guard<std::mutex> g(m_);
    out4[0] = (int64_t)entries_.size(); out4[1] = (int64_t)tail_; out4[2] = hits_; out4[3] = misses_;
  }

 private:
  StreamCache(size_t bytes, bool debug);
  struct Slot {
    std::shared_ptr<const Record> rec;
    std::shared_ptr<ImageCache::Fence> fence;  // null once the copy is known to have finished
  };
  size_t size_, tail_ = 0;
  bool debug_, full_ = false;
  uint8_t *blob_ = nullptr;
  mutable std::mutex m_;
  std::unordered_map<std::string, Slot> entries_;
  std::unordered_map<std::string, std::pair<uint8_t *, size_t>> pending_;   // slot, bytes it occupies in the blob
  int64_t hits_ = 0, misses_ = 0;
};

// Finished code tables of the GPU entropy decoder, per device and distinct table set (daliamdJpegHuffDesc.tables, round 5).
// The tables depend on the DHT contents and the MCU structure of a stream and on nothing else, and nearly every file of a
// data set carries the same ones (T.81 Annex K): they are built ONCE on the host (daliamdJpegHuffmanTablesBuild: the
// functions the device runs), uploaded, and every later launch reads them - no table-building workgroups, and for a batch
// of indexed resident streams no first kernel at all.  A set is built when it is seen the SECOND time (a file with
// optimised tables of its own is seen once per epoch), at most kMaxSets sets are kept; anything else gets nullptr and has
// its tables built inside the launch as before.  Process-wide, never freed (a few times 59 KB).
class HuffTableStore {
 public:
  static const uint8_t *Get(int device_id, const daliamdJpegScan &scan);

 private:
  static constexpr size_t kMaxSets = 32;
  struct Set {
    daliamdJpegScan scan;   // (the fields that decide the tables)
    int device_id;
    int seen;
    const uint8_t *tables;  // device; null until built
  };
  static bool SameTables(const daliamdJpegScan &a, const daliamdJpegScan &b);
};

// Parse results of files the decoder has seen before (round 5): epoch >= 2 of a shard that is NOT resident still reads its
// files, but their headers say what they said in epoch 1 - the reference keeps such epoch-invariant facts in index files
// (tools/tfrecord2idx, tools/wds2idx.py).  Keyed by the sample's source (the file path), valid while the size of the
// encoded stream is the one recorded; a hit spares the header parse and scan analysis (4 us of host time per image, a third of a
// decoder thread's work when eight ranks share a host).  The table-heavy part of a scan analysis (DHT / DQT contents,
// 2.9 KB) is shared between the files that have the same: an entry costs about 250 bytes.  Process-wide, at most
// DALI_AMD_HEADER_CACHE_ENTRIES entries (default 2 M; 0 switches it off), filled until full.
class HeaderCache {
 public:
  // `data`: the stream as it is now - an entry only answers for the header bytes it was made from (a file rewritten in place
  // under the same name and size is parsed anew)
  static bool Find(const std::string &key, const uint8_t *data, int64_t stream_size, daliamdJpegInfo *info, daliamdJpegScan *scan);
  static void Put(const std::string &key, const uint8_t *data, int64_t stream_size, const daliamdJpegInfo &info, const daliamdJpegScan &scan);
  static void Invalidate(const std::string &key);   // (a decode of the stream failed: whatever the entry says is suspect)
};

// `skip_cached_images` of the readers: is the sample held by a decoder cache (of either kind) of the device?
bool DecoderCacheHolds(int device_id, const std::string &key);

}  // namespace daliamd_host

#endif  // DALI_AMD_HOST_IMAGE_CACHE_H_
