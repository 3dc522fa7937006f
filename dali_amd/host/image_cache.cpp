// Decoded-image cache: see image_cache.h for the reference files this follows.
#include "image_cache.h"

#include <algorithm>
#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "dali_amd_host.h"
#include "framework.h"
#include "host_common.h"

namespace daliamd_host {

// ------------------------------------------------------------------------------------------ policy
ImageCachePolicy::ImageCachePolicy(const std::string &type, size_t cache_size, size_t threshold)
    : largest_(type == "largest"), cache_size_(cache_size), threshold_(type == "largest" ? 0 : threshold) {
  if (type != "threshold" && type != "largest") DALI_FAIL("unexpected cache policy `", type, "`");
  DALI_ENFORCE(threshold_ <= cache_size_, "Cache size should fit at least one image");
}

int64_t ImageCachePolicy::Find(const std::string &key) const {
  auto it = stored_.find(key);
  return it == stored_.end() ? -1 : it->second;
}

void ImageCachePolicy::Erase(const std::string &key) { stored_.erase(key); }  // the space is not reclaimed

// ImageCacheBlob::Add (image_cache_blob.cc:89-115): append at the tail while there is room
int64_t ImageCachePolicy::Store(const std::string &key, size_t data_size, size_t stored_size, size_t threshold) {
  if (data_size < threshold) return -1;
  DALI_ENFORCE(!key.empty(), "internal: cache key is empty");
  if (stored_.count(key)) return -1;
  if (cache_size_ - tail_ < stored_size) {
    full_ = true;
    return -1;
  }
  int64_t at = (int64_t)tail_;
  stored_[key] = at;
  tail_ += stored_size;
  return at;
}

int64_t ImageCachePolicy::OnDecode(const std::string &key, size_t data_size, size_t stored_size) {
  if (!largest_) return Store(key, data_size, stored_size, threshold_);
  // ImageCacheLargest::Add (image_cache_largest.cc:25-89).  Until a key comes round a second time we only rank:
  // `biggest_` holds the set of largest images seen so far whose sizes sum to at most the blob size.
  if (!start_caching_) {
    if (images_.count(key)) {
      // second pass begins: the ranked set becomes the set of images to keep
      start_caching_ = true;
      images_.clear();
      for (; !biggest_.empty(); biggest_.pop()) images_.insert(biggest_.top().second);
    } else {
      images_.insert(key);
      if (biggest_total_ + stored_size <= cache_size_) {
        biggest_.push({stored_size, key});
        biggest_total_ += stored_size;
      } else {
        full_ = true;
        if (stored_size <= cache_size_) {
          // make room by dropping strictly smaller candidates, smallest first; whatever still fits comes back
          std::vector<Candidate> dropped;
          while (!biggest_.empty() && biggest_total_ + stored_size > cache_size_ && biggest_.top().first < stored_size) {
            biggest_total_ -= biggest_.top().first;
            dropped.push_back(biggest_.top());
            biggest_.pop();
          }
          if (biggest_total_ + stored_size <= cache_size_) {
            biggest_.push({stored_size, key});
            biggest_total_ += stored_size;
          }
          for (auto it = dropped.rbegin(); it != dropped.rend(); ++it) {  // largest of the dropped ones first
            if (biggest_total_ + it->first <= cache_size_) {
              biggest_total_ += it->first;
              biggest_.push(*it);
            }
          }
        }
      }
    }
  }
  if (start_caching_ && images_.count(key)) return Store(key, data_size, stored_size, 0);
  return -1;
}

// ------------------------------------------------------------------------------------------ cache
ImageCache::Fence::~Fence() {
  if (event) daliamdEventDestroy(event);
}

namespace {
std::mutex g_factory_mutex;
struct Instance { std::weak_ptr<ImageCache> cache; ImageCache::Params params; };
std::map<int, Instance> g_caches;
}  // namespace

std::shared_ptr<ImageCache> ImageCache::Get(int device_id, const Params &params) {
  std::lock_guard<std::mutex> g(g_factory_mutex);
  auto it = g_caches.find(device_id);
  if (it != g_caches.end()) {
    if (auto c = it->second.cache.lock()) {
      DALI_ENFORCE(it->second.params == params, "Cache for device ", device_id,
                   " was already initialized with other parameters");
      return c;
    }
  }
  std::shared_ptr<ImageCache> c(new ImageCache(params));
  g_caches[device_id] = {c, params};
  return c;
}

std::shared_ptr<ImageCache> ImageCache::Find(int device_id) {
  std::lock_guard<std::mutex> g(g_factory_mutex);
  auto it = g_caches.find(device_id);
  return it == g_caches.end() ? nullptr : it->second.cache.lock();
}

bool ImageCache::IsCached(const std::string &key) const {
  std::lock_guard<std::mutex> g(m_);
  return entries_.count(key) != 0;
}

ImageCache::ImageCache(const Params &params) : params_(params), policy_(params.type, params.size, params.threshold) {
  void *p = nullptr;
  KCHECK(daliamdMalloc(&p, std::max<size_t>(params.size, 256)));
  blob_ = static_cast<uint8_t *>(p);
}

ImageCache::~ImageCache() {
  // every stream that reads or writes the blob belongs to a pipeline holding this cache: they are idle by now
  if (params_.debug && !stats_.empty()) PrintStats();
  entries_.clear();
  pending_.clear();
  if (blob_) daliamdFree(blob_);
}

bool ImageCache::Lookup(const std::string &key, Entry *entry, daliamdStream_t stream) {
  if (key.empty()) return false;
  std::lock_guard<std::mutex> g(m_);
  auto it = entries_.find(key);
  if (it == entries_.end()) return false;
  Entry &e = it->second;
  if (e.fence) {
    if (!e.fence->done) {
      int done = 0;
      KCHECK(daliamdEventQuery(e.fence->event, &done));
      e.fence->done = done != 0;
    }
    if (e.fence->done) e.fence.reset();
    else KCHECK(daliamdStreamWaitEvent(stream, e.fence->event));
  }
  *entry = e;
  if (params_.debug) stats_[key].reads++;
  return true;
}

uint8_t *ImageCache::Reserve(const std::string &key, int h, int w, int c, int64_t pitch) {
  if (key.empty()) return nullptr;
  std::lock_guard<std::mutex> g(m_);
  if (params_.debug) stats_[key].decodes++;
  if (entries_.count(key) || pending_.count(key)) return nullptr;
  const size_t data_size = (size_t)h * w * c;
  const size_t stored = ((size_t)h * pitch + 255) & ~(size_t)255;  // every slot starts at a 256-byte boundary
  int64_t at = policy_.OnDecode(key, data_size, stored);
  if (at < 0) return nullptr;
  Entry e;
  e.data = blob_ + at;
  e.h = h; e.w = w; e.c = c; e.pitch = pitch;
  pending_[key] = e;
  return e.data;
}

void ImageCache::Commit(const std::vector<std::string> &keys, daliamdStream_t stream) {
  if (keys.empty()) return;
  auto fence = std::make_shared<Fence>();
  KCHECK(daliamdEventCreate(&fence->event, 0));
  KCHECK(daliamdEventRecord(fence->event, stream));
  std::lock_guard<std::mutex> g(m_);
  for (auto &k : keys) {
    auto it = pending_.find(k);
    if (it == pending_.end()) continue;
    it->second.fence = fence;
    entries_[k] = it->second;
    pending_.erase(it);
    if (params_.debug) stats_[k].cached = true;
  }
}

void ImageCache::Invalidate(const std::string &key) {
  std::lock_guard<std::mutex> g(m_);
  entries_.erase(key);
  pending_.erase(key);
  policy_.Erase(key);
  if (params_.debug) stats_[key].cached = false;
}

// same report as ImageCacheBlob::print_stats (image_cache_blob.cc:129-160); DALI_LOG_FILE redirects it
void ImageCache::PrintStats() const {
  static std::mutex stats_mutex;
  std::lock_guard<std::mutex> g(stats_mutex);
  size_t cached = 0;
  for (auto &kv : stats_) cached += kv.second.cached;
  const char *log_filename = std::getenv("DALI_LOG_FILE");
  std::ofstream log_file;
  if (log_filename) log_file.open(log_filename);
  std::ostream &out = log_filename ? log_file : std::cout;
  out << "#################### CACHE STATS ####################" << std::endl;
  out << "cache_size: " << policy_.cache_size() << std::endl;
  out << "cache_threshold: " << policy_.threshold() << std::endl;
  out << "is_cache_full: " << (int)policy_.is_full() << std::endl;
  out << "images_seen: " << stats_.size() << std::endl;
  out << "images_cached: " << cached << std::endl;
  out << "images_not_cached: " << stats_.size() - cached << std::endl;
  for (auto &kv : stats_) {
    out << "image[" << kv.first << "] : is_cached[" << (int)kv.second.cached << "] decodes[" << kv.second.decodes
        << "] reads[" << kv.second.reads << "]";
    auto it = entries_.find(kv.first);
    if (it != entries_.end()) out << " shape[" << it->second.h << ", " << it->second.w << ", " << it->second.c << "]";
    out << std::endl;
  }
  out << "#################### END   STATS ####################" << std::endl;
}

// ------------------------------------------------------------------------------------------ encoded streams
namespace {
struct StreamInstance { std::weak_ptr<StreamCache> cache; size_t bytes; };
std::map<int, StreamInstance> g_stream_caches;
}  // namespace

std::shared_ptr<StreamCache> StreamCache::Get(int device_id, size_t bytes, bool debug) {
  std::lock_guard<std::mutex> g(g_factory_mutex);
  auto it = g_stream_caches.find(device_id);
  if (it != g_stream_caches.end()) {
    if (auto c = it->second.cache.lock()) {
      DALI_ENFORCE(it->second.bytes == bytes, "Encoded-stream cache for device ", device_id,
                   " was already initialized with another size");
      return c;
    }
  }
  std::shared_ptr<StreamCache> c(new StreamCache(bytes, debug));
  g_stream_caches[device_id] = {c, bytes};
  return c;
}

std::shared_ptr<StreamCache> StreamCache::Find(int device_id) {
  std::lock_guard<std::mutex> g(g_factory_mutex);
  auto it = g_stream_caches.find(device_id);
  return it == g_stream_caches.end() ? nullptr : it->second.cache.lock();
}

StreamCache::StreamCache(size_t bytes, bool debug) : size_(bytes), debug_(debug) {
  void *p = nullptr;
  KCHECK(daliamdMalloc(&p, std::max<size_t>(bytes, 256)));
  blob_ = static_cast<uint8_t *>(p);
}

StreamCache::~StreamCache() {
  if (debug_)
    fprintf(stderr, "[dali_amd] encoded-stream cache: %zu streams, %.1f of %.1f MB used%s, %lld hits, %lld misses\n",
            entries_.size(), tail_ / 1048576.0, size_ / 1048576.0, full_ ? " (full)" : "", (long long)hits_,
            (long long)misses_);
  entries_.clear();
  if (blob_) daliamdFree(blob_);
}

bool StreamCache::IsCached(const std::string &key) const {
  std::lock_guard<std::mutex> g(m_);
  return entries_.count(key) != 0;
}

int StreamCache::Lookup(const std::vector<std::string> &keys, const std::vector<uint8_t> &skip,
                        std::vector<std::shared_ptr<const Record>> *out, daliamdStream_t stream, daliamdStream_t also) {
  int found = 0;
  const ImageCache::Fence *waited = nullptr;  // a batch is usually behind ONE fence: wait for it once
  std::lock_guard<std::mutex> g(m_);
  for (size_t i = 0; i < out->size(); i++) {
    (*out)[i].reset();
    if (i >= keys.size() || keys[i].empty() || (i < skip.size() && skip[i])) continue;
    auto it = entries_.find(keys[i]);
    if (it == entries_.end()) { misses_++; continue; }
    Slot &s = it->second;
    if (s.fence && s.fence.get() != waited) {
      if (!s.fence->done) {
        int done = 0;
        KCHECK(daliamdEventQuery(s.fence->event, &done));
        s.fence->done = done != 0;
      }
      if (!s.fence->done) {
        KCHECK(daliamdStreamWaitEvent(stream, s.fence->event));
        if (also && also != stream) KCHECK(daliamdStreamWaitEvent(also, s.fence->event));
        waited = s.fence.get();
      }
    }
    if (s.fence && s.fence->done) s.fence.reset();
    (*out)[i] = s.rec;
    found++;
  }
  hits_ += found;
  return found;
}

uint8_t *StreamCache::Reserve(const std::string &key, size_t bytes) {
  if (key.empty() || full_) return nullptr;
  // (segments start at 64-byte multiples and own 64 bytes of slack behind their last byte: the kernels read whole
  // 16-byte chunks)
  const size_t stored = ((bytes + 64 + 63) / 64) * 64;
  std::lock_guard<std::mutex> g(m_);
  if (entries_.count(key) || pending_.count(key)) return nullptr;
  if (size_ - tail_ < stored) {
    full_ = true;
    return nullptr;
  }
  uint8_t *at = blob_ + tail_;
  tail_ += stored;
  pending_[key] = {at, stored};
  return at;
}

void StreamCache::Commit(const std::vector<std::string> &keys, const std::vector<const daliamdJpegInfo *> &infos,
                         const std::vector<const daliamdJpegScan *> &scans, daliamdStream_t stream,
                         const std::vector<uint8_t> &indexed, int device_id) {
  if (keys.empty()) return;
  auto fence = std::make_shared<ImageCache::Fence>();
  KCHECK(daliamdEventCreate(&fence->event, 0));
  KCHECK(daliamdEventRecord(fence->event, stream));
  // (records are built outside the lock: 3 KB of tables each)
  std::vector<std::shared_ptr<Record>> recs(keys.size());
  for (size_t k = 0; k < keys.size(); k++) {
    recs[k] = std::make_shared<Record>();
    recs[k]->info = *infos[k];
    recs[k]->scan = *scans[k];
    if (device_id >= 0) recs[k]->tables = HuffTableStore::Get(device_id, *scans[k]);
  }
  std::lock_guard<std::mutex> g(m_);
  for (size_t k = 0; k < keys.size(); k++) {
    auto it = pending_.find(keys[k]);
    if (it == pending_.end()) continue;
    if (k < indexed.size() && indexed[k]) recs[k]->index = it->second.first;
    else recs[k]->ecs = it->second.first;
    entries_[keys[k]] = Slot{recs[k], fence};
    pending_.erase(it);
  }
}

void StreamCache::CommitRaster(const std::string &key, int h, int w, int c, int64_t pitch, int image_type, daliamdStream_t stream) {
  auto fence = std::make_shared<ImageCache::Fence>();
  KCHECK(daliamdEventCreate(&fence->event, 0));
  KCHECK(daliamdEventRecord(fence->event, stream));
  auto rec = std::make_shared<Record>();
  rec->info = daliamdJpegInfo{};
  rec->scan = daliamdJpegScan{};
  rec->h = h; rec->w = w; rec->c = c; rec->pitch = pitch; rec->image_type = image_type;
  std::lock_guard<std::mutex> g(m_);
  auto it = pending_.find(key);
  if (it == pending_.end()) return;
  rec->pixels = it->second.first;
  entries_[key] = Slot{rec, fence};
  pending_.erase(it);
}

void StreamCache::Invalidate(const std::string &key) {
  std::lock_guard<std::mutex> g(m_);
  entries_.erase(key);   // (the space is not reclaimed)
  pending_.erase(key);
}

void StreamCache::Abandon(const std::vector<std::string> &keys) {
  std::lock_guard<std::mutex> g(m_);
  for (auto k = keys.rbegin(); k != keys.rend(); ++k) {   // latest reservation first: the blob is a stack
    auto it = pending_.find(*k);
    if (it == pending_.end()) continue;
    if (it->second.first + it->second.second == blob_ + tail_) {
      tail_ -= it->second.second;
      full_ = false;
    }
    pending_.erase(it);
  }
}

// ------------------------------------------------------------------------------------------ code tables
bool HuffTableStore::SameTables(const daliamdJpegScan &a, const daliamdJpegScan &b) {
  return a.blocks_per_mcu == b.blocks_per_mcu && !memcmp(a.comp_of_block, b.comp_of_block, sizeof(a.comp_of_block)) &&
         !memcmp(a.dc_sel, b.dc_sel, sizeof(a.dc_sel)) && !memcmp(a.ac_sel, b.ac_sel, sizeof(a.ac_sel)) &&
         !memcmp(a.dc_bits, b.dc_bits, 2 * sizeof(a.dc_bits[0])) && !memcmp(a.ac_bits, b.ac_bits, 2 * sizeof(a.ac_bits[0])) &&
         !memcmp(a.dc_vals, b.dc_vals, 2 * sizeof(a.dc_vals[0])) && !memcmp(a.ac_vals, b.ac_vals, 2 * sizeof(a.ac_vals[0]));
}

const uint8_t *HuffTableStore::Get(int device_id, const daliamdJpegScan &scan) {
  static std::mutex m;
  static std::vector<Set> sets;
  static const bool off = getenv("DALI_AMD_NO_HOST_TABLES") && atoi(getenv("DALI_AMD_NO_HOST_TABLES"));   // (read once: ADVICE r05)
  if (off) return nullptr;
  std::lock_guard<std::mutex> g(m);
  Set *hit = nullptr;
  for (auto &s : sets)
    if (s.device_id == device_id && SameTables(s.scan, scan)) { hit = &s; break; }
  if (!hit) {
    if (sets.size() >= kMaxSets) {
      // full: the sets seen only once make room (streams with tables of their own), the built ones stay
      auto it = std::find_if(sets.begin(), sets.end(), [](const Set &s) { return !s.tables; });
      if (it == sets.end()) return nullptr;
      sets.erase(it);
    }
    sets.push_back(Set{scan, device_id, 1, nullptr});
    return nullptr;
  }
  hit->seen++;
  if (hit->tables || hit->seen < 2) return hit->tables;
  // second sighting: build on the host, upload, wait (once per set: a 59 KB copy)
  daliamdJpegHuffDesc d{};
  d.blocks_per_mcu = scan.blocks_per_mcu;
  memcpy(d.comp_of_block, scan.comp_of_block, 10);
  memcpy(d.dc_sel, scan.dc_sel, 4);
  memcpy(d.ac_sel, scan.ac_sel, 4);
  for (int t = 0; t < 2; t++) {
    memcpy(d.bits[t], scan.dc_bits[t], 16);
    memcpy(d.bits[2 + t], scan.ac_bits[t], 16);
    memcpy(d.vals[t], scan.dc_vals[t], 256);
    memcpy(d.vals[2 + t], scan.ac_vals[t], 256);
  }
  size_t bytes = 0;
  if (daliamdJpegHuffmanTablesBytes(&bytes) != DALIAMD_SUCCESS) return nullptr;
  std::vector<uint8_t> host(bytes);
  if (daliamdJpegHuffmanTablesBuild(&d, host.data()) != DALIAMD_SUCCESS) return nullptr;   // (the launch will say why)
  void *dev = nullptr;
  if (daliamdMalloc(&dev, bytes) != DALIAMD_SUCCESS) return nullptr;
  // (on a stream of its own, never the legacy NULL stream: in a training process that one is the framework's default stream,
  // and waiting for it means waiting for everything the training loop has queued - ADVICE r05)
  // (one per device: the caller - a pipeline's device-stage thread - has its device current; a stream belongs to the device
  // it was made on)
  static daliamdStream_t upload_streams[64] = {};
  daliamdStream_t &upload_stream = upload_streams[(unsigned)device_id % 64];
  if (!upload_stream) daliamdStreamCreate(&upload_stream, 1);
  if (!upload_stream || daliamdMemcpyH2DAsync(dev, host.data(), bytes, upload_stream) != DALIAMD_SUCCESS ||
      daliamdStreamSynchronize(upload_stream) != DALIAMD_SUCCESS) {
    daliamdFree(dev);
    return nullptr;
  }
  hit->tables = static_cast<const uint8_t *>(dev);
  return hit->tables;
}

// ------------------------------------------------------------------------------------------ parse results
namespace {
struct HeaderEntry {
  int64_t size;
  uint64_t header_hash;   // of the bytes in front of the entropy-coded segment (the headers the entry describes)
  daliamdJpegInfo info;
  int32_t eligible, mcus_x, mcus_y, length_is_upper_bound;
  int64_t ecs_offset, ecs_length;
  std::shared_ptr<const daliamdJpegScan> common;   // everything else of the scan analysis, shared between files
};
// The key - source path + stream size - does not change when a file is rewritten in place with another image of the same
// size (tests and jobs that regenerate a data set in one process); the entry would then describe headers that
// no longer exist: stale dimensions, tables, segment offset.  A hit therefore also needs the same HEADER BYTES (a 64-bit
// hash of everything in front of the segment, at most 4 KB: a few hundred bytes, 0.1 us).
uint64_t HeaderHash(const uint8_t *data, size_t size, int64_t ecs_offset) {
  size_t n = (size_t)std::max<int64_t>(0, std::min<int64_t>(ecs_offset, 4096));
  n = std::min(n, size);
  uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, data + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  for (; i < n; i++) h = (h ^ data[i]) * 0x100000001b3ull;
  return h;
}
constexpr int kHeaderShards = 32;
struct HeaderShard {
  std::mutex m;
  std::unordered_map<std::string, HeaderEntry> map;
};
HeaderShard g_header_shards[kHeaderShards];
std::mutex g_header_common_m;
std::vector<std::shared_ptr<const daliamdJpegScan>> g_header_common;
std::atomic<int64_t> g_header_entries{0};
int64_t HeaderCacheCap() {
  static const int64_t cap = [] { const char *e = getenv("DALI_AMD_HEADER_CACHE_ENTRIES"); return e ? atoll(e) : (int64_t)2000000; }();
  return cap;
}
daliamdJpegScan CommonPart(const daliamdJpegScan &s) {
  daliamdJpegScan c = s;
  c.eligible = 0; c.mcus_x = c.mcus_y = 0; c.ecs_offset = c.ecs_length = 0; c.length_is_upper_bound = 0;
  return c;
}
}  // namespace

bool HeaderCache::Find(const std::string &key, const uint8_t *data, int64_t stream_size, daliamdJpegInfo *info, daliamdJpegScan *scan) {
  if (key.empty() || !data || HeaderCacheCap() <= 0) return false;
  HeaderShard &sh = g_header_shards[std::hash<std::string>()(key) % kHeaderShards];
  std::lock_guard<std::mutex> g(sh.m);
  auto it = sh.map.find(key);
  if (it == sh.map.end() || it->second.size != stream_size) return false;
  const HeaderEntry &e = it->second;
  if (e.ecs_offset > stream_size || HeaderHash(data, (size_t)stream_size, e.ecs_offset) != e.header_hash) {
    sh.map.erase(it);   // another file under the same name and size: parsed anew (and stored again) by the caller
    g_header_entries.fetch_sub(1, std::memory_order_relaxed);
    return false;
  }
  *info = e.info;
  *scan = *e.common;
  scan->eligible = e.eligible; scan->mcus_x = e.mcus_x; scan->mcus_y = e.mcus_y;
  scan->ecs_offset = e.ecs_offset; scan->ecs_length = e.ecs_length; scan->length_is_upper_bound = e.length_is_upper_bound;
  return true;
}

void HeaderCache::Invalidate(const std::string &key) {
  if (key.empty()) return;
  HeaderShard &sh = g_header_shards[std::hash<std::string>()(key) % kHeaderShards];
  std::lock_guard<std::mutex> g(sh.m);
  if (sh.map.erase(key)) g_header_entries.fetch_sub(1, std::memory_order_relaxed);
}

void HeaderCache::Put(const std::string &key, const uint8_t *data, int64_t stream_size, const daliamdJpegInfo &info,
                      const daliamdJpegScan &scan) {
  if (key.empty() || !data || g_header_entries.load(std::memory_order_relaxed) >= HeaderCacheCap()) return;
  const daliamdJpegScan common = CommonPart(scan);
  std::shared_ptr<const daliamdJpegScan> shared;
  {
    std::lock_guard<std::mutex> g(g_header_common_m);
    for (auto &c : g_header_common)
      if (!memcmp(c.get(), &common, sizeof(common))) { shared = c; break; }
    if (!shared) {
      if (g_header_common.size() >= 4096) return;   // (a data set of files with tables of their own: not worth keeping)
      shared = std::make_shared<const daliamdJpegScan>(common);
      g_header_common.push_back(shared);
    }
  }
  HeaderShard &sh = g_header_shards[std::hash<std::string>()(key) % kHeaderShards];
  std::lock_guard<std::mutex> g(sh.m);
  auto ins = sh.map.emplace(key, HeaderEntry{});
  if (ins.second) g_header_entries.fetch_add(1, std::memory_order_relaxed);
  ins.first->second = HeaderEntry{stream_size, HeaderHash(data, (size_t)stream_size, scan.ecs_offset), info, scan.eligible, scan.mcus_x, scan.mcus_y, scan.length_is_upper_bound,
                                  scan.ecs_offset, scan.ecs_length, shared};
}

bool DecoderCacheHolds(int device_id, const std::string &key) {
  if (auto c = ImageCache::Find(device_id))
    if (c->IsCached(key)) return true;
  if (auto c = StreamCache::Find(device_id))
    if (c->IsCached(key)) return true;
  return false;
}

}  // namespace daliamd_host

// ------------------------------------------------------------------------------------------ C ABI (bookkeeping only)
extern "C" {
void *daliamdImageCachePolicyCreate(const char *type, uint64_t cache_size, uint64_t threshold) {
  try {
    return new daliamd_host::ImageCachePolicy(type ? type : "", (size_t)cache_size, (size_t)threshold);
  } catch (const std::exception &e) {
    daliamd_host::Fail("%s", e.what());
    return nullptr;
  }
}
void daliamdImageCachePolicyDestroy(void *policy) { delete static_cast<daliamd_host::ImageCachePolicy *>(policy); }
int64_t daliamdImageCachePolicyOnDecode(void *policy, const char *key, uint64_t data_size, uint64_t stored_size) {
  try {
    return static_cast<daliamd_host::ImageCachePolicy *>(policy)->OnDecode(key, (size_t)data_size, (size_t)stored_size);
  } catch (const std::exception &e) {
    daliamd_host::Fail("%s", e.what());
    return -1;
  }
}
int64_t daliamdImageCachePolicyFind(void *policy, const char *key) {
  return static_cast<daliamd_host::ImageCachePolicy *>(policy)->Find(key);
}
}
