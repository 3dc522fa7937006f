// Source operators: ExternalSource, readers.file (with the reference's shard / shuffle / padding
// semantics) and the CPU -> GPU copy inserted for `.gpu()`.
//   FileReader / FileLabelLoader   dali/operators/reader/file_reader_op.{h,cc}, loader/file_label_loader.cc:34-92
//   Loader (shards, shuffle)       dali/operators/reader/loader/loader.h:78-503, loader.cc:78-87
//   discovery (sorted dirs/files)  dali/operators/reader/loader/discover_files.cc:38-143
//   ExternalSource                 dali/pipeline/operator/builtin/external_source.{h,cc}
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <dirent.h>
#include <fcntl.h>
#include <fnmatch.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <random>

#include "image_cache.h"
#include "ops.h"
#include "pipeline.h"

namespace daliamd_host {

// =============================================================================================
// ExternalSource
// =============================================================================================
DALI_SCHEMA(ExternalSource)
    .DocStr("Allows externally provided data to be passed as an input to the pipeline "
            "(fed with `Pipeline.feed_input`).")
    .NumInput(0)
    .NumOutput(1)
    .AddOptionalArg("blocking", "Whether to block when no data is available (ignored; an empty queue is an error).",
                    ArgValue::Bool(false))
    .AddOptionalArg("no_copy", "Ignored: fed data is always copied into pipeline-owned memory.", ArgValue::Bool(false))
    .AddOptionalTypeArg("layout", "Layout of the fed data.", ArgType::STRING)
    .AddOptionalTypeArg("dtype", "Expected data type (checked when set).", ArgType::INT)
    .AddOptionalTypeArg("ndim", "Expected number of dimensions (checked when set).", ArgType::INT);

class ExternalSourceOp : public OperatorBase {
 public:
  explicit ExternalSourceOp(const OpSpec &spec) : OperatorBase(spec) {}

  void Feed(const std::vector<const void *> &data, const std::vector<TensorShape> &shapes, DALIDataType type,
            const std::string &layout) {
    DALI_ENFORCE(data.size() == shapes.size(), "ExternalSource: data / shape count mismatch");
    DALI_ENFORCE((int)data.size() <= max_batch_size_, "ExternalSource expects a batch of at most ", max_batch_size_,
                 " samples, got ", data.size());
    if (const ArgValue *d = spec_.TryArg("dtype"))
      DALI_ENFORCE(d->i == (int)type, "ExternalSource expected data of type ", TypeName((DALIDataType)d->i), " and got ",
                   TypeName(type));
    if (const ArgValue *nd = spec_.TryArg("ndim"))
      for (auto &s : shapes) DALI_ENFORCE((int64_t)s.size() == nd->i, "ExternalSource expected ", nd->i, "-D data");
    // Storage is recycled: a page-locked block that no ring slot refers to any more takes the next batch.  Allocating
    // (and, worse, FREEING: hipHostFree waits for the device to go idle) a pinned block per fed batch stalled the
    // host stage for about a millisecond per iteration with several batches in flight.
    std::shared_ptr<TensorList> tl;
    {
      std::lock_guard<std::mutex> g(m_);
      for (auto it = recycle_.begin(); it != recycle_.end(); ++it)
        if ((*it)->buffer_use_count() == 1) {
          tl = std::move(*it);
          recycle_.erase(it);
          break;
        }
    }
    if (!tl) tl = std::make_shared<TensorList>(StorageDevice::CPU);
    tl->Resize(shapes, type);
    tl->SetLayout(layout);
    for (size_t i = 0; i < data.size(); i++) memcpy(tl->raw((int)i), data[i], tl->nbytes((int)i));
    std::lock_guard<std::mutex> g(m_);
    queue_.push_back(std::move(tl));
  }

  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }
  void RunImpl(Workspace &ws) override {
    std::shared_ptr<TensorList> tl;
    {
      std::lock_guard<std::mutex> g(m_);
      DALI_ENFORCE(!queue_.empty(), "No data was provided to the ExternalSource. Make sure to feed it properly "
                   "(call feed_input before each run).");
      tl = std::move(queue_.front());
      queue_.pop_front();
    }
    ws.Output(0).ShareData(*tl);
    {
      std::lock_guard<std::mutex> g(m_);
      if (recycle_.size() < 32) recycle_.push_back(std::move(tl));   // (its block is free again once the ring slot moves on)
    }
    if (const ArgValue *l = spec_.TryArg("layout"))
      if (ws.Output(0).layout().empty()) ws.Output(0).SetLayout(l->s);
  }

 private:
  std::mutex m_;
  std::deque<std::shared_ptr<TensorList>> queue_;
  std::vector<std::shared_ptr<TensorList>> recycle_;
};
DALI_REGISTER_OPERATOR(ExternalSource, ExternalSourceOp, CPU);

void FeedExternalSource(OperatorBase *op, const std::vector<const void *> &data, const std::vector<TensorShape> &shapes,
                        DALIDataType type, const std::string &layout) {
  auto *es = dynamic_cast<ExternalSourceOp *>(op);
  DALI_ENFORCE(es, "The operator is not an ExternalSource");
  es->Feed(data, shapes, type, layout);
}

// =============================================================================================
// readers.file
// =============================================================================================
DALI_SCHEMA(LoaderBase)
    .DocStr("Common reader arguments.")
    .MakeInternal()
    .AddOptionalArg("random_shuffle", "Determines whether to randomly shuffle data. A prefetch buffer with a size equal "
                    "to ``initial_fill`` is used to read data sequentially, and then samples are selected randomly to "
                    "form a batch.", ArgValue::Bool(false))
    .AddOptionalArg("initial_fill", "Size of the buffer that is used for shuffling.", ArgValue::Int(1024))
    .AddOptionalArg("num_shards", "Partitions the data into the specified number of parts (shards).", ArgValue::Int(1))
    .AddOptionalArg("shard_id", "Index of the shard to read.", ArgValue::Int(0))
    .AddOptionalArg("tensor_init_bytes", "Hint for how much memory to allocate per image.", ArgValue::Int(1048576))
    .AddOptionalArg("stick_to_shard", "Determines whether the reader should stick to a data shard instead of going "
                    "through the entire dataset.", ArgValue::Bool(false))
    .AddOptionalArg("read_ahead", "Determines whether the accessed data should be read ahead (readers.file: the kernel is "
                    "asked to fetch the whole file when it is opened).", ArgValue::Bool(false))
    .AddOptionalArg("prefetch_queue_depth", "Number of batches prefetched by the internal loader (readers.file: its "
                    "reader threads run this many batches ahead of the executor).", ArgValue::Int(1))
    .AddOptionalArg("skip_cached_images", "If set to True, the loading data will be skipped when the sample is in the "
                    "decoder cache. In this case, the output of the loader will be empty.", ArgValue::Bool(false))
    .AddOptionalArg("lazy_init", "Parse and prepare the dataset metadata only during the first run.", ArgValue::Bool(false))
    .AddOptionalArg("pad_last_batch", "If set to True, pads the shard by repeating the last sample.", ArgValue::Bool(false))
    .AddOptionalArg("dont_use_mmap", "If set to True, the loader uses plain file I/O instead of mapping the files in "
                    "memory.  (readers.file: mapped files stay mapped - all readers of the process together up to the environment "
                    "variable DALI_AMD_READER_MMAP_MB, 4096 by default, 0 = plain reads - and later epochs copy them from the "
                    "mapping; a reader keeps at most DALI_AMD_READER_FD_CAP descriptors open, by default a quarter of the "
                    "process's soft RLIMIT_NOFILE, which it never changes; the other readers always use plain I/O.  The data set is "
                    "taken to be static while a reader lives: a mapped file must not be truncated - the copy out of the mapping "
                    "then ends in SIGBUS, as with the reference's mmap loader.  With DALI_AMD_READER_ZERO_COPY=1 - the default "
                    "only when the process has at most four CPUs to itself - readers.file keeps a page-locked copy of every "
                    "file, up to DALI_AMD_READER_PINNED_MB = 4096 per process, which mixed decoders fetch with a device-side "
                    "copy; later changes to such a file are not seen, and cannot disturb the device.)", ArgValue::Bool(false))
    .AddRandomSeedArg();

DALI_SCHEMA(readers__File)
    .DocStr("Reads file contents and returns file-label pairs.\n\nThe labels are the indices of the alphabetically "
            "sorted sub-directories of ``file_root`` (or come from ``file_list`` / ``labels``).")
    .NumInput(0)
    .NumOutput(2)
    .AddOptionalTypeArg("file_root", "Path to a directory that contains the data files.", ArgType::STRING)
    .AddOptionalTypeArg("file_list", "Path to a text file with rows ``filename label``.", ArgType::STRING)
    .AddOptionalTypeArg("files", "A list of file paths to read the data from.", ArgType::STRING_VEC)
    .AddOptionalTypeArg("labels", "Labels accompanying ``files`` (default: the file index).", ArgType::INT_VEC)
    .AddOptionalTypeArg("file_filters", "Glob patterns to filter the files (default: known image extensions).",
                        ArgType::STRING_VEC)
    .AddOptionalTypeArg("dir_filters", "Glob patterns to filter the sub-directories.", ArgType::STRING_VEC)
    .AddOptionalArg("case_sensitive_filter", "Match the filters case-sensitively.", ArgValue::Bool(false))
    .AddOptionalArg("shuffle_after_epoch", "Reshuffle the whole dataset after each epoch.", ArgValue::Bool(false))
    .AddOptionalArg("shuffle_after_epoch_seed", "Seed for shuffle_after_epoch.", ArgValue::Int(0))
    .AddOptionalTypeArg("index_path", "MI355X extension: root of a tree of indexed JPEG containers made offline by "
                        "``tools/jpeg2idx.py`` (``<index_path>/<relative file name>.didx``).  Where a container exists it is read "
                        "in place of the file; ``decoders.image(device=\"mixed\")`` decodes it from its index (no position passes "
                        "in the GPU entropy decoder, also in the first epoch).  The sample keeps the file's name and label.  The "
                        "reference indexes its containers offline the same way (tools/tfrecord2idx, tools/wds2idx.py).",
                        ArgType::STRING)
    .AddParent("LoaderBase");

DALI_SCHEMA(FileReader).DocStr("Legacy alias of readers.file").NumInput(0).NumOutput(2).AddParent("readers__File");

static size_t start_index(size_t shard_id, size_t shard_num, size_t size) { return size * shard_id / shard_num; }
static int64_t num_samples(size_t shard_num, size_t size) { return (int64_t)std::ceil(size * 1.0 / shard_num); }

static std::vector<std::string> ListDir(const std::string &dir, bool want_dirs, const std::vector<std::string> &filters,
                                        bool case_sensitive) {
  DIR *d = opendir(dir.c_str());
  DALI_ENFORCE(d != nullptr, "Failed to open ", dir);
  std::vector<std::string> out;
  while (dirent *e = readdir(d)) {
    std::string name = e->d_name;
    if (name == "." || name == "..") continue;
    struct stat s;
    if (stat((dir + "/" + name).c_str(), &s) != 0) { closedir(d); DALI_FAIL("Could not access ", dir, "/", name); }
    bool is_dir = S_ISDIR(s.st_mode);
    if (is_dir != want_dirs) continue;
    bool ok = filters.empty();
    for (auto &f : filters) ok |= fnmatch(f.c_str(), name.c_str(), case_sensitive ? 0 : FNM_CASEFOLD) == 0;
    if (ok) out.push_back(name);
  }
  closedir(d);
  std::sort(out.begin(), out.end());
  return out;
}

// ---------------------------------------------------------------------------------------------- Loader
Loader::Loader(const OpSpec &spec)
    : shuffle_(spec.GetBool("random_shuffle")),
      initial_fill_(shuffle_ ? (int)spec.GetInt("initial_fill") : 1),
      num_shards_((int)spec.GetInt("num_shards")),
      shard_id_((int)spec.GetInt("shard_id")),
      stick_to_shard_(spec.GetBool("stick_to_shard")),
      pad_last_batch_(spec.GetBool("pad_last_batch")) {
  DALI_ENFORCE(num_shards_ > shard_id_, "num_shards needs to be greater than shard_id");
  DALI_ENFORCE(shard_id_ >= 0, "shard_id must be non-negative");
  DALI_ENFORCE(initial_fill_ > 0, "initial_fill must be positive");
  std::seed_seq seq({spec.GetInt("seed")});
  rng_ = std::default_random_engine(seq);
  virtual_shard_id_ = shard_id_;
}

void Loader::Init(int64_t size) {
  size_ = size;
  DALI_ENFORCE((int64_t)num_shards_ <= size_, "The number of input samples: ", size_,
               ", needs to be at least equal to the requested number of shards: ", num_shards_, ".");
  Reset(true);
}

ReaderMeta Loader::Meta() const {
  ReaderMeta m;
  m.epoch_size = size_;
  m.epoch_size_padded = pad_last_batch_ ? num_samples(num_shards_, size_) * num_shards_ : size_;
  m.number_of_shards = num_shards_;
  m.shard_id = shard_id_;
  m.pad_last_batch = pad_last_batch_;
  m.stick_to_shard = stick_to_shard_;
  return m;
}

// checkpoint (loader.h:279,335,485-503): the COMPLETE state - position of the sequential stream, shard bookkeeping,
// the samples sitting in the shuffle buffer (read ahead but not yet returned), the epoch end marks and the rng - so a
// restored reader continues with exactly the sample the saved one would have returned next, whether it is a fresh
// instance or one that has already run.
std::string Loader::Save() const {
  std::ostringstream ss;
  ss << current_index_ << " " << virtual_shard_id_ << " " << read_in_shard_ << " " << total_read_ << " " << consumed_
     << " " << returned_ << " " << epoch_ << " " << last_pick_ << " " << (filled_ ? 1 : 0) << " " << buffer_.size();
  for (auto &b : buffer_) ss << " " << b.first << " " << b.second;
  ss << " " << shard_ends_.size();
  for (int64_t e : shard_ends_) ss << " " << e;
  ss << " " << rng_;
  return ss.str();
}
void Loader::Restore(const std::string &s) {
  std::istringstream ss(s);
  int filled = 0;
  size_t nbuf = 0, nends = 0;
  ss >> current_index_ >> virtual_shard_id_ >> read_in_shard_ >> total_read_ >> consumed_ >> returned_ >> epoch_ >>
      last_pick_ >> filled >> nbuf;
  DALI_ENFORCE(!ss.fail() && nbuf <= (size_t)initial_fill_, "reader: malformed checkpoint");
  buffer_.assign(nbuf, {0, 0});
  for (auto &b : buffer_) ss >> b.first >> b.second;
  ss >> nends;
  DALI_ENFORCE(!ss.fail() && nends <= (1u << 20), "reader: malformed checkpoint");
  shard_ends_.assign(nends, 0);
  for (auto &e : shard_ends_) ss >> e;
  ss >> std::ws >> rng_;  // libstdc++ reads the engine with skipws cleared
  DALI_ENFORCE(!ss.fail(), "reader: malformed checkpoint");
  for (auto &b : buffer_) DALI_ENFORCE(b.second >= 0 && b.second < size_, "reader: checkpoint of another dataset");
  DALI_ENFORCE(last_pick_ < size_ && current_index_ >= 0 && current_index_ <= size_, "reader: checkpoint of another dataset");
  filled_ = filled != 0;
}

// sequential stream over the dataset, starting at this shard and (unless stick_to_shard) moving
// on to the next shard every epoch (loader.h:413-452)
void Loader::Reset(bool wrap_to_shard) {
  current_index_ = wrap_to_shard ? (int64_t)start_index(virtual_shard_id_, num_shards_, size_) : 0;
}
bool Loader::IsNextShard(int64_t idx) const {
  return idx >= size_ || (stick_to_shard_ && shard_id_ + 1 < num_shards_ &&
                          idx >= (int64_t)start_index(shard_id_ + 1, num_shards_, size_));
}
int64_t Loader::ReadSequential() {
  if (IsNextShard(current_index_)) Reset(stick_to_shard_);
  int64_t idx = current_index_++;
  // shard bookkeeping (IncreaseReadSampleCounter, loader.h:440-457)
  read_in_shard_++;
  int64_t rel_end = (int64_t)start_index(virtual_shard_id_ + 1, num_shards_, size_) -
                    (int64_t)start_index(virtual_shard_id_, num_shards_, size_);
  if (read_in_shard_ >= rel_end) {
    shard_ends_.push_back(total_read_ + 1);
    if (!stick_to_shard_) virtual_shard_id_ = (virtual_shard_id_ + 1) % num_shards_;
    read_in_shard_ = 0;
  }
  total_read_++;
  return idx;
}

// one sample of the output stream: shuffle buffer + last-batch padding (loader.h:207-345)
int64_t Loader::NextIndex(bool is_new_batch) {
  if (buffer_.empty() && !filled_) {
    for (int i = 0; i < initial_fill_; i++) {
      int64_t seq = total_read_;  // by value: ReadSequential() advances total_read_
      int64_t idx = ReadSequential();
      buffer_.push_back({seq, idx});
    }
    filled_ = true;
  }
  // the current epoch (shard) is depleted when everything read before its end mark was returned
  if (!shard_ends_.empty() && consumed_ >= shard_ends_.front()) {
    bool pad = (returned_ < num_samples(num_shards_, size_) || !is_new_batch) && pad_last_batch_;
    if (pad && last_pick_ >= 0) {
      returned_++;
      return last_pick_;
    }
    shard_ends_.pop_front();
    returned_ = 0;
    epoch_++;
  }
  // candidates: buffered samples that belong to the current epoch
  int64_t limit = shard_ends_.empty() ? total_read_ : shard_ends_.front();
  // Every buffered sample was read before total_read_: while the epoch's end mark has not been read past (limit >=
  // total_read_) ALL of them are candidates and the pick is a position - O(1).  Only the tail of an epoch (its last
  // initial_fill samples) counts and walks the buffer.  (Round 6: the two walks per SAMPLE cost a shuffling reader 0.64 ms
  // per 256-image batch with initial_fill = 4096 - more than the GPU needs for the batch.)
  int ncand = (int)buffer_.size();
  const bool all = limit >= total_read_;
  if (!all) {
    ncand = 0;
    for (auto &b : buffer_) ncand += b.first < limit;
  }
  DALI_ENFORCE(ncand > 0, "Internal error: shuffle buffer has no sample of the current epoch");
  int pick = 0;
  if (shuffle_) pick = std::uniform_int_distribution<>(0, ncand - 1)(rng_);
  int pos = all ? pick : -1;
  for (int i = 0, k = 0; !all && i < (int)buffer_.size(); i++) {
    if (buffer_[i].first < limit) {
      if (k == pick) { pos = i; break; }
      k++;
    }
  }
  int64_t idx = buffer_[pos].second;
  {
    int64_t seq = total_read_;
    int64_t next = ReadSequential();
    buffer_[pos] = {seq, next};
  }
  consumed_++;
  returned_++;
  last_pick_ = idx;
  return idx;
}

// ---------------------------------------------------------------------------------------------- readers.file
// The reader runs AHEAD of the executor (DataReader: prefetch thread + queue of `prefetch_queue_depth` batches that is
// independent of the executor, dali/operators/reader/reader_op.h:57-183,386-415; Loader::ReadOne / PrepareEmpty,
// loader/loader.h:231-272).  Three kinds of threads:
//   planner  (1)  draws the indices of the next batch from the Loader, looks the samples up in the decoder caches
//                 (skip_cached_images), lays the batch out in a page-locked block and publishes its read tasks
//   readers  (W)  take chunks of tasks of the oldest unfinished batch: pread() from a long-lived file descriptor into the
//                 sample's place in the block.  They only sleep when no planned batch has tasks left, so while the
//                 executor consumes they never pay a wake-up per batch (the thread pool's RunAll did: a 0.3 ms read
//                 pass took 1.0-1.3 ms on a host with 256 logical CPUs and a 16-CPU quota)
//   RunImpl       (the executor's host stage) pops the oldest batch once its reads are done and SHARES its block with
//                 the output - no copy; the block goes back to the planner when no ring slot refers to it any more.
//                 The decoder transfers the page-locked block to the device as it is.
// Checkpoints describe what has been HANDED OUT: every planned batch carries the Loader's state behind its picks.
// CPUs this process can count on.  The cgroup's CPU quota (v2: cpu.max, v1: cfs_quota_us / cfs_period_us) belongs to the
// node's job and is shared between the ranks a launcher put on this node (LOCAL_WORLD_SIZE; 1 when unset).  The affinity
// mask is the PROCESS's: a mask narrower than the machine means the launcher (or Pipeline(set_affinity=True)) already gave
// this rank its own CPUs - that share is not divided again; a mask that still spans the whole machine is
// shared like the quota.  0: unknown.
static double UsableCpusPerRank() {
  cpu_set_t set;
  CPU_ZERO(&set);
  double cpus = sched_getaffinity(0, sizeof(set), &set) == 0 ? (double)CPU_COUNT(&set) : 0.0;
  if (cpus <= 0) return 0.0;
  auto read_pair = [](const char *path, double *a, double *b) {
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    char first[64] = {0};
    double second = 0;
    const int n = fscanf(f, "%63s %lf", first, &second);
    fclose(f);
    if (n < 1 || !strcmp(first, "max")) return 0;
    *a = atof(first);
    if (b) *b = second;
    return n;
  };
  int ranks = 1;
  if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
  const long online = sysconf(_SC_NPROCESSORS_ONLN);
  if (online <= 0 || cpus >= (double)online) cpus /= ranks;   // the whole machine: every rank sees the same mask
  double quota = 0, period = 0;
  if (read_pair("/sys/fs/cgroup/cpu.max", &quota, &period) == 2 && quota > 0 && period > 0) {
    cpus = std::min(cpus, quota / period / ranks);
  } else if (read_pair("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &quota, nullptr) >= 1 && quota > 0 &&
             read_pair("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &period, nullptr) >= 1 && period > 0) {
    cpus = std::min(cpus, quota / period / ranks);
  }
  return cpus;
}

class FileReaderOp : public OperatorBase {
 public:
  explicit FileReaderOp(const OpSpec &spec)
      : OperatorBase(spec), loader_(spec), skip_cached_(spec.GetBool("skip_cached_images")),
        read_ahead_(spec.GetBool("read_ahead")), use_mmap_(!spec.GetBool("dont_use_mmap")),
        device_id_((int)spec.GetInt("device_id")),
        depth_(std::max(1, (int)spec.GetInt("prefetch_queue_depth"))), consumed_state_(loader_) {
    Discover();
    loader_.Init((int64_t)entries_.size());
    consumed_state_ = loader_;
    paths_.reserve(entries_.size());
    for (auto &e : entries_)
      paths_.push_back((root_.empty() || (!e.first.empty() && e.first[0] == '/')) ? e.first : root_ + "/" + e.first);
    // index_path (round 6): a tree of indexed JPEG containers made offline by tools/jpeg2idx.py - <index_path>/<name as
    // under file_root>.didx.  Where one exists it is READ IN PLACE OF the file (the mixed decoders take it wherever they take
    // a JPEG and decode it from its index entry); the sample keeps the file's name - labels, error messages and the decoder
    // caches' keys do not change.  Files without one (progressive, PNG ...) are read as they are.
    read_paths_ = paths_;
    if (const ArgValue *ip = spec.TryArg("index_path")) {
      const std::string base = ip->s;
      for (size_t i = 0; i < entries_.size() && !base.empty(); i++) {
        std::string rel = entries_[i].first;
        if (!rel.empty() && rel[0] == '/') {   // an absolute name: relative to file_root when it lies below it, else its base name
          if (!root_.empty() && rel.compare(0, root_.size(), root_) == 0 && rel.size() > root_.size() && rel[root_.size()] == '/')
            rel = rel.substr(root_.size() + 1);
          else
            rel = rel.substr(rel.rfind('/') + 1);
        }
        const std::string cand = base + "/" + rel + ".didx";
        if (access(cand.c_str(), R_OK) == 0) { read_paths_[i] = cand; indexed_files_++; }
      }
    }
    size_cache_.assign(entries_.size(), -1);
    fds_ = std::make_unique<std::atomic<int>[]>(entries_.size());
    for (size_t i = 0; i < entries_.size(); i++) fds_[i].store(-1, std::memory_order_relaxed);
    // long-lived mappings (the reference's default, file_loader / dont_use_mmap = False: FileStream::Open(.., mmap)): a
    // file is mapped - populated - the first time it is read and stays mapped, so from the second epoch on a sample is
    // one memcpy out of the mapping: no system call and none of the page cache's per-page work (the part of pread that
    // does not scale with the number of reader threads).  Mappings are never taken down while the reader lives (munmap
    // means TLB shoot-downs on every core the process runs on); files beyond the budget are read with pread.
    maps_ = std::make_unique<std::atomic<const char *>[]>(entries_.size());
    for (size_t i = 0; i < entries_.size(); i++) maps_[i].store(nullptr, std::memory_order_relaxed);
    // Device-side fetch ("zero copy", round 5; re-built in round 6): from its second sighting on a file is not copied by
    // any host core: the sample the reader hands out is a page-locked RESIDENT COPY of the file (made once, on the first
    // read, in blocks from daliamdHostAlloc) and a mixed decoder fetches the bytes with a device-side copy - they cross the
    // bus once (the copy out of the mapping was 60 % of the pipeline's host time per image).
    // Round 5 handed out the file MAPPINGS themselves, registered with the device (hipHostRegister on page-cache pages).
    // That tied the device to the files: truncating one made the driver evict the process's queues for minutes, with no
    // error.  The resident copy is anonymous memory of this process: whatever happens to the file
    // afterwards - truncate, rewrite, unlink - the reader keeps handing out the bytes it read, exactly like the decoder
    // caches do.  Budget: DALI_AMD_READER_PINNED_MB (process-wide, default 4096); files beyond it are copied per epoch as
    // before.
    // WHEN: measured on the bench box (gpurun_out/r05_r, r05_t - r05_v), the device-side fetch is the better way when host
    // cores are what a rank lacks - one rank of eight on 2 of 16 CPUs: 314 k img/s against 238 k, 3.9 ms of CPU per
    // 512-image batch against 5.6 - and the worse one when they are not: alone it runs at the bus rate (56 GB/s, the copy
    // engine out of page-locked memory: 52), but inside the busy pipeline it takes 0.6-0.8 ms per batch instead of 0.43
    // and the kernels next to it run 1.3-1.8 times longer (one rank on 16 CPUs: 300 k against 410-480 k).  Hence the
    // default: on when this process has at most four CPUs to itself (UsableCpusPerRank above), off otherwise (the copying
    // path needs 3.1-3.5 ms of CPU per 256-image batch: with four CPUs it is host-bound at about the rate the device-side
    // fetch reaches); DALI_AMD_READER_ZERO_COPY=1 / 0 decides.
    resident_ = std::make_unique<std::atomic<const char *>[]>(entries_.size());
    for (size_t i = 0; i < entries_.size(); i++) resident_[i].store(nullptr, std::memory_order_relaxed);
    {
      const double cpus = UsableCpusPerRank();
      zero_copy_ = cpus > 0 && cpus <= 4.0;
    }
    if (const char *e = getenv("DALI_AMD_READER_ZERO_COPY")) zero_copy_ = atoi(e) != 0;
    zero_copy_ = zero_copy_ && HaveDevice();
    pinned_budget_ = (int64_t)4096 << 20;
    if (const char *e = getenv("DALI_AMD_READER_PINNED_MB")) pinned_budget_ = (int64_t)(std::max(0.0, atof(e)) * 1048576.0);
    // Both budgets are PROCESS-wide: the mappings of all readers of the process together stay below
    // DALI_AMD_READER_MMAP_MB (4096 by default; 0: plain reads) - SharedMappedBytes() - and a reader keeps at most
    // DALI_AMD_READER_FD_CAP descriptors open, by default a quarter of the soft limit the process was STARTED with.  The
    // limit itself is the host application's: an operator does not raise it (descriptors above 1023 break select()-based
    // code elsewhere in the process).
    map_budget_ = (int64_t)4096 << 20;
    if (const char *e = getenv("DALI_AMD_READER_MMAP_MB")) map_budget_ = (int64_t)(std::max(0.0, atof(e)) * 1048576.0);
    if (!use_mmap_) map_budget_ = 0;
    struct rlimit rl;
    size_t cap = 256;
    if (getrlimit(RLIMIT_NOFILE, &rl) == 0)
      cap = rl.rlim_cur == RLIM_INFINITY ? 16384 : (size_t)std::min<rlim_t>(rl.rlim_cur / 4, 16384);
    fd_cap_ = std::max<size_t>(16, cap);
    if (const char *e = getenv("DALI_AMD_READER_FD_CAP")) fd_cap_ = (size_t)std::max(1, atoi(e));
    int workers = std::max(1, (int)spec.GetInt("num_threads"));
    if (const char *e = getenv("DALI_AMD_READER_THREADS")) workers = std::max(1, atoi(e));
    // eight readers copy a 25 MB batch out of the page cache in 0.3 ms; more only contend inside the kernel (MI355X host,
    // 16-CPU quota: 6 / 8 / 12 / 16 readers spend 3.4 / 2.8 / 5.5 / 6.2 ms of CPU per batch, gpurun_out/r4d)
    num_workers_ = getenv("DALI_AMD_READER_THREADS") ? std::min(workers, 64) : std::min(workers, 8);
  }

  ~FileReaderOp() override {
    StopThreads();
    for (size_t i = 0; i < entries_.size(); i++) {
      const int fd = fds_[i].load(std::memory_order_relaxed);
      if (fd >= 0) close(fd);
      const char *m = maps_[i].load(std::memory_order_relaxed);
      if (m && m != kNoMapping) {
        munmap(const_cast<char *>(m), (size_t)size_cache_[i]);
        SharedMappedBytes().fetch_sub((int64_t)size_cache_[i], std::memory_order_relaxed);
      }
    }
    // (the resident copies go with the last batch that refers to them: every such batch holds the arena)
  }

  ReaderMeta GetReaderMeta() const override { return loader_.Meta(); }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }

  void RunImpl(Workspace &ws) override {
    if (!started_) StartThreads();   // lazily: every operator of the graph (the decoder and its caches) exists by now
    std::shared_ptr<Prefetched> b;
    {
      std::unique_lock<std::mutex> lk(m_);
      cv_ready_.wait(lk, [&] { return !queue_.empty() && queue_.front()->remaining.load(std::memory_order_acquire) == 0; });
      b = std::move(queue_.front());
      queue_.pop_front();
      consumed_state_ = b->after;
    }
    const std::string error = b->error;
    if (error.empty()) {
      ws.Output(0).ShareData(b->data);
      ws.Output(1).ShareData(b->labels);
    }
    {
      // (only now: the planner recycles a batch of this list as soon as nobody else holds its blocks)
      std::lock_guard<std::mutex> g(m_);
      in_use_.push_back(std::move(b));
    }
    cv_space_.notify_one();
    if (!error.empty()) DALI_FAIL(error);
  }

  std::string SaveState() const override {
    std::lock_guard<std::mutex> g(m_);
    return consumed_state_.Save();
  }
  void RestoreState(const std::string &s) override {
    StopThreads();          // what was read ahead belongs to the old position
    queue_.clear();
    loader_.Restore(s);
    consumed_state_ = loader_;
  }

 private:
  struct Prefetched {
    TensorList data{StorageDevice::CPU}, labels{StorageDevice::CPU};
    std::vector<int64_t> picks;
    std::vector<off_t> sizes;
    std::vector<int> tasks;              // samples that need a file read
    size_t next_task = 0;                // (under m_)
    std::atomic<int> remaining{0};       // reads not finished yet
    std::mutex err_m;
    std::string error;
    Loader after;                        // the index stream behind this batch's picks
    explicit Prefetched(const Loader &l) : after(l) {}
  };
  static constexpr int kChunk = 8;       // samples a reader thread takes at a time

  void StartThreads() {
    std::lock_guard<std::mutex> g(m_);
    if (started_) return;
    stop_ = false;
    started_ = true;
    planner_ = std::thread([this] { NameThisThread("dali-rd-plan"); PlannerLoop(); });
    for (int i = 0; i < num_workers_; i++)
      workers_.emplace_back([this, i] { NameThisThread("dali-rd" + std::to_string(i)); WorkerLoop(); });
  }
  void StopThreads() {
    {
      std::lock_guard<std::mutex> g(m_);
      if (!started_) return;
      stop_ = true;
    }
    cv_space_.notify_all();
    cv_work_.notify_all();
    if (planner_.joinable()) planner_.join();
    for (auto &t : workers_) t.join();
    workers_.clear();
    std::lock_guard<std::mutex> g(m_);
    started_ = false;
  }

  // ---- planner ----
  void PlannerLoop() {
    if (HaveDevice()) daliamdSetDevice(device_id_);   // page-locked allocations belong to this pipeline's device
    for (;;) {
      std::shared_ptr<Prefetched> b;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_space_.wait(lk, [&] { return stop_ || (int)queue_.size() < depth_; });
        if (stop_) return;
        for (auto it = in_use_.begin(); it != in_use_.end(); ++it)
          if ((*it)->data.buffer_use_count() == 1 && (*it)->labels.buffer_use_count() == 1) {
            b = std::move(*it);
            in_use_.erase(it);
            break;
          }
      }
      if (!b) b = std::make_shared<Prefetched>(loader_);
      try {
        Plan(*b);
      } catch (const std::exception &e) {
        b->error = e.what();
        b->tasks.clear();
        b->remaining.store(0);
      }
      bool has_tasks;
      {
        std::lock_guard<std::mutex> g(m_);
        has_tasks = !b->tasks.empty();
        queue_.push_back(b);
      }
      if (has_tasks) cv_work_.notify_all();
      else cv_ready_.notify_all();
    }
  }

  void Plan(Prefetched &b) {
    const int n = max_batch_size_;
    b.picks.resize(n);
    for (int i = 0; i < n; i++) b.picks[i] = loader_.NextIndex(i == 0);
    b.after = loader_;
    b.error.clear();
    b.sizes.assign(n, 0);
    b.tasks.clear();
    b.next_task = 0;
    std::vector<TensorShape> shapes(n), lshape(n, TensorShape{1});
    std::vector<void *> ext(n, nullptr);
    bool any_ext = false;
    // skip_cached_images (loader.h:466-480, file_label_loader.cc:49-56): a sample the decoder cache of this device holds
    // is not read - its tensor is empty, the decoder finds it by its source_info.  The caches are looked up at run time:
    // the decoder that owns them may be constructed after the reader.
    std::shared_ptr<ImageCache> decoded = skip_cached_ ? ImageCache::Find(device_id_) : nullptr;
    std::shared_ptr<StreamCache> encoded = skip_cached_ ? StreamCache::Find(device_id_) : nullptr;
    for (int i = 0; i < n; i++) {
      const int64_t idx = b.picks[i];
      if ((decoded && decoded->IsCached(paths_[idx])) || (encoded && encoded->IsCached(paths_[idx]))) {
        shapes[i] = {0};
        continue;
      }
      off_t &cached = size_cache_[idx];  // the dataset is static: one stat() per file, not one per epoch
      if (cached < 0) {
        struct stat st;
        DALI_ENFORCE(stat(read_paths_[idx].c_str(), &st) == 0, "Could not open file ", read_paths_[idx]);
        cached = st.st_size;
      }
      b.sizes[i] = cached;
      shapes[i] = {(int64_t)cached};
      if (zero_copy_ && cached > 0) {
        // resident in page-locked memory of this reader (which stays until the reader goes): the sample IS that copy
        const char *r = resident_[idx].load(std::memory_order_acquire);
        if (r && r != kFilling && r != kNoRoom) {
          ext[i] = const_cast<char *>(r);
          any_ext = true;
          continue;
        }
      }
      b.tasks.push_back(i);
    }
    if (any_ext) {
      b.data.Resize(shapes, DALI_UINT8, 1, ext, std::vector<int64_t>(n, 0), arena_);
      b.data.SetExtDeviceVisible(true);
    } else {
      b.data.Resize(shapes, DALI_UINT8);
    }
    b.labels.Resize(lshape, DALI_INT32);
    b.data.source_info.resize(n);
    for (int i = 0; i < n; i++) {
      *static_cast<int32_t *>(b.labels.raw(i)) = entries_[b.picks[i]].second;
      b.data.source_info[i] = paths_[b.picks[i]];
    }
    b.remaining.store((int)b.tasks.size(), std::memory_order_release);
    EvictDescriptors();
  }

  // Descriptors above the cap are closed oldest first - never one that a batch with unfinished reads may be using.
  void EvictDescriptors() {
    std::lock_guard<std::mutex> fg(fd_m_);
    if (open_fifo_.size() <= fd_cap_) return;
    std::vector<int64_t> busy;
    {
      std::lock_guard<std::mutex> g(m_);
      for (auto &q : queue_)
        if (q->remaining.load(std::memory_order_acquire) != 0) busy.insert(busy.end(), q->picks.begin(), q->picks.end());
    }
    std::sort(busy.begin(), busy.end());
    size_t scanned = 0;
    const size_t limit = open_fifo_.size();
    while (open_fifo_.size() > fd_cap_ && scanned++ < limit) {
      const int64_t idx = open_fifo_.front();
      open_fifo_.pop_front();
      if (std::binary_search(busy.begin(), busy.end(), idx)) {
        open_fifo_.push_back(idx);
        continue;
      }
      const int fd = fds_[idx].exchange(-1, std::memory_order_acq_rel);
      if (fd >= 0) close(fd);
    }
  }

  // ---- readers ----
  void WorkerLoop() {
    if (HaveDevice()) daliamdSetDevice(device_id_);   // mappings are registered with this pipeline's device
    for (;;) {
      std::shared_ptr<Prefetched> b;
      size_t t0 = 0, t1 = 0;
      {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
          if (stop_) return;
          for (auto &q : queue_)
            if (q->next_task < q->tasks.size()) { b = q; break; }
          if (b) break;
          cv_work_.wait(lk);
        }
        t0 = b->next_task;
        t1 = std::min(b->tasks.size(), t0 + kChunk);
        b->next_task = t1;
      }
      for (size_t t = t0; t < t1; t++) {
        const int i = b->tasks[t];
        std::string err = ReadSample(*b, i);
        if (!err.empty()) {
          std::lock_guard<std::mutex> g(b->err_m);
          if (b->error.empty()) b->error = err;
        }
      }
      if (b->remaining.fetch_sub((int)(t1 - t0), std::memory_order_acq_rel) == (int)(t1 - t0)) {
        std::lock_guard<std::mutex> g(m_);   // the consumer checks the counter under this lock: no lost wake-up
        cv_ready_.notify_all();
      }
    }
  }

  int Descriptor(int64_t idx) {
    int fd = fds_[idx].load(std::memory_order_acquire);
    if (fd >= 0) return fd;
    fd = open(read_paths_[idx].c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -1;
    // read_ahead: the whole file is wanted, tell the kernel before the first byte is asked for
    if (read_ahead_) posix_fadvise(fd, 0, 0, POSIX_FADV_WILLNEED);
    int expected = -1;
    if (!fds_[idx].compare_exchange_strong(expected, fd, std::memory_order_acq_rel)) {
      close(fd);   // another reader opened the same file (a sample repeated inside the batches in flight)
      return expected;
    }
    std::lock_guard<std::mutex> g(fd_m_);
    open_fifo_.push_back(idx);
    return fd;
  }

  // The file's persistent mapping, made on first use while the budget lasts (nullptr: read it with pread).
  const char *Mapping(int64_t idx, int fd, off_t size) {
    const char *m = maps_[idx].load(std::memory_order_acquire);
    if (m) return m == kNoMapping ? nullptr : m;
    if (size <= 0 || SharedMappedBytes().fetch_add((int64_t)size, std::memory_order_relaxed) + (int64_t)size > map_budget_) {
      SharedMappedBytes().fetch_sub(size > 0 ? (int64_t)size : 0, std::memory_order_relaxed);
      maps_[idx].store(kNoMapping, std::memory_order_release);
      return nullptr;
    }
    // (a file that is no longer as long as the index says must fail with a message, not with a bus error in memcpy:
    // such a file is read with pread, which notices)
    struct stat st;
    void *p = fstat(fd, &st) == 0 && st.st_size == size
                  ? mmap(nullptr, (size_t)size, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0) : MAP_FAILED;
    if (p == MAP_FAILED) {
      SharedMappedBytes().fetch_sub((int64_t)size, std::memory_order_relaxed);
      maps_[idx].store(kNoMapping, std::memory_order_release);
      return nullptr;
    }
    const char *expected = nullptr;
    if (!maps_[idx].compare_exchange_strong(expected, static_cast<const char *>(p), std::memory_order_acq_rel)) {
      munmap(p, (size_t)size);   // another reader mapped the same file (a sample repeated inside the batches in flight)
      SharedMappedBytes().fetch_sub((int64_t)size, std::memory_order_relaxed);
      return expected == kNoMapping ? nullptr : expected;
    }
    return static_cast<const char *>(p);
  }

  // A file's bytes from its mapping into the page-locked block the host->device transfer reads.  Nothing on the host looks
  // at most of these bytes again (the decoder parses the headers, the copy engine takes the rest): streaming stores do not
  // pull the destination lines into the cache first - 8.2 against 6.2 GB/s per core for 94 KB copies out of cold memory
  // (tools/microbench note in HISTORY.md section 6c), a quarter of a reader thread's time per image.
  static void CopyOut(char *dst, const char *src, size_t n) {
#if !defined(__SSE2__)
    memcpy(dst, src, n);   // (no streaming stores on this host: aarch64 nodes with AMD GPUs exist)
    return;
#else
    if (n < 4096) { memcpy(dst, src, n); return; }
    size_t head = (64 - (reinterpret_cast<uintptr_t>(dst) & 63)) & 63;
    memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    const size_t blocks = n / 64;
    for (size_t k = 0; k < blocks; k++) {
      const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 64 * k));
      const __m128i b2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 64 * k + 16));
      const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 64 * k + 32));
      const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + 64 * k + 48));
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 64 * k), a);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 64 * k + 16), b2);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 64 * k + 32), c);
      _mm_stream_si128(reinterpret_cast<__m128i *>(dst + 64 * k + 48), e);
    }
    _mm_sfence();
    memcpy(dst + 64 * blocks, src + 64 * blocks, n - 64 * blocks);
#endif
  }

  // Page-locked room for one file's resident copy (nullptr: budget spent, or the file is larger than a block).  Blocks of
  // 32 MB, bump-allocated, 64-byte aligned, never handed back before the reader goes.
  char *ReserveResident(size_t size) {
    constexpr size_t kBlock = (size_t)32 << 20;
    const size_t need = (size + 63) & ~(size_t)63;
    if (need == 0 || need > kBlock / 4) return nullptr;
    std::lock_guard<std::mutex> g(arena_->m);
    auto &pinned_blocks_ = arena_->blocks;
    auto &pinned_used_ = arena_->used;
    if (pinned_blocks_.empty() || pinned_used_ + need > kBlock) {
      if (SharedPinnedBytes().fetch_add((int64_t)kBlock, std::memory_order_relaxed) + (int64_t)kBlock > pinned_budget_) {
        SharedPinnedBytes().fetch_sub((int64_t)kBlock, std::memory_order_relaxed);
        return nullptr;
      }
      void *p = nullptr;
      if (daliamdHostAlloc(&p, kBlock) != DALIAMD_SUCCESS || !p) {
        SharedPinnedBytes().fetch_sub((int64_t)kBlock, std::memory_order_relaxed);
        return nullptr;
      }
      pinned_blocks_.push_back({p, kBlock});
      pinned_used_ = 0;
    }
    char *slot = static_cast<char *>(pinned_blocks_.back().first) + pinned_used_;
    pinned_used_ += need;
    return slot;
  }

  std::string ReadSample(Prefetched &b, int i) {
    const int64_t idx = b.picks[i];
    char *dst = static_cast<char *>(b.data.raw(i));
    if (zero_copy_) {
      // first sighting of a file that may become resident: read it INTO its page-locked place (one thread per file; a
      // second reader of the same file inside the batches in flight takes the ordinary path below), then copy it out for
      // this batch, whose layout was planned before the copy existed
      const char *expected = nullptr;
      if (resident_[idx].compare_exchange_strong(expected, kFilling, std::memory_order_acq_rel)) {
        char *slot = ReserveResident((size_t)b.sizes[i]);
        if (!slot) {
          resident_[idx].store(kNoRoom, std::memory_order_release);
        } else {
          const int fd = Descriptor(idx);
          off_t got = 0;
          while (fd >= 0 && got < b.sizes[i]) {
            const ssize_t r = pread(fd, slot + got, (size_t)(b.sizes[i] - got), got);
            if (r <= 0) break;
            got += r;
          }
          if (got != b.sizes[i]) {
            resident_[idx].store(nullptr, std::memory_order_release);   // (the slot is lost; the next sighting tries again)
            return make_string(fd < 0 ? "Could not open file " : "Failed to read file ", read_paths_[idx]);
          }
          CopyOut(dst, slot, (size_t)b.sizes[i]);
          resident_[idx].store(slot, std::memory_order_release);
          return "";
        }
      } else if (expected != kFilling && expected != kNoRoom) {
        CopyOut(dst, expected, (size_t)b.sizes[i]);   // became resident after this batch was planned
        return "";
      }
    }
    if (map_budget_ > 0) {   // already mapped: no descriptor needed (it may have been evicted)
      const char *m = maps_[idx].load(std::memory_order_acquire);
      if (m && m != kNoMapping) {
        CopyOut(dst, m, (size_t)b.sizes[i]);
        return "";
      }
    }
    const int fd = Descriptor(idx);
    if (fd < 0) return make_string("Could not open file ", read_paths_[idx]);
    if (map_budget_ > 0) {
      if (const char *m = Mapping(idx, fd, b.sizes[i])) {
        CopyOut(dst, m, (size_t)b.sizes[i]);
        return "";
      }
    }
    off_t got = 0;
    while (got < b.sizes[i]) {
      const ssize_t r = pread(fd, dst + got, (size_t)(b.sizes[i] - got), got);
      if (r <= 0) break;
      got += r;
    }
    if (got != b.sizes[i]) return make_string("Failed to read file ", read_paths_[idx]);
    return "";
  }

  static bool HaveDevice() {
    static const bool have = [] { int n = 0; daliamdDeviceCount(&n); return n > 0; }();
    return have;
  }

  Loader loader_;                 // planner thread only once the threads run
  bool skip_cached_, read_ahead_, use_mmap_;
  // per file: nullptr = not seen yet, kFilling = a reader thread is making the resident copy, kNoRoom = stays a copied
  // file, else the page-locked resident copy (see ReadSample)
  std::unique_ptr<std::atomic<const char *>[]> resident_;
  static inline const char *const kFilling = reinterpret_cast<const char *>(1);
  static inline const char *const kNoRoom = reinterpret_cast<const char *>(2);
  bool zero_copy_ = false;
  struct ResidentArena {
    std::mutex m;
    std::vector<std::pair<void *, size_t>> blocks;
    size_t used = 0;                                    // of the newest block
    ~ResidentArena() {
      for (auto &blk : blocks) {
        daliamdHostFree(blk.first);
        SharedPinnedBytes().fetch_sub((int64_t)blk.second, std::memory_order_relaxed);
      }
    }
  };
  std::shared_ptr<ResidentArena> arena_ = std::make_shared<ResidentArena>();
  int64_t pinned_budget_ = 0;
  static std::atomic<int64_t> &SharedPinnedBytes() { static std::atomic<int64_t> v{0}; return v; }
  std::unique_ptr<std::atomic<const char *>[]> maps_;   // per file: nullptr = not tried yet, kNoMapping = pread, else the mapping
  static inline const char *const kNoMapping = reinterpret_cast<const char *>(1);
  // bytes mapped by every readers.file of the process (the budget is per process, not per reader)
  static std::atomic<int64_t> &SharedMappedBytes() { static std::atomic<int64_t> v{0}; return v; }
  int64_t map_budget_ = 0;
  int device_id_, depth_, num_workers_ = 1;
  std::vector<off_t> size_cache_;  // planner thread
  std::vector<std::string> paths_;       // the samples' names (source_info, cache keys)
  std::vector<std::string> read_paths_;  // what is opened for them: the same, or the file's indexed container (index_path)
  int64_t indexed_files_ = 0;
  std::unique_ptr<std::atomic<int>[]> fds_;
  std::mutex fd_m_;
  std::deque<int64_t> open_fifo_;
  size_t fd_cap_ = 256;

  mutable std::mutex m_;
  std::condition_variable cv_space_, cv_work_, cv_ready_;
  std::deque<std::shared_ptr<Prefetched>> queue_;     // planned batches, oldest first (reads finished or not)
  std::vector<std::shared_ptr<Prefetched>> in_use_;   // handed out; recycled when their blocks are free again
  Loader consumed_state_;                             // the index stream behind the last batch handed out
  bool started_ = false, stop_ = false;
  std::thread planner_;
  std::vector<std::thread> workers_;

  void Discover() {
    if (const ArgValue *files = spec_.TryArg("files")) {
      std::vector<std::string> names = files->type == ArgType::STRING ? std::vector<std::string>{files->s} : files->sv;
      std::vector<int64_t> labels;
      if (spec_.TryArg("labels")) labels = spec_.GetIntVec("labels");
      DALI_ENFORCE(labels.empty() || labels.size() == names.size(), "Provided ", labels.size(), " labels for ",
                   names.size(), " files.");
      for (size_t i = 0; i < names.size(); i++) entries_.push_back({names[i], labels.empty() ? (int)i : (int)labels[i]});
      if (spec_.TryArg("file_root")) root_ = spec_.GetString("file_root");
    } else if (spec_.TryArg("file_list")) {
      std::string list = spec_.GetString("file_list");
      if (spec_.TryArg("file_root")) root_ = spec_.GetString("file_root");
      else { size_t p = list.rfind('/'); root_ = p == std::string::npos ? "." : list.substr(0, p); }
      std::ifstream f(list);
      DALI_ENFORCE(f.good(), "Failed to open ", list);
      std::string line;
      while (std::getline(f, line)) {
        if (line.empty()) continue;
        size_t sp = line.rfind(' ');
        DALI_ENFORCE(sp != std::string::npos, "Malformed line in file_list: ", line);
        entries_.push_back({line.substr(0, sp), std::stoi(line.substr(sp + 1))});
      }
    } else {
      DALI_ENFORCE(spec_.TryArg("file_root"), "Either `file_root`, `file_list` or `files` must be specified");
      root_ = spec_.GetString("file_root");
      std::vector<std::string> filters = {"*.jpg", "*.jpeg", "*.png", "*.bmp", "*.tif", "*.tiff", "*.pnm", "*.ppm",
                                          "*.pgm", "*.pbm", "*.jp2", "*.webp", "*.flac", "*.ogg", "*.wav"};
      std::vector<std::string> dir_filters;
      if (const ArgValue *ff = spec_.TryArg("file_filters")) filters = ff->type == ArgType::STRING ? std::vector<std::string>{ff->s} : ff->sv;
      if (const ArgValue *df = spec_.TryArg("dir_filters")) dir_filters = df->type == ArgType::STRING ? std::vector<std::string>{df->s} : df->sv;
      bool cs = spec_.GetBool("case_sensitive_filter");
      auto subdirs = ListDir(root_, true, dir_filters, cs);
      for (size_t d = 0; d < subdirs.size(); d++)
        for (auto &f : ListDir(root_ + "/" + subdirs[d], false, filters, cs)) entries_.push_back({subdirs[d] + "/" + f, (int)d});
    }
    DALI_ENFORCE(!entries_.empty(), "No files found.");
  }

  std::string root_;
  std::vector<std::pair<std::string, int>> entries_;
};
DALI_REGISTER_OPERATOR(readers__File, FileReaderOp, CPU);
DALI_REGISTER_OPERATOR(FileReader, FileReaderOp, CPU);

// =============================================================================================
// CPU -> GPU copy (`.gpu()`): pinned host tensor list -> device, asynchronous on the pipeline stream
// (the reference's __Copy_CpuToGpu_ nodes, pipeline.cc:805-810)
// =============================================================================================
DALI_SCHEMA(_CopyToGpu).DocStr("Copies a CPU batch to the GPU.").NumInput(1).NumOutput(1).MakeInternal();

class CopyToGpuOp : public OperatorBase {
 public:
  explicit CopyToGpuOp(const OpSpec &spec) : OperatorBase(spec) {}
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    desc[0].type = in.type();
    desc[0].shape.clear();
    for (int i = 0; i < in.num_samples(); i++) desc[0].shape.push_back(in.shape(i));
    // A list that is a VIEW of one host block (decoders.audio over the files of a batch: every sample is the data chunk of
    // a file the reader put into one page-locked block) goes over as that block, in ONE transfer - 64 transfers of 0.4 MB
    // run at a third of the rate of one of 25 MB - and the output views the device copy the same way.
    block_ = false;
    const int n = in.num_samples();
    if (n > 1) {
      uintptr_t lo = UINTPTR_MAX, hi = 0;
      size_t sum = 0;
      bool all_ext = true;
      for (int i = 0; i < n && all_ext; i++) {
        all_ext = in.is_external(i) && in.row_pitch(i) == 0;
        const uintptr_t p = reinterpret_cast<uintptr_t>(in.raw(i));
        lo = std::min(lo, p);
        hi = std::max(hi, p + in.nbytes(i));
        sum += in.nbytes(i);
      }
      if (all_ext && hi - lo <= sum + sum / 4 + (64 << 10)) {
        block_ = true;
        block_lo_ = lo;
        block_bytes_ = hi - lo;
      }
    }
    return !block_;
  }
  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    if (block_) {
      if ((int)blocks_.size() < ws.ring) blocks_.resize(ws.ring);
      auto &blk = blocks_[ws.iteration % ws.ring];   // (its last reader finished before this ring slot came round again)
      if (!blk) blk = std::make_shared<Buffer>(StorageDevice::GPU);
      blk->Reserve(block_bytes_ + 512);
      // (device addresses congruent to the host's modulo 256: whatever alignment the consumer's loads found there holds here)
      uint8_t *dst = static_cast<uint8_t *>(blk->data()) + (block_lo_ & 255);
      const int n = in.num_samples();
      std::vector<TensorShape> shapes(n);
      std::vector<void *> ptrs(n);
      for (int i = 0; i < n; i++) {
        shapes[i] = in.shape(i);
        ptrs[i] = dst + (reinterpret_cast<uintptr_t>(in.raw(i)) - block_lo_);
      }
      out.Resize(shapes, in.type(), 1, ptrs, std::vector<int64_t>(n, 0), blk);
      out.SetLayout(in.layout());
      out.source_info = in.source_info;
      KCHECK(daliamdMemcpyH2DAsync(dst, reinterpret_cast<const void *>(block_lo_), block_bytes_, ws.stream));
      NoteLaunch(ws, "h2d_copy");
      return;
    }
    out.SetLayout(in.layout());
    out.source_info = in.source_info;
    // both lists use the same 256-byte-aligned packing: one copy
    if (in.num_samples() > 0 && in.total_bytes() == out.total_bytes()) {
      KCHECK(daliamdMemcpyH2DAsync(out.raw(0), in.raw(0), in.total_bytes(), ws.stream));
    } else {
      for (int i = 0; i < in.num_samples(); i++)
        KCHECK(daliamdMemcpyH2DAsync(out.raw(i), in.raw(i), in.nbytes(i), ws.stream));
    }
    NoteLaunch(ws, "h2d_copy");
  }

 private:
  bool block_ = false;
  uintptr_t block_lo_ = 0;
  size_t block_bytes_ = 0;
  std::vector<std::shared_ptr<Buffer>> blocks_;   // one device copy of the host block per ring slot
};
DALI_REGISTER_OPERATOR(_CopyToGpu, CopyToGpuOp, MIXED);

}  // namespace daliamd_host
