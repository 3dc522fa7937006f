// C++ host framework of dali_amd: tensor lists, thread pool, operator schema / spec / registry and
// the workspace handed to operators.  It mirrors the reference's drop-in boundary for out-of-tree
// operators at the scale this hot path needs:
//   OpSchema + DALI_SCHEMA            dali/pipeline/operator/op_schema.h:1096-1109
//   OpSpec                            dali/pipeline/operator/op_spec.h
//   OperatorBase / Setup / Run        dali/pipeline/operator/operator.h:76-252
//   DALI_REGISTER_OPERATOR + registry dali/pipeline/operator/operator.h:327-333, operator_factory.h:37-139
//   Workspace                         dali/pipeline/workspace/workspace.h
//   TensorList                        dali/pipeline/data/tensor_list.h
//   ThreadPool::AddWork / RunAll      dali/pipeline/util/thread_pool.h
// Host code never includes HIP headers: the device is reached only through the C ABI in
// include/dali_amd_kernels.h.
#ifndef DALI_AMD_HOST_FRAMEWORK_H_
#define DALI_AMD_HOST_FRAMEWORK_H_

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "dali_amd_kernels.h"

// Everything declared here is the plug-in interface: the library is built with hidden visibility, but an operator
// library loaded later (daliamdLoadLibrary / dali_amd.plugin_manager.load_library; reference
// dali/plugin/plugin_manager.cc:26-41) resolves these classes and registries against libdali_amd_host.so.
#pragma GCC visibility push(default)
namespace daliamd_host {

// ---------------------------------------------------------------------------------------------
// errors (DALI_ENFORCE / DALI_FAIL -> std::runtime_error, dali/core/error_handling.h)
// ---------------------------------------------------------------------------------------------
template <typename... Args>
std::string make_string(const Args &...args) {
  std::ostringstream ss;
  (void)std::initializer_list<int>{(ss << args, 0)...};
  return ss.str();
}
#define DALI_FAIL(...) throw std::runtime_error(::daliamd_host::make_string(__VA_ARGS__))
#define DALI_ENFORCE(cond, ...)                                                                  \
  do {                                                                                           \
    if (!(cond)) throw std::runtime_error(::daliamd_host::make_string("Assert on \"", #cond,      \
                                                                       "\" failed: ", __VA_ARGS__)); \
  } while (0)
// kernel-library call that must succeed
#define KCHECK(expr)                                                                             \
  do {                                                                                           \
    int _rc = (expr);                                                                            \
    if (_rc != 0) DALI_FAIL("device library error ", _rc, ": ", daliamdGetLastErrorMessage());    \
  } while (0)

// ---------------------------------------------------------------------------------------------
// types
// ---------------------------------------------------------------------------------------------
// values follow DALIDataType (include/dali/core/dali_data_type.h) for the types used here
enum DALIDataType : int {
  DALI_NO_TYPE = -1, DALI_UINT8 = 0, DALI_UINT16 = 1, DALI_UINT32 = 2, DALI_UINT64 = 3, DALI_INT8 = 4,
  DALI_INT16 = 5, DALI_INT32 = 6, DALI_INT64 = 7, DALI_FLOAT16 = 8, DALI_FLOAT = 9, DALI_FLOAT64 = 10,
  DALI_BOOL = 11
};
int TypeSize(DALIDataType t);
const char *TypeName(DALIDataType t);
int ToKernelDType(DALIDataType t);  // daliamdDType_t or throws

enum class StorageDevice { CPU = 0, GPU = 1 };
enum class OpType { CPU = 0, GPU = 1, MIXED = 2 };
OpType ParseOpType(const std::string &device);
const char *OpTypeName(OpType t);

using TensorShape = std::vector<int64_t>;
inline int64_t volume(const TensorShape &s) {
  int64_t v = 1;
  for (auto e : s) v *= e;
  return v;
}

// ---------------------------------------------------------------------------------------------
// Buffer + TensorList
// ---------------------------------------------------------------------------------------------
// Grow-only allocation (like the reference's Buffer with DALI_BUFFER_GROWTH_FACTOR): host memory is
// pinned when a GPU is present (so H2D copies are asynchronous), plain otherwise.
class Buffer {
 public:
  explicit Buffer(StorageDevice dev) : dev_(dev) {}
  ~Buffer();
  Buffer(const Buffer &) = delete;
  Buffer &operator=(const Buffer &) = delete;
  void Reserve(size_t bytes);
  void *data() const { return ptr_; }
  size_t capacity() const { return cap_; }
  bool pinned() const { return pinned_; }
  StorageDevice device() const { return dev_; }

 private:
  StorageDevice dev_;
  void *ptr_ = nullptr;
  size_t cap_ = 0;
  bool pinned_ = false;
};

struct DeferredResample;  // see ops_image.cpp: RandomResizedCrop -> CropMirrorNormalize fusion
struct DeferredPointwise;  // see ops_augment.cpp: ColorTwist -> Erase fusion
struct DeferredBlur;       // see ops_augment.cpp: GaussianBlur -> ColorTwist / Erase fusion
struct DeferredAudio;      // see ops_audio.cpp: Spectrogram -> MelFilterBank (-> ToDecibels) fusion

class TensorList {
 public:
  explicit TensorList(StorageDevice dev) : buf_(std::make_shared<Buffer>(dev)), dev_(dev) {}
  StorageDevice device() const { return dev_; }
  int num_samples() const { return (int)shapes_.size(); }
  DALIDataType type() const { return type_; }
  const std::string &layout() const { return layout_; }
  void SetLayout(const std::string &l) { layout_ = l; }
  const TensorShape &shape(int i) const { return shapes_[i]; }
  // bytes between rows of sample i for image-like (HWC) data; dense when 0
  int64_t row_pitch(int i) const { return pitch_[i]; }
  void *raw(int i) const { return ext_.empty() || !ext_[i] ? static_cast<char *>(buf_->data()) + offsets_[i] : ext_[i]; }
  size_t nbytes(int i) const { return sizes_[i]; }
  bool is_dense() const;
  size_t total_bytes() const { return total_; }
  size_t min_reserve_ = 0;

  // Allocates one contiguous block; every sample starts at a 256-byte boundary.  For 3-D u8 samples
  // `pitch_align` > 1 pads each row to that many bytes (internal hand-off between device operators).
  // Storage is never smaller than this from the next Resize on (a producer whose sample sizes change from iteration to
  // iteration - decoded crop windows - names its upper bound once instead of growing, i.e. re-allocating device memory,
  // whenever a batch sets a new record)
  void SetMinReserve(size_t bytes) { if (bytes > min_reserve_) min_reserve_ = bytes; }
  void Resize(const std::vector<TensorShape> &shapes, DALIDataType type, int pitch_align = 1);
  // Same, but sample i lives in caller-owned memory when ext_ptr[i] != nullptr (row pitch ext_pitch[i]): no space is
  // set aside for it in the block and raw(i) returns that pointer.  `keepalive` owns the external memory (the decoded
  // image cache hands out views of its entries this way - no copy on a hit).
  void Resize(const std::vector<TensorShape> &shapes, DALIDataType type, int pitch_align, const std::vector<void *> &ext_ptr,
              const std::vector<int64_t> &ext_pitch, std::shared_ptr<void> keepalive);
  bool is_external(int i) const { return !ext_.empty() && ext_[i]; }
  bool has_external() const { for (void *p : ext_) if (p) return true; return false; }
  // external CPU samples that the device can read where they are (host memory registered with the device at the same
  // address: readers.file hands out its registered file mappings this way; daliamdHostRegister)
  bool ext_device_visible() const { return ext_device_visible_; }
  void SetExtDeviceVisible(bool v) { ext_device_visible_ = v; }
  // the one block behind the samples (raw(i) - base() is sample i's offset) and whether it is page-locked: a device
  // operator may then transfer straight from it instead of staging a copy
  const void *base() const { return buf_->data(); }
  // holders of the storage block (this list included): 1 = nobody else reads it any more
  long buffer_use_count() const { return buf_.use_count(); }
  bool pinned() const { return buf_->pinned(); }
  // shares storage and metadata (zero-copy pass-through)
  void ShareData(const TensorList &other);
  // metadata only; used when an operator's work is deferred to its consumer
  std::shared_ptr<DeferredResample> deferred;
  std::shared_ptr<DeferredPointwise> deferred_pointwise;
  std::shared_ptr<DeferredAudio> deferred_audio;
  std::shared_ptr<DeferredBlur> deferred_blur;

  // source info (readers): file name per sample, used in error messages like the reference's
  std::vector<std::string> source_info;

 private:
  std::shared_ptr<Buffer> buf_;
  StorageDevice dev_;
  DALIDataType type_ = DALI_NO_TYPE;
  std::string layout_;
  std::vector<TensorShape> shapes_;
  std::vector<int64_t> offsets_, pitch_;
  std::vector<size_t> sizes_;
  size_t total_ = 0;
  std::vector<void *> ext_;
  std::shared_ptr<void> ext_owner_;
  bool ext_device_visible_ = false;
};

// ---------------------------------------------------------------------------------------------
// ThreadPool (AddWork(fn(thread_id), priority) + RunAll(), dali/pipeline/util/thread_pool.h)
// ---------------------------------------------------------------------------------------------
class ThreadPool {
 public:
  // cpus: when not empty, every worker is bound to this CPU set (Pipeline(set_affinity=True): the cores local to the
  // GPU's NUMA node; the reference binds through NVML, dali/pipeline/util/thread_pool.cc)
  // name: the workers show up as <name><index> in /proc/<pid>/task/*/comm (15 characters: tools, bench.py's CPU account)
  explicit ThreadPool(int num_threads, const std::vector<int> &cpus = {}, const char *name = "dali-pool");
  ~ThreadPool();
  using Work = std::function<void(int)>;   // argument: worker index in [0, NumThreads()]; NumThreads() = the thread that called RunAll
  void AddWork(Work w, int64_t priority = 0);
  // runs everything added so far (highest priority first) and waits; rethrows the first exception
  void RunAll();
  // drops work that was added but never run (an operator threw between AddWork and RunAll: the closures refer to
  // its dead stack frame and must not execute in the next RunAll)
  void Discard() { pending_.clear(); }
  int NumThreads() const { return (int)threads_.size(); }

 private:
  // One RunAll = one Batch.  Workers pick it up under the lock when the generation changes and then drain it with
  // atomic counters only: a batch of 256 tasks of 20 us each must not serialise on a mutex per task.  A worker that
  // is late only ever touches the batch it picked up (kept alive by the shared_ptr).
  struct Batch {
    std::vector<std::pair<int64_t, Work>> tasks;
    std::atomic<size_t> next{0}, done{0};
    std::mutex err_m;
    std::vector<std::string> errors;
  };
  void Loop(int tid);
  void Drain(Batch &batch, int tid);
  std::vector<std::thread> threads_;
  std::vector<std::pair<int64_t, Work>> pending_;
  std::shared_ptr<Batch> batch_;
  uint64_t generation_ = 0;
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  bool stop_ = false;
};

// CPUs local to a device: /sys/bus/pci/devices/<bus id>/local_cpulist intersected with the CPUs this process may
// run on (empty when unknown: callers then leave the affinity alone).  DALI_AFFINITY_MASK ("0,1,4-7") overrides,
// as in the reference.
std::vector<int> DeviceLocalCpus(int device_id);
std::vector<int> ParseCpuList(const std::string &list);
// binds the calling thread to `cpus` (no-op for an empty set)
void BindThisThread(const std::vector<int> &cpus);
// names the calling thread (pthread_setname_np; truncated to 15 characters)
void NameThisThread(const std::string &name);

// ---------------------------------------------------------------------------------------------
// Arguments, OpSchema, OpSpec
// ---------------------------------------------------------------------------------------------
enum class ArgType { INT, FLOAT, BOOL, STRING, INT_VEC, FLOAT_VEC, STRING_VEC, NONE };

struct ArgValue {
  ArgType type = ArgType::NONE;
  int64_t i = 0;
  double f = 0;
  std::string s;
  std::vector<int64_t> iv;
  std::vector<double> fv;
  std::vector<std::string> sv;
  static ArgValue Int(int64_t v) { ArgValue a; a.type = ArgType::INT; a.i = v; return a; }
  static ArgValue Float(double v) { ArgValue a; a.type = ArgType::FLOAT; a.f = v; return a; }
  static ArgValue Bool(bool v) { ArgValue a; a.type = ArgType::BOOL; a.i = v; return a; }
  static ArgValue Str(std::string v) { ArgValue a; a.type = ArgType::STRING; a.s = std::move(v); return a; }
  static ArgValue IntVec(std::vector<int64_t> v) { ArgValue a; a.type = ArgType::INT_VEC; a.iv = std::move(v); return a; }
  static ArgValue FloatVec(std::vector<double> v) { ArgValue a; a.type = ArgType::FLOAT_VEC; a.fv = std::move(v); return a; }
  static ArgValue StrVec(std::vector<std::string> v) { ArgValue a; a.type = ArgType::STRING_VEC; a.sv = std::move(v); return a; }
};
const char *ArgTypeName(ArgType t);

struct ArgDef {
  std::string name, doc;
  ArgType type = ArgType::NONE;
  bool required = false;
  bool tensor_ok = false;   // may be supplied per sample through an argument input
  bool deprecated = false;
  ArgValue def;
};

class OpSchema {
 public:
  explicit OpSchema(std::string name) : name_(std::move(name)) {}
  OpSchema &DocStr(std::string d) { doc_ = std::move(d); return *this; }
  OpSchema &NumInput(int n) { min_in_ = max_in_ = n; return *this; }
  OpSchema &NumInput(int lo, int hi) { min_in_ = lo; max_in_ = hi; return *this; }
  OpSchema &NumOutput(int n) { num_out_ = n; return *this; }
  OpSchema &AddArg(const std::string &name, const std::string &doc, ArgType type, bool tensor_ok = false);
  OpSchema &AddOptionalArg(const std::string &name, const std::string &doc, ArgValue def, bool tensor_ok = false);
  // optional argument without a default (absent unless given), e.g. `seed`, `fill_value`
  OpSchema &AddOptionalTypeArg(const std::string &name, const std::string &doc, ArgType type, bool tensor_ok = false);
  OpSchema &DeprecateArg(const std::string &name);
  OpSchema &AddParent(const std::string &parent) { parents_.push_back(parent); return *this; }
  OpSchema &AddRandomSeedArg();
  OpSchema &InputLayout(int idx, std::vector<std::string> layouts) { in_layouts_[idx] = std::move(layouts); return *this; }
  OpSchema &AllowSequences() { return *this; }
  OpSchema &MakeInternal() { internal_ = true; return *this; }

  const std::string &name() const { return name_; }
  const std::string &doc() const { return doc_; }
  int MinNumInput() const { return min_in_; }
  int MaxNumInput() const { return max_in_; }
  int NumOutput() const { return num_out_; }
  bool IsInternal() const { return internal_; }
  bool HasRandomSeedArg() const;
  // own + inherited
  std::vector<ArgDef> AllArgs() const;
  const ArgDef *FindArg(const std::string &name) const;
  const std::vector<std::string> *InputLayouts(int idx) const;

 private:
  std::string name_, doc_;
  int min_in_ = 0, max_in_ = 0, num_out_ = 1;
  bool internal_ = false;
  std::vector<ArgDef> args_;
  std::vector<std::string> parents_;
  std::map<int, std::vector<std::string>> in_layouts_;
};

class SchemaRegistry {
 public:
  static OpSchema &RegisterSchema(const std::string &name);
  static const OpSchema &GetSchema(const std::string &name);
  static const OpSchema *TryGetSchema(const std::string &name);
  static std::vector<std::string> Names();
};

#define DALI_AMD_CONCAT_(a, b) a##b
#define DALI_AMD_CONCAT(a, b) DALI_AMD_CONCAT_(a, b)
// DALI_SCHEMA(Name).DocStr(..).NumInput(..)...;   (op_schema.h:1096-1109)
#define DALI_SCHEMA(OpName)                                                               \
  static ::daliamd_host::OpSchema &DALI_AMD_CONCAT(schema_reg_, __LINE__) [[maybe_unused]] = \
      ::daliamd_host::SchemaRegistry::RegisterSchema(#OpName)

class OpSpec {
 public:
  OpSpec() = default;
  explicit OpSpec(std::string schema_name) : name_(std::move(schema_name)) {}
  const std::string &SchemaName() const { return name_; }
  const OpSchema &GetSchema() const { return SchemaRegistry::GetSchema(name_); }
  OpSpec &AddArg(const std::string &name, ArgValue v) { args_[name] = std::move(v); return *this; }
  OpSpec &AddInput(const std::string &name, StorageDevice dev) { inputs_.push_back({name, dev}); return *this; }
  OpSpec &AddOutput(const std::string &name, StorageDevice dev) { outputs_.push_back({name, dev}); return *this; }
  OpSpec &AddArgumentInput(const std::string &arg, const std::string &tensor) { arg_inputs_[arg] = tensor; return *this; }

  bool ArgumentDefined(const std::string &name) const { return args_.count(name) || arg_inputs_.count(name); }
  bool HasTensorArgument(const std::string &name) const { return arg_inputs_.count(name) != 0; }
  // explicit value, else schema default, else throws
  const ArgValue &Arg(const std::string &name) const;
  const ArgValue *TryArg(const std::string &name) const;
  int64_t GetInt(const std::string &name) const;
  double GetFloat(const std::string &name) const;
  bool GetBool(const std::string &name) const;
  std::string GetString(const std::string &name) const;
  std::vector<int64_t> GetIntVec(const std::string &name) const;      // scalar promoted to 1-vector
  std::vector<double> GetFloatVec(const std::string &name) const;

  struct IO { std::string name; StorageDevice dev; };
  const std::vector<IO> &Inputs() const { return inputs_; }
  const std::vector<IO> &Outputs() const { return outputs_; }
  const std::map<std::string, std::string> &ArgumentInputs() const { return arg_inputs_; }
  const std::map<std::string, ArgValue> &Args() const { return args_; }
  // checks names/types against the schema; fills nothing
  void Validate() const;

 private:
  std::string name_;
  std::map<std::string, ArgValue> args_;
  std::vector<IO> inputs_, outputs_;
  std::map<std::string, std::string> arg_inputs_;
};

// ---------------------------------------------------------------------------------------------
// Workspace + OperatorBase + registry
// ---------------------------------------------------------------------------------------------
struct OutputDesc {
  std::vector<TensorShape> shape;
  DALIDataType type = DALI_NO_TYPE;
};

class Pipeline;

class Workspace {
 public:
  Pipeline *pipeline = nullptr;
  std::vector<std::shared_ptr<TensorList>> inputs, outputs;
  std::map<std::string, std::shared_ptr<TensorList>> argument_inputs;
  ThreadPool *thread_pool = nullptr;
  OpType backend = OpType::GPU;  // the backend the running operator instance was created for
  daliamdStream_t stream = nullptr;  // device operators enqueue here and must not synchronise
  // Second stream for bulk host->device transfers: a copy issued here for iteration i+1 overlaps the kernels of
  // iteration i on `stream`.  The operator orders the two with an event (record on copy_stream, wait on stream).
  daliamdStream_t copy_stream = nullptr;
  // A third stream for small set-up launches that depend on host data only (descriptor uploads, the resampling tables):
  // they run while `stream` is still busy with the iteration's earlier kernels; same ordering rule (event).
  daliamdStream_t aux_stream = nullptr;
  int ring = 3;                      // iterations that may be in flight (prefetch_queue_depth + 1)
  int batch_size = 0;                // requested (max) batch size of this iteration
  int64_t iteration = 0;
  // Checks that can only be made once the device work of this iteration has finished (e.g. status words written
  // by a kernel).  The pipeline runs them in Outputs(), after waiting for the iteration; a check reports by throwing.
  std::vector<std::function<void()>> *completion_checks = nullptr;
  void AddCompletionCheck(std::function<void()> fn) const {
    if (completion_checks) completion_checks->push_back(std::move(fn));
  }

  const TensorList &Input(int i) const { return *inputs.at(i); }
  TensorList &Output(int i) const { return *outputs.at(i); }
  const TensorList &ArgumentInput(const std::string &name) const;
  bool HasArgumentInput(const std::string &name) const { return argument_inputs.count(name) != 0; }
  int NumInput() const { return (int)inputs.size(); }
  int NumOutput() const { return (int)outputs.size(); }
  int GetInputBatchSize(int i) const { return inputs.at(i)->num_samples(); }
  ThreadPool &GetThreadPool() const { return *thread_pool; }
};

// ReaderMeta (dali/pipeline/operator/operator.h:46-58)
struct ReaderMeta {
  int64_t epoch_size = -1, epoch_size_padded = -1;
  int number_of_shards = -1, shard_id = -1;
  int pad_last_batch = -1, stick_to_shard = -1;
};

class OperatorBase {
 public:
  explicit OperatorBase(const OpSpec &spec);
  virtual ~OperatorBase() = default;
  // returns true when the executor should allocate outputs from `output_desc`
  virtual bool SetupImpl(std::vector<OutputDesc> &output_desc, const Workspace &ws) = 0;
  virtual void RunImpl(Workspace &ws) = 0;
  virtual ReaderMeta GetReaderMeta() const { return {}; }
  // checkpointing (operator.h:186-215): textual state, empty for stateless operators
  virtual std::string SaveState() const { return ""; }
  virtual void RestoreState(const std::string &) {}
  // row pitch alignment requested for the image-like output `idx` (1 = dense)
  virtual int OutputPitchAlign(int) const { return 1; }
  const OpSpec &spec() const { return spec_; }

 protected:
  OpSpec spec_;
  int num_threads_, max_batch_size_, device_id_;
};

using OpFactory = std::function<std::unique_ptr<OperatorBase>(const OpSpec &)>;
class OperatorRegistry {
 public:
  static void Register(const std::string &name, OpType type, OpFactory f);
  static std::unique_ptr<OperatorBase> Create(const std::string &name, OpType type, const OpSpec &spec);
  static bool IsRegistered(const std::string &name, OpType type);
  static std::vector<OpType> Backends(const std::string &name);
};
struct OpRegisterer {
  OpRegisterer(const std::string &name, OpType t, OpFactory f) { OperatorRegistry::Register(name, t, std::move(f)); }
};
// DALI_REGISTER_OPERATOR(Name, Class, CPU|GPU|MIXED)   (operator.h:327-333)
#define DALI_REGISTER_OPERATOR(OpName, OpClass, Device)                                              \
  static ::daliamd_host::OpRegisterer DALI_AMD_CONCAT(op_reg_, __LINE__)(                              \
      #OpName, ::daliamd_host::OpType::Device,                                                        \
      [](const ::daliamd_host::OpSpec &s) -> std::unique_ptr<::daliamd_host::OperatorBase> {          \
        return std::make_unique<OpClass>(s);                                                          \
      })

// per-sample (or broadcast) argument access: value from the spec or from an argument input
// (ArgValue<T>, dali/pipeline/operator/arg_helper.h)
std::vector<float> GetPerSampleFloat(const OpSpec &spec, const Workspace &ws, const std::string &name, int nsamples);
std::vector<int> GetPerSampleInt(const OpSpec &spec, const Workspace &ws, const std::string &name, int nsamples);
// vector-valued argument: per-sample tensors (any length) or one broadcast list
std::vector<std::vector<float>> GetPerSampleFloatVec(const OpSpec &spec, const Workspace &ws, const std::string &name,
                                                     int nsamples);

// Pinned staging + asynchronous upload of descriptor tables, shared by device operators.
// Waits for `event` without keeping the calling thread on a CPU: asks, sleeps 50 us, asks again.  hipEventSynchronize - on
// an event made with hipEventBlockingSync as well - stayed on a CPU for the whole wait on the bench box: the consumer's wait
// for an iteration's device work cost 1.5 ms of the main thread's CPU per 2.9 ms batch for one rank's share of an 8-GPU node
// (40 % of the pipeline's host cost), 0.65 ms per 0.62 ms batch with 16 CPUs; asking and sleeping: 0.03-0.05 ms, the same rate
// (gpurun_out/r05_w; the consumer is `prefetch_queue_depth` iterations behind the producer - the wait usually finds the work
// done).  DALI_AMD_OUTPUT_WAIT=block: hipEventSynchronize as before.
void SleepWaitEvent(daliamdEvent_t event);

class DescUploader {
 public:
  // copies `bytes` to a device buffer that stays valid for `min_slots` further uploads.  Consecutive iterations run
  // on different streams, so the reuse of a buffer is NOT ordered by a stream: callers pass the pipeline's ring
  // (Workspace::ring) + 1, and the executor guarantees that the iteration `ring` steps back has completed.
  // `scratch_bytes` more device bytes are set aside behind the table (256-byte aligned; nothing is copied there):
  // kernel-side scratch that lives and dies with the table's slot; Scratch() returns it for the last upload.
  void *Upload(const void *host, size_t bytes, daliamdStream_t stream, int min_slots = 4, size_t scratch_bytes = 0);
  void *Scratch() const { return scratch_; }
  // the event of the slot the last Upload used (recorded behind its copy): a caller that enqueues more work on the upload's
  // stream and wants the slot guarded by that too records it again behind that work
  daliamdEvent_t LastEvent() const { return last_ev_; }
  ~DescUploader();

 private:
  struct Slot {
    void *pinned = nullptr, *dev = nullptr;
    size_t cap = 0, dev_cap = 0;
    daliamdEvent_t ev = nullptr;
    bool used = false;
  };
  void *scratch_ = nullptr;
  daliamdEvent_t last_ev_ = nullptr;
  std::vector<Slot> slots_ = std::vector<Slot>(4);
  int next_ = 0;
};

}  // namespace daliamd_host
#pragma GCC visibility pop
#endif  // DALI_AMD_HOST_FRAMEWORK_H_
