#include "framework.h"
#include <time.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>

namespace daliamd_host {

// ------------------------------------------------------------------------------------------ types
int TypeSize(DALIDataType t) {
  switch (t) {
    case DALI_UINT8: case DALI_INT8: case DALI_BOOL: return 1;
    case DALI_UINT16: case DALI_INT16: case DALI_FLOAT16: return 2;
    case DALI_UINT32: case DALI_INT32: case DALI_FLOAT: return 4;
    case DALI_UINT64: case DALI_INT64: case DALI_FLOAT64: return 8;
    default: DALI_FAIL("Unknown data type ", (int)t);
  }
}

const char *TypeName(DALIDataType t) {
  switch (t) {
    case DALI_UINT8: return "uint8"; case DALI_UINT16: return "uint16"; case DALI_UINT32: return "uint32";
    case DALI_UINT64: return "uint64"; case DALI_INT8: return "int8"; case DALI_INT16: return "int16";
    case DALI_INT32: return "int32"; case DALI_INT64: return "int64"; case DALI_FLOAT16: return "float16";
    case DALI_FLOAT: return "float"; case DALI_FLOAT64: return "double"; case DALI_BOOL: return "bool";
    default: return "<no type>";
  }
}

int ToKernelDType(DALIDataType t) {
  switch (t) {
    case DALI_UINT8: return DALIAMD_UINT8;
    case DALI_FLOAT16: return DALIAMD_FLOAT16;
    case DALI_FLOAT: return DALIAMD_FLOAT;
    case DALI_INT8: return DALIAMD_INT8;
    default: DALI_FAIL("Data type ", TypeName(t), " is not supported by the device kernels. Supported types are: "
                       "uint8, int8, float16, float");
  }
}

OpType ParseOpType(const std::string &device) {
  if (device == "cpu") return OpType::CPU;
  if (device == "gpu") return OpType::GPU;
  if (device == "mixed") return OpType::MIXED;
  DALI_FAIL("Invalid device \"", device, "\". Valid options are \"cpu\", \"gpu\" or \"mixed\"");
}
const char *OpTypeName(OpType t) { return t == OpType::CPU ? "cpu" : t == OpType::GPU ? "gpu" : "mixed"; }

// ------------------------------------------------------------------------------------------ Buffer
Buffer::~Buffer() {
  if (!ptr_) return;
  if (dev_ == StorageDevice::GPU) daliamdFree(ptr_);
  else if (pinned_) daliamdHostFree(ptr_);
  else free(ptr_);
}

void Buffer::Reserve(size_t bytes) {
  if (bytes <= cap_) return;
  size_t want = std::max(bytes, cap_ + cap_ / 10);  // growth factor 1.1
  want = (want + 4095) & ~(size_t)4095;
  void *p = nullptr;
  bool pinned = false;
  if (dev_ == StorageDevice::GPU) {
    KCHECK(daliamdMalloc(&p, want));
  } else {
    // pinned when a device is present (asynchronous H2D); plain memory on GPU-less hosts
    static const bool have_gpu = [] { int n = 0; daliamdDeviceCount(&n); return n > 0; }();
    if (have_gpu && daliamdHostAlloc(&p, want) == DALIAMD_SUCCESS) pinned = true;
    else {
      daliamdClearLastError();
      if (posix_memalign(&p, 256, want) != 0) DALI_FAIL("Out of host memory allocating ", want, " bytes");
    }
  }
  // contents are not preserved: every producer rewrites its output each iteration
  if (ptr_) {
    if (dev_ == StorageDevice::GPU) daliamdFree(ptr_);
    else if (pinned_) daliamdHostFree(ptr_);
    else free(ptr_);
  }
  ptr_ = p; cap_ = want; pinned_ = pinned;
}

// ------------------------------------------------------------------------------------------ TensorList
void TensorList::Resize(const std::vector<TensorShape> &shapes, DALIDataType type, int pitch_align) {
  Resize(shapes, type, pitch_align, {}, {}, nullptr);
}

void TensorList::Resize(const std::vector<TensorShape> &shapes, DALIDataType type, int pitch_align,
                        const std::vector<void *> &ext_ptr, const std::vector<int64_t> &ext_pitch,
                        std::shared_ptr<void> keepalive) {
  type_ = type;
  shapes_ = shapes;
  int n = (int)shapes.size();
  offsets_.assign(n, 0); pitch_.assign(n, 0); sizes_.assign(n, 0);
  bool any_ext = false;
  for (void *p : ext_ptr) any_ext |= p != nullptr;
  if (any_ext) { ext_ = ext_ptr; ext_.resize(n, nullptr); ext_owner_ = std::move(keepalive); }
  else { ext_.clear(); ext_owner_.reset(); }
  ext_device_visible_ = false;
  size_t off = 0;
  int esz = TypeSize(type);
  for (int i = 0; i < n; i++) {
    const auto &s = shapes[i];
    size_t bytes;
    if (any_ext && ext_[i]) {
      pitch_[i] = ext_pitch[i];
      sizes_[i] = s.size() == 3 && ext_pitch[i] ? (size_t)(ext_pitch[i] * s[0]) : (size_t)volume(s) * esz;
      offsets_[i] = (int64_t)off;
      continue;
    }
    if (pitch_align > 1 && s.size() == 3) {
      int64_t row = s[1] * s[2] * esz;
      int64_t p = (row + pitch_align - 1) / pitch_align * pitch_align;
      pitch_[i] = p;
      bytes = (size_t)(p * s[0]);
    } else {
      bytes = (size_t)volume(s) * esz;
    }
    offsets_[i] = (int64_t)off;
    sizes_[i] = bytes;
    off += (bytes + 255) & ~(size_t)255;
  }
  total_ = off;
  if (!buf_ || buf_.use_count() > 1) buf_ = std::make_shared<Buffer>(dev_);  // never resize shared storage
  buf_->Reserve(std::max<size_t>(std::max(off, min_reserve_), 256));
  deferred.reset();
  deferred_pointwise.reset();
  deferred_audio.reset();
  deferred_blur.reset();
}

bool TensorList::is_dense() const {
  for (void *p : ext_) if (p) return false;
  for (size_t i = 0; i < shapes_.size(); i++) {
    if (pitch_[i] && shapes_[i].size() == 3 && pitch_[i] != shapes_[i][1] * shapes_[i][2] * TypeSize(type_)) return false;
  }
  return true;
}

void TensorList::ShareData(const TensorList &o) {
  buf_ = o.buf_; dev_ = o.dev_; type_ = o.type_; layout_ = o.layout_; shapes_ = o.shapes_;
  offsets_ = o.offsets_; pitch_ = o.pitch_; sizes_ = o.sizes_; total_ = o.total_;
  ext_ = o.ext_; ext_owner_ = o.ext_owner_; ext_device_visible_ = o.ext_device_visible_;
  deferred = o.deferred; deferred_pointwise = o.deferred_pointwise; deferred_audio = o.deferred_audio;
  deferred_blur = o.deferred_blur; source_info = o.source_info;
}

// ------------------------------------------------------------------------------------------ ThreadPool
void NameThisThread(const std::string &name) { pthread_setname_np(pthread_self(), name.substr(0, 15).c_str()); }

ThreadPool::ThreadPool(int n, const std::vector<int> &cpus, const char *name) {
  n = std::max(1, n);
  const std::string base = name;
  for (int i = 0; i < n; i++)
    threads_.emplace_back([this, i, cpus, base] {
      NameThisThread(base + std::to_string(i));
      BindThisThread(cpus);
      Loop(i);
    });
}

std::vector<int> ParseCpuList(const std::string &list) {
  std::vector<int> out;
  size_t pos = 0;
  while (pos < list.size()) {
    size_t end = list.find(',', pos);
    if (end == std::string::npos) end = list.size();
    std::string tok = list.substr(pos, end - pos);
    pos = end + 1;
    int a = 0, b = 0;
    if (sscanf(tok.c_str(), "%d-%d", &a, &b) == 2) {
      for (int c = a; c <= b && c < 4096; c++) out.push_back(c);
    } else if (sscanf(tok.c_str(), "%d", &a) == 1) {
      out.push_back(a);
    }
  }
  return out;
}

void BindThisThread(const std::vector<int> &cpus) {
  if (cpus.empty()) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus)
    if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
  pthread_setaffinity_np(pthread_self(), sizeof(set), &set);  // best effort
}

std::vector<int> DeviceLocalCpus(int device_id) {
  std::vector<int> local;
  if (const char *env = getenv("DALI_AFFINITY_MASK")) {
    local = ParseCpuList(env);
  } else {
    char bus[64] = "";
    if (daliamdDevicePciBusId(device_id, bus, sizeof(bus)) != DALIAMD_SUCCESS) return {};
    std::string id = bus;
    for (auto &ch : id) ch = (char)tolower(ch);
    std::ifstream f("/sys/bus/pci/devices/" + id + "/local_cpulist");
    std::string line;
    if (!f || !std::getline(f, line)) return {};
    local = ParseCpuList(line);
  }
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return {};
  std::vector<int> out;
  for (int c : local)
    if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) out.push_back(c);
  return out;
}
ThreadPool::~ThreadPool() {
  {
    std::lock_guard<std::mutex> g(m_);
    stop_ = true;
  }
  cv_work_.notify_all();
  for (auto &t : threads_) t.join();
}
void ThreadPool::AddWork(Work w, int64_t priority) { pending_.emplace_back(priority, std::move(w)); }

void ThreadPool::RunAll() {
  if (pending_.empty()) return;
  std::stable_sort(pending_.begin(), pending_.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
  auto batch = std::make_shared<Batch>();
  batch->tasks = std::move(pending_);
  pending_.clear();
  {
    std::lock_guard<std::mutex> g(m_);
    batch_ = batch;
    generation_++;
  }
  // wake no more workers than the batch can feed (a worker woken from a deep idle state for one 10 us task costs more
  // than it contributes), and work on the batch from this thread as well: its share needs no wake-up at all
  const size_t total = batch->tasks.size();
  // (this thread is one of the `active` ones, so the pool never runs more threads than it was sized for)
  const size_t active = std::min(threads_.size(), (total + 3) / 4);
  for (size_t i = 0; i + 1 < active; i++) cv_work_.notify_one();
  Drain(*batch, (int)threads_.size());
  {
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return batch->done.load(std::memory_order_acquire) == total; });
    batch_.reset();
  }
  if (!batch->errors.empty()) throw std::runtime_error(batch->errors.front());
}

void ThreadPool::Drain(Batch &batch, int tid) {
  const size_t total = batch.tasks.size();
  for (;;) {
    const size_t idx = batch.next.fetch_add(1, std::memory_order_relaxed);
    if (idx >= total) break;
    std::string err;
    try { batch.tasks[idx].second(tid); } catch (const std::exception &e) { err = e.what(); } catch (...) { err = "unknown error"; }
    if (!err.empty()) {
      std::lock_guard<std::mutex> g(batch.err_m);
      batch.errors.push_back(err);
    }
    if (batch.done.fetch_add(1, std::memory_order_acq_rel) + 1 == total) {
      std::lock_guard<std::mutex> g(m_);   // the waiter checks the counter under this lock: no lost wake-up
      cv_done_.notify_all();
    }
  }
}

void ThreadPool::Loop(int tid) {
  uint64_t seen = 0;
  for (;;) {
    std::shared_ptr<Batch> batch;
    {
      std::unique_lock<std::mutex> lk(m_);
      cv_work_.wait(lk, [&] { return stop_ || (generation_ != seen && batch_); });
      if (stop_) return;
      seen = generation_;
      batch = batch_;
    }
    Drain(*batch, tid);
  }
}

// ------------------------------------------------------------------------------------------ schema
const char *ArgTypeName(ArgType t) {
  switch (t) {
    case ArgType::INT: return "int"; case ArgType::FLOAT: return "float"; case ArgType::BOOL: return "bool";
    case ArgType::STRING: return "str"; case ArgType::INT_VEC: return "int or list of int";
    case ArgType::FLOAT_VEC: return "float or list of float"; case ArgType::STRING_VEC: return "str or list of str";
    default: return "none";
  }
}

OpSchema &OpSchema::AddArg(const std::string &name, const std::string &doc, ArgType type, bool tensor_ok) {
  ArgDef d; d.name = name; d.doc = doc; d.type = type; d.required = true; d.tensor_ok = tensor_ok;
  args_.push_back(d);
  return *this;
}
OpSchema &OpSchema::AddOptionalArg(const std::string &name, const std::string &doc, ArgValue def, bool tensor_ok) {
  ArgDef d; d.name = name; d.doc = doc; d.type = def.type; d.def = std::move(def); d.tensor_ok = tensor_ok;
  args_.push_back(d);
  return *this;
}
OpSchema &OpSchema::AddOptionalTypeArg(const std::string &name, const std::string &doc, ArgType type, bool tensor_ok) {
  ArgDef d; d.name = name; d.doc = doc; d.type = type; d.tensor_ok = tensor_ok;  // def.type == NONE: absent
  args_.push_back(d);
  return *this;
}
OpSchema &OpSchema::DeprecateArg(const std::string &name) {
  for (auto &a : args_) if (a.name == name) a.deprecated = true;
  return *this;
}
OpSchema &OpSchema::AddRandomSeedArg() {
  return AddOptionalTypeArg("seed", "Random seed; if not set, one will be assigned automatically.", ArgType::INT);
}
bool OpSchema::HasRandomSeedArg() const { return FindArg("seed") != nullptr; }

std::vector<ArgDef> OpSchema::AllArgs() const {
  std::vector<ArgDef> out = args_;
  for (auto &p : parents_) {
    for (auto &a : SchemaRegistry::GetSchema(p).AllArgs()) {
      bool dup = false;
      for (auto &o : out) dup |= o.name == a.name;
      if (!dup) out.push_back(a);
    }
  }
  return out;
}
const ArgDef *OpSchema::FindArg(const std::string &name) const {
  for (auto &a : args_) if (a.name == name) return &a;
  for (auto &p : parents_) {
    if (auto *s = SchemaRegistry::TryGetSchema(p))
      if (auto *a = s->FindArg(name)) return a;
  }
  return nullptr;
}
const std::vector<std::string> *OpSchema::InputLayouts(int idx) const {
  auto it = in_layouts_.find(idx);
  return it == in_layouts_.end() ? nullptr : &it->second;
}

static std::map<std::string, std::unique_ptr<OpSchema>> &Schemas() {
  static std::map<std::string, std::unique_ptr<OpSchema>> m;
  return m;
}
static std::mutex &RegistryMutex() { static std::mutex m; return m; }

OpSchema &SchemaRegistry::RegisterSchema(const std::string &name) {
  std::lock_guard<std::mutex> g(RegistryMutex());
  auto &m = Schemas();
  DALI_ENFORCE(!m.count(name), "OpSchema already registered for operator '", name,
               "'. DALI_SCHEMA(op) should only be called once per op.");
  m[name] = std::make_unique<OpSchema>(name);
  return *m[name];
}
const OpSchema *SchemaRegistry::TryGetSchema(const std::string &name) {
  std::lock_guard<std::mutex> g(RegistryMutex());
  auto it = Schemas().find(name);
  return it == Schemas().end() ? nullptr : it->second.get();
}
const OpSchema &SchemaRegistry::GetSchema(const std::string &name) {
  auto *s = TryGetSchema(name);
  DALI_ENFORCE(s, "Schema for operator '", name, "' not registered");
  return *s;
}
std::vector<std::string> SchemaRegistry::Names() {
  std::lock_guard<std::mutex> g(RegistryMutex());
  std::vector<std::string> out;
  for (auto &kv : Schemas()) out.push_back(kv.first);
  return out;
}

// ------------------------------------------------------------------------------------------ OpSpec
const ArgValue *OpSpec::TryArg(const std::string &name) const {
  auto it = args_.find(name);
  if (it != args_.end()) return &it->second;
  if (auto *s = SchemaRegistry::TryGetSchema(name_)) {
    const ArgDef *d = s->FindArg(name);
    if (d && d->def.type != ArgType::NONE) return &d->def;
  }
  return nullptr;
}
const ArgValue &OpSpec::Arg(const std::string &name) const {
  const ArgValue *a = TryArg(name);
  DALI_ENFORCE(a, "Argument \"", name, "\" of operator \"", name_, "\" was not provided and has no default value");
  return *a;
}
int64_t OpSpec::GetInt(const std::string &name) const {
  const ArgValue &a = Arg(name);
  if (a.type == ArgType::INT || a.type == ArgType::BOOL) return a.i;
  if (a.type == ArgType::FLOAT && a.f == (int64_t)a.f) return (int64_t)a.f;
  if (a.type == ArgType::INT_VEC && a.iv.size() == 1) return a.iv[0];
  DALI_FAIL("Argument \"", name, "\" of operator \"", name_, "\" must be an integer");
}
double OpSpec::GetFloat(const std::string &name) const {
  const ArgValue &a = Arg(name);
  if (a.type == ArgType::FLOAT) return a.f;
  if (a.type == ArgType::INT || a.type == ArgType::BOOL) return (double)a.i;
  if (a.type == ArgType::FLOAT_VEC && a.fv.size() == 1) return a.fv[0];
  if (a.type == ArgType::INT_VEC && a.iv.size() == 1) return (double)a.iv[0];
  DALI_FAIL("Argument \"", name, "\" of operator \"", name_, "\" must be a number");
}
bool OpSpec::GetBool(const std::string &name) const { return GetInt(name) != 0; }
std::string OpSpec::GetString(const std::string &name) const {
  const ArgValue &a = Arg(name);
  DALI_ENFORCE(a.type == ArgType::STRING, "Argument \"", name, "\" of operator \"", name_, "\" must be a string");
  return a.s;
}
std::vector<int64_t> OpSpec::GetIntVec(const std::string &name) const {
  const ArgValue &a = Arg(name);
  if (a.type == ArgType::INT_VEC) return a.iv;
  if (a.type == ArgType::INT || a.type == ArgType::BOOL) return {a.i};
  if (a.type == ArgType::FLOAT_VEC) {
    std::vector<int64_t> v;
    for (double f : a.fv) { DALI_ENFORCE(f == (int64_t)f, "Argument \"", name, "\" must hold integers"); v.push_back((int64_t)f); }
    return v;
  }
  if (a.type == ArgType::FLOAT && a.f == (int64_t)a.f) return {(int64_t)a.f};
  DALI_FAIL("Argument \"", name, "\" of operator \"", name_, "\" must be an integer or a list of integers");
}
std::vector<double> OpSpec::GetFloatVec(const std::string &name) const {
  const ArgValue &a = Arg(name);
  if (a.type == ArgType::FLOAT_VEC) return a.fv;
  if (a.type == ArgType::FLOAT) return {a.f};
  if (a.type == ArgType::INT || a.type == ArgType::BOOL) return {(double)a.i};
  if (a.type == ArgType::INT_VEC) return std::vector<double>(a.iv.begin(), a.iv.end());
  DALI_FAIL("Argument \"", name, "\" of operator \"", name_, "\" must be a number or a list of numbers");
}

static bool Convertible(ArgType have, ArgType want) {
  if (have == want) return true;
  auto num = [](ArgType t) { return t == ArgType::INT || t == ArgType::FLOAT || t == ArgType::BOOL; };
  auto numvec = [](ArgType t) { return t == ArgType::INT_VEC || t == ArgType::FLOAT_VEC; };
  if (num(have) && (num(want) || numvec(want))) return true;
  if (numvec(have) && numvec(want)) return true;
  if (have == ArgType::STRING && want == ArgType::STRING_VEC) return true;
  return false;
}

void OpSpec::Validate() const {
  const OpSchema &s = GetSchema();
  static const char *injected[] = {"device", "max_batch_size", "num_threads", "device_id", "preserve", "name",
                                   "bytes_per_sample_hint", "checkpointing", "gpu_prefetch_queue_depth",
                                   "cpu_prefetch_queue_depth"};
  for (auto &kv : args_) {
    bool inj = false;
    for (auto *i : injected) inj |= kv.first == i;
    if (inj) continue;
    const ArgDef *d = s.FindArg(kv.first);
    DALI_ENFORCE(d, "Operator \"", name_, "\" got an unexpected argument \"", kv.first, "\"");
    DALI_ENFORCE(Convertible(kv.second.type, d->type), "Argument \"", kv.first, "\" of operator \"", name_,
                 "\" has type ", ArgTypeName(kv.second.type), " but ", ArgTypeName(d->type), " was expected");
  }
  for (auto &kv : arg_inputs_) {
    const ArgDef *d = s.FindArg(kv.first);
    DALI_ENFORCE(d, "Operator \"", name_, "\" got an unexpected argument \"", kv.first, "\"");
    DALI_ENFORCE(d->tensor_ok, "Argument \"", kv.first, "\" of operator \"", name_,
                 "\" cannot be provided as a tensor (per-sample) argument");
  }
  for (auto &d : s.AllArgs())
    DALI_ENFORCE(!d.required || ArgumentDefined(d.name), "Operator \"", name_, "\" is missing the required argument \"",
                 d.name, "\"");
  int nin = (int)inputs_.size();
  DALI_ENFORCE(nin >= s.MinNumInput() && nin <= s.MaxNumInput(), "Operator \"", name_, "\" expects between ",
               s.MinNumInput(), " and ", s.MaxNumInput(), " inputs, but received ", nin);
}

// ------------------------------------------------------------------------------------------ operators
const TensorList &Workspace::ArgumentInput(const std::string &name) const {
  auto it = argument_inputs.find(name);
  DALI_ENFORCE(it != argument_inputs.end(), "Argument input \"", name, "\" not found in the workspace");
  return *it->second;
}

OperatorBase::OperatorBase(const OpSpec &spec) : spec_(spec) {
  num_threads_ = (int)spec.GetInt("num_threads");
  max_batch_size_ = (int)spec.GetInt("max_batch_size");
  device_id_ = (int)spec.GetInt("device_id");
}

using RegKey = std::pair<std::string, int>;
static std::map<RegKey, OpFactory> &Factories() {
  static std::map<RegKey, OpFactory> m;
  return m;
}
void OperatorRegistry::Register(const std::string &name, OpType type, OpFactory f) {
  std::lock_guard<std::mutex> g(RegistryMutex());
  RegKey k{name, (int)type};
  DALI_ENFORCE(!Factories().count(k), "Operator \"", name, "\" already registered for backend ", OpTypeName(type));
  Factories()[k] = std::move(f);
}
bool OperatorRegistry::IsRegistered(const std::string &name, OpType type) {
  std::lock_guard<std::mutex> g(RegistryMutex());
  return Factories().count({name, (int)type}) != 0;
}
std::vector<OpType> OperatorRegistry::Backends(const std::string &name) {
  std::vector<OpType> out;
  for (OpType t : {OpType::CPU, OpType::GPU, OpType::MIXED})
    if (IsRegistered(name, t)) out.push_back(t);
  return out;
}
std::unique_ptr<OperatorBase> OperatorRegistry::Create(const std::string &name, OpType type, const OpSpec &spec) {
  OpFactory f;
  {
    std::lock_guard<std::mutex> g(RegistryMutex());
    auto it = Factories().find({name, (int)type});
    if (it != Factories().end()) f = it->second;
  }
  if (!f) {
    std::string avail;
    for (OpType t : Backends(name)) avail += std::string(avail.empty() ? "" : ", ") + "\"" + OpTypeName(t) + "\"";
    if (avail.empty()) DALI_FAIL("Operator \"", name, "\" is not registered");
    DALI_FAIL("Operator \"", name, "\" is not available for device \"", OpTypeName(type),
              "\" in this MI355X-native build (registered backends: ", avail, "). There is no CPU fallback for "
              "device operators.");
  }
  return f(spec);
}

std::vector<float> GetPerSampleFloat(const OpSpec &spec, const Workspace &ws, const std::string &name, int n) {
  std::vector<float> out(n);
  if (spec.HasTensorArgument(name)) {
    const TensorList &t = ws.ArgumentInput(name);
    DALI_ENFORCE(t.num_samples() == n, "Argument input \"", name, "\" has ", t.num_samples(), " samples, expected ", n);
    for (int i = 0; i < n; i++) {
      const void *p = t.raw(i);
      switch (t.type()) {
        case DALI_FLOAT: out[i] = *static_cast<const float *>(p); break;
        case DALI_INT32: out[i] = (float)*static_cast<const int32_t *>(p); break;
        case DALI_INT64: out[i] = (float)*static_cast<const int64_t *>(p); break;
        case DALI_UINT8: case DALI_BOOL: out[i] = (float)*static_cast<const uint8_t *>(p); break;
        default: DALI_FAIL("Unsupported type ", TypeName(t.type()), " for argument input \"", name, "\"");
      }
    }
  } else {
    float v = (float)spec.GetFloat(name);
    std::fill(out.begin(), out.end(), v);
  }
  return out;
}
std::vector<int> GetPerSampleInt(const OpSpec &spec, const Workspace &ws, const std::string &name, int n) {
  auto f = GetPerSampleFloat(spec, ws, name, n);
  std::vector<int> out(n);
  for (int i = 0; i < n; i++) out[i] = (int)f[i];
  return out;
}

std::vector<std::vector<float>> GetPerSampleFloatVec(const OpSpec &spec, const Workspace &ws, const std::string &name,
                                                     int n) {
  std::vector<std::vector<float>> out(n);
  if (spec.HasTensorArgument(name)) {
    const TensorList &t = ws.ArgumentInput(name);
    DALI_ENFORCE(t.num_samples() == n, "Argument input \"", name, "\" has ", t.num_samples(), " samples, expected ", n);
    for (int i = 0; i < n; i++) {
      int64_t cnt = volume(t.shape(i));
      out[i].resize(cnt);
      for (int64_t k = 0; k < cnt; k++) {
        switch (t.type()) {
          case DALI_FLOAT: out[i][k] = static_cast<const float *>(t.raw(i))[k]; break;
          case DALI_FLOAT64: out[i][k] = (float)static_cast<const double *>(t.raw(i))[k]; break;
          case DALI_INT32: out[i][k] = (float)static_cast<const int32_t *>(t.raw(i))[k]; break;
          case DALI_INT64: out[i][k] = (float)static_cast<const int64_t *>(t.raw(i))[k]; break;
          default: DALI_FAIL("Unsupported type ", TypeName(t.type()), " for argument input \"", name, "\"");
        }
      }
    }
  } else {
    std::vector<float> v;
    for (double f : spec.GetFloatVec(name)) v.push_back((float)f);
    for (auto &o : out) o = v;
  }
  return out;
}

void SleepWaitEvent(daliamdEvent_t event) {
  static const bool poll = [] { const char *e = getenv("DALI_AMD_OUTPUT_WAIT"); return !(e && std::string(e) == "block"); }();
  if (!poll) {
    KCHECK(daliamdEventSynchronize(event));
    return;
  }
  for (;;) {
    int done = 0;
    KCHECK(daliamdEventQuery(event, &done));
    if (done) return;
    struct timespec ts{0, 50000};
    nanosleep(&ts, nullptr);
  }
}

// ------------------------------------------------------------------------------------------ DescUploader
void *DescUploader::Upload(const void *host, size_t bytes, daliamdStream_t stream, int min_slots, size_t scratch_bytes) {
  if ((int)slots_.size() < min_slots) {  // grow: new (unused) slots go behind the cursor
    slots_.resize(min_slots);
  }
  if (next_ >= (int)slots_.size()) next_ = 0;
  Slot &s = slots_[next_];
  next_ = (next_ + 1) % (int)slots_.size();
  if (s.used) SleepWaitEvent(s.ev);  // the previous copy from this slot must be done
  if (bytes > s.cap) {
    if (s.pinned) daliamdHostFree(s.pinned);
    s.cap = (bytes * 3 / 2 + 4095) & ~(size_t)4095;
    KCHECK(daliamdHostAlloc(&s.pinned, s.cap));
  }
  const size_t table_bytes = (bytes + 255) & ~(size_t)255;
  if (table_bytes + scratch_bytes > s.dev_cap) {
    if (s.dev) daliamdFree(s.dev);
    s.dev_cap = ((table_bytes + scratch_bytes) * 3 / 2 + 4095) & ~(size_t)4095;
    KCHECK(daliamdMalloc(&s.dev, s.dev_cap));
  }
  scratch_ = scratch_bytes ? static_cast<char *>(s.dev) + table_bytes : nullptr;
  if (!s.ev) KCHECK(daliamdEventCreate(&s.ev, 0));
  memcpy(s.pinned, host, bytes);
  KCHECK(daliamdMemcpyH2DAsync(s.dev, s.pinned, bytes, stream));
  KCHECK(daliamdEventRecord(s.ev, stream));
  last_ev_ = s.ev;
  s.used = true;
  return s.dev;
}
DescUploader::~DescUploader() {
  for (auto &s : slots_) {
    if (s.ev) daliamdEventDestroy(s.ev);
    if (s.pinned) daliamdHostFree(s.pinned);
    if (s.dev) daliamdFree(s.dev);
  }
}

}  // namespace daliamd_host
