// Flat C API over the C++ host framework, bound from Python with ctypes (dali_amd/_backend.py).
// It plays the role of the reference's pybind11 bridge (dali/python/backend_impl.cc: Pipeline :2475,
// OpSpec :3345, OpSchema :3445, TensorList :1542/:1876) for the subset the hot path needs.
#include <cstring>

#include "dali_amd_host.h"
#include "host_common.h"
#include <dlfcn.h>

#include "image_cache.h"
#include "pipeline.h"

using namespace daliamd_host;

namespace {
struct PipelineHandle {
  std::unique_ptr<Pipeline> pipe;
  std::vector<std::shared_ptr<TensorList>> outputs;  // held until the next Outputs() call
};

template <typename F>
int Guard(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    return Fail("%s", e.what());
  } catch (...) {
    return Fail("unknown error");
  }
}

int CopyOut(const std::string &s, char *buf, int len) {
  if (!buf || len <= 0) return (int)s.size() + 1;
  if ((int)s.size() + 1 > len) return (int)s.size() + 1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

std::string JsonEscape(const std::string &s) {
  std::string o;
  for (char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\t': o += "\\t"; break;
      default: o += c;
    }
  }
  return o;
}

std::string ArgValueJson(const ArgValue &v) {
  std::ostringstream ss;
  ss.precision(17);
  switch (v.type) {
    case ArgType::INT: ss << v.i; break;
    case ArgType::BOOL: ss << (v.i ? "true" : "false"); break;
    case ArgType::FLOAT: ss << v.f; break;
    case ArgType::STRING: ss << '"' << JsonEscape(v.s) << '"'; break;
    case ArgType::INT_VEC: ss << '['; for (size_t i = 0; i < v.iv.size(); i++) ss << (i ? "," : "") << v.iv[i]; ss << ']'; break;
    case ArgType::FLOAT_VEC: ss << '['; for (size_t i = 0; i < v.fv.size(); i++) ss << (i ? "," : "") << v.fv[i]; ss << ']'; break;
    case ArgType::STRING_VEC:
      ss << '[';
      for (size_t i = 0; i < v.sv.size(); i++) ss << (i ? "," : "") << '"' << JsonEscape(v.sv[i]) << '"';
      ss << ']';
      break;
    default: ss << "null";
  }
  return ss.str();
}
}  // namespace

extern "C" {

#define API __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------- schemas
API int daliamdSchemaList(char *buf, int len) {
  std::string s;
  for (auto &n : SchemaRegistry::Names()) s += n + "\n";
  return CopyOut(s, buf, len);
}

// JSON: {"name":..,"doc":..,"min_inputs":..,"max_inputs":..,"num_outputs":..,"internal":..,"backends":[..],
//        "args":[{"name":..,"doc":..,"type":..,"required":..,"tensor_ok":..,"default":..},..]}
API int daliamdSchemaInfo(const char *name, char *buf, int len) {
  std::string out;
  int rc = Guard([&] {
    const OpSchema &s = SchemaRegistry::GetSchema(name);
    std::ostringstream ss;
    ss << "{\"name\":\"" << s.name() << "\",\"doc\":\"" << JsonEscape(s.doc()) << "\",\"min_inputs\":" << s.MinNumInput()
       << ",\"max_inputs\":" << s.MaxNumInput() << ",\"num_outputs\":" << s.NumOutput()
       << ",\"internal\":" << (s.IsInternal() ? "true" : "false") << ",\"backends\":[";
    bool first = true;
    for (OpType t : OperatorRegistry::Backends(name)) { ss << (first ? "" : ",") << '"' << OpTypeName(t) << '"'; first = false; }
    ss << "],\"args\":[";
    first = true;
    for (auto &a : s.AllArgs()) {
      ss << (first ? "" : ",") << "{\"name\":\"" << a.name << "\",\"doc\":\"" << JsonEscape(a.doc) << "\",\"type\":\""
         << ArgTypeName(a.type) << "\",\"required\":" << (a.required ? "true" : "false") << ",\"tensor_ok\":"
         << (a.tensor_ok ? "true" : "false") << ",\"deprecated\":" << (a.deprecated ? "true" : "false")
         << ",\"default\":" << ArgValueJson(a.def) << "}";
      first = false;
    }
    ss << "]}";
    out = ss.str();
  });
  if (rc) return -1;
  return CopyOut(out, buf, len);
}

// ---------------------------------------------------------------------------------------- plug-ins
// dlopen of an operator library: its static initialisers (DALI_SCHEMA / DALI_REGISTER_OPERATOR of
// dali_amd/host/framework.h) add schemas and factories to the registries (PluginManager::LoadLibrary,
// dali/plugin/plugin_manager.cc:26-41).
API int daliamdLoadLibrary(const char *path, int global_symbols) {
  return Guard([&] {
    DALI_ENFORCE(path && *path, "Failed to load library: empty path");
    void *handle = dlopen(path, (global_symbols ? RTLD_GLOBAL : RTLD_LOCAL) | RTLD_LAZY);
    if (handle == nullptr) {
      const char *err = dlerror();
      DALI_FAIL("Failed to load library ", path, ": ", err ? err : "unknown error");
    }
  });
}

// ---------------------------------------------------------------------------------------- OpSpec
API void *daliamdOpSpecCreate(const char *schema_name) { return new OpSpec(schema_name); }
API void daliamdOpSpecDestroy(void *spec) { delete static_cast<OpSpec *>(spec); }
API void daliamdOpSpecAddArgInt(void *s, const char *n, int64_t v) { static_cast<OpSpec *>(s)->AddArg(n, ArgValue::Int(v)); }
API void daliamdOpSpecAddArgBool(void *s, const char *n, int v) { static_cast<OpSpec *>(s)->AddArg(n, ArgValue::Bool(v != 0)); }
API void daliamdOpSpecAddArgFloat(void *s, const char *n, double v) { static_cast<OpSpec *>(s)->AddArg(n, ArgValue::Float(v)); }
API void daliamdOpSpecAddArgStr(void *s, const char *n, const char *v) { static_cast<OpSpec *>(s)->AddArg(n, ArgValue::Str(v)); }
API void daliamdOpSpecAddArgIntVec(void *s, const char *n, const int64_t *v, int count) {
  static_cast<OpSpec *>(s)->AddArg(n, ArgValue::IntVec(std::vector<int64_t>(v, v + count)));
}
API void daliamdOpSpecAddArgFloatVec(void *s, const char *n, const double *v, int count) {
  static_cast<OpSpec *>(s)->AddArg(n, ArgValue::FloatVec(std::vector<double>(v, v + count)));
}
API void daliamdOpSpecAddArgStrVec(void *s, const char *n, const char *const *v, int count) {
  static_cast<OpSpec *>(s)->AddArg(n, ArgValue::StrVec(std::vector<std::string>(v, v + count)));
}
API void daliamdOpSpecAddInput(void *s, const char *name, int gpu) {
  static_cast<OpSpec *>(s)->AddInput(name, gpu ? StorageDevice::GPU : StorageDevice::CPU);
}
API void daliamdOpSpecAddOutput(void *s, const char *name, int gpu) {
  static_cast<OpSpec *>(s)->AddOutput(name, gpu ? StorageDevice::GPU : StorageDevice::CPU);
}
API void daliamdOpSpecAddArgumentInput(void *s, const char *arg, const char *tensor) {
  static_cast<OpSpec *>(s)->AddArgumentInput(arg, tensor);
}

// ---------------------------------------------------------------------------------------- Pipeline
API void *daliamdPipelineCreate(int batch_size, int num_threads, int device_id, int64_t seed, int prefetch_queue_depth,
                                int exec_async) {
  void *h = nullptr;
  Guard([&] {
    PipelineParams p;
    p.batch_size = batch_size; p.num_threads = num_threads; p.device_id = device_id; p.seed = seed;
    p.prefetch_queue_depth = prefetch_queue_depth; p.exec_async = exec_async != 0;
    auto *ph = new PipelineHandle;
    ph->pipe = std::make_unique<Pipeline>(p);
    h = ph;
  });
  return h;
}
// before Build(): Pipeline(set_affinity=True) (reference pipeline.py:164)
API int daliamdPipelineSetAffinity(void *h, int on) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->SetAffinity(on != 0); });
}
// per-operator device time (timing events around the launches of every mixed / gpu operator); enable before Build()
API int daliamdPipelineEnableOperatorTiming(void *h, int on) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->EnableOperatorTiming(on != 0); });
}
// "name\tms\n" per operator: average device milliseconds over the iterations handed out so far
API int daliamdPipelineOperatorTimes(void *h, char *buf, int len) {
  std::string s;
  Guard([&] {
    for (auto &kv : static_cast<PipelineHandle *>(h)->pipe->OperatorDeviceTimesMs()) s += kv.first + "\t" + std::to_string(kv.second) + "\n";
  });
  return CopyOut(s, buf, len);
}
// "name\tms\n": host milliseconds per iteration of every operator (SetupImpl + RunImpl on its stage thread) since the last
// call, then "<host stage>", "<device stage>", "<slot wait>" and "<iterations>"; the call resets the window
API int daliamdPipelineOperatorHostTimes(void *h, char *buf, int len) {
  static thread_local std::string s;   // (the two-call protocol - size, then contents - must see ONE snapshot)
  if (!buf || len <= 0) {
    s.clear();
    Guard([&] {
      for (auto &kv : static_cast<PipelineHandle *>(h)->pipe->OperatorHostTimesMs()) s += kv.first + "\t" + std::to_string(kv.second) + "\n";
    });
  }
  return CopyOut(s, buf, len);
}
// {streams resident, bytes used, lookups that hit, lookups that missed} of the encoded-stream cache of the device
// (decoders.image(cache_type="encoded")); zeros when there is none
API int daliamdEncodedCacheStats(int device_id, int64_t *out4) {
  return Guard([&] {
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (auto c = StreamCache::Find(device_id)) c->Stats(out4);
  });
}
API void daliamdPipelineDestroy(void *h) { delete static_cast<PipelineHandle *>(h); }
API int64_t daliamdPipelineSeed(void *h) { return static_cast<PipelineHandle *>(h)->pipe->seed(); }

API int daliamdPipelineAddOperator(void *h, void *spec, const char *inst_name) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->AddOperator(*static_cast<OpSpec *>(spec), inst_name); });
}
API int daliamdPipelineBuild(void *h, const char *const *names, const int *gpu, int n) {
  return Guard([&] {
    std::vector<std::pair<std::string, std::string>> outs;
    for (int i = 0; i < n; i++) outs.push_back({names[i], gpu[i] ? "gpu" : "cpu"});
    static_cast<PipelineHandle *>(h)->pipe->Build(outs);
  });
}
API int daliamdPipelineRun(void *h) { return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->Run(); }); }
API int daliamdPipelineOutputs(void *h, int *num_outputs) {
  return Guard([&] {
    auto *ph = static_cast<PipelineHandle *>(h);
    ph->outputs.clear();  // releases the previous iteration
    ph->outputs = ph->pipe->Outputs();
    *num_outputs = (int)ph->outputs.size();
  });
}
API int daliamdPipelineOutputsOnStream(void *h, void *consumer_stream, int *num_outputs) {
  return Guard([&] {
    auto *ph = static_cast<PipelineHandle *>(h);
    ph->outputs.clear();
    ph->outputs = ph->pipe->OutputsOnStream(consumer_stream);
    *num_outputs = (int)ph->outputs.size();
  });
}
API int daliamdPipelineWaitEnqueued(void *h) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->WaitEnqueued(); });
}
API int daliamdPipelineReleaseOnStream(void *h, void *consumer_stream) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->ReleaseOnStream(consumer_stream); });
}
API int daliamdPipelineFlushChecks(void *h) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->FlushChecks(); });
}
// info: [0] device (0 cpu / 1 gpu), [1] dtype, [2] num_samples, [3] dense (1) or row-padded (0)
API int daliamdPipelineOutputInfo(void *h, int idx, int64_t *info, char *layout, int layout_len) {
  return Guard([&] {
    auto &tl = *static_cast<PipelineHandle *>(h)->outputs.at(idx);
    info[0] = tl.device() == StorageDevice::GPU;
    info[1] = tl.type();
    info[2] = tl.num_samples();
    info[3] = tl.is_dense();
    CopyOut(tl.layout(), layout, layout_len);
  });
}
API int daliamdPipelineOutputSample(void *h, int idx, int sample, void **ptr, int64_t *shape, int *ndim, int64_t *row_pitch) {
  return Guard([&] {
    auto &tl = *static_cast<PipelineHandle *>(h)->outputs.at(idx);
    DALI_ENFORCE(sample >= 0 && sample < tl.num_samples(), "Sample index ", sample, " out of range");
    *ptr = tl.raw(sample);
    const TensorShape &s = tl.shape(sample);
    DALI_ENFORCE(s.size() <= 8, "Too many dimensions");
    *ndim = (int)s.size();
    for (size_t d = 0; d < s.size(); d++) shape[d] = s[d];
    *row_pitch = tl.row_pitch(sample);
  });
}
// All samples of one output in one call: ptrs[n], shapes[n][8], ndims[n], row_pitches[n] (n = num_samples of OutputInfo)
API int daliamdPipelineOutputSamples(void *h, int idx, void **ptrs, int64_t *shapes, int *ndims, int64_t *row_pitches) {
  return Guard([&] {
    auto &tl = *static_cast<PipelineHandle *>(h)->outputs.at(idx);
    for (int i = 0; i < tl.num_samples(); i++) {
      const TensorShape &s = tl.shape(i);
      DALI_ENFORCE(s.size() <= 8, "Too many dimensions");
      ptrs[i] = tl.raw(i);
      ndims[i] = (int)s.size();
      for (size_t d = 0; d < s.size(); d++) shapes[(size_t)i * 8 + d] = s[d];
      row_pitches[i] = tl.row_pitch(i);
    }
  });
}
API int daliamdPipelineFeedInput(void *h, const char *op_name, const void *const *data, const int64_t *shapes, int ndim, int n,
                                 int dtype, const char *layout) {
  return Guard([&] {
    std::vector<const void *> d(data, data + n);
    std::vector<TensorShape> s(n);
    for (int i = 0; i < n; i++) s[i].assign(shapes + (size_t)i * ndim, shapes + (size_t)(i + 1) * ndim);
    static_cast<PipelineHandle *>(h)->pipe->FeedInput(op_name, d, s, (DALIDataType)dtype, layout ? layout : "");
  });
}
// meta: epoch_size, epoch_size_padded, number_of_shards, shard_id, pad_last_batch, stick_to_shard
API int daliamdPipelineReaderMeta(void *h, const char *op_name, int64_t *meta) {
  return Guard([&] {
    ReaderMeta m = static_cast<PipelineHandle *>(h)->pipe->GetReaderMeta(op_name);
    meta[0] = m.epoch_size; meta[1] = m.epoch_size_padded; meta[2] = m.number_of_shards; meta[3] = m.shard_id;
    meta[4] = m.pad_last_batch; meta[5] = m.stick_to_shard;
  });
}
API int daliamdPipelineReaderNames(void *h, char *buf, int len) {
  std::string s;
  int rc = Guard([&] { for (auto &n : static_cast<PipelineHandle *>(h)->pipe->ReaderNames()) s += n + "\n"; });
  return rc ? -1 : CopyOut(s, buf, len);
}
API int daliamdPipelineCheckpoint(void *h, char *buf, int len) {
  std::string s;
  int rc = Guard([&] { s = static_cast<PipelineHandle *>(h)->pipe->SaveCheckpoint(); });
  return rc ? -1 : CopyOut(s, buf, len);
}
API int daliamdPipelineRestore(void *h, const char *cpt) {
  return Guard([&] { static_cast<PipelineHandle *>(h)->pipe->RestoreCheckpoint(cpt); });
}
API int daliamdPipelineLastLaunches(void *h, char *buf, int len) {
  std::string s;
  int rc = Guard([&] { for (auto &n : static_cast<PipelineHandle *>(h)->pipe->LastLaunches()) s += n + "\n"; });
  return rc ? -1 : CopyOut(s, buf, len);
}
API void *daliamdPipelineStream(void *h) { return static_cast<PipelineHandle *>(h)->pipe->stream(); }

}  // extern "C"
