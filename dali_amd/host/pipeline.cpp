#include "pipeline.h"
#include <time.h>

#include <algorithm>
#include <chrono>

#include "ops.h"

namespace daliamd_host {

static constexpr int kMaxSeeds = 1024;  // pipeline.h:714

Pipeline::Pipeline(const PipelineParams &p) : params_(p) {
  DALI_ENFORCE(p.batch_size > 0, "Batch size must be greater than 0, got ", p.batch_size);
  DALI_ENFORCE(p.num_threads > 0, "num_threads must be greater than 0, got ", p.num_threads);
  DALI_ENFORCE(p.prefetch_queue_depth > 0, "prefetch_queue_depth must be greater than 0");
  using Clock = std::chrono::high_resolution_clock;
  original_seed_ = p.seed >= 0 ? p.seed : (int64_t)Clock::now().time_since_epoch().count();
  // per-operator seeds: std::seed_seq{seed}.generate(...) like pipeline.cc:303-308
  seeds_.resize(kMaxSeeds);
  std::seed_seq ss{original_seed_};
  ss.generate(seeds_.begin(), seeds_.end());
  ring_ = p.prefetch_queue_depth + 1;
  if (const char *t = getenv("DALI_AMD_TRACE")) trace_ = atoi(t) != 0;
  int ndev = 0;
  daliamdDeviceCount(&ndev);
  have_gpu_ = ndev > 0;
}

Pipeline::~Pipeline() {
  {
    std::lock_guard<std::mutex> g(m_);
    stop_ = true;
  }
  cv_req_.notify_all();
  cv_mid_.notify_all();
  cv_dev_done_.notify_all();
  if (cpu_worker_.joinable()) cpu_worker_.join();
  if (worker_.joinable()) worker_.join();
  for (auto st : streams_) daliamdStreamSynchronize(st);
  // a deferred completion check that nobody asked for any more (the last stream-ordered hand-over before the pipeline
  // went away): a destructor cannot throw - say what it found
  try {
    RunPendingChecks();
  } catch (const std::exception &e) {
    fprintf(stderr, "[dali_amd] error of the last iteration handed out, found when the pipeline was destroyed: %s\n", e.what());
  }
  if (trace_ && traced_iterations_ > 0) {
    fprintf(stderr, "[dali_amd trace] host time per iteration over %lld iterations (worker thread):\n",
            (long long)traced_iterations_);
    for (auto &n : nodes_)
      fprintf(stderr, "[dali_amd trace]   %-40s %8.3f ms\n", n.name.c_str(), 1e3 * n.host_seconds / traced_iterations_);
    fprintf(stderr, "[dali_amd trace]   %-40s %8.3f ms\n", "(host stage waiting for its ring slot)",
            1e3 * slot_wait_seconds_ / traced_iterations_);
  }
  nodes_.clear();
  for (auto e : slot_events_) if (e) daliamdEventDestroy(e);
  for (auto e : release_events_) if (e) daliamdEventDestroy(e);
  if (copy_stream_) daliamdStreamDestroy(copy_stream_);
  if (aux_stream_) daliamdStreamDestroy(aux_stream_);
  for (auto st : streams_) daliamdStreamDestroy(st);
}

static std::string TensorKey(const std::string &name, StorageDevice d) {
  return name + (d == StorageDevice::GPU ? "_gpu" : "_cpu");
}

int Pipeline::AddOperator(OpSpec spec, const std::string &inst_name) {
  DALI_ENFORCE(!built_, "Alterations to the pipeline after \"Build()\" has been called are not allowed");
  for (auto &n : nodes_) DALI_ENFORCE(n.name != inst_name, "Operator instance name \"", inst_name, "\" is not unique");
  const OpSchema &schema = spec.GetSchema();  // throws for unknown operators
  std::string dev = spec.TryArg("device") ? spec.GetString("device") : "cpu";
  OpType type = ParseOpType(dev);
  // inject what the executor owns (pipeline.cc:812-832)
  spec.AddArg("max_batch_size", ArgValue::Int(params_.batch_size));
  spec.AddArg("num_threads", ArgValue::Int(params_.num_threads));
  spec.AddArg("device_id", ArgValue::Int(params_.device_id));
  if (type != OpType::CPU) spec.AddArg("gpu_prefetch_queue_depth", ArgValue::Int(params_.prefetch_queue_depth));
  if (schema.HasRandomSeedArg() && !spec.ArgumentDefined("seed")) {
    spec.AddArg("seed", ArgValue::Int(seeds_[current_seed_]));
    current_seed_ = (current_seed_ + 1) % kMaxSeeds;
  }
  spec.Validate();
  Node n;
  n.name = inst_name;
  n.type = type;
  // inputs must already exist with the requested storage device
  for (auto &in : spec.Inputs()) {
    auto it = tensor_producer_.find(TensorKey(in.name, in.dev));
    if (it == tensor_producer_.end()) {
      bool other = tensor_producer_.count(TensorKey(in.name, in.dev == StorageDevice::GPU ? StorageDevice::CPU
                                                                                           : StorageDevice::GPU));
      DALI_ENFORCE(!other, "Operator \"", inst_name, "\" (", spec.SchemaName(), ") requests input \"", in.name,
                   "\" on the ", in.dev == StorageDevice::GPU ? "GPU" : "CPU", " but it is produced on the other "
                   "device. Use `.gpu()` to move the data to the device; device-to-host transfers inside the graph "
                   "are not supported.");
      DALI_FAIL("Input \"", in.name, "\" of operator \"", inst_name, "\" is not produced by any operator");
    }
    if (type == OpType::CPU || type == OpType::MIXED)
      DALI_ENFORCE(in.dev == StorageDevice::CPU, (type == OpType::CPU ? "CPU" : "Mixed"), " operator \"", inst_name,
                   "\" cannot take a GPU input");
    n.in_node.push_back(it->second.first);
    n.in_idx.push_back(it->second.second);
  }
  for (auto &kv : spec.ArgumentInputs()) {
    auto it = tensor_producer_.find(TensorKey(kv.second, StorageDevice::CPU));
    DALI_ENFORCE(it != tensor_producer_.end(), "Argument input \"", kv.first, "\" of operator \"", inst_name,
                 "\" must be a CPU tensor produced earlier in the graph");
    n.arg_in.push_back({kv.first, it->second});
  }
  int idx = (int)nodes_.size();
  int k = 0;
  for (auto &out : spec.Outputs()) {
    StorageDevice expect = type == OpType::CPU ? StorageDevice::CPU : StorageDevice::GPU;
    DALI_ENFORCE(out.dev == expect, "Operator \"", inst_name, "\" produces ", expect == StorageDevice::GPU ? "GPU" : "CPU",
                 " outputs");
    std::string key = TensorKey(out.name, out.dev);
    DALI_ENFORCE(!tensor_producer_.count(key), "Tensor \"", out.name, "\" is produced twice");
    tensor_producer_[key] = {idx, k++};
  }
  n.spec = std::move(spec);
  nodes_.push_back(std::move(n));
  return idx;
}

void Pipeline::Build(const std::vector<std::pair<std::string, std::string>> &outputs) {
  DALI_ENFORCE(!built_, "\"Build()\" can only be called once");
  DALI_ENFORCE(!outputs.empty(), "There must be at least one output");
  bool needs_gpu = false;
  for (auto &n : nodes_) needs_gpu |= n.type != OpType::CPU;
  if (needs_gpu) {
    DALI_ENFORCE(have_gpu_, "The pipeline contains device (\"gpu\"/\"mixed\") operators but no MI355X/ROCm device "
                 "is available. There is no CPU fallback for device operators.");
    KCHECK(daliamdSetDevice(params_.device_id));
    // Compute streams: iteration i runs on stream i mod kComputeStreams (chosen by ITERATION, not by ring slot: with a
    // ring size that is not a multiple of the stream count the last slot and slot 0 would share a stream and two
    // consecutive iterations would be serialised once per revolution; the host stage waits for the slot's event
    // before it reuses a slot, so a slot may be written from any stream).  More batches than that in
    // flight do not help - two position passes of the entropy decoder side by side each take twice as long - but
    // which streams end up SHARING an in-order hardware queue does: the runtime spreads the streams of one priority
    // over four hardware queues in creation order, so with one stream per ring slot the throughput depended on the
    // ring size and on every other stream of the process (prefetch_queue_depth 4 / 5 / 6: 370 / 450 / 413 k images/s
    // on the resident hot path; one unrelated stream created first: 386 / 390 / 400).  Three streams - the batch
    // whose positions are being decoded, the one in its throughput-bound kernels, the one finishing - map to three
    // hardware queues whatever the depth: 431 / 437 k at depth 4 / 5 (gpurun_out/r03a[d-g]_env, HISTORY.md section 6).
    // DALI_AMD_PIPELINE_STREAMS overrides the count (0 = one per ring slot).
    int distinct = std::min(ring_, kComputeStreams);
    if (const char *e = getenv("DALI_AMD_PIPELINE_STREAMS")) {
      const int v = atoi(e);
      distinct = v <= 0 ? ring_ : std::min(ring_, v);
    }
    streams_.assign(distinct, nullptr);
    for (int i = 0; i < distinct; i++) KCHECK(daliamdStreamCreate(&streams_[i], 1));
    // The copy stream carries the descriptor tables (and the JPEG bytes) of the NEXT iteration: highest priority = a
    // hardware queue it does not share with any compute stream, or the upload waits behind a 0.3 ms kernel
    // (DALI_AMD_COPY_STREAM_PRIORITY: -1 / 0 / 1, for measurements)
    KCHECK(daliamdStreamCreateWithPriority(&copy_stream_, 1, getenv("DALI_AMD_COPY_STREAM_PRIORITY") ? atoi(getenv("DALI_AMD_COPY_STREAM_PRIORITY")) : -1));
    // ... and one for the small set-up launches of an iteration that need host data only (DALI_AMD_AUX_STREAM=0: none,
    // they stay on the compute stream)
    if (!(getenv("DALI_AMD_AUX_STREAM") && atoi(getenv("DALI_AMD_AUX_STREAM")) == 0))
      KCHECK(daliamdStreamCreateWithPriority(&aux_stream_, 1, -1));
  }
  for (auto &o : outputs) {
    StorageDevice d = o.second == "gpu" ? StorageDevice::GPU : StorageDevice::CPU;
    auto it = tensor_producer_.find(TensorKey(o.first, d));
    DALI_ENFORCE(it != tensor_producer_.end(), "Requested output \"", o.first, "\" on device \"", o.second,
                 "\" is not produced by any operator");
    outputs_.push_back(it->second);
  }
  // set_affinity: worker threads (and with them the first touch of their pinned staging buffers) stay on the
  // GPU's NUMA node - with 8 GPUs behind two sockets the H2D copies otherwise cross the socket link
  if (params_.set_affinity && needs_gpu) local_cpus_ = DeviceLocalCpus(params_.device_id);
  thread_pool_ = std::make_unique<ThreadPool>(params_.num_threads, local_cpus_, "dali-devpool");
  cpu_thread_pool_ = std::make_unique<ThreadPool>(params_.num_threads, local_cpus_, "dali-cpupool");
  // instantiate operators (InstantiateOperator, operator.cc:157-169) and their output rings
  for (auto &n : nodes_) {
    try {
      n.op = OperatorRegistry::Create(n.spec.SchemaName(), n.type, n.spec);
    } catch (const std::exception &e) {
      DALI_FAIL("Error when constructing operator \"", n.name, "\" (", n.spec.SchemaName(), "): ", e.what());
    }
    n.out_ring.resize(n.spec.Outputs().size());
    for (size_t k = 0; k < n.out_ring.size(); k++)
      for (int s = 0; s < ring_; s++) n.out_ring[k].push_back(std::make_shared<TensorList>(n.spec.Outputs()[k].dev));
  }
  // graph-level fusion: RandomResizedCrop / Resize feeding ONLY a CropMirrorNormalize is deferred into it
  std::vector<int> consumers(nodes_.size(), 0);
  for (auto &n : nodes_) for (int p : n.in_node) consumers[p]++;
  for (auto &o : outputs_) consumers[o.first] += 2;  // pipeline outputs are never deferred
  for (auto &n : nodes_) {
    if (n.spec.SchemaName() != "CropMirrorNormalize" || n.in_node.empty()) continue;
    Node &p = nodes_[n.in_node[0]];
    if (consumers[n.in_node[0]] == 1 && p.type == OpType::GPU) TryEnableFusion(p.op.get(), n.op.get());
  }
  // ... and a decoders.image feeding ONLY a RandomResizedCrop decodes just the windows that operator draws
  {
    const char *env = getenv("DALI_AMD_ROI_FUSION");
    if (!env || atoi(env) != 0)
      for (auto &n : nodes_) {
        if (n.spec.SchemaName() != "RandomResizedCrop" || n.in_node.empty() || n.type != OpType::GPU) continue;
        Node &p = nodes_[n.in_node[0]];
        const std::string &pn = p.spec.SchemaName();
        if (consumers[n.in_node[0]] == 1 && p.type == OpType::MIXED &&
            (pn == "decoders__Image" || pn == "ImageDecoder" || pn == "experimental__decoders__Image"))
          roi_decode_fused_ = TryEnableRoiDecodeFusion(p.op.get(), n.op.get()) || roi_decode_fused_;
      }
  }
  // ... and a ColorTwist feeding ONLY an Erase becomes part of its launch
  for (auto &n : nodes_) {
    if (n.spec.SchemaName() != "Erase" || n.in_node.empty() || n.type != OpType::GPU) continue;
    Node &p = nodes_[n.in_node[0]];
    if (consumers[n.in_node[0]] == 1 && p.type == OpType::GPU) TryEnablePointwiseFusion(p.op.get(), n.op.get());
  }
  // ... and a GaussianBlur feeding ONLY a pointwise operator applies it in its own write-out
  for (auto &n : nodes_) {
    const std::string &sn = n.spec.SchemaName();
    if ((sn != "ColorTwist" && sn != "Hsv" && sn != "Hue" && sn != "Saturation" && sn != "Erase") || n.in_node.empty() ||
        n.type != OpType::GPU)
      continue;
    Node &p = nodes_[n.in_node[0]];
    if (consumers[n.in_node[0]] == 1 && p.type == OpType::GPU && p.spec.SchemaName() == "GaussianBlur")
      TryEnableBlurFusion(p.op.get(), n.op.get());
  }
  // ... and Spectrogram -> MelFilterBank (-> ToDecibels) chains with single consumers are one launch (two when the decibel
  // reference is the sample's maximum); in graph order, so that the mel operator knows about its input before the
  // decibel operator asks it
  for (auto &n : nodes_) {
    const std::string &sn = n.spec.SchemaName();
    if ((sn != "MelFilterBank" && sn != "ToDecibels") || n.in_node.empty() || n.type != OpType::GPU) continue;
    Node &p = nodes_[n.in_node[0]];
    if (consumers[n.in_node[0]] == 1 && p.type == OpType::GPU) TryEnableAudioFusion(p.op.get(), n.op.get());
  }
  // ... and the audio samples of a decoders.audio whose ONLY consumer is the copy to the device in front of a gpu
  // Spectrogram (itself that copy's only consumer) cross the bus as the 16-bit PCM the files hold
  for (auto &n : nodes_) {
    if (n.spec.SchemaName() != "Spectrogram" || n.type != OpType::GPU || n.in_node.empty()) continue;
    const int c = n.in_node[0];
    Node &copy = nodes_[c];
    if (copy.spec.SchemaName() != "_CopyToGpu" || consumers[c] != 1 || copy.in_node.empty() || copy.in_idx[0] != 0) continue;
    const int q = copy.in_node[0];
    const std::string &dn = nodes_[q].spec.SchemaName();
    if (dn != "decoders__Audio" && dn != "AudioDecoder") continue;
    int uses = 0;   // of the decoder's first output (the second one is the sampling rate)
    for (auto &m : nodes_) {
      for (size_t k = 0; k < m.in_node.size(); k++) uses += m.in_node[k] == q && m.in_idx[k] == 0;
      for (auto &a : m.arg_in) uses += a.second.first == q && a.second.second == 0;
    }
    for (auto &o : outputs_) uses += 2 * (o.first == q && o.second == 0);
    if (uses == 1) TryEnablePcm16Fusion(nodes_[q].op.get(), n.op.get());
  }
  slot_events_.assign(ring_, nullptr);
  if (!streams_.empty())
    for (auto &e : slot_events_) KCHECK(daliamdEventCreate(&e, 2));  // the consumer sleeps in Outputs(), it does not poll
  release_events_.assign(ring_, nullptr);
  release_pending_.assign(ring_, 0);
  if (!streams_.empty())
    for (auto &e : release_events_) KCHECK(daliamdEventCreate(&e, 2));
  if (op_timing_ && !streams_.empty())
    for (auto &n : nodes_) {
      if (n.type == OpType::CPU) continue;
      n.ev_begin.assign(ring_, nullptr);
      n.ev_end.assign(ring_, nullptr);
      for (int s2 = 0; s2 < ring_; s2++) {
        KCHECK(daliamdEventCreate(&n.ev_begin[s2], 1));
        KCHECK(daliamdEventCreate(&n.ev_end[s2], 1));
      }
    }
  built_ = true;
  if (params_.exec_async) {
    cpu_worker_ = std::thread([this] { NameThisThread("dali-cpustage"); BindThisThread(local_cpus_); CpuWorkerLoop(); });
    worker_ = std::thread([this] { NameThisThread("dali-devstage"); BindThisThread(local_cpus_); DeviceWorkerLoop(); });
  }
}

// launches are noted by the stage thread that runs the operator: the list of the iteration it is working on
static thread_local std::vector<std::string> *tl_launches = nullptr;
void Pipeline::NoteLaunch(const std::string &what) {
  if (tl_launches) tl_launches->push_back(what);
}
std::vector<std::string> Pipeline::LastLaunches() const {
  std::lock_guard<std::mutex> g(launches_m_);
  return last_launches_;
}
void NoteLaunch(const Workspace &ws, const std::string &what) {
  if (ws.pipeline) ws.pipeline->NoteLaunch(what);
}

void Pipeline::RunStage(bool device_stage, int64_t it, int slot, Iteration &res) {
  res.slot = slot;
  struct LaunchScope {  // host stage and device stage of different iterations run concurrently: per-iteration lists
    explicit LaunchScope(std::vector<std::string> *l) { tl_launches = l; }
    ~LaunchScope() { tl_launches = nullptr; }
  } launch_scope(&res.launches);
  struct Range {  // profiler range (roctx), closed on every exit path
    explicit Range(const std::string &n) { daliamdRangePush(n.c_str()); }
    ~Range() { daliamdRangePop(); }
  };
  Range stage_range(device_stage ? "dali_amd device stage" : "dali_amd host stage");
  try {
    if (res.failed) throw std::runtime_error(res.error);  // the host stage already failed: nothing to enqueue
    if (!streams_.empty() && device_stage) KCHECK(daliamdSetDevice(params_.device_id));
    if (!streams_.empty() && !device_stage) {
      KCHECK(daliamdSetDevice(params_.device_id));
      // the buffers of this ring slot (pinned staging included) were last used `ring_` iterations ago: wait until
      // that iteration's device work has been enqueued (its event recorded) and has completed
      if (it >= ring_) {
        auto t_wait = std::chrono::steady_clock::now();
        {
          std::unique_lock<std::mutex> lk(m_);
          cv_dev_done_.wait(lk, [&] { return stop_ || device_stages_done_ > it - ring_; });
          if (stop_) return;
        }
        SleepWaitEvent(slot_events_[slot]);
        if (release_pending_[slot]) {  // a stream-ordered consumer may still be reading the slot's outputs
          SleepWaitEvent(release_events_[slot]);
          release_pending_[slot] = 0;
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
        slot_wait_seconds_ += waited;
        std::lock_guard<std::mutex> g(host_times_m_);
        slot_wait_window_ += waited;
      }
    }
    for (auto &n : nodes_) {
      if ((n.type != OpType::CPU) != device_stage) continue;
      Workspace ws;
      ws.pipeline = this;
      ws.backend = n.type;
      ws.thread_pool = device_stage ? thread_pool_.get() : cpu_thread_pool_.get();
      ws.stream = streams_.empty() ? nullptr : streams_[it % (int64_t)streams_.size()];
      ws.copy_stream = copy_stream_;
      ws.aux_stream = aux_stream_;
      ws.ring = ring_;
      ws.batch_size = params_.batch_size;
      ws.iteration = it;
      std::vector<std::function<void()>> node_checks;
      ws.completion_checks = &node_checks;
      for (size_t i = 0; i < n.in_node.size(); i++) ws.inputs.push_back(nodes_[n.in_node[i]].out_ring[n.in_idx[i]][slot]);
      for (auto &a : n.arg_in) ws.argument_inputs[a.first] = nodes_[a.second.first].out_ring[a.second.second][slot];
      for (auto &r : n.out_ring) ws.outputs.push_back(r[slot]);
      auto t_node = std::chrono::steady_clock::now();
      Range op_range(n.name);
      try {
        std::vector<OutputDesc> desc(ws.outputs.size());
        if (n.op->SetupImpl(desc, ws)) {
          for (size_t k = 0; k < desc.size(); k++) ws.outputs[k]->Resize(desc[k].shape, desc[k].type, n.op->OutputPitchAlign((int)k));
        }
        if (!n.ev_begin.empty()) daliamdEventRecord(n.ev_begin[slot], ws.stream);
        n.op->RunImpl(ws);
        if (!n.ev_end.empty()) daliamdEventRecord(n.ev_end[slot], ws.stream);
        for (auto &chk : node_checks) {
          // same decoration as synchronous errors
          std::string where = make_string("Error in ", OpTypeName(n.type), " operator `", n.spec.SchemaName(),
                                          "` (instance \"", n.name, "\"): ");
          res.checks.push_back([chk, where] {
            try {
              chk();
            } catch (const std::exception &e) {
              throw std::runtime_error(where + e.what());
            }
          });
        }
      } catch (const std::exception &e) {
        ws.thread_pool->Discard();
        // error_reporting.h: decorate with the operator's origin
        DALI_FAIL("Error in ", OpTypeName(n.type), " operator `", n.spec.SchemaName(), "` (instance \"", n.name, "\"): ",
                  e.what());
      }
      {
        const double spent = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_node).count();
        std::lock_guard<std::mutex> g(host_times_m_);
        n.host_seconds += spent;
        n.host_seconds_window += spent;
      }
    }
    if (device_stage) {
      traced_iterations_++;
      std::lock_guard<std::mutex> g(host_times_m_);
      window_iterations_++;
    }
  } catch (const std::exception &e) {
    res.failed = true;
    res.error = e.what();
  }
  if (device_stage) {
    // recorded even after a failure: the slot's next user waits for this event
    if (!streams_.empty()) daliamdEventRecord(slot_events_[slot], streams_[it % (int64_t)streams_.size()]);
    {
      std::lock_guard<std::mutex> g(m_);
      device_stages_done_ = it + 1;
    }
    cv_dev_done_.notify_all();
  }
}

void Pipeline::CpuWorkerLoop() {
  for (;;) {
    int64_t it;
    {
      std::unique_lock<std::mutex> lk(m_);
      cv_req_.wait(lk, [this] { return stop_ || !requests_.empty(); });
      if (stop_) return;
      it = requests_.front();
      requests_.pop_front();
    }
    Iteration res;
    RunStage(false, it, (int)(it % ring_), res);
    {
      std::lock_guard<std::mutex> g(m_);
      if (stop_) return;
      mid_.emplace_back(it, std::move(res));
    }
    cv_mid_.notify_all();
  }
}

void Pipeline::DeviceWorkerLoop() {
  for (;;) {
    std::pair<int64_t, Iteration> job;
    {
      std::unique_lock<std::mutex> lk(m_);
      cv_mid_.wait(lk, [this] { return stop_ || !mid_.empty(); });
      if (stop_) return;
      job = std::move(mid_.front());
      mid_.pop_front();
    }
    RunStage(true, job.first, (int)(job.first % ring_), job.second);
    {
      std::lock_guard<std::mutex> g(m_);
      results_.push_back(std::move(job.second));
    }
    cv_res_.notify_all();
  }
}

void Pipeline::Run() {
  DALI_ENFORCE(built_, "\"Build()\" must be called before \"Run()\"");
  // never run further ahead than the ring allows while the consumer still holds one iteration
  if (scheduled_ - consumed_ >= params_.prefetch_queue_depth + (holding_ ? 0 : 1))
    DALI_FAIL("Trying to schedule more iterations than the prefetch queue depth (", params_.prefetch_queue_depth,
              ") allows; call Outputs() first");
  int64_t it = scheduled_++;
  if (params_.exec_async) {
    {
      std::lock_guard<std::mutex> g(m_);
      requests_.push_back(it);
    }
    cv_req_.notify_all();
  } else {
    Iteration res;
    RunStage(false, it, (int)(it % ring_), res);
    RunStage(true, it, (int)(it % ring_), res);
    std::lock_guard<std::mutex> g(m_);
    results_.push_back(std::move(res));
  }
}

// The consumer's wait for an iteration's device work.  hipEventSynchronize - on an event made with hipEventBlockingSync as
// well - kept the calling thread on a CPU for the whole wait on the bench box: 1.5 ms of the main thread's CPU per 2.9 ms
// batch for one rank's share of an 8-GPU node (40 % of the pipeline's host cost), 0.65 ms per 0.62 ms batch with 16 CPUs
// (gpurun_out/r05_w).  So the consumer asks whether the event is done and sleeps 50 us in between: 0.03-0.05 ms of CPU per
// batch, the same rate (the consumer is `prefetch_queue_depth` iterations behind the producer - the wait usually finds
// the work done).  DALI_AMD_OUTPUT_WAIT=block: hipEventSynchronize as before.
void Pipeline::WaitForSlot(int slot) { SleepWaitEvent(slot_events_[slot]); }

void Pipeline::RunPendingChecks() {
  if (pending_checks_.slot < 0) return;
  PendingChecks p = std::move(pending_checks_);
  pending_checks_ = PendingChecks{};
  SleepWaitEvent(slot_events_[p.slot]);
  for (auto &chk : p.checks) chk();
}

std::vector<std::shared_ptr<TensorList>> Pipeline::Outputs() { return TakeOutputs(nullptr, false); }
std::vector<std::shared_ptr<TensorList>> Pipeline::OutputsOnStream(daliamdStream_t consumer_stream) {
  return TakeOutputs(consumer_stream, true);
}

std::vector<std::shared_ptr<TensorList>> Pipeline::TakeOutputs(daliamdStream_t consumer_stream, bool on_stream) {
  DALI_ENFORCE(built_, "\"Build()\" must be called before \"Outputs()\"");
  DALI_ENFORCE(scheduled_ > consumed_, "There are no iterations scheduled; call Run() before Outputs()");
  RunPendingChecks();  // of the previous stream-ordered hand-over (throws that iteration's error)
  Iteration res;
  {
    std::unique_lock<std::mutex> lk(m_);
    cv_res_.wait(lk, [this] { return !results_.empty(); });
    res = std::move(results_.front());
    results_.pop_front();
  }
  consumed_++;
  holding_ = true;
  held_slot_ = res.slot;
  {
    std::lock_guard<std::mutex> g(launches_m_);
    last_launches_ = res.launches;
  }
  if (res.failed) throw std::runtime_error(res.error);
  bool complete = true;
  if (!streams_.empty()) {
    if (on_stream) {
      KCHECK(daliamdStreamWaitEvent(consumer_stream, slot_events_[res.slot]));
      int done = 0;
      KCHECK(daliamdEventQuery(slot_events_[res.slot], &done));
      complete = done != 0;
    } else {
      WaitForSlot(res.slot);
    }
  }
  if (op_timing_ && complete)
    for (auto &n : nodes_) {
      if (n.ev_begin.empty()) continue;
      float ms = 0;
      if (daliamdEventElapsedMs(n.ev_begin[res.slot], n.ev_end[res.slot], &ms) == DALIAMD_SUCCESS) {
        n.device_ms += ms;
        n.device_ms_count++;
      }
    }
  if (complete) {
    for (auto &chk : res.checks) chk();
  } else if (!res.checks.empty()) {
    pending_checks_.slot = res.slot;
    pending_checks_.checks = std::move(res.checks);
  }
  std::vector<std::shared_ptr<TensorList>> out;
  for (auto &o : outputs_) out.push_back(nodes_[o.first].out_ring[o.second][res.slot]);
  return out;
}

void Pipeline::WaitEnqueued() {
  std::unique_lock<std::mutex> lk(m_);
  cv_res_.wait(lk, [this] { return (int64_t)results_.size() >= scheduled_ - consumed_; });
}

void Pipeline::ReleaseOnStream(daliamdStream_t consumer_stream) {
  if (streams_.empty() || held_slot_ < 0) return;
  KCHECK(daliamdEventRecord(release_events_[held_slot_], consumer_stream));
  release_pending_[held_slot_] = 1;
  held_slot_ = -1;
}

void Pipeline::FeedInput(const std::string &op_name, const std::vector<const void *> &data,
                         const std::vector<TensorShape> &shapes, DALIDataType type, const std::string &layout) {
  for (auto &n : nodes_) {
    if (n.name != op_name) continue;
    DALI_ENFORCE(n.op, "\"Build()\" must be called before feeding inputs");
    FeedExternalSource(n.op.get(), data, shapes, type, layout);
    return;
  }
  DALI_FAIL("Could not find an ExternalSource operator named \"", op_name, "\"");
}

std::vector<std::pair<std::string, double>> Pipeline::OperatorDeviceTimesMs() const {
  std::vector<std::pair<std::string, double>> out;
  for (auto &n : nodes_)
    if (n.device_ms_count > 0) out.push_back({n.name, n.device_ms / (double)n.device_ms_count});
  return out;
}

std::vector<std::pair<std::string, double>> Pipeline::OperatorHostTimesMs() {
  std::vector<std::pair<std::string, double>> out;
  std::lock_guard<std::mutex> g(host_times_m_);
  const double per = window_iterations_ > 0 ? 1e3 / (double)window_iterations_ : 0.0;
  double stage[2] = {0, 0};
  for (auto &n : nodes_) {
    out.push_back({n.name, n.host_seconds_window * per});
    stage[n.type == OpType::CPU ? 0 : 1] += n.host_seconds_window;
    n.host_seconds_window = 0;
  }
  out.push_back({"<host stage>", stage[0] * per});
  out.push_back({"<device stage>", stage[1] * per});
  out.push_back({"<slot wait>", slot_wait_window_ * per});
  out.push_back({"<iterations>", (double)window_iterations_});
  slot_wait_window_ = 0;
  window_iterations_ = 0;
  return out;
}

ReaderMeta Pipeline::GetReaderMeta(const std::string &op_name) const {
  for (auto &n : nodes_) {
    if (n.name != op_name) continue;
    DALI_ENFORCE(n.op, "\"Build()\" must be called first");
    ReaderMeta m = n.op->GetReaderMeta();
    DALI_ENFORCE(m.epoch_size >= 0, "Operator \"", op_name, "\" is not a reader");
    return m;
  }
  DALI_FAIL("Operator \"", op_name, "\" not found in the pipeline");
}

std::vector<std::string> Pipeline::ReaderNames() const {
  std::vector<std::string> out;
  for (auto &n : nodes_)
    if (n.op && n.op->GetReaderMeta().epoch_size >= 0) out.push_back(n.name);
  return out;
}

std::string Pipeline::SaveCheckpoint() const {
  DALI_ENFORCE(scheduled_ == consumed_, "Checkpoints can only be taken when no iterations are in flight");
  std::string out;
  for (auto &n : nodes_) {
    std::string s = n.op ? n.op->SaveState() : "";
    if (!s.empty()) out += n.name + "=" + s + "\n";
  }
  return out;
}

void Pipeline::RestoreCheckpoint(const std::string &cpt) {
  DALI_ENFORCE(built_ && scheduled_ == consumed_, "Checkpoints can only be restored on an idle, built pipeline");
  size_t pos = 0;
  while (pos < cpt.size()) {
    size_t nl = cpt.find('\n', pos);
    if (nl == std::string::npos) nl = cpt.size();
    std::string line = cpt.substr(pos, nl - pos);
    pos = nl + 1;
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string name = line.substr(0, eq), state = line.substr(eq + 1);
    bool found = false;
    for (auto &n : nodes_)
      if (n.name == name) { n.op->RestoreState(state); found = true; }
    DALI_ENFORCE(found, "Checkpoint refers to an unknown operator \"", name, "\"");
  }
}

}  // namespace daliamd_host
