// Pipeline + HIP stream/event executor.
// Reference counterparts: Pipeline (dali/pipeline/pipeline.h:87-589, pipeline.cc:566-832) and the
// exec2 executor (dali/pipeline/executor/executor2/exec2.cc:97-145, exec_node_task.cc:329-414).
//
// MI355X design: one pipeline = one GPU.  A worker thread runs iterations ahead of the consumer
// (prefetch_queue_depth): host stages (file read, Huffman on the thread pool, random parameters) of
// iteration i+1 overlap the device stages of iteration i, which are only ENQUEUED on the pipeline's
// HIP stream (pinned H2D staging -> decode kernels -> fused resample/normalise).  Every operator
// output lives in a ring of depth+1 buffers; an iteration-done event guards pinned-buffer reuse and
// is what Outputs() waits on.  No collective, no cross-GPU traffic: multi-GPU = N pipelines with
// disjoint shards.
#ifndef DALI_AMD_HOST_PIPELINE_H_
#define DALI_AMD_HOST_PIPELINE_H_

#include <deque>
#include <random>
#include "framework.h"

namespace daliamd_host {

struct PipelineParams {
  int batch_size = 1;
  int num_threads = 1;
  int device_id = 0;
  int64_t seed = -1;  // < 0: derived from the clock
  int prefetch_queue_depth = 2;
  bool exec_async = true;
  bool set_affinity = false;  // bind worker threads to the CPUs local to the device (pipeline.py:164)
};

class Pipeline {
 public:
  explicit Pipeline(const PipelineParams &p);
  ~Pipeline();

  // adds an operator instance; `inst_name` must be unique (the Python layer generates one when the
  // user gives no `name=`).  Returns the operator index.
  int AddOperator(OpSpec spec, const std::string &inst_name);
  void Build(const std::vector<std::pair<std::string, std::string>> &outputs);  // (tensor name, device)
  bool IsBuilt() const { return built_; }

  // schedules one iteration
  void Run();
  // waits for the oldest scheduled iteration and returns its outputs (released at the next call)
  std::vector<std::shared_ptr<TensorList>> Outputs();
  // The same for a consumer that works on a HIP stream of its own (a framework iterator): the host does not wait for
  // the iteration's device work - `consumer_stream` does (stream-ordered hand-over), so the consumer's copy queues
  // behind the batch and the call returns as soon as the iteration has been enqueued.  The operators' completion
  // checks (decoder status words) run here when the work happens to be complete and otherwise at the next Outputs*()
  // call: an error of iteration i is raised by the call for i or for i + 1.
  std::vector<std::shared_ptr<TensorList>> OutputsOnStream(daliamdStream_t consumer_stream);
  // The consumer's reads of the outputs handed out last are enqueued on `consumer_stream`: their ring slot is not
  // written again before everything in that stream up to this call has completed.  (Without this call a consumer
  // that took the outputs with OutputsOnStream must have finished reading on its own.)
  void ReleaseOnStream(daliamdStream_t consumer_stream);
  // Completion checks of a stream-ordered hand-over that were deferred (the iteration was still running): waits for that
  // iteration and raises its error now - the iterator calls this behind the last batch of an epoch, where no later
  // Outputs*() call would.
  void FlushChecks() { RunPendingChecks(); }
  // blocks until the device work of every scheduled iteration has been ENQUEUED (its host and device stages have run);
  // a device synchronisation behind this call then covers everything Run() has asked for (benchmarks: the end of a
  // timed region)
  void WaitEnqueued();
  // feed one batch to an ExternalSource operator instance
  void FeedInput(const std::string &op_name, const std::vector<const void *> &data,
                 const std::vector<TensorShape> &shapes, DALIDataType type, const std::string &layout);
  ReaderMeta GetReaderMeta(const std::string &op_name) const;
  std::vector<std::string> ReaderNames() const;
  // textual checkpoint: one line per stateful operator ("<name>=<state>")
  std::string SaveCheckpoint() const;
  void RestoreCheckpoint(const std::string &cpt);
  // which device kernels the iteration last returned by Outputs() launched ("fused_resample_cmn", "jpeg_idct", ...)
  std::vector<std::string> LastLaunches() const;

  const PipelineParams &params() const { return params_; }
  int64_t seed() const { return original_seed_; }
  void SetAffinity(bool on) { params_.set_affinity = on; }  // before Build()
  // Brackets every mixed / gpu operator's launches with timing events on its stream (before Build()); the elapsed
  // times are collected when an iteration's outputs are handed out.  For benchmarks: per-operator device time
  // measured live, on the stream the kernels run on.
  void EnableOperatorTiming(bool on) { op_timing_ = on; }
  std::vector<std::pair<std::string, double>> OperatorDeviceTimesMs() const;
  // Host time of the stage threads since the last call (which resets the counters): per operator the average
  // milliseconds per iteration its SetupImpl + RunImpl took (enqueueing included), "<host stage>" / "<device stage>" =
  // the sums per stage, "<slot wait>" = the host stage blocked on the ring slot's previous user.  Always collected (two
  // clock reads per operator and iteration).
  std::vector<std::pair<std::string, double>> OperatorHostTimesMs();
  daliamdStream_t stream() const { return streams_.empty() ? nullptr : streams_[0]; }
  int ring() const { return ring_; }

 private:
  struct Node {
    std::string name;
    OpSpec spec;
    OpType type;
    std::unique_ptr<OperatorBase> op;
    std::vector<int> in_node, in_idx;                 // producer of each regular input
    std::vector<std::pair<std::string, std::pair<int, int>>> arg_in;  // arg name -> producer
    std::vector<std::vector<std::shared_ptr<TensorList>>> out_ring;   // [output][slot]
    double host_seconds = 0;  // time the worker spent in SetupImpl + RunImpl (enqueueing included)
    double host_seconds_window = 0;   // ... since the last OperatorHostTimesMs()
    // device time of the operator (EnableOperatorTiming): events around its launches, one pair per ring slot
    std::vector<daliamdEvent_t> ev_begin, ev_end;
    double device_ms = 0;
    int64_t device_ms_count = 0;
  };
  struct Iteration {
    int slot;
    daliamdEvent_t done = nullptr;
    std::string error;
    bool failed = false;
    std::vector<std::function<void()>> checks;  // run by Outputs() once the device work is complete
    std::vector<std::string> launches;          // kernels (device, or host_* for the CPU backend) of this iteration
  };

  // One iteration = host stage (CPU operators: readers, random numbers, external sources) followed by the device
  // stage (mixed + gpu operators: their host-side preparation and the enqueueing of device work).  With
  // exec_async the two stages run on two threads, so the file reads of iteration i+1 overlap the header parsing /
  // descriptor building / launches of iteration i (the reference's separate CPU and mixed/GPU stage threads).
  void RunStage(bool device_stage, int64_t it, int slot, Iteration &res);
  void CpuWorkerLoop();
  void DeviceWorkerLoop();

  PipelineParams params_;
  int64_t original_seed_;
  std::vector<int64_t> seeds_;
  size_t current_seed_ = 0;
  std::vector<Node> nodes_;
  std::map<std::string, std::pair<int, int>> tensor_producer_;  // "name_device" -> (node, out idx)
  std::vector<std::pair<int, int>> outputs_;
  bool built_ = false;
  bool op_timing_ = false;
  bool roi_decode_fused_ = false;   // a decoders.image -> random_resized_crop pair decodes windows only (Build)
  bool trace_ = false;  // DALI_AMD_TRACE=1: per-operator host time summary on stderr when the pipeline is destroyed
  int64_t traced_iterations_ = 0;
  double slot_wait_seconds_ = 0;  // host stage blocked on the ring slot's previous user
  std::mutex host_times_m_;       // the *_window counters: written by the stage threads, read + reset by the consumer
  double slot_wait_window_ = 0;
  int64_t window_iterations_ = 0;
  bool have_gpu_ = false;
  std::unique_ptr<ThreadPool> thread_pool_;      // device-stage operators (e.g. the decoder's header parsing)
  std::unique_ptr<ThreadPool> cpu_thread_pool_;  // host-stage operators (e.g. the reader's file reads)
  // the compute streams (kComputeStreams distinct ones, see Build()): iteration i runs on stream i mod their count, so
  // consecutive iterations always use different streams and the latency-bound tail of one batch's kernels (the entropy
  // decoder's relaxation rounds) overlaps the start of the next batch's
  static constexpr int kComputeStreams = 3;
  std::vector<daliamdStream_t> streams_;
  daliamdStream_t aux_stream_ = nullptr;   // small set-up launches that depend on host data only (Workspace::aux_stream)
  daliamdStream_t copy_stream_ = nullptr;  // bulk H2D staging (highest stream priority), overlaps the compute streams
  int ring_ = 3;

  // scheduling
  std::thread worker_, cpu_worker_;
  std::vector<int> local_cpus_;  // set_affinity: CPUs of the device's NUMA node (empty: unbound)
  std::mutex m_;
  std::condition_variable cv_req_, cv_res_, cv_mid_, cv_dev_done_;
  std::deque<int64_t> requests_;
  std::deque<std::pair<int64_t, Iteration>> mid_;  // host stage done, device stage pending
  std::deque<Iteration> results_;
  int64_t device_stages_done_ = 0;  // iterations whose device stage has been enqueued (slot event recorded)
  std::vector<daliamdEvent_t> slot_events_;
  // OutputsOnStream / ReleaseOnStream: per ring slot, the event the consumer's stream recorded behind its reads
  // (written by the consumer thread before it schedules the slot's next user: Run() publishes it under m_)
  std::vector<daliamdEvent_t> release_events_;
  std::vector<char> release_pending_;
  int held_slot_ = -1;
  struct PendingChecks {
    int slot = -1;
    std::vector<std::function<void()>> checks;
  } pending_checks_;
  void RunPendingChecks();
  void WaitForSlot(int slot);
  std::vector<std::shared_ptr<TensorList>> TakeOutputs(daliamdStream_t consumer_stream, bool on_stream);
  int64_t scheduled_ = 0, consumed_ = 0;
  bool stop_ = false;
  bool holding_ = false;  // the consumer holds the outputs of iteration consumed_-1
  std::vector<std::string> last_launches_;  // of the iteration handed out last
  mutable std::mutex launches_m_;

 public:
  void NoteLaunch(const std::string &what);
};

// Operators report the kernels they enqueue so tests can assert that the device path really ran.
void NoteLaunch(const Workspace &ws, const std::string &what);

}  // namespace daliamd_host
#endif  // DALI_AMD_HOST_PIPELINE_H_
