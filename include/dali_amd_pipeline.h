/*
 * dali_amd_pipeline.h -- the flat C API of the pipeline / operator framework in libdali_amd_host.so
 * (dali_amd/host/c_api.cpp): what a language binding builds graphs and runs them with.  dali_amd/_backend.py binds it
 * with ctypes; it plays the role of the reference's C API (include/dali/c_api.h:120-834: daliCreatePipeline, daliRun,
 * daliOutput, daliShapeAtSample, daliGetReaderMetadata, daliGetSerializedCheckpoint ...) and of the pybind layer the
 * reference's Python front end uses (dali/python/backend_impl.cc), with one difference: a graph is not handed over as a
 * serialized protobuf but built call by call from OpSpecs, the way `Pipeline::AddOperator` is used in C++
 * (dali/pipeline/pipeline.h:87-140).
 *
 * Conventions: handles are opaque pointers; functions returning int return 0 on success and non-zero after an error whose
 * text daliamdHostGetLastErrorMessage() (dali_amd_host.h) returns (the operator-decorated messages of the reference's
 * error_reporting.h); functions that return text fill `buf` (`len` bytes) and return 0, or return the size needed when
 * `buf` is NULL or too small (negative on error).  Nothing here throws across the ABI.
 */
#ifndef DALI_AMD_PIPELINE_H_
#define DALI_AMD_PIPELINE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DALIAMD_PIPE_API __attribute__((visibility("default")))

/* ---- operator registry (OpSchema / SchemaRegistry: dali/pipeline/operator/op_schema.h, operator_factory.h) ---- */
DALIAMD_PIPE_API int daliamdSchemaList(char *buf, int len);                       /* registered schema names, one per line */
DALIAMD_PIPE_API int daliamdSchemaInfo(const char *name, char *buf, int len);     /* JSON: doc, inputs / outputs, arguments */
/* dlopen an operator plugin; its DALI_SCHEMA / DALI_REGISTER_OPERATOR statics register on load (plugin_manager.cc:26-41) */
DALIAMD_PIPE_API int daliamdLoadLibrary(const char *path, int global_symbols);

/* ---- OpSpec (dali/pipeline/operator/op_spec.h): one operator instance of the graph ---- */
DALIAMD_PIPE_API void *daliamdOpSpecCreate(const char *schema_name);
DALIAMD_PIPE_API void daliamdOpSpecDestroy(void *spec);
DALIAMD_PIPE_API void daliamdOpSpecAddArgInt(void *spec, const char *name, int64_t value);
DALIAMD_PIPE_API void daliamdOpSpecAddArgBool(void *spec, const char *name, int value);
DALIAMD_PIPE_API void daliamdOpSpecAddArgFloat(void *spec, const char *name, double value);
DALIAMD_PIPE_API void daliamdOpSpecAddArgStr(void *spec, const char *name, const char *value);
DALIAMD_PIPE_API void daliamdOpSpecAddArgIntVec(void *spec, const char *name, const int64_t *values, int count);
DALIAMD_PIPE_API void daliamdOpSpecAddArgFloatVec(void *spec, const char *name, const double *values, int count);
DALIAMD_PIPE_API void daliamdOpSpecAddArgStrVec(void *spec, const char *name, const char *const *values, int count);
DALIAMD_PIPE_API void daliamdOpSpecAddInput(void *spec, const char *tensor_name, int gpu);
DALIAMD_PIPE_API void daliamdOpSpecAddOutput(void *spec, const char *tensor_name, int gpu);
DALIAMD_PIPE_API void daliamdOpSpecAddArgumentInput(void *spec, const char *arg_name, const char *tensor_name);

/* ---- Pipeline (dali/pipeline/pipeline.h; c_api.h:135-160 daliCreatePipeline) ---- */
DALIAMD_PIPE_API void *daliamdPipelineCreate(int batch_size, int num_threads, int device_id, int64_t seed,
                                             int prefetch_queue_depth, int exec_async);
DALIAMD_PIPE_API void daliamdPipelineDestroy(void *pipe);                         /* c_api.h:683 daliDeletePipeline */
DALIAMD_PIPE_API int64_t daliamdPipelineSeed(void *pipe);
DALIAMD_PIPE_API int daliamdPipelineSetAffinity(void *pipe, int on);              /* before Build: NUMA-local worker threads */
DALIAMD_PIPE_API int daliamdPipelineAddOperator(void *pipe, void *spec, const char *instance_name);
/* outputs by tensor name; gpu[i] = 1 for a device output.  Instantiates the operators, runs the graph-level fusions
 * (resample -> crop/mirror/normalize, colour twist -> erase, spectrogram -> mel -> dB), starts the stage threads */
DALIAMD_PIPE_API int daliamdPipelineBuild(void *pipe, const char *const *output_names, const int *gpu, int num_outputs);
DALIAMD_PIPE_API int daliamdPipelineRun(void *pipe);                               /* c_api.h:456 daliRun: schedule one iteration */
/* c_api.h:487-498 daliOutput / daliShareOutput: waits for the oldest scheduled iteration, releases the previous one;
 * the outputs stay valid until the next call */
DALIAMD_PIPE_API int daliamdPipelineOutputs(void *pipe, int *num_outputs);
/* c_api.h:560-590 daliOutputCopy with a caller stream / dali/python/nvidia/dali/plugin/pytorch/torch_utils.py:34-75
 * feed_ndarray(cuda_stream=..., non_blocking): the stream-ordered hand-over.  OutputsOnStream does not wait on the host for
 * the iteration's device work; `consumer_stream` (hipStream_t) is made to wait for it, so whatever the consumer enqueues
 * there afterwards reads finished outputs.  ReleaseOnStream marks the point in that stream behind which the outputs are no
 * longer read: their buffers are not written again before the stream has passed it.  Completion checks (decoder status)
 * of an iteration are raised by the Outputs* call for it or by the next one. */
DALIAMD_PIPE_API int daliamdPipelineOutputsOnStream(void *pipe, void *consumer_stream, int *num_outputs);
DALIAMD_PIPE_API int daliamdPipelineReleaseOnStream(void *pipe, void *consumer_stream);
/* The completion checks of the last OutputsOnStream hand-over, if they were deferred: waits for that iteration's device work
 * and reports its error (the reference raises at the iteration that failed; a stream-ordered hand-over raises at the next
 * Outputs* call, or here - the iterator calls this behind the last batch of an epoch). */
DALIAMD_PIPE_API int daliamdPipelineFlushChecks(void *pipe);
/* Blocks until the device work of every iteration scheduled so far has been enqueued on the pipeline's streams: a device
 * synchronisation behind it covers all of it (the end of a benchmark's timed region). */
DALIAMD_PIPE_API int daliamdPipelineWaitEnqueued(void *pipe);
/* info: [0] device (0 cpu / 1 gpu), [1] dtype (DALIDataType), [2] num_samples, [3] 1 = dense rows */
DALIAMD_PIPE_API int daliamdPipelineOutputInfo(void *pipe, int output, int64_t *info4, char *layout, int layout_len);
DALIAMD_PIPE_API int daliamdPipelineOutputSample(void *pipe, int output, int sample, void **ptr, int64_t *shape8, int *ndim,
                                                 int64_t *row_pitch);
/* every sample of one output in one call: ptrs[n], shapes[n][8], ndims[n], row_pitches[n] */
DALIAMD_PIPE_API int daliamdPipelineOutputSamples(void *pipe, int output, void **ptrs, int64_t *shapes, int *ndims,
                                                  int64_t *row_pitches);
/* c_api.h:276-392 daliSetExternalInput*: one batch for the ExternalSource operator `op_name` (copied) */
DALIAMD_PIPE_API int daliamdPipelineFeedInput(void *pipe, const char *op_name, const void *const *data, const int64_t *shapes,
                                              int ndim, int num_samples, int dtype, const char *layout);
/* c_api.h:711 daliGetReaderMetadata: epoch_size, epoch_size_padded, number_of_shards, shard_id, pad_last_batch, stick_to_shard */
DALIAMD_PIPE_API int daliamdPipelineReaderMeta(void *pipe, const char *op_name, int64_t *meta6);
DALIAMD_PIPE_API int daliamdPipelineReaderNames(void *pipe, char *buf, int len);
/* c_api.h:796-816 daliGetSerializedCheckpoint / daliRestoreFromSerializedCheckpoint (text: one line per stateful operator) */
DALIAMD_PIPE_API int daliamdPipelineCheckpoint(void *pipe, char *buf, int len);
DALIAMD_PIPE_API int daliamdPipelineRestore(void *pipe, const char *checkpoint);
DALIAMD_PIPE_API void *daliamdPipelineStream(void *pipe);                         /* the HIP stream of ring slot 0 (hipStream_t) */

/* ---- instrumentation ---- */
DALIAMD_PIPE_API int daliamdPipelineLastLaunches(void *pipe, char *buf, int len);  /* kernels the last handed-out iteration ran */
DALIAMD_PIPE_API int daliamdPipelineEnableOperatorTiming(void *pipe, int on);      /* before Build: device time per operator */
DALIAMD_PIPE_API int daliamdPipelineOperatorTimes(void *pipe, char *buf, int len); /* "name\tms\n" */
/* host milliseconds per iteration of every operator on its stage thread since the last call, then "<host stage>",
 * "<device stage>", "<slot wait>", "<iterations>"; resets the window */
DALIAMD_PIPE_API int daliamdPipelineOperatorHostTimes(void *pipe, char *buf, int len);
/* encoded-stream cache of the device (decoders.image(cache_type="encoded")): streams resident, bytes used, look-ups that
 * hit, look-ups that missed */
DALIAMD_PIPE_API int daliamdEncodedCacheStats(int device_id, int64_t *out4);

#ifdef __cplusplus
}
#endif
#endif /* DALI_AMD_PIPELINE_H_ */
