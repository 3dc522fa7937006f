"""Pipeline / pipeline_def: the graph-definition front end (reference:
dali/python/nvidia/dali/pipeline.py:97-2330).  The graph is a list of OpSpecs handed to the C++ host
framework (dali_amd/host), which owns the executor, the thread pool and the HIP stream."""
import functools
import inspect
import threading

import numpy as np

from . import _backend as _b
from . import tensors, types
from .data_node import DataNode

_tls = threading.local()


class Pipeline:
    """See nvidia.dali.Pipeline.  Supported arguments: batch_size, num_threads, device_id, seed,
    prefetch_queue_depth, exec_pipelined/exec_async (async execution = the prefetching worker thread), set_affinity
    (worker threads bound to the CPUs of the GPU's NUMA node; DALI_AFFINITY_MASK overrides the set);
    the remaining reference arguments are accepted and ignored."""

    def __init__(self, batch_size=-1, num_threads=-1, device_id=-1, seed=-1, exec_pipelined=True,
                 prefetch_queue_depth=2, exec_async=True, bytes_per_sample=0, set_affinity=False, max_streams=None,
                 default_cuda_stream_priority=None, *, enable_memory_stats=False, enable_checkpointing=False,
                 checkpoint=None, py_num_workers=1, py_start_method="fork", py_callback_pickler=None,
                 output_dtype=None, output_ndim=None, exec_dynamic=False, experimental_exec_dynamic=None,
                 stream_policy=None, concurrency=None):
        if batch_size is None or batch_size < 1:
            raise ValueError("`batch_size` must be a positive integer")
        self._max_batch_size = int(batch_size)
        self._num_threads = int(num_threads) if num_threads and num_threads > 0 else 1
        self._device_id = 0 if device_id is None or device_id < 0 else int(device_id)
        self._seed = -1 if seed is None else int(seed)
        if isinstance(prefetch_queue_depth, dict):
            prefetch_queue_depth = max(prefetch_queue_depth.get("cpu_size", 2), prefetch_queue_depth.get("gpu_size", 2))
        self._prefetch_queue_depth = int(prefetch_queue_depth) if exec_pipelined else 1
        self._exec_async = bool(exec_async and exec_pipelined)
        self._set_affinity = bool(set_affinity)
        self._operator_timing = False
        self._enable_checkpointing = enable_checkpointing
        self._restore_from = checkpoint
        self._ops = []          # (schema_name, instance_name, device, init_args, inputs, arg_inputs, outputs)
        self._names = set()
        self._counter = 0
        self._outputs = None
        self._built = False
        self._backend = None
        self._scheduled = 0
        self._consumed = 0
        self._input_callbacks = []
        self._first_run = True
        self._held = None

    # ------------------------------------------------------------------ definition scope
    @staticmethod
    def current():
        return getattr(_tls, "pipeline", None)

    def __enter__(self):
        self._prev = Pipeline.current()
        _tls.pipeline = self
        return self

    def __exit__(self, *exc):
        _tls.pipeline = self._prev
        return False

    @property
    def max_batch_size(self):
        return self._max_batch_size

    batch_size = max_batch_size

    @property
    def num_threads(self):
        return self._num_threads

    @property
    def device_id(self):
        return self._device_id

    @property
    def seed(self):
        return self._backend.seed() if self._backend else self._seed

    # ------------------------------------------------------------------ graph construction
    def _unique_name(self, schema_name, requested):
        if requested is not None:
            if requested in self._names:
                raise RuntimeError(f'Operator instance name "{requested}" is not unique')
            self._names.add(requested)
            return requested
        while True:
            name = f"__{schema_name}_{self._counter}"
            self._counter += 1
            if name not in self._names:
                self._names.add(name)
                return name

    def _add_op(self, schema_name, device, init_args, inputs, arg_inputs, num_outputs, name=None):
        if self._built:
            raise RuntimeError("The pipeline is already built; operators cannot be added")
        inst = self._unique_name(schema_name, name)
        out_dev = "cpu" if device == "cpu" else "gpu"
        outs = [DataNode(f"{inst}[{i}]" if num_outputs > 1 else inst, out_dev, source=inst) for i in range(num_outputs)]
        self._ops.append((schema_name, inst, device, dict(init_args), list(inputs), dict(arg_inputs), outs))
        return outs

    def _to_gpu(self, node):
        key = ("_gpu_copy", node.name)
        cache = self.__dict__.setdefault("_gpu_copies", {})
        if key not in cache:
            inst = self._unique_name("_CopyToGpu", None)
            out = DataNode(node.name, "gpu", source=inst)
            self._ops.append(("_CopyToGpu", inst, "mixed", {}, [node], {}, [out]))
            cache[key] = out
        return cache[key]

    def set_outputs(self, *outputs):
        flat = []
        for o in outputs:
            if isinstance(o, (list, tuple)):
                flat.extend(o)
            else:
                flat.append(o)
        for o in flat:
            if not isinstance(o, DataNode):
                raise TypeError(f"Pipeline outputs must be DataNodes, got {type(o).__name__}")
        self._outputs = flat

    def define_graph(self):
        raise NotImplementedError("Use set_outputs() / @pipeline_def, or override define_graph()")

    def build(self):
        if self._built:
            return
        if self._outputs is None:
            with self:
                outs = self.define_graph()
            self.set_outputs(*(outs if isinstance(outs, (list, tuple)) else [outs]))
        be = _b.BackendPipeline(self._max_batch_size, self._num_threads, self._device_id, self._seed,
                                self._prefetch_queue_depth, self._exec_async, self._set_affinity)
        if getattr(self, "_operator_timing", False):
            be.enable_operator_timing()
        # Operators that do not contribute to an output are pruned, like the reference's graph lowering does
        # (pipeline.cc: "prune" of unused operators; `preserve=True` keeps one alive)
        by_inst = {op[1]: op for op in self._ops}
        needed, stack = set(), [o.source for o in self._outputs]
        stack += [op[1] for op in self._ops if op[3].get("preserve")]
        while stack:
            inst = stack.pop()
            if inst in needed or inst not in by_inst:
                continue
            needed.add(inst)
            op = by_inst[inst]
            stack += [n.source for n in op[4]] + [n.source for n in op[5].values()]
        for schema_name, inst, device, init_args, inputs, arg_inputs, outs in self._ops:
            if inst not in needed:
                continue
            spec = _b.OpSpec(schema_name)
            spec.add_arg("device", device)
            for k, v in init_args.items():
                spec.add_arg(k, v)
            for n in inputs:
                spec.add_input(n.name, n.device)
            for k, n in arg_inputs.items():
                spec.add_argument_input(k, n.name)
            for o in outs:
                spec.add_output(o.name, o.device)
            be.add_operator(spec, inst)
        be.build([(o.name, o.device) for o in self._outputs])
        self._backend = be
        self._built = True
        if self._restore_from:
            be.restore(self._restore_from)

    # ------------------------------------------------------------------ execution
    def _feed_callbacks(self):
        for cb in self._input_callbacks:
            cb(self)

    def schedule_run(self):
        self.build()
        self._feed_callbacks()
        self._backend.run()
        self._scheduled += 1

    def _prefetch(self):
        while self._scheduled - self._consumed < self._prefetch_queue_depth:
            self.schedule_run()

    def share_outputs(self, cuda_stream=None):
        """`cuda_stream` (a hipStream_t handle, e.g. torch.cuda.Stream.cuda_stream): stream-ordered hand-over - the host
        does not wait for the iteration's device work, that stream does; release with release_outputs(cuda_stream)."""
        if self._scheduled <= self._consumed:
            raise RuntimeError("There are no scheduled runs; call schedule_run() first")
        try:
            n = self._backend.outputs() if cuda_stream is None else self._backend.outputs_on_stream(int(cuda_stream))
        except _b.PipelineError:
            # an iteration that failed is consumed all the same (the executor has popped it and goes on with the next one):
            # the pipeline stays usable, the error belongs to this call only.  Anything raised in FRONT of the executor's
            # call (a bad stream handle, an interrupt) has consumed nothing and does not touch the accounting.
            self._consumed += 1
            raise
        self._consumed += 1
        outs = []
        for i in range(n):
            info = self._backend.output_info(i)
            outs.append((tensors.TensorListGPU if info["gpu"] else tensors.TensorListCPU)(self._backend, i))
        self._held = outs
        return tuple(outs)

    def release_outputs(self, cuda_stream=None):
        """`cuda_stream`: the outputs are still being read by work enqueued on that stream; their buffers are not
        reused before the stream has passed this point."""
        if cuda_stream is not None and self._held is not None:
            self._backend.release_on_stream(int(cuda_stream))
        self._held = None

    def flush_checks(self):
        """Stream-ordered hand-over (share_outputs(cuda_stream=...)): the completion checks of an iteration whose device work
        was still running when it was handed out are raised by the NEXT share_outputs call - or by this one, which waits for
        that iteration (the iterators call it behind the last batch of an epoch)."""
        if self._built:
            self._backend.flush_checks()

    def outputs(self):
        self.release_outputs()
        return self.share_outputs()

    def run(self):
        """Runs the pipeline and returns the outputs of the oldest scheduled iteration (prefetching
        `prefetch_queue_depth` iterations ahead, like the reference's pipelined executor)."""
        self.build()
        self._prefetch()
        out = self.outputs()
        return out

    def feed_input(self, data_node, data, layout=None):
        self.build()
        name = data_node if isinstance(data_node, str) else data_node.source
        def to_host(d):
            # anything that speaks DLPack or the array interfaces (torch / cupy tensors on either device, DALI tensors):
            # external_source.py:95-130 accepts them; the operator copies from host memory, so device data comes down
            if isinstance(d, (bytes, bytearray)):
                return np.frombuffer(d, np.uint8)
            if isinstance(d, np.ndarray):
                return d
            if hasattr(d, "__dlpack__") or hasattr(d, "__cuda_array_interface__"):
                import torch
                t = torch.from_dlpack(d) if hasattr(d, "__dlpack__") else torch.as_tensor(d, device="cuda")
                return t.detach().cpu().numpy()
            return np.asarray(d)
        if isinstance(data, np.ndarray) and data.dtype != object:
            arrays = [np.ascontiguousarray(data[i]) for i in range(data.shape[0])]
        elif not isinstance(data, (list, tuple)) and (hasattr(data, "__dlpack__") or hasattr(data, "__cuda_array_interface__")):
            whole = to_host(data)      # one tensor = a batch along its first axis
            arrays = [np.ascontiguousarray(whole[i]) for i in range(whole.shape[0])]
        else:
            arrays = [np.ascontiguousarray(to_host(d)) for d in data]
        if not arrays:
            raise ValueError("Cannot feed an empty batch")
        dt = arrays[0].dtype
        nd = arrays[0].ndim
        for a in arrays:
            if a.dtype != dt or a.ndim != nd:
                raise TypeError("All samples in a batch must have the same dtype and number of dimensions")
        self._backend.feed_input(name, arrays, int(types.from_numpy_type(dt)), layout)

    def reader_meta(self, name=None):
        self.build()
        if name is not None:
            return self._backend.reader_meta(name)
        return {n: self._backend.reader_meta(n) for n in self._backend.reader_names()}

    def epoch_size(self, name=None):
        meta = self.reader_meta(name)
        if name is not None:
            return meta["epoch_size_padded"]
        return {k: v["epoch_size_padded"] for k, v in meta.items()}

    def checkpoint(self, filename=None):
        """Serialized state of the stateful operators (readers, random generators)."""
        self.build()
        # drain what is in flight so that the checkpoint is taken at an iteration boundary
        while self._scheduled > self._consumed:
            self.outputs()
        cpt = self._backend.checkpoint()
        if filename:
            with open(filename, "w") as f:
                f.write(cpt)
        return cpt

    def enable_operator_timing(self):
        """Benchmarks: time the device work of every mixed / gpu operator with events on its stream (call before
        build()); read the averages with operator_device_times()."""
        if self._built:
            raise RuntimeError("enable_operator_timing() must be called before the pipeline is built")
        self._operator_timing = True

    def operator_device_times(self):
        return self._backend.operator_times()

    def operator_host_times(self):
        """Host milliseconds per iteration of every operator on its stage thread since the last call."""
        return self._backend.operator_host_times()

    def executed_kernels(self):
        """Names of the device kernels the most recent iteration launched (testing aid)."""
        return self._backend.last_launches()

    def reset(self):
        pass

    def empty(self):
        return self._scheduled == self._consumed

    def _check_api_type_scope(self, *_):
        import contextlib
        return contextlib.nullcontext()


_PIPELINE_KWARGS = set(inspect.signature(Pipeline.__init__).parameters) - {"self"}


def pipeline_def(fn=None, **pipeline_kwargs):
    """Decorator that converts a graph-definition function into a pipeline factory
    (reference: pipeline.py:2179-2330)."""

    def actual_decorator(func):
        @functools.wraps(func)
        def create_pipeline(*args, **kwargs):
            fn_params = inspect.signature(func).parameters
            ctor = dict(pipeline_kwargs)
            fn_kwargs = {}
            for k, v in kwargs.items():
                if k in _PIPELINE_KWARGS:
                    ctor[k] = v
                    if k in fn_params:
                        fn_kwargs[k] = v
                else:
                    fn_kwargs[k] = v
            pipe = Pipeline(**ctor)
            with pipe:
                outs = func(*args, **fn_kwargs)
                if isinstance(outs, (list, tuple)):
                    pipe.set_outputs(*outs)
                elif outs is not None:
                    pipe.set_outputs(outs)
            return pipe

        create_pipeline._is_pipeline_def = True
        return create_pipeline

    return actual_decorator(fn) if fn is not None else actual_decorator
