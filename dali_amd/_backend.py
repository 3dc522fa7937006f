"""ctypes binding of the C++ host framework's flat C API (dali_amd/host/c_api.cpp) -- the counterpart
of the reference's pybind11 module `nvidia.dali.backend_impl`."""
import ctypes as C
import json

from . import _capi as capi


def _lib():
    capi.kernels()      # torch + the kernel library first (single HIP runtime, see _capi.kernels)
    lib = capi.host()
    if not getattr(lib, "_pipeline_api_ready", False):
        lib.daliamdOpSpecCreate.restype = C.c_void_p
        lib.daliamdPipelineCreate.restype = C.c_void_p
        lib.daliamdPipelineStream.restype = C.c_void_p
        lib.daliamdPipelineSeed.restype = C.c_int64
        lib.daliamdPipelineCreate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int]
        for f in ("daliamdOpSpecDestroy", "daliamdPipelineDestroy"):
            getattr(lib, f).argtypes = [C.c_void_p]
            getattr(lib, f).restype = None
        lib.daliamdOpSpecAddArgInt.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.daliamdOpSpecAddArgBool.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.daliamdOpSpecAddArgFloat.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        lib.daliamdOpSpecAddArgStr.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        lib.daliamdOpSpecAddArgIntVec.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.c_int]
        lib.daliamdOpSpecAddArgFloatVec.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        lib.daliamdOpSpecAddArgStrVec.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
        lib.daliamdOpSpecAddInput.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.daliamdOpSpecAddOutput.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.daliamdOpSpecAddArgumentInput.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        lib.daliamdPipelineAddOperator.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        lib.daliamdPipelineBuild.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int]
        lib.daliamdPipelineRun.argtypes = [C.c_void_p]
        lib.daliamdPipelineOutputs.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.daliamdPipelineOutputsOnStream.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        lib.daliamdPipelineReleaseOnStream.argtypes = [C.c_void_p, C.c_void_p]
        lib.daliamdPipelineWaitEnqueued.argtypes = [C.c_void_p]
        lib.daliamdPipelineFlushChecks.argtypes = [C.c_void_p]
        lib.daliamdPipelineOutputInfo.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_char_p, C.c_int]
        lib.daliamdPipelineOutputSample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        lib.daliamdPipelineOutputSamples.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.daliamdPipelineFeedInput.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                                 C.c_int, C.c_int, C.c_int, C.c_char_p]
        lib.daliamdPipelineReaderMeta.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        for f in ("daliamdPipelineReaderNames", "daliamdPipelineCheckpoint", "daliamdPipelineLastLaunches"):
            getattr(lib, f).argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.daliamdPipelineRestore.argtypes = [C.c_void_p, C.c_char_p]
        lib.daliamdPipelineStream.argtypes = [C.c_void_p]
        lib.daliamdPipelineSeed.argtypes = [C.c_void_p]
        lib._pipeline_api_ready = True
    return lib


class PipelineError(RuntimeError):
    """An error the executor reports through the C ABI (an operator failed in the iteration that was being handed out, ...):
    a RuntimeError, as before; its own type so that a caller can tell it from an error raised in front of the call."""


def check(rc):
    if rc != 0:
        msg = _lib().daliamdHostGetLastErrorMessage()
        raise PipelineError(msg.decode(errors="replace") if msg else "unknown error")


def _string_out(fn, *args):
    lib = _lib()
    need = fn(*args, None, 0)
    if need < 0:
        check(1)
    buf = C.create_string_buffer(max(need, 1))
    rc = fn(*args, buf, need)
    if rc < 0:
        check(1)
    return buf.value.decode()


_schema_cache = {}


def schema_names():
    return [s for s in _string_out(_lib().daliamdSchemaList).split("\n") if s]


def get_schema(name):
    if name not in _schema_cache:
        lib = _lib()
        lib.daliamdSchemaInfo.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        _schema_cache[name] = json.loads(_string_out(lib.daliamdSchemaInfo, name.encode()))
    return _schema_cache[name]


class OpSpec:
    def __init__(self, schema_name):
        self._lib = _lib()
        self.schema_name = schema_name
        self._h = self._lib.daliamdOpSpecCreate(schema_name.encode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.daliamdOpSpecDestroy(self._h)
            self._h = None

    def add_arg(self, name, value):
        import numpy as np
        from . import types
        lib, h, n = self._lib, self._h, name.encode()
        if isinstance(value, types._DALIEnum):
            value = int(value)
        if isinstance(value, (bool, np.bool_)):
            lib.daliamdOpSpecAddArgBool(h, n, int(value))
        elif isinstance(value, (int, np.integer)):
            lib.daliamdOpSpecAddArgInt(h, n, int(value))
        elif isinstance(value, (float, np.floating)):
            lib.daliamdOpSpecAddArgFloat(h, n, float(value))
        elif isinstance(value, str):
            lib.daliamdOpSpecAddArgStr(h, n, value.encode())
        elif isinstance(value, (list, tuple, np.ndarray)):
            vals = [int(v) if isinstance(v, types._DALIEnum) else v for v in (value.tolist() if isinstance(value, np.ndarray) else value)]
            if len(vals) and all(isinstance(v, str) for v in vals):
                arr = (C.c_char_p * len(vals))(*[v.encode() for v in vals])
                lib.daliamdOpSpecAddArgStrVec(h, n, arr, len(vals))
            elif all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in vals):
                arr = (C.c_int64 * len(vals))(*[int(v) for v in vals])
                lib.daliamdOpSpecAddArgIntVec(h, n, arr, len(vals))
            else:
                arr = (C.c_double * len(vals))(*[float(v) for v in vals])
                lib.daliamdOpSpecAddArgFloatVec(h, n, arr, len(vals))
        else:
            raise TypeError(f"Unsupported value for argument `{name}` of operator `{self.schema_name}`: "
                            f"{type(value).__name__}")

    def add_input(self, name, device):
        self._lib.daliamdOpSpecAddInput(self._h, name.encode(), 1 if device == "gpu" else 0)

    def add_output(self, name, device):
        self._lib.daliamdOpSpecAddOutput(self._h, name.encode(), 1 if device == "gpu" else 0)

    def add_argument_input(self, arg, tensor_name):
        self._lib.daliamdOpSpecAddArgumentInput(self._h, arg.encode(), tensor_name.encode())


def encoded_cache_stats(device_id=0):
    """Encoded-stream cache of the device (decoders.image(cache_type="encoded")): resident streams, bytes used, lookups
    that hit / missed so far."""
    out = (C.c_int64 * 4)()
    check(_lib().daliamdEncodedCacheStats(int(device_id), out))
    return dict(streams=int(out[0]), bytes_used=int(out[1]), hits=int(out[2]), misses=int(out[3]))


class BackendPipeline:
    def __init__(self, batch_size, num_threads, device_id, seed, prefetch_queue_depth, exec_async, set_affinity=False):
        self._lib = _lib()
        self.device_id = max(0, int(device_id))
        self.generation = 0      # of the outputs currently handed out (TensorLists fetch their sample tables lazily)
        self._h = self._lib.daliamdPipelineCreate(batch_size, num_threads, device_id, seed, prefetch_queue_depth,
                                                  1 if exec_async else 0)
        if not self._h:
            check(1)
        if set_affinity:
            check(self._lib.daliamdPipelineSetAffinity(C.c_void_p(self._h), 1))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.daliamdPipelineDestroy(self._h)
            self._h = None

    def enable_operator_timing(self):
        check(self._lib.daliamdPipelineEnableOperatorTiming(C.c_void_p(self._h), 1))

    def operator_times(self):
        """{operator instance name: average device ms} (after enable_operator_timing)."""
        out = {}
        for line in _string_out(self._lib.daliamdPipelineOperatorTimes, C.c_void_p(self._h)).split("\n"):
            if line:
                k, v = line.rsplit("\t", 1)
                out[k] = float(v)
        return out

    def operator_host_times(self):
        """{operator instance name: host ms per iteration on its stage thread} since the last call (+ the "<host stage>",
        "<device stage>", "<slot wait>" sums and "<iterations>"); resets the window."""
        out = {}
        for line in _string_out(self._lib.daliamdPipelineOperatorHostTimes, C.c_void_p(self._h)).split("\n"):
            if line:
                k, v = line.rsplit("\t", 1)
                out[k] = float(v)
        return out

    def add_operator(self, spec, name):
        check(self._lib.daliamdPipelineAddOperator(self._h, spec._h, name.encode()))

    def build(self, outputs):
        names = (C.c_char_p * len(outputs))(*[n.encode() for n, _ in outputs])
        gpu = (C.c_int * len(outputs))(*[1 if d == "gpu" else 0 for _, d in outputs])
        check(self._lib.daliamdPipelineBuild(self._h, names, gpu, len(outputs)))

    def run(self):
        check(self._lib.daliamdPipelineRun(self._h))

    def outputs(self):
        n = C.c_int(0)
        self.generation += 1
        check(self._lib.daliamdPipelineOutputs(self._h, C.byref(n)))
        return n.value

    def outputs_on_stream(self, stream_handle):
        """Stream-ordered hand-over: `stream_handle` (hipStream_t as an integer) waits for the iteration, the host does not."""
        n = C.c_int(0)
        self.generation += 1
        check(self._lib.daliamdPipelineOutputsOnStream(self._h, C.c_void_p(stream_handle), C.byref(n)))
        return n.value

    def wait_enqueued(self):
        """Every scheduled iteration's device work is on the streams when this returns (then synchronise the device)."""
        check(self._lib.daliamdPipelineWaitEnqueued(self._h))

    def release_on_stream(self, stream_handle):
        check(self._lib.daliamdPipelineReleaseOnStream(self._h, C.c_void_p(stream_handle)))

    def flush_checks(self):
        """Raises the deferred completion checks (decoder status) of the last stream-ordered hand-over, if any."""
        check(self._lib.daliamdPipelineFlushChecks(self._h))

    def output_info(self, idx):
        info = (C.c_int64 * 4)()
        layout = C.create_string_buffer(16)
        check(self._lib.daliamdPipelineOutputInfo(self._h, idx, info, layout, 16))
        return dict(gpu=bool(info[0]), dtype=int(info[1]), num_samples=int(info[2]), dense=bool(info[3]),
                    layout=layout.value.decode())

    def output_sample(self, idx, i):
        ptr, shape, ndim, pitch = C.c_void_p(), (C.c_int64 * 8)(), C.c_int(0), C.c_int64(0)
        check(self._lib.daliamdPipelineOutputSample(self._h, idx, i, C.byref(ptr), shape, C.byref(ndim), C.byref(pitch)))
        return ptr.value or 0, tuple(shape[:ndim.value]), int(pitch.value)

    def output_samples(self, idx, n):
        """(ptr, shape, row_pitch) of every sample of output `idx` with ONE call into the library."""
        import numpy as np
        if n == 0:
            return []
        ptrs = np.zeros(n, np.uint64)
        shapes = np.zeros((n, 8), np.int64)
        ndims = np.zeros(n, np.int32)
        pitches = np.zeros(n, np.int64)
        check(self._lib.daliamdPipelineOutputSamples(self._h, idx, ptrs.ctypes.data_as(C.c_void_p),
                                                     shapes.ctypes.data_as(C.c_void_p), ndims.ctypes.data_as(C.c_void_p),
                                                     pitches.ctypes.data_as(C.c_void_p)))
        pl, sl, nl, tl = ptrs.tolist(), shapes.tolist(), ndims.tolist(), pitches.tolist()
        return [(pl[i], tuple(sl[i][:nl[i]]), tl[i]) for i in range(n)]

    def output_samples_arrays(self, idx, n):
        """The same table as numpy arrays (pointers, shapes [n, 8], ndims, pitches): whole-batch questions - are the samples
        uniform and back to back? - without a Python object per sample."""
        import numpy as np
        ptrs = np.zeros(max(n, 1), np.uint64)
        shapes = np.zeros((max(n, 1), 8), np.int64)
        ndims = np.zeros(max(n, 1), np.int32)
        pitches = np.zeros(max(n, 1), np.int64)
        if n:
            check(self._lib.daliamdPipelineOutputSamples(self._h, idx, ptrs.ctypes.data_as(C.c_void_p),
                                                         shapes.ctypes.data_as(C.c_void_p), ndims.ctypes.data_as(C.c_void_p),
                                                         pitches.ctypes.data_as(C.c_void_p)))
        return ptrs[:n].astype(np.int64), shapes[:n], ndims[:n], pitches[:n]

    def feed_input(self, name, arrays, dtype, layout):
        import numpy as np
        n = len(arrays)
        ndim = arrays[0].ndim if n else 1
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrays])
        shapes = np.array([a.shape for a in arrays], np.int64).reshape(-1)
        check(self._lib.daliamdPipelineFeedInput(self._h, name.encode(), ptrs,
                                                 shapes.ctypes.data_as(C.POINTER(C.c_int64)), ndim, n, dtype,
                                                 (layout or "").encode()))

    def reader_meta(self, name):
        meta = (C.c_int64 * 6)()
        check(self._lib.daliamdPipelineReaderMeta(self._h, name.encode(), meta))
        return dict(epoch_size=int(meta[0]), epoch_size_padded=int(meta[1]), number_of_shards=int(meta[2]),
                    shard_id=int(meta[3]), pad_last_batch=int(meta[4]), stick_to_shard=int(meta[5]))

    def reader_names(self):
        return [s for s in _string_out(self._lib.daliamdPipelineReaderNames, self._h).split("\n") if s]

    def checkpoint(self):
        return _string_out(self._lib.daliamdPipelineCheckpoint, self._h)

    def restore(self, cpt):
        check(self._lib.daliamdPipelineRestore(self._h, cpt.encode()))

    def last_launches(self):
        return [s for s in _string_out(self._lib.daliamdPipelineLastLaunches, self._h).split("\n") if s]

    def stream(self):
        return self._lib.daliamdPipelineStream(self._h)

    def seed(self):
        return int(self._lib.daliamdPipelineSeed(self._h))
