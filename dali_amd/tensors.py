"""TensorListCPU / TensorListGPU / TensorCPU / TensorGPU: the objects Pipeline.run() returns
(reference: dali/python/backend_impl.cc:1542-2470).  Device tensors expose __cuda_array_interface__ so
torch / cupy can view them without a copy; the memory belongs to the pipeline and stays valid until the
next run()/outputs() call, like the reference's share_outputs/release_outputs contract."""
import ctypes as C

import numpy as np

from . import types


class _Tensor:
    def __init__(self, owner, ptr, shape, pitch, dtype, layout, gpu):
        self._owner, self._ptr, self._shape, self._pitch = owner, ptr, tuple(shape), pitch
        self.dtype, self._layout, self._gpu = dtype, layout, gpu

    def shape(self):
        return list(self._shape)

    def layout(self):
        return self._layout

    def data_ptr(self):
        return self._ptr

    def _strides(self):
        item = np.dtype(types.to_numpy_type(self.dtype)).itemsize
        strides = [item]
        for d in reversed(self._shape[1:]):
            strides.insert(0, strides[0] * d)
        if self._pitch and len(self._shape) == 3:
            strides[0] = self._pitch
        return tuple(strides)


class TensorCPU(_Tensor):
    def __array__(self, dtype=None, copy=None):
        return self.as_array() if dtype is None else self.as_array().astype(dtype)

    # DLPack (dali/python/backend_impl.cc:623-740): a host tensor exports a copy through numpy's own exporter
    def __dlpack__(self, stream=None, **kwargs):
        return self.as_array().__dlpack__()

    def __dlpack_device__(self):
        return (1, 0)   # kDLCPU

    def as_array(self):
        np_t = np.dtype(types.to_numpy_type(self.dtype))
        n = int(np.prod(self._shape)) if len(self._shape) else 1
        if n == 0:
            return np.zeros(self._shape, np_t)
        buf = (C.c_char * (n * np_t.itemsize)).from_address(self._ptr)
        return np.frombuffer(buf, dtype=np_t).reshape(self._shape).copy()


class TensorGPU(_Tensor):
    @property
    def __cuda_array_interface__(self):
        np_t = np.dtype(types.to_numpy_type(self.dtype))
        return {"shape": self._shape, "typestr": np_t.str, "data": (self._ptr, False), "version": 3,
                "strides": self._strides()}

    def as_torch(self):
        import torch
        if int(np.prod(self._shape)) == 0:
            return torch.empty(self._shape, dtype=_torch_dtype(self.dtype), device="cuda")
        return torch.as_tensor(self, device="cuda")

    def as_cpu(self):
        t = self.as_torch().contiguous().cpu()
        return t.numpy()

    # DLPack: a zero-copy view of the pipeline-owned buffer (valid until the next run()/outputs(), like the
    # __cuda_array_interface__ view), exported through torch so that device type / id and strides follow the
    # consumer's conventions (kDLROCM on this platform).  `stream`: the consumer's stream, as in the protocol.
    def __dlpack__(self, stream=None, **kwargs):
        t = self.as_torch()
        return t.__dlpack__(stream=stream) if stream is not None else t.__dlpack__()

    def __dlpack_device__(self):
        return tuple(int(v) for v in self.as_torch().__dlpack_device__())


def _torch_dtype(dali_type):
    import torch
    return {0: torch.uint8, 4: torch.int8, 5: torch.int16, 6: torch.int32, 7: torch.int64, 8: torch.float16,
            9: torch.float32, 10: torch.float64, 11: torch.bool}[int(dali_type)]


class _TensorList:
    def __init__(self, backend_pipe, idx):
        info = backend_pipe.output_info(idx)
        self._pipe, self._idx = backend_pipe, idx
        self.dtype = [e for e in vars(types.DALIDataType).values()
                      if isinstance(e, types.DALIDataType) and int(e) == info["dtype"]][0]
        self._layout, self._n, self._dense, self._gpu = info["layout"], info["num_samples"], info["dense"], info["gpu"]
        # the per-sample table (pointer, shape, pitch) is fetched when something asks for it: an iterator that only
        # counts batches, or a benchmark loop, pays for one call into the library per output, not for 3 x N Python objects
        self._generation = getattr(backend_pipe, "generation", 0)
        self._sample_table = None

    @property
    def _samples(self):
        if self._sample_table is None:
            if getattr(self._pipe, "generation", 0) != self._generation:
                raise RuntimeError("The outputs of this iteration were released (a later run() / outputs() call): "
                                   "the TensorList can no longer be read")
            self._sample_table = self._pipe.output_samples(self._idx, self._n)
        return self._sample_table

    def __len__(self):
        return self._n

    def layout(self):
        return self._layout

    def shape(self):
        return [list(s[1]) for s in self._samples]

    def is_dense_tensor(self):
        return self._n > 0 and all(s[1] == self._samples[0][1] for s in self._samples)

    def at(self, i):
        ptr, shape, pitch = self._samples[i]
        cls = TensorGPU if self._gpu else TensorCPU
        t = cls(self, ptr, shape, pitch, self.dtype, self._layout, self._gpu)
        return t if self._gpu else t.as_array()

    def __getitem__(self, i):
        ptr, shape, pitch = self._samples[i]
        return (TensorGPU if self._gpu else TensorCPU)(self, ptr, shape, pitch, self.dtype, self._layout, self._gpu)

    def __iter__(self):
        return (self[i] for i in range(self._n))


class TensorListCPU(_TensorList):
    def as_array(self):
        assert self.is_dense_tensor(), "All samples must have the same shape to form a dense array"
        # samples of one shape sitting back to back in the pipeline's buffer (labels, fixed-size outputs): ONE view + one
        # copy instead of a Python object, a ctypes buffer and a copy per sample (1 ms for 256 labels: it was what kept
        # DALIGenericIterator at half the pipeline's rate)
        ptr0, shape, pitch = self._samples[0]
        np_t = np.dtype(types.to_numpy_type(self.dtype))
        nbytes = int(np.prod(shape)) * np_t.itemsize if len(shape) else np_t.itemsize
        if nbytes and not (pitch and len(shape) == 3 and pitch != shape[1] * shape[2] * np_t.itemsize):
            ptrs = np.fromiter((smp[0] for smp in self._samples), np.int64, self._n)
            stride = int(ptrs[1] - ptrs[0]) if self._n > 1 else nbytes      # (samples start at 256-byte multiples)
            if stride >= nbytes and np.array_equal(ptrs, ptr0 + stride * np.arange(self._n, dtype=np.int64)):
                buf = (C.c_char * (stride * (self._n - 1) + nbytes)).from_address(ptr0)
                inner = np.empty(shape, np_t).strides
                return np.ndarray((self._n,) + tuple(shape), np_t, buf, 0, (stride,) + inner).copy()
        return np.stack([self[i].as_array() for i in range(self._n)])

    def as_tensor(self):
        return self.as_array()

    def as_cpu(self):
        return self


class TensorListGPU(_TensorList):
    def device_id(self):
        return getattr(self._pipe, "device_id", 0)

    def _contiguous_view(self):
        """Zero-copy [N, ...] view when the (uniform) samples sit back to back in the pipeline's buffer, else None."""
        import torch
        if self._n == 0:
            return None
        if self._sample_table is None and hasattr(self._pipe, "output_samples_arrays"):
            # whole-batch check on arrays: the per-sample tuples (0.1 ms for 256 samples) are only built when asked for
            if getattr(self._pipe, "generation", 0) != self._generation:
                raise RuntimeError("The outputs of this iteration were released (a later run() / outputs() call): "
                                   "the TensorList can no longer be read")
            ptrs, shapes, ndims, pitches = self._pipe.output_samples_arrays(self._idx, self._n)
            nd = int(ndims[0])
            if (ndims != nd).any() or (shapes[:, :nd] != shapes[0, :nd]).any():
                return None
            ptr0, shape, pitch = int(ptrs[0]), tuple(int(v) for v in shapes[0, :nd]), int(pitches[0])
            np_t = np.dtype(types.to_numpy_type(self.dtype))
            nbytes = int(np.prod(shape)) * np_t.itemsize
            if nbytes == 0 or (pitch and len(shape) == 3 and pitch != shape[1] * shape[2] * np_t.itemsize):
                return None
            if (ptrs != ptr0 + nbytes * np.arange(self._n, dtype=np.int64)).any():
                return None
            iface = {"shape": (self._n,) + shape, "typestr": np_t.str, "data": (ptr0, False), "version": 3}
            return torch.as_tensor(_Interface(iface, self), device="cuda")
        if not self.is_dense_tensor():
            return None
        ptr0, shape, pitch = self._samples[0]
        np_t = np.dtype(types.to_numpy_type(self.dtype))
        nbytes = int(np.prod(shape)) * np_t.itemsize
        if nbytes == 0 or (pitch and len(shape) == 3 and pitch != shape[1] * shape[2] * np_t.itemsize):
            return None
        if any(self._samples[i][0] != ptr0 + i * nbytes for i in range(self._n)):
            return None
        iface = {"shape": (self._n,) + tuple(shape), "typestr": np_t.str, "data": (ptr0, False), "version": 3}
        return torch.as_tensor(_Interface(iface, self), device="cuda")

    def as_tensor(self):
        """Dense [N, ...] torch tensor (device).  Uniform samples that are contiguous in the pipeline's buffer are
        viewed IN PLACE (no copy; the memory belongs to the pipeline and is valid until the next run()/outputs(),
        the reference's share_outputs contract); otherwise they are gathered with one torch.stack."""
        import torch
        assert self.is_dense_tensor(), "All samples must have the same shape to form a dense tensor"
        view = self._contiguous_view()
        return view if view is not None else torch.stack([self[i].as_torch() for i in range(self._n)])

    def as_cpu(self):
        return _HostCopy([self[i].as_cpu() for i in range(self._n)], self.dtype, self._layout)

    def copy_to_external(self, ptr, cuda_stream=None, non_blocking=False):
        """Copies the (uniform) batch into external device memory, like feed_ndarray does."""
        import torch
        t = self.as_tensor().contiguous()
        nbytes = t.numel() * t.element_size()
        dst = torch.as_tensor(_RawDevice(ptr, nbytes), device="cuda")
        dst.copy_(t.view(torch.uint8).reshape(-1), non_blocking=non_blocking)


class _Interface:
    def __init__(self, iface, owner):
        self.__cuda_array_interface__, self._owner = iface, owner


class _RawDevice:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


class _HostCopy:
    """Result of TensorListGPU.as_cpu()."""

    def __init__(self, arrays, dtype, layout):
        self._a, self.dtype, self._layout = arrays, dtype, layout

    def __len__(self):
        return len(self._a)

    def at(self, i):
        return self._a[i]

    def __getitem__(self, i):
        return self._a[i]

    def layout(self):
        return self._layout

    def shape(self):
        return [list(a.shape) for a in self._a]

    def as_array(self):
        return np.stack(self._a)

    as_tensor = as_array
