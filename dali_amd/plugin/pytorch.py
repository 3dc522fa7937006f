"""PyTorch (ROCm) iterators over dali_amd pipelines: DALIGenericIterator / DALIClassificationIterator
(reference: dali/python/nvidia/dali/plugin/pytorch/__init__.py:43-283, torch_utils.py:34-102).

Each output batch becomes a dense torch tensor on the pipeline's GPU (or in host memory for CPU outputs).
The device copy is ONE copy of the batch (viewed in place in the pipeline's buffer) into a tensor the caller owns
-- the equivalent of the reference's feed_ndarray / copy_to_external -- issued on a side stream of the iterator and
completed before __next__ returns: the pipeline reuses the buffer a few iterations later on its own streams, so the
read must not be left pending on torch's stream (and waiting for torch's CURRENT stream would wait for the
consumer's training step as well).  Everything the copy touches is therefore created ON the side stream: the
destination tensor (the caching allocator keeps one pool per stream, so the block cannot still be in use by kernels
queued on the consumer's stream), the gathered source when the samples are not back to back (torch.stack), and the
copy itself; the side stream is drained before the tensor is handed out and the tensor is recorded on the consumer's
stream so that its block is not recycled under it."""
import numpy as np
import torch

from .. import types
from ..tensors import TensorListGPU
from .base_iterator import LastBatchPolicy, _DaliBaseIterator  # noqa: F401


def feed_ndarray(tensor_or_tl, arr, cuda_stream=None, non_blocking=False, _src=None):
    """Copies a dali_amd TensorList into a preallocated torch tensor (API parity with
    nvidia.dali.plugin.pytorch.feed_ndarray, torch_utils.py:34-75).  `cuda_stream`: the torch stream to copy on
    (default: the current one).  Unless `non_blocking`, the copy has completed on return, i.e. the pipeline may
    reuse the source buffer."""
    stream = None
    if arr.is_cuda:
        stream = cuda_stream if isinstance(cuda_stream, torch.cuda.Stream) else torch.cuda.current_stream(arr.device)
    if _src is not None:
        src = _src
    elif isinstance(tensor_or_tl, TensorListGPU):
        if stream is not None:      # a gathered source (torch.stack) must be ordered in front of the copy
            with torch.cuda.stream(stream):
                src = tensor_or_tl.as_tensor()
        else:
            src = tensor_or_tl.as_tensor()
    else:
        src = torch.from_numpy(np.ascontiguousarray(tensor_or_tl.as_array()))
    assert tuple(src.shape) == tuple(arr.shape), f"Shapes do not match: DALI {tuple(src.shape)} vs torch {tuple(arr.shape)}"
    if arr.is_cuda:
        with torch.cuda.stream(stream):
            arr.copy_(src, non_blocking=True)
        if not non_blocking:
            stream.synchronize()
    else:
        arr.copy_(src)
    return arr


class DALIGenericIterator(_DaliBaseIterator):
    def __init__(self, pipelines, output_map, size=-1, reader_name=None, auto_reset=False, fill_last_batch=None,
                 dynamic_shape=False, last_batch_padded=False, last_batch_policy=LastBatchPolicy.FILL,
                 prepare_first_batch=True):
        assert len(set(output_map)) == len(output_map), "output_map names should be distinct"
        self.output_map = list(output_map)
        self._copy_streams = {}
        super().__init__(pipelines, size, reader_name, auto_reset, fill_last_batch, last_batch_padded,
                         last_batch_policy, prepare_first_batch)

    def _convert(self, outputs_per_pipe, valid_per_pipe):
        result = []
        for g, outs in enumerate(outputs_per_pipe):
            assert len(outs) == len(self.output_map), \
                f"The pipeline returns {len(outs)} outputs but output_map has {len(self.output_map)} names"
            valid = None if valid_per_pipe is None else int(valid_per_pipe[g])
            entry = {}
            for name, tl in zip(self.output_map, outs):
                if isinstance(tl, TensorListGPU):
                    dev = torch.device("cuda", tl.device_id())
                    side = self._copy_streams.get(dev)
                    if side is None:
                        # (default priority: measured, tools/iterator_trace.py - the copy waits 0.17 ms behind the
                        # prefetched batches' kernels, but a high-priority side stream slows the pipeline's own streams
                        # down by more than that: 395 000 against 351 000 images/s)
                        side = self._copy_streams[dev] = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(side):
                        src = tl.as_tensor()    # in-place view of the pipeline's buffer, or a copy gathered on `side`
                        t = torch.empty(src.shape, dtype=src.dtype, device=src.device)    # a block of side's pool
                    feed_ndarray(tl, t, cuda_stream=side, _src=src)     # complete on return: the slot may be reused
                    t.record_stream(torch.cuda.current_stream(dev))
                else:
                    t = torch.from_numpy(np.ascontiguousarray(tl.as_array()))
                if valid is not None and valid < t.shape[0]:
                    t = t[:valid]
                entry[name] = t
            result.append(entry)
        return result


class DALIClassificationIterator(DALIGenericIterator):
    """Returns 2 outputs (data and label) as a list of dicts with keys "data" and "label"."""

    def __init__(self, pipelines, size=-1, reader_name=None, auto_reset=False, fill_last_batch=None,
                 dynamic_shape=False, last_batch_padded=False, last_batch_policy=LastBatchPolicy.FILL,
                 prepare_first_batch=True):
        super().__init__(pipelines, ["data", "label"], size, reader_name, auto_reset, fill_last_batch, dynamic_shape,
                         last_batch_padded, last_batch_policy, prepare_first_batch)
