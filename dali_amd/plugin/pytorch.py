"""PyTorch (ROCm) iterators over dali_amd pipelines: DALIGenericIterator / DALIClassificationIterator
(reference: dali/python/nvidia/dali/plugin/pytorch/__init__.py:43-283, torch_utils.py:34-102).

Each output batch becomes a dense torch tensor on the pipeline's GPU (or in host memory for CPU outputs).
The device copy is ONE copy of the batch (viewed in place in the pipeline's buffer) into a tensor the caller owns
-- the equivalent of the reference's feed_ndarray / copy_to_external(stream, non_blocking=True) -- and the hand-over is
STREAM ORDERED (round 4; Pipeline.share_outputs(cuda_stream=...) / release_outputs(cuda_stream=...)): the consumer's
current stream is made to wait for the event behind the batch's kernels, the destination tensor, the gathered source
(torch.stack, when the samples are not back to back) and the copy are all enqueued on that stream, and an event recorded
behind them tells the pipeline when the ring slot may be written again.  __next__ therefore returns as soon as the
iteration has been ENQUEUED; nothing waits on the host, and the tensor is an ordinary tensor of the consumer's stream.
(Until round 3 every output was copied on a side stream that was drained before __next__ returned: 0.17 ms per batch
behind the prefetched batches' kernels, 342 000 against 447 000 images/s.)"""
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
        self._handed = []    # (pipeline, torch stream) of the outputs taken by _run_pipe and not yet released
        super().__init__(pipelines, size, reader_name, auto_reset, fill_last_batch, last_batch_padded,
                         last_batch_policy, prepare_first_batch)

    @staticmethod
    def _consumer_stream(pipe):
        dev = pipe.device_id
        if dev is None or dev < 0 or not torch.cuda.is_available():
            return None
        return torch.cuda.current_stream(torch.device("cuda", dev))

    def _run_pipe(self, pipe):
        stream = self._consumer_stream(pipe)
        if stream is None:
            return pipe.run()
        pipe.build()
        pipe._prefetch()
        pipe.release_outputs()
        outs = pipe.share_outputs(cuda_stream=stream.cuda_stream)   # `stream` now waits for the batch; the host does not
        self._handed.append((pipe, stream))
        return outs

    def _epoch_ended(self):
        # outputs taken by _run_pipe for a batch that is dropped (the epoch ends instead of a conversion): nothing reads
        # them, release them without a stream; then the deferred checks of the last batch that WAS handed out
        handed, self._handed = self._handed, []
        for pipe, _stream in handed:
            pipe.release_outputs()
        super()._epoch_ended()

    def _convert(self, outputs_per_pipe, valid_per_pipe):
        result = []
        handed, self._handed = self._handed, []
        # the stream each pipeline's outputs were handed over on (share_outputs made THAT stream wait for the batch): the
        # copies run there and the release is recorded there, whatever the caller's current stream has become since
        stream_of = {id(pipe): stream for pipe, stream in handed}
        try:
            for g, outs in enumerate(outputs_per_pipe):
                assert len(outs) == len(self.output_map), \
                    f"The pipeline returns {len(outs)} outputs but output_map has {len(self.output_map)} names"
                valid = None if valid_per_pipe is None else int(valid_per_pipe[g])
                stream = stream_of.get(id(self._pipes[g])) or self._consumer_stream(self._pipes[g])
                entry = {}
                for name, tl in zip(self.output_map, outs):
                    if isinstance(tl, TensorListGPU):
                        dev = torch.device("cuda", tl.device_id())
                        if stream is None:
                            stream = torch.cuda.current_stream(dev)
                        with torch.cuda.stream(stream):
                            src = tl.as_tensor()    # in-place view of the pipeline's buffer, or a copy gathered on `stream`
                            t = torch.empty(src.shape, dtype=src.dtype, device=src.device)
                        feed_ndarray(tl, t, cuda_stream=stream, non_blocking=True, _src=src)
                    else:
                        t = torch.from_numpy(np.ascontiguousarray(tl.as_array()))
                    if valid is not None and valid < t.shape[0]:
                        t = t[:valid]
                    entry[name] = t
                result.append(entry)
        finally:
            # the copies above are the last reads of the pipelines' buffers: their slots may be written again once the
            # consumer's stream has passed this point
            for pipe, stream in handed:
                pipe.release_outputs(cuda_stream=stream.cuda_stream)
        return result


class DALIClassificationIterator(DALIGenericIterator):
    """Returns 2 outputs (data and label) as a list of dicts with keys "data" and "label"."""

    def __init__(self, pipelines, size=-1, reader_name=None, auto_reset=False, fill_last_batch=None,
                 dynamic_shape=False, last_batch_padded=False, last_batch_policy=LastBatchPolicy.FILL,
                 prepare_first_batch=True):
        super().__init__(pipelines, ["data", "label"], size, reader_name, auto_reset, fill_last_batch, dynamic_shape,
                         last_batch_padded, last_batch_policy, prepare_first_batch)
