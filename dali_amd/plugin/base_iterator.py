"""Framework-agnostic iterator logic: epoch accounting over sharded readers and the last-batch
policies (behavioural counterpart of dali/python/nvidia/dali/plugin/base_iterator.py:55-608)."""
import enum
import math

import numpy as np


class LastBatchPolicy(enum.Enum):
    """What to do with the last batch when the shard size is not a multiple of the batch size."""
    FILL = 0      # return a full batch (padded by the reader or wrapping into the next epoch)
    DROP = 1      # drop the incomplete batch
    PARTIAL = 2   # return the incomplete batch


class _EpochBook:
    """Per-epoch bookkeeping for pipelines that read one shard each.

    `size` is the number of samples each pipeline contributes in the current epoch (a multiple of the batch
    size unless policy is DROP/PARTIAL); shard sizes follow floor((k+1)*N/S) - floor(k*N/S) and rotate with the
    reader when it does not stick to its shard."""

    def __init__(self, metas, batch_size, policy):
        first = metas[0]
        for key, what in (("epoch_size", "size value"), ("number_of_shards", "`num_shards` argument set")):
            if any(m[key] != first[key] for m in metas):
                raise AssertionError(f"Reader Operator should have the same {what} in all the pipelines.")
        for key, what in (("pad_last_batch", "`pad_last_batch` argument set"),
                          ("stick_to_shard", "`stick_to_shard` argument set")):
            vals = [bool(m[key]) for m in metas]
            if any(vals) and not all(vals):
                raise AssertionError(f"Reader Operator should have the same {what} in all the pipelines.")
        self.n = first["epoch_size"]
        self.shards = first["number_of_shards"]
        self.padded = bool(first["pad_last_batch"])
        self.sticky = bool(first["stick_to_shard"])
        self.ids = np.array([m["shard_id"] for m in metas], np.int64)
        self.bs = batch_size
        self.policy = policy
        k = np.arange(self.shards, dtype=np.int64)
        self.initial_sizes = ((k + 1) * self.n // self.shards) - (k * self.n // self.shards)
        self.sizes = self.initial_sizes.copy()       # current size of the shard each *position* reads
        self.carry = np.zeros(self.shards, np.int64)  # samples already taken from the current shard (FILL wrap)
        if policy == LastBatchPolicy.DROP:
            self.size = self.n // self.shards
        elif self.padded:
            self.size = first["epoch_size_padded"] // self.shards
        else:
            self.size = math.ceil(math.ceil(self.n / self.shards) / batch_size) * batch_size

    def should_drop_next(self, counter):
        return self.policy == LastBatchPolicy.DROP and bool(np.any(self.carry + counter + self.bs > self.sizes))

    def valid_in_last_batch(self, counter):
        """PARTIAL: how many samples of the batch that ended at `counter` are real, per pipeline."""
        left = self.bs - (counter - self.initial_sizes[self.ids])
        return np.where(left < self.bs, np.maximum(left, 0), self.bs)

    def next_epoch(self, counter):
        """Called at the end of an epoch; returns the starting counter of the next one."""
        wrap = self.policy == LastBatchPolicy.FILL and not self.padded
        start = 0
        if wrap:
            taken = self.carry + (counter - self.carry.min())
            self.carry = taken - self.sizes        # read-ahead into the following shard
            start = int(self.carry.min())
        if not self.sticky:
            self.ids = (self.ids + 1) % self.shards
        if wrap:
            if not self.sticky:
                self.sizes = np.roll(self.sizes, 1)
            todo = self.sizes - self.carry
            self.size = math.ceil(int(todo.max()) / self.bs) * self.bs
            if self.size == 0:   # everything of the next epoch was already consumed: skip it
                self.carry[:] = 0
                start = 0
                self.sizes = np.roll(self.sizes, 1)
                self.size = math.ceil(int(self.sizes.max()) / self.bs) * self.bs
        return start


class _DaliBaseIterator:
    def __init__(self, pipelines, size=-1, reader_name=None, auto_reset=False, fill_last_batch=None,
                 last_batch_padded=False, last_batch_policy=LastBatchPolicy.FILL, prepare_first_batch=True):
        if not isinstance(pipelines, (list, tuple)):
            pipelines = [pipelines]
        assert len(pipelines) > 0, "Number of provided pipelines has to be at least 1"
        self._pipes = list(pipelines)
        self._num_gpus = len(self._pipes)
        self.batch_size = self._pipes[0].max_batch_size
        assert all(p.max_batch_size == self.batch_size for p in self._pipes), \
            "All pipelines should have the same batch size set"
        if fill_last_batch is not None:
            last_batch_policy = LastBatchPolicy.FILL if fill_last_batch else LastBatchPolicy.PARTIAL
        if not isinstance(last_batch_policy, LastBatchPolicy):
            raise ValueError("last_batch_policy must be a LastBatchPolicy")
        self._policy = last_batch_policy
        self._auto_reset = "yes" if auto_reset in (True, "yes") else ("no" if auto_reset in (False, None, "no") else auto_reset)
        self._reader_name = reader_name
        self._last_batch_padded = last_batch_padded
        assert self._reader_name is None or size == -1, "When reader_name is provided, size should not be set"
        assert self._reader_name is not None or size != 0, "`size` must not be 0 without a reader_name"
        self._size = int(size)
        self._counter = 0
        self._book = None
        self._ever_consumed = False
        for p in self._pipes:
            p.build()
        if self._reader_name:
            self._book = _EpochBook([p.reader_meta(self._reader_name) for p in self._pipes], self.batch_size,
                                    self._policy)
            self._size = self._book.size
            self._last_batch_padded = self._book.padded
        self._first_batch = None
        if prepare_first_batch:
            try:
                self._first_batch = self._fetch()
            except StopIteration:
                raise RuntimeError("It seems that there is no data in the pipeline. This may happen if "
                                   "`last_batch_policy` is set to PARTIAL and the requested batch size is greater than "
                                   "the shard size.")

    # ---- to be provided by the framework plugin
    def _convert(self, outputs_per_pipe, valid_per_pipe):
        raise NotImplementedError

    def _run_pipe(self, pipe):
        """One iteration of one pipeline; a plugin whose consumer works on a device stream overrides this with the
        stream-ordered hand-over (Pipeline.share_outputs(cuda_stream=...))."""
        return pipe.run()

    def _epoch_ended(self):
        """Hook: the epoch is over, nothing handed out so far is followed by another fetch."""
        for p in self._pipes:
            p.flush_checks()

    # ---- epoch logic
    def _fetch(self):
        if self._size > 0 and self._counter >= self._size:
            self._end_epoch()
        drop = (self._book.should_drop_next(self._counter) if self._book
                else self._policy == LastBatchPolicy.DROP and self._size > 0 and
                self._counter + self._num_gpus * self.batch_size > self._size)
        outs = [self._run_pipe(p) for p in self._pipes]
        self._counter += self.batch_size if self._book else self._num_gpus * self.batch_size
        if drop:
            self._end_epoch()
        valid = None
        if self._policy == LastBatchPolicy.PARTIAL and self._size > 0 and self._counter > self._size:
            if self._book:
                valid = self._book.valid_in_last_batch(self._counter)
            else:
                over = self._counter - self._size
                per = np.full(self._num_gpus, self.batch_size, np.int64)
                for g in range(self._num_gpus - 1, -1, -1):  # trailing pipelines hold the padding
                    cut = min(self.batch_size, over)
                    per[g] -= cut
                    over -= cut
                valid = per
        return self._convert(outs, valid)

    def _end_epoch(self):
        self._epoch_ended()
        if self._auto_reset == "yes":
            self.reset()
        raise StopIteration

    def reset(self):
        """Resets the iterator after a full epoch (ignored mid-epoch, like the reference)."""
        if self._size < 0 or self._counter >= self._size or self._policy == LastBatchPolicy.DROP:
            if self._book:
                self._counter = self._book.next_epoch(self._counter)
                self._size = self._book.size
            elif self._policy == LastBatchPolicy.FILL and not self._last_batch_padded and self._size > 0:
                self._counter = self._counter % self._size
            else:
                self._counter = 0

    def __next__(self):
        self._ever_consumed = True
        if self._first_batch is not None:
            batch, self._first_batch = self._first_batch, None
            return batch
        return self._fetch()

    next = __next__

    def __iter__(self):
        if self._counter != 0 and self._ever_consumed and self._first_batch is None and \
                self._size > 0 and self._counter >= self._size:
            self.reset()
        return self

    @property
    def size(self):
        return self._size

    def __len__(self):
        per_step = self.batch_size if self._reader_name else self._num_gpus * self.batch_size
        if self._policy != LastBatchPolicy.DROP:
            return math.ceil(self.size / per_step)
        return self.size // per_step
