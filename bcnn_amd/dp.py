"""Data-parallel helpers of the bench / launcher: the bucketed, overlapped gradient all-reduce.

The C executor reports, from inside bcnn_backward, growing TAIL ranges [first, first + count) of the flat
gradient arena as the nodes that own them finish (bcnn_set_gradient_ready_callback, include/bcnn/bcnn.h).
BucketedAllReduce gathers those ranges into buckets of ~bucket_floats and queues one asynchronous
all-reduce(sum) per bucket, so the collective runs on RCCL's stream while backward keeps computing; finish()
queues the remainder and makes the compute stream wait for every bucket (no host block). The same object
drives a CPU tensor over gloo in tests/test_dp_gloo.py (stream_ctx = nullcontext)."""
import contextlib

import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, arena, bucket_floats, stream_ctx=None):
        self.arena = arena                      # 1-D tensor aliasing the gradient arena
        self.size = arena.numel()
        self.bucket = max(1, int(bucket_floats))
        self.stream_ctx = stream_ctx or contextlib.nullcontext
        self.failed = None                      # first exception raised inside the C callback, if any
        self.buckets = []                       # (lo, hi) of the buckets queued in the current step
        self.begin()

    def begin(self):
        self.lo = self.hi = self.size
        self.works = []
        self.buckets = []

    def _flush(self):
        if self.lo < self.hi:
            with self.stream_ctx():
                self.works.append(dist.all_reduce(self.arena[self.lo:self.hi], async_op=True))
            self.buckets.append((self.lo, self.hi))
            self.hi = self.lo

    def on_ready(self, first, count):
        """the gradient-ready callback: runs inside bcnn_backward; exceptions are recorded, never raised"""
        if self.failed is not None:
            return
        try:
            assert first + count == self.lo, "ranges must arrive as a growing tail: got [%d, %d) below %d" % (
                first, first + count, self.lo)
            self.lo = first
            if self.hi - self.lo >= self.bucket:
                self._flush()
        except Exception as e:  # noqa: BLE001
            self.failed = e

    def finish(self):
        """queue what is left and order the compute stream behind every bucket; returns the first range index
        that was NOT reduced (0 when everything was), so a caller can fall back after a failure"""
        if self.failed is None:
            try:
                self._flush()
            except Exception as e:  # noqa: BLE001
                self.failed = e
        with self.stream_ctx():
            for w in self.works:
                w.wait()
        self.works = []
        return self.hi if self.failed is not None else 0
