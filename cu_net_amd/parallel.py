"""Data parallelism for the CU-Net hot path: one process per GPU, RCCL over xGMI.

Replaces `torch.nn.DataParallel(net).cuda()` (cu-net.py:59).  What DataParallel does every
iteration -- broadcast all parameters, scatter the batch, gather outputs, reduce gradients to GPU 0
through one python thread per GPU -- becomes:
  * parameters/buffers live resident on every rank (one broadcast at start, `broadcast_state`);
  * every rank feeds its own shard of the batch and computes its own loss (BatchNorm statistics stay
    per rank, exactly as DataParallel replicas keep them per replica -- no SyncBN in the reference);
  * the only exchange is a SUM all-reduce of the flat fp32 gradient arena, cut into one bucket per
    U-Net (include/cunet.h, cunet_backward_ex) and issued on a side stream as soon as backward has
    enqueued the last kernel writing that bucket, so communication overlaps the rest of backward;
    the 1/world_size factor is folded into the fused RMSprop kernel.
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs over "gloo" on CPU tensors,
which is how the bucketing logic is tested without GPUs.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


class BucketAllReducer:
    """SUM all-reduce of contiguous [begin, begin+count) ranges of one flat tensor, bucket by bucket."""

    def __init__(self, buckets: Sequence[Tuple[int, int]], process_group=None, overlap: bool = True):
        self.buckets = list(buckets)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if process_group is not None else 1
        self.overlap = overlap
        # gloo on GPU tensors (two ranks sharing one device in tests/test_gpu_dp.py): the bucket is staged through the host
        self._host_staged = process_group is not None and dist.get_backend(process_group) == 'gloo'
        self._stream = None
        self.reduced: List[int] = []        # bucket ids in the order they were reduced (tests)

    def begin_step(self):
        self.reduced = []

    def reduce_bucket(self, flat: torch.Tensor, b: int, join_side=None):
        """Called when every kernel writing bucket b has been enqueued.  `join_side(stream_ptr)` makes a raw
        stream wait for the library's internal weight-gradient stream (cunet_side_stream_join)."""
        import ctypes
        begin, count = self.buckets[b]
        self.reduced.append(b)
        if self.pg is None or count == 0:
            return
        g = flat[begin:begin + count]
        if self.overlap and flat.is_cuda:
            dev = flat.device
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._stream.wait_event(ev)
            if join_side is not None:
                join_side(ctypes.c_void_p(self._stream.cuda_stream))
            with torch.cuda.stream(self._stream):
                if self._host_staged:
                    h = g.to('cpu')                      # (synchronises the communication stream)
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
                    g.copy_(h)
                else:
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            if join_side is not None and flat.is_cuda:
                join_side(ctypes.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream))
            if self._host_staged and flat.is_cuda:
                h = g.to('cpu')
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
                g.copy_(h)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg)

    def finish(self, flat: torch.Tensor):
        """Make the current stream wait for every outstanding bucket reduction."""
        if self._stream is not None and flat.is_cuda:
            torch.cuda.current_stream(flat.device).wait_stream(self._stream)

    def covers(self, numel: int) -> bool:
        """True when the buckets tile [0, numel) without gaps larger than alignment padding or overlap."""
        pos = 0
        for begin, count in self.buckets:
            if begin < pos:
                return False
            pos = begin + count
        return pos <= numel


def broadcast_state(tensors: Sequence[torch.Tensor], src: int = 0, process_group=None):
    """One-time replacement of DataParallel's per-iteration replicate."""
    if process_group is None:
        return
    staged = dist.get_backend(process_group) == 'gloo'
    for t in tensors:
        if staged and t.is_cuda:
            h = t.to('cpu')
            dist.broadcast(h, src, group=process_group)
            t.copy_(h)
        else:
            dist.broadcast(t, src, group=process_group)


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk of the batch owned by `rank` (DataParallel chunks dim 0 in device order)."""
    per = (global_batch + world - 1) // world
    lo = min(rank * per, global_batch)
    return lo, min(lo + per, global_batch)
