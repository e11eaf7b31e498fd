"""average_gradients / handle_gradients of st/tools/tftools/gradientTools.py:7-67.

The reference averages per-tower gradients on /cpu:0.  Here each GPU is one process; the flat
fp32 gradient arena is cut into contiguous buckets that are all-reduced (sum) over RCCL on a side
stream as soon as backward has enqueued every gradient kernel of the bucket, overlapping with the
rest of backward; the 1/world_size of the mean is folded into the Adam kernel.
"""
import torch
import os

import torch.distributed as dist

_DEBUG = bool(os.environ.get('NEURST_DEBUG_BUCKETS'))


class GradBucketer:
    def __init__(self, store, bucket_bytes=32 << 20, group=None, comm_stream=None, producer_streams=None):
        self.store = store
        self.producer_streams = producer_streams or (lambda: ())   # other streams that may hold gradient kernels
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.use_stream = store.grads.is_cuda
        self.comm_stream = comm_stream or (torch.cuda.Stream() if (self.use_stream and self.world > 1) else None)
        elems = max(bucket_bytes // 4, 1)
        # buckets follow arena order; a parameter belongs to the bucket of its first element
        self.bounds = []
        start = 0
        while start < store.total:
            end = min(start + elems, store.total)
            self.bounds.append((start, end))
            start = end
        self.param_bucket = {}
        self.bucket_params = [0] * len(self.bounds)
        for p in store.vars.values():
            first = min(p.offset // elems, len(self.bounds) - 1)
            last = min((p.offset + p.size - 1) // elems, len(self.bounds) - 1)
            self.param_bucket[p.name] = (first, last)
            for b in range(first, last + 1):
                self.bucket_params[b] += 1
        self.reset()

    def reset(self):
        self.pending = list(self.bucket_params)
        self.done_params = set()
        self.handles = []
        self.launched = [False] * len(self.bounds)

    def param_done(self, p):
        """Called by backward once every kernel writing p.grad has been enqueued."""
        if self.world == 1 or p.name in self.done_params:
            return
        self.done_params.add(p.name)
        first, last = self.param_bucket[p.name]
        for b in range(first, last + 1):
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if _DEBUG:
            print('[bucketer r%d] launch bucket %d of %d (%d handles so far)' % (dist.get_rank(), b, len(self.bounds), len(self.handles)), flush=True)
        s, e = self.bounds[b]
        view = self.store.grads[s:e]
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            for st in self.producer_streams():
                if st is not None:
                    self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever is left (parameters without gradient this step) and wait for all buckets."""
        if self.world == 1:
            return 1.0
        for b in range(len(self.bounds)):
            self._launch(b)
        for i, h in enumerate(self.handles):
            h.wait()
            if _DEBUG:
                print('[bucketer r%d] handle %d done' % (dist.get_rank(), i), flush=True)
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        scale = 1.0 / self.world
        self.reset()
        return scale


def average_gradients(bucketer):
    """Mean over replicas (gradientTools.py:7-37): returns the scale Adam must apply after the summed all-reduce."""
    return bucketer.finish()


def handle_gradients(args):
    """gradientTools.py:40-67: L2 / clipping are all 0.0 in both templates; anything else is refused
    rather than silently ignored."""
    for k in ('lambda_l2', 'grad_clip_value', 'grad_clip_norm', 'grad_clip_global_norm'):
        if getattr(args, k):
            raise NotImplementedError('{} > 0 is outside the hot path (both templates use 0.0)'.format(k))
