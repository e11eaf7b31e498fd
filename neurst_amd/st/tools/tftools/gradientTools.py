"""average_gradients / handle_gradients of st/tools/tftools/gradientTools.py:7-67.

The reference averages per-tower gradients on /cpu:0.  Here each GPU is one process; the flat
fp32 gradient arena is cut into contiguous buckets that are all-reduced (sum) over RCCL on a side
stream as soon as backward has enqueued every gradient kernel of the bucket, overlapping with the
rest of backward; the 1/world_size of the mean is folded into the Adam kernel.
"""
import torch
import os

import numpy as np

import torch.distributed as dist


def _shared_stream(kind, device):
    from ...models.graph import shared_stream       # one side stream of each kind per process and device
    return shared_stream(kind, device)

_DEBUG = bool(os.environ.get('NEURST_DEBUG_BUCKETS'))


class _StreamDone:
    """Handle of a bucket whose whole reduction (cast, collective, cast back) is already enqueued on the communication
    stream: wait() = the current stream waits for that stream."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)


class _HostStagedWork:
    """Handle of a bucket that gloo reduces in pinned host memory: wait() = the host waits for the reduction, then the
    copy back to the device is enqueued on the communication stream and the current stream is made to wait for it."""

    def __init__(self, work, host, dev, stream, widen=None):
        self.work, self.host, self.dev, self.stream, self.widen = work, host, dev, stream, widen

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            with torch.cuda.stream(self.stream):
                self.dev.copy_(self.host, non_blocking=True)
                if self.widen is not None:          # bf16 buckets: dev is the bf16 staging slice, widen() casts it back into the fp32 arena
                    self.widen()
        torch.cuda.current_stream().wait_stream(self.stream)


class GradBucketer:
    """Completion tracking of the gradient arena in contiguous buckets.

    world > 1: a bucket is all-reduced (sum) on the communication stream as soon as backward has enqueued every kernel
    that writes it.  If `apply_fn(start, end)` is set (it enqueues the optimizer update of arena[start:end] on the
    current stream), a bucket is also UPDATED during backward, on that same side stream, behind its all-reduce: Adam is
    HBM-bound and hides under the matrix-core work of the remaining backward.  The update of a bucket is enqueued one
    step late because the data-gradient GEMM of the layer that completed it still
    reads the weights and is enqueued right after the completion call.  "Late" is counted in backward NODES: buckets
    completed while a node (one layer's backward closure) runs are updated when `node_done()` is called after it, by which
    time every kernel of that node - the data-gradient GEMM included - is in the stream the update waits for."""

    def __init__(self, store, bucket_bytes=32 << 20, group=None, comm_stream=None, producer_streams=None, cuts=(), bf16=None):
        """cuts: arena offsets (elements) at which a bucket must end - the model cuts the arena so that the gradients that
        complete LAST (the first layers of the acoustic encoder, at the start of the arena) form a small bucket of their
        own: what is reduced after backward has ended is then a few MB instead of a whole bucket.
        bf16 (default: NEURST_BF16_BUCKETS=1): a bucket travels as bf16 - cast on the communication stream, summed by the
        collective in bf16, widened back into the fp32 arena (Adam still accumulates in fp32): half the bytes per link,
        at 2^-9 relative rounding per addend (tests state the tolerance)."""
        self.store = store
        self.producer_streams = producer_streams or (lambda: ())   # other streams that may hold gradient kernels
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # the collectives run when there is more than one rank - or, for tests of the RCCL / stream plumbing on a one-GPU
        # box, whenever a process group exists and NEURST_FORCE_BUCKETS is set (an all-reduce over one rank is the identity)
        self.reduce = self.world > 1 or (bool(os.environ.get('NEURST_FORCE_BUCKETS')) and dist.is_available() and dist.is_initialized())
        self.use_stream = store.grads.is_cuda
        self.comm_stream = comm_stream or (_shared_stream('comm', store.grads.device) if (self.use_stream and self.reduce) else None)
        # gloo with device tensors (two ranks rehearsing on ONE GPU; RCCL refuses that): gloo's own CUDA path - a pinned
        # copy per call on pool streams, host threads that synchronise them - hangs on this ROCm stack as soon as a few
        # asynchronous device-tensor all-reduces are in flight, with no code of this package involved
        # (tools/exp/gloo_cuda_check.py: one at a time passes, 2 or 4 or 60 in flight hang, 8 passed once; staged by hand
        # passes).  So gloo only ever sees HOST tensors here: a bucket is copied to pinned memory on the communication
        # stream, reduced there, and copied back before anything waits for its handle.  The host waits for the copy, so
        # this mode has no overlap to measure - it is for checking the N > 1 arithmetic and call order on one GPU.
        self.host_stage = bool(self.reduce and self.use_stream and dist.get_backend(group) == 'gloo')
        self.host_buf = None
        self.bf16 = bool(int(os.environ.get('NEURST_BF16_BUCKETS', '0'))) if bf16 is None else bool(bf16)
        self.g16 = None                  # bf16 staging arena of the bf16 mode (allocated on first use)
        # NEURST_NO_COMM=1 / bench.py --no-comm: everything a step does for the exchange - bucket bookkeeping, the
        # communication stream's waits for the producer streams, the cast kernels of the bf16 mode - except the collective
        # itself: the comm-off leg of the overlap report.  The gradients stay local sums.
        self.skip_comm = bool(int(os.environ.get('NEURST_NO_COMM', '0')))
        self.apply_fn = None
        self.apply_in_backward = True    # False: buckets are only updated in finish(), in launch order, behind their all-reduce
        elems = max(bucket_bytes // 4, 1)
        # buckets follow arena order; a parameter belongs to the bucket of its first element
        self.bounds = []
        edges = sorted({int(c) for c in cuts if 0 < int(c) < store.total} | {store.total})
        start = 0
        for edge in edges:
            while start < edge:
                end = min(start + elems, edge)
                self.bounds.append((start, end))
                start = end
        self.param_bucket = {}
        self.bucket_params = [0] * len(self.bounds)
        starts = [b[0] for b in self.bounds]
        import bisect
        for p in store.vars.values():
            first = bisect.bisect_right(starts, p.offset) - 1
            last = bisect.bisect_right(starts, p.offset + p.size - 1) - 1
            self.param_bucket[p.name] = (first, last)
            for b in range(first, last + 1):
                self.bucket_params[b] += 1
        self.reset()

    def reset(self):
        self.pending = list(self.bucket_params)
        self.done_params = set()
        self.handles = [None] * len(self.bounds)
        self.launched = [False] * len(self.bounds)
        self.applied = [False] * len(self.bounds)
        self.order = []              # buckets in the order their all-reduce was launched
        self.ready = []              # completed buckets whose update is held back until the running backward node is done

    def _side_stream(self):
        if self.use_stream and self.comm_stream is None:
            self.comm_stream = _shared_stream('comm', self.store.grads.device)
        return self.comm_stream

    def param_done(self, p):
        """Called by backward once every kernel writing p.grad has been enqueued."""
        if not self.reduce and self.apply_fn is None:
            return
        first, last = self.param_bucket[p.name]
        if p.name in self.done_params:
            # a variable used by two layers: its first user already reported it.  Harmless while the bucket is untouched
            # (kernels run in stream order), an error once the bucket has been reduced or updated without this gradient.
            if any(self.applied[b] or (self.launched[b] and self.reduce) for b in range(first, last + 1)):
                raise RuntimeError('gradient of shared variable {} produced after its bucket was reduced/updated'.format(p.name))
            return
        self.done_params.add(p.name)
        for b in range(first, last + 1):
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._complete(b)

    def _complete(self, b):
        if self.reduce:
            self._launch(b)
        if self.apply_fn is not None and self.apply_in_backward:
            self.ready.append(b)

    def node_done(self):
        """Called by the tape after every backward node."""
        if self.ready and self.apply_fn is not None:
            for b in self.ready:
                self._apply(b)
            self.ready = []

    def _wait_for_producers(self, stream):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        stream.wait_event(ev)
        for st in self.producer_streams():
            if st is not None:
                stream.wait_stream(st)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        self.order.append(b)
        if _DEBUG:
            print('[bucketer r%d] launch bucket %d of %d' % (dist.get_rank(), b, len(self.bounds)), flush=True)
        s, e = self.bounds[b]
        view = self.store.grads[s:e]
        if self.bf16 and self.g16 is None:
            self.g16 = torch.empty(self.store.total, dtype=torch.bfloat16, device=view.device)
        wire = self.g16[s:e] if self.bf16 else view        # what the collective sees

        def narrow():
            if view.is_cuda:
                from .... import ops
                ops.cast_f32_to_bf16(view, wire)
            else:
                wire.copy_(view)                           # host tensors: test plumbing only

        def widen():
            if view.is_cuda:
                from .... import ops
                ops.cast_bf16_to_f32(wire, view)
            else:
                view.copy_(wire)

        def collective(t):
            if self.skip_comm:
                return None
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.host_stage:
            if self.host_buf is None:
                self.host_buf = torch.empty(self.store.total, dtype=wire.dtype).pin_memory()
            self._wait_for_producers(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                if self.bf16:
                    narrow()
                self.host_buf[s:e].copy_(wire, non_blocking=True)
            self.comm_stream.synchronize()
            self.handles[b] = _HostStagedWork(collective(self.host_buf[s:e]), self.host_buf[s:e], wire, self.comm_stream,
                                              widen if self.bf16 else None)
        elif self.comm_stream is not None:
            self._wait_for_producers(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                if self.bf16:
                    narrow()
                    h = collective(wire)
                    if h is not None:
                        h.wait()                           # RCCL work: makes the communication stream wait, not the host
                    widen()
                    self.handles[b] = _StreamDone(self.comm_stream)
                else:
                    self.handles[b] = collective(view)
        else:
            if self.bf16:
                narrow()
                h = collective(wire)
                if h is not None:
                    h.wait()
                widen()
            else:
                self.handles[b] = collective(view)

    def _apply(self, b):
        """Optimizer update of bucket b behind everything enqueued so far (current stream, producer streams, its all-reduce)."""
        if self.applied[b]:
            return
        self.applied[b] = True
        s, e = self.bounds[b]
        side = self._side_stream()
        if side is None:                      # host tensors (gloo tests): everything is synchronous
            if self.handles[b] is not None:
                self.handles[b].wait()
            self.apply_fn(s, e)
            return
        self._wait_for_producers(side)
        with torch.cuda.stream(side):
            if self.handles[b] is not None:
                self.handles[b].wait()        # NCCL work: makes `side` wait, not the host
            self.apply_fn(s, e)

    def finish(self):
        """Launch whatever is left (parameters without gradient this step), wait for all buckets; with apply_fn set also
        update every bucket not updated yet.  Returns the 1/world of the mean (already used by apply_fn if that is set)."""
        scale = 1.0 / self.world
        if not self.reduce and self.apply_fn is None:
            return scale
        if self.reduce:
            for b in range(len(self.bounds)):
                self._launch(b)
        if self.apply_fn is not None:
            # in launch order: the update of an early bucket runs while the all-reduces of the later ones are still in flight
            for b in self.order + [b for b in reversed(range(len(self.bounds))) if b not in self.order]:
                self._apply(b)
        else:
            for i, h in enumerate(self.handles):
                if h is not None:
                    h.wait()
                    if _DEBUG:
                        print('[bucketer r%d] handle %d done' % (dist.get_rank(), i), flush=True)
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()
        return scale


def average_gradients(bucketer):
    """Mean over replicas (gradientTools.py:7-37): returns the scale Adam must apply after the summed all-reduce."""
    return bucketer.finish()


def handle_gradients(args, store=None, scale=1.0, produced=None):
    """gradientTools.py:40-67 on the averaged gradients (`scale` * the summed arena): L2 penalty on every variable without
    'bias' in its name (:47-48), tf.clip_by_global_norm (:60-65).  grad_clip_value / grad_clip_norm only touch variables
    with 'lstm' in their name (:51-56) - none of this model's - so they are accepted and change nothing.  Returns the
    scale Adam has to apply (the clip factor rides on it).  Called without a store (graph construction): checks only.
    `produced`: names of the variables whose gradient exists this step - the reference skips the penalty where `g is None`
    (:48; such a variable is then not updated at all); None = every variable has one."""
    lam = float(args.lambda_l2 or 0.0)
    clip = float(args.grad_clip_global_norm or 0.0)
    if lam < 0 or clip < 0:
        raise ValueError('lambda_l2 / grad_clip_global_norm must be >= 0')
    if store is None or (lam == 0.0 and clip == 0.0):
        return scale
    from .... import ops
    if lam > 0.0:
        for p in store.vars.values():
            if 'bias' not in p.name and (produced is None or p.name in produced):
                # scale * (g + (lam / scale) v) = scale * g + lam * v
                ops.axpy_f32(store.grads[p.offset:p.offset + p.size], store.params[p.offset:p.offset + p.size], lam / scale)
    if clip > 0.0:
        norm = scale * float(np.sqrt(ops.sumsq_f32(store.grads)))      # one 8 KB read back: clipping needs the norm on the host
        scale = scale * clip / max(norm, clip)
    return scale
