"""Execution context that stands in for the pieces of TensorFlow the reference leans on:
variable scopes + `tf.get_variable` (a flat-arena VariableStore with the TF variable
names), and `optimizer.compute_gradients` (a reverse-mode Tape whose nodes launch the
backward HIP kernels).  No arithmetic happens here - only bookkeeping of device buffers.
"""
import contextlib
import os

import numpy as np
import torch

from ... import ops

ALIGN = 64  # every parameter starts on a 256-byte boundary of the fp32 arena (128 B of the bf16 shadow)


class Param:
    __slots__ = ('name', 'shape', 'offset', 'size', 'kind', 'store', '_views')

    def __init__(self, name, shape, offset, kind, store):
        self.name, self.shape, self.offset, self.kind, self.store = name, tuple(shape), offset, kind, store
        self.size = int(np.prod(shape))
        self._views = {}

    def _view(self, which, arena):
        # the arenas live as long as the store, so a view is made once (hundreds of these per step otherwise)
        ent = self._views.get(which)
        if ent is None or ent[0] is not arena:
            ent = self._views[which] = (arena, arena[self.offset:self.offset + self.size].view(self.shape))
        return ent[1]

    @property
    def master(self):      # fp32 master copy
        return self._view('p', self.store.params)

    @property
    def w(self):           # operand in the compute dtype (bf16 shadow or the fp32 master)
        return self._view('w', self.store.shadow if self.store.shadow is not None else self.store.params)

    @property
    def grad(self):        # fp32 gradient (accumulated into; zeroed at the start of a step)
        return self._view('g', self.store.grads)


class VariableStore:
    """All trainable variables in one fp32 arena (+ grads, Adam m/v, optional bf16 shadow),
    laid out in creation (= forward) order so backward completes gradient buckets from the
    end of the arena towards the start."""

    def __init__(self, specs, device, compute_dtype, seed=1):
        self.device = device
        self.compute_dtype = compute_dtype
        self.vars = {}
        off = 0
        for name, shape, kind in specs:
            if name in self.vars:
                continue
            self.vars[name] = Param(name, shape, off, kind, self)
            off += -(-int(np.prod(shape)) // ALIGN) * ALIGN
        self.total = off
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=device) if compute_dtype == ops.BF16 else None
        self.adam_t = 0
        self.initialize(seed)

    def initialize(self, seed):
        """Glorot-uniform for matrices (TF default of get_variable), ones / zeros for LN and biases."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.total, np.float32)
        for p in self.vars.values():
            if p.kind == 'glorot':
                if len(p.shape) == 2:
                    fi, fo = p.shape
                else:
                    rf = int(np.prod(p.shape[:-2]))
                    fi, fo = rf * p.shape[-2], rf * p.shape[-1]
                lim = np.sqrt(6.0 / (fi + fo))
                host[p.offset:p.offset + p.size] = rng.uniform(-lim, lim, size=p.size).astype(np.float32)
            elif p.kind == 'ones':
                host[p.offset:p.offset + p.size] = 1.0
        self.params.copy_(torch.from_numpy(host))
        self.sync_shadow()

    def sync_shadow(self):
        if self.shadow is not None:
            ops.cast_f32_to_bf16(self.params, self.shadow)

    def load_params(self, named):
        """named: {tf_variable_name: numpy array} (same names / layouts as the reference's checkpoint)."""
        host = self.params.cpu().numpy()
        for name, arr in named.items():
            p = self.vars[name]
            assert tuple(arr.shape) == p.shape, (name, arr.shape, p.shape)
            host[p.offset:p.offset + p.size] = np.asarray(arr, np.float32).reshape(-1)
        self.params.copy_(torch.from_numpy(host))
        self.sync_shadow()

    def load_slots(self, m_named, v_named):
        """Adam first / second moments by variable name (checkpoint restore)."""
        for arena, named in ((self.m, m_named), (self.v, v_named)):
            host = arena.cpu().numpy()
            for name, arr in named.items():
                p = self.vars[name]
                assert tuple(arr.shape) == p.shape, (name, arr.shape, p.shape)
                host[p.offset:p.offset + p.size] = np.asarray(arr, np.float32).reshape(-1)
            arena.copy_(torch.from_numpy(host))

    def export(self, arena=None):
        host = (self.params if arena is None else arena).float().cpu().numpy()
        return {n: host[p.offset:p.offset + p.size].reshape(p.shape).copy() for n, p in self.vars.items()}

    def export_grads(self):
        return self.export(self.grads)

    def zero_grads(self):
        self.grads.zero_()   # hipMemsetAsync on the current stream

    def get(self, name, shape=None):
        p = self.vars[name]
        if shape is not None:
            assert tuple(shape) == p.shape, (name, shape, p.shape)
        return p


class Tape:
    """Reverse-mode record of one forward pass.  `push(fn)` registers the backward launch of an
    op; gradients of intermediate tensors are kept by tensor identity; an op whose output
    gradient must be added to an existing one either accumulates in its kernel (GEMM) or goes
    through `accumulate`."""

    def __init__(self):
        self.nodes = []
        self.early = []             # nodes whose output gradient exists before backward starts (see push)
        self.grads = {}
        self.keep = {}
        self.relu_outputs = set()   # ids of tensors produced by a fused ReLU: their gradient is gated where it is produced
        self.bias_hooks = {}        # id(dense output) -> BiasGradHook (tools/utils.py): who accumulates that layer's bias gradient

    def push(self, fn, early=False):
        """early=True: the node's output gradient is set by the loss itself (the CTC head), so backward may run it first,
        on the second stream, beside the latency-bound decoder backward instead of after it."""
        (self.early if early else self.nodes).append(fn)

    def grad(self, t):
        return self.grads.get(id(t))

    def pop(self, t):
        self.keep.pop(id(t), None)
        g = self.grads.pop(id(t), None)
        if g is not None and _current and _current[-1].side_writes:
            _current[-1].before_read(g)
        return g

    def has(self, t):
        return id(t) in self.grads

    def set(self, t, g):
        self.keep[id(t)] = t
        self.grads[id(t)] = g

    def accumulate(self, t, g):
        if id(t) in self.grads:
            if _current:
                _current[-1].before_inplace_write(self.grads[id(t)])
            ops.add_inplace(self.grads[id(t)], g)
        else:
            self.set(t, g)

    def backward(self):
        g = _current[-1] if _current else None
        done = g.on_node_done if (g is not None and g.on_node_done is not None) else (lambda: None)
        for fn in self.early:
            if not (g is not None and g.run_early(fn)):
                fn()
            done()
        for fn in reversed(self.nodes):
            fn()
            done()
        self.nodes.clear()
        self.early.clear()


def side_wgrad_rows_from_env():
    v = os.environ.get('NEURST_SIDE_WGRAD_ROWS')
    if not v:
        return None
    lo, _, hi = v.partition(':')
    return (int(lo), int(hi)) if hi else (1, int(lo))


class _Scope:
    """`with g.variable_scope(name):` - a plain object instead of a generator-based context manager: a step opens ~400."""
    __slots__ = ('g', 'name')

    def __init__(self, g, name):
        self.g, self.name = g, name

    def __enter__(self):
        g, name = self.g, self.name
        g.scopes.append(name)
        g.prefixes.append(g.prefixes[-1] + name + '/' if name else g.prefixes[-1])

    def __exit__(self, *exc):
        self.g.scopes.pop()
        self.g.prefixes.pop()
        return False


_SHARED_STREAMS = {}


def shared_stream(kind, device=None):
    """The process's side stream of one kind ('wgrad', 'ctc', 'comm') on `device`.  HIP gives a process a handful of hardware
    queues and streams that share one serialise (Hardware queues): two models in one process (bench.py's LUT and
    COSTT steps, run_lut's train and dev graphs) must not each bring their own set.  Models of one process run one after the
    other on the same host thread, so sharing a stream only adds the stream order they had anyway."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = (kind, dev.index)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class Graph:
    """One model's execution context (TF: default graph + variable scope + train/eval flag)."""

    def __init__(self, store, dtype, seed=0):
        self.store = store
        self.dtype = dtype
        self.seed = seed
        self.step = 0
        self.tape = None
        self.scopes = []
        self.prefixes = ['']        # per open scope: the joined names so far, '/'-terminated (hundreds of lookups per step)
        self.auto_site = 10_000
        self.name_counts = {}
        self.dropout_enabled = True
        self.on_param_grad = None   # callback(Param) once a parameter's gradient kernels are enqueued
        self.produced = set()
        self.on_node_done = None    # callback() after every backward node (all its kernels are enqueued)
        # Weight gradients run on a second HIP stream beside the data-gradient chain: small layers cannot fill the chip on
        # their own, and for large ones each kernel's ramp-up and drain (~12 us per launch, measured) is covered by the
        # other stream's steady state.  side_wgrad_rows = (min, max) input rows of the layers that take part
        # (NEURST_SIDE_WGRAD_ROWS=min:max overrides; 0:0 turns it off).  side_reads holds the storages such kernels may
        # still be reading, so an in-place update of one of them on the main stream first waits for the side stream.
        self.side_streams = os.environ.get('NEURST_SIDE_STREAMS', '1') != '0'   # False: everything on the current stream
        self.wgrad_stream = None
        self.side_wgrad_rows = side_wgrad_rows_from_env() or (1, 1 << 40)
        self.side_reads = set()
        self.side_keep = []
        self.side_used = False
        # The CTC head's chip-wide persistent GEMMs (forward on the CTC stream, data gradient as an early node) run beside the
        # decoder's chain of small launches: with one workgroup on every CU they leave that chain nothing until a tile ends;
        # launched on this many CUs they take longer themselves (they have the slack) and the chain keeps moving.
        # Measured: 128 -> 12.51 against 12.56 ms per step, 160 12.52, 96 and 64 slower (12.62-12.65); 0 = the whole chip.
        self.side_gemm_grid = int(os.environ.get('NEURST_SIDE_GRID', '128'))
        self.in_early = False
        self.early_stream = None     # stream for early backward nodes (default: the weight-gradient stream)
        self.early_used = set()
        self.early_ptrs = {}         # storage -> early stream that writes it
        self.side_writes = set()
        # Encoder-size weight gradients (thousands of rows) are not launched one by one: each has only 12-48 output tiles
        # of 256 x 192, and the whole-K tile kernel runs one workgroup per tile.  They are collected until their tiles fill
        # the 256 CUs and go out as ONE grouped launch (ops.gemm_wgrad_grouped) on the weight-gradient stream.
        # NEURST_GROUP_WGRAD=0 turns it off (one split-K launch per gradient).
        self.group_wgrads = os.environ.get('NEURST_GROUP_WGRAD', '1') != '0'
        # decoder-size gradients (1280 rows) grouped as well: measured 13.93 -> 14.43 ms per step (every flush that an in-place
        # update forces makes the main stream wait for a whole-K launch), so only encoder-size layers take part.  The line sits
        # between the two at every batch size in use: 3072 rows = 7 utterances of 500 frames or 77 label rows of 40 (it was
        # 8192 until 16 utterances - 8000 rows - turned out to fall under it: 9.65 -> 9.08 ms per step there, 7.21 -> 6.41 at 8)
        self.group_min_rows = int(os.environ.get('NEURST_GROUP_MIN_ROWS', '3072'))
        self.group_launch_tiles = int(os.environ.get('NEURST_GROUP_TILES', '256'))
        self.pending_wgrads = []
        self.pending_tiles = 0
        self.pending_reads = {}

    def variable_scope(self, name):
        return _Scope(self, name)

    def default_scope(self, default_name):
        """tf.variable_scope(None, default_name=...): first use keeps the name, later uses in the
        same enclosing scope get _1, _2, ... (reset by begin_forward)."""
        k = (self.prefixes[-1], default_name)
        n = self.name_counts.get(k, 0)
        self.name_counts[k] = n + 1
        return self.variable_scope(default_name if n == 0 else '{}_{}'.format(default_name, n))

    def begin_forward(self, tape=None, step=None):
        self.name_counts = {}
        self.auto_site = 10_000
        self.tape = tape
        if tape is not None:
            self.produced = set()   # names of the variables whose gradient this step's backward produced
        if step is not None:
            self.step = step

    def scope_name(self):
        return self.prefixes[-1][:-1]

    def get_variable(self, name, shape=None):
        return self.store.get(self.prefixes[-1] + name, shape)

    def key(self, site=None):
        if site is None:
            site = self.auto_site
            self.auto_site += 1
        return ops.site_key(self.seed, site, self.step)

    def rate(self, r):
        return float(r) if (self.dropout_enabled and r) else 0.0

    def param_grad_done(self, p):
        self.produced.add(p.name)
        if self.on_param_grad is not None:
            self.on_param_grad(p)

    # ---- second stream for small weight gradients -------------------------------------------------------------
    def side_stream_for(self, rows, device):
        """The stream a weight gradient over `rows` input rows should run on, or None for the current stream."""
        lo, hi = self.side_wgrad_rows
        if not self.side_streams or device.type != 'cuda' or rows < lo or rows > hi:
            return None
        if self.wgrad_stream is None:
            self.wgrad_stream = shared_stream('wgrad', device)
        return self.wgrad_stream

    def run_ahead(self, launch, device):
        """Runs `launch()` (kernels that read no activations, e.g. the dropout decisions of the next attention) on the
        weight-gradient stream - idle during forward - behind everything enqueued so far, and returns (its result, an
        event the consumer's stream has to wait for).  The result is allocated from the main stream's pool as usual."""
        if self.wgrad_stream is None:
            self.wgrad_stream = shared_stream('wgrad', device)
        side = self.wgrad_stream
        side.wait_stream(torch.cuda.current_stream())      # the block handed out may still be read by earlier main-stream kernels
        ops.set_stream(side)
        try:
            out = launch()
        finally:
            ops.set_stream(None)
        ev = torch.cuda.Event()
        ev.record(side)
        return out, ev

    def run_on_side(self, side, launch, *read_tensors):
        """Enqueue `launch()` (kernels that allocate nothing and read `read_tensors`) on `side`, after everything the
        current stream holds so far.  The tensors stay referenced until join_side_streams, so the caching allocator
        cannot hand their memory to a later main-stream kernel while the side stream still reads it."""
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        ops.set_stream(side)
        try:
            launch()
        finally:
            ops.set_stream(None)
        for t in read_tensors:
            self.side_reads.add(t.untyped_storage().data_ptr())
        self.side_keep.append(read_tensors)
        self.side_used = True

    def run_early(self, fn):
        """Run a whole backward node (weight AND data gradient) on the second stream, after everything enqueued so far.
        Returns False when there is no second stream to use."""
        if not self.side_streams or self.store.grads.device.type != 'cuda':
            return False
        if self.wgrad_stream is None:
            self.wgrad_stream = shared_stream('wgrad', self.store.grads.device)
        side = self.early_stream
        if side is None:
            side = self.wgrad_stream
            side.wait_stream(torch.cuda.current_stream())
        else:
            # early_stream produced the node's output gradient itself and already follows everything else the node reads
            # (the activations of the forward, the zeroed gradient arena): no wait for the main stream, the node starts as
            # soon as the loss kernels before it are done
            self.early_used.add(side)
        self.in_early = True
        grid = self.side_gemm_grid if self.early_stream is not None else 0    # only beside the decoder's chain (deferred CTC branch)
        try:
            if grid:
                ops.set_option('gemm256_max_grid', grid)
            with torch.cuda.stream(side):
                fn()
        finally:
            if grid:
                ops.set_option('gemm256_max_grid', 0)
            self.in_early = False
        self.side_used = True
        return True

    def side_written(self, *tensors):
        """Tensors produced on the second stream: the main stream waits for it before it updates one of them in place;
        a plain READ on the main stream must be preceded by that wait too (callers: only gradients that are always
        accumulated into before anything reads them)."""
        st = self.early_stream if self.in_early else None
        for t in tensors:
            p = t.untyped_storage().data_ptr()
            if st is not None:
                self.early_ptrs[p] = st         # produced on the early nodes' own stream: wait for THAT stream only
            else:
                self.side_reads.add(p)
            self.side_writes.add(p)
        self.side_keep.append(tensors)

    def before_read(self, t):
        """The main stream is about to read a gradient: wait if the second stream produced it (and nothing was accumulated
        into it since, which would already have waited)."""
        p = t.untyped_storage().data_ptr()
        if p in self.side_writes:
            self.side_writes.discard(p)
            if p in self.side_reads or p in self.early_ptrs:
                self.before_inplace_write(t)

    # ---- grouped weight gradients ----------------------------------------------------------------------------
    def defer_wgrad(self, dy2, x2, w, layout, bias):
        """Take the weight (+ bias) gradient of one dense layer into the pending group; False = not a case for the grouped
        kernel (the caller launches it the usual way).  param_grad_done is reported when the group is launched."""
        if x2.shape[0] < self.group_min_rows:       # first: the answer for every decoder-size layer (most calls)
            return False
        if (not self.group_wgrads or dy2.device.type != 'cuda' or dy2.dtype != torch.bfloat16
                or not dy2.is_contiguous() or not x2.is_contiguous()):
            return False
        if layout == 'out_in':
            a, b = dy2, x2                      # grad[n_out, n_in] += dy^T . x, bias.grad += column sums of dy
        elif bias is None:
            a, b = x2, dy2                      # grad[n_in, n_out] += x^T . dy
        else:
            return False
        rows, cols = a.shape[1], b.shape[1]
        tiles = -(-rows // 256) * -(-cols // 192)
        if tiles > 64 or rows % 8 or cols % 8:
            return False
        # the pending gradients are a QUEUE of output tiles: whenever 256 are waiting, exactly 256 go out as one launch (one
        # workgroup per CU), the last gradient of the launch possibly only in part - its other tiles lead the next launch
        self.pending_wgrads.append([a, b, w, bias if layout == 'out_in' else None, bias, 0, tiles])
        self.pending_tiles += tiles
        pr = self.pending_reads                 # storage -> number of waiting gradients that read it (kept incrementally: the
        for t in (a, b):                        # queue holds dozens of entries when decoder-size layers take part)
            k = t.untyped_storage().data_ptr()
            pr[k] = pr.get(k, 0) + 1
        while self.pending_tiles >= self.group_launch_tiles:
            self.launch_wgrad_tiles(self.group_launch_tiles)
        return True

    def launch_wgrad_tiles(self, budget):
        """One grouped launch of up to `budget` tiles from the head of the queue (on the weight-gradient stream when there
        is one); gradients whose last tile went out are reported done."""
        probs, ranges, done = [], [], []
        nmax = ops.wgrad_group_max()
        while self.pending_wgrads and budget > 0 and len(probs) < nmax:
            e = self.pending_wgrads[0]
            a, b, w, bs, bias, t0, tiles = e
            take = min(tiles - t0, budget)
            probs.append((a, b, w.grad, (bs.grad if bs is not None else None)))
            ranges.append((t0, t0 + take))
            budget -= take
            self.pending_tiles -= take
            e[5] = t0 + take
            if e[5] == tiles:
                self.pending_wgrads.pop(0)
                done.append((w, bias))
                for t in (a, b):
                    k = t.untyped_storage().data_ptr()
                    n = self.pending_reads.get(k, 0) - 1
                    if n > 0:
                        self.pending_reads[k] = n
                    else:
                        self.pending_reads.pop(k, None)
        if not probs:
            return
        K = probs[0][0].shape[0]
        side = self.side_stream_for(K, probs[0][0].device)

        def launch():
            ops.gemm_wgrad_grouped(probs, ranges)
        if side is not None:
            self.run_on_side(side, launch, *[t for a, b, _, _ in probs for t in (a, b)])
        else:
            launch()
        for w, bias in done:
            self.param_grad_done(w)
            if bias is not None:
                self.param_grad_done(bias)

    def flush_wgrads(self):
        """Launch everything that is pending."""
        while self.pending_wgrads:
            self.launch_wgrad_tiles(self.group_launch_tiles)
        self.pending_reads = {}

    def before_inplace_write(self, t):
        """Main-stream kernels are about to update `t` in place: if a side-stream kernel may still read that storage,
        wait for the side stream's tail (covers every reader enqueued so far)."""
        if self.pending_reads and t.untyped_storage().data_ptr() in self.pending_reads:
            self.flush_wgrads()                 # a gradient still waiting for its group reads this tensor: launch it first
        if self.early_ptrs:
            st = self.early_ptrs.pop(t.untyped_storage().data_ptr(), None)
            if st is not None:
                torch.cuda.current_stream().wait_stream(st)
        if self.side_reads and t.untyped_storage().data_ptr() in self.side_reads:
            ev = torch.cuda.Event()
            ev.record(self.wgrad_stream)
            torch.cuda.current_stream().wait_event(ev)
            self.side_reads.clear()

    def join_side_streams(self):
        """Everything enqueued on the side stream happens-before what the current stream does next (Adam, all-reduce wait)."""
        self.flush_wgrads()
        if self.side_used:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self.side_used = False
        for st in self.early_used:
            torch.cuda.current_stream().wait_stream(st)
        self.early_used.clear()
        self.early_ptrs.clear()
        self.side_reads.clear()
        self.side_writes.clear()
        self.side_keep.clear()


_current = []


def current():
    if not _current:
        raise RuntimeError('no active Graph: layer functions must run inside `with use_graph(g):`')
    return _current[-1]


@contextlib.contextmanager
def use_graph(g):
    _current.append(g)
    try:
        yield g
    finally:
        _current.pop()


def variable_scope(name):
    return current().variable_scope(name)
