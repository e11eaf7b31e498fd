#!/usr/bin/env python
"""Headline benchmark: LUT train-step encoder frames/s on MI355X (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full LUT train step (a1 splice/down-sample of raw fbank resident in HBM, acoustic
encoder, CTC head + loss, semantic encoder, decoder + label-smoothed CE, KD attention + MSE, backward,
bucketed RCCL all-reduce when N > 1, Adam) on a synthetic batch of 32 x 1500 raw frames per GPU
(T = 500 encoder frames), bf16 operands / fp32 accumulate, dropout 0.1, V = 10000, L = U = Lb = 40.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import neurst_amd  # noqa: E402
neurst_amd.want_hw_queues()   # before the HIP runtime starts: 6 hardware queues for the step's streams (neurst_amd/__init__.py)
import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def lut_step_flops(B, T, L, Lb, V, d=768, din=480, nenc=4, nenc2=4, ndec=4, T2=None, kd=True):
    """ALGORITHMIC forward FLOPs of one train step (SURVEY.md 8d formulas, 1 MAC = 2 flops); x3 for fwd+bwd.  T2 = frames the
    semantic encoder / decoder memory see (COSTT: the padded length after the shrink layer; LUT: T); kd: LUT's KD attention."""
    F = 4 * d
    T2 = T if T2 is None else T2
    enc_blk = lambda rows, Tk: rows * (2 * d * 3 * d + 2 * d * d + 2 * 2 * d * F + 4 * Tk * d)
    fwd = B * T * 2 * din * d + nenc * enc_blk(B * T, T) + B * T * 2 * d * V + nenc2 * enc_blk(B * T2, T2)
    fwd += ndec * B * T2 * 4 * d * d                                    # decoder memory K/V projections
    fwd += ndec * B * L * (8 * d * d + 4 * d * d + 16 * d * d + 4 * L * d + 4 * T2 * d) + B * L * 2 * d * V
    if kd:
        fwd += B * T2 * 4 * d * d + B * Lb * (4 * d * d + 4 * T2 * d)   # KD attention
    return fwd, 3 * fwd


def kernel_sources_digest():
    """sha256 over the kernel sources the library is built from: counters collected for one build say nothing about another"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'neurst_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_label, tag):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/<tag>_hbm_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md 'HBM').  The file records the digest
    of the kernel sources it was collected for; when that differs from the tree this run was built from, the figure is
    stale and is NOT reported (traffic = null, traffic_stale = true)."""
    path = os.path.join(ROOT, 'profiles', tag + '_hbm_traffic.json')
    info = {'traffic_source': 'profiles/%s_hbm_traffic.json' % tag, 'traffic_stale': None, 'traffic_sources_digest': None}
    if not os.path.exists(path):
        return None, info
    d = json.load(open(path))
    info['traffic_sources_digest'] = d.get('kernel_sources_digest')
    info['traffic_stale'] = d.get('kernel_sources_digest') != kernel_sources_digest()
    val = d.get('bytes_per_launch', {}).get(kernel_label)
    return (None if info['traffic_stale'] else val), info


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are issued on (ops.active_stream(): the torch
    current stream, or the side stream small weight gradients are routed to)."""

    def __init__(self):
        self.records = {}
        self.enabled = False

    def wrap_gemm(self, ops):
        orig = ops.gemm
        timer = self

        def gemm(A, B, C, M, N, K, lda, ldb, ldc, trans_a, trans_b, **kw):
            if not timer.enabled:
                return orig(A, B, C, M, N, K, lda, ldb, ldc, trans_a, trans_b, **kw)
            name = ops.gemm_kernel_label(A, C, M, N, K, trans_a, trans_b, kw.get('split_k', 1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = ops.active_stream()
            e0.record(st)
            r = orig(A, B, C, M, N, K, lda, ldb, ldc, trans_a, trans_b, **kw)
            e1.record(st)
            timer.records.setdefault(name, []).append((e0, e1, 2.0 * M * N * K))
            return r
        ops.gemm = gemm
        orig_w = ops.gemm_wgrad_bias

        def gemm_wgrad_bias(A, B, C, opsum, which, M, N, K, lda, ldb, ldc, **kw):
            if not timer.enabled:
                return orig_w(A, B, C, opsum, which, M, N, K, lda, ldb, ldc, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = ops.active_stream()
            e0.record(st)
            r = orig_w(A, B, C, opsum, which, M, N, K, lda, ldb, ldc, **kw)
            e1.record(st)
            timer.records.setdefault(ops.gemm_kernel_label(A, C, M, N, K, True, False, kw.get('split_k', 1), True), []).append((e0, e1, 2.0 * M * N * K))
            return r
        ops.gemm_wgrad_bias = gemm_wgrad_bias
        orig_g = ops.gemm_wgrad_grouped

        def gemm_wgrad_grouped(problems, tile_ranges=None):
            if not timer.enabled:
                return orig_g(problems, tile_ranges)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = ops.active_stream()
            e0.record(st)
            r = orig_g(problems, tile_ranges)
            e1.record(st)
            if tile_ranges is None:
                fl = sum(2.0 * a_.shape[1] * b_.shape[1] * a_.shape[0] for a_, b_, _, _ in problems)
            else:                                          # the launch's 256 x 192 tiles, edge tiles at their true size
                fl = 0.0
                for (a_, b_, _, _), (t0, t1) in zip(problems, tile_ranges):
                    Mm, Nn = a_.shape[1], b_.shape[1]
                    tm_n, tn_n = -(-Mm // 256), -(-Nn // 192)
                    for lt in range(t0, t1):               # the kernel's order: the shorter tile dimension runs fastest
                        tm, tn = (lt % tm_n, lt // tm_n) if tm_n <= tn_n else (lt // tn_n, lt % tn_n)
                        fl += 2.0 * min(256, Mm - 256 * tm) * min(192, Nn - 192 * tn) * a_.shape[0]
            timer.records.setdefault('gemm256_group_kernel', []).append((e0, e1, fl))
            return r
        ops.gemm_wgrad_grouped = gemm_wgrad_grouped

    def wrap_others(self, ops):
        """The other launch classes of the step, each with its ALGORITHMIC work: attention 4 / 10 B H Tq Tk dh flops
        (forward / backward), LayerNorm 4 rows C elt bytes each way (3 in a backward without dropout), the CTC loss
        B T V (4 + elt) bytes (fp32 logits in, gradient out), Adam 30 bytes per parameter (fp32 p, m, v read and written, g read,
        bf16 shadow written).  A class is timed from the first to the last kernel of ONE host call."""
        timer = self
        self.bound = {}

        def wrap(name, label, work, kind):
            orig = getattr(ops, name)

            def fn(*a, **kw):
                if not timer.enabled:
                    return orig(*a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st = ops.active_stream()
                e0.record(st)
                r = orig(*a, **kw)
                e1.record(st)
                lab = label(*a, **kw)
                timer.records.setdefault(lab, []).append((e0, e1, float(work(*a, **kw))))
                timer.bound[lab] = kind
                return r
            setattr(ops, name, fn)
        es = lambda t: t.element_size()
        wrap('mha_fwd', lambda q, k, v, ldq, ldk, ldv, B, H, Tq, Tk, dh, *a, **kw: 'attention forward (dh=%d, Tq=%d, Tk=%d)' % (dh, Tq, Tk),
             lambda q, k, v, ldq, ldk, ldv, B, H, Tq, Tk, dh, *a, **kw: 4.0 * B * H * Tq * Tk * dh, 'mfma')
        wrap('mha_bwd', lambda q, k, v, o, d_o, lse, dq, dk, dv, ldq, ldk, ldv, lddq, lddk, lddv, B, H, Tq, Tk, dh, *a, **kw:
             'attention backward (dh=%d, Tq=%d, Tk=%d)' % (dh, Tq, Tk),
             lambda q, k, v, o, d_o, lse, dq, dk, dv, ldq, ldk, ldv, lddq, lddk, lddv, B, H, Tq, Tk, dh, *a, **kw: 10.0 * B * H * Tq * Tk * dh, 'mfma')
        wrap('dropout_add_layernorm_fwd', lambda x, *a, **kw: 'layernorm forward (%d rows)' % (x.numel() // x.shape[-1]),
             lambda x, *a, **kw: 4.0 * x.numel() * es(x), 'hbm')
        wrap('dropout_add_layernorm_bwd', lambda dout, z, gamma, mean, rstd, dgamma, dbeta, rate, *a, **kw: 'layernorm backward (%d rows)' % (dout.numel() // dout.shape[-1]),
             lambda dout, z, gamma, mean, rstd, dgamma, dbeta, rate, *a, **kw: (4.0 if rate else 3.0) * dout.numel() * es(dout), 'hbm')
        wrap('ctc_loss_fwd_bwd', lambda logits, *a, **kw: 'ctc loss + gradient (prep, frame, alpha-beta, fix)',
             lambda logits, len_logits, labels, len_labels, grad_dtype, *a, **kw: logits.numel() * (4.0 + (2.0 if grad_dtype == ops.BF16 else 4.0)), 'hbm')
        wrap('adam_step', lambda p, *a, **kw: 'adam', lambda p, g, m, v, shadow, *a, **kw: p.numel() * (28.0 + (2.0 if shadow is not None else 0.0)), 'hbm')

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            fl = [f for _, _, f in recs]
            kind = getattr(self, 'bound', {}).get(name, 'mfma')
            out[name] = {'launches': len(recs), 'total_ms': float(sum(ms)), 'avg_us': 1e3 * float(np.mean(ms)), 'bound': kind,
                         'flops_per_launch': float(np.mean(fl)), 'tflops': float(sum(fl) / (sum(ms) * 1e-3) / 1e12)}
        return out


def cpu_baseline(threads):
    """The reference graph on the host cores (TF 1.15 itself cannot run here), on a bounded sample of the benchmark's own
    workload: LUT widths at BASELINE config 2's shape with B = 8 of the 32 utterances (every op of the step is per
    utterance, so frames/s does not depend on B beyond BLAS efficiency), T = 500, L = U = Lb = 40, V = 10000, dropout 0.
    Two legs: (i) the numpy fp32 oracle, forward + hand-written backward + Adam (`value`);
    (ii) the same graph in torch-CPU fp32 (library ops + autograd + the same Adam arithmetic)."""
    import yaml
    from oracle.model import OracleTPE, TPEConfig, init_params, train_step
    from oracle import ops as oops
    from oracle import torch_graph
    from neurst_amd.synthetic import make_batch
    cfg = TPEConfig.from_yaml_dict(yaml.safe_load(open(os.path.join(ROOT, 'configs', 'lut.yaml'))), 10000).zero_dropout()
    B, Traw, Ln = 8, 1500, 40
    batch = make_batch(B=B, Traw=Traw, V=10000, L=Ln, U=Ln, Lb=Ln, bert_dim=768, seed=7)
    batch['feature'], batch['len_fea'] = oops.splice_downsample_batch(batch['feature'], batch['len_fea'], 5, 3)
    T = batch['feature'].shape[1]
    P = init_params(cfg, seed=1, dtype=np.float32)
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    model = OracleTPE(cfg, is_train=True, dropout=False)
    nstep = 3
    train_step(model, P, M, V, batch, 1, 50000, 4e-4, 50000)      # untimed: page in BLAS, allocate
    t0 = time.time()
    for i in range(nstep):
        train_step(model, P, M, V, batch, 2 + i, 50000, 4e-4, 50000)
    dt = (time.time() - t0) / nstep
    # (ii) torch-CPU
    torch.set_num_threads(threads)
    Pt = {k: torch.tensor(v, requires_grad=True) for k, v in init_params(cfg, seed=1, dtype=np.float32).items()}
    Mt = {k: torch.zeros_like(v) for k, v in Pt.items()}
    Vt = {k: torch.zeros_like(v) for k, v in Pt.items()}

    def torch_step(t):
        loss, _ = torch_graph.loss_fn(Pt, cfg, batch)
        grads = torch.autograd.grad(loss, list(Pt.values()), allow_unused=True)
        lr_t = 4e-4 * t / 50000 * math.sqrt(1 - 0.98 ** t) / (1 - 0.9 ** t)
        with torch.no_grad():
            for (k, p), g in zip(Pt.items(), grads):
                if g is None:
                    continue
                Mt[k].mul_(0.9).add_(g, alpha=0.1)
                Vt[k].mul_(0.98).addcmul_(g, g, value=0.02)
                p.sub_(lr_t * Mt[k] / (Vt[k].sqrt() + 1e-9))
        return float(loss.detach())
    torch_step(1)
    t0 = time.time()
    for i in range(nstep):
        torch_step(2 + i)
    dt_t = (time.time() - t0) / nstep
    return {'value': B * T / dt, 'unit': 'encoder frames/s', 'cores': threads, 'kind': 'port',
            'sample': 'numpy fp32 oracle (BLAS threads = cores), LUT train step at BASELINE config 2\'s shape with B=%d of its 32 utterances: %d raw '
                      'frames (T=%d), L=U=Lb=%d, V=10000, dropout 0; %.2f s/step over %d timed steps' % (B, Traw, T, Ln, dt, nstep),
            'torch_cpu': {'value': B * T / dt_t, 'unit': 'encoder frames/s', 'cores': threads, 'kind': 'port',
                          'sample': 'the same step as torch-CPU fp32 autograd (oracle/torch_graph.py) + Adam, %d threads; %.2f s/step over %d timed steps'
                                    % (threads, dt_t, nstep)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--raw-frames', type=int, default=1500)
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'lut.yaml'))
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--ragged', action='store_true')
    ap.add_argument('--profile-tag', default='r03', help='profiles/<tag>_hbm_traffic.json holds the PMC traffic of the dominant kernel')
    ap.add_argument('--bucket-mb', type=float, default=None, help='size of the gradient buckets that are all-reduced as units (default: the config value or 64)')
    ap.add_argument('--serial-streams', action='store_true',
                    help='no second HIP streams (CTC recursion, weight gradients): every kernel has the chip to itself, as in '
                         'the per-kernel timing pass; used for the rocprofv3 runs whose durations profiles/ holds')
    ap.add_argument('--no-secondary', action='store_true', help='skip the COSTT (BASELINE config 3) figure that rides along in the LUT line')
    ap.add_argument('--no-comm', action='store_true',
                    help='N > 1: the step with the gradient all-reduce skipped (same streams, waits and bucket bookkeeping): the comm-off leg '
                         'of the overlap report; the default run measures it as well, after the timed region, and reports both')
    ap.add_argument('--bf16-buckets', action='store_true', help='N > 1: gradient buckets travel as bf16 (cast on the communication stream)')
    ap.add_argument('--rehearse-one-gpu', action='store_true',
                    help='debug: every rank uses cuda:0 and the gloo backend (exercises the N>1 code path on a 1-GPU box; not a measurement)')
    a = ap.parse_args()
    # stdout carries exactly ONE line, the JSON result: everything else that writes to file descriptor 1 on the way (RCCL
    # prints a version banner there when its communicator comes up) is sent to stderr until the result is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get('NEURST_BENCH_WATCHDOG'):      # debug aid: dump every thread's stack if the run is still going after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['NEURST_BENCH_WATCHDOG']), exit=True)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(0 if a.rehearse_one_gpu else local_rank)
    use_dist = world > 1 or 'RANK' in os.environ       # under torch.distributed.run also with one rank: same code path as N > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.rehearse_one_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world)

    from neurst_amd import ops
    from neurst_amd.st.layers.arguments import build_args
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    from neurst_amd.synthetic import make_batch

    timer = KernelTimer()
    if not a.no_kernel_timing:
        timer.wrap_gemm(ops)
        timer.wrap_others(ops)

    V, Ln = 10000, 40
    args = build_args(a.config, dim_output=V, compute_dtype=a.dtype)
    if a.bucket_mb:
        args.grad_bucket_mb = a.bucket_mb
    is_lut = args.model.loss_type == 'ctc_kd_ce'
    if not is_lut:
        Ln = 80
    model = args.Model(GlobalStep(1000), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    if a.serial_streams:
        model.graph.side_streams = False
    if a.bf16_buckets:
        model.bucketer.bf16 = True
    if a.no_comm:
        model.bucketer.skip_comm = True
    B, Traw = a.batch, a.raw_frames
    host = make_batch(B=B, Traw=Traw, V=V, L=Ln, U=40, Lb=40, bert_dim=768 if is_lut else None, seed=1234 + rank, ragged=a.ragged)
    batch = {k: torch.from_numpy(v).cuda() for k, v in host.items()}     # resident in HBM before the timed region
    T = -(-Traw // args.data.downsample)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        model.train_step_async(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out4 = model.train_step_async(batch)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # N > 1: the same steps once more with the collective switched off (every rank, same streams and waits), outside the
    # timed region: step time with and without the exchange shows how much of it backward hides
    comm = None
    if world > 1:
        bk = model.bucketer
        comm = {'buckets': len(bk.bounds), 'bucket_mb': [round((e - s) * 4 / 2 ** 20, 1) for s, e in bk.bounds], 'dtype': 'bf16' if bk.bf16 else 'f32',
                'bytes_per_step_per_rank': int(bk.store.total * (2 if bk.bf16 else 4)), 'comm_on': not bk.skip_comm}
        if not bk.skip_comm:
            bk.skip_comm = True
            for _ in range(2):
                model.train_step_async(batch)
            barrier()
            t1 = time.perf_counter()
            nn = max(a.steps // 2, 3)
            for _ in range(nn):
                model.train_step_async(batch)
            barrier()
            t_off = torch.tensor([(time.perf_counter() - t1) / nn], device='cuda')
            dist.all_reduce(t_off, op=dist.ReduceOp.MAX)
            comm['ms_per_step_comm_off'] = 1e3 * float(t_off.item())
            comm['ms_per_step_comm_on'] = 1e3 * dt / a.steps
            bk.skip_comm = False
    # per-kernel pass, OUTSIDE the timed region (an event pair around every GEMM launch costs ~1.4 ms per step):
    # live HIP-event durations of the dominant kernel on the stream it is launched on.  The second streams are off in
    # this pass: in the timed region above weight gradients and the CTC recursion share the chip with the main
    # stream's kernels, which is faster for the step but makes a single launch's duration say nothing about the kernel.
    ksteps = 0
    if not a.no_kernel_timing:            # every rank runs it: the steps contain the gradient all-reduce
        side_streams = model.graph.side_streams
        model.graph.side_streams = False
        model.train_step_async(batch)     # one untimed step in this mode
        timer.enabled = True
        ksteps = 3
        for _ in range(ksteps):
            model.train_step_async(batch)
        torch.cuda.synchronize()
        timer.enabled = False
        model.graph.side_streams = side_streams
    # BASELINE config 3 rides along (rank 0 of a one-GPU run of the default LUT config): the COSTT step - d = 512, 6 + 6 + 6
    # blocks, shrink layer, SpecAugment, L = 80 - measured AFTER the LUT regions on the same GPU, same batch shape
    secondary = None
    losses = out4.cpu().numpy()
    T2_main = int(model.last['encoded'].shape[1])
    if world == 1 and is_lut and not a.no_secondary and os.path.basename(a.config) == 'lut.yaml':
        # the LUT model is done: its arenas and activations go back to the allocator before the second model is built (with the
        # first model alive the COSTT step - close to the host's launch rate - measured 9.8-10.2 ms here against 9.3 in a
        # process of its own; freed first: 9.05-9.3)
        import gc
        del model, batch, out4
        gc.collect()
        torch.cuda.empty_cache()
        try:
            cargs = build_args(os.path.join(ROOT, 'configs', 'costt.yaml'), dim_output=V, compute_dtype=a.dtype)
            cmodel = cargs.Model(GlobalStep(1000), encoder=cargs.model.encoder.type, decoder=cargs.model.decoder.type, is_train=True, args=cargs)
            chost = make_batch(B=B, Traw=Traw, V=V, L=80, U=40, Lb=40, bert_dim=None, seed=1234, ragged=a.ragged)
            cbatch = {k: torch.from_numpy(v).cuda() for k, v in chost.items()}
            # warm-up in two legs: the first steps run at the full length (the shrink layer's bound is learnt from finished
            # steps), the next ones at the learnt bound (new buffer shapes for the allocator); only then the timed steps.
            # Twelve steps in all: the bound is 1.25 x the maximum over the last EIGHT settled steps, and the first step of a
            # freshly initialised model keeps nearly every frame (478 of 500), so until that step has left the history the
            # shrunk tensor is sized for the full length (12.4 instead of 8.6 ms per step)
            for leg in (2, 10):
                for _ in range(leg):
                    cmodel.train_step_async(cbatch)
                torch.cuda.synchronize()
            c0 = time.perf_counter()
            csteps = max(min(a.steps, 30), 2)
            for _ in range(csteps):
                cout = cmodel.train_step_async(cbatch)
            torch.cuda.synchronize()
            cdt = (time.perf_counter() - c0) / csteps
            T2c = int(cmodel.last['encoded'].shape[1])
            cfl = lut_step_flops(B, T, 80, 40, V, d=cargs.model.encoder.num_cell_units, nenc=cargs.model.encoder.num_blocks,
                                 nenc2=cargs.model.encoder2.num_blocks, ndec=cargs.model.decoder.num_blocks, T2=T2c, kd=False)[1]
            secondary = {'costt': {'workload': 'COSTT train step (configs/costt.yaml: shrink layer, SpecAugment), %d x %d raw frames (T=%d, %d after '
                                               'the shrink layer), V=10000, L=80, dropout 0.1' % (B, Traw, T, T2c),
                                   'ms_per_step': 1e3 * cdt, 'value': B * T / cdt, 'unit': 'encoder frames/s', 'steps': csteps,
                                   'step_achieved_tflops': cfl / cdt / 1e12, 'loss': float(cout.cpu().numpy()[0]),
                                   'shrink_replays': int(cmodel.last.get('shrink_replays', 0)),
                                   'note': 'host-bound (launch rate); the shrunk tensor is sized from earlier steps, no mid-step read-back '
                                           '(st/models/tools/blocks.py:8-38, ShrinkSpeculation)'}}
            del cmodel, cbatch
        except Exception as e:      # the headline line must not depend on the secondary figure
            secondary = {'costt': {'error': repr(e)}}
    if use_dist:
        dist.barrier()
    assert np.isfinite(losses).all(), losses
    step_s = dt / a.steps
    frames = B * T * world

    if rank == 0:
        d = args.model.encoder.num_cell_units
        T2 = T2_main if not is_lut else T      # COSTT: padded length after the shrink layer (data dependent)
        fwd, tot = lut_step_flops(B, T, Ln, 40, V, d=d, nenc=args.model.encoder.num_blocks, nenc2=args.model.encoder2.num_blocks,
                                  ndec=args.model.decoder.num_blocks, T2=T2, kd=is_lut)
        peak = PEAK_BF16_TFLOPS if a.dtype == 'bf16' else PEAK_F32_TFLOPS
        kern = timer.summary()
        roof = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': peak, 'traffic': None,
                'step_algorithmic_tflop': tot / 1e12, 'step_achieved': tot / step_s / 1e12, 'step_frac': tot / step_s / 1e12 / peak}
        if kern:
            # the dominant kernel over EVERY timed launch class (GEMM instantiations, attention, LayerNorm, CTC, Adam)
            dom = max(kern, key=lambda k: kern[k]['total_ms'])
            roof['traffic'], tinfo = pmc_traffic(dom, a.profile_tag)
            roof.update(tinfo)
            roof['kernel_timing'] = ('HIP events around each launch in %d extra steps with the second streams off (one kernel on the '
                                     'chip at a time; same mode as profiles/: bench.py --serial-streams under rocprofv3)' % ksteps)
            if kern[dom]['bound'] == 'hbm':        # work is bytes: achieved in GB/s against the HBM peak
                gbs = kern[dom]['tflops'] * 1e3
                roof.update(bound='hbm', unit='GB/s', peak=PEAK_HBM_GBS, kernel=dom, achieved=gbs, frac=gbs / PEAK_HBM_GBS,
                            avg_launch_us=kern[dom]['avg_us'], bytes_per_launch=kern[dom]['flops_per_launch'])
            else:
                roof.update(kernel=dom, achieved=kern[dom]['tflops'], frac=kern[dom]['tflops'] / peak,
                            avg_launch_us=kern[dom]['avg_us'], flops_per_launch=kern[dom]['flops_per_launch'])
            roof.update(gemm_ms_per_step={k: v['total_ms'] / ksteps for k, v in kern.items() if 'gemm' in k},
                        gemm_tflops={k: v['tflops'] for k, v in kern.items() if 'gemm' in k},
                        other_ms_per_step={k: v['total_ms'] / ksteps for k, v in kern.items() if 'gemm' not in k},
                        other_achieved={k: ('%.0f GB/s' % (v['tflops'] * 1e3) if v['bound'] == 'hbm' else '%.0f TFLOP/s' % v['tflops'])
                                        for k, v in kern.items() if 'gemm' not in k})
        else:
            roof.update(kernel='train step', achieved=roof['step_achieved'], frac=roof['step_frac'])
        res = {
            'metric': 'train-step encoder frames/sec', 'value': frames / step_s, 'unit': 'encoder frames/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * step_s, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'LUT train step (configs/lut.yaml = configs_template/lut.yaml dims), %d x %d raw 80-dim fbank frames per GPU '
                                   '(T=%d), V=10000, L=U=Lb=%d, dropout 0.1, a1 splice on device' % (B, Traw, T, Ln) if is_lut else
                                   'COSTT train step (configs/costt.yaml = configs_template/costt.yaml dims: shrink layer, SpecAugment), %d x %d raw frames '
                                   'per GPU (T=%d, %d after the shrink layer), V=10000, L=%d, dropout 0.1' % (B, Traw, T, T2, Ln),
                       'global_batch': B * world, 'frames_per_step': frames, 'raw_frames_per_s': 3 * frames / step_s,
                       'parallelism': 'dp%d' % world, 'loss': float(losses[0])},
            'roofline': roof,
        }
        if comm is not None:
            res['comm'] = comm
        if secondary is not None:
            res['secondary'] = secondary
        if not a.no_cpu_baseline and world == 1:
            threads = min(os.cpu_count() or 1, 16)
            res['cpu_baseline'] = cpu_baseline(threads)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
