"""Thin Python wrappers over the C ABI (include/neurst_hip.h).  torch tensors are used
only as device buffers (`data_ptr()`) and for the current HIP stream; no torch op does
arithmetic on the hot path.  Every wrapper raises if the kernel library reports an error.
"""
import math
import os

import numpy as np
import ctypes

import torch

from . import lib as _lib
from .lib import BF16, F32, MASK_CAUSAL, MASK_KEYPAD, MASK_NONE  # noqa: F401

_L = None


def L():
    global _L
    if _L is None:
        _L = _lib.load()
    return _L


class _Symbols:
    """C entry points as plain instance attributes after their first use (an attribute read instead of a call to L() and
    a ctypes symbol lookup on every launch: the wrappers below run ~300 times per train step)."""

    def __getattr__(self, name):
        fn = getattr(L(), name)
        setattr(self, name, fn)
        return fn


_C = _Symbols()
_DT = {torch.bfloat16: BF16, torch.float32: F32}


def _s():
    """Raw hipStream_t the next launch goes to: the stream set by `on_stream`, else torch's current stream (the C-level
    query: ~0.2 us instead of ~4 us through torch.cuda.Stream)."""
    return _override_raw or _raw_stream(_cur_dev())


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_dev = torch._C._cuda_getDevice
_override_raw = 0
_override_stream = None


def set_stream(stream):
    """Route the wrappers' launches to `stream` (a torch.cuda.Stream) until set_stream(None).  Cheaper than
    `with torch.cuda.stream(...)`; only for launches that allocate nothing (torch's allocator keeps using its own
    current stream)."""
    global _override_raw, _override_stream
    _override_stream = stream
    _override_raw = 0 if stream is None else stream.cuda_stream


def set_option(name, value):
    """Kernel-selection switch of the library (neurst_set_option: tests and A/B timing)."""
    _chk(_C.neurst_set_option(name.encode(), int(value)), 'set_option(%s)' % name)


def active_stream():
    return _override_stream if _override_stream is not None else torch.cuda.current_stream()


def _p(t):
    return 0 if t is None else t.data_ptr()


def dt_of(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f'unsupported dtype {t.dtype}')


def torch_dtype(dt):
    return torch.bfloat16 if dt == BF16 else torch.float32


def _chk(rc, what):
    if rc != 0:
        _lib.check(rc, what)


def _c(t):
    assert t.is_contiguous(), 'tensor must be contiguous'
    return t


# ---- dropout keys (same construction as oracle/ops.py:site_key) -----------------
def fmix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def site_key(seed, site, step=0):
    return fmix32((int(seed) * 0x9E3779B9 + int(site) * 0x85EBCA6B + int(step) * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF)


# ---- a5 timing signal table (host, fp32; st/layers/common_attention_v1.py:56-69) ---
_signal_cache = {}


def timing_signal_table(length, channels, device):
    key = (length, channels, str(device))
    if key not in _signal_cache:
        f = np.float32
        position = np.arange(length).astype(np.float32)
        nts = channels // 2
        inc = f(math.log(1.0e4 / 1.0)) / (f(nts) - f(1))
        inv = f(1.0) * np.exp(np.arange(nts).astype(np.float32) * -inc)
        st = position[:, None] * inv[None, :]
        sig = np.concatenate([np.sin(st), np.cos(st)], axis=1)
        sig = np.pad(sig, [[0, 0], [0, channels % 2]]).astype(np.float32)
        _signal_cache[key] = torch.from_numpy(sig).to(device)
    return _signal_cache[key]


# ---- a1 ------------------------------------------------------------------------------
def splice_downsample(x, len_raw, num_context, rate, out_dtype):
    B, Traw, F = x.shape
    Tout = -(-Traw // rate)
    out = torch.empty((B, Tout, F * (num_context + 1)), dtype=torch_dtype(out_dtype), device=x.device)
    len_out = torch.empty_like(len_raw)
    _chk(_C.neurst_splice_downsample(_p(_c(x)), _p(len_raw), _p(out), _p(len_out), B, Traw, F, num_context, rate, out_dtype, _s()),
         'splice_downsample')
    return out, len_out


# ---- a3 ------------------------------------------------------------------------------
def specaug_params(x, F, Tmax, p, key):
    B, T, v = x.shape
    params = torch.empty((B, 8), dtype=torch.int32, device=x.device)
    _chk(_C.neurst_specaug_params(_p(_c(x)), _p(params), B, T, v, F, Tmax, p, key, _s()), 'specaug_params')
    return params


def specaug_apply_(x, params):
    B, T, v = x.shape
    _chk(_C.neurst_specaug_apply(_p(_c(x)), _p(_c(params)), B, T, v, _s()), 'specaug_apply')
    return x


# ---- a5 ------------------------------------------------------------------------------
def add_timing_dropout(x, out_dtype, rate, key):
    B, T, C = x.shape
    y = torch.empty((B, T, C), dtype=torch_dtype(out_dtype), device=x.device)
    sig = timing_signal_table(T, C, x.device)
    _chk(_C.neurst_add_timing_dropout_fwd(_p(_c(x)), dt_of(x), _p(sig), _p(y), out_dtype, B, T, C, rate, key, _s()), 'add_timing_dropout')
    return y


def dropout_bwd(dy, rate, key, out=None):
    dx = torch.empty_like(dy) if out is None else out
    _chk(_C.neurst_dropout_bwd(_p(_c(dy)), _p(dx), dy.numel(), dt_of(dy), rate, key, _s()), 'dropout_bwd')
    return dx


def add_inplace(dst, src):
    assert dst.dtype == src.dtype and dst.numel() == src.numel()
    _chk(_C.neurst_add_inplace(_p(_c(dst)), _p(_c(src)), dst.numel(), dt_of(dst), _s()), 'add_inplace')
    return dst


# ---- GEMM ----------------------------------------------------------------------------
def gemm(A, B, C, M, N, K, lda, ldb, ldc, trans_a, trans_b, bias=None, relu=False, relu_src=None, accumulate=False, split_k=1):
    assert A.dtype == B.dtype
    rc = _C.neurst_gemm_bias_act(A.data_ptr(), B.data_ptr(), C.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                 0 if relu_src is None else relu_src.data_ptr(), M, N, K, lda, ldb, ldc, trans_a, trans_b,
                                 _DT[A.dtype], _DT[C.dtype], relu, accumulate, split_k, _override_raw or _raw_stream(_cur_dev()))
    if rc:
        _lib.check(rc, 'gemm_bias_act')
    return C


THIN_MAX_TILES = max(int(os.environ.get('NEURST_GEMM_THIN_MAX_TILES', '64')),    # csrc/gemm.hip: thin_max_tiles() (32-row tiles)
                     int(os.environ.get('NEURST_GEMM_THIN64_MAX_TILES', '128')))  # and thin64_max_tiles() (64-row tiles)
PERSIST_MIN_TILES = int(os.environ.get('NEURST_GEMM_PERSIST_MIN_TILES', '1024'))   # csrc/gemm.hip: persist_min_tiles()


def gemm_kernel_label(A, C, M, N, K, trans_a, trans_b, split_k=1, bias_grad=False):
    """Which instantiation of gemm_kernel (csrc/gemm.hip) a call runs - mirrors the host dispatch there; used by bench.py
    to group its per-launch timings the way rocprofv3 groups kernels."""
    ab = 'bf16' if A.dtype == torch.bfloat16 else 'f32'
    c = 'bf16' if C.dtype == torch.bfloat16 else 'f32'
    lay = {(False, True): 'NT', (False, False): 'NN', (True, False): 'TN', (True, True): 'TT'}[(bool(trans_a), bool(trans_b))]
    fast = ab == 'bf16' and K % 8 == 0 and (not trans_a or M % 8 == 0) and (trans_b or N % 8 == 0)
    # csrc/gemm256.hip neurst_gemm256_eligible: long reductions and the wide fp32 result, from 128 tiles of 256 x 256 up
    if fast and not trans_a and not bias_grad and -(-M // 256) * -(-N // 256) >= 128 and (
            (K >= 2560 and trans_b) or K >= 8192 or (c == 'f32' and N >= 4096)):
        return 'gemm256_kernel<bf16,%s,%s>' % (c, lay)
    tiles = -(-M // 128) * -(-N // 128) * max(split_k, 1)
    ns = 4 if (fast and tiles <= 256 and K >= 256) else 2
    if ns == 4 and not trans_a and split_k <= 1 and not bias_grad and tiles <= THIN_MAX_TILES:
        return 'gemm_thin_kernel<bf16,%s,%s>' % (c, lay)
    if fast and lay == 'NT' and ns == 2 and split_k <= 1 and not bias_grad and K % 64 == 0 and K >= 192 and tiles >= PERSIST_MIN_TILES:
        return 'gemm_persist_kernel<%s,%s,NT>' % (ab, c)
    return 'gemm_kernel<%s,%s,%s,%s,stages=%d%s>' % (ab, c, lay, 'direct-to-LDS' if fast else 'register-staged', ns, ',+bias-grad' if bias_grad else '')


def gemm_wgrad_bias(A, B, C, opsum, sum_operand, M, N, K, lda, ldb, ldc, split_k=1):
    """C[M,N] += A^T.B (A [K,M], B [K,N]) and opsum += column sums of A (sum_operand 1) or B (2)."""
    rc = _C.neurst_gemm_wgrad_bias(A.data_ptr(), B.data_ptr(), C.data_ptr(), opsum.data_ptr(), sum_operand, M, N, K, lda, ldb, ldc,
                                   _DT[A.dtype], split_k, _override_raw or _raw_stream(_cur_dev()))
    if rc:
        _lib.check(rc, 'gemm_wgrad_bias')


def gemm_wgrad_grouped(problems, tile_ranges=None):
    """[(A [K, M] = dY, B [K, N] = X, C [M, N] fp32, opsum [M] fp32 or None), ...] (at most wgrad_group_max()): C += A^T . B
    and opsum += column sums of A for every problem in one launch (neurst_gemm_wgrad_bias_grouped_tiles).  tile_ranges
    [(begin, end), ...]: only those 256 x 192 output tiles of each problem (wgrad_group_tiles(M, N) per problem)."""
    n = len(problems)
    P = ctypes.c_void_p * n
    L = ctypes.c_int64 * n
    A = P(*[a.data_ptr() for a, _, _, _ in problems])
    B = P(*[b.data_ptr() for _, b, _, _ in problems])
    C = P(*[c.data_ptr() for _, _, c, _ in problems])
    S = P(*[(o.data_ptr() if o is not None else None) for _, _, _, o in problems])
    M = L(*[a.shape[1] for a, _, _, _ in problems])
    N = L(*[b.shape[1] for _, b, _, _ in problems])
    K = L(*[a.shape[0] for a, _, _, _ in problems])
    T0 = T1 = None
    if tile_ranges is not None:
        I = ctypes.c_int * n
        T0, T1 = I(*[r[0] for r in tile_ranges]), I(*[r[1] for r in tile_ranges])
    rc = _C.neurst_gemm_wgrad_bias_grouped_tiles(n, A, B, C, S, M, N, K, M, N, N, T0, T1, _DT[problems[0][0].dtype],
                                                 _override_raw or _raw_stream(_cur_dev()))
    if rc:
        _lib.check(rc, 'gemm_wgrad_bias_grouped_tiles')


def wgrad_group_tiles(M, N):
    return int(_C.neurst_gemm_wgrad_group_tiles(M, N))


def wgrad_group_max():
    return int(_C.neurst_gemm_wgrad_group_max())


def colsum(x2d, out, accumulate=True):
    M, N = x2d.shape
    _chk(_C.neurst_colsum(_p(x2d), _p(out), M, N, x2d.stride(0), dt_of(x2d), int(accumulate), _s()), 'colsum')
    return out


def pick_split_k(M, N, K, n_cu=256):
    """wgrad-style GEMMs have few 128x128 output tiles and a long reduction.  Measured on MI355X
    (tools/sweep_splitk.py, K = 16000): the best split puts ~430 workgroups in flight, is never worth more
    than 6 (fp32 atomic traffic grows with it) and never pays below K = 2048."""
    tiles = -(-M // 128) * -(-N // 128)
    if K < 2048:
        return 1
    return int(max(1, min(6, round(430.0 / tiles), K // 1024)))


# ---- attention -------------------------------------------------------------------------
def mha_keep_bits(B, H, Tq, Tk, rate, key, device):
    """the attention-dropout decisions of one layer from (key, shape) alone, for mha_fwd(..., keep_bits=<this tensor>)"""
    kb = torch.empty(_C.neurst_mha_keep_bits_bytes(B, H, Tq, Tk) // 2, dtype=torch.int16, device=device)
    _chk(_C.neurst_mha_keep_bits(_p(kb), B, H, Tq, Tk, rate, key, _s()), 'mha_keep_bits')
    return kb


def mha_fwd(q, k, v, ldq, ldk, ldv, B, H, Tq, Tk, dh, mask_mode, len_k, rate, key, keep_bits=False):
    """keep_bits=True (training with dropout): also returns the stored dropout decisions for mha_bwd; keep_bits=<tensor of
    mha_keep_bits>: the decisions were generated ahead, the forward loads them."""
    o = torch.empty((B, Tq, H * dh), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
    if isinstance(keep_bits, torch.Tensor):
        _chk(_C.neurst_mha_fwd_pregen(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(keep_bits), ldq, ldk, ldv, H * dh, B, H, Tq, Tk, dh, mask_mode,
                                       _p(len_k), rate, key, dt_of(q), _s()), 'mha_fwd_pregen')
        return o, lse, keep_bits
    if keep_bits:
        kb = torch.empty(_C.neurst_mha_keep_bits_bytes(B, H, Tq, Tk) // 2, dtype=torch.int16, device=q.device) if rate > 0 else None
        _chk(_C.neurst_mha_fwd_train(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(kb), ldq, ldk, ldv, H * dh, B, H, Tq, Tk, dh, mask_mode,
                                      _p(len_k), rate, key, dt_of(q), _s()), 'mha_fwd_train')
        return o, lse, kb
    _chk(_C.neurst_mha_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), ldq, ldk, ldv, H * dh, B, H, Tq, Tk, dh, mask_mode, _p(len_k), rate, key,
                            dt_of(q), _s()), 'mha_fwd')
    return o, lse


def mha_bwd(q, k, v, o, d_o, lse, dq, dk, dv, ldq, ldk, ldv, lddq, lddk, lddv, B, H, Tq, Tk, dh, mask_mode, len_k, rate, key, keep_bits=None):
    delta = torch.empty((B, H, Tq), dtype=torch.float32, device=q.device)
    # fp32 partial dQ of the fused kernel (encoder-size bf16 problems with more than one block of 256 keys); the caching
    # allocator hands the same block back every layer
    wsb = _C.neurst_mha_bwd_workspace_bytes(B, H, Tq, Tk, dh) if q.dtype == torch.bfloat16 else 0
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=q.device) if wsb else None
    _chk(_C.neurst_mha_bwd_train_ws(_p(q), _p(k), _p(v), _p(o), _p(_c(d_o)), _p(lse), _p(keep_bits), _p(delta), _p(dq), _p(dk), _p(dv), ldq, ldk,
                                     ldv, H * dh, lddq, lddk, lddv, B, H, Tq, Tk, dh, mask_mode, _p(len_k), rate, key, dt_of(q),
                                     _p(ws), wsb, _s()), 'mha_bwd')


# ---- layer norm ------------------------------------------------------------------------
def dropout_add_layernorm_fwd(x, y_inout, gamma, beta, rate, key, lens=None, T=0, bufs=None):
    """bufs = (out, mean, rstd) to write into buffers of the caller (decoding loops) instead of new ones."""
    C = x.shape[-1]
    rows = x.numel() // C
    if bufs is not None:
        out, mean, rstd = bufs
    else:
        out = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _chk(_C.neurst_dropout_add_layernorm_fwd(_p(_c(x)), _p(_c(y_inout)), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), rows, C, rate, key,
                                              _p(lens), T, dt_of(x), _s()), 'dropout_add_layernorm_fwd')
    return out, mean, rstd


def dropout_add_layernorm_bwd(dout, z, gamma, mean, rstd, dgamma, dbeta, rate, key, lens=None, T=0, deterministic=False, dbias=None):
    """deterministic=True: column sums through per-workgroup partial buffers and a second reduction launch instead of
    fp32 atomics straight into dgamma / dbeta.  dbias (fp32 [C]): += column sums of dy."""
    C = z.shape[-1]
    rows = z.numel() // C
    dz = torch.empty_like(z)
    dy = torch.empty_like(z) if rate > 0 else dz
    partial = None
    if deterministic:
        nblk = _C.neurst_layernorm_bwd_blocks(rows)
        partial = torch.empty(2 * nblk * C, dtype=torch.float32, device=z.device)
    _chk(_C.neurst_dropout_add_layernorm_bwd(_p(_c(dout)), _p(z), _p(gamma), _p(mean), _p(rstd), _p(dz), _p(dy), _p(dgamma), _p(dbeta),
                                              _p(partial), _p(dbias), rows, C, rate, key, _p(lens), T, dt_of(z), _s()), 'dropout_add_layernorm_bwd')
    return dz, dy


def mask_rows_(x, lens):
    B, T, C = x.shape
    _chk(_C.neurst_mask_rows(_p(_c(x)), _p(lens), B, T, C, dt_of(x), _s()), 'mask_rows')
    return x


# ---- decoder inputs / embedding -----------------------------------------------------------
def build_decoder_input(labels, len_labels, sos):
    B, Ln = labels.shape
    inp = torch.empty_like(labels)
    _chk(_C.neurst_build_decoder_input(_p(_c(labels)), _p(len_labels), _p(inp), B, Ln, sos, _s()), 'build_decoder_input')
    return inp


def embedding_fwd(ids, table, out_dtype, rate, key, pos0=0, signal_len=None, out=None):
    B, Ln = ids.shape
    V, D = table.shape
    sig = timing_signal_table(signal_len or (pos0 + Ln), D, ids.device)
    if out is None:
        out = torch.empty((B, Ln, D), dtype=torch_dtype(out_dtype), device=ids.device)
    sig_ptr = sig.data_ptr() + pos0 * D * 4
    _chk(_C.neurst_embedding_fwd(_p(_c(ids)), _p(table), sig_ptr, _p(out), B, Ln, D, V, rate, key, out_dtype, _s()), 'embedding_fwd')
    return out


def embedding_bwd(ids, dout, dtable, rate, key):
    B, Ln = ids.shape
    V, D = dtable.shape
    _chk(_C.neurst_embedding_bwd(_p(ids), _p(_c(dout)), _p(dtable), B, Ln, D, V, rate, key, dt_of(dout), _s()), 'embedding_bwd')


# ---- CTC / greedy / shrink -------------------------------------------------------------------
def ctc_loss_fwd_bwd(logits, len_logits, labels, len_labels, grad_dtype, grad_scale, in_place=False, recursion_stream=None):
    """recursion_stream: run the alpha-beta recursion and the gradient correction there (after the per-frame part on the
    current stream): arg-max is then available on the current stream right away, loss / grad once that stream is joined."""
    B, T, V = logits.shape
    Umax = labels.shape[1]
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = logits if (in_place and grad_dtype == F32) else torch.empty((B, T, V), dtype=torch_dtype(grad_dtype), device=logits.device)
    argmax = torch.empty((B, T), dtype=torch.int32, device=logits.device)
    ws = torch.empty(_C.neurst_ctc_workspace_bytes(B, T, Umax), dtype=torch.uint8, device=logits.device)
    labels = _c(labels)
    args = (_p(_c(logits)), _p(len_logits), _p(labels), _p(len_labels), _p(loss), _p(grad), grad_dtype, _p(argmax), _p(ws), B, T, V, Umax,
            grad_scale)
    if recursion_stream is None:
        _chk(_C.neurst_ctc_loss_fwd_bwd_phased(*args, 3, _s()), 'ctc_loss_fwd_bwd')
        return loss, grad, argmax
    _chk(_C.neurst_ctc_loss_fwd_bwd_phased(*args, 1, _s()), 'ctc_loss_fwd_bwd(frames)')
    recursion_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(recursion_stream):
        _chk(_C.neurst_ctc_loss_fwd_bwd_phased(*args, 2, _s()), 'ctc_loss_fwd_bwd(recursion)')
    for t in (ws, loss, grad, labels, len_logits, len_labels):   # allocated on the current stream, still in use on the other one
        t.record_stream(recursion_stream)
    return loss, grad, argmax


def argmax_rows(logits2d):
    rows, V = logits2d.shape
    out = torch.empty(rows, dtype=torch.int32, device=logits2d.device)
    _chk(_C.neurst_argmax_rows(_p(_c(logits2d)), _p(out), rows, V, _s()), 'argmax_rows')
    return out


def ctc_greedy(argmax, len_logits, blank):
    B, T = argmax.shape
    out = torch.empty((B, T), dtype=torch.int32, device=argmax.device)
    out_len = torch.empty(B, dtype=torch.int32, device=argmax.device)
    _chk(_C.neurst_ctc_greedy(_p(_c(argmax)), _p(len_logits), _p(out), _p(out_len), B, T, blank, _s()), 'ctc_greedy')
    return out, out_len


def shrink_index(argmax, lens, blank):
    B, T = argmax.shape
    index = torch.empty((B, T), dtype=torch.int32, device=argmax.device)
    new_len = torch.empty(B, dtype=torch.int32, device=argmax.device)
    _chk(_C.neurst_shrink_index(_p(_c(argmax)), _p(lens), _p(index), _p(new_len), B, T, blank, _s()), 'shrink_index')
    return index, new_len


def shrink_clamp_(new_len, bound, state):
    """new_len[b] = min(new_len[b], bound) in place; state[0] = overflow flag, state[1] = max before clamping (int32, device)"""
    _chk(_C.neurst_shrink_clamp(_p(new_len), new_len.numel(), int(bound), _p(state), _s()), 'shrink_clamp')


def shrink_fwd(x, index, Tout):
    B, T, D = x.shape
    out = torch.empty((B, Tout, D), dtype=x.dtype, device=x.device)
    _chk(_C.neurst_shrink_fwd(_p(_c(x)), _p(index), _p(out), B, T, Tout, D, dt_of(x), _s()), 'shrink_fwd')
    return out


def shrink_bwd(dout, index, T):
    B, Tout, D = dout.shape
    dx = torch.empty((B, T, D), dtype=dout.dtype, device=dout.device)
    _chk(_C.neurst_shrink_bwd(_p(_c(dout)), _p(index), _p(dx), B, T, Tout, D, dt_of(dout), _s()), 'shrink_bwd')
    return dx


# ---- losses --------------------------------------------------------------------------------------
def ls_xent_fwd_bwd(logits, labels, len_labels, confidence, grad_dtype, grad_scale, in_place=False):
    B, Ln, V = logits.shape
    xent = torch.empty((B, Ln), dtype=torch.float32, device=logits.device)
    grad = logits if (in_place and grad_dtype == F32) else torch.empty((B, Ln, V), dtype=torch_dtype(grad_dtype), device=logits.device)
    _chk(_C.neurst_ls_xent_fwd_bwd(_p(_c(logits)), _p(_c(labels)), _p(len_labels), _p(xent), _p(grad), grad_dtype, B, Ln, V, confidence,
                                    grad_scale, _s()), 'ls_xent_fwd_bwd')
    return xent, grad


def mse_fwd_bwd(pred, label, grad_scale):
    n = pred.numel()
    nblk = _C.neurst_mse_blocks(n)
    partial = torch.empty(nblk, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred)
    _chk(_C.neurst_mse_fwd_bwd(_p(_c(pred)), _p(_c(label)), dt_of(label), _p(partial), _p(dpred), n, grad_scale, dt_of(pred), _s()), 'mse_fwd_bwd')
    return partial, dpred


def loss_combine(xent, ctc, mse_partial, mse_numel, B, Ln, w_ast, w_ctc, w_kd, device):
    out = torch.empty(4, dtype=torch.float32, device=device)
    _chk(_C.neurst_loss_combine(_p(xent), _p(ctc), _p(mse_partial), 0 if mse_partial is None else mse_partial.numel(), mse_numel, B, Ln,
                                 w_ast, w_ctc, w_kd, _p(out), _s()), 'loss_combine')
    return out


# ---- conv front end ----------------------------------------------------------------------------------
def same_out(n, s):
    return -(-n // s)


def conv2d_ln_relu_fwd(x, w, bias, gamma, beta, kernel, stride):
    B, H, W, Cin = x.shape
    Cout = w.shape[-1]
    Ho, Wo = same_out(H, stride[0]), same_out(W, stride[1])
    conv_out = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    out = torch.empty_like(conv_out)
    mean = torch.empty(B * Ho * Wo, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    _chk(_C.neurst_conv2d_ln_relu_fwd(_p(_c(x)), _p(w), _p(bias), _p(gamma), _p(beta), _p(conv_out), _p(out), _p(mean), _p(rstd), B, H, W, Cin,
                                       Cout, kernel[0], kernel[1], stride[0], stride[1], dt_of(x), _s()), 'conv2d_ln_relu_fwd')
    return out, conv_out, mean, rstd


def conv2d_ln_relu_bwd(dout, x, w, gamma, conv_out, out, mean, rstd, dw, dbias, dgamma, dbeta, kernel, stride, need_dx):
    B, H, W, Cin = x.shape
    Cout = w.shape[-1]
    ws = torch.empty_like(conv_out)
    dx = torch.empty_like(x) if need_dx else None
    nb = int(_C.neurst_conv2d_bwd_workspace_bytes(B, H, W, Cin, Cout, kernel[0], kernel[1], stride[0], stride[1], dt_of(x)))
    col = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb > 0 else None      # im2col image for the weight-gradient GEMM
    _chk(_C.neurst_conv2d_ln_relu_bwd(_p(_c(dout)), _p(x), _p(w), _p(gamma), _p(conv_out), _p(out), _p(mean), _p(rstd), _p(ws), _p(col), nb,
                                       _p(dx), _p(dw), _p(dbias), _p(dgamma), _p(dbeta), B, H, W, Cin, Cout, kernel[0], kernel[1], stride[0],
                                       stride[1], dt_of(x), _s()), 'conv2d_ln_relu_bwd')
    return dx


def maxpool3x3_fwd(x):
    B, H, W, C = x.shape
    out = torch.empty_like(x)
    arg = torch.empty((B, H, W, C), dtype=torch.int8, device=x.device)
    _chk(_C.neurst_maxpool3x3_fwd(_p(_c(x)), _p(out), _p(arg), B, H, W, C, dt_of(x), _s()), 'maxpool3x3_fwd')
    return out, arg


def maxpool3x3_bwd(dout, arg):
    B, H, W, C = dout.shape
    dx = torch.empty_like(dout)
    _chk(_C.neurst_maxpool3x3_bwd(_p(_c(dout)), _p(arg), _p(dx), B, H, W, C, dt_of(dout), _s()), 'maxpool3x3_bwd')
    return dx


# ---- optimiser / casts / greedy step ---------------------------------------------------------------------
def adam_step(p, g, m, v, shadow, lr_t, beta1, beta2, eps, grad_scale, skip_flag=None):
    """skip_flag: int32 device tensor; the update is a no-op when its first element is non-zero at execution time"""
    if skip_flag is None:
        _chk(_C.neurst_adam_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), lr_t, beta1, beta2, eps, grad_scale, _s()), 'adam_step')
    else:
        _chk(_C.neurst_adam_step_guarded(_p(p), _p(g), _p(m), _p(v), _p(shadow), p.numel(), lr_t, beta1, beta2, eps, grad_scale, _p(skip_flag), _s()),
             'adam_step_guarded')


def axpy_f32(y, x, alpha):
    """y += alpha * x (fp32, in place)."""
    _chk(_C.neurst_axpy_f32(_p(y), _p(x), float(alpha), y.numel(), _s()), 'axpy_f32')


def sumsq_f32(x):
    """Sum of squares of an fp32 tensor: fp64 block partials on the device, summed on the host (one small read back)."""
    nb = int(_C.neurst_sumsq_blocks(x.numel()))
    part = torch.empty(nb, dtype=torch.float64, device=x.device)
    _chk(_C.neurst_sumsq_f32(_p(x), x.numel(), _p(part), _s()), 'sumsq_f32')
    return float(part.cpu().numpy().sum())


def cast_f32_to_bf16(src, dst):
    _chk(_C.neurst_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _s()), 'cast_f32_to_bf16')


def cast_bf16_to_f32(src, dst):
    _chk(_C.neurst_cast_bf16_to_f32(_p(src), _p(dst), src.numel(), _s()), 'cast_bf16_to_f32')


def greedy_step(logits2d, preds, finished, len_decoded, n_unfinished, step, eos):
    B, V = logits2d.shape
    _chk(_C.neurst_greedy_step(_p(_c(logits2d)), _p(preds), _p(finished), _p(len_decoded), _p(n_unfinished), B, V, step, preds.shape[1], eos, _s()),
         'greedy_step')


def selftest_mma(A, Bt):
    K = A.shape[1]
    C = torch.empty((16, 16), dtype=torch.float32, device=A.device)
    _chk(_C.neurst_selftest_mma(_p(_c(A)), _p(_c(Bt)), _p(C), K, dt_of(A), _s()), 'selftest_mma')
    return C


def lens_ceil_div(lens, div):
    out = torch.empty_like(lens)
    _chk(_C.neurst_lens_ceil_div(_p(lens), _p(out), lens.numel(), div, _s()), 'lens_ceil_div')
    return out


def mha_fwd_strided(q, k, v, o, lse, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, Tq, Tk, dh, mask_mode, len_k):
    _chk(_C.neurst_mha_fwd_strided(_p(q), _p(k), _p(v), _p(o), _p(lse), ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, Tq, Tk, dh, mask_mode,
                                    _p(len_k), 0.0, 0, dt_of(q), _s()), 'mha_fwd_strided')


# ---- beam search bookkeeping (transformer_decoder.py:248-431) ---------------------------------------------------
def beam_topk(logits2d, k):
    R, V = logits2d.shape
    scores = torch.empty((R, k), dtype=torch.float32, device=logits2d.device)
    ids = torch.empty((R, k), dtype=torch.int32, device=logits2d.device)
    _chk(_C.neurst_beam_topk(_p(_c(logits2d)), _p(scores), _p(ids), R, V, k, _s()), 'beam_topk')
    return scores, ids


def beam_prune(next_scores, next_ids, scores, len_decoded, finished, n_unfinished, beam, eos, length_penalty_weight,
               prev_unfinished=None):
    R = scores.numel()
    new_scores, new_len, new_fin = torch.empty_like(scores), torch.empty_like(len_decoded), torch.empty_like(finished)
    tokens = torch.empty(R, dtype=torch.int32, device=scores.device)
    parents = torch.empty(R, dtype=torch.int32, device=scores.device)
    _chk(_C.neurst_beam_prune(_p(_c(next_scores)), _p(_c(next_ids)), _p(scores), _p(len_decoded), _p(finished), _p(new_scores), _p(new_len),
                              _p(new_fin), _p(tokens), _p(parents), _p(n_unfinished), _p(prev_unfinished) if prev_unfinished is not None else None,
                              R // beam, beam, eos, float(length_penalty_weight),
                              _s()), 'beam_prune')
    return new_scores, new_len, new_fin, tokens, parents


def gather_rows(src, idx, out=None, row_bytes=None):
    """out[r] = src[idx[r]] along dim 0 (src contiguous); row_bytes: only that many leading bytes of every row."""
    src = _c(src)
    out = torch.empty_like(src) if out is None else out
    stride = src[0].numel() * src.element_size()
    _chk(_C.neurst_gather_rows(_p(src), _p(out), _p(idx), idx.numel(), stride if row_bytes is None else row_bytes, stride, stride, _s()),
         'gather_rows')
    return out
