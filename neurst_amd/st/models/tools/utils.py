"""Layer primitives with the reference's names and argument meaning
(st/models/tools/utils.py: dense :367-398, ff_hidden :401-410, residual :324-337,
multihead_attention :479-554).  Each call launches the forward HIP kernel(s) and, when a
Tape is active, registers the matching backward launches."""
import os

import torch

from .. import graph as G
from ...layers import common_layers_v1
from .... import ops


def _flat2d(x):
    return x.reshape(-1, x.shape[-1])


def _wgrad(dy2, x2, param, layout, bias=None):
    """param.grad += dy^T . x ('out_in' kernels) or x^T . dy ('in_out' kernels); bias.grad += column sums of dy
    from the same launch."""
    M = x2.shape[0]
    n_out, n_in = dy2.shape[1], x2.shape[1]
    if layout == 'out_in':
        sk = ops.pick_split_k(n_out, n_in, M)
        if bias is not None:
            ops.gemm_wgrad_bias(dy2, x2, param.grad, bias.grad, 1, n_out, n_in, M, n_out, n_in, n_in, split_k=sk)
        else:
            ops.gemm(dy2, x2, param.grad, n_out, n_in, M, n_out, n_in, n_in, True, False, accumulate=True, split_k=sk)
    else:
        sk = ops.pick_split_k(n_in, n_out, M)
        if bias is not None:
            ops.gemm_wgrad_bias(x2, dy2, param.grad, bias.grad, 2, n_in, n_out, M, n_in, n_out, n_out, split_k=sk)
        else:
            ops.gemm(x2, dy2, param.grad, n_in, n_out, M, n_in, n_out, n_out, True, False, accumulate=True, split_k=sk)


# off by default: measured on LUT config 2 the third column sum makes ln_bwd (main stream) slower by more than the operand
# sum costs the weight-gradient GEMMs (second stream): 16.49 ms per step with it, 16.32 without
_FUSE_BIAS_GRAD = os.environ.get('NEURST_FUSE_BIAS_GRAD', '0') != '0'
# attention-dropout decisions hashed ahead on the second stream: off by default - measured 16.43 against 16.32 ms per step (the extra
# launch, its two stream hand-offs and the share of the chip it takes from the projection GEMM cost more than the hash saves)
_PREGEN_KEEP_BITS = os.environ.get('NEURST_PREGEN_KEEP_BITS', '0') != '0'
_PREGEN_MIN_WEIGHTS = 1 << 22                                                  # not for decoder-size attention: one more launch costs more than it saves


class BiasGradHook:
    """Set by a dense layer on its output; `residual` marks it taken when it accumulates the bias gradient itself."""
    __slots__ = ('bias', 'taken')

    def __init__(self, bias):
        self.bias, self.taken = bias, False


def _linear(inputs, w, b, units, relu, layout, out_dtype, stop_gradient, early_backward=False):
    g = G.current()
    n_in = inputs.shape[-1]
    x2 = _flat2d(inputs)
    M = x2.shape[0]
    out = torch.empty(inputs.shape[:-1] + (units,), dtype=ops.torch_dtype(g.dtype if out_dtype is None else out_dtype), device=inputs.device)
    bias = b.master if b is not None else None
    if layout == 'out_in':      # x . W^T, kernel [units, in]
        ops.gemm(x2, w.w, out, M, units, n_in, n_in, n_in, units, False, True, bias=bias, relu=relu)
    else:                       # x . W, kernel [in, units]
        ops.gemm(x2, w.w, out, M, units, n_in, n_in, units, units, False, False, bias=bias, relu=relu)
    tape = g.tape
    if tape is not None:
        if relu:
            tape.relu_outputs.add(id(out))
        # a `residual` that consumes `out` directly sums dy over the rows anyway (LayerNorm's dgamma / dbeta pass): it
        # takes the bias gradient along (fused_bias_grad) and the weight-gradient GEMM below runs without its operand sum
        hook = None
        if b is not None and not relu and _FUSE_BIAS_GRAD:
            hook = tape.bias_hooks[id(out)] = BiasGradHook(b)

        def bwd():
            dy = tape.pop(out)
            if dy is None:
                return
            dy2 = _flat2d(dy)
            if dy2.dtype != x2.dtype:
                raise TypeError('gradient dtype {} does not match operand dtype {}'.format(dy2.dtype, x2.dtype))
            bw = None if (hook is not None and hook.taken) else b
            if bw is b and g.defer_wgrad(dy2, x2, w, layout, b):
                pass                 # encoder-size layer: goes out with its group (graph.flush_wgrads reports the parameters)
            else:
                side = None if g.in_early else g.side_stream_for(M, dy2.device)   # an early node already runs off the main stream
                if side is not None:     # small layer: weight (and bias) gradient beside the data-gradient chain
                    g.run_on_side(side, lambda: _wgrad(dy2, x2, w, layout, bw), dy2, x2)
                else:
                    _wgrad(dy2, x2, w, layout, bw)
                g.param_grad_done(w)
                if b is not None:
                    g.param_grad_done(b)
            if not stop_gradient:
                have = tape.has(inputs)
                dx = tape.grad(inputs) if have else torch.empty_like(inputs)
                if have:
                    g.before_inplace_write(dx)
                gate = inputs if id(inputs) in tape.relu_outputs else None
                if layout == 'out_in':   # dX = dY . W
                    ops.gemm(dy2, w.w, dx, M, n_in, units, units, n_in, n_in, False, False, relu_src=gate, accumulate=have)
                else:                    # dX = dY . W^T
                    ops.gemm(dy2, w.w, dx, M, n_in, units, units, units, n_in, False, True, relu_src=gate, accumulate=have)
                if not have:
                    tape.set(inputs, dx)
                if g.in_early:
                    g.side_written(dx)
        tape.push(bwd, early=early_backward)
    return out


def dense(inputs, units, activation=None, use_bias=True, kernel=None, reuse=None, name=None, out_dtype=None, stop_gradient=False):
    """x . kernel^T (+ bias), kernel [units, in] (tools/utils.py:388-393). activation: None or 'relu'."""
    g = G.current()
    with (g.variable_scope(name) if name else g.default_scope('dense')):
        w = g.get_variable('kernel', [units, inputs.shape[-1]])
        b = g.get_variable('bias', [units]) if use_bias else None
    return _linear(inputs, w, b, units, activation == 'relu', 'out_in', out_dtype, stop_gradient)


def tf_layers_dense(inputs, units, use_bias=False, name=None, out_dtype=None, stop_gradient=False, early_backward=False):
    """tf.layers.dense(use_bias=False): x . kernel, kernel [in, units]
    (transformer_encoder.py:117-122, fc_decoder.py:33-38, transformer_decoder.py:42-46)."""
    assert not use_bias
    g = G.current()
    with g.variable_scope(name):
        w = g.get_variable('kernel', [inputs.shape[-1], units])
    return _linear(inputs, w, None, units, False, 'in_out', out_dtype, stop_gradient, early_backward)


def ff_hidden(inputs, hidden_size, output_size, activation='relu', use_bias=True, reuse=None, name=None):
    """relu(x W1^T + b1) W2^T + b2 (tools/utils.py:401-410)."""
    g = G.current()
    with (g.variable_scope(name) if name else g.default_scope('ff_hidden')):
        hidden_outputs = dense(inputs, hidden_size, activation, use_bias)
        outputs = dense(hidden_outputs, output_size, None, use_bias)
    return outputs


def residual(inputs, outputs, dropout_rate, site=None, mask_lens=None):
    """LN(inputs + dropout(outputs)) (tools/utils.py:324-337).  `outputs` is overwritten with the
    pre-norm sum (kept for backward).  mask_lens zeroes rows t >= len (transformer_encoder.py:151)."""
    g = G.current()
    C = inputs.shape[-1]
    scale, bias = common_layers_v1.layer_norm_vars(C)
    rate, key = g.rate(dropout_rate), g.key(site)
    T = inputs.shape[-2]
    out, mean, rstd = ops.dropout_add_layernorm_fwd(inputs, outputs, scale.master, bias.master, rate, key, mask_lens, T)
    tape = g.tape
    if tape is not None:
        hook = tape.bias_hooks.get(id(outputs))
        if hook is not None:
            hook.taken = True

        def bwd():
            dout = tape.pop(out)
            if dout is None:
                return
            dz, dy = ops.dropout_add_layernorm_bwd(dout, outputs, scale.master, mean, rstd, scale.grad, bias.grad, rate, key, mask_lens, T,
                                                   dbias=hook.bias.grad if hook is not None else None)
            g.param_grad_done(scale)
            g.param_grad_done(bias)
            # With rate == 0 dy IS dz (one buffer): the producer of `outputs` reads it in the very next
            # backward nodes, and only later does another consumer of `inputs` accumulate into it (stream
            # order), so sharing the buffer between the two branches is safe.
            tape.set(outputs, dy)
            tape.accumulate(inputs, dz)
        tape.push(bwd)
    return out


def multihead_attention(query_antecedent, memory_antecedent, bias, total_key_depth, total_value_depth, output_depth, num_heads,
                        dropout_rate, num_queries=None, query_eq_key=False, summaries=False, image_shapes=None, name=None, site=None):
    """tools/utils.py:479-554.  `bias` is an AttentionBias descriptor or None.  Self-attention runs one
    d->3d projection, cross-attention d->d on the queries and d->2d on the memory; q is scaled by
    dh^-0.5, softmax(q k^T + bias), dropout on the weights, . v, heads merged, d->d projection."""
    assert not query_eq_key and num_queries is None, 'only the training forms are routed through this function'
    assert total_key_depth == total_value_depth
    g = G.current()
    d, H = total_key_depth, num_heads
    dh = d // H
    B, Tq = query_antecedent.shape[0], query_antecedent.shape[1]
    mode = bias.mode if bias is not None else ops.MASK_NONE
    lens = bias.lens if bias is not None else None
    rate, key = g.rate(dropout_rate), g.key(site)
    tape = g.tape
    Tk_ = Tq if memory_antecedent is None else memory_antecedent.shape[1]
    ahead = None
    if (tape is not None and rate > 0 and _PREGEN_KEEP_BITS and g.side_streams and query_antecedent.is_cuda
            and B * H * Tq * Tk_ >= _PREGEN_MIN_WEIGHTS):
        # the dropout decisions depend on (key, shape) only: hash them on the second stream while the projection GEMM(s)
        # below keep the matrix pipes busy; the forward then loads them like the backward does
        ahead = g.run_ahead(lambda: ops.mha_keep_bits(B, H, Tq, Tk_, rate, key, query_antecedent.device), query_antecedent.device)
    with (g.variable_scope(name) if name else g.default_scope('multihead_attention')):
        if memory_antecedent is None:
            combined = dense(query_antecedent, 3 * d, name='qkv_transform')
            Tk = Tq
            q = (combined, 0, 3 * d)
            k = (combined, d, 3 * d)
            v = (combined, 2 * d, 3 * d)
            qbuf = kvbuf = combined
        else:
            qbuf = dense(query_antecedent, d, name='q_transform')
            kvbuf = dense(memory_antecedent, 2 * d, name='kv_transform')
            Tk = memory_antecedent.shape[1]
            q = (qbuf, 0, d)
            k = (kvbuf, 0, 2 * d)
            v = (kvbuf, d, 2 * d)
        view = lambda t: t[0].view(-1)[t[1]:]       # pointer to the first element of the (strided) operand
        if tape is not None:   # training: keep the dropout decisions for backward (the TF graph keeps the mask tensor)
            pre = True
            if ahead is not None:
                pre, ready = ahead
                torch.cuda.current_stream().wait_event(ready)
            x, lse, kb = ops.mha_fwd(view(q), view(k), view(v), q[2], k[2], v[2], B, H, Tq, Tk, dh, mode, lens, rate, key, keep_bits=pre)
        else:
            x, lse = ops.mha_fwd(view(q), view(k), view(v), q[2], k[2], v[2], B, H, Tq, Tk, dh, mode, lens, rate, key)
            kb = None
        if tape is not None:
            def bwd():
                dx = tape.pop(x)
                if dx is None:
                    return
                if memory_antecedent is None:
                    dcomb = torch.empty_like(combined)
                    dq, dk, dv = dcomb.view(-1), dcomb.view(-1)[d:], dcomb.view(-1)[2 * d:]
                    ops.mha_bwd(view(q), view(k), view(v), x, dx, lse, dq, dk, dv, q[2], k[2], v[2], 3 * d, 3 * d, 3 * d,
                                B, H, Tq, Tk, dh, mode, lens, rate, key, keep_bits=kb)
                    tape.set(combined, dcomb)
                else:
                    dqb, dkvb = torch.empty_like(qbuf), torch.empty_like(kvbuf)
                    ops.mha_bwd(view(q), view(k), view(v), x, dx, lse, dqb.view(-1), dkvb.view(-1), dkvb.view(-1)[d:], q[2], k[2], v[2],
                                d, 2 * d, 2 * d, B, H, Tq, Tk, dh, mode, lens, rate, key, keep_bits=kb)
                    tape.set(qbuf, dqb)
                    tape.set(kvbuf, dkvb)
            tape.push(bwd)
        out = dense(x, output_depth, name='output_transform')
    return out
