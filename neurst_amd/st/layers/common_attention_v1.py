"""Attention helpers of st/layers/common_attention_v1.py used on the hot path.
Biases are descriptors, not [B,1,1,T] / [1,1,L,L] tensors: the fused attention kernel applies
the same additive -1e9 (lines 190, 203) from lengths / indices, so nothing T x T is built."""
from ..models import graph as G
from ... import ops


class AttentionBias:
    def __init__(self, mode, lens=None):
        self.mode, self.lens = mode, lens


def attention_bias_lower_triangle(length):
    """common_attention_v1.py:180-191."""
    return AttentionBias(ops.MASK_CAUSAL)


def attention_bias_ignore_padding(len_memory):
    """common_attention_v1.py:194-204; takes the lengths the padding mask was built from."""
    return AttentionBias(ops.MASK_KEYPAD, len_memory)


def add_timing_signal_1d(x, dropout_rate=0.0, site=None, out_dtype=None, stop_gradient=False):
    """x + sinusoid table (common_attention_v1.py:28-70) fused with the tf.layers.dropout that
    always follows it on the hot path (transformer_encoder.py:110-115)."""
    g = G.current()
    rate = g.rate(dropout_rate)
    key = g.key(site)
    y = ops.add_timing_dropout(x, g.dtype if out_dtype is None else out_dtype, rate, key)
    tape = g.tape
    if tape is not None and not stop_gradient:
        def bwd():
            dy = tape.pop(y)
            if dy is None:
                return
            if rate > 0:
                g.before_inplace_write(dy)
                dy = ops.dropout_bwd(dy, rate, key, out=dy)
            tape.accumulate(x, dy)
        tape.push(bwd)
    return y
