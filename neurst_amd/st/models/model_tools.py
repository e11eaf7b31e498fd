"""Losses / LR schedule of st/models/model_tools.py."""
import numpy as np

from . import graph as G
from ... import ops


def smoothing_cross_entropy(logits, labels, vocab_size, confidence, len_labels, grad_scale):
    """model_tools.py:17-38 fused with the length mask and per-utterance normalisation of
    seq2seqModel.py:169-189.  logits [B,L,V] fp32.  Returns xent [B,L] (masked, already divided by the
    utterance length, so loss_b = sum_l xent[b,l]) and registers d(loss)/d(logits) on the tape.
    The dense [B,L,V] soft-target tensor of the reference is never built."""
    g = G.current()
    assert logits.shape[-1] == vocab_size
    in_place = g.dtype == ops.F32
    xent, grad = ops.ls_xent_fwd_bwd(logits, labels, len_labels, confidence, g.dtype, grad_scale, in_place=in_place)
    if g.tape is not None:
        g.tape.set(logits, grad)
    return xent


def warmup_exponential_decay(global_step, warmup_steps, peak, decay_rate, decay_steps):
    """model_tools.py:80-87, evaluated in fp32 like the TF graph."""
    gs, ws = np.float32(global_step), np.float32(warmup_steps)
    if gs <= ws:
        return float(np.float32(peak) * gs / ws)
    return float(np.float32(peak) * np.float32(decay_rate) ** ((gs - ws) / np.float32(decay_steps)))
