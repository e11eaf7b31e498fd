"""shrink_layer of st/models/tools/blocks.py:8-38 (COSTT): keep frame t iff its CTC arg-max is not
blank, differs from the previous frame's and t < len; gather the kept rows per utterance and zero-pad
to max(max count, 1).  The reference's serial tf.while_loop with growing concat is replaced by one
prefix-count kernel and one gather kernel; one [B] int32 read-back sizes the output like the reference."""
import collections
import os

import torch

from .. import graph as G
from .... import ops


class ShrinkSpeculation:
    """Sizes the shrunk tensor WITHOUT the device -> host read of the reference (blocks.py:26-36 pads to the longest kept
    sequence, known only after the acoustic encoder has run - the host stalled there for 2-3 ms per COSTT step and then
    enqueued the rest of the step against an idle GPU).  The host allocates `bound` rows per utterance, guessed from the
    maxima of earlier steps; lengths stay on the device (neurst_shrink_clamp), every later kernel masks by them, so any
    bound >= the true maximum gives the reference's values on every valid position.  When a step's maximum did not fit, the
    device-side flag vetoes that step's Adam update (neurst_adam_step_guarded) and the model repeats the batch with the
    full length as soon as the host learns of it (TransformerTPE._settle_shrink): at once in train_step, whose read-back
    of the losses waits for the GPU anyway; at a later call in train_step_async."""

    def __init__(self, device, margin=1.25, slack=8, quantum=32, keep=8):
        self.state = torch.zeros(2, dtype=torch.int32, device=device)     # [overflow flag, max kept length] of the step in flight
        self.margin, self.slack, self.quantum = margin, slack, quantum
        self.history = collections.deque(maxlen=keep)
        self.pending = collections.deque()     # (event, pinned host copy of `state`, bound, batch, step) per step not yet looked at
        self.free = []                         # pinned host buffers of settled steps
        self.force_full = False
        self.last = None
        # measurement aid: never size the shrunk tensor below this many frames (what a step costs when about that many frames
        # are kept: a synthetic batch trains to all-blank CTC output - one kept frame - within a few steps)
        self.min_bound = int(os.environ.get('NEURST_SHRINK_MIN_BOUND', '0'))

    def next_bound(self, T):
        if self.force_full or not self.history:
            return int(T)
        b = int(self.margin * max(self.history)) + self.slack
        b = -(-b // self.quantum) * self.quantum
        return max(1, min(int(T), max(b, self.min_bound)))

    def clamp(self, len_labels, T):
        """enqueue: clamp the lengths to the bound of this step, copy (flag, max) to pinned memory, remember the event"""
        bound = self.next_bound(T)
        ops.shrink_clamp_(len_labels, bound, self.state)
        ev, host = self.free.pop() if self.free else (torch.cuda.Event(), torch.empty(2, dtype=torch.int32).pin_memory())
        host.copy_(self.state, non_blocking=True)
        ev.record()
        self.last = (ev, host, bound)
        return bound


def shrink_layer(encoded, len_encoded, logits, dim_hidden, argmax=None):
    g = G.current()
    B, T, D = encoded.shape
    blank = logits.shape[-1] - 1
    if argmax is None:
        argmax = ops.argmax_rows(logits.view(B * T, -1)).view(B, T)
    index, len_labels = ops.shrink_index(argmax, len_encoded, blank)
    spec = getattr(g, 'shrink_spec', None)
    if spec is not None and g.tape is not None:
        max_label_len = spec.clamp(len_labels, T)               # no host sync: bound from earlier steps, lengths stay on the device
    else:
        max_label_len = max(int(len_labels.cpu().numpy().max()), 1)   # host sync: the reference output shape is data dependent too
    out = ops.shrink_fwd(encoded, index, max_label_len)
    tape = g.tape
    if tape is not None:
        def bwd():
            d = tape.pop(out)
            if d is not None:
                tape.accumulate(encoded, ops.shrink_bwd(d, index, T))
        tape.push(bwd)
    return out, len_labels
