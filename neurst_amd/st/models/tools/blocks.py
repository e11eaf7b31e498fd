"""shrink_layer of st/models/tools/blocks.py:8-38 (COSTT): keep frame t iff its CTC arg-max is not
blank, differs from the previous frame's and t < len; gather the kept rows per utterance and zero-pad
to max(max count, 1).  The reference's serial tf.while_loop with growing concat is replaced by one
prefix-count kernel and one gather kernel; one [B] int32 read-back sizes the output like the reference."""
from .. import graph as G
from .... import ops


def shrink_layer(encoded, len_encoded, logits, dim_hidden, argmax=None):
    g = G.current()
    B, T, D = encoded.shape
    blank = logits.shape[-1] - 1
    if argmax is None:
        argmax = ops.argmax_rows(logits.view(B * T, -1)).view(B, T)
    index, len_labels = ops.shrink_index(argmax, len_encoded, blank)
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
