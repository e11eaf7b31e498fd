"""Tensor helpers of st/tools/tftools/tfTools.py that sit on the hot path."""
from .... import ops


def right_shift_rows(p, shift, pad, lens=None):
    """tfTools.py:195-198 ([pad, p[:, :-1]]) fused with the label mask of
    st/models/TransformerTPE.py:335-338 when `lens` is given (positions >= len -> 0)."""
    assert type(shift) is int and shift == 1
    assert lens is not None, 'the fused kernel applies the length mask; pass the label lengths'
    return ops.build_decoder_input(p, lens, int(pad))
