"""Feature transforms of st/tools/tftools/tfAudioTools.py on the front edge of the hot path."""
from .... import ops


def splice_down_sample(features, len_raw, left_num, right_num, rate, out_dtype=ops.F32):
    """splice(left=0, right=num_context) (:69-88) then down_sample(rate) (:91-98), batched:
    row t' = concat(x[rate*t'], ..., x[rate*t'+right_num]) with zeros past each utterance's end.
    Returns (features [B, ceil(Traw/rate), F*(right_num+1)], lengths ceil(len/rate))."""
    assert left_num == 0, 'both templates use left_num = 0 (tfRecord.py:834-837)'
    return ops.splice_downsample(features, len_raw, right_num, rate, out_dtype)


def apply_specaugmant(inputs, F, T, mF, mT, p, key, params=None):
    """SpecAugment (:134-212) with 2 frequency and 2 time masks, applied IN PLACE on
    inputs [B, length, v*channels] fp32.  `params` ([B,8] int32) overrides the sampled masks."""
    assert mF == 2 and mT == 2
    if params is None:
        params = ops.specaug_params(inputs, F, T, p, key)
    return ops.specaug_apply_(inputs, params), params
