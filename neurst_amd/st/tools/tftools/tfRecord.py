"""process_raw_feature of st/tools/tftools/tfRecord.py:827-845 as one device kernel (the reference
runs it per utterance on CPU queue-runner threads before batching)."""
from . import tfAudioTools as tfAudio


def process_raw_feature(seq_raw_features, len_raw, args, out_dtype=0):
    assert not args.data.add_delta, 'add_delta is False in both templates and is not on the hot path'
    return tfAudio.splice_down_sample(seq_raw_features, len_raw, left_num=0, right_num=args.data.num_context,
                                      rate=args.data.downsample, out_dtype=out_dtype)
