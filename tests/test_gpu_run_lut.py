"""-m gpu: the caller of the path end to end (st/bin/run_lut.py train / infer loops): records -> bucketed batches ->
train steps -> dev decode + scores -> checkpoint -> restore -> the same scores."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


def _write(dirname, n, seed):
    from neurst_amd.st.tools.tftools import tfRecord as R
    rng = np.random.default_rng(seed)
    samples = []
    for i in range(n):
        # a learnable toy task: the label sequence is a function of the utterance's "class", which shifts the features
        cls = i % 3
        L = 2 + cls
        lab = np.append(np.full(L, 4 + cls), 3).astype(np.int32)
        T = int(rng.integers(30, 60))       # >= 10 encoder frames: CTC needs 2 L - 1 = 7 for a label of 4 equal tokens
        feat = (rng.standard_normal((T, 4)) * 0.3 + (cls - 1) * 1.5).astype(np.float32)
        samples.append({'id': np.array([i], np.int32), 'feature': feat, 'label': lab, 'aux_label': lab,
                        'bert_feature': rng.standard_normal((L + 1, 16)).astype(np.float32) * 0.1})
    R.save2tfrecord_ctcbert(samples, dirname, size_file=400)


def test_train_dev_checkpoint_infer_round_trip(tmp_path):
    from neurst_amd.st.bin import run_lut
    from neurst_amd.st.tools.tftools import checkpointTools
    _write(tmp_path / 'train', 48, 0)
    _write(tmp_path / 'dev', 9, 1)
    args = ge.tiny_args(compute_dtype='fp32')
    args.mode, args.num_batch_tokens, args.bucket_boundaries, args.list_bucket_boundaries = 'train', 96, '14,17', None
    args.model.use_ctc_bert, args.data.bert_feat_size, args.data.unit, args.data.max_train_len = True, 16, 'word', 100
    args.idx2token = {i: 'w%d' % i for i in range(11)}
    args.idx2token[3] = '<eos>'
    args.dirs = {'train': {'tfdata': str(tmp_path / 'train')}, 'dev': {'tfdata': str(tmp_path / 'dev')}}
    args.dir_log, args.dir_checkpoint = str(tmp_path / 'log'), str(tmp_path / 'ckpt')
    args.data.train = {'size_dataset': 48}
    args.num_epochs, args.dev_step, args.metrics = 1000, 40, 'WER'
    args.warmup_steps, args.peak, args.decay_steps = 20, 3e-3, 1000
    losses = run_lut.train(args, max_steps=121)
    assert len(losses) == 121 and np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < 0.6 * np.mean(losses[:10]), (losses[:10], losses[-10:])
    ckpts = [l.strip() for l in open(os.path.join(args.dir_checkpoint, 'checkpoint'))]
    assert ckpts[0] == 'model-0.npz' and len(ckpts) >= 2                 # the initial evaluation saves, later ones when WER improves
    assert any(f.startswith('events.out.tfevents') for f in os.listdir(args.dir_log))
    # infer mode: a fresh model restored from the newest checkpoint reproduces the dev scores of that step
    args2 = ge.tiny_args(compute_dtype='fp32')
    for k in ('num_batch_tokens', 'bucket_boundaries', 'idx2token', 'dirs', 'dir_log', 'dir_checkpoint'):
        args2[k] = args[k]
    args2.mode = 'infer'
    args2.model.use_ctc_bert, args2.data.bert_feat_size, args2.data.unit = True, 16, 'word'
    cer, wer, bleu = run_lut.infer(args2)
    step = int(checkpointTools.load_variable(args.dir_checkpoint, 'global_step'))
    assert step == int(ckpts[-1].split('-')[1].split('.')[0]) and step > 0
    assert 0 <= wer and np.isfinite(cer) and 0 <= bleu <= 100
    # the training run logged the same WER for that step in its event file: read it back
    from tests.test_metrics import _event_classes
    from oracle import records as OR
    Event = _event_classes()
    logged = {}
    for f in sorted(os.listdir(args.dir_log)):
        if f.startswith('events.out.tfevents'):
            for payload in OR.unframe(open(os.path.join(args.dir_log, f), 'rb').read()):
                e = Event()
                e.ParseFromString(payload)
                for v in e.summary.value:
                    logged[(v.tag, e.step)] = v.simple_value
    assert ('loss', 0) in logged and ('learning_rate', 5) in logged
    assert logged[('dev_wer', step)] == pytest.approx(wer, rel=1e-6)
