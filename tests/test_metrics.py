"""Dev-loop metrics: st/tools/{textTools,wer,bleu,performanceTools,summaryTools}.py of the build
against oracle/metrics.py and known answers.  Known answers held by the reference: the `unpadding` docstring example
(st/tools/textTools.py:17-20).  BLEU: hand-derived cases below; otherwise parity unpinned (see oracle/metrics.py)."""
import os
import struct

import numpy as np
import pytest

import __graft_entry__ as ge
from neurst_amd.st.tools import bleu as B
from neurst_amd.st.tools import performanceTools as PT
from neurst_amd.st.tools import textTools as TT
from neurst_amd.st.tools import wer as W
from neurst_amd.st.tools.summaryTools import Summary
from oracle import metrics as OM
from oracle import records as OR


def test_unpadding_docstring_example_of_the_reference():
    a = np.array([2, 2, 3, 4, 5, 1, 0, 0, 0])
    assert TT.unpadding(a, 1).tolist() == [2, 2, 3, 4, 5]
    assert TT.unpadding(a, None, min_idx=0, max_idx=5).tolist() == [2, 2, 3, 4, 1]
    assert TT.unpadding(np.array([], np.int32), 3).tolist() == []


def test_edit_distance_and_alignment_counts_match_the_oracle():
    assert TT.edit_distance(list('kitten'), list('sitting')) == 3
    assert TT.edit_distance([], []) == 0 and TT.edit_distance([1, 2], []) == 2 and TT.edit_distance([], [5]) == 1
    assert W.wer(['a', 'b', 'c'], ['a', 'x', 'c']) == (1.0, 0.0, 0.0)
    assert W.wer(['a', 'b', 'c'], ['a', 'c']) == (0.0, 0.0, 1.0)
    assert W.wer(['a', 'c'], ['a', 'b', 'c']) == (0.0, 1.0, 0.0)
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 4, size=int(rng.integers(0, 12))).tolist()
        b = rng.integers(0, 4, size=int(rng.integers(0, 12))).tolist()
        assert TT.edit_distance(a, b) == OM.levenshtein(a, b)
        s, i, d = W.wer(a, b)
        assert (s, i, d) == tuple(float(x) for x in OM.wer(a, b))       # same tie-breaking, not only the same total
        assert s + i + d == OM.levenshtein(a, b)


def test_batch_cer_wer_and_array2text():
    idx2token = {0: '<pad>', 1: '<unk>', 2: '<sos>', 3: '<eos>', 4: 'he@@', 5: 'llo', 6: 'world', 7: 'wor@@', 8: 'ld', 9: '<blk>'}
    args = ge.tiny_args(compute_dtype='fp32')
    ref = np.array([[4, 5, 6, 3, 0, 0], [7, 8, 3, 0, 0, 0]])
    res = np.array([[4, 5, 7, 8, 3, 6], [6, 6, 3, 0, 0, 0]])
    assert TT.array2text(ref[0], 'subword', idx2token, 3, 0, 9, args=args) == 'hello world'
    assert TT.array2text(res[0], 'subword', idx2token, 3, 0, 9, args=args) == 'hello world'
    assert TT.array2text(ref[0], 'word', idx2token, 3, 0, 9) == 'he@@ llo world'
    assert TT.array2text(ref[0], 'char', idx2token, 3, 0, 9) == 'he@@lloworld'
    assert TT.array2text(ref[0], 'subword', idx2token, 3, 0, 9, delimiter='@@') == 'he llo world'
    with pytest.raises(NotImplementedError):
        TT.array2text(ref[0], 'phone', idx2token)
    assert TT.batch_cer(res, ref, eos_idx=3, min_idx=0, max_idx=9) == (2 + 2, 3 + 2)
    assert TT.batch_wer(res, ref, idx2token, 'subword', eos_idx=3, min_idx=0, max_idx=9, args=args) == (0 + 1, 2 + 1)      # 'world world' vs 'world'


HAND = [
    # identical corpus: every precision 1, no brevity penalty
    ([['the cat sat on the mat', 'a b c d e']], ['the cat sat on the mat', 'a b c d e'], False,
     'BLEU = 100.00, 100.0/100.0/100.0/100.0 (BP=1.000, ratio=1.000, hyp_len=11, ref_len=11)'),
    # one substitution in 7 words: 6/7, 4/6, 2/5, 0/4 -> log(0) = -9999999999 makes the mean vanish
    ([['a b c x d e f']], ['a b c y d e f'], False, 'BLEU = 0.00, 85.7/66.7/40.0/0.0 (BP=1.000, ratio=1.000, hyp_len=7, ref_len=7)'),
    # clipping: 'the' occurs twice in the reference, seven times in the output -> 2/7; no bigram matches
    ([['the cat the mat']], ['the the the the the the the'], False, 'BLEU = 0.00, 28.6/0.0/0.0/0.0 (BP=1.000, ratio=1.750, hyp_len=7, ref_len=4)'),
    # prefix of 5 of 10 words: precisions 1, BP = exp(1 - 10/5) = 0.368
    ([['a b c d e f g h i j']], ['a b c d e'], False, 'BLEU = 36.79, 100.0/100.0/100.0/100.0 (BP=0.368, ratio=0.500, hyp_len=5, ref_len=10)'),
    # one substitution in 8 words: 7/8, 5/7, 3/6, 2/5 - their product is exactly 1/8, so BLEU = 100 * 8^(-1/4) = 59.46
    ([['a b c d e f g h']], ['a b c d e x g h'], False, 'BLEU = 59.46, 87.5/71.4/50.0/40.0 (BP=1.000, ratio=1.000, hyp_len=8, ref_len=8)'),
    # -lc
    ([['The Cat sat on the mat']], ['the cat SAT on the mat'], True, 'BLEU = 100.00, 100.0/100.0/100.0/100.0 (BP=1.000, ratio=1.000, hyp_len=6, ref_len=6)'),
    # case-sensitive: unigrams the(1 of 2)/on/mat = 3/6, bigrams 'on the', 'the mat' = 2/5, trigram 'on the mat' = 1/4, no 4-gram
    ([['The Cat sat on the mat']], ['the cat SAT on the mat'], False, 'BLEU = 0.00, 50.0/40.0/25.0/0.0 (BP=1.000, ratio=1.000, hyp_len=6, ref_len=6)'),
    # two references of lengths 4 and 8 around a 6-word output: equally close, the shorter counts
    ([['a b c d'], ['a b c d e f g h']], ['a b c d e f'], False, 'BLEU = 100.00, 100.0/100.0/100.0/100.0 (BP=1.000, ratio=1.500, hyp_len=6, ref_len=4)'),
    # empty reference corpus length
    ([['']], [''], False, 'BLEU = 0, 0/0/0/0 (BP=0, ratio=0, hyp_len=0, ref_len=0)'),
]


@pytest.mark.parametrize('refs,hyp,lc,want', HAND)
def test_bleu_hand_derived_cases(refs, hyp, lc, want):
    assert OM.multi_bleu(refs, hyp, lc) == want
    assert B.multi_bleu(refs, hyp, lc)[1] == want


def test_bleu_matches_the_oracle_on_random_corpora(tmp_path):
    rng = np.random.default_rng(1)
    vocab = ['the', 'a', 'cat', 'Dog', 'sat', 'ran', 'on', 'mat', 'tree', 'big', ',', '.']
    sent = lambda lo, hi: ' '.join(rng.choice(vocab, size=int(rng.integers(lo, hi + 1))))

    def noisy(line, p):
        return ' '.join(str(rng.choice(vocab)) if rng.random() < p else w for w in line.split() if rng.random() > p / 2)
    for trial in range(30):
        n = int(rng.integers(1, 25))
        ref = [sent(0 if trial % 5 == 0 else 1, 15) for _ in range(n)]
        refs = [ref] + [[noisy(l, 0.2) for l in ref] for _ in range(trial % 3)]
        hyp = [noisy(l, [0.0, 0.1, 0.4][trial % 3]) for l in ref]
        lc = trial % 4 == 0
        if sum(len(h.split()) for h in hyp) == 0:
            continue
        score, text = B.multi_bleu(refs, hyp, lc)
        assert text == OM.multi_bleu(refs, hyp, lc)
        assert abs(score - float(text.split()[2].rstrip(','))) < 0.006
    ref, hyp = [sent(3, 12) for _ in range(9)], None
    hyp = [noisy(l, 0.15) for l in ref]
    (tmp_path / 'ref.txt').write_text(''.join(l + '\n' for l in ref))
    (tmp_path / 'hyp.txt').write_text(''.join(l + '\n' for l in hyp))
    assert B.bleu_from_files(str(tmp_path / 'ref.txt'), str(tmp_path / 'hyp.txt')) == float(OM.multi_bleu([ref], hyp).split()[2].rstrip(','))


def test_calc_wer_cer_on_files(tmp_path, capsys):
    (tmp_path / 'ref.txt').write_text('the cat sat\nhello world\n')
    (tmp_path / 'dec.txt').write_text('the cat sat down\nhallo world\n')
    assert W.calc_wer(str(tmp_path / 'ref.txt'), str(tmp_path / 'dec.txt')) == pytest.approx(2 / 5)
    assert W.calc_cer(str(tmp_path / 'ref.txt'), str(tmp_path / 'dec.txt')) == pytest.approx((4 + 1) / (9 + 10))
    assert 'word error rate' in capsys.readouterr().out


def _event_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='oracle_event.proto', package='tensorflow', syntax='proto3')
    m = fd.message_type.add(name='HistogramProto')
    for i, n in enumerate(['min', 'max', 'num', 'sum', 'sum_squares'], 1):
        m.field.add(name=n, number=i, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    m.field.add(name='bucket_limit', number=6, type=F.TYPE_DOUBLE, label=F.LABEL_REPEATED)
    m.field.add(name='bucket', number=7, type=F.TYPE_DOUBLE, label=F.LABEL_REPEATED)
    m = fd.message_type.add(name='Summary')
    v = m.nested_type.add(name='Value')
    v.field.add(name='tag', number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    v.field.add(name='simple_value', number=2, type=F.TYPE_FLOAT, label=F.LABEL_OPTIONAL)
    v.field.add(name='histo', number=5, type=F.TYPE_MESSAGE, type_name='.tensorflow.HistogramProto', label=F.LABEL_OPTIONAL)
    m.field.add(name='value', number=1, type=F.TYPE_MESSAGE, type_name='.tensorflow.Summary.Value', label=F.LABEL_REPEATED)
    m = fd.message_type.add(name='Event')
    m.field.add(name='wall_time', number=1, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    m.field.add(name='step', number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    m.field.add(name='file_version', number=3, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    m.field.add(name='summary', number=5, type=F.TYPE_MESSAGE, type_name='.tensorflow.Summary', label=F.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.Event'))


def test_summary_event_file_is_readable_as_tensorboard_events(tmp_path):
    s = Summary(str(tmp_path))
    s.summary_scalar('loss', 1.25, 7)
    s.summary_scalar('dev/wer', 0.5, 1 << 40)
    vals = np.random.default_rng(0).standard_normal(500)
    s.histogram('weights', vals, 9, bins=10)
    s.summary_distribution(np.array([[0.25, 0.75], [0.5, 0.5]]), 'dist')
    s.close()
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].startswith('events.out.tfevents.')
    Event = _event_classes()
    events = []
    for payload in OR.unframe(open(tmp_path / files[0], 'rb').read()):       # framing + both CRCs
        e = Event()
        e.ParseFromString(payload)
        events.append(e)
    assert events[0].file_version == 'brain.Event:2' and events[0].wall_time > 1e9
    assert (events[1].step, events[1].summary.value[0].tag, events[1].summary.value[0].simple_value) == (7, 'loss', 1.25)
    assert (events[2].step, events[2].summary.value[0].tag, events[2].summary.value[0].simple_value) == (1 << 40, 'dev/wer', 0.5)
    h = events[3].summary.value[0].histo
    counts, edges = np.histogram(vals, bins=10)
    assert events[3].summary.value[0].tag == 'weights' and events[3].step == 9
    assert (h.min, h.max, h.num) == (vals.min(), vals.max(), 500.0)
    assert h.sum == pytest.approx(vals.sum()) and h.sum_squares == pytest.approx((vals ** 2).sum())
    assert list(h.bucket) == counts.tolist() and np.allclose(list(h.bucket_limit), edges[1:])
    tags = [(e.step, e.summary.value[0].tag, e.summary.value[0].simple_value) for e in events[4:]]
    assert tags == [(0, 'dist_0', 0.25), (0, 'dist_1', 0.75), (1, 'dist_0', 0.5), (1, 'dist_1', 0.5)]


class _FakeModel:
    """stands in for the inference model: returns the labels with every 5th utterance's first token replaced"""

    def __init__(self):
        self.last = {}
        self.n = 0

    def infer(self, features, len_feas):
        self.last['shape_batch'] = tuple(np.asarray(features).shape)
        out = self.labels.copy()
        for row in out:
            if self.n % 5 == 0:
                row[0] = 6 if row[0] != 6 else 7
            self.n += 1
        return out, self.aux.copy()


def test_dev_test_scores_and_files(tmp_path):
    idx2token = {0: '<pad>', 1: '<unk>', 2: '<sos>', 3: '<eos>', 4: 'aa', 5: 'bb', 6: 'cc', 7: 'dd', 8: 'ee', 9: 'ff', 10: '<blk>'}
    args = ge.tiny_args(compute_dtype='fp32')
    args.idx2token, args.data.unit = idx2token, 'word'
    rng = np.random.default_rng(2)
    model = _FakeModel()
    batches, refs_txt, hyp_txt = [], [], []
    for _ in range(4):
        lab = rng.integers(4, 10, size=(5, 8)).astype(np.int32)
        lab[:, -1] = 3
        batches.append({'feature': np.zeros((5, 12, 4), np.float32), 'len_fea': np.full(5, 12, np.int32), 'label': lab, 'aux_label': lab.copy()})

    def loader():
        for b in batches:
            model.labels, model.aux = b['label'], b['aux_label']
            yield b
    cer, wer, bleu = PT.dev_test(args, 0, loader(), model, None, 'word', idx2token, eos_idx=3, min_idx=0, max_idx=10, out_dir=str(tmp_path))
    assert cer == pytest.approx(4 / (20 * 7)) and wer == pytest.approx(4 / (20 * 7))
    ref_lines = open(tmp_path / 'dev_ast_ref.txt').read().split('\n')[:-1]
    res_lines = open(tmp_path / 'dev_ast_res.txt').read().split('\n')[:-1]
    assert len(ref_lines) == len(res_lines) == 20 and sum(a != b for a, b in zip(ref_lines, res_lines)) == 4
    assert ref_lines[1] == ' '.join(idx2token[int(i)] for i in batches[0]['label'][1][:-1]) + ' '      # '<eos>' removed, its blank kept (:64)
    assert bleu == float(OM.multi_bleu([ref_lines], res_lines).split()[2].rstrip(','))
    assert open(tmp_path / 'dev_asr_ref.txt').read() == open(tmp_path / 'dev_asr_res.txt').read() != ''


@pytest.mark.gpu
def test_dev_test_with_the_model_and_the_record_reader(tmp_path):
    """records -> TFReader (infer mode, one pass) -> TransformerTPE.infer (beam 2) -> dev_test; the scores equal what the
    oracle computes from the text files dev_test wrote"""
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    from neurst_amd.st.tools.tftools import tfRecord as R
    args = ge.tiny_args(compute_dtype='fp32')
    args.mode, args.num_batch_tokens, args.bucket_boundaries, args.list_bucket_boundaries = 'infer', 40, '8,16', None
    args.model.use_ctc_bert, args.data.bert_feat_size, args.beam_size = True, 16, 2
    args.idx2token = {i: 'w%d' % i for i in range(11)}
    args.idx2token[3] = '<eos>'
    args.data.unit = 'word'
    rng = np.random.default_rng(3)
    samples = []
    for i in range(13):
        lab = np.append(rng.integers(4, 10, size=int(rng.integers(1, 6))), 3).astype(np.int32)
        samples.append({'id': np.array([i], np.int32), 'feature': rng.standard_normal((int(rng.integers(9, 60)), 4)).astype(np.float32),
                        'label': lab, 'aux_label': lab, 'bert_feature': np.zeros((2, 16), np.float32)})
    R.save2tfrecord_ctcbert(samples, tmp_path / 'data')
    model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=False, args=args)
    reader = R.TFReader(str(tmp_path / 'data'), args, is_train=False)
    cer, wer, bleu = PT.dev_test(args, 0, reader, model, None, 'word', args.idx2token, eos_idx=3, min_idx=0, max_idx=10, out_dir=str(tmp_path))
    ref_lines = open(tmp_path / 'dev_ast_ref.txt').read().split('\n')[:-1]
    res_lines = open(tmp_path / 'dev_ast_res.txt').read().split('\n')[:-1]
    assert len(ref_lines) == len(res_lines) == 13
    dist = sum(OM.levenshtein(a.split(), b.split()) for a, b in zip(res_lines, ref_lines))
    assert wer == pytest.approx(dist / sum(len(r.split()) for r in ref_lines))
    assert np.isfinite(cer) and 0 <= bleu <= 100
    assert bleu == float(OM.multi_bleu([ref_lines], res_lines).split()[2].rstrip(','))
