"""Input pipeline: the native TFRecord / tf.train.Example reader, the writers and the bucketed
batching of neurst_amd/st/tools/tftools/tfRecord.py against oracle/records.py (bitwise CRC-32C, struct framing,
google.protobuf's wire format, a plain restatement of bucket_by_sequence_length).  CPU only, except the last test.
parity unpinned against TensorFlow's own files (the reference holds none); CRC-32C is pinned by RFC 3720 B.4."""
import os
import struct

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from neurst_amd import lib
from neurst_amd.st.tools.tftools import tfRecord as R
from oracle import records as O


def test_crc32c_known_answers_rfc3720():
    L = lib.load()
    vectors = [(b'123456789', 0xE3069283), (bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E),
               (bytes(range(31, -1, -1)), 0x113FDB5C), (b'', 0)]
    for data, want in vectors:
        assert L.neurst_crc32c(data, len(data)) == want
        assert O.crc32c(data) == want
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 1000, 4099):                  # every head / tail alignment of the 8-byte loop
        blob = rng.integers(0, 256, size=n + 3, dtype=np.uint8)
        for shift in range(3):
            part = blob[shift:shift + n]
            assert L.neurst_crc32c(part.ctypes.data, n) == O.crc32c(part.tobytes())


def _samples(n, rng, dim=4, bert=None, min_len=1, max_len=40):
    out = []
    for i in range(n):
        T = int(rng.integers(min_len, max_len + 1))
        s = {'id': np.array([i], np.int32), 'feature': rng.standard_normal((T, dim)).astype(np.float32),
             'label': rng.integers(4, 10, size=int(rng.integers(1, 7))).astype(np.int32),
             'aux_label': rng.integers(4, 10, size=int(rng.integers(1, 5))).astype(np.int32)}
        if bert:
            s['bert_feature'] = rng.standard_normal((int(rng.integers(1, 6)), bert)).astype(np.float32)
        out.append(s)
    return out


def test_writer_output_is_what_protobuf_and_the_framing_oracle_read(tmp_path):
    rng = np.random.default_rng(1)
    samples = _samples(7, rng, bert=3) + [{'id': np.zeros(0, np.int32), 'feature': np.zeros((0, 4), np.float32),
                                           'label': np.zeros(0, np.int32), 'aux_label': np.zeros(0, np.int32),
                                           'bert_feature': np.zeros((0, 3), np.float32)}]
    R.save2tfrecord_ctcbert(samples, tmp_path, size_file=50)
    files = sorted(R.fentch_filelist(str(tmp_path)), key=lambda p: int(os.path.basename(p).split('.')[0]))
    assert len(files) > 1                                    # rolled over every 50 frames (:357-360)
    got = []
    for f in files:
        for payload in O.unframe(open(f, 'rb').read()):      # checks both checksums of every record
            got.append(O.decode_example(payload))
    assert len(got) == len(samples)
    for g, s in zip(got, samples):
        assert set(g) == set(s)
        for k in s:
            assert g[k] == s[k].tobytes()
    info = open(tmp_path / 'tfdata.info').read()
    assert 'dim_feature 4' in info and 'size_dataset 8' in info and 'num_tokens {}'.format(sum(len(s['feature']) for s in samples)) in info


def test_native_reader_reads_what_protobuf_wrote(tmp_path):
    """the other direction: payloads serialized by google.protobuf (its own key order, plus an unknown extra field and
    a non-bytes feature), framed by the struct oracle"""
    rng = np.random.default_rng(2)
    samples = _samples(9, rng, bert=5)
    path = tmp_path / '0.recode'
    Example = O.example_class()
    with open(path, 'wb') as f:
        for s in samples:
            payload = O.encode_example(s)
            payload = payload + bytes([0x10, 0x05])           # an unknown varint field 2 after `features`: skipped
            f.write(O.frame(payload))
    rf = R.RecordFile(str(path))
    assert len(rf) == len(samples)
    for i, s in enumerate(samples):
        assert np.array_equal(rf.field(i, 'feature', np.float32).reshape(-1, 4), s['feature'])
        assert np.array_equal(rf.field(i, 'label', np.int32), s['label'])
        assert np.array_equal(rf.field(i, 'bert_feature', np.float32).reshape(-1, 5), s['bert_feature'])
        assert np.array_equal(rf.field(i, 'id', np.int32), s['id'])
    with pytest.raises(lib.NeurstLibError, match='no feature'):
        rf.field(0, 'lang_label', np.int32)
    assert Example is O.example_class()


def test_empty_truncated_and_corrupted_files_are_reported(tmp_path):
    (tmp_path / 'empty.recode').write_bytes(b'')
    assert len(R.RecordFile(str(tmp_path / 'empty.recode'))) == 0
    good = O.frame(O.encode_example({'feature': np.ones((2, 4), np.float32)})) * 3
    for cut in (5, 12, len(good) - 1, len(good) // 3 + 20):
        p = tmp_path / 'cut{}.recode'.format(cut)
        p.write_bytes(good[:cut])
        with pytest.raises(lib.NeurstLibError, match='truncated|runs past'):
            R.RecordFile(str(p))
    one = len(good) // 3
    for pos, what in ((3, 'length'), (9, 'length'), (40, 'data'), (one - 2, 'data'), (one + 30, 'data of record 1')):
        bad = bytearray(good)
        bad[pos] ^= 0x10
        p = tmp_path / 'bad{}.recode'.format(pos)
        p.write_bytes(bytes(bad))
        with pytest.raises(lib.NeurstLibError, match=what):
            R.RecordFile(str(p))
    bad = bytearray(good)
    bad[40] ^= 0x10
    (tmp_path / 'unchecked.recode').write_bytes(bytes(bad))
    assert len(R.RecordFile(str(tmp_path / 'unchecked.recode'), verify_crc=False)) == 3
    # a length that claims more than the file holds must not be followed
    huge = struct.pack('<Q', 1 << 62)
    (tmp_path / 'huge.recode').write_bytes(huge + struct.pack('<I', O.masked_crc(huge)) + b'abcd')
    with pytest.raises(lib.NeurstLibError, match='runs past'):
        R.RecordFile(str(tmp_path / 'huge.recode'))


def test_native_parser_survives_mutated_files_under_sanitizers(tmp_path):
    """tests/native/fuzz_records.cpp: csrc/records.cpp compiled for the host with AddressSanitizer + UBSan, fed bit-flipped,
    truncated and random record files / Example payloads from exact-size heap buffers; every accepted view must lie
    inside its record"""
    import shutil
    import subprocess
    if shutil.which('g++') is None:
        pytest.skip('no host compiler')
    root = ge.ROOT
    exe = str(tmp_path / 'fuzz_records')
    subprocess.check_call(['g++', '-O1', '-g', '-std=c++17', '-msse4.2', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                           '-I', os.path.join(root, 'include'), os.path.join(root, 'tests', 'native', 'fuzz_records.cpp'),
                           os.path.join(root, 'neurst_amd', 'csrc', 'records.cpp'), '-o', exe])
    out = subprocess.run([exe, '40000'], capture_output=True, text=True, timeout=300)
    if out.returncode != 0 and 'ASan runtime does not come first' in out.stderr:
        pytest.skip('the environment preloads a library ahead of the sanitizer runtime')
    assert out.returncode == 0 and 'fuzz ok' in out.stdout, out.stdout[-500:] + out.stderr[-2000:]


def _args(mode='train', **over):
    args = ge.tiny_args(compute_dtype='fp32', **over)
    args.mode = mode
    args.num_batch_tokens, args.bucket_boundaries = 24, '4,8,12'
    args.list_bucket_boundaries = None
    args.data.max_train_len = 30
    args.data.bert_feat_size = 16
    args.model.use_ctc_bert, args.model.use_multilabel = True, False
    return args


@pytest.mark.parametrize('mode', ['train', 'infer'])
def test_bucketed_batches_match_the_oracle(tmp_path, mode):
    rng = np.random.default_rng(3)
    samples = _samples(57, rng, bert=16)
    R.save2tfrecord_ctcbert(samples, tmp_path, size_file=300)
    args = _args(mode)
    reader = R.TFReader(str(tmp_path), args, is_train=False)
    assert reader.list_bucket_boundaries == [4, 8, 12] and reader.list_batch_size == [6, 3, 2, 1]     # arguments.py:86-87
    # one pass in sorted file order ('10.recode' < '2.recode' as strings, like list.sort() in the reference)
    order = []
    for f in sorted(R.fentch_filelist(str(tmp_path))):
        order += [int(np.frombuffer(O.decode_example(p)['id'], np.int32)[0]) for p in O.unframe(open(f, 'rb').read())]
    cut = 30 if mode == 'train' else None
    lens = [O.transformed_len(len(samples[i]['feature']), 3, cut) for i in order]
    if mode == 'train':          # num_epochs=None (:629): the data repeats and remainders stay queued across the epoch boundary
        order, lens = order * 2, lens * 2
    want = O.bucket_batches(lens, [4, 8, 12], [6, 3, 2, 1])
    if mode == 'train':
        want = want[:-4]         # without the oracle's end-of-input flush (at most one batch per bucket)
        assert any(max(g) >= 57 > min(g) for g in want)          # some batch does straddle the two epochs
    it = iter(reader.fentch_ctcbert_batch_bucket())
    for group in want:
        batch = next(it)
        ids = [order[j] for j in group]
        assert batch['id'].numpy().reshape(-1).tolist() == ids
        feats = [samples[i]['feature'][:cut] for i in ids]
        assert np.array_equal(batch['feature'].numpy(), O.pad_batch(feats))
        assert batch['len_fea'].tolist() == [len(f) for f in feats]
        for name, key in (('label', 'label'), ('aux_label', 'aux_label'), ('bert_feat', 'bert_feature')):
            assert np.array_equal(batch[name].numpy(), O.pad_batch([samples[i][key] for i in ids]))
            assert batch['len_' + name].tolist() == [len(samples[i][key]) for i in ids]
    if mode == 'infer':
        with pytest.raises(StopIteration):
            next(it)
    else:
        assert next(it)['feature'].shape[0] >= 1


def test_train_reader_shuffles_files_per_epoch_and_prefetcher_preserves_order(tmp_path):
    rng = np.random.default_rng(4)
    samples = _samples(40, rng, dim=4)
    R.save2tfrecord_multilabel(samples, tmp_path, size_file=60)
    assert len(R.fentch_filelist(str(tmp_path))) >= 5
    args = _args('train', loss_type='ce_ctc')
    args.model.use_multilabel, args.model.use_ctc_bert = True, False
    direct = iter(R.TFReader(str(tmp_path), args, is_train=True, seed=5))
    pre = R.Prefetcher(R.TFReader(str(tmp_path), args, is_train=True, seed=5), depth=2)
    for _ in range(60):                                    # more than two epochs
        a, b = next(direct), next(pre)
        assert sorted(a) == sorted(b) == sorted(['id', 'feature', 'len_fea', 'label', 'len_label', 'aux_label', 'len_aux_label'])
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]))
    pre.close()
    ex = R.TFReader(str(tmp_path), args, is_train=True, seed=5)._epochs()
    seen = [int(next(ex)[0][0]) for _ in range(80)]
    first, second = seen[:40], seen[40:80]
    assert sorted(first) == list(range(40))                # an epoch sees every example once...
    assert first != second                                 # ...in a new file order each time


def test_reader_skips_utterances_without_frames(tmp_path):
    rng = np.random.default_rng(8)
    samples = _samples(6, rng, bert=16)
    samples[2]['feature'] = np.zeros((0, 4), np.float32)
    R.save2tfrecord_ctcbert(samples, tmp_path)
    ids = [int(ex[0][0]) for ex in R.readTFRecord_ctcbert(str(tmp_path), _args('infer'))]
    assert ids == [0, 1, 3, 4, 5]


def test_prefetcher_hands_errors_to_the_consumer(tmp_path):
    def gen():
        yield 1
        raise ValueError('boom')
    p = R.Prefetcher(gen())
    assert next(p) == 1
    with pytest.raises(ValueError, match='boom'):
        next(p)


@pytest.mark.gpu
def test_train_step_from_records_equals_train_step_from_arrays(tmp_path):
    """records -> TFReader (pinned batches, prefetch thread) -> train_step gives the same losses as the same padded
    arrays handed over directly: the pipeline changes no bit on the way to the device"""
    from neurst_amd.st.models.TransformerTPE import GlobalStep
    rng = np.random.default_rng(6)
    samples = _samples(12, rng, dim=4, bert=16, min_len=20, max_len=40)
    for s in samples:
        s['label'][-1] = 3
        s['aux_label'][-1] = 3
    R.save2tfrecord_ctcbert(samples, tmp_path, size_file=100)
    args = _args('train', loss_type='ctc_kd_ce')
    args.num_batch_tokens, args.bucket_boundaries, args.list_bucket_boundaries = 64, '16', None
    model = args.Model(GlobalStep(0), encoder=args.model.encoder.type, decoder=args.model.decoder.type, is_train=True, args=args)
    reader = R.TFReader(str(tmp_path), args, is_train=False, pin_memory=True)
    pre = R.Prefetcher(reader, depth=2)
    batch = next(pre)
    pre.close()
    assert batch['feature'].is_pinned() and batch['feature'].shape[0] == 4
    a = model.forward_backward(batch).cpu().numpy()
    ga = model.store.grads.clone()
    plain = {k: (v.numpy().copy() if isinstance(v, torch.Tensor) else v.copy()) for k, v in batch.items()}
    b = model.forward_backward(plain).cpu().numpy()
    assert np.isfinite(a).all() and np.array_equal(a, b)
    assert torch.allclose(ga, model.store.grads, rtol=1e-5, atol=1e-7)
