"""The input pipeline either side of the hot path, mirroring st/tools/tftools/tfRecord.py:

* `save2tfrecord*`  (:197-458)  writers of `<n>.recode` files + `tfdata.info`
* `readTFRecord*`   (:474-672)  per-example readers (id, feature, label[, aux_label][, bert_feature])
* `TFReader`        (:858-1062) length-bucketed, token-budgeted batches
* `process_raw_feature` (:827-845) splice + down-sample - here ONE device kernel applied to the padded batch (a1),
  not per utterance on 64 CPU queue threads.

The reference hands all of this to TensorFlow (TFRecordReader, parse_single_example, decode_raw,
bucket_by_sequence_length).  Here a file is memory-mapped and indexed once by the native scanner
(`neurst_tfrecord_scan`, CRC-32C checked), fields are numpy views into the mapping, and a batch is assembled with one
copy per field into (optionally pinned) host memory, so the only other copy is the H2D transfer.  Batching follows
bucket_by_sequence_length for one enqueuing thread: deterministic, where the reference's 64 threads are not.
"""
import ctypes
import glob
import os
import queue
import random
import struct
import threading

import numpy as np

from .... import lib as _lib
from . import tfAudioTools as tfAudio

_TORCH_DTYPE = {}


def _init_dtypes():
    import torch
    _TORCH_DTYPE.update({'float32': torch.float32, 'int32': torch.int32, 'int64': torch.int64, 'uint8': torch.uint8, 'float64': torch.float64})


_init_dtypes()


def process_raw_feature(seq_raw_features, len_raw, args, out_dtype=0):
    assert not args.data.add_delta, 'add_delta is False in both templates and is not on the hot path'
    return tfAudio.splice_down_sample(seq_raw_features, len_raw, left_num=0, right_num=args.data.num_context,
                                      rate=args.data.downsample, out_dtype=out_dtype)


# ------------------------------------------------------------------------------------------------ container + payload
def _masked_crc(data):
    c = _lib.load().neurst_crc32c(data, len(data))
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _ld(field, payload):
    """length-delimited protobuf field"""
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def serialize_example(sample):
    """tf.train.Example with one bytes feature per key (tfRecord.py:362-370); values are numpy arrays (`.tostring()`)."""
    entries = b''
    for key, value in sample.items():
        raw = value if isinstance(value, (bytes, bytearray)) else np.ascontiguousarray(value).tobytes()
        feature = _ld(1, _ld(1, raw))                                  # Feature.bytes_list = BytesList{value: [raw]}
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feature))     # Features.feature map entry
    return _ld(1, entries)                                             # Example.features


class TFRecordWriter:
    """tf.python_io.TFRecordWriter (:122, :228 ...)"""

    def __init__(self, path):
        self.f = open(path, 'wb')

    def write(self, payload):
        head = struct.pack('<Q', len(payload))
        self.f.write(head + struct.pack('<I', _masked_crc(head)))
        self.f.write(payload)
        self.f.write(struct.pack('<I', _masked_crc(payload)))

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RecordFile:
    """One `*.recode` file: read-only mapping + record index."""

    def __init__(self, path, verify_crc=True):
        L = _lib.load()
        self.path = path
        size = os.path.getsize(path)
        self.buf = np.memmap(path, dtype=np.uint8, mode='r') if size else np.zeros(0, np.uint8)
        self.base = self.buf.ctypes.data
        n = L.neurst_tfrecord_scan(self.base, size, None, None, 0, 1 if verify_crc else 0)
        if n < 0:
            _lib.check(int(n), 'tfrecord_scan({})'.format(path))
        self.offsets = np.zeros(n, np.int64)
        self.lengths = np.zeros(n, np.int64)
        if n:
            got = L.neurst_tfrecord_scan(self.base, size, self.offsets.ctypes.data, self.lengths.ctypes.data, n, 0)
            assert got == n
        self._off, self._size = ctypes.c_int64(), ctypes.c_int64()
        self._fn = L.neurst_example_bytes_feature

    def __len__(self):
        return len(self.offsets)

    def field(self, i, key, dtype):
        """zero-copy view of feature `key` of record i, decoded like tf.decode_raw(..., dtype)"""
        off, n = int(self.offsets[i]), int(self.lengths[i])
        rc = self._fn(self.base + off, n, key.encode(), ctypes.byref(self._off), ctypes.byref(self._size))
        if rc != 0:
            _lib.check(rc, 'example_bytes_feature({}, record {})'.format(self.path, i))
        a = off + self._off.value
        raw = self.buf[a:a + self._size.value]
        item = np.dtype(dtype).itemsize
        if raw.size % item:
            raise ValueError('{} record {}: feature {!r} has {} bytes, not a multiple of {}'.format(self.path, i, key, raw.size, item))
        return raw.view(dtype)            # payloads start at arbitrary bytes: numpy handles the unaligned view, no copy


def fentch_filelist(dir_data):
    """:848-855 - every `*.recode` of the comma-separated directories"""
    file_list = []
    for dir_split in str(dir_data).split(','):
        assert os.path.isdir(dir_split), dir_split
        file_list += glob.glob(os.path.join(dir_split, '*.recode'))
    return file_list


# ------------------------------------------------------------------------------------------------ writers (:197-458)
def _save(dataset, dir_save, keys, size_file):
    """the common body of save2tfrecord*: a new `<n>.recode` every `size_file` feature frames, then tfdata.info"""
    os.makedirs(str(dir_save), exist_ok=True)
    num_token, idx_file, num_damaged, writer, dim_feature, count = 0, -1, 0, None, None, 0
    for sample in dataset:
        if not sample:
            num_damaged += 1
            continue
        dim_feature = sample['feature'].shape[-1]
        if (num_token // size_file) > idx_file:
            idx_file = num_token // size_file
            if writer:
                writer.close()
            writer = TFRecordWriter(os.path.join(str(dir_save), '{}.recode'.format(idx_file)))
        writer.write(serialize_example({k: sample[k] for k in keys}))
        num_token += len(sample['feature'])
        count += 1
    if writer:
        writer.close()
    with open(os.path.join(str(dir_save), 'tfdata.info'), 'w') as fw:
        print('data_file {}'.format(getattr(dataset, 'list_files', None)), file=fw)
        print('dim_feature {}'.format(dim_feature), file=fw)
        print('num_tokens {}'.format(num_token), file=fw)
        print('size_dataset {}'.format(count), file=fw)
        print('damaged samples: {}'.format(num_damaged), file=fw)


def save2tfrecord(dataset, dir_save, size_file=5000000):
    _save(dataset, dir_save, ('id', 'feature', 'label'), size_file)


def save2tfrecord_multilabel(dataset, dir_save, size_file=5000000):
    _save(dataset, dir_save, ('id', 'feature', 'label', 'aux_label'), size_file)


def save2tfrecord_ctcbert(dataset, dir_save, size_file=5000000):
    _save(dataset, dir_save, ('id', 'feature', 'label', 'aux_label', 'bert_feature'), size_file)


# ------------------------------------------------------------------------------------------------ readers (:474-672)
def _examples(dir_data, args, _shuffle, fields, epoch_seed=None, verify_crc=True, mode=None):
    """One pass over the files: tuples (id, feature, *fields).  Train mode cuts the raw frames to data.max_train_len
    (:656-657).  The file order is shuffled when `_shuffle` (random.shuffle, :624-625) and sorted otherwise; the
    reference additionally passes `shuffle=shuffle` - the imported FUNCTION, always true - to string_input_producer, so
    its file order is random in every mode; that accident is not reproduced."""
    files = fentch_filelist(dir_data)
    files.sort()
    if _shuffle:
        random.Random(epoch_seed).shuffle(files)
    d = args.data
    cut = d.max_train_len if (mode or args.mode) == 'train' and d.max_train_len else None
    for path in files:
        rf = RecordFile(path, verify_crc=verify_crc)
        for i in range(len(rf)):
            ident = np.array([1], np.int32) if d.withoutid else rf.field(i, 'id', np.int32)
            feature = rf.field(i, 'feature', np.float32).reshape(-1, d.dim_feature)[:cut]
            if feature.shape[0] == 0:       # nothing to align a label to (the reference drops damaged samples when it writes, :353-355)
                continue
            yield (ident, feature) + tuple(rf.field(i, key, dt).reshape(shape) for key, dt, shape in fields)


def readTFRecord(dir_data, args, _shuffle=False, transform=False, **kw):
    return _examples(dir_data, args, _shuffle, [('label', np.int32, (-1,))], **kw)


def readTFRecord_multilabel(dir_data, args, _shuffle=False, transform=False, **kw):
    return _examples(dir_data, args, _shuffle, [('label', np.int32, (-1,)), ('aux_label', np.int32, (-1,))], **kw)


def readTFRecord_ctcbert(dir_data, args, _shuffle=False, transform=False, **kw):
    bert = ('bert_feature', np.int32, (-1,)) if args.model.pt_decoder else ('bert_feature', np.float32, (-1, args.data.bert_feat_size))
    return _examples(dir_data, args, _shuffle, [('label', np.int32, (-1,)), ('aux_label', np.int32, (-1,)), bert], **kw)


# ------------------------------------------------------------------------------------------------ batching (:858-1062)
def bucket_by_sequence_length(examples, input_length, batch_size, bucket_boundaries):
    """tf.contrib.training.bucket_by_sequence_length(dynamic_pad=True, allow_smaller_final_batch=True) for one
    enqueuing thread: example -> bucket `number of boundaries <= length`; a bucket is emitted, in arrival order, when it
    holds batch_size[bucket] examples; what is left at the end of `examples` is emitted bucket by bucket."""
    assert len(batch_size) == len(bucket_boundaries) + 1
    bounds = np.asarray(bucket_boundaries, np.int64)
    buckets = [[] for _ in batch_size]
    for ex in examples:
        b = int(np.searchsorted(bounds, input_length(ex), side='right'))
        buckets[b].append(ex)
        if len(buckets[b]) >= batch_size[b]:
            yield buckets[b]
            buckets[b] = []
    for b in buckets:
        if b:
            yield b


def _pad(arrays, pin):
    """dynamic_pad: zero padding to the longest of the batch.  One copy per example into fresh (optionally pinned)
    memory; only the padding is zeroed.  Pinned blocks come from torch's caching host allocator, which hands a block out
    again only after the H2D copies that read it have finished."""
    import torch
    shape = (len(arrays),) + tuple(max(a.shape[d] for a in arrays) for d in range(arrays[0].ndim))
    out = torch.empty(shape, dtype=_TORCH_DTYPE[arrays[0].dtype.name], pin_memory=pin)
    view = out.numpy()
    for i, a in enumerate(arrays):
        if a.ndim == 0 or a.shape == shape[1:]:
            view[i] = a
            continue
        n = a.shape[0]
        view[i, :n][(slice(None),) + tuple(slice(0, s) for s in a.shape[1:])] = a
        view[i, n:] = 0
        if a.shape[1:] != shape[2:]:      # ragged inner axes (not produced by the readers above): clear and refill
            view[i] = 0
            view[(i,) + tuple(slice(0, s) for s in a.shape)] = a
    return out


class TFReader:
    """:858-1062.  `fentch_*_batch_bucket()` return an iterator of batch dicts with the key names the model's
    `train_step` / `forward_backward` take (`feature` RAW frames + `len_fea` raw lengths: process_raw_feature runs on
    the device on the padded batch; bucketing uses the length AFTER it, as the reference does).  Train mode repeats the
    data for ever with a new file order per epoch (num_epochs=None, :629); any other mode makes one pass."""

    def __init__(self, dir_tfdata, args, is_train=True, seed=0, pin_memory=False, verify_crc=True, mode=None):
        self.dir_tfdata, self.args, self.is_train = dir_tfdata, args, is_train
        self.mode = mode or args.mode          # `mode`: a dev reader inside a training run (args.mode stays 'train')
        self.seed, self.pin, self.verify_crc = seed, pin_memory, verify_crc
        if not args.list_bucket_boundaries:      # arguments.py:77-87
            if args.bucket_boundaries:
                args.list_bucket_boundaries = [int(i) for i in str(args.bucket_boundaries).split(',')]
            else:
                args.list_bucket_boundaries = list(range(args.size_bucket_start, args.size_bucket_end, args.size_bucket_gap))
            n_tow = args.towers_per_batch or args.num_gpus or 1       # arguments.build_args: 1 per process under torchrun
            args.list_batch_size = [int(args.num_batch_tokens / b) * n_tow for b in args.list_bucket_boundaries] + [n_tow]
        self.list_batch_size = args.list_batch_size
        self.list_bucket_boundaries = args.list_bucket_boundaries
        m = args.model
        if m.use_multilabel:
            self.read, self.names = readTFRecord_multilabel, ('label', 'aux_label')
        elif m.use_ctc_bert:
            self.read, self.names = readTFRecord_ctcbert, ('label', 'aux_label', 'bert_feat')
        else:
            self.read, self.names = readTFRecord, ('label',)

    def _epochs(self):
        epoch = 0
        while True:
            yield from self.read(self.dir_tfdata, self.args, _shuffle=self.is_train, transform=True,
                                 epoch_seed=self.seed + epoch, verify_crc=self.verify_crc and epoch == 0, mode=self.mode)
            epoch += 1
            if self.mode != 'train':
                return

    def transformed_len(self, example):
        return -(-example[1].shape[0] // int(self.args.data.downsample or 1))

    def _batches(self):
        for group in bucket_by_sequence_length(self._epochs(), self.transformed_len, self.list_batch_size, self.list_bucket_boundaries):
            batch = {'id': _pad([ex[0] for ex in group], self.pin), 'feature': _pad([ex[1] for ex in group], self.pin),
                     'len_fea': np.array([ex[1].shape[0] for ex in group], np.int32)}
            for j, name in enumerate(self.names):
                batch[name] = _pad([ex[2 + j] for ex in group], self.pin)
                batch['len_' + name] = np.array([ex[2 + j].shape[0] for ex in group], np.int32)
            yield batch

    def __iter__(self):
        return self._batches()

    # the reference's three entry points differ only in the tuple they return; the batch dict carries every field
    fentch_batch_bucket = fentch_multi_label_batch_bucket = fentch_ctcbert_batch_bucket = _batches


class Prefetcher:
    """A bounded queue filled by one host thread (the record views, the padding copies and the CRC check release the
    GIL inside numpy / the native library), in place of the reference's 64 queue-runner threads (:1051)."""

    def __init__(self, iterable, depth=4):
        self.q = queue.Queue(maxsize=depth)
        self.stop = threading.Event()
        self.thread = threading.Thread(target=self._run, args=(iter(iterable),), daemon=True)
        self.thread.start()

    def _run(self, it):
        try:
            for item in it:
                while not self.stop.is_set():
                    try:
                        self.q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        pass
                if self.stop.is_set():
                    return
            self.q.put(StopIteration)
        except BaseException as e:       # handed to the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is StopIteration:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        self.stop.set()
