"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the input pipeline either side of the hot path
 - the TFRecord container, the tf.train.Example payload and the length-bucketed batching that
st/tools/tftools/tfRecord.py gets from TensorFlow.  Written with DIFFERENT means than the product so that the two can
check each other: a bitwise CRC-32C, `struct` framing, and google.protobuf's own encoder/decoder over a run-time
descriptor of tensorflow/core/example/{example,feature}.proto (message and field numbers as published there).

parity unpinned against TensorFlow itself: the reference holds no `*.recode` file and TF is not importable here.  What IS
pinned: CRC-32C by the RFC 3720 B.4 known answers (tests/test_records.py), the wire format by protobuf's implementation.
The batching restates the documented behaviour of tf.contrib.training.bucket_by_sequence_length for ONE enqueuing
thread (the reference uses 64 threads, whose interleaving is not deterministic): bucket i takes lengths in
[boundaries[i-1], boundaries[i]), a bucket is emitted in arrival order as soon as it holds batch_size[i] examples, the
remainders are emitted at the end of the input (allow_smaller_final_batch), tensors are zero-padded to the longest of
the batch (dynamic_pad)."""
import struct

import numpy as np

_POLY = 0x82F63B78     # reflected Castagnoli polynomial


def crc32c(data):
    crc = 0xFFFFFFFF
    for b in bytes(data):
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (_POLY if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def frame(payload):
    """one record of the TFRecord container (tensorflow/core/lib/io/record_writer.cc)"""
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', masked_crc(head)) + payload + struct.pack('<I', masked_crc(payload))


def unframe(blob):
    out, pos = [], 0
    while pos < len(blob):
        (n,) = struct.unpack_from('<Q', blob, pos)
        assert struct.unpack_from('<I', blob, pos + 8)[0] == masked_crc(blob[pos:pos + 8]), 'length crc'
        payload = blob[pos + 12:pos + 12 + n]
        assert len(payload) == n, 'truncated'
        assert struct.unpack_from('<I', blob, pos + 12 + n)[0] == masked_crc(payload), 'data crc'
        out.append(payload)
        pos += 16 + n
    return out


_EXAMPLE = None


def example_class():
    global _EXAMPLE
    if _EXAMPLE is None:
        from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
        fd = descriptor_pb2.FileDescriptorProto(name='oracle_example.proto', package='tensorflow', syntax='proto3')
        F = descriptor_pb2.FieldDescriptorProto
        m = fd.message_type.add(name='BytesList')
        m.field.add(name='value', number=1, type=F.TYPE_BYTES, label=F.LABEL_REPEATED)
        m = fd.message_type.add(name='Feature')
        m.field.add(name='bytes_list', number=1, type=F.TYPE_MESSAGE, type_name='.tensorflow.BytesList', label=F.LABEL_OPTIONAL)
        m = fd.message_type.add(name='Features')
        e = m.nested_type.add(name='FeatureEntry')
        e.options.map_entry = True
        e.field.add(name='key', number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
        e.field.add(name='value', number=2, type=F.TYPE_MESSAGE, type_name='.tensorflow.Feature', label=F.LABEL_OPTIONAL)
        m.field.add(name='feature', number=1, type=F.TYPE_MESSAGE, type_name='.tensorflow.Features.FeatureEntry', label=F.LABEL_REPEATED)
        m = fd.message_type.add(name='Example')
        m.field.add(name='features', number=1, type=F.TYPE_MESSAGE, type_name='.tensorflow.Features', label=F.LABEL_OPTIONAL)
        pool = descriptor_pool.DescriptorPool()
        pool.Add(fd)
        _EXAMPLE = message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.Example'))
    return _EXAMPLE


def encode_example(sample):
    """sample: {key: numpy array}; one bytes feature per key holding array.tobytes() (tfRecord.py:362-370)"""
    ex = example_class()()
    for k, v in sample.items():
        ex.features.feature[k].bytes_list.value.append(np.ascontiguousarray(v).tobytes())
    return ex.SerializeToString()


def decode_example(payload):
    ex = example_class()()
    ex.ParseFromString(bytes(payload))
    return {k: bytes(v.bytes_list.value[0]) for k, v in ex.features.feature.items()}


def transformed_len(raw_len, downsample, max_train_len=None):
    """frames after process_raw_feature (tfRecord.py:827-845): splice keeps the length, down_sample takes every
    `downsample`-th frame from 0; train mode cuts the raw frames to max_train_len first (:656-657)"""
    n = raw_len if max_train_len is None else min(raw_len, max_train_len)
    return -(-n // downsample)


def bucket_batches(lengths, boundaries, batch_sizes):
    """-> list of lists of example indices, in emission order (see the module docstring)"""
    assert len(batch_sizes) == len(boundaries) + 1
    buckets = [[] for _ in batch_sizes]
    out = []
    for i, n in enumerate(lengths):
        b = sum(1 for edge in boundaries if n >= edge)
        buckets[b].append(i)
        if len(buckets[b]) == batch_sizes[b]:
            out.append(buckets[b])
            buckets[b] = []
    for b in buckets:
        if b:
            out.append(b)
    return out


def pad_batch(arrays):
    """dynamic_pad: zero-pad along every axis to the longest of the batch"""
    shape = tuple(max(a.shape[d] for a in arrays) for d in range(arrays[0].ndim))
    out = np.zeros((len(arrays),) + shape, arrays[0].dtype)
    for i, a in enumerate(arrays):
        out[(i,) + tuple(slice(0, s) for s in a.shape)] = a
    return out
