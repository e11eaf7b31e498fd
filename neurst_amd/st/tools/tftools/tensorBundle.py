"""TensorFlow's checkpoint container ("TensorBundle", what tf.train.Saver writes and st/bin/run_lut.py:140-276,
st/tools/tftools/checkpointTools.py:44-97 and st/tools/avg_checkpoints.py read through pywrap_tensorflow): SURVEY 8(f) rank 1.

  <prefix>.index                 a LevelDB-format table (tensorflow/core/lib/io/table*.cc) mapping
                                   ""            -> BundleHeaderProto {num_shards = 1, endianness = 2 (LITTLE = 0), version = 3}
                                   tensor name   -> BundleEntryProto  {dtype = 1, shape = 2 {dim = 2 {size = 1}}, shard_id = 3,
                                                                       offset = 4, size = 5, crc32c = 6 (fixed32, masked), slices = 7}
  <prefix>.data-SSSSS-of-NNNNN   the tensors' bytes (little endian, C order) at [offset, offset + size)

Table file: data blocks, a meta-index block, an index block, and a 48-byte footer (two varint block handles padded to
40 bytes, magic 0xdb4775248b80fb57).  Every block is followed by a one-byte compression type (0 = none, 1 = snappy) and
the masked CRC-32C of block + type.  Block entries are prefix-compressed: varint32 shared, non_shared, value_len, then the
key suffix and the value; the block ends with its restart offsets (uint32 each) and their count.

parity unpinned: the reference repository holds no checkpoint and TensorFlow is not importable here, so reader and writer
are checked against each other, against the format constants above, and the reader against hand-built index files with
prefix-compressed keys and several blocks (tests/test_host_logic.py).  Partitioned variables (`slices`) and string tensors
are not supported; snappy-compressed blocks are (BundleWriter asks for no compression, other table writers may not)."""
import os
import struct

import numpy as np

from .... import lib as _lib

MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16,
          19: np.float16, 22: np.uint32, 23: np.uint64}
DT_BFLOAT16 = 14
ENUM_OF = {np.dtype(v).name: k for k, v in DTYPES.items()}


def _crc(data):
    c = _lib.load().neurst_crc32c(bytes(data), len(data))
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _get_varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise ValueError('malformed varint')


def _put_varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _snappy_decompress(src):
    """the raw snappy format: varint uncompressed length, then literal / copy elements"""
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 2], 'little')
            pos += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('malformed snappy block')
        for _ in range(ln):              # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy block decodes to %d bytes, header says %d' % (len(out), n))
    return bytes(out)


def _read_block(buf, offset, size, verify=True):
    raw, kind = buf[offset:offset + size], buf[offset + size]
    if len(raw) != size or offset + size + 5 > len(buf):
        raise ValueError('table block runs past the end of the index file')
    if verify and struct.unpack_from('<I', buf, offset + size + 1)[0] != _crc(buf[offset:offset + size + 1]):
        raise ValueError('table block at %d: checksum mismatch' % offset)
    if kind == 1:
        raw = _snappy_decompress(raw)
    elif kind != 0:
        raise ValueError('table block at %d: unknown compression type %d' % (offset, kind))
    return raw


def _block_entries(block):
    (nrestarts,) = struct.unpack_from('<I', block, len(block) - 4)
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """-> list of (key bytes, value bytes) in file order"""
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow index table (bad magic number)' % path)
    footer = buf[len(buf) - 48:]
    _, p = _get_varint(footer, 0)          # meta-index handle (unused)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        out += list(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


def _fields(msg):
    """one level of a protobuf message -> {field: [values]} (varints as int, fixed32 as int, length-delimited as bytes)"""
    out, pos = {}, 0
    while pos < len(msg):
        tag, pos = _get_varint(msg, pos)
        wt, f = tag & 7, tag >> 3
        if wt == 0:
            v, pos = _get_varint(msg, pos)
        elif wt == 2:
            n, pos = _get_varint(msg, pos)
            v = bytes(msg[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', msg, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from('<Q', msg, pos)[0]
            pos += 8
        else:
            raise ValueError('unsupported wire type %d' % wt)
        out.setdefault(f, []).append(v)
    return out


class BundleReader:
    def __init__(self, prefix, verify=True):
        self.prefix, self.verify = prefix, verify
        self.entries, self.num_shards = {}, 1
        for key, value in read_table(prefix + '.index', verify):
            f = _fields(value)
            if key == b'':
                self.num_shards = f.get(1, [1])[0]
                if f.get(2, [0])[0] != 0:
                    raise ValueError('%s: big-endian bundles are not supported' % prefix)
                continue
            if 7 in f:
                raise NotImplementedError('%s: %s is a partitioned variable (slices)' % (prefix, key.decode()))
            dims = []
            for shape in f.get(2, []):
                for dim in _fields(shape).get(2, []):
                    size = _fields(dim).get(1, [0])[0]
                    dims.append(size - (1 << 64) if size >> 63 else size)
            self.entries[key.decode()] = dict(dtype=f.get(1, [0])[0], shape=tuple(dims), shard=f.get(3, [0])[0], offset=f.get(4, [0])[0],
                                              size=f.get(5, [0])[0], crc=f.get(6, [0])[0])
        self._data = {}

    def _shard(self, i):
        if i not in self._data:
            self._data[i] = np.memmap('%s.data-%05d-of-%05d' % (self.prefix, i, self.num_shards), dtype=np.uint8, mode='r')
        return self._data[i]

    def has_tensor(self, name):
        return name in self.entries

    def get_variable_to_shape_map(self):
        return {k: list(e['shape']) for k, e in self.entries.items()}

    def get_tensor(self, name):
        e = self.entries[name]
        raw = self._shard(e['shard'])[e['offset']:e['offset'] + e['size']]
        if raw.size != e['size']:
            raise ValueError('%s: tensor %s runs past the end of its data file' % (self.prefix, name))
        if self.verify and _crc(raw.tobytes()) != e['crc']:
            raise ValueError('%s: tensor %s: checksum mismatch' % (self.prefix, name))
        if e['dtype'] == DT_BFLOAT16:      # widened to fp32: numpy has no bfloat16
            return (np.frombuffer(raw.tobytes(), np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(e['shape'])
        if e['dtype'] not in DTYPES:
            raise NotImplementedError('%s: tensor %s has unsupported dtype enum %d' % (self.prefix, name, e['dtype']))
        return np.frombuffer(raw.tobytes(), DTYPES[e['dtype']]).reshape(e['shape']).copy()


# ------------------------------------------------------------------------------------------------------------ writer
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.interval = bytearray(), [0], 0, b'', restart_interval

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def write_table(path, items, block_size=4096):
    """items: (key bytes, value bytes) pairs in ascending key order"""
    out = bytearray()

    def emit(block):
        handle = _put_varint(len(out)) + _put_varint(len(block))
        trailer = block + b'\x00'
        out.extend(trailer + struct.pack('<I', _crc(trailer)))
        return handle
    index, builder = _BlockBuilder(restart_interval=1), _BlockBuilder()
    for key, value in items:
        assert builder.count == 0 or key > builder.last, 'keys must ascend'
        builder.add(key, value)
        if builder.size() >= block_size:
            index.add(builder.last, emit(builder.finish()))
            builder = _BlockBuilder()
    if builder.count:
        index.add(builder.last, emit(builder.finish()))
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    footer = meta + idx
    out.extend(footer + bytes(40 - len(footer)) + struct.pack('<Q', MAGIC))
    with open(path, 'wb') as f:
        f.write(bytes(out))


def _pb_varint(field, v):
    return _put_varint(field << 3) + _put_varint(v & 0xFFFFFFFFFFFFFFFF)


def _pb_bytes(field, payload):
    return _put_varint((field << 3) | 2) + _put_varint(len(payload)) + payload


def write_bundle(prefix, tensors):
    """tensors: {name: numpy array}; one data shard"""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or '.', exist_ok=True)
    header = _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))           # num_shards = 1, (endianness LITTLE = default), version.producer = 1
    items, offset = [(b'', header)], 0
    with open(prefix + '.data-00000-of-00001', 'wb') as data:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])           # (ascontiguousarray would turn a scalar into shape [1])
            if a.dtype.name not in ENUM_OF:
                raise NotImplementedError('dtype %s of %s' % (a.dtype, name))
            raw = a.tobytes(order='C')
            shape = b''.join(_pb_bytes(2, _pb_varint(1, d)) for d in a.shape)
            entry = _pb_varint(1, ENUM_OF[a.dtype.name]) + _pb_bytes(2, shape)
            if offset:
                entry += _pb_varint(4, offset)
            entry += _pb_varint(5, len(raw)) + _put_varint((6 << 3) | 5) + struct.pack('<I', _crc(raw))
            items.append((name.encode(), entry))
            data.write(raw)
            offset += len(raw)
    write_table(prefix + '.index', items)
