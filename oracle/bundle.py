"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): an independent restatement of TensorFlow's checkpoint container
("TensorBundle": `<prefix>.index` + `<prefix>.data-SSSSS-of-NNNNN`), the format behind tf.train.Saver that the reference
reads and writes through TensorFlow in st/tools/tftools/checkpointTools.py:80-97, st/tools/avg_checkpoints.py:46-117 and
st/bin/run_lut.py:140-276.  The arithmetic lives in a dependency that is absent here (tensorflow,
unpinned in requirements.txt, "Tensorflow 1.15" per README.md:20); this file restates its published algorithms:

  tensorflow/core/lib/io/table_builder.cc   TableBuilder::Add / Flush / WriteBlock / Finish, FindShortestSeparator /
                                            FindShortSuccessor for the index keys, block_restart_interval 16 (index block 1),
                                            block_size 262144 (table_options.h), the 48-byte footer (format.cc)
  tensorflow/core/lib/io/block_builder.cc   prefix-compressed entries + restart array
  tensorflow/core/lib/io/block.cc           Block::Iter::Seek - binary search over the restart points, then a linear scan
  tensorflow/core/lib/io/table.cc           Table::InternalGet - seek in the index block, then in the data block it names
  tensorflow/core/util/tensor_bundle/tensor_bundle.cc   BundleWriter::Add / Finish, BundleReader::GetValue
  tensorflow/core/protobuf/tensor_bundle.proto, framework/tensor_shape.proto, framework/versions.proto   (field numbers)
  the raw snappy format (format_description.txt): varint length, literal / copy elements

Written with OTHER means than neurst_amd/st/tools/tftools/tensorBundle.py so that the two can check each other
(tests/test_host_logic.py): messages are encoded and decoded by google.protobuf on run-time descriptors (the product packs
and unpacks the wire format by hand), the CRC-32C is table-driven from the polynomial (the product calls the SSE4.2
instruction in libneurst_hip.so), lookups go through the index block and the restart arrays as LevelDB's Seek does (the
product scans every block front to back), the writer emits shortest-separator index keys, arbitrary block sizes / restart
intervals and snappy-compressed blocks (the product writes neither).

parity unpinned against TensorFlow itself: the reference holds no checkpoint and TF is not importable here.  Pinned: the
CRC by the RFC 3720 B.4 vectors (tests/test_records.py uses the same polynomial), the protobuf wire format by protobuf's
own implementation, the snappy codec by the format description's worked elements (tests)."""
import struct

import numpy as np

MAGIC = 0xdb4775248b80fb57
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)

# tensorflow/core/framework/types.proto
DT = {'float32': 1, 'float64': 2, 'int32': 3, 'uint8': 4, 'int16': 5, 'int8': 6, 'int64': 9, 'bool': 10, 'bfloat16': 14, 'uint16': 17,
      'float16': 19, 'uint32': 22, 'uint64': 23}
NP_OF = {v: k for k, v in DT.items()}


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask(c):
    """crc32c::Mask (tensorflow/core/lib/hash/crc32c.h): rotate right by 15, add a constant"""
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# --------------------------------------------------------------------------------------------------------- protobuf side
_MSG = None


def messages():
    """-> (BundleHeaderProto, BundleEntryProto) classes built from run-time descriptors"""
    global _MSG
    if _MSG is None:
        from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
        F = descriptor_pb2.FieldDescriptorProto
        fd = descriptor_pb2.FileDescriptorProto(name='oracle_tensor_bundle.proto', package='tensorflow', syntax='proto3')
        m = fd.message_type.add(name='VersionDef')
        m.field.add(name='producer', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
        m.field.add(name='min_consumer', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
        m.field.add(name='bad_consumers', number=3, type=F.TYPE_INT32, label=F.LABEL_REPEATED)
        m = fd.message_type.add(name='TensorShapeProto')
        d = m.nested_type.add(name='Dim')
        d.field.add(name='size', number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
        d.field.add(name='name', number=2, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
        m.field.add(name='dim', number=2, type=F.TYPE_MESSAGE, type_name='.tensorflow.TensorShapeProto.Dim', label=F.LABEL_REPEATED)
        m.field.add(name='unknown_rank', number=3, type=F.TYPE_BOOL, label=F.LABEL_OPTIONAL)
        m = fd.message_type.add(name='TensorSliceProto')
        e = m.nested_type.add(name='Extent')
        e.field.add(name='start', number=1, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
        e.field.add(name='length', number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
        m.field.add(name='extent', number=1, type=F.TYPE_MESSAGE, type_name='.tensorflow.TensorSliceProto.Extent', label=F.LABEL_REPEATED)
        m = fd.message_type.add(name='BundleHeaderProto')
        m.field.add(name='num_shards', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
        m.field.add(name='endianness', number=2, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)      # enum Endianness {LITTLE = 0; BIG = 1}
        m.field.add(name='version', number=3, type=F.TYPE_MESSAGE, type_name='.tensorflow.VersionDef', label=F.LABEL_OPTIONAL)
        m = fd.message_type.add(name='BundleEntryProto')
        m.field.add(name='dtype', number=1, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)           # enum DataType
        m.field.add(name='shape', number=2, type=F.TYPE_MESSAGE, type_name='.tensorflow.TensorShapeProto', label=F.LABEL_OPTIONAL)
        m.field.add(name='shard_id', number=3, type=F.TYPE_INT32, label=F.LABEL_OPTIONAL)
        m.field.add(name='offset', number=4, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
        m.field.add(name='size', number=5, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
        m.field.add(name='crc32c', number=6, type=F.TYPE_FIXED32, label=F.LABEL_OPTIONAL)
        m.field.add(name='slices', number=7, type=F.TYPE_MESSAGE, type_name='.tensorflow.TensorSliceProto', label=F.LABEL_REPEATED)
        pool = descriptor_pool.DescriptorPool()
        pool.Add(fd)
        _MSG = tuple(message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.' + n))
                     for n in ('BundleHeaderProto', 'BundleEntryProto'))
    return _MSG


# ------------------------------------------------------------------------------------------------------------- varints
def varint(n):
    out = b''
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def unvarint(buf, pos):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if b < 0x80:
            return n, pos


# --------------------------------------------------------------------------------------------------------------- snappy
def snappy_compress(data, min_match=4):
    """A valid raw-snappy stream: greedy matching against a hash of 4-byte windows; copies with 2-byte offsets (element
    type 2, lengths 1..64) and 1-byte offsets where they fit (type 1: lengths 4..11, offsets < 2048); the rest as literals."""
    data = bytes(data)
    out = bytearray(varint(len(data)))
    seen = {}
    lit_start = i = 0

    def flush_literal(end):
        nonlocal lit_start
        while lit_start < end:
            n = min(end - lit_start, 65536)
            if n <= 60:
                out.append((n - 1) << 2)
            elif n <= 256:
                out.extend([60 << 2, n - 1])
            else:
                out.append(61 << 2)
                out.extend(struct.pack('<H', n - 1))
            out.extend(data[lit_start:lit_start + n])
            lit_start += n
    while i + min_match <= len(data):
        key = data[i:i + 4]
        j = seen.get(key)
        seen[key] = i
        if j is not None and i - j <= 65535:
            ln = 4
            while i + ln < len(data) and ln < 64 and data[j + ln] == data[i + ln]:
                ln += 1
            flush_literal(i)
            off = i - j
            if 4 <= ln <= 11 and off < 2048:
                out.extend([((off >> 8) << 5) | ((ln - 4) << 2) | 1, off & 0xFF])
            else:
                out.append(((ln - 1) << 2) | 2)
                out.extend(struct.pack('<H', off))
            i += ln
            lit_start = i
        else:
            i += 1
    flush_literal(len(data))
    return bytes(out)


def snappy_decompress(src):
    src = bytes(src)
    n, pos = unvarint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        t = tag & 3
        if t == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(src[pos:pos + nb], 'little') + 1
                pos += nb
            out += src[pos:pos + ln]
            pos += ln
        else:
            if t == 1:
                ln, off = 4 + ((tag >> 2) & 7), ((tag >> 5) << 8) | src[pos]
                pos += 1
            elif t == 2:
                ln, off = (tag >> 2) + 1, struct.unpack_from('<H', src, pos)[0]
                pos += 2
            else:
                ln, off = (tag >> 2) + 1, struct.unpack_from('<I', src, pos)[0]
                pos += 4
            assert 0 < off <= len(out), 'copy reaches before the start of the output'
            start = len(out) - off
            for k in range(ln):
                out.append(out[start + k])
    assert len(out) == n, 'length header disagrees with the elements'
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------- table writer
class BlockBuilder:
    """block_builder.cc: entries `varint shared | varint non_shared | varint value_len | key suffix | value`, a restart
    point (shared = 0) every `interval` entries, then the restart offsets and their count as fixed32."""

    def __init__(self, interval):
        self.interval, self.body, self.restarts, self.counter, self.last_key = interval, bytearray(), [0], 0, b''

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            lim = min(len(self.last_key), len(key))
            while shared < lim and self.last_key[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.body))
            self.counter = 0
        self.body += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.counter += 1

    def estimate(self):
        return len(self.body) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.body

    def finish(self):
        return bytes(self.body) + struct.pack('<%dI' % len(self.restarts), *self.restarts) + struct.pack('<I', len(self.restarts))


def shortest_separator(start, limit):
    """BytewiseComparator::FindShortestSeparator: a short key k with start <= k < limit"""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d >= n:
        return start                      # one is a prefix of the other
    if start[d] < 0xFF and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def short_successor(key):
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def write_table(path, items, block_size=262144, restart_interval=16, compress=False):
    """TableBuilder: items = ascending (key, value) pairs.  compress=True stores a block snappy-compressed when that saves
    at least 12.5 % (table_builder.cc WriteBlock), else raw."""
    out = bytearray()

    def write_block(raw):
        kind = 0
        if compress:
            z = snappy_compress(raw)
            if len(z) < len(raw) - len(raw) // 8:
                raw, kind = z, 1
        handle = varint(len(out)) + varint(len(raw))
        out.extend(raw)
        out.append(kind)
        out.extend(struct.pack('<I', mask(crc32c(bytes([kind]), crc32c(raw)))))
        return handle
    data, index = BlockBuilder(restart_interval), BlockBuilder(1)
    last_key, pending, first = b'', None, True
    for key, value in items:
        assert first or key > last_key, 'keys must ascend'
        first = False
        if pending is not None:
            index.add(shortest_separator(last_key, key), pending)
            pending = None
        last_key = key
        data.add(key, value)
        if data.estimate() >= block_size:
            pending = write_block(data.finish())
            data = BlockBuilder(restart_interval)
    if not data.empty():
        pending = write_block(data.finish())
    meta_handle = write_block(BlockBuilder(restart_interval).finish())
    if pending is not None:
        index.add(short_successor(last_key), pending)
    index_handle = write_block(index.finish())
    footer = meta_handle + index_handle
    out.extend(footer + bytes(40 - len(footer)) + struct.pack('<II', MAGIC & 0xFFFFFFFF, MAGIC >> 32))
    with open(path, 'wb') as f:
        f.write(bytes(out))


def to_bytes(a):
    """-> (dtype enum, shape, little-endian C-order bytes); a (values, 'bfloat16') pair stores the upper halves of fp32"""
    if isinstance(a, tuple) and a[1] == 'bfloat16':
        v = np.asarray(a[0], np.float32)
        return DT['bfloat16'], v.shape, (v.view(np.uint32) >> 16).astype('<u2').tobytes()
    a = np.asarray(a)
    return DT[a.dtype.name], a.shape, a.astype(a.dtype.newbyteorder('<')).tobytes()


def write_bundle(prefix, tensors, **table_options):
    """BundleWriter (one shard): header under the empty key, then one BundleEntryProto per tensor in key order."""
    Header, Entry = messages()
    h = Header(num_shards=1, endianness=0)
    h.version.producer = 1                 # kTensorBundleVersion
    items, offset = [(b'', h.SerializeToString())], 0
    with open(prefix + '.data-00000-of-00001', 'wb') as data:
        for name in sorted(tensors, key=lambda s: s.encode()):
            dt, shape, raw = to_bytes(tensors[name])
            e = Entry(dtype=dt, shard_id=0, offset=offset, size=len(raw), crc32c=mask(crc32c(raw)))
            e.shape.SetInParent()
            for d in shape:
                e.shape.dim.add(size=int(d))
            items.append((name.encode(), e.SerializeToString()))
            data.write(raw)
            offset += len(raw)
    write_table(prefix + '.index', items, **table_options)


# ---------------------------------------------------------------------------------------------------------- table reader
class Block:
    def __init__(self, raw):
        self.raw = raw
        (self.nrestarts,) = struct.unpack_from('<I', raw, len(raw) - 4)
        self.restart_off = len(raw) - 4 - 4 * self.nrestarts
        assert self.nrestarts >= 1 and self.restart_off >= 0, 'bad restart array'

    def restart(self, i):
        return struct.unpack_from('<I', self.raw, self.restart_off + 4 * i)[0]

    def entry(self, pos, prev_key):
        shared, pos = unvarint(self.raw, pos)
        non_shared, pos = unvarint(self.raw, pos)
        vlen, pos = unvarint(self.raw, pos)
        assert shared <= len(prev_key), 'shared prefix longer than the previous key'
        key = prev_key[:shared] + self.raw[pos:pos + non_shared]
        pos += non_shared
        return key, self.raw[pos:pos + vlen], pos + vlen

    def seek(self, target):
        """Block::Iter::Seek: -> (key, value) of the first entry with key >= target, or None"""
        lo, hi = 0, self.nrestarts - 1
        while lo < hi:                                   # last restart point whose key is < target
            mid = (lo + hi + 1) // 2
            key, _, _ = self.entry(self.restart(mid), b'')   # a restart entry shares nothing
            if key < target:
                lo = mid
            else:
                hi = mid - 1
        pos, key = self.restart(lo), b''
        while pos < self.restart_off:
            key, value, pos = self.entry(pos, key)
            if key >= target:
                return key, value
        return None

    def __iter__(self):
        pos, key = 0, b''
        while pos < self.restart_off:
            key, value, pos = self.entry(pos, key)
            yield key, value


class Table:
    def __init__(self, path):
        self.buf = open(path, 'rb').read()
        assert len(self.buf) >= 48, 'file shorter than a footer'
        lo, hi = struct.unpack_from('<II', self.buf, len(self.buf) - 8)
        assert (hi << 32) | lo == MAGIC, 'bad magic number'
        footer = self.buf[-48:]
        _, p = unvarint(footer, 0)
        _, p = unvarint(footer, p)
        ioff, p = unvarint(footer, p)
        isize, p = unvarint(footer, p)
        self.index = self.block(ioff, isize)

    def block(self, off, size):
        raw, kind = self.buf[off:off + size], self.buf[off + size]
        (stored,) = struct.unpack_from('<I', self.buf, off + size + 1)
        assert stored == mask(crc32c(bytes([kind]), crc32c(raw))), 'block checksum mismatch at %d' % off
        if kind == 1:
            raw = snappy_decompress(raw)
        else:
            assert kind == 0, 'unknown block type %d' % kind
        return Block(raw)

    def handle_block(self, handle):
        off, p = unvarint(handle, 0)
        size, _ = unvarint(handle, p)
        return self.block(off, size)

    def get(self, key):
        """Table::InternalGet"""
        hit = self.index.seek(key)
        if hit is None:
            return None
        found = self.handle_block(hit[1]).seek(key)
        return found[1] if found is not None and found[0] == key else None

    def items(self):
        for _, handle in self.index:
            yield from self.handle_block(handle)

    def num_blocks(self):
        return sum(1 for _ in self.index)


class Reader:
    """BundleReader: Lookup(name) through the table's Get, value checked against its crc32c"""

    def __init__(self, prefix):
        self.prefix, self.table = prefix, Table(prefix + '.index')
        Header, self.Entry = messages()
        raw = self.table.get(b'')
        assert raw is not None, 'no header entry'
        self.header = Header()
        self.header.ParseFromString(raw)
        assert self.header.endianness == 0, 'big-endian bundle'

    def names(self):
        return [k.decode() for k, _ in self.table.items() if k != b'']

    def entry(self, name):
        raw = self.table.get(name.encode())
        if raw is None:
            raise KeyError(name)
        e = self.Entry()
        e.ParseFromString(raw)
        return e

    def tensor(self, name):
        e = self.entry(name)
        assert not e.slices, 'partitioned variable'
        with open('%s.data-%05d-of-%05d' % (self.prefix, e.shard_id, self.header.num_shards), 'rb') as f:
            f.seek(e.offset)
            raw = f.read(e.size)
        assert len(raw) == e.size, 'data file too short'
        assert mask(crc32c(raw)) == e.crc32c, 'tensor checksum mismatch'
        shape = tuple(d.size for d in e.shape.dim)
        kind = NP_OF[e.dtype]
        if kind == 'bfloat16':
            return (np.frombuffer(raw, '<u2').astype(np.uint32) << 16).view(np.float32).reshape(shape)
        return np.frombuffer(raw, np.dtype(kind).newbyteorder('<')).reshape(shape).astype(kind)
