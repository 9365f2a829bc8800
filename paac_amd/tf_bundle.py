"""Weights-only reader / writer for the reference's checkpoint container: TensorFlow's V2 "tensor bundle"
(`<prefix>.index` + `<prefix>.data-00000-of-00001`), what tf.train.Saver.save / restore produce and consume in
actor_learner.py:79-82,89-93,112-115 and networks.py:122-135 (pretrained/*/checkpoints/-80000000.index is one).

  .index                 a LevelDB-style sorted string table: prefix-compressed data blocks (restart every 16 keys), an
                         empty metaindex block, an index block (one separator key + block handle per data block) and a
                         48-byte footer; every block is followed by a 1-byte compression type (0 = none) and the masked
                         CRC-32C of block + type.  Key "" holds the BundleHeaderProto, every other key is a variable name
                         and its value a BundleEntryProto {1 dtype, 2 shape, 3 shard_id, 4 offset, 5 size, 6 crc32c}.
  .data-00000-of-00001   the tensors' little-endian bytes back to back, in key order, no padding.

TensorFlow itself is not needed (and not installed): the protobuf wire format and the table format are written and
parsed directly; float32 tensors only (all the reference has).  Nothing in a file is ever executed.  The upstream
repository ships no `.data` blob (.MISSING_LARGE_BLOBS), so VALUES read from a reference-written bundle are "parity
unpinned"; the container itself is pinned by the reference's own `.index` (tests/test_tf_bundle.py: entry set, header,
block layout and CRCs of pretrained/breakout/checkpoints/-80000000.index).
"""
import os
import struct

import numpy as np

DT_FLOAT = 1
TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE = 4096
RESTART_INTERVAL = 16
DATA_SUFFIX = ".data-00000-of-00001"

# ---- CRC-32C (Castagnoli), the checksum of both the table blocks and the tensors ---------------------------------------
_POLY = 0x82F63B78


def _make_table():
    t = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (_POLY if c & 1 else 0)
        t[i] = c
    return t


_TABLE = _make_table()
_TABLE_LIST = [int(v) for v in _TABLE]


def _crc_raw_small(data, crc):
    for b in data:
        crc = _TABLE_LIST[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc


def _zeros_operator(nbytes):
    """32x32 GF(2) matrix (as 32 column words) that advances a raw CRC register over `nbytes` zero bytes."""
    def times(mat, vec):
        out, i = 0, 0
        while vec:
            if vec & 1:
                out ^= mat[i]
            vec >>= 1
            i += 1
        return out

    def square(mat):
        return [times(mat, mat[i]) for i in range(32)]

    one_bit = [_POLY] + [1 << (i - 1) for i in range(1, 32)]     # one zero BIT
    op = square(square(square(one_bit)))                          # one zero byte
    result = [1 << i for i in range(32)]                          # identity
    n = nbytes
    while n:
        if n & 1:
            result = [times(op, result[i]) for i in range(32)]
        op = square(op)
        n >>= 1
    return result, times


def crc32c(data):
    """CRC-32C of a bytes-like object.  Large buffers are cut into equal chunks whose registers advance in lockstep
    (one vectorised table lookup per byte position), then the chunk CRCs are chained with the zero-advance operator."""
    buf = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8)
    n = len(buf)
    if n < (1 << 14):
        return _crc_raw_small(buf.tolist(), 0xFFFFFFFF) ^ 0xFFFFFFFF
    chunk = 2048
    nchunks = n // chunk
    body = buf[:nchunks * chunk].reshape(nchunks, chunk)
    reg = np.zeros(nchunks, dtype=np.uint32)          # raw registers with a ZERO initial value (linear part only)
    for i in range(chunk):
        reg = _TABLE[(reg ^ body[:, i]) & np.uint32(0xFF)] ^ (reg >> np.uint32(8))
    op, times = _zeros_operator(chunk)
    crc = 0xFFFFFFFF                                   # the real initial value travels through the chain
    for r in reg.tolist():
        crc = times(op, crc) ^ r
    crc = _crc_raw_small(buf[nchunks * chunk:].tolist(), crc)
    return crc ^ 0xFFFFFFFF


def masked_crc32c(data):
    crc = crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---- protobuf wire format (the four messages the bundle needs) ------------------------------------------------------------
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    out, pos = [], 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("tensor bundle: unsupported protobuf wire type %d" % wt)
        out.append((num, v))
    return out


def _shape_proto(shape):
    return b"".join(b"\x12" + _varint(len(d)) + d for d in (b"\x08" + _varint(int(s)) for s in shape))


def _entry_proto(shape, offset, size, crc):
    out = b"\x08" + _varint(DT_FLOAT)
    sp = _shape_proto(shape)
    out += b"\x12" + _varint(len(sp)) + sp
    if offset:
        out += b"\x20" + _varint(offset)
    out += b"\x28" + _varint(size)
    out += b"\x35" + struct.pack("<I", crc)
    return out


_HEADER_PROTO = b"\x08\x01" + b"\x1a\x02\x08\x01"      # num_shards = 1, (little endian,) version { producer = 1 }


# ---- sorted string table ------------------------------------------------------------------------------------------------------
class _BlockBuilder(object):
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _separator(start, limit):
    """Shortest key k with start <= k < limit (LevelDB's BytewiseComparator::FindShortestSeparator)."""
    m = min(len(start), len(limit))
    i = 0
    while i < m and start[i] == limit[i]:
        i += 1
    if i < m and start[i] < 0xFF and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _successor(key):
    """Short key >= key (FindShortSuccessor): the first byte that is not 0xff incremented, the rest dropped."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


def _table_bytes(items):
    """items: sorted list of (key bytes, value bytes) -> the table file's bytes."""
    out = bytearray()
    index = _BlockBuilder()

    def emit(block_bytes):
        handle = _varint(len(out)) + _varint(len(block_bytes))
        out.extend(block_bytes + b"\x00" + struct.pack("<I", masked_crc32c(block_bytes + b"\x00")))
        return handle

    block, pending = _BlockBuilder(), None            # pending: (last key of the finished block, its handle)
    for key, value in items:
        if pending is not None:
            index.add(_separator(pending[0], key), pending[1])
            pending = None
        block.add(key, value)
        if block.size() >= BLOCK_SIZE:
            pending = (block.last, emit(block.finish()))
            block = _BlockBuilder()
    if block.count:
        pending = (block.last, emit(block.finish()))
    if pending is not None:
        index.add(_successor(pending[0]), pending[1])
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    return bytes(out)


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def _read_table(data, verify=True):
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("not a tensor-bundle index (bad table magic)")
    footer = data[-48:]
    pos = 0
    _, pos = _read_varint(footer, pos)
    _, pos = _read_varint(footer, pos)
    idx_off, pos = _read_varint(footer, pos)
    idx_len, pos = _read_varint(footer, pos)

    def block(off, ln):
        if off + ln + 5 > len(data):
            raise ValueError("tensor-bundle index is truncated")
        if data[off + ln] != 0:
            raise ValueError("compressed table block (type %d): not supported" % data[off + ln])
        if verify and struct.unpack_from("<I", data, off + ln + 1)[0] != masked_crc32c(data[off:off + ln + 1]):
            raise ValueError("tensor-bundle index block fails its CRC-32C")
        return data[off:off + ln]

    out = []
    for _, handle in _block_entries(block(idx_off, idx_len)):
        off, p = _read_varint(handle, 0)
        ln, p = _read_varint(handle, p)
        out.extend(_block_entries(block(off, ln)))
    return out


# ---- the bundle -----------------------------------------------------------------------------------------------------------------
def write(prefix, arrays):
    """arrays: {variable name: array} -> `<prefix>.index` and `<prefix>.data-00000-of-00001` (float32, key order).
    The data file is complete (flushed) before the index appears, and the index is renamed into place: a reader that
    finds an index finds a whole checkpoint."""
    items, offset = [(b"", _HEADER_PROTO)], 0
    tmp_data = prefix + DATA_SUFFIX + ".tmp-%d" % os.getpid()
    with open(tmp_data, "wb") as f:
        for name in sorted(arrays, key=lambda s: s.encode()):
            a = np.ascontiguousarray(np.asarray(arrays[name], dtype="<f4"))
            raw = a.tobytes()
            f.write(raw)
            items.append((name.encode(), _entry_proto(a.shape, offset, len(raw), masked_crc32c(raw))))
            offset += len(raw)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp_data, prefix + DATA_SUFFIX)
    tmp_index = prefix + ".index.tmp-%d" % os.getpid()
    with open(tmp_index, "wb") as f:
        f.write(_table_bytes(items))
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp_index, prefix + ".index")
    return prefix + ".index"


def entries(prefix):
    """-> {variable name: dict(shape, offset, size, crc32c, dtype, shard)} of `<prefix>.index` (the header entry checked)."""
    with open(prefix + ".index", "rb") as f:
        table = _read_table(f.read())
    out = {}
    for key, value in table:
        fs = _fields(value)
        if key == b"":
            if dict(fs).get(1, 1) != 1:
                raise ValueError("tensor bundle with %d shards: only single-shard bundles are supported" % dict(fs)[1])
            if dict(fs).get(2, 0) != 0:
                raise ValueError("big-endian tensor bundle: not supported")
            continue
        d = dict(fs)
        shape = tuple(dict(_fields(dim)).get(1, 0) for num, dim in _fields(d.get(2, b"")) if num == 2)
        out[key.decode()] = dict(dtype=d.get(1, 0), shape=shape, shard=d.get(3, 0), offset=d.get(4, 0), size=d.get(5, 0),
                                 crc32c=d.get(6, 0))
    return out


def read(prefix, verify=True):
    """-> {variable name: float32 array}.  Every tensor's extent and (verify) CRC-32C are checked."""
    meta = entries(prefix)
    out = {}
    with open(prefix + DATA_SUFFIX, "rb") as f:
        blob = f.read()
    for name, e in meta.items():
        if e["dtype"] != DT_FLOAT:
            raise ValueError("%s: dtype %d, only float32 (1) is supported" % (name, e["dtype"]))
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if e["size"] != 4 * n or e["offset"] + e["size"] > len(blob):
            raise ValueError("%s: entry of %d bytes at %d does not fit shape %s / the %d-byte data file"
                             % (name, e["size"], e["offset"], e["shape"], len(blob)))
        raw = blob[e["offset"]:e["offset"] + e["size"]]
        if verify and masked_crc32c(raw) != e["crc32c"]:
            raise ValueError("%s fails its CRC-32C" % name)
        out[name] = np.frombuffer(raw, dtype="<f4").reshape(e["shape"]).astype(np.float32)
    return out


def readable(prefix):
    """True when `<prefix>.index` parses and the data file holds every tensor it lists (no CRC pass over the data)."""
    try:
        meta = entries(prefix)
        size = os.path.getsize(prefix + DATA_SUFFIX)
        return len(meta) > 0 and all(e["offset"] + e["size"] <= size for e in meta.values())
    except (OSError, ValueError, IndexError, struct.error):
        return False


def write_state_file(folder, prefix_name, all_prefix_names=None):
    """The `checkpoint` text file tf.train.latest_checkpoint reads (networks.py:125): the newest prefix, then every prefix
    the saver still keeps (oldest first), one `all_model_checkpoint_paths` line each like tf.train.Saver writes them."""
    tmp = os.path.join(folder, ".checkpoint.tmp-%d" % os.getpid())
    with open(tmp, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % prefix_name)
        for name in (all_prefix_names or [prefix_name]):
            f.write('all_model_checkpoint_paths: "%s"\n' % name)
    os.replace(tmp, os.path.join(folder, "checkpoint"))
