"""Test-side readers for the two TensorFlow artefact formats the reference ships (pretrained/*/checkpoints/):
  * `.index` -- a LevelDB-style sorted string table (one prefix-compressed data block per 4 KiB) whose keys are the
    checkpoint's variable names and whose values are BundleEntryProto messages (dtype, shape, ...);
  * `.meta`  -- a serialized MetaGraphDef protobuf.
TensorFlow's generated classes are not installed, so the protobuf WIRE FORMAT is walked directly (field numbers
from tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape}.proto and
tensorflow/core/protobuf/{meta_graph,tensor_bundle}.proto, TF 1.0).  Pure data readers: nothing from the files is
executed.  Test infrastructure only.
"""
import struct


# ---- protobuf wire format ----------------------------------------------------------------------------
def varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def fields(buf):
    """-> list of (field number, wire type, value); value = int (varint / fixed) or bytes (length-delimited)."""
    out, pos, n = [], 0, len(buf)
    while pos < n:
        key, pos = varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.append((num, wt, v))
    return out


def first(fs, num, default=None):
    for n, _, v in fs:
        if n == num:
            return v
    return default


def every(fs, num):
    return [v for n, _, v in fs if n == num]


def packed_varints(b):
    out, pos = [], 0
    while pos < len(b):
        v, pos = varint(b, pos)
        out.append(v)
    return out


# ---- .index (sorted string table) ----------------------------------------------------------------------
def _block_entries(block):
    """Entries of one table block: [shared varint][non_shared varint][value_len varint][key delta][value], followed by
    the restart array (u32 offsets + u32 count)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = varint(block, pos)
        non_shared, pos = varint(block, pos)
        vlen, pos = varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path):
    """-> {key (str): value (bytes)} of an uncompressed table file (TF writes checkpoint indexes uncompressed)."""
    data = open(path, "rb").read()
    footer = data[-48:]
    assert footer[-8:] == struct.pack("<Q", 0xdb4775248b80fb57), "not a table file"
    pos = 0
    _, pos = varint(footer, pos)        # metaindex handle
    _, pos = varint(footer, pos)
    idx_off, pos = varint(footer, pos)
    idx_len, pos = varint(footer, pos)
    out = {}
    for _, handle in _block_entries(data[idx_off:idx_off + idx_len]):
        off, p = varint(handle, 0)
        ln, p = varint(handle, p)
        assert data[off + ln] == 0, "compressed block"       # 1-byte type trailer: 0 = none
        for k, v in _block_entries(data[off:off + ln]):
            out[k.decode("latin-1")] = v
    return out


def bundle_entries(path):
    """-> {variable name: shape tuple} from a checkpoint .index (BundleEntryProto: 1 dtype, 2 shape; the header
    entry under the empty key is skipped).  TensorShapeProto: repeated dim = 2 { size = 1 }."""
    out = {}
    for k, v in read_table(path).items():
        if k == "":
            continue
        fs = fields(v)
        shape = first(fs, 2, b"")
        out[k] = tuple(first(fields(d), 1, 0) for d in every(fields(shape), 2))
    return out


# ---- .meta (MetaGraphDef -> GraphDef -> NodeDef) ---------------------------------------------------------
class Node(object):
    def __init__(self, raw):
        fs = fields(raw)
        self.name = first(fs, 1, b"").decode()
        self.op = first(fs, 2, b"").decode()
        self.inputs = [v.decode() for v in every(fs, 3)]
        self.attr = {}
        for entry in every(fs, 5):                   # map<string, AttrValue>: entries {1: key, 2: value}
            es = fields(entry)
            self.attr[first(es, 1, b"").decode()] = first(es, 2, b"")

    # AttrValue: 2 s, 3 i, 4 f, 5 b, 6 type, 7 shape, 8 tensor, 1 list
    def attr_s(self, key):
        return first(fields(self.attr[key]), 2, b"").decode()

    def attr_ints(self, key):
        lst = first(fields(self.attr[key]), 1, b"")
        out = []
        for n, wt, v in fields(lst):
            if n == 3:
                out.extend(packed_varints(v) if wt == 2 else [v])
        return out

    def const_floats(self):
        """Values of a float32 Const node (TensorProto: 4 tensor_content bytes, 5 float_val) and its dims."""
        import numpy as np
        t = fields(first(fields(self.attr["value"]), 8, b""))
        shape = first(t, 2, b"")
        dims = [first(fields(d), 1, 0) for d in every(fields(shape), 2)]
        content = first(t, 4)
        if content:
            vals = np.frombuffer(content, dtype="<f4")
        else:
            vals = []
            for n, wt, v in t:
                if n == 5:
                    vals.extend(np.frombuffer(v, dtype="<f4") if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]])
            vals = np.asarray(vals, dtype=np.float32)
        return vals, dims


def graph_nodes(meta_path):
    """-> {node name: Node} of the MetaGraphDef's graph_def (MetaGraphDef field 2; GraphDef.node = 1)."""
    meta = fields(open(meta_path, "rb").read())
    graph = fields(first(meta, 2, b""))
    nodes = [Node(raw) for raw in every(graph, 1)]
    return {n.name: n for n in nodes}


def producer(nodes, input_name):
    """Node that produces `input_name` ('name', 'name:1', '^control')."""
    return nodes[input_name.lstrip("^").split(":")[0]]


def through_identity(nodes, input_name):
    """Follow Identity ('/read') nodes back to the real producer."""
    n = producer(nodes, input_name)
    while n.op == "Identity":
        n = producer(nodes, n.inputs[0])
    return n
