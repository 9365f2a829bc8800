"""paac_amd/tf_bundle.py -- the reference's checkpoint container (TensorFlow V2 tensor bundle) -- pinned to the one
bundle index the reference ships, pretrained/breakout/checkpoints/-80000000.index (committed here as data:
tests/golden/tf_index_breakout.bin is that file's 1,283 bytes; no .data blob exists upstream, .MISSING_LARGE_BLOBS)."""
import os
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tfproto
from paac_amd import tf_bundle

GOLDEN_INDEX = os.path.join(HERE, "golden", "tf_index_breakout.bin")


def test_crc32c_known_answers_and_chunked_path():
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    assert tf_bundle.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 zero bytes
    assert tf_bundle.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # ... 32 bytes of 0xff
    rs = np.random.RandomState(0)
    for n in (0, 1, 2047, 2048, 16384, 16385, 70001):                        # both sides of the chunked path's threshold
        data = rs.randint(0, 256, n, dtype=np.uint8).tobytes()
        assert tf_bundle.crc32c(data) == (tf_bundle._crc_raw_small(data, 0xFFFFFFFF) ^ 0xFFFFFFFF), n


def test_reference_index_blocks_carry_our_crc():
    """Every block of the reference-written index passes OUR masked CRC-32C (table layout + checksum pinned)."""
    data = open(GOLDEN_INDEX, "rb").read()
    table = tf_bundle._read_table(data, verify=True)             # raises on a CRC mismatch
    assert table[0] == (b"", tf_bundle._HEADER_PROTO)            # header entry: one shard, producer version 1
    assert len(table) == 31


def test_writer_reproduces_the_reference_index_layout(tmp_path):
    """Writing tensors of the reference's shapes under its names gives an index whose entry set (names, dtypes, shapes,
    offsets, sizes) equals the reference's, parsed by the independent test-side reader (tests/tfproto.py) -- and, the
    CRCs of the tensor bytes apart, the same bytes: same blocks, same prefix compression, same separator key, same footer."""
    ref = tf_bundle.entries(_as_prefix(tmp_path, GOLDEN_INDEX))
    rs = np.random.RandomState(1)
    arrays = {k: rs.randn(*e["shape"]).astype(np.float32) for k, e in ref.items()}
    prefix = str(tmp_path / "-80000000")
    tf_bundle.write(prefix, arrays)
    ours = tf_bundle.entries(prefix)
    assert set(ours) == set(ref) and len(ours) == 30
    for k in ref:
        for f in ("dtype", "shape", "shard", "offset", "size"):
            assert ours[k][f] == ref[k][f], (k, f)
    assert tfproto.bundle_entries(prefix + ".index") == {k: e["shape"] for k, e in ref.items()}
    # byte-level: identical up to the 30 four-byte tensor CRCs and the two block CRCs that cover them
    a, b = open(prefix + ".index", "rb").read(), open(GOLDEN_INDEX, "rb").read()
    assert len(a) == len(b) and a[-48:] == b[-48:]
    assert sum(x != y for x, y in zip(a, b)) <= 4 * 30 + 4 + 4
    back = tf_bundle.read(prefix)
    assert all(np.array_equal(back[k], arrays[k]) for k in arrays)


def _as_prefix(tmp_path, index_file):
    """tf_bundle.entries wants `<prefix>.index`: link the committed bytes under such a name."""
    p = str(tmp_path / "ref")
    with open(p + ".index", "wb") as f:
        f.write(open(index_file, "rb").read())
    return p


def test_multi_block_table_round_trip(tmp_path):
    """More keys than one 4 KiB block holds: separators between blocks, restart points every 16 keys."""
    rs = np.random.RandomState(2)
    arrays = {"scope_%d/layer%03d/%s" % (i % 3, i, "w" * (i % 7 + 1)): rs.randn(i % 5 + 1, 3).astype(np.float32) for i in range(300)}
    prefix = str(tmp_path / "-7")
    tf_bundle.write(prefix, arrays)
    assert os.path.getsize(prefix + ".index") > 2 * 4096
    assert tfproto.bundle_entries(prefix + ".index") == {k: v.shape for k, v in arrays.items()}
    back = tf_bundle.read(prefix)
    assert set(back) == set(arrays) and all(np.array_equal(back[k], arrays[k]) for k in arrays)


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "-9")
    tf_bundle.write(prefix, {"a/b": np.arange(6, dtype=np.float32).reshape(2, 3)})
    assert tf_bundle.readable(prefix)
    blob = bytearray(open(prefix + tf_bundle.DATA_SUFFIX, "rb").read())
    blob[5] ^= 1
    open(prefix + tf_bundle.DATA_SUFFIX, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="CRC"):
        tf_bundle.read(prefix)
    open(prefix + tf_bundle.DATA_SUFFIX, "wb").write(bytes(blob[:10]))           # torn data file
    assert not tf_bundle.readable(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        tf_bundle.entries(prefix)


def test_saver_writes_and_resumes_from_either_container(tmp_path):
    from paac_amd.session import Saver
    state = {"x/w": np.arange(12, dtype=np.float32).reshape(3, 4)}
    got = {}
    for fmt, step in (("tf", 10), ("npz", 20), ("tf", 30)):
        s = Saver(lambda: state, got.update, max_to_keep=2, fmt=fmt)
        state["x/w"] = state["x/w"] + 1
        s.save(None, str(tmp_path), step)
    names = sorted(os.listdir(tmp_path))
    assert names == ["-20.npz", "-30.data-00000-of-00001", "-30.index", "checkpoint"], names      # step 10 pruned, both files
    assert 'model_checkpoint_path: "-30"' in open(tmp_path / "checkpoint").read()
    latest = Saver.latest_checkpoint(str(tmp_path))
    assert latest.endswith("-30.index") and Saver.step_of(latest) == 30
    Saver(lambda: state, got.update).restore(None, latest)
    assert np.array_equal(got["x/w"], state["x/w"])
    os.truncate(tmp_path / "-30.data-00000-of-00001", 8)                      # a torn newest checkpoint is skipped
    assert Saver.latest_checkpoint(str(tmp_path)).endswith("-20.npz")


def test_state_file_lists_every_kept_bundle_and_same_step_ties_go_to_the_newer_container(tmp_path):
    """tf.train.Saver's `checkpoint` state file names every checkpoint still on disk (oldest first), so a reference saver
    resuming from the folder keeps pruning them; and after switching --checkpoint_format at an unchanged step the container
    written last is the one resumed from."""
    import time
    from paac_amd.session import Saver
    state = {"x/w": np.zeros((2, 2), dtype=np.float32)}
    s = Saver(lambda: state, lambda d: None, max_to_keep=2, fmt="tf")
    for step in (5, 6, 7):
        s.save(None, str(tmp_path), step)
    lines = open(tmp_path / "checkpoint").read().splitlines()
    assert lines == ['model_checkpoint_path: "-7"', 'all_model_checkpoint_paths: "-6"', 'all_model_checkpoint_paths: "-7"']
    assert not os.path.exists(tmp_path / "-5.index")
    # same step, other container, written later
    state["x/w"] = state["x/w"] + 3
    time.sleep(0.02)
    Saver(lambda: state, lambda d: None, max_to_keep=5, fmt="npz").save(None, str(tmp_path), 7)
    assert Saver.latest_checkpoint(str(tmp_path)).endswith("-7.npz")
    time.sleep(0.02)
    state["x/w"] = state["x/w"] + 3
    s.save(None, str(tmp_path), 7)
    assert Saver.latest_checkpoint(str(tmp_path)).endswith("-7.index")
