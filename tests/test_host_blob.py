"""Host half of the blob reader (no GPU): IPC metadata and LZ4 frame headers through ssa_blob_describe, on blobs written
by Arrow (tests/golden/ipc) and on damaged ones -- every outcome is a description or an error, never a crash."""
import ctypes
import json
import os

import numpy as np
import pytest

from ydb_amd import _lib
from ydb_amd.arrow_c import Col, HostBatch

IPC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ipc")
CASES = json.load(open(os.path.join(IPC, "cases.json")))


def schema_batch():
    """A HostBatch whose schema is the fixtures' (for lone RecordBatch messages)."""
    from decimal import Decimal
    return HostBatch({
        "i64": np.zeros(1, np.int64), "i32n": np.zeros(1, np.int32), "u8": np.zeros(1, np.uint8), "u16": np.zeros(1, np.uint16),
        "f64": np.zeros(1, np.float64), "f32n": np.zeros(1, np.float32), "flag": np.zeros(1, np.bool_), "s": Col(["x"], type="string"),
        "bn": Col([b"x"], type="binary"), "ts": Col(np.zeros(1, np.int64), type="timestamp"),
        "dec": Col(np.zeros((1, 2), np.uint64), decimal=(20, 2)),
    })


def describe(blob, schema=None):
    lib = _lib.load()
    err = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(bytes(blob), len(blob))
    p = lib.ssa_blob_describe(ctypes.cast(buf, ctypes.c_void_p), len(blob), ctypes.byref(schema.schema) if schema is not None else None,
                              ctypes.byref(err))
    msg = _lib.take_string(lib, err.value)
    return (_lib.take_string(lib, p) if p else None), msg


@pytest.mark.parametrize("case", CASES["cases"], ids=lambda c: c["name"])
def test_describe_fixture(case):
    blob = open(os.path.join(IPC, case["name"] + ".bin"), "rb").read()
    text, err = describe(blob, schema_batch() if case["form"] == "payload" else None)
    assert text is not None, err
    head = text.splitlines()[0]
    assert f"rows={case['rows']} " in head
    assert {"raw": "codec=none", "lz4": "codec=lz4_frame", "zstd": "codec=zstd"}[case["codec"]] in head
    assert len(text.splitlines()) == 1 + len(CASES["dictionary_schema"] if case.get("dictionary") else CASES["schema"])
    if case.get("dictionary"):
        assert text.count("dictionary(id") == 3 and "7 entries" in text and "2 entries" in text and "5 entries" in text
    if case["codec"] == "lz4" and case["rows"] >= 30000:
        assert "linked" in text and "blocks of 65536" in text   # what Arrow's codec writes: 64 KB blocks that refer back


def test_payload_needs_a_schema():
    blob = open(os.path.join(IPC, "small_lz4_payload.bin"), "rb").read()
    text, err = describe(blob)
    assert text is None and "schema" in err


def test_damaged_blobs_are_errors():
    rng = np.random.default_rng(3)
    blob = bytearray(open(os.path.join(IPC, "small_lz4_stream.bin"), "rb").read())
    outcomes = {"ok": 0, "error": 0}
    for _ in range(600):
        b = bytearray(blob)
        kind = rng.integers(3)
        if kind == 0:
            b = b[:rng.integers(0, len(b))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 8))):
                b[rng.integers(0, min(len(b), 2500))] = rng.integers(0, 256)
        else:
            pos = rng.integers(0, len(b) - 8)
            b[pos:pos + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
        text, err = describe(bytes(b))
        outcomes["ok" if text is not None else "error"] += 1
        assert text is not None or err
    assert outcomes["error"] > 100


def test_a_row_count_beyond_32_bits_is_refused():
    """A batch that announces 2^61 rows: the sizes of its buffers, as products of that count, would wrap to a few bytes."""
    import struct

    import pyarrow as pa
    rows = 12345
    batch = pa.record_batch({"i64": pa.array(np.arange(rows, dtype=np.int64)), "s": pa.array(["x"] * rows)})
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    blob = sink.getvalue().to_pybytes()
    text, err = describe(blob)
    assert text is not None and f"rows={rows} " in text, err
    # the RecordBatch message's length and its field nodes' lengths: every 64-bit 12345 ahead of the body
    body = blob.index(np.arange(8, dtype=np.int64).tobytes())
    head = blob[:body]
    needle = struct.pack("<q", rows)
    assert head.count(needle) >= 3
    for huge in (1 << 61, (1 << 62) + 5, (1 << 31)):
        damaged = head.replace(needle, struct.pack("<q", huge)) + blob[body:]
        text, err = describe(damaged)
        assert text is None and "row count" in err, (text, err)
