#!/usr/bin/env python3
"""Writes the IPC blob fixtures under tests/golden/ipc/ with pyarrow (the same upstream as the reference's Arrow; run in
the build container, where pyarrow is importable):

    python tests/golden/make_ipc_fixtures.py

Each case is <name>.bin (the blob, as TNativeSerializer::DoSerializeFull / DoSerializePayload would store it) and
<name>.npz (the expected columns: <col>.values / <col>.valid, strings as <col>.offsets / <col>.chars), plus cases.json.
"""
import io
import json
import os

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipc")


def table(n, rng, compressible=True):
    words = ["", "a", "abc", "google", "the quick brown fox jumps over the lazy dog", "x" * 90]
    ids = rng.integers(0, len(words), n)
    strs = [words[i] for i in ids]
    ints = (rng.integers(0, 50, n) if compressible else rng.integers(-2**62, 2**62, n)).astype(np.int64)
    dbl = (rng.integers(0, 11, n) / 100.0) if compressible else rng.normal(size=n)
    valid = rng.random(n) > 0.2
    cols = {
        "i64": pa.array(ints, type=pa.int64()),
        "i32n": pa.array(rng.integers(-1000, 1000, n).astype(np.int32), mask=~valid),
        "u8": pa.array(rng.integers(0, 3, n).astype(np.uint8)),
        "u16": pa.array((np.arange(n) % 1000).astype(np.uint16)),
        "f64": pa.array(dbl, type=pa.float64()),
        "f32n": pa.array(rng.normal(size=n).astype(np.float32), mask=~valid),
        "flag": pa.array(rng.random(n) < 0.3),
        "s": pa.array(strs, type=pa.utf8()),
        "bn": pa.array([None if not v else w.encode() for v, w in zip(valid, strs)], type=pa.binary()),
        "ts": pa.array((np.arange(n, dtype=np.int64) * 1000003), type=pa.timestamp("us")),
        "dec": pa.array([None if i % 7 == 3 else __import__("decimal").Decimal(int(ints[i]) % 10**9) / 100 for i in range(n)], type=pa.decimal128(20, 2)),
    }
    return pa.table(cols)


def expected(t):
    out = {}
    for name in t.column_names:
        arr = t.column(name).combine_chunks()
        valid = np.array([v is not None for v in arr.to_pylist()], dtype=np.bool_) if arr.null_count else np.ones(len(arr), np.bool_)
        out[name + ".valid"] = valid
        typ = arr.type
        if pa.types.is_string(typ) or pa.types.is_binary(typ):
            vals = [b"" if v is None else (v.encode() if isinstance(v, str) else v) for v in arr.to_pylist()]
            out[name + ".offsets"] = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int32)
            out[name + ".chars"] = np.frombuffer(b"".join(vals), dtype=np.uint8)
        elif pa.types.is_boolean(typ):
            out[name + ".values"] = np.array([bool(v) for v in arr.to_pylist()], dtype=np.uint8)
        elif pa.types.is_decimal(typ):
            words = []
            for v in arr.to_pylist():
                x = 0 if v is None else int(v.scaleb(typ.scale))
                x &= (1 << 128) - 1
                words += [x & (2**64 - 1), x >> 64]
            out[name + ".values"] = np.array(words, dtype=np.uint64)
        elif pa.types.is_timestamp(typ):
            out[name + ".values"] = np.asarray(arr.cast(pa.int64()).fill_null(0))
        else:
            out[name + ".values"] = np.asarray(arr.fill_null(0))
    return out


def full_stream(t, codec):
    sink = io.BytesIO()
    opts = ipc.IpcWriteOptions(compression=codec)
    with ipc.new_stream(sink, t.schema, options=opts) as w:
        if t.num_rows:
            w.write_batch(t.to_batches()[0])
        elif any(pa.types.is_dictionary(f.type) for f in t.schema):
            w.write_batch(pa.RecordBatch.from_arrays([c.combine_chunks() for c in t.columns], schema=t.schema))  # (keeps the dictionaries)
        else:
            w.write_batch(pa.RecordBatch.from_pylist([], schema=t.schema))
    return sink.getvalue()


def payload_only(t, codec):
    """The lone RecordBatch message (DoSerializePayload): the stream without its schema message and end marker."""
    whole = full_stream(t, codec)
    import struct
    pos = 0
    marker, meta = struct.unpack_from("<II", whole, 0)
    assert marker == 0xFFFFFFFF
    # the schema message has no body
    pos = 8 + meta
    m2, meta2 = struct.unpack_from("<II", whole, pos)
    assert m2 == 0xFFFFFFFF
    return whole[pos:len(whole) - 8]


def dictionary_table(n, rng):
    """Dictionary-encoded string columns as the dictionary transformer stores them (ArrayToDictionary,
    ydb/core/formats/arrow/dictionary/conversion.h:11-12): codes of several widths, nulls among the rows, entries no row
    uses, the empty string as an entry; a plain column next to them."""
    words = ["", "N", "R", "A", "google", "the quick brown fox jumps over the lazy dog", "never used"]
    codes = rng.integers(0, len(words) - 1, n)
    valid = rng.random(n) > 0.25
    few = ["F", "O"]
    cols = {
        "ds": pa.DictionaryArray.from_arrays(pa.array(codes.astype(np.int32), mask=~valid), pa.array(words, type=pa.utf8())),
        "flag": pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 2, n).astype(np.int8)), pa.array(few, type=pa.utf8())),
        "db": pa.DictionaryArray.from_arrays(pa.array((codes % 5).astype(np.uint16)), pa.array([w.encode() for w in words[:5]], type=pa.binary())),
        "v": pa.array(rng.integers(0, 100, n).astype(np.int64)),
    }
    return pa.table(cols)


def expected_decoded(t):
    return expected(pa.table({name: (t.column(name).combine_chunks().dictionary_decode() if pa.types.is_dictionary(t.schema.field(name).type)
                                     else t.column(name)) for name in t.column_names}))


def main():
    os.makedirs(HERE, exist_ok=True)
    rng = np.random.default_rng(2025)
    cases = []
    for name, n, compressible in (("empty", 0, True), ("one", 1, True), ("small", 1000, True), ("blocks", 30000, True), ("random", 9000, False)):
        t = table(n, rng, compressible)
        exp = expected(t)
        for codec in (None, "lz4", "zstd"):
            if codec == "zstd" and name != "small":
                continue
            if codec is None and name in ("blocks", "random"):   # (large and adds nothing over the small raw cases)
                continue
            for form in ("stream", "payload"):
                if form == "payload" and name not in ("small", "blocks"):
                    continue
                blob = full_stream(t, codec) if form == "stream" else payload_only(t, codec)
                case = f"{name}_{codec or 'raw'}_{form}"
                open(os.path.join(HERE, case + ".bin"), "wb").write(blob)
                cases.append({"name": case, "rows": n, "codec": codec or "raw", "form": form, "expected": name, "bytes": len(blob)})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **exp)
    schema = [(f.name, str(f.type)) for f in table(1, rng).schema]
    drng = np.random.default_rng(2026)
    for name, n in (("dict_empty", 0), ("dict_small", 700), ("dict_blocks", 40000)):
        t = dictionary_table(n, drng)
        for codec in (None, "lz4"):
            blob = full_stream(t, codec)
            case = f"{name}_{codec or 'raw'}_stream"
            open(os.path.join(HERE, case + ".bin"), "wb").write(blob)
            cases.append({"name": case, "rows": n, "codec": codec or "raw", "form": "stream", "expected": name, "bytes": len(blob), "dictionary": True})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **expected_decoded(t))
    dict_schema = [("ds", "string"), ("flag", "string"), ("db", "binary"), ("v", "int64")]
    json.dump({"cases": cases, "schema": schema, "dictionary_schema": dict_schema,
               "about": "made by tests/golden/make_ipc_fixtures.py with pyarrow " + pa.__version__},
              open(os.path.join(HERE, "cases.json"), "w"), indent=1)
    print(sum(c["bytes"] for c in cases), "bytes of blobs in", len(cases), "cases")


if __name__ == "__main__":
    main()
