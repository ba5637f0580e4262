"""Column blobs decoded on the GPU (ssa_batch_from_blob / ssa_batches_from_blobs) against the columns Arrow wrote them
from (tests/golden/ipc) and against the CPU oracle reading the same blob; then a program over the decoded batch."""
import json
import os

import numpy as np
import pytest

from test_host_blob import CASES, IPC, schema_batch
from test_oracle_ipc import DICT_SCHEMA, SCHEMA, check_columns, schema_of

pytestmark = pytest.mark.gpu


def container(schema=None):
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import Program
    pc = ProgramContainer()
    ok, err = pc.Init([(i + 1, n) for i, (n, _) in enumerate(schema or SCHEMA)], Program().project([1]))
    assert ok, err
    return pc


def as_columns(exported):
    """Export() result -> the shape check_columns takes."""
    out = {}
    for name, c in exported.items():
        col = {"type": c.type, "valid": np.asarray(c.valid, dtype=np.bool_)}
        if c.type in ("string", "binary"):
            vals = [bytes(v) if v is not None else b"" for v in c.values]
            col["offsets"] = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int32)
            col["chars"] = b"".join(vals)
        elif c.type == "decimal128":
            col["values"] = np.asarray(c.values, dtype=np.uint64).reshape(-1)
        elif c.type == "bool":
            col["values"] = np.asarray(c.values).astype(np.uint8)
        else:
            col["values"] = np.asarray(c.values)
        out[name] = col
    return out


@pytest.mark.parametrize("case", [c for c in CASES["cases"] if c["codec"] != "zstd"], ids=lambda c: c["name"])
def test_blob_decodes_to_the_columns_arrow_wrote(case):
    from oracle import ipc_oracle
    schema = schema_of(case)
    pc = container(schema)
    blob = open(os.path.join(IPC, case["name"] + ".bin"), "rb").read()
    payload = case["form"] == "payload"
    dev = pc.ImportBlob(blob, schema_batch() if payload else None)
    assert dev.rows == case["rows"]
    got = as_columns(pc.Export(dev))  # (dictionary-encoded columns leave as the columns they stand for)
    check_columns(got, np.load(os.path.join(IPC, case["expected"] + ".npz")), case["rows"], schema)
    # and byte for byte what the oracle reads, null slots included
    exp = ipc_oracle.read_blob(blob, SCHEMA if payload else None)
    for name, typ in schema:
        if typ in ("string", "binary"):
            assert got[name]["chars"] == exp[name]["chars"][exp[name]["offsets"][0]:exp[name]["offsets"][-1]] or case["rows"] == 0
        else:
            valid = exp[name]["valid"]
            g, e = np.asarray(got[name]["values"]), np.asarray(exp[name]["values"])
            if typ == "decimal128":
                g, e, valid = g.reshape(-1, 2), e.reshape(-1, 2), valid
            assert np.array_equal(g[valid], e[valid]), name


def test_many_blobs_in_one_call():
    pc = container()
    names = [c["name"] for c in CASES["cases"] if c["codec"] == "lz4" and c["form"] == "stream"] * 3
    blobs = [open(os.path.join(IPC, n + ".bin"), "rb").read() for n in names]
    devs = pc.ImportBlobs(blobs)
    assert len(devs) == len(blobs)
    by_name = {c["name"]: c for c in CASES["cases"]}
    for n, d in zip(names, devs):
        check_columns(as_columns(pc.Export(d)), np.load(os.path.join(IPC, by_name[n]["expected"] + ".npz")), by_name[n]["rows"], schema_of(by_name[n]))


def shapes_blob(rows, seed):
    """An LZ4 stream whose buffers hold every shape of block the decoder knows: stored blocks between compressed ones
    (a stretch of random numbers inside a column that otherwise compresses), long runs of literals (text), matches of
    tens of kilobytes (zeros: lengths of many 255 bytes), sequences without literals (a repeating pattern), and frames of
    many linked blocks."""
    import io
    import pyarrow as pa
    import pyarrow.ipc as ipc
    rng = np.random.default_rng(seed)
    third = rows // 3
    mixed = np.concatenate([np.tile(np.arange(7, dtype=np.int64), third // 7 + 1)[:third], rng.integers(-2**62, 2**62, third),
                            np.arange(rows - 2 * third, dtype=np.int64) * 3])
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 12))).astype(np.uint8)) for _ in range(3000)]
    text = [b" ".join(words[j] for j in rng.integers(0, 3000, int(rng.integers(1, 9)))) for _ in range(rows // 4)] + [b""] * (rows - rows // 4)
    sparse = np.where(rng.random(rows) < 0.02, rng.integers(0, 2**31 - 1, rows), 0).astype(np.int32)
    t = pa.table({"mixed": pa.array(mixed), "text": pa.array(text, type=pa.binary()), "zeros": pa.array(np.zeros(rows, dtype=np.int64)),
                  "sparse": pa.array(sparse)})
    sink = io.BytesIO()
    with ipc.new_stream(sink, t.schema, options=ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(t, max_chunksize=rows)
    return sink.getvalue()


@pytest.mark.parametrize("split", ["1", "0"], ids=["parse_and_copy_apart", "one_wave_does_both"])
@pytest.mark.parametrize("rows,seed", [(120_000, 77), (50_001, 78), (700_000, 79)])
def test_lz4_blocks_of_every_shape(rows, seed, split, monkeypatch):
    """Both ways the library decodes LZ4 frames: the blocks parsed side by side into records and the frames' waves left
    with the copies (the default), and one wave per frame doing both (what it falls back to without room for the
    records; SSA_LZ4_SPLIT=0 asks for it)."""
    from oracle import ipc_oracle
    monkeypatch.setenv("SSA_LZ4_SPLIT", split)
    schema = [("mixed", "int64"), ("text", "binary"), ("zeros", "int64"), ("sparse", "int32")]
    blob = shapes_blob(rows, seed)
    pc = container(schema)
    dev = pc.ImportBlob(blob)
    assert dev.rows == rows
    got = as_columns(pc.Export(dev))
    exp = ipc_oracle.read_blob(blob, None)
    for name, typ in schema:
        if typ == "binary":
            assert got[name]["chars"] == bytes(exp[name]["chars"][exp[name]["offsets"][0]:exp[name]["offsets"][-1]])
            assert np.array_equal(got[name]["offsets"], np.asarray(exp[name]["offsets"]) - exp[name]["offsets"][0])
        else:
            assert np.array_equal(np.asarray(got[name]["values"]), np.asarray(exp[name]["values"])), name


def test_an_import_of_tens_of_megabytes_goes_through_the_pinned_slots():
    """Blobs of 16 MB and more in all reach the device through slots of pinned memory filled by several threads: the
    pieces cut the blobs wherever a slot ends."""
    from oracle import ipc_oracle
    schema = [("mixed", "int64"), ("text", "binary"), ("zeros", "int64"), ("sparse", "int32")]
    distinct = [shapes_blob(700_000, 80), shapes_blob(650_001, 81), shapes_blob(300_000, 82)]
    blobs = [distinct[i % 3] for i in range(7)]
    assert sum(len(b) for b in blobs) > (24 << 20)
    pc = container(schema)
    devs = pc.ImportBlobs(blobs)
    expected = [ipc_oracle.read_blob(b, None) for b in distinct]
    for i, dev in enumerate(devs):
        got = as_columns(pc.Export(dev))
        exp = expected[i % 3]
        for name, typ in schema:
            if typ == "binary":
                assert got[name]["chars"] == bytes(exp[name]["chars"][exp[name]["offsets"][0]:exp[name]["offsets"][-1]]), (i, name)
            else:
                assert np.array_equal(np.asarray(got[name]["values"]), np.asarray(exp[name]["values"])), (i, name)


@pytest.mark.parametrize("split", ["1", "0"], ids=["parse_and_copy_apart", "one_wave_does_both"])
def test_zstd_is_declined_and_damage_is_an_error(split, monkeypatch):
    from ydb_amd import _lib
    monkeypatch.setenv("SSA_LZ4_SPLIT", split)
    from ydb_amd.program import SsaError
    pc = container()
    with pytest.raises(SsaError) as e:
        pc.ImportBlob(open(os.path.join(IPC, "small_zstd_stream.bin"), "rb").read())
    assert e.value.code == _lib.SSA_UNSUPPORTED and "ZSTD" in str(e.value)
    blob = bytearray(open(os.path.join(IPC, "blocks_lz4_stream.bin"), "rb").read())
    rng = np.random.default_rng(9)
    outcomes = {"ok": 0, "invalid": 0}
    for _ in range(40):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 30))):
            b[rng.integers(2000, len(b))] = rng.integers(0, 256)   # somewhere in the compressed body
        try:
            pc.Export(pc.ImportBlob(bytes(b)))
            outcomes["ok"] += 1     # (a flipped literal byte decodes fine, to other data)
        except SsaError as err:
            assert err.code in (_lib.SSA_INVALID, _lib.SSA_UNSUPPORTED), err
            outcomes["invalid"] += 1
    assert outcomes["invalid"] > 0
    # the handle still works
    good = pc.ImportBlob(bytes(blob))
    assert good.rows == 30000


def test_program_over_a_decoded_blob():
    from cases import compare_batches
    from oracle import ssa_oracle
    from ydb_amd.arrow_c import Col
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    exp_cols = np.load(os.path.join(IPC, "blocks.npz"))
    resolver = [(i + 1, n) for i, (n, _) in enumerate(SCHEMA)]
    ids = {n: i for i, n in resolver}
    p = (Program().assign_const(100, "Int64", 25).assign_func(101, F.CMP_LESS, [ids["i64"], 100]).filter(101)
         .group_by([Aggregate(110, AGG.SUM, [ids["f64"]]), Aggregate(111, AGG.COUNT, [ids["i32n"]]), Aggregate(112, AGG.MAX, [ids["u16"]])],
                   keys=[ids["u8"], ids["s"]]))
    pc = ProgramContainer()
    assert pc.Init(resolver, p)[0]
    dev = pc.ImportBlob(open(os.path.join(IPC, "blocks_lz4_stream.bin"), "rb").read())
    got = pc.Export(pc.ApplyDevice(dev))
    so, sc = exp_cols["s.offsets"], exp_cols["s.chars"].tobytes()
    host = {"i64": exp_cols["i64.values"], "f64": exp_cols["f64.values"], "u16": exp_cols["u16.values"], "u8": exp_cols["u8.values"],
            "i32n": Col(exp_cols["i32n.values"], valid=exp_cols["i32n.valid"]),
            "s": Col([sc[so[i]:so[i + 1]].decode() for i in range(len(so) - 1)], type="string")}
    exp = ssa_oracle.Oracle(resolver, p).apply(host)
    compare_batches(got, exp, key_names=["u8", "s"], float_sum_names=["110"])


def test_program_over_a_blob_with_dictionary_columns():
    """Dictionary-encoded columns of a blob stay encoded in HBM: a filter on one of them is evaluated per entry, the
    GROUP BY groups by code -- with the results of the decoded columns (dictionary/conversion.h:13-15)."""
    import ctypes
    from cases import compare_batches
    from oracle import ssa_oracle
    from ydb_amd import _lib
    from ydb_amd.arrow_c import Col
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    exp_cols = np.load(os.path.join(IPC, "dict_blocks.npz"))
    resolver = [(i + 1, n) for i, (n, _) in enumerate(DICT_SCHEMA)]
    ids = {n: i for i, n in resolver}
    p = (Program().assign_const(100, "Bytes", b"o").assign_func(101, F.STR_MATCH, [ids["ds"], 100]).filter(101)
         .group_by([Aggregate(110, AGG.SUM, [ids["v"]]), Aggregate(111, AGG.COUNT, [ids["ds"]]), Aggregate(112, AGG.COUNT, [])],
                   keys=[ids["flag"], ids["db"]]))
    pc = ProgramContainer()
    assert pc.Init(resolver, p)[0]

    def on_codes():
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.load().ssa_dictionary_stats(ctypes.byref(a), ctypes.byref(b))
        return a.value

    before = on_codes()
    dev = pc.ImportBlob(open(os.path.join(IPC, "dict_blocks_lz4_stream.bin"), "rb").read())
    got = pc.Export(pc.ApplyDevice(dev))
    assert on_codes() > before

    def strings(name, as_text):
        o, c, v = exp_cols[name + ".offsets"], exp_cols[name + ".chars"].tobytes(), exp_cols[name + ".valid"]
        vals = [c[o[i]:o[i + 1]] for i in range(len(o) - 1)]
        return Col([x.decode() for x in vals] if as_text else vals, type="string" if as_text else "binary", valid=None if v.all() else v)

    host = {"ds": strings("ds", True), "flag": strings("flag", True), "db": strings("db", False), "v": exp_cols["v.values"]}
    exp = ssa_oracle.Oracle(resolver, p).apply(host)
    compare_batches(got, exp, key_names=["flag", "db"])


def long_match_blob(period, match=100_000):
    """(blob, expected bytes): an IPC stream with one uint8 column whose data buffer is an LZ4 frame with one 256 KB block
    (Arrow writes 64 KB ones; the format allows up to 4 MB) -- the pattern as literals, one match `period` bytes back that
    repeats it for `match` bytes, and the closing literals."""
    import struct

    import pyarrow as pa
    import xxhash
    pattern = bytes((17 * i + 5) % 251 for i in range(period))
    for extra in range(5, 14):  # (until the buffer Arrow writes is not a multiple of eight bytes long: see below)
        tail = bytes(range(1, extra + 1))
        expected = (pattern * (match // period + 2))[:period + match] + tail
        lit = len(pattern)
        block = bytearray([(min(lit, 15) << 4) | 15])
        if lit >= 15:
            block += b"\xff" * ((lit - 15) // 255) + bytes([(lit - 15) % 255])
        block += pattern + struct.pack("<H", period)
        rest = match - 4 - 15
        block += b"\xff" * (rest // 255) + bytes([rest % 255])
        block += bytes([len(tail) << 4]) + tail
        descriptor = bytes([0x60, 0x50])  # version 1, independent blocks, no checksums; blocks of up to 256 KB
        frame = struct.pack("<I", 0x184D2204) + descriptor + bytes([(xxhash.xxh32(descriptor, seed=0).intdigest() >> 8) & 0xFF])
        frame += struct.pack("<I", len(block)) + bytes(block) + struct.pack("<I", 0)
        # a stream written by Arrow for a column of the same length; the frame of its one data buffer is replaced, and with it
        # the two lengths in the metadata that describe it: the buffer's and the (padded) body's
        n = len(expected)
        batch = pa.record_batch({"u8": pa.array(np.frombuffer(expected, dtype=np.uint8))})
        sink = pa.BufferOutputStream()
        with pa.ipc.new_stream(sink, batch.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
            w.write_batch(batch)
        blob = sink.getvalue().to_pybytes()
        at = blob.index(struct.pack("<I", 0x184D2204))
        assert struct.unpack_from("<q", blob, at - 8)[0] == n
        eos = blob.rindex(struct.pack("<iI", -1, 0))
        head, body = blob[:at - 8], eos - (at - 8)
        lengths = [c for c in range(body - 7, body) if head.count(struct.pack("<q", c)) == 1]
        if head.count(struct.pack("<q", body)) != 1 or len(lengths) != 1:
            continue
        new_buffer = 8 + len(frame)
        padded = (new_buffer + 7) // 8 * 8
        head = head.replace(struct.pack("<q", lengths[0]), struct.pack("<q", new_buffer)).replace(struct.pack("<q", body), struct.pack("<q", padded))
        return head + struct.pack("<q", n) + frame + b"\0" * (padded - new_buffer) + blob[eos:], expected
    raise AssertionError("no stream whose two lengths can be told apart")


@pytest.mark.parametrize("split", ["1", "0"], ids=["parse_and_copy_apart", "one_wave_does_both"])
@pytest.mark.parametrize("period", [3, 7, 64])
def test_a_match_longer_than_64_kb_that_repeats_a_short_pattern(split, period, monkeypatch):
    """Longer than the 64 KB of output the decode kernels keep in LDS: the match's later bytes come from its own output."""
    monkeypatch.setenv("SSA_LZ4_SPLIT", split)
    crafted, expected = long_match_blob(period)
    pc = container([("u8", None)])
    got = pc.Export(pc.ImportBlob(crafted))
    assert bytes(np.asarray(got["u8"].values).astype(np.uint8)) == expected
