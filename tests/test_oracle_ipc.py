"""The blob-reading oracle (oracle/ipc_oracle.py + oracle/lz4_oracle.c) against blobs written by Arrow itself
(tests/golden/ipc, made by tests/golden/make_ipc_fixtures.py) and, where it was built (the build container), against
the reference's own LZ4 library compiled from its sources (oracle/_ref/liblz4_ref.so)."""
import ctypes
import json
import os
import struct

import numpy as np
import pytest

from oracle import ipc_oracle

IPC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ipc")
CASES = json.load(open(os.path.join(IPC, "cases.json")))
TYPE_OF = {"int64": "int64", "int32": "int32", "uint8": "uint8", "uint16": "uint16", "double": "double", "float": "float", "bool": "bool",
           "string": "string", "binary": "binary", "timestamp[us]": "timestamp", "decimal128(20, 2)": "decimal128"}
SCHEMA = [(n, TYPE_OF[t]) for n, t in CASES["schema"]]
DICT_SCHEMA = [tuple(x) for x in CASES["dictionary_schema"]]  # the cases with dictionary-encoded columns, as decoded


def schema_of(case):
    return DICT_SCHEMA if case.get("dictionary") else SCHEMA


def check_columns(got, exp, rows, schema=None):
    """got: {name: {"type", "valid", "values" | "offsets" + "chars"}}; exp: the .npz of the case."""
    schema = schema or SCHEMA
    assert list(got.keys()) == [n for n, _ in schema]
    for name, typ in schema:
        g = got[name]
        assert g["type"] == typ, (name, g["type"])
        valid = exp[name + ".valid"]
        assert np.array_equal(np.asarray(g["valid"], dtype=np.bool_), valid), name
        if typ in ("string", "binary"):
            go, eo = np.asarray(g["offsets"], dtype=np.int64), exp[name + ".offsets"].astype(np.int64)
            # (null slots may hold any bytes: compare the valid rows)
            gc, ec = bytes(g["chars"]), exp[name + ".chars"].tobytes()
            for r in np.nonzero(valid)[0][:2000]:
                assert gc[go[r]:go[r + 1]] == ec[eo[r]:eo[r + 1]], (name, r)
            assert len(go) == rows + 1
        elif typ == "decimal128":
            gv, ev = np.asarray(g["values"]).reshape(-1, 2), exp[name + ".values"].reshape(-1, 2)
            assert np.array_equal(gv[valid], ev[valid]), name
        else:
            gv, ev = np.asarray(g["values"]), exp[name + ".values"]
            assert gv.dtype == ev.dtype or typ in ("bool", "timestamp"), (name, gv.dtype, ev.dtype)
            assert np.array_equal(gv[valid].view(np.uint8), ev[valid].astype(gv.dtype).view(np.uint8)), name


@pytest.mark.parametrize("case", [c for c in CASES["cases"] if c["codec"] != "zstd"], ids=lambda c: c["name"])
def test_oracle_reads_arrow_blobs(case):
    blob = open(os.path.join(IPC, case["name"] + ".bin"), "rb").read()
    exp = np.load(os.path.join(IPC, case["expected"] + ".npz"))
    got = ipc_oracle.read_blob(blob, SCHEMA if case["form"] == "payload" else None)
    check_columns(got, exp, case["rows"], schema_of(case))
    if case.get("dictionary"):  # read through the codes: the dictionary came in a message of its own
        assert got["ds"]["entries"] == 7 and got["flag"]["entries"] == 2 and got["db"]["entries"] == 5
        assert got["flag"]["codes"].max(initial=0) <= 1


def lz4_frames_of(blob):
    """(compressed frame bytes, announced length) of every compressed buffer of a stream blob's RecordBatch."""
    import io
    out = []
    pos = 0
    while True:
        marker, meta = struct.unpack_from("<II", blob, pos)
        assert marker == 0xFFFFFFFF
        if meta == 0:
            break
        fb = ipc_oracle._Fb(blob[pos + 8:pos + 8 + meta])
        msg = fb.root()
        body_len = fb.scalar(msg, 3, "q")
        body = pos + 8 + meta
        if fb.scalar(msg, 1, "B") == 3:
            header = fb.table(msg, 2)
            bufs, n = fb.vector(header, 2)
            for i in range(n):
                off, ln = struct.unpack_from("<qq", fb.b, bufs + 16 * i)
                if ln > 8:
                    ulen, = struct.unpack_from("<q", blob, body + off)
                    if ulen >= 0:
                        out.append((blob[body + off + 8: body + off + ln], ulen))
        pos = body + body_len
    return out


def test_lz4_restatement_equals_the_reference_library():
    ref_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "liblz4_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref is built from /root/reference in the build container only")
    ref = ctypes.CDLL(ref_path)
    ref.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    ref.LZ4F_createDecompressionContext.restype = ctypes.c_size_t
    ref.LZ4F_decompress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p,
                                    ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    ref.LZ4F_decompress.restype = ctypes.c_size_t
    ref.LZ4F_freeDecompressionContext.argtypes = [ctypes.c_void_p]
    ref.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    ref.LZ4F_compressFrame.restype = ctypes.c_size_t
    ref.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    ref.LZ4F_compressFrameBound.restype = ctypes.c_size_t

    def ref_decode(frame, size):
        ctx = ctypes.c_void_p()
        assert ref.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100) == 0
        out = ctypes.create_string_buffer(max(1, size))
        dn, sn = ctypes.c_size_t(size), ctypes.c_size_t(len(frame))
        rc = ref.LZ4F_decompress(ctx, out, ctypes.byref(dn), frame, ctypes.byref(sn), None)
        ref.LZ4F_freeDecompressionContext(ctx)
        assert rc == 0 and dn.value == size, (rc, dn.value, size)
        return out.raw[:size]

    nframes = 0
    for case in CASES["cases"]:
        if case["codec"] != "lz4" or case["form"] != "stream":
            continue
        blob = open(os.path.join(IPC, case["name"] + ".bin"), "rb").read()
        for frame, ulen in lz4_frames_of(blob):
            assert ipc_oracle.lz4_frame_decode(frame, ulen) == ref_decode(frame, ulen)
            nframes += 1
    assert nframes > 20
    # frames written by the reference's library itself (LZ4F_compressFrame with zeroed preferences, as Arrow's codec
    # calls it): runs, short periods, random bytes, a length that is not a multiple of the block size
    rng = np.random.default_rng(5)
    for data in (b"", b"a", b"ab" * 70000, bytes(rng.integers(0, 4, 300001, dtype=np.uint8)), bytes(rng.integers(0, 256, 200000, dtype=np.uint8)),
                 (b"0123456789abcdef" * 5000 + bytes(rng.integers(0, 256, 1000, dtype=np.uint8))) * 3):
        cap = ref.LZ4F_compressFrameBound(len(data), None)
        buf = ctypes.create_string_buffer(cap)
        n = ref.LZ4F_compressFrame(buf, cap, data, len(data), None)
        assert n < (1 << 40)
        assert ipc_oracle.lz4_frame_decode(buf.raw[:n], len(data)) == data


@pytest.mark.parametrize("period", [3, 7, 64])
def test_a_frame_with_a_256_kb_block_and_a_match_longer_than_64_kb(period):
    """The blob tests/test_gpu_blob.py crafts for the decode kernels, read by the oracle (and, where it was built, by the
    reference's own LZ4): the crafted stream is a valid one."""
    from oracle import ipc_oracle
    from test_gpu_blob import long_match_blob
    blob, expected = long_match_blob(period)
    got = ipc_oracle.read_blob(blob)
    assert bytes(np.asarray(got["u8"]["values"]).astype(np.uint8)) == expected
