"""CPU restatement of reading a column blob: Arrow IPC messages with optional LZ4_FRAME body compression.
TEST INFRASTRUCTURE -- the GPU product never imports this.

Follows NArrow::NSerialization::TNativeSerializer::DoDeserialize (ydb/core/formats/arrow/serializer/native.cpp:18-81),
i.e. arrow::ipc::RecordBatchStreamReader / ReadRecordBatch of the vendored Arrow 5.0
(contrib/libs/apache/arrow/cpp/src/arrow/ipc/reader.cc: message framing, FieldNode / Buffer lists, DecompressBuffers:
int64 uncompressed length, -1 = stored as is) over the flatbuffer layouts of format/Message.fbs and Schema.fbs.  Buffers
are decompressed by oracle/lz4_oracle.c.  Pinned on blobs written by Arrow itself (tests/golden/ipc, made by
tests/golden/make_ipc_fixtures.py) and, for the LZ4 part, on the reference's own library (oracle/_ref/liblz4_ref.so).
"""
from __future__ import annotations

import ctypes
import os
import struct
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lz4 = None


def lz4_lib():
    global _lz4
    if _lz4 is None:
        path = os.path.join(_HERE, "liblz4_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lz4 = ctypes.CDLL(path)
        _lz4.orc_lz4_frame_decode.restype = ctypes.c_int64
        _lz4.orc_lz4_frame_decode.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    return _lz4


def lz4_frame_decode(data: bytes, size: int) -> bytes:
    out = ctypes.create_string_buffer(max(1, size))
    n = lz4_lib().orc_lz4_frame_decode(data, len(data), out, size)
    if n != size:
        raise ValueError(f"LZ4 frame decodes to {n} bytes, {size} announced")
    return out.raw[:size]


class _Fb:
    """Flatbuffer reads (tables through vtables, forward offsets to strings / vectors / tables)."""

    def __init__(self, b: bytes):
        self.b = b

    def u(self, fmt, pos):
        return struct.unpack_from("<" + fmt, self.b, pos)[0]

    def root(self):
        return self.u("I", 0)

    def field(self, table, i):
        vt = table - self.u("i", table)
        if 4 + 2 * i + 2 > self.u("H", vt):
            return 0
        off = self.u("H", vt + 4 + 2 * i)
        return table + off if off else 0

    def scalar(self, table, i, fmt, default=0):
        p = self.field(table, i)
        return self.u(fmt, p) if p else default

    def table(self, table, i):
        p = self.field(table, i)
        return p + self.u("I", p) if p else 0

    def vector(self, table, i):
        p = self.field(table, i)
        if not p:
            return 0, 0
        v = p + self.u("I", p)
        return v + 4, self.u("I", v)

    def string(self, table, i):
        p, n = self.vector(table, i)
        return self.b[p:p + n].decode() if p else ""


_INT = {(8, 1): "int8", (8, 0): "uint8", (16, 1): "int16", (16, 0): "uint16", (32, 1): "int32", (32, 0): "uint32", (64, 1): "int64", (64, 0): "uint64"}
_NP = {"bool": np.uint8, "int8": np.int8, "int16": np.int16, "int32": np.int32, "int64": np.int64, "uint8": np.uint8, "uint16": np.uint16,
       "uint32": np.uint32, "uint64": np.uint64, "float": np.float32, "double": np.float64, "timestamp": np.int64, "duration": np.int64}


def _field_dictionary(fb: _Fb, f: int):
    """(dictionary id, code type) of a dictionary-encoded field (Schema.fbs: Field.dictionary = DictionaryEncoding {id,
    indexType, isOrdered, dictionaryKind}; an absent indexType is int32), else None."""
    enc = fb.table(f, 4)
    if not enc:
        return None
    it = fb.table(enc, 1)
    return fb.scalar(enc, 0, "q"), (_INT[(fb.scalar(it, 0, "i"), fb.scalar(it, 1, "B"))] if it else "int32")


def _field_type(fb: _Fb, f: int) -> str:
    tt = fb.scalar(f, 2, "B")
    t = fb.table(f, 3)
    if tt == 2:
        return _INT[(fb.scalar(t, 0, "i"), fb.scalar(t, 1, "B"))]
    if tt == 3:
        return {1: "float", 2: "double"}[fb.scalar(t, 0, "h")]
    return {4: "binary", 5: "string", 6: "bool", 7: "decimal128", 10: "timestamp", 18: "duration"}[tt]


def read_blob(blob: bytes, schema: Optional[List[Tuple[str, str]]] = None) -> Dict[str, dict]:
    """{column name: {"type", "values" | ("offsets", "chars"), "valid"}}; `schema` = [(name, type)] for a lone
    RecordBatch message."""
    pos = 0
    fields = list(schema) if schema else None
    encodings = [None] * len(fields) if fields else []
    dictionaries: Dict[int, dict] = {}

    def buffers_of(fb, body, record_batch):
        """A reader of the buffers a RecordBatch table lists, one after the other, decoded."""
        bufs, _ = fb.vector(record_batch, 2)
        comp = fb.table(record_batch, 3)
        codec = fb.scalar(comp, 0, "b") if comp else -1
        if codec == 1:
            raise ValueError("Unsupported: ZSTD")
        state = {"i": 0}

        def take():
            off, ln = struct.unpack_from("<qq", fb.b, bufs + 16 * state["i"])
            state["i"] += 1
            raw = blob[body + off: body + off + ln]
            if codec == -1 or ln == 0:
                return raw
            ulen, = struct.unpack_from("<q", raw, 0)
            return raw[8:] if ulen == -1 else lz4_frame_decode(raw[8:], ulen)
        return take

    while True:
        marker, = struct.unpack_from("<I", blob, pos)
        if marker == 0xFFFFFFFF:
            meta_len, = struct.unpack_from("<I", blob, pos + 4)
            pos += 8
        else:
            meta_len = marker
            pos += 4
        if meta_len == 0:
            raise ValueError("no RecordBatch message")
        fb = _Fb(blob[pos:pos + meta_len])
        msg = fb.root()
        header_type = fb.scalar(msg, 1, "B")
        header = fb.table(msg, 2)
        body_len = fb.scalar(msg, 3, "q")
        body = pos + meta_len
        if header_type == 1:
            p, n = fb.vector(header, 1)
            fields, encodings = [], []
            for i in range(n):
                f = p + 4 * i + fb.u("I", p + 4 * i)
                fields.append((fb.string(f, 0), _field_type(fb, f)))
                encodings.append(_field_dictionary(fb, f))
            pos = body + body_len
            continue
        if header_type == 2:  # DictionaryBatch {id, data: RecordBatch of one string column, isDelta}
            data = fb.table(header, 1)
            if fb.scalar(header, 2, "B"):
                raise ValueError("Unsupported: delta dictionary")
            entries = fb.scalar(data, 0, "q")
            take = buffers_of(fb, body, data)
            take()  # validity: dictionaries with null entries are not decoded
            offsets = take()
            dictionaries[fb.scalar(header, 0, "q")] = {
                "offsets": np.frombuffer(offsets, dtype=np.int32, count=entries + 1).copy() if entries or len(offsets) >= 4 else np.zeros(1, np.int32),
                "chars": bytes(take())}
            pos = body + body_len
            continue
        if header_type != 3:
            raise ValueError(f"unexpected message type {header_type}")
        rows = fb.scalar(header, 0, "q")
        nodes, nnodes = fb.vector(header, 1)
        take = buffers_of(fb, body, header)

        out: Dict[str, dict] = {}
        for i in range(nnodes):
            name, typ = fields[i]
            length, nulls = struct.unpack_from("<qq", fb.b, nodes + 16 * i)
            vbits = take()
            valid = np.ones(length, dtype=np.bool_)
            if nulls > 0:
                valid = np.unpackbits(np.frombuffer(vbits, dtype=np.uint8), bitorder="little")[:length].astype(np.bool_)
            col = {"type": typ, "valid": valid}
            data = take()
            if i < len(encodings) and encodings[i] is not None:
                # a dictionary-encoded column read as the column it stands for (dictionary/conversion.h:13-15,
                # DictionaryToArray): every row the entry its code names, a null row the empty string
                d = dictionaries[encodings[i][0]]
                codes = np.frombuffer(data, dtype=_NP[encodings[i][1]], count=length).astype(np.int64)
                strs = [d["chars"][d["offsets"][c]:d["offsets"][c + 1]] if ok else b"" for c, ok in zip(codes, valid)]
                col["offsets"] = np.concatenate([[0], np.cumsum([len(x) for x in strs])]).astype(np.int32)
                col["chars"] = b"".join(strs)
                col["codes"] = codes
                col["entries"] = len(d["offsets"]) - 1
            elif typ in ("string", "binary"):
                col["offsets"] = np.frombuffer(data, dtype=np.int32, count=length + 1).copy() if length or len(data) >= 4 else np.zeros(1, np.int32)
                col["chars"] = bytes(take())
            elif typ == "bool":
                col["values"] = np.unpackbits(np.frombuffer(data, dtype=np.uint8), bitorder="little")[:length].astype(np.uint8)
            elif typ == "decimal128":
                col["values"] = np.frombuffer(data, dtype=np.uint64, count=2 * length).copy()
            else:
                col["values"] = np.frombuffer(data, dtype=_NP[typ], count=length).copy()
            out[name] = col
        return out
