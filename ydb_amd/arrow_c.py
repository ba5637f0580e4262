"""numpy <-> Arrow C Data Interface, without pyarrow.

The C-ABI of the executor (include/ydb_ssa_hip.h) exchanges record batches as
`struct ArrowArray` / `struct ArrowSchema` (the layout published by Apache Arrow, also vendored by
the reference under contrib/libs/apache/arrow/cpp/src/arrow/c/abi.h).  This module builds those
structs from numpy arrays for inputs and reads them back into numpy arrays for results, so that the
tests and the benchmark exercise exactly the boundary a C++ host shim would use.
"""
from __future__ import annotations

import ctypes
from collections.abc import Mapping
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np


class ArrowSchema(ctypes.Structure):
    pass


class ArrowArray(ctypes.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", ctypes.c_char_p),
    ("name", ctypes.c_char_p),
    ("metadata", ctypes.c_char_p),
    ("flags", ctypes.c_int64),
    ("n_children", ctypes.c_int64),
    ("children", ctypes.POINTER(ctypes.POINTER(ArrowSchema))),
    ("dictionary", ctypes.POINTER(ArrowSchema)),
    ("release", ctypes.CFUNCTYPE(None, ctypes.POINTER(ArrowSchema))),
    ("private_data", ctypes.c_void_p),
]

ArrowArray._fields_ = [
    ("length", ctypes.c_int64),
    ("null_count", ctypes.c_int64),
    ("offset", ctypes.c_int64),
    ("n_buffers", ctypes.c_int64),
    ("n_children", ctypes.c_int64),
    ("buffers", ctypes.POINTER(ctypes.c_void_p)),
    ("children", ctypes.POINTER(ctypes.POINTER(ArrowArray))),
    ("dictionary", ctypes.POINTER(ArrowArray)),
    ("release", ctypes.CFUNCTYPE(None, ctypes.POINTER(ArrowArray))),
    ("private_data", ctypes.c_void_p),
]

# type name -> (Arrow format string, numpy dtype of the value buffer)
_TYPES = {
    "bool": ("b", np.uint8),
    "int8": ("c", np.int8),
    "uint8": ("C", np.uint8),
    "int16": ("s", np.int16),
    "uint16": ("S", np.uint16),
    "int32": ("i", np.int32),
    "uint32": ("I", np.uint32),
    "int64": ("l", np.int64),
    "uint64": ("L", np.uint64),
    "float": ("f", np.float32),
    "double": ("g", np.float64),
    "timestamp": ("tsu:", np.int64),
    "duration": ("tDu", np.int64),
    "string": ("u", None),
    "binary": ("z", None),
}
_FORMAT_TO_TYPE = {v[0]: k for k, v in _TYPES.items()}

_NUMPY_TO_TYPE = {
    np.dtype(np.bool_): "bool", np.dtype(np.int8): "int8", np.dtype(np.uint8): "uint8",
    np.dtype(np.int16): "int16", np.dtype(np.uint16): "uint16", np.dtype(np.int32): "int32",
    np.dtype(np.uint32): "uint32", np.dtype(np.int64): "int64", np.dtype(np.uint64): "uint64",
    np.dtype(np.float32): "float", np.dtype(np.float64): "double",
}


@dataclass
class Col:
    """One column of a host batch.

    values: numpy array (fixed width / bool) or a sequence of str / bytes / None (string, binary).
    valid:  optional boolean mask, True = non-null.  For string sequences, None entries are nulls.
    type:   one of _TYPES; inferred from the numpy dtype (or "string"/"binary" from the first element).
    offset: number of leading elements of `values` that sit *before* row 0 (models a sliced Arrow array:
            ArrowArray.offset != 0, as produced by NArrow::SliceToRecordBatches in the reference).
    dictionary: optional sequence of str/bytes: `values` are then integer codes into it.
    """
    values: object
    valid: Optional[np.ndarray] = None
    type: Optional[str] = None
    offset: int = 0
    dictionary: Optional[Sequence] = None
    decimal: Optional[Tuple[int, int]] = None  # (precision, scale): values is an (n, 2) uint64 array [lo, hi]


def pack_bits(mask: np.ndarray) -> np.ndarray:
    """bool array -> Arrow bitmap (LSB first)."""
    mask = np.asarray(mask, dtype=np.bool_)
    return np.packbits(mask, bitorder="little")


def unpack_bits(buf: np.ndarray, n: int, offset: int = 0) -> np.ndarray:
    bits = np.unpackbits(np.asarray(buf, dtype=np.uint8), bitorder="little")
    return bits[offset:offset + n].astype(np.bool_)


def _encode_strings(values: Sequence) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    offsets = np.zeros(len(values) + 1, dtype=np.int32)
    chunks = []
    valid = np.ones(len(values), dtype=np.bool_)
    pos = 0
    for i, v in enumerate(values):
        if v is None:
            valid[i] = False
            b = b""
        elif isinstance(v, str):
            b = v.encode()
        else:
            b = bytes(v)
        chunks.append(b)
        pos += len(b)
        offsets[i + 1] = pos
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if pos else np.zeros(0, dtype=np.uint8)
    return offsets, data, valid


class HostBatch:
    """Owns the ArrowArray / ArrowSchema structs of one record batch and every buffer they point to."""

    def __init__(self, columns: Dict[str, Union[Col, np.ndarray, Sequence]], rows: Optional[int] = None):
        self._keep: List[object] = []
        self.names = list(columns.keys())
        cols = [c if isinstance(c, Col) else Col(c) for c in columns.values()]
        child_arrays, child_schemas = [], []
        nrows = rows
        for name, c in zip(self.names, cols):
            arr, sch, n = self._build_column(name, c)
            child_arrays.append(arr)
            child_schemas.append(sch)
            if nrows is None:
                nrows = n
            elif n != nrows:
                raise ValueError(f"column {name!r} has {n} rows, expected {nrows}")
        self.rows = nrows or 0
        self.schema = self._schema(b"+s", b"", child_schemas)
        self.array = ArrowArray()
        self.array.length = self.rows
        self.array.null_count = 0
        self.array.offset = 0
        bufs = (ctypes.c_void_p * 1)(None)
        self._keep.append(bufs)
        self.array.n_buffers = 1
        self.array.buffers = ctypes.cast(bufs, ctypes.POINTER(ctypes.c_void_p))
        self.array.n_children = len(child_arrays)
        ptrs = (ctypes.POINTER(ArrowArray) * max(1, len(child_arrays)))(*[ctypes.pointer(a) for a in child_arrays])
        self._keep.append(ptrs)
        self._keep.append(child_arrays)
        self.array.children = ctypes.cast(ptrs, ctypes.POINTER(ctypes.POINTER(ArrowArray)))

    # -- helpers ---------------------------------------------------------------------------------
    def _schema(self, fmt: bytes, name: bytes, children=(), dictionary=None) -> ArrowSchema:
        s = ArrowSchema()
        s.format = fmt
        s.name = name
        s.metadata = None
        s.flags = 2
        s.n_children = len(children)
        if children:
            ptrs = (ctypes.POINTER(ArrowSchema) * len(children))(*[ctypes.pointer(c) for c in children])
            self._keep.append(ptrs)
            self._keep.append(list(children))
            s.children = ctypes.cast(ptrs, ctypes.POINTER(ctypes.POINTER(ArrowSchema)))
        if dictionary is not None:
            self._keep.append(dictionary)
            s.dictionary = ctypes.pointer(dictionary)
        self._keep.append((fmt, name))
        return s

    def _array(self, length: int, null_count: int, offset: int, buffers: List[Optional[np.ndarray]], dictionary=None) -> ArrowArray:
        a = ArrowArray()
        a.length = length
        a.null_count = null_count
        a.offset = offset
        a.n_buffers = len(buffers)
        raw = (ctypes.c_void_p * len(buffers))()
        for i, b in enumerate(buffers):
            if b is None:
                raw[i] = None
            else:
                self._keep.append(b)
                raw[i] = b.ctypes.data if b.size else None
                if b.size == 0:
                    pad = np.zeros(8, dtype=np.uint8)  # Arrow allows non-null pointers to empty buffers
                    self._keep.append(pad)
                    raw[i] = pad.ctypes.data
        self._keep.append(raw)
        a.buffers = ctypes.cast(raw, ctypes.POINTER(ctypes.c_void_p))
        a.n_children = 0
        if dictionary is not None:
            self._keep.append(dictionary)
            a.dictionary = ctypes.pointer(dictionary)
        return a

    def _build_column(self, name: str, c: Col):
        off = c.offset
        values = c.values
        tname = c.type
        if c.dictionary is not None:
            codes = np.ascontiguousarray(values)
            n = len(codes) - off
            doffs, ddata, _ = _encode_strings(list(c.dictionary))
            dtype_name = tname or "string"
            dict_arr = self._array(len(c.dictionary), 0, 0, [None, doffs, ddata])
            dict_sch = self._schema(_TYPES[dtype_name][0].encode(), b"")
            valid = c.valid
            vbuf, nulls = self._validity(valid, len(codes), off)
            arr = self._array(n, nulls, off, [vbuf, codes], dictionary=dict_arr)
            idx_fmt = _TYPES[_NUMPY_TO_TYPE[codes.dtype]][0].encode()
            sch = self._schema(idx_fmt, name.encode(), dictionary=dict_sch)
            return arr, sch, n
        if c.decimal is not None:
            vals = np.ascontiguousarray(values, dtype=np.uint64)
            n = vals.shape[0] - off
            vbuf, nulls = self._validity(c.valid, vals.shape[0], off)
            arr = self._array(n, nulls, off, [vbuf, vals.reshape(-1)])
            sch = self._schema(f"d:{c.decimal[0]},{c.decimal[1]}".encode(), name.encode())
            return arr, sch, n
        if isinstance(values, np.ndarray) and values.dtype.kind in "biuf":
            if tname is None:
                tname = _NUMPY_TO_TYPE[values.dtype]
            fmt, dt = _TYPES[tname]
            total = len(values)
            n = total - off
            vbuf, nulls = self._validity(c.valid, total, off)
            if tname == "bool":
                data = pack_bits(values.astype(np.bool_))
            else:
                data = np.ascontiguousarray(values.astype(dt, copy=False))
            arr = self._array(n, nulls, off, [vbuf, data])
            sch = self._schema(fmt.encode(), name.encode())
            return arr, sch, n
        if isinstance(values, np.ndarray) and values.dtype.kind in "Mm":
            tname = tname or ("timestamp" if values.dtype.kind == "M" else "duration")
            data = np.ascontiguousarray(values.astype("datetime64[us]" if values.dtype.kind == "M" else "timedelta64[us]").astype(np.int64))
            total = len(data)
            n = total - off
            vbuf, nulls = self._validity(c.valid, total, off)
            arr = self._array(n, nulls, off, [vbuf, data])
            sch = self._schema(_TYPES[tname][0].encode(), name.encode())
            return arr, sch, n
        # strings / binary
        seq = list(values)
        if tname is None:
            first = next((v for v in seq if v is not None), "")
            tname = "string" if isinstance(first, str) else "binary"
        offsets, data, valid = _encode_strings(seq)
        if c.valid is not None:
            valid = valid & np.asarray(c.valid, dtype=np.bool_)
        total = len(seq)
        n = total - off
        vbuf, nulls = self._validity(None if valid.all() else valid, total, off)
        arr = self._array(n, nulls, off, [vbuf, offsets, data])
        sch = self._schema(_TYPES[tname][0].encode(), name.encode())
        return arr, sch, n

    @staticmethod
    def _validity(valid, total: int, off: int):
        if valid is None:
            return None, 0
        valid = np.asarray(valid, dtype=np.bool_)
        if len(valid) != total:
            raise ValueError("validity mask length mismatch")
        nulls = int((~valid[off:]).sum())
        return pack_bits(valid), nulls


@dataclass
class ResultColumn:
    name: str
    type: str
    values: object          # numpy array (fixed width), or list of bytes (strings)
    valid: np.ndarray       # bool mask
    decimal: Optional[Tuple[int, int]] = None

    def to_pylist(self):
        out = []
        for i in range(len(self.valid)):
            if not self.valid[i]:
                out.append(None)
            else:
                v = self.values[i]
                if self.type == "decimal128":  # two's-complement 128-bit integer from its (lo, hi) words
                    x = int(v[0]) | (int(v[1]) << 64)
                    out.append(x - (1 << 128) if x >> 127 else x)
                    continue
                if isinstance(v, (bytes, bytearray)):
                    out.append(v.decode() if self.type == "string" else bytes(v))
                elif isinstance(v, np.generic):
                    out.append(v.item())
                else:
                    out.append(v)
        return out


def _buffer(ptr, nbytes: int) -> np.ndarray:
    if not ptr or nbytes == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).copy()


def read_batch(array: ArrowArray, schema: ArrowSchema, release: bool = True) -> Dict[str, ResultColumn]:
    """Copies an exported struct array into numpy and (optionally) runs the release callbacks."""
    out: Dict[str, ResultColumn] = {}
    order: List[str] = []
    for i in range(array.n_children):
        a = array.children[i].contents
        s = schema.children[i].contents
        name = s.name.decode() if s.name else ""
        fmt = s.format.decode()
        n, off = a.length, a.offset
        if a.n_buffers > 0 and a.buffers[0]:
            valid = unpack_bits(_buffer(a.buffers[0], (off + n + 7) // 8), n, off)
        else:
            valid = np.ones(n, dtype=np.bool_)
        dec = None
        if fmt.startswith("d:"):
            p, sc = fmt[2:].split(",")[:2]
            dec = (int(p), int(sc))
            raw = _buffer(a.buffers[1], (off + n) * 16).view(np.uint64).reshape(-1, 2)[off:off + n]
            col = ResultColumn(name, "decimal128", raw, valid, dec)
        elif fmt in ("u", "z"):
            offs = _buffer(a.buffers[1], (off + n + 1) * 4).view(np.int32)[off:off + n + 1]
            total = int(offs[-1]) if n + 1 > 0 and len(offs) else 0
            data = _buffer(a.buffers[2], total).tobytes()
            vals = [data[offs[j]:offs[j + 1]] for j in range(n)]
            col = ResultColumn(name, _FORMAT_TO_TYPE[fmt], vals, valid)
        elif fmt == "b":
            vals = unpack_bits(_buffer(a.buffers[1], (off + n + 7) // 8), n, off)
            col = ResultColumn(name, "bool", vals, valid)
        else:
            key = "tsu:" if fmt.startswith("tsu:") else fmt
            tname = _FORMAT_TO_TYPE[key]
            dt = np.dtype(_TYPES[tname][1])
            vals = _buffer(a.buffers[1], (off + n) * dt.itemsize).view(dt)[off:off + n]
            col = ResultColumn(name, tname, vals, valid)
        key = name
        while key in out:  # duplicate names are legal in a record batch
            key += "'"
        out[key] = col
        order.append(key)
    if release:
        if array.release:
            array.release(ctypes.byref(array))
        if schema.release:
            schema.release(ctypes.byref(schema))
    return out


class ExportedBatch(Mapping):
    """A batch the library exported -- Arrow C data interface structs in host memory, what a C++ caller of
    ssa_batch_export holds -- that reads as the dictionary read_batch makes of it; the numpy copies are made when a
    column is first looked at, and the structs released then (or when the object goes away unread)."""

    def __init__(self, array: ArrowArray, schema: ArrowSchema):
        self._array, self._schema, self._cols = array, schema, None

    def columns(self) -> Dict[str, ResultColumn]:
        if self._cols is None:
            self._cols = read_batch(self._array, self._schema)
            self._array = self._schema = None
        return self._cols

    @property
    def rows(self) -> int:
        if self._cols is None:
            return int(self._array.length)
        return len(next(iter(self._cols.values())).valid) if self._cols else 0

    def __getitem__(self, key):
        return self.columns()[key]

    def __iter__(self):
        return iter(self.columns())

    def __len__(self):
        return len(self.columns())

    def __del__(self):
        if self._cols is None and self._array is not None:
            if self._array.release:
                self._array.release(ctypes.byref(self._array))
            if self._schema.release:
                self._schema.release(ctypes.byref(self._schema))
            self._array = self._schema = None
