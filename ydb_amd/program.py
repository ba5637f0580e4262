"""Host-side mirror of the reference's program objects over the C-ABI.

`ProgramContainer` follows NOlap::TProgramContainer (ydb/core/tx/program/program.h:55-127):
`Init(resolver, serialized)` parses a TOlapProgram / TProgram and returns (ok, error);
`ApplyProgram(batch)` runs NSsa::TProgram::ApplyTo on the GPU.  The error behaviour matches the
reference: parse failures are soft (False + message), execution failures raise `SsaError` carrying the
arrow::Status-like code.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib
from .arrow_c import ArrowArray, ArrowSchema, Col, ExportedBatch, HostBatch, ResultColumn, read_batch
from .ssa_pb import Program, olap_program


class SsaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{_lib.STATUS_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class DeviceBatch:
    """Record batch resident in HBM (ssa_batch_t)."""

    def __init__(self, lib, handle, keep=None):
        self._lib = lib
        self._h = handle
        self._keep = keep

    @property
    def rows(self) -> int:
        return self._lib.ssa_batch_rows(self._h)

    @property
    def num_columns(self) -> int:
        return self._lib.ssa_batch_columns(self._h)

    def close(self):
        if self._h:
            self._lib.ssa_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


ColumnResolver = Sequence[Tuple[int, str]]  # (column id, column name), the IColumnResolver of the reference


class ProgramContainer:
    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._h = None
        self._device = device

    # -- TProgramContainer::Init -------------------------------------------------------------------
    def Init(self, resolver: ColumnResolver, program: Union[Program, bytes], olap_wrapper: Optional[bool] = None,
             parameters: Optional[HostBatch] = None, kernel_ops: Optional[Sequence[int]] = None) -> Tuple[bool, str]:
        """`kernel_ops`: what each KernelIdx of the program's YQL kernels stands for (ssa_program_create_with_kernels):
        ssa_pb.KERNEL_UNARY_BASE + YQL_UNARY.x, a YQL.x value, KERNEL_IF or KERNEL_UNKNOWN."""
        direct = False
        if isinstance(program, Program):
            data = olap_program(program)
            wrapper = True
            direct = program.direct
        else:
            data = bytes(program)
            wrapper = True if olap_wrapper is None else olap_wrapper
        ids = (ctypes.c_uint32 * max(1, len(resolver)))(*[i for i, _ in resolver])
        names_b = [n.encode() for _, n in resolver]
        names = (ctypes.c_char_p * max(1, len(resolver)))(*names_b)
        out = ctypes.c_void_p()
        err = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(data, len(data))
        pa = ctypes.byref(parameters.array) if parameters is not None else None
        ps = ctypes.byref(parameters.schema) if parameters is not None else None
        flags = (1 if wrapper else 0) | (2 if direct else 0)
        if kernel_ops is None:
            rc = self._lib.ssa_program_create(ctypes.cast(buf, ctypes.c_void_p), len(data), flags, ids, names,
                                              len(resolver), pa, ps, self._device, ctypes.byref(out), ctypes.byref(err))
        else:
            ops = (ctypes.c_uint32 * max(1, len(kernel_ops)))(*kernel_ops)
            rc = self._lib.ssa_program_create_with_kernels(ctypes.cast(buf, ctypes.c_void_p), len(data), flags, ids, names, len(resolver), pa, ps,
                                                           ops, len(kernel_ops), self._device, ctypes.byref(out), ctypes.byref(err))
        msg = _lib.take_string(self._lib, err.value)
        if rc != _lib.SSA_OK:
            if rc != _lib.SSA_PARSE_ERROR:
                raise SsaError(rc, msg)
            return False, msg
        self.close()
        self._h = out.value
        return True, ""

    # -- CH::GroupByOptions ("ch.group_by") ------------------------------------------------------------
    def InitGroupBy(self, assigns: Sequence[Tuple[int, str, Optional[str]]]) -> Tuple[bool, str]:
        """`assigns`: (CH::AggFunctionId, result column, argument column or None) in the order of the result; function 0
        marks a key (ssa_group_by_create)."""
        arr = (_lib.GroupByAssign * max(1, len(assigns)))()
        keep = []
        for i, (fn, res, arg) in enumerate(assigns):
            keep.append((res.encode(), arg.encode() if arg is not None else None))
            arr[i].function = fn
            arr[i].result_column = keep[-1][0]
            arr[i].argument = keep[-1][1]
        out = ctypes.c_void_p()
        err = ctypes.c_void_p()
        rc = self._lib.ssa_group_by_create(arr, len(assigns), self._device, ctypes.byref(out), ctypes.byref(err))
        msg = _lib.take_string(self._lib, err.value)
        if rc != _lib.SSA_OK:
            if rc == _lib.SSA_HIP_ERROR:
                raise SsaError(rc, msg)
            return False, msg
        self.close()
        self._h = out.value
        return True, ""

    def HasProgram(self) -> bool:
        return self._h is not None

    def DebugString(self) -> str:
        return _lib.take_string(self._lib, self._lib.ssa_program_debug_string(self._h))

    def GetSourceColumns(self) -> List[str]:
        s = _lib.take_string(self._lib, self._lib.ssa_program_source_columns(self._h))
        return [x for x in s.split("\n") if x]

    def GetEarlyFilterColumns(self) -> List[str]:
        s = _lib.take_string(self._lib, self._lib.ssa_program_early_filter_columns(self._h))
        return [x for x in s.split("\n") if x]

    def StepCount(self) -> int:
        return self._lib.ssa_program_step_count(self._h)

    def Explain(self, batch: Union[HostBatch, Dict[str, object]]) -> str:
        """Scan plans + output schema for this input schema (host only, no GPU needed)."""
        hb = batch if isinstance(batch, HostBatch) else HostBatch(batch)
        err = ctypes.c_void_p()
        ptr = self._lib.ssa_program_explain(self._h, ctypes.byref(hb.schema), ctypes.byref(err))
        msg = _lib.take_string(self._lib, err.value)
        if not ptr:
            raise SsaError(_lib.SSA_INVALID if "nsupported" not in msg else _lib.SSA_UNSUPPORTED, msg)
        return _lib.take_string(self._lib, ptr)

    # -- TProgram::ApplyTo ---------------------------------------------------------------------------
    def _check(self, rc: int, err: ctypes.c_void_p):
        msg = _lib.take_string(self._lib, err.value)
        if rc != _lib.SSA_OK:
            raise SsaError(rc, msg)

    def ApplyProgram(self, batch: Union[HostBatch, Dict[str, object]]) -> Dict[str, ResultColumn]:
        """Host batch in, host batch out (the drop-in seam, PCIe crossing included)."""
        hb = batch if isinstance(batch, HostBatch) else HostBatch(batch)
        out_a, out_s, err = ArrowArray(), ArrowSchema(), ctypes.c_void_p()
        rc = self._lib.ssa_program_apply(self._h, ctypes.byref(hb.array), ctypes.byref(hb.schema), ctypes.byref(out_a),
                                         ctypes.byref(out_s), ctypes.byref(err))
        self._check(rc, err)
        return read_batch(out_a, out_s)

    def BuildFilter(self, batch: Union[HostBatch, Dict[str, object]], step: int = 0) -> Tuple[np.ndarray, int]:
        """TProgramStep::BuildFilter: boolean mask of the rows that pass the step's filters."""
        hb = batch if isinstance(batch, HostBatch) else HostBatch(batch)
        bits = np.zeros((hb.rows + 7) // 8 + 8, dtype=np.uint8)
        selected = ctypes.c_int64(0)
        err = ctypes.c_void_p()
        rc = self._lib.ssa_program_build_filter(self._h, step, ctypes.byref(hb.array), ctypes.byref(hb.schema),
                                                bits.ctypes.data, ctypes.byref(selected), ctypes.byref(err))
        self._check(rc, err)
        mask = np.unpackbits(bits, bitorder="little")[:hb.rows].astype(np.bool_)
        return mask, selected.value

    # -- HBM-resident path ----------------------------------------------------------------------------
    def Import(self, batch: Union[HostBatch, Dict[str, object]]) -> DeviceBatch:
        hb = batch if isinstance(batch, HostBatch) else HostBatch(batch)
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._lib.ssa_batch_import(self._h, ctypes.byref(hb.array), ctypes.byref(hb.schema), ctypes.byref(out),
                                        ctypes.byref(err))
        self._check(rc, err)
        return DeviceBatch(self._lib, out.value)

    def ImportBlobs(self, blobs: Sequence[bytes], schema: Optional[HostBatch] = None) -> List[DeviceBatch]:
        """TNativeSerializer::DoDeserialize on the GPU: IPC blobs (whole streams, or lone RecordBatch messages read with
        `schema`: a HostBatch whose schema is the chunk's) -> resident batches, one launch for all compressed buffers."""
        n = len(blobs)
        # (the library reads the blobs where they lie: a bytes object's buffer is handed over as it is, anything else
        # through one copy)
        keep = [ctypes.c_char_p(b) if isinstance(b, bytes) else ctypes.create_string_buffer(bytes(b), len(b)) for b in blobs]
        ptrs = (ctypes.c_void_p * max(1, n))(*[ctypes.cast(k, ctypes.c_void_p).value for k in keep])
        sizes = (ctypes.c_size_t * max(1, n))(*[len(b) for b in blobs])
        outs = (ctypes.c_void_p * max(1, n))()
        err = ctypes.c_void_p()
        rc = self._lib.ssa_batches_from_blobs(self._h, ptrs, sizes, n, ctypes.byref(schema.schema) if schema is not None else None, outs,
                                              ctypes.byref(err))
        self._check(rc, err)
        return [DeviceBatch(self._lib, outs[i]) for i in range(n)]

    def ImportBlob(self, blob: bytes, schema: Optional[HostBatch] = None) -> DeviceBatch:
        return self.ImportBlobs([blob], schema)[0]

    def ApplyDevice(self, batch: DeviceBatch) -> DeviceBatch:
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._lib.ssa_program_apply_device(self._h, batch._h, ctypes.byref(out), ctypes.byref(err))
        self._check(rc, err)
        return DeviceBatch(self._lib, out.value)

    def ApplyDeviceMany(self, batches: Sequence[DeviceBatch]) -> List[DeviceBatch]:
        """The program on each of the batches (slices of a table), one result per slice (program.h:423-433 of the
        reference runs them one by one); slices of one schema under an aggregation are scanned by one launch."""
        n = len(batches)
        ins = (ctypes.c_void_p * max(n, 1))(*[b._h for b in batches])
        outs = (ctypes.c_void_p * max(n, 1))()
        err = ctypes.c_void_p()
        rc = self._lib.ssa_program_apply_many(self._h, ins, n, outs, ctypes.byref(err))
        self._check(rc, err)
        return [DeviceBatch(self._lib, outs[i]) for i in range(n)]

    # -- result batches between GPUs (the collective itself is the caller's: torch.distributed over RCCL) -------
    def PackLayout(self, batch: DeviceBatch) -> Tuple[bytes, int]:
        """(header, data_bytes): the description of `batch` as one flat buffer and the size of its data."""
        hb, db, err = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_void_p()
        rc = self._lib.ssa_batch_pack_layout(self._h, batch._h, None, 0, ctypes.byref(hb), ctypes.byref(db), ctypes.byref(err))
        self._check(rc, err)
        buf = (ctypes.c_uint8 * max(hb.value, 1))()
        rc = self._lib.ssa_batch_pack_layout(self._h, batch._h, buf, hb.value, ctypes.byref(hb), ctypes.byref(db), ctypes.byref(err))
        self._check(rc, err)
        return bytes(buf[:hb.value]), db.value

    def PackData(self, batch: DeviceBatch, device_ptr: int, capacity: int) -> None:
        err = ctypes.c_void_p()
        self._check(self._lib.ssa_batch_pack_data(self._h, batch._h, device_ptr, capacity, ctypes.byref(err)), err)

    def Unpack(self, header: bytes, device_ptr: int, data_bytes: int, keep=None) -> DeviceBatch:
        """A batch of views of the device memory at `device_ptr` (kept alive through `keep`)."""
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        hb = (ctypes.c_uint8 * max(len(header), 1)).from_buffer_copy(header or b"\0")
        self._check(self._lib.ssa_batch_unpack(hb, len(header), device_ptr, data_bytes, ctypes.byref(out), ctypes.byref(err)), err)
        return DeviceBatch(self._lib, out.value, keep=keep)

    def Concat(self, batches: Sequence[DeviceBatch]) -> DeviceBatch:
        n = len(batches)
        ins = (ctypes.c_void_p * max(n, 1))(*[b._h for b in batches])
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.ssa_batch_concat(self._h, ins, n, ctypes.byref(out), ctypes.byref(err)), err)
        return DeviceBatch(self._lib, out.value)

    def Slice(self, batch: DeviceBatch, row0: int, rows: int) -> DeviceBatch:
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.ssa_batch_slice(batch._h, row0, rows, ctypes.byref(out), ctypes.byref(err)), err)
        return DeviceBatch(self._lib, out.value, keep=batch)

    def Rename(self, batch: DeviceBatch, names: Sequence[str]) -> DeviceBatch:
        arr = (ctypes.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        self._check(self._lib.ssa_batch_rename(batch._h, arr, len(names), ctypes.byref(out), ctypes.byref(err)), err)
        return DeviceBatch(self._lib, out.value, keep=batch)

    def ColumnNames(self, batch: DeviceBatch) -> List[str]:
        return [_lib.take_string(self._lib, self._lib.ssa_batch_column_name(batch._h, i)) for i in range(batch.num_columns)]

    def Export(self, batch: DeviceBatch, lazy: bool = False) -> Dict[str, ResultColumn]:
        """The batch on the host.  `lazy`: as the Arrow structs ssa_batch_export filled (what a C++ caller holds), turned
        into numpy arrays only when a column is looked at."""
        out_a, out_s, err = ArrowArray(), ArrowSchema(), ctypes.c_void_p()
        rc = self._lib.ssa_batch_export(self._h, batch._h, ctypes.byref(out_a), ctypes.byref(out_s), ctypes.byref(err))
        self._check(rc, err)
        return ExportedBatch(out_a, out_s) if lazy else read_batch(out_a, out_s)

    # -- compute sharding of a result batch (THashConstructor::BuildHashUI64 + TShardingSplitIndex) ------------
    @staticmethod
    def _names(keys: Sequence[str]):
        arr = (ctypes.c_char_p * max(1, len(keys)))()
        for i, k in enumerate(keys):
            arr[i] = k.encode()
        return arr

    def ShardingHash(self, batch: DeviceBatch, keys: Sequence[str], seed: int = 0) -> np.ndarray:
        """NHash::TXX64::Execute: XXH64(seed) per row over the non-null key fields."""
        out = np.zeros(max(1, batch.rows), dtype=np.uint64)
        err = ctypes.c_void_p()
        rc = self._lib.ssa_sharding_hash(self._h, batch._h, self._names(keys), len(keys), seed, out.ctypes.data, ctypes.byref(err))
        self._check(rc, err)
        return out[:batch.rows]

    def ShardingSplit(self, batch: DeviceBatch, keys: Sequence[str], shards: int) -> Tuple[DeviceBatch, List[int]]:
        """The batch with its rows grouped shard after shard, and the number of rows of every shard."""
        sizes = np.zeros(max(1, shards), dtype=np.uint32)
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._lib.ssa_sharding_split(self._h, batch._h, self._names(keys), len(keys), shards, ctypes.byref(out),
                                          sizes.ctypes.data, ctypes.byref(err))
        self._check(rc, err)
        return DeviceBatch(self._lib, out.value), [int(x) for x in sizes[:shards]]

    def PartitionByKey(self, batch: DeviceBatch, keys: Sequence[str], parts: int) -> Tuple[DeviceBatch, List[int]]:
        """The batch with its rows grouped part after part by the hash of their keys alone (no balancing of the parts:
        equal keys are in the same part whichever batch they come from), and the number of rows of every part."""
        sizes = np.zeros(max(1, parts), dtype=np.uint32)
        out, err = ctypes.c_void_p(), ctypes.c_void_p()
        rc = self._lib.ssa_batch_partition_by_key(self._h, batch._h, self._names(keys), len(keys), parts, ctypes.byref(out),
                                                  sizes.ctypes.data, ctypes.byref(err))
        self._check(rc, err)
        return DeviceBatch(self._lib, out.value), [int(x) for x in sizes[:parts]]

    def EnableProfile(self, enabled: bool = True):
        self._lib.ssa_profile_enable(self._h, 1 if enabled else 0)

    def ReadProfile(self) -> Tuple[float, int, int]:
        ms, n, b = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._lib.ssa_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b))
        return ms.value, n.value, b.value

    def SpecializationStats(self) -> Tuple[int, float, int]:
        """Process-wide: scan kernels built at run time, host milliseconds spent building, launches that ran one."""
        built, ms, n = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_int64(0)
        self._lib.ssa_specialization_stats(ctypes.byref(built), ctypes.byref(ms), ctypes.byref(n))
        return built.value, ms.value, n.value

    def close(self):
        if self._h:
            self._lib.ssa_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_batch(columns: Sequence[Tuple[str, int, int, int, int, int, int]], rows: int, keep=None) -> DeviceBatch:
    """Wraps device pointers: (name, ssa_type, values_ptr, validity_ptr, chars_ptr, chars_size, null_count) and, for a
    dictionary-encoded string column, three more: (..., index_type, dict_offsets_ptr, dict_size) with values_ptr = the
    codes and chars_ptr = the dictionary's bytes."""
    lib = _lib.load()
    arr = (_lib.DeviceColumn * len(columns))()
    names = []
    for i, column in enumerate(columns):
        name, typ, values, validity, chars, chars_size, null_count = column[:7]
        if len(column) > 7:
            arr[i].index_type, arr[i].dict_offsets, arr[i].dict_size = column[7], column[8], column[9]
        nb = name.encode()
        names.append(nb)
        arr[i].name = nb
        arr[i].type = typ
        arr[i].values = values
        arr[i].validity = validity or None
        arr[i].chars = chars or None
        arr[i].chars_size = chars_size
        arr[i].null_count = null_count
    out, err = ctypes.c_void_p(), ctypes.c_void_p()
    rc = lib.ssa_batch_from_device(arr, len(columns), rows, ctypes.byref(out), ctypes.byref(err))
    msg = _lib.take_string(lib, err.value)
    if rc != _lib.SSA_OK:
        raise SsaError(rc, msg)
    return DeviceBatch(lib, out.value, keep=(keep, names))
