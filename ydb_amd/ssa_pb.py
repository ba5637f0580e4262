"""Wire-format encoder for NKikimrSSA::TProgram / TOlapProgram.

The field numbers follow ydb/core/formats/arrow/protos/ssa.proto:19-209 of the
reference (proto2, package NKikimrSSA).  There is no protoc in the build
environment, so the few messages the scan path needs are encoded by hand; the
bytes this module emits are exactly what `TProgram::SerializeToString` would
produce for the same field values (fields are written in field-number order,
varints are minimal).

Only *encoding* lives here: tests, the benchmark and the fixtures use it to
build programs the way the reference's unit tests do with the generated C++
classes (ydb/core/tx/columnshard/ut_rw/ut_columnshard_read_write.cpp:1108-1223).
The library and the oracle each carry their own independent decoder.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union


# --- EFunction (ssa.proto:72-111) -------------------------------------------
class F:
    UNSPECIFIED = 0
    CMP_EQUAL = 1
    CMP_NOT_EQUAL = 2
    CMP_LESS = 3
    CMP_LESS_EQUAL = 4
    CMP_GREATER = 5
    CMP_GREATER_EQUAL = 6
    IS_NULL = 7
    STR_LENGTH = 8
    STR_MATCH = 9
    BINARY_NOT = 10
    BINARY_AND = 11
    BINARY_OR = 12
    BINARY_XOR = 13
    MATH_ADD = 14
    MATH_SUBTRACT = 15
    MATH_MULTIPLY = 16
    MATH_DIVIDE = 17
    CAST_TO_BOOLEAN = 18
    CAST_TO_INT8 = 19
    CAST_TO_INT16 = 20
    CAST_TO_INT32 = 21
    CAST_TO_INT64 = 22
    CAST_TO_UINT8 = 23
    CAST_TO_UINT16 = 24
    CAST_TO_UINT32 = 25
    CAST_TO_UINT64 = 26
    CAST_TO_FLOAT = 27
    CAST_TO_DOUBLE = 28
    CAST_TO_BINARY = 29
    CAST_TO_FIXED_SIZE_BINARY = 30
    CAST_TO_TIMESTAMP = 31
    STR_MATCH_LIKE = 32
    STR_STARTS_WITH = 33
    STR_ENDS_WITH = 34
    STR_MATCH_IGNORE_CASE = 35
    STR_STARTS_WITH_IGNORE_CASE = 36
    STR_ENDS_WITH_IGNORE_CASE = 37


# --- EAggregateFunction (ssa.proto:137-155) ----------------------------------
class AGG:
    UNSPECIFIED = 0
    SOME = 1
    COUNT = 2
    MIN = 3
    MAX = 4
    SUM = 5


# --- EFunctionType (ssa.proto:66-69) -----------------------------------------
SIMPLE_ARROW = 1
YQL_KERNEL = 2


# --- NYql::TKernelRequestBuilder::EBinaryOp
#     (ydb/library/yql/core/arrow_kernels/request/request.h:19-42) -------------
class YQL:
    And = 0
    Or = 1
    Xor = 2
    Add = 3
    Sub = 4
    Mul = 5
    Div = 6
    Mod = 7
    StartsWith = 8
    EndsWith = 9
    StringContains = 10
    Equals = 11
    NotEquals = 12
    Less = 13
    LessOrEqual = 14
    Greater = 15
    GreaterOrEqual = 16
    Coalesce = 17


class YQL_UNARY:
    """NYql::TKernelRequestBuilder::EUnaryOp (request.h:11-17).  KQP sends these by KernelIdx alone; the table that says
    which index is which operation goes to ProgramContainer.Init(kernel_ops=...) as KERNEL_UNARY_BASE + the value."""
    Not = 0
    Size = 1
    Minus = 2
    Abs = 3
    Just = 4


KERNEL_UNARY_BASE = 0x100
KERNEL_IF = 0x10000
KERNEL_UNKNOWN = 0xFFFFFFFF


# --- primitive wire encoders --------------------------------------------------
class OP:
    """NKikimr::NKernels::EOperation (ydb/library/arrow_kernels/operations.h:5-84), by position.  The ones below have no
    TFunction id: a step gets them through TAssign(column, operation, arguments) objects built in code
    (ydb/core/formats/arrow/ut/ut_program_step.cpp:30-41), carried here as TFunction ids of DIRECT_BASE + operation."""
    IsValid = 17
    Abs = 33
    Negate = 34
    Gcd = 35
    Lcm = 36
    Modulo = 37
    ModuloOrZero = 38
    AddNotNull = 39        # add_checked
    SubtractNotNull = 40
    MultiplyNotNull = 41
    DivideNotNull = 42
    Acosh = 48
    Atanh = 49
    Cbrt = 50
    Cosh = 51
    E = 52
    Erf = 53
    Erfc = 54
    Exp = 55
    Exp2 = 56
    Exp10 = 57
    Hypot = 58
    Lgamma = 59
    Pi = 60
    Sinh = 61
    Sqrt = 62
    Tgamma = 63
    Floor = 64
    Ceil = 65
    Trunc = 66
    Round = 67
    RoundBankers = 68
    RoundToExp2 = 69


DIRECT_BASE = 1000


def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64  # negative int32/int64 are sign-extended to 10 bytes
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _tag(field_no: int, wire: int) -> bytes:
    return _varint((field_no << 3) | wire)


def _f_varint(field_no: int, v: int) -> bytes:
    return _tag(field_no, 0) + _varint(v)


def _f_bytes(field_no: int, b: bytes) -> bytes:
    return _tag(field_no, 2) + _varint(len(b)) + b


def _f_fixed32(field_no: int, raw: bytes) -> bytes:
    return _tag(field_no, 5) + raw


def _f_fixed64(field_no: int, raw: bytes) -> bytes:
    return _tag(field_no, 1) + raw


# --- messages -----------------------------------------------------------------
@dataclass
class Column:
    """TProgram.TColumn (ssa.proto:20-23)."""
    id: Optional[int] = None
    name: Optional[str] = None

    def encode(self) -> bytes:
        out = b""
        if self.id is not None:
            out += _f_varint(1, self.id)
        if self.name is not None:
            out += _f_bytes(2, self.name.encode())
        return out


ColLike = Union[Column, int, str]


def col(c: ColLike) -> Column:
    if isinstance(c, Column):
        return c
    if isinstance(c, int):
        return Column(id=c)
    return Column(name=c)


@dataclass
class Constant:
    """TProgram.TConstant (ssa.proto:25-42): exactly one field set."""
    kind: str
    value: object

    _FIELDS = {
        "Bool": 1, "Int32": 2, "Uint32": 3, "Int64": 4, "Uint64": 5, "Float": 6,
        "Double": 7, "Bytes": 8, "Text": 9, "Int8": 10, "Uint8": 11, "Int16": 12,
        "Uint16": 13, "Timestamp": 14,
    }

    def encode(self) -> bytes:
        no = self._FIELDS[self.kind]
        k = self.kind
        if k == "Bool":
            return _f_varint(no, 1 if self.value else 0)
        if k in ("Int32", "Int64", "Int8", "Int16", "Uint32", "Uint64", "Uint8",
                 "Uint16", "Timestamp"):
            return _f_varint(no, int(self.value))
        if k == "Float":
            return _f_fixed32(no, struct.pack("<f", self.value))
        if k == "Double":
            return _f_fixed64(no, struct.pack("<d", self.value))
        if k == "Bytes":
            v = self.value if isinstance(self.value, bytes) else str(self.value).encode()
            return _f_bytes(no, v)
        if k == "Text":
            v = self.value if isinstance(self.value, bytes) else str(self.value).encode()
            return _f_bytes(no, v)
        raise ValueError(k)


@dataclass
class Function:
    """TProgram.TAssignment.TFunction (ssa.proto:113-119)."""
    id: int = 0
    args: Sequence[ColLike] = ()
    function_type: Optional[int] = None
    kernel_idx: Optional[int] = None
    yql_op: Optional[int] = None

    def encode(self) -> bytes:
        out = b""
        if self.id:
            out += _f_varint(1, self.id)
        for a in self.args:
            out += _f_bytes(2, col(a).encode())
        if self.function_type is not None:
            out += _f_varint(3, self.function_type)
        if self.kernel_idx is not None:
            out += _f_varint(4, self.kernel_idx)
        if self.yql_op is not None:
            out += _f_varint(5, self.yql_op)
        return out


@dataclass
class Assign:
    """TProgram.TAssignment (ssa.proto:126-133)."""
    column: Optional[ColLike]
    function: Optional[Function] = None
    constant: Optional[Constant] = None
    null: Optional[bool] = None
    parameter: Optional[str] = None
    external: bool = False

    def encode(self) -> bytes:
        out = b""
        if self.column is not None:
            out += _f_bytes(1, col(self.column).encode())
        if self.function is not None:
            out += _f_bytes(2, self.function.encode())
        if self.external:
            out += _f_bytes(3, b"")
        if self.constant is not None:
            out += _f_bytes(4, self.constant.encode())
        if self.null is not None:
            out += _f_varint(5, 1 if self.null else 0)
        if self.parameter is not None:
            out += _f_bytes(6, _f_bytes(1, self.parameter.encode()))
        return _f_bytes(1, out)  # TCommand.Assign = 1


@dataclass
class Filter:
    """TProgram.TFilter (ssa.proto:174-179)."""
    predicate: ColLike

    def encode(self) -> bytes:
        return _f_bytes(3, _f_bytes(1, col(self.predicate).encode()))  # TCommand.Filter = 3


@dataclass
class Projection:
    """TProgram.TProjection (ssa.proto:170-172)."""
    columns: Sequence[ColLike]

    def encode(self) -> bytes:
        body = b"".join(_f_bytes(1, col(c).encode()) for c in self.columns)
        return _f_bytes(2, body)  # TCommand.Projection = 2


@dataclass
class Aggregate:
    """TProgram.TAggregateAssignment (ssa.proto:157-167)."""
    column: ColLike
    id: int
    args: Sequence[ColLike] = ()

    def encode(self) -> bytes:
        fn = b""
        if self.id:
            fn += _f_varint(1, self.id)
        for a in self.args:
            fn += _f_bytes(2, col(a).encode())
        return _f_bytes(1, col(self.column).encode()) + _f_bytes(2, fn)


@dataclass
class GroupBy:
    """TProgram.TGroupBy (ssa.proto:181-184)."""
    aggregates: Sequence[Aggregate]
    keys: Sequence[ColLike] = ()

    def encode(self) -> bytes:
        body = b"".join(_f_bytes(1, a.encode()) for a in self.aggregates)
        body += b"".join(_f_bytes(2, col(k).encode()) for k in self.keys)
        return _f_bytes(4, body)  # TCommand.GroupBy = 4


@dataclass
class Program:
    """TProgram (ssa.proto:196-198)."""
    commands: List[object] = field(default_factory=list)
    version: Optional[int] = None
    kernels: Optional[bytes] = None

    # -- builder helpers mirroring how the reference's tests fill the proto ----
    def assign_const(self, column: ColLike, kind: str, value) -> "Program":
        self.commands.append(Assign(column, constant=Constant(kind, value)))
        return self

    def assign_func(self, column: Optional[ColLike], func_id: int, args: Sequence[ColLike]) -> "Program":
        self.commands.append(Assign(column, function=Function(func_id, args)))
        return self

    def assign_op(self, column: Optional[ColLike], operation: int, args: Sequence[ColLike] = ()) -> "Program":
        """TAssign(column, EOperation, arguments) for an operation of class OP."""
        self.commands.append(Assign(column, function=Function(DIRECT_BASE + operation, args)))
        return self

    @property
    def direct(self) -> bool:
        return any(isinstance(c, Assign) and c.function is not None and c.function.id >= DIRECT_BASE for c in self.commands)

    def assign_yql(self, column: Optional[ColLike], yql_op: int, args: Sequence[ColLike],
                   kernel_idx: int = 0, func_id: int = 0) -> "Program":
        self.commands.append(Assign(column, function=Function(
            func_id, args, function_type=YQL_KERNEL, kernel_idx=kernel_idx, yql_op=yql_op)))
        return self

    def assign_param(self, column: ColLike, name: str) -> "Program":
        self.commands.append(Assign(column, parameter=name))
        return self

    def filter(self, predicate: ColLike) -> "Program":
        self.commands.append(Filter(predicate))
        return self

    def project(self, columns: Sequence[ColLike]) -> "Program":
        self.commands.append(Projection(columns))
        return self

    def group_by(self, aggregates: Sequence[Aggregate], keys: Sequence[ColLike] = ()) -> "Program":
        self.commands.append(GroupBy(aggregates, keys))
        return self

    def encode(self) -> bytes:
        out = b"".join(_f_bytes(1, c.encode()) for c in self.commands)
        if self.version is not None:
            out += _f_varint(2, self.version)
        if self.kernels is not None:
            out += _f_bytes(3, self.kernels)
        return out


def olap_program(program: Union[Program, bytes], parameters_schema: Optional[bytes] = None,
                 parameters: Optional[bytes] = None) -> bytes:
    """TOlapProgram (ssa.proto:201-209) wrapping a serialized TProgram."""
    body = program.encode() if isinstance(program, Program) else program
    out = _f_bytes(1, body)
    if parameters_schema is not None:
        out += _f_bytes(2, parameters_schema)
    if parameters is not None:
        out += _f_bytes(3, parameters)
    return out
