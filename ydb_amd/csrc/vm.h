// Shared host/device definitions: column types, the scan-VM instruction set and the
// argument blocks passed to the HIP kernels.
//
// One program step (Assigns -> Filters -> Aggregates|Projection, the order fixed by
// ydb/core/formats/arrow/program.cpp:869-902 of the reference) is lowered by the plan
// compiler into ONE pass over the rows: a small accumulator-machine program evaluates
// every Assign expression that is still needed, ANDs the Filter predicates into a row
// predicate, and leaves the aggregate arguments / group-by keys / projected values in
// "output registers" which the kernel's sink consumes.  Intermediates never touch HBM.
#pragma once
#if defined(__HIPCC_RTC__)
// built at run time: the compiler brings the HIP runtime declarations and the fixed-width integer types with it
#define SSA_HD __host__ __device__ __forceinline__
typedef signed char int8_t;
typedef short int16_t;
typedef int int32_t;
typedef long int64_t;
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long uint64_t;
typedef unsigned long uintptr_t;
typedef unsigned long size_t;
#ifndef INT64_MAX
#define INT64_MAX 9223372036854775807LL
#endif
#ifndef INT64_MIN
#define INT64_MIN (-9223372036854775807LL - 1)
#endif
#ifndef INT32_MAX
#define INT32_MAX 2147483647
#endif
#else
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SSA_HD __host__ __device__ __forceinline__
#else
#define SSA_HD inline
#endif
#endif

namespace ssa {

// Arrow types the column-shard scan path can see (ydb/core/formats/arrow/switch/switch_type.h:125-172).
enum TypeId : int32_t {
    T_NONE = 0,
    T_BOOL = 1,  // bit-packed, LSB first
    T_I8 = 2,
    T_I16 = 3,
    T_I32 = 4,
    T_I64 = 5,
    T_U8 = 6,
    T_U16 = 7,
    T_U32 = 8,
    T_U64 = 9,
    T_F32 = 10,
    T_F64 = 11,
    T_TS = 12,      // timestamp[us], int64
    T_DUR = 13,     // duration[us], int64
    T_STRING = 14,  // utf8, int32 offsets
    T_BINARY = 15,  // binary, int32 offsets
    T_DEC128 = 16,  // decimal128(p, s), 16-byte little-endian two's complement
};

SSA_HD bool IsSignedInt(int t) { return t >= T_I8 && t <= T_I64; }
SSA_HD bool IsUnsignedInt(int t) { return t >= T_U8 && t <= T_U64; }
SSA_HD bool IsInt(int t) { return t >= T_I8 && t <= T_U64; }
SSA_HD bool IsFloat(int t) { return t == T_F32 || t == T_F64; }
SSA_HD bool IsNumeric(int t) { return t >= T_I8 && t <= T_F64; }
SSA_HD bool IsStringLike(int t) { return t == T_STRING || t == T_BINARY; }
SSA_HD bool IsTemporal(int t) { return t == T_TS || t == T_DUR; }
SSA_HD int TypeWidth(int t) {  // bytes of one fixed-width value, 0 for bit-packed / variable
    switch (t) {
        case T_I8: case T_U8: return 1;
        case T_I16: case T_U16: return 2;
        case T_I32: case T_U32: case T_F32: return 4;
        case T_I64: case T_U64: case T_F64: case T_TS: case T_DUR: return 8;
        case T_DEC128: return 16;
        default: return 0;
    }
}

// Bytes of string data a wave can stage per string column and tile: its rows' strings from the first row's start to
// 8 bytes past the last row's start (enough for every string of up to 8 bytes when the average is 8), plus alignment slack.
// Per-wave slot of a staged string column: the end offset of the wave's last row in its first 4 bytes, and from byte 16
// on one 64-bit word per tile row of the lanes (bit t: the string of lane t's row matched) for the wave-wide search.
constexpr int kEndSlotBytes = 16 + 16 * 8;
// Head area of one wave and string column: 8 bytes for each of its 64 x U rows, the copy's alignment skew and slack.
SSA_HD constexpr int CharsStageBytes(int rowsPerThread) { return 64 * rowsPerThread * 8 + 32; }

// Bytes per row of the buffer a scan stages into LDS: the values of a fixed-width column, the int32 offsets of a
// string column (0: nothing to stage).
SSA_HD int StageWidth(int t) {
    const int w = TypeWidth(t);
    if (w >= 1 && w <= 8) return w;
    return (t == T_STRING || t == T_BINARY) ? 4 : 0;
}

// Value classes: how a 64-bit VM payload is interpreted by an instruction.
enum ValClass : int32_t {
    CLS_I = 0,    // signed integer, sign-extended to 64 bits
    CLS_U = 1,    // unsigned integer, zero-extended (also BOOL 0/1)
    CLS_F64 = 2,  // IEEE double
    CLS_F32 = 3,  // float value held exactly in a double payload; arithmetic rounds to float
    CLS_IU = 4,   // compare only: A signed, B unsigned (value-correct mixed compare)
    CLS_UI = 5,   // compare only: A unsigned, B signed
};
SSA_HD int ClassOfType(int t) {
    if (IsSignedInt(t) || IsTemporal(t)) return CLS_I;
    if (IsUnsignedInt(t) || t == T_BOOL) return CLS_U;
    if (t == T_F32) return CLS_F32;
    return CLS_F64;
}

enum CmpOp : int32_t { CMP_EQ = 0, CMP_NE = 1, CMP_LT = 2, CMP_LE = 3, CMP_GT = 4, CMP_GE = 5 };

// Null/zero-division conventions of the two kernel families the reference dispatches to.
enum Dialect : int32_t {
    DIALECT_ARROW = 0,  // Arrow 5.0 compute: validity = AND of inputs; int x/0 -> error Status
    DIALECT_YQL = 1,    // YQL block kernels: x/0 and MIN/-1 -> NULL; And/Or are Kleene
};

enum SrcKind : int32_t { SRC_NONE = 0, SRC_COL = 1, SRC_CONST = 2, SRC_REG = 3 };

enum StrMatchKind : int32_t { STR_CONTAINS = 0, STR_STARTS = 1, STR_ENDS = 2, STR_LIKE = 3 /* the pattern is a whole SQL LIKE pattern */ };

enum OpCode : int32_t {
    OP_END = 0,
    OP_LD,        // A = src
    OP_ST,        // R[si] = A
    OP_ADD,       // A = A + B            a = class
    OP_SUB,       // A = A - B
    OP_RSUB,      // A = B - A
    OP_MUL,
    OP_DIV,       // A = A / B            a = class, b = dialect | (result TypeId << 1)
    OP_RDIV,      // A = B / A
    OP_MOD,       // A = A % B            (YQL Mod)
    OP_RMOD,
    OP_CMP,       // A = A cmp B          a = class, b = CmpOp
    OP_AND,       // b = dialect (ARROW: null if either null; YQL: Kleene)
    OP_OR,
    OP_XOR,
    OP_NOT,
    OP_NEG,       // a = class
    OP_ABS,       // a = class
    OP_ISNULL,
    OP_ISVALID,
    OP_WRAP,      // a = TypeId: wrap integer payload to that type's width
    OP_CAST,      // a = source class, b = target TypeId (C++ static_cast semantics)
    OP_CHECK,     // a = source class, b = target TypeId: flag an error if the value does not fit (safe implicit cast)
    OP_COALESCE,  // A = A.valid ? A : B
    OP_STRLEN,    // A = byte length of string column si (sk = SRC_NONE: the column values are not fetched)
    OP_STRMATCH,  // A = match(string column si, pattern const a); b = StrMatchKind | (ignore_case << 4)
    OP_STRCMP,    // A = string column si  cmp  (const a | column a); b = CmpOp | (rhs_is_column << 4)
    OP_PRED,      // P &= (A != 0)   -- value bit only, validity ignored (reference: program.cpp:517-527)
    OP_ROWIDX,    // A.value = global row index; A's validity is kept.  a = RowIdxFlags (the order the reference's
                  // single-byte-key path visits rows in; a NaN tag in bit 0 of the result)
    OP_PREDCMP,   // P &= (column si  cmp  constant a); b = CmpOp | (class << 4).  Fusion of LD col; CMP const; PRED
    OP_DECPART,   // A = part a of decimal128 column si: 0 = bits 0..31, 1 = bits 32..63, 2 = bits 64..127 (validity of the column)
    OP_FUNC,      // A = f(A, B)          a = TypeId of the operands, b = FuncId; unary functions ignore B
    OP_RFUNC,     // A = f(B, A)
    OP_SELECT,    // A = (A != 0) ? B : R[a]   value and validity of the chosen side (YQL BlockIf)
    OP_LOOKUP,    // A = table a [A]: A holds a dictionary code; null and out-of-range codes read the table's last entry
};

// Functions of OP_FUNC: the custom-registry kernels of ydb/library/arrow_kernels/func_{gcd,lcm,modulo,modulo_or_zero,
// round,math}.h and Arrow's abs / negate / floor / ceil / trunc.
enum FuncId : int32_t {
    FN_GCD = 1, FN_LCM, FN_MOD, FN_MODZ,
    FN_ABS, FN_NEG,
    FN_ROUND, FN_ROUNDBANKERS, FN_ROUNDEXP2, FN_FLOOR, FN_CEIL, FN_TRUNC,
    FN_SQRT, FN_ACOSH, FN_ATANH, FN_CBRT, FN_COSH, FN_ERF, FN_ERFC, FN_EXP, FN_EXP2, FN_LGAMMA, FN_SINH, FN_TGAMMA,  // double result
    FN_HYPOT,
    FN_ADD_CHECKED, FN_SUB_CHECKED, FN_MUL_CHECKED, FN_DIV_CHECKED,  // computed on non-null slots only
};
SSA_HD bool IsBinaryFunc(int fn) { return (fn >= FN_GCD && fn <= FN_MODZ) || fn >= FN_HYPOT; }
SSA_HD bool IsNotNullFunc(int fn) { return fn >= FN_ADD_CHECKED && fn <= FN_DIV_CHECKED; }

SSA_HD uint32_t MakeInsn(int op, int sk = 0, int si = 0, int a = 0, int b = 0) {
    return uint32_t(op & 0xFF) | (uint32_t(sk & 3) << 8) | (uint32_t(si & 0x3F) << 10) | (uint32_t(a & 0xFF) << 16) |
           (uint32_t(b & 0xFF) << 24);
}
SSA_HD int InsnOp(uint32_t i) { return i & 0xFF; }
SSA_HD int InsnSk(uint32_t i) { return (i >> 8) & 3; }
SSA_HD int InsnSi(uint32_t i) { return (i >> 10) & 0x3F; }
SSA_HD int InsnA(uint32_t i) { return (i >> 16) & 0xFF; }
SSA_HD int InsnB(uint32_t i) { return (i >> 24) & 0xFF; }

constexpr int kMaxCols = 16;     // source columns referenced by one scan program
constexpr int kMaxRegs = 16;     // R registers (outputs first, then temporaries); they live in LDS
constexpr int kMaxConsts = 24;
constexpr int kMaxCode = 64;      // instructions per step (one per lane of a wave)
constexpr int kMaxAggs = 16;
constexpr int kMaxKeys = 6;
constexpr int kMaxOuts = 12;
constexpr int kMaxPatternBytes = 256;  // all string constants of one program, concatenated
constexpr int kMaxOffsetChecks = 4;     // string columns scanned as fixed-width values (see OffsetCheck)
constexpr int kMaxLookupTables = 8;     // per-dictionary-entry results a step reads through OP_LOOKUP

// Error bits a scan kernel can raise (OR-ed into ScanArgs::status).
enum ScanStatus : uint32_t {
    SCAN_OK = 0,
    SCAN_DIV_BY_ZERO = 1,      // Arrow integer divide by zero -> Status::Invalid("divide by zero")
    SCAN_CAST_OVERFLOW = 2,    // safe implicit cast failed -> Status::Invalid("Integer value ... not in range")
    SCAN_TABLE_FULL = 4,       // group-by spill table ran out of slots (host retries with a bigger one)
    SCAN_OVERFLOW = 8,         // checked integer arithmetic -> Status::Invalid("overflow")
    SCAN_NAN_ORDER = 16,       // a NaN reached a float MIN / MAX whose result would depend on the reference's visiting order
    SCAN_BAD_BLOB = 64,        // a compressed buffer of an IPC blob does not decode to the length it announces
    SCAN_NOT_UNIFORM = 32,     // a string column scanned as fixed-width values has strings of another length (host re-runs the step)
};

// OP_ROWIDX variants.  The reference's aggregator visits the rows of a batch grouped by ONE single-byte key in four
// interleaved streams (rows 0, 4, 8, ..; 1, 5, ..; ..) and then the last rows % 4 of the batch
// (ydb/library/arrow_clickhouse/AggregateFunctions/IAggregateFunction.h:384-459): "the first row of a group" is the
// smallest row in that order.
enum RowIdxFlags : int32_t {
    ROWIDX_LOOKUP_ORDER = 1,  // result = (stream << kRowStreamShift) | row, stream = row % 4, or 4 for the last nrows % 4 rows
    ROWIDX_NAN_TAG = 2,       // result = (result << 1) | (A is NaN)
    ROWIDX_ZERO_TAG = 4,      // result = A is +-0.0 ? (result << 1) | (sign of A) : ~0 (the identity of an unsigned MIN)
};
constexpr int kRowStreamShift = 58;
constexpr uint64_t kRowIndexMask = (1ull << kRowStreamShift) - 1ull;

struct DevColumn {
    const void* values;       // fixed width: values; BOOL: bitmap; strings: int32 offsets (rows + 1 entries)
    const uint8_t* validity;  // validity bitmap or nullptr
    const uint8_t* chars;     // string bytes
    int32_t type;
    int32_t val_bit_off;      // BOOL columns: bit offset of row 0 inside `values`
    int32_t valid_bit_off;    // bit offset of row 0 inside `validity`
    int32_t chars_bytes;      // strings: readable bytes at `chars` (an 8-byte read at offset o is safe while o + 8 <= chars_bytes)
};

struct DevConst {
    uint64_t bits;     // payload (numeric) or (offset << 32 | length) into ScanArgs::pattern for strings
    int32_t valid;
    int32_t type;
};

// Pre-decoded dispatch: how the operand of an instruction is fetched ...
enum FetchKind : uint32_t {
    FK_NONE = 0,
    FK_LDS_64,    // staged 8-byte column (int64 / uint64 / double / timestamp / duration): raw bits
    FK_LDS_I32, FK_LDS_U32, FK_LDS_I16, FK_LDS_U16, FK_LDS_I8, FK_LDS_U8,
    FK_LDS_F32,
    FK_REG,       // R register (LDS)
    FK_CONST,
    FK_SLOW,      // unstaged, bit-packed or string column: generic path
    FK_HAS_VALIDITY = 0x80,  // flag: the column carries a validity bitmap
};
// ... and which handler runs it.  Everything not listed executes through the generic form.
enum OpKind : uint32_t {
    OK_GENERIC = 0,
    OK_LD, OK_ST, OK_PREDCMP, OK_CMP, OK_PRED,
    OK_ADD_F64, OK_SUB_F64, OK_RSUB_F64, OK_MUL_F64,
    OK_ADD_I, OK_SUB_I, OK_MUL_I,
};
// Comparison as a mask over {lt = 1, eq = 2, gt = 4, unordered = 8}.
SSA_HD uint32_t CmpMaskOf(int cmp) {
    switch (cmp) {
        case CMP_EQ: return 2;
        case CMP_NE: return 1 | 4 | 8;
        case CMP_LT: return 1;
        case CMP_LE: return 1 | 2;
        case CMP_GT: return 4;
        default: return 4 | 2;
    }
}

enum AggKind : int32_t {
    AGG_NONE = 0,
    AGG_COUNT,     // counts rows where the argument is valid (or all rows when reg < 0)
    AGG_SUM_I,     // wrapping 64-bit integer sum (signed and unsigned share the bits)
    AGG_SUM_F,     // double sum
    AGG_MIN_I, AGG_MAX_I,
    AGG_MIN_U, AGG_MAX_U,
    AGG_MIN_F, AGG_MAX_F,
    AGG_SOME,      // first non-null in row order
    AGG_MIN_S, AGG_MAX_S,  // strings, bytewise: the accumulator is the row index of the winning string (argument = row index)
};

struct AggSpec {
    int32_t kind;
    int32_t reg;        // >= 0: R register holding the argument; -1: none (COUNT of rows); <= -2: column slot (-2 - reg)
    int32_t arg_class;  // ValClass of the argument payload
    int32_t check_valid; // skip rows whose argument is null (0: the argument's validity is not consulted)
    uint32_t fetch_kind;    // pre-decoded operand (FetchKind, payload) for `reg`
    uint32_t fetch_payload;
    int32_t flags;      // AggFlags
    int32_t str_col;    // AGG_MIN_S / AGG_MAX_S: column slot of the string argument
};
enum AggFlags : int32_t {
    AGG_NAN_UNSUPPORTED = 1,  // float MIN / MAX: a NaN argument raises SCAN_NAN_ORDER instead of being skipped
};

struct KeySpec {
    int32_t reg;        // fixed width keys: >= 0 R register, <= -2 column slot (-2 - reg); -1 for string keys
    int32_t col;        // string keys: source column index
    int32_t type;       // TypeId
    int32_t nullable;   // key contributes a null flag
    uint32_t fetch_kind;    // pre-decoded operand for `reg`
    uint32_t fetch_payload;
};

enum SinkKind : int32_t {
    SINK_AGG_NOKEY = 0,   // per-block partial aggregates -> finalize
    SINK_FILTER_BITS = 1, // predicate bitmap + per-tile selected counts
    SINK_PROJECT = 2,     // compaction of output registers at precomputed positions
    SINK_AGG_KEYED = 3,   // LDS-staged hash aggregation with a global spill table
};

// A string column whose strings all have `len` bytes is scanned as a column of `len`-byte values (its chars buffer);
// the scan still reads the column's offsets and checks offsets[row] == first + len * row for every row, raising
// SCAN_NOT_UNIFORM otherwise.  (offsets[nrows] is checked by the host.)
struct OffsetCheck {
    const int32_t* offsets;  // 16-byte aligned
    uint32_t first;          // offsets[0]
    uint32_t len;
};

// A function of one dictionary-encoded column (and constants) evaluated once per dictionary entry: entry e < n holds
// the result for code e, entry n the result for a null (or out-of-range) code.  Values are VM payloads.
struct LookupTable {
    const uint64_t* payload;  // n + 1 entries
    const uint8_t* valid;     // n + 1 entries, 1 = non-null
    uint64_t n;
};

// One slice of a scan over several batches of one schema ("slices" of a table: the reference runs the program on
// each, ydb/core/formats/arrow/program.h:423-433).  A launch covers them all: workgroups [first_block, first_block +
// nblocks) scan this slice's rows with this slice's column addresses and aggregate into this slice's part of the
// level-2 table, slots [table_off, table_off + capacity).
struct SliceDesc {
    int64_t nrows;
    uint64_t table_off;
    int32_t first_block, nblocks;
    DevColumn cols[kMaxCols];      // types as in ScanArgs::cols; addresses and bit offsets of this slice
    OffsetCheck checks[kMaxOffsetChecks];
};

struct ScanArgs {
    int64_t nrows;
    int32_t ncols;
    int32_t ncode;
    int32_t nconsts;
    int32_t nullable;        // any nullable input / null-producing op: validity is tracked
    DevColumn cols[kMaxCols];
    DevConst consts[kMaxConsts];
    uint32_t code[kMaxCode * 3];   // pre-decoded: {dispatch word, operand payload, generic form} per instruction
    uint8_t pattern[kMaxPatternBytes];
    uint32_t* status;        // device word, OR-ed with ScanStatus bits
    // LDS plan of one workgroup (byte offsets into dynamic shared memory)
    int32_t lds_col_off[kMaxCols];  // staged tile of column c, or -1: read from global memory on demand
    int32_t lds_chars_off[kMaxCols]; // string column c: staging area of the heads of its strings (CharsStageBytes(U) per wave), or -1
    int32_t lds_reg_offs[kMaxRegs]; // R register r: tile rows x 8 bytes at this offset (may reuse a dead staged column)
    int32_t lds_regok_off;          // R register validity: nregs x lanes bytes
    int32_t nregs;
    int32_t lds_sink_off;           // first byte available to the sink (group-by table)
    int32_t lds_end_off[kMaxCols];  // staged string column c: one kEndSlotBytes slot per wave (end offset of the wave's last row; match words), or -1
    int32_t debug_flags;            // diagnostics only: 1 = skip staging, 2 = skip evaluation (timing experiments)
    int32_t nchecks;
    OffsetCheck checks[kMaxOffsetChecks];
    LookupTable tables[kMaxLookupTables];
    // a scan over several slices (else nullptr): the slices, and the slice of every workgroup of the grid
    const SliceDesc* slices;
    const int32_t* block_slice;
};

}  // namespace ssa
