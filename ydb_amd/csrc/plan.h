// Plan compiler: one TProgramStep + the schema of its input batch -> one fused scan.
//
// The reference executes a step as "one Arrow call per Assign, each materialising a column"
// (ydb/core/formats/arrow/program.cpp:677-698) followed by a row loop that ANDs the filter columns
// (:773-796), a gather per needed column (:798-832) and one pass per aggregate (:700-771).  Here the
// whole step becomes a typed expression DAG which is flattened into scan-VM code (vm.h); type
// inference reproduces Arrow's DispatchBest promotion rules for SIMPLE_ARROW functions and the YQL
// kernel typing table for YQL_KERNEL functions, so every output column has the reference's type.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "program.h"
#include "vm.h"

namespace ssa {

enum StatusCode : int {
    ST_OK = 0,
    ST_INVALID = 1,        // arrow::Status::Invalid / ExecutionError from the reference path
    ST_PARSE_ERROR = 2,    // TProgramContainer::Init returned false
    ST_UNSUPPORTED = 3,    // valid program, but not executable on the GPU path: caller falls back to CPU ApplyTo
    ST_HIP_ERROR = 4,
};

struct Status {
    int Code = ST_OK;
    std::string Msg;
    bool ok() const { return Code == ST_OK; }
    static Status OK() { return {}; }
    static Status Invalid(std::string m) { return {ST_INVALID, std::move(m)}; }
    static Status Unsupported(std::string m) { return {ST_UNSUPPORTED, std::move(m)}; }
    static Status Hip(std::string m) { return {ST_HIP_ERROR, std::move(m)}; }
};

// Schema entry of a step's input batch (a TDatumBatch column: array or scalar).
struct FieldDesc {
    std::string Name;
    int32_t Type = T_NONE;
    bool Nullable = false;   // a validity buffer is present
    bool IsScalar = false;
    TScalar Scalar;
    int32_t Precision = 0, Scale = 0;  // decimal128
};

struct OutColumn {
    enum ESource { FROM_INPUT, FROM_REG, FROM_SCALAR, FROM_AGG, FROM_KEY } Source = FROM_INPUT;
    std::string Name;
    int32_t Type = T_NONE;
    int Index = -1;        // input column index / R register / aggregate index / key index
    bool Nullable = false;
    TScalar Scalar;
    int32_t Precision = 0, Scale = 0;
};

struct AggPlan {
    AggSpec Spec;
    int32_t OutType;
    int32_t ArgType;
    bool ArgNullable;
    int ArgInputColumn = -1;  // argument is a plain input column (index into the step's input batch), else -1
    int SomeInputColumn = -1; // SOME: the kernel aggregates MIN(row index); the value is gathered from this input column
    // decimal128 SUM: this entry sums bits 0..31 of the values; two hidden entries appended after the step's own
    // aggregates sum bits 32..63 and the high words.  The executor recombines the three 64-bit sums modulo 2^128.
    int DecMid = -1, DecHi = -1;
    int32_t Precision = 0, Scale = 0;
    bool Hidden = false;
    // Keyed float MIN / MAX: index of the hidden aggregate MIN((row << 1) | is-NaN) over the same argument.  The
    // reference's state takes the group's first non-null value whatever it is and then compares with `<` / `>`, which a
    // NaN never passes (AggregateFunctionMinMaxAny.h:96-138): the result is NaN exactly when that first value is one.
    int NanTag = -1;
    // The same compares never replace a zero by the zero of the other sign: a zero result has the sign of the group's
    // first zero argument.  Index of the hidden aggregate MIN((row << 1) | sign) over the zero arguments.
    int ZeroTag = -1;
};

struct KeyPlan {
    KeySpec Spec;
    int InputColumn = -1;     // batch column index when the key is a plain input column
};

struct StepPlan {
    enum EKind { PASS_THROUGH, PROJECT, AGG_NOKEY, AGG_KEYED, FILTER_ONLY } Kind = PASS_THROUGH;
    // scan program
    std::vector<uint32_t> Code;
    std::vector<DevConst> Consts;
    std::string PatternBytes;
    std::vector<int> SourceColumns;   // VM column slot -> index into the step's input batch
    bool Nullable = false;
    bool HasFilter = false;
    bool UnreadFallible = false;  // assigns that nothing reads but that can fail: evaluated for their errors
    // The step groups by one single-byte key and an aggregate's result depends on the positions of the rows in the
    // FILTERED batch (RowIdxFlags): a step with filters is run on the compacted batch, as the reference does.
    bool NeedsFilteredPositions = false;
    bool HasStringMinMax = false;  // MIN / MAX / SOME over a string column: only the keyed sink has them (a key-less step runs as one group)
    int NumRegsUsed = 0;
    // sink description
    std::vector<OutColumn> Outs;      // output batch columns in order
    std::vector<AggPlan> Aggs;
    std::vector<KeyPlan> Keys;
    std::string DebugString() const;
};

enum ECompileMode {
    CM_FULL = 0,          // filters and outputs in one scan (aggregating steps)
    CM_FILTER_ONLY = 1,   // Assigns + Filters only (TProgramStep::BuildFilter, program.cpp:921-946; pass 1 of a compaction)
    CM_OUTPUTS_ONLY = 2,  // outputs without the filter predicates (pass 2 of a compaction)
};

// Compile `step` against the input schema.
// `serializedKeys`: the GROUP BY takes the reference's `serialized` method whatever the key types are (the caller has
// replaced string keys by fixed-width stand-ins): no null-aware keys, no single-byte-key lookup table.
Status CompileStep(const TProgramStep& step, const std::vector<FieldDesc>& input, ECompileMode mode, StepPlan& out,
                   bool serializedKeys = false);

}  // namespace ssa
