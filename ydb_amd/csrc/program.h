// Host-side mirror of the reference's SSA program objects:
//   NSsa::TColumnInfo / TAssign / TAggregateAssign / TProgramStep / TProgram
//     (ydb/core/formats/arrow/program.h:53-460)
//   NOlap::TProgramContainer::Init / ParseProgram and TProgramBuilder
//     (ydb/core/tx/program/program.cpp:19-612)
//   NSsa::OptimizeProgram (ydb/core/formats/arrow/ssa_program_optimizer.cpp:7-30)
// Same names, argument meaning and error behaviour; the parsing is done from the raw
// protobuf bytes with the hand-written reader (no generated classes).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "vm.h"

namespace ssa {

// NKikimr::NKernels::EOperation (ydb/library/arrow_kernels/operations.h:5-84), same enumerators in the same order.  The
// protobuf parser produces the ones with a TFunction id (ydb/core/tx/program/program.cpp:126-250); the others reach a
// program only through TAssign objects built in code (ydb/core/formats/arrow/ut/ut_program_step.cpp:30-41), which the
// C-ABI carries as TFunction ids of kDirectOperationBase + the enumerator.
enum class EOperation {
    Unspecified = 0,
    Constant,
    CastBoolean, CastInt8, CastInt16, CastInt32, CastInt64,
    CastUInt8, CastUInt16, CastUInt32, CastUInt64, CastFloat, CastDouble,
    CastBinary, CastFixedSizeBinary, CastString, CastTimestamp,
    IsValid, IsNull,
    Equal, NotEqual, Less, LessEqual, Greater, GreaterEqual,
    Invert, And, Or, Xor,
    Add, Subtract, Multiply, Divide, Abs, Negate, Gcd, Lcm, Modulo, ModuloOrZero,
    AddNotNull, SubtractNotNull, MultiplyNotNull, DivideNotNull,
    BinaryLength, MatchSubstring, MatchLike, StartsWith, EndsWith,
    Acosh, Atanh, Cbrt, Cosh, E, Erf, Erfc, Exp, Exp2, Exp10, Hypot, Lgamma, Pi, Sinh, Sqrt, Tgamma,
    Floor, Ceil, Trunc, Round, RoundBankers, RoundToExp2,
};
constexpr uint32_t kDirectOperationBase = 1000;

// ydb/core/formats/arrow/program.h:16-24
enum class EAggregate { Unspecified = 0, Some = 1, Count = 2, Min = 3, Max = 4, Sum = 5 };

// NYql::TKernelRequestBuilder::EBinaryOp (ydb/library/yql/core/arrow_kernels/request/request.h:19-42)
enum class EYqlOp : uint32_t {
    And = 0, Or, Xor, Add, Sub, Mul, Div, Mod, StartsWith, EndsWith, StringContains,
    Equals, NotEquals, Less, LessOrEqual, Greater, GreaterOrEqual, Coalesce,
    // Not an EBinaryOp: TKernelRequestBuilder::AddIf (request.h:49) is the only kernel of three arguments and is sent
    // with a KernelIdx alone (ydb/core/kqp/query_compiler/kqp_olap_compiler.cpp:740-759); recognised by its arity.
    If = 0x10000,
    // NYql::TKernelRequestBuilder::EUnaryOp (request.h:11-17) + kYqlUnaryBase.  KQP sends these with a KernelIdx alone
    // (kqp_olap_compiler.cpp:655-676, every pushed NOT at :804-812): which operation an index stands for is written in
    // the program's serialized MiniKQL kernel set, which this build does not read -- the caller hands the operations of
    // the kernel indexes over beside the program (TProgramContainer::Init, kernelOps).
    Not = 0x100, Size, Minus, Abs, Just,
};
constexpr uint32_t kYqlUnaryBase = 0x100;
constexpr uint32_t kYqlKernelUnknown = 0xFFFFFFFFu;

const char* GetFunctionName(EOperation op);          // program.cpp:163-301
const char* GetFunctionName(EAggregate op);          // program.cpp:391-409
const char* GetHouseFunctionName(EAggregate op);     // program.cpp:411-431
EOperation ValidateOperation(EOperation op, uint32_t argsSize);  // program.cpp:303-389

struct TColumnInfo {
    std::string ColumnName;
    uint32_t ColumnId = 0;
    bool GeneratedFlag = false;
    static TColumnInfo Generated(uint32_t id, const std::string& name) { return {name, id, true}; }
    static TColumnInfo Original(uint32_t id, const std::string& name) { return {name, id, false}; }
    bool IsGenerated() const { return GeneratedFlag; }
    const std::string& GetColumnName() const { return ColumnName; }
    std::string DebugString() const { return (GeneratedFlag ? "G:" : "") + ColumnName; }
};

// An Arrow scalar as the parser produces it (program.cpp:253-295 of the tx/program file).
struct TScalar {
    int32_t Type = T_NONE;   // TypeId
    bool Valid = true;
    uint64_t Bits = 0;       // VM payload for numeric / bool / timestamp
    std::string Bytes;       // string / binary value
    std::string ToString() const;
};

struct TAssign {
    TColumnInfo Column;
    EOperation Operation = EOperation::Unspecified;
    std::vector<TColumnInfo> Arguments;
    std::optional<TScalar> Constant;
    // function options
    int32_t CastTo = T_NONE;          // arrow::compute::CastOptions(false).to_type
    bool HasMatchOptions = false;     // arrow::compute::MatchSubstringOptions
    std::string Pattern;
    bool IgnoreCase = false;
    // YQL kernel
    bool IsYqlKernel = false;
    uint32_t KernelIdx = 0;
    std::optional<uint32_t> YqlOperationId;

    // Set by the executor, never by the parser: the function has been evaluated once per entry of a dictionary and
    // this assign reads the result through the dictionary codes in Arguments[0] (lookup table `DictTable` of the step).
    int DictTable = -1;
    int32_t DictResultType = T_NONE;
    bool DictResultNullable = false;

    bool IsConstant() const { return Operation == EOperation::Constant; }
    bool IsOk() const { return Operation != EOperation::Unspecified || IsYqlKernel; }
    const std::string& GetName() const { return Column.ColumnName; }
    std::string DebugString() const;
};

struct TAggregateAssign {
    TColumnInfo Column;
    EAggregate Operation = EAggregate::Unspecified;
    std::vector<TColumnInfo> Arguments;
    bool IsOk() const { return Operation != EAggregate::Unspecified; }
    const std::string& GetName() const { return Column.ColumnName; }
};

// Group of commands that finishes with a Projection or GroupBy (program.h:304-410).
struct TProgramStep {
    std::vector<TAssign> Assignes;
    std::vector<TColumnInfo> Filters;
    std::set<uint32_t> FilterOriginalColumnIds;
    std::vector<TAggregateAssign> GroupBy;
    std::vector<TColumnInfo> GroupByKeys;
    std::vector<TColumnInfo> Projection;

    void AddAssigne(const TAssign& a);
    void AddFilter(const TColumnInfo& f);
    bool Empty() const {
        return Assignes.empty() && Filters.empty() && Projection.empty() && GroupBy.empty() && GroupByKeys.empty();
    }
    bool IsFilterOnly() const { return Filters.size() && (!GroupBy.size() && !GroupByKeys.size()); }
    std::set<std::string> GetColumnsInUsage(bool originalOnly = false) const;
    std::string DebugString() const;
};

struct TProgram {
    std::vector<std::shared_ptr<TProgramStep>> Steps;
    // THashMap<ui32, TColumnInfo> in the reference; ordered by first use here so that the
    // COUNT(*) rewrite below is deterministic.
    std::vector<std::pair<uint32_t, TColumnInfo>> SourceColumns;
    std::set<std::string> GetEarlyFilterColumns() const;
    std::set<std::string> GetProcessingColumns() const;
    std::string DebugString() const;
};

// IColumnResolver (ydb/core/tx/program/program.h:17-25)
class IColumnResolver {
public:
    virtual ~IColumnResolver() = default;
    virtual std::string GetColumnName(uint32_t id, bool required = true) const = 0;
    virtual std::optional<uint32_t> GetColumnIdOptional(const std::string& name) const = 0;
    virtual TColumnInfo GetDefaultColumn() const = 0;
};

class TMapColumnResolver : public IColumnResolver {
public:
    void Add(uint32_t id, const std::string& name) {
        ById[id] = name;
        ByName[name] = id;
        if (Order.empty()) First = TColumnInfo::Original(id, name);
        Order.push_back(id);
    }
    std::string GetColumnName(uint32_t id, bool) const override {
        auto it = ById.find(id);
        return it == ById.end() ? std::string() : it->second;
    }
    std::optional<uint32_t> GetColumnIdOptional(const std::string& name) const override {
        auto it = ByName.find(name);
        if (it == ByName.end()) return {};
        return it->second;
    }
    TColumnInfo GetDefaultColumn() const override { return First; }
    bool Empty() const { return Order.empty(); }

private:
    std::unordered_map<uint32_t, std::string> ById;
    std::unordered_map<std::string, uint32_t> ByName;
    std::vector<uint32_t> Order;
    TColumnInfo First;
};

// A one-row batch of named scalars (TOlapProgram.Parameters after deserialization).
using TParameters = std::map<std::string, TScalar>;

// TProgramContainer (ydb/core/tx/program/program.h:55-127): owns the parsed program.
class TProgramContainer {
public:
    // `serialized` is a TOlapProgram when olapWrapper is true (OLAP_PROGRAM_SSA_PROGRAM_WITH_PARAMETERS),
    // else a bare TProgram.  Returns false and fills `error` on a malformed / unsupported program;
    // an empty error from the builder becomes "Wrong olap program" like the reference does.
    // `directOperations`: accept TFunction ids of kDirectOperationBase + EOperation as well (see EOperation).
    // `kernelOps`: per KernelIdx the EYqlOp the kernel stands for, kYqlKernelUnknown where the caller does not say.
    bool Init(const IColumnResolver& columnResolver, const uint8_t* serialized, size_t size, bool olapWrapper,
              const TParameters& parameters, std::string& error, bool directOperations = false,
              const std::vector<uint32_t>* kernelOps = nullptr);
    // A program built from objects (the operator-level seams), not parsed.
    void Adopt(std::shared_ptr<TProgram> program) { Program = std::move(program); }
    bool HasProgram() const { return !!Program; }
    const std::shared_ptr<TProgram>& GetProgram() const { return Program; }

private:
    bool ParseProgram(const IColumnResolver& columnResolver, std::string_view program, const TParameters& parameters,
                      std::string& error, bool directOperations, const std::vector<uint32_t>* kernelOps);
    std::shared_ptr<TProgram> Program;
};

void OptimizeProgram(TProgram& program);

}  // namespace ssa
