#include "program.h"

#include <cstring>

#include "pb_reader.h"
#include "vm_ops.h"

namespace ssa {

// ---------------------------------------------------------------------------------------------
// names / arity (ydb/core/formats/arrow/program.cpp:163-431 of the reference)
// ---------------------------------------------------------------------------------------------
const char* GetFunctionName(EOperation op) {
    switch (op) {
        case EOperation::CastBoolean: case EOperation::CastInt8: case EOperation::CastInt16:
        case EOperation::CastInt32: case EOperation::CastInt64: case EOperation::CastUInt8:
        case EOperation::CastUInt16: case EOperation::CastUInt32: case EOperation::CastUInt64:
        case EOperation::CastFloat: case EOperation::CastDouble: case EOperation::CastTimestamp:
            return "ydb.cast";
        case EOperation::IsValid: return "is_valid";
        case EOperation::IsNull: return "is_null";
        case EOperation::Equal: return "equal";
        case EOperation::NotEqual: return "not_equal";
        case EOperation::Less: return "less";
        case EOperation::LessEqual: return "less_equal";
        case EOperation::Greater: return "greater";
        case EOperation::GreaterEqual: return "greater_equal";
        case EOperation::Invert: return "invert";
        case EOperation::And: return "and";
        case EOperation::Or: return "or";
        case EOperation::Xor: return "xor";
        case EOperation::Add: return "add";
        case EOperation::Subtract: return "subtract";
        case EOperation::Multiply: return "multiply";
        case EOperation::Divide: return "divide";
        case EOperation::Abs: return "abs";
        case EOperation::Negate: return "negate";
        case EOperation::Gcd: return "gcd";
        case EOperation::Lcm: return "lcm";
        case EOperation::Modulo: return "mod";
        case EOperation::ModuloOrZero: return "modOrZero";
        case EOperation::AddNotNull: return "add_checked";
        case EOperation::SubtractNotNull: return "subtract_checked";
        case EOperation::MultiplyNotNull: return "multiply_checked";
        case EOperation::DivideNotNull: return "divide_checked";
        case EOperation::Acosh: return "acosh";
        case EOperation::Atanh: return "atanh";
        case EOperation::Cbrt: return "cbrt";
        case EOperation::Cosh: return "cosh";
        case EOperation::E: return "e";
        case EOperation::Erf: return "erf";
        case EOperation::Erfc: return "erfc";
        case EOperation::Exp: return "exp";
        case EOperation::Exp2: return "exp2";
        case EOperation::Exp10: return "exp10";
        case EOperation::Hypot: return "hypot";
        case EOperation::Lgamma: return "lgamma";
        case EOperation::Pi: return "pi";
        case EOperation::Sinh: return "sinh";
        case EOperation::Sqrt: return "sqrt";
        case EOperation::Tgamma: return "tgamma";
        case EOperation::Floor: return "floor";
        case EOperation::Ceil: return "ceil";
        case EOperation::Trunc: return "trunc";
        case EOperation::Round: return "round";
        case EOperation::RoundBankers: return "roundBankers";
        case EOperation::RoundToExp2: return "roundToExp2";
        case EOperation::BinaryLength: return "binary_length";
        case EOperation::MatchSubstring: return "match_substring";
        case EOperation::MatchLike: return "match_like";
        case EOperation::StartsWith: return "starts_with";
        case EOperation::EndsWith: return "ends_with";
        default: return "";
    }
}

const char* GetFunctionName(EAggregate op) {
    switch (op) {
        case EAggregate::Count: return "count";
        case EAggregate::Min: case EAggregate::Max: return "min_max";
        case EAggregate::Sum: return "sum";
        default: return "";
    }
}

const char* GetHouseFunctionName(EAggregate op) {
    switch (op) {
        case EAggregate::Some: return "ch.any";
        case EAggregate::Count: return "ch.count";
        case EAggregate::Min: return "ch.min";
        case EAggregate::Max: return "ch.max";
        case EAggregate::Sum: return "ch.sum";
        default: return "";
    }
}

EOperation ValidateOperation(EOperation op, uint32_t argsSize) {
    uint32_t need;
    switch (op) {
        case EOperation::Equal: case EOperation::NotEqual: case EOperation::Less: case EOperation::LessEqual:
        case EOperation::Greater: case EOperation::GreaterEqual: case EOperation::And: case EOperation::Or:
        case EOperation::Xor: case EOperation::Add: case EOperation::Subtract: case EOperation::Multiply:
        case EOperation::Divide: case EOperation::Modulo: case EOperation::AddNotNull: case EOperation::SubtractNotNull:
        case EOperation::MultiplyNotNull: case EOperation::DivideNotNull: case EOperation::ModuloOrZero:
        case EOperation::Gcd: case EOperation::Lcm:
            need = 2;
            break;
        case EOperation::CastBoolean: case EOperation::CastInt8: case EOperation::CastInt16:
        case EOperation::CastInt32: case EOperation::CastInt64: case EOperation::CastUInt8:
        case EOperation::CastUInt16: case EOperation::CastUInt32: case EOperation::CastUInt64:
        case EOperation::CastFloat: case EOperation::CastDouble: case EOperation::CastTimestamp:
        case EOperation::IsValid: case EOperation::IsNull: case EOperation::BinaryLength: case EOperation::Invert:
        case EOperation::Abs: case EOperation::Negate:
        case EOperation::StartsWith: case EOperation::EndsWith: case EOperation::MatchSubstring:
        case EOperation::MatchLike:
            need = 1;
            break;
        case EOperation::Acosh: case EOperation::Atanh: case EOperation::Cbrt: case EOperation::Cosh: case EOperation::E:
        case EOperation::Erf: case EOperation::Erfc: case EOperation::Exp: case EOperation::Exp2: case EOperation::Exp10:
        case EOperation::Hypot: case EOperation::Lgamma: case EOperation::Pi: case EOperation::Sinh: case EOperation::Sqrt:
        case EOperation::Tgamma: case EOperation::Floor: case EOperation::Ceil: case EOperation::Trunc:
        case EOperation::Round: case EOperation::RoundBankers: case EOperation::RoundToExp2:
            return op;  // the reference does not check these ("TODO: check", program.cpp:361-383)
        default:
            return EOperation::Unspecified;
    }
    return argsSize == need ? op : EOperation::Unspecified;
}

// ---------------------------------------------------------------------------------------------
// debug strings
// ---------------------------------------------------------------------------------------------
std::string TScalar::ToString() const {
    if (!Valid) return "null";
    if (IsStringLike(Type)) return Bytes;
    if (IsFloat(Type)) return std::to_string(BitsToF64(Bits));
    if (IsUnsignedInt(Type) || Type == T_BOOL) return std::to_string(Bits);
    return std::to_string(int64_t(Bits));
}

std::string TAssign::DebugString() const {
    std::string sb = "{";
    if (Operation != EOperation::Unspecified) sb += "op=" + std::to_string(int(Operation)) + ";";
    if (YqlOperationId) sb += "yql_op=" + std::to_string(*YqlOperationId) + ";";
    if (!Arguments.empty()) {
        sb += "arguments=[";
        for (auto& a : Arguments) sb += a.DebugString() + ";";
        sb += "];";
    }
    if (Constant) sb += "const=" + Constant->ToString() + ";";
    if (IsYqlKernel) sb += "kernel=" + std::to_string(KernelIdx) + ";";
    sb += "column=" + Column.DebugString() + ";}";
    return sb;
}

std::string TProgramStep::DebugString() const {
    std::string sb = "{";
    if (!Assignes.empty()) {
        sb += "assignes=[";
        for (auto& a : Assignes) sb += a.DebugString() + ";";
        sb += "];";
    }
    if (!Filters.empty()) {
        sb += "filters=[";
        for (auto& f : Filters) sb += f.DebugString() + ";";
        sb += "];";
    }
    if (!GroupBy.empty()) sb += "group_by_count=" + std::to_string(GroupBy.size()) + "; ";
    if (!GroupByKeys.empty()) sb += "group_by_keys_count=" + std::to_string(GroupByKeys.size()) + ";";
    sb += "projections=[";
    for (auto& p : Projection) sb += p.DebugString() + ";";
    sb += "];}";
    return sb;
}

std::string TProgram::DebugString() const {
    std::string sb = "[";
    for (auto& s : Steps) sb += s->DebugString() + ";";
    return sb + "]";
}

// ---------------------------------------------------------------------------------------------
// steps
// ---------------------------------------------------------------------------------------------
void TProgramStep::AddAssigne(const TAssign& a) {
    if (!a.Column.IsGenerated()) FilterOriginalColumnIds.insert(a.Column.ColumnId);
    for (auto& arg : a.Arguments) {
        if (!arg.IsGenerated()) FilterOriginalColumnIds.insert(arg.ColumnId);
    }
    Assignes.push_back(a);
}

void TProgramStep::AddFilter(const TColumnInfo& f) {
    if (!f.IsGenerated()) FilterOriginalColumnIds.insert(f.ColumnId);
    Filters.push_back(f);
}

std::set<std::string> TProgramStep::GetColumnsInUsage(bool originalOnly) const {
    std::set<std::string> result;
    for (auto& f : Filters) {
        if (!originalOnly || !f.IsGenerated()) result.insert(f.ColumnName);
    }
    for (auto& a : Assignes) {
        for (auto& arg : a.Arguments) {
            if (!originalOnly || !arg.IsGenerated()) result.insert(arg.ColumnName);
        }
    }
    return result;
}

std::set<std::string> TProgram::GetEarlyFilterColumns() const {
    std::set<std::string> result;
    for (auto& step : Steps) {
        if (!step->IsFilterOnly()) break;
        auto fields = step->GetColumnsInUsage();
        result.insert(fields.begin(), fields.end());
    }
    return result;
}

std::set<std::string> TProgram::GetProcessingColumns() const {
    std::set<std::string> result;
    for (auto& kv : SourceColumns) result.insert(kv.second.ColumnName);
    return result;
}

// COUNT(*) -> COUNT(first GROUP BY key | any source column): ssa_program_optimizer.cpp:7-24.
void OptimizeProgram(TProgram& program) {
    for (auto& step : program.Steps) {
        for (auto& groupBy : step->GroupBy) {
            if (groupBy.Operation == EAggregate::Count && groupBy.Arguments.empty()) {
                if (!step->GroupByKeys.empty()) {
                    groupBy.Arguments.push_back(step->GroupByKeys[0]);
                } else if (!program.SourceColumns.empty()) {
                    groupBy.Arguments.push_back(program.SourceColumns.front().second);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// protobuf -> steps   (TProgramBuilder, tx/program/program.cpp:19-442)
// ---------------------------------------------------------------------------------------------
namespace {

struct PbColumn {
    bool HasId = false, HasName = false;
    uint64_t Id = 0;
    std::string Name;
};

PbColumn ReadColumn(std::string_view bytes) {
    PbColumn c;
    PbReader r(bytes);
    while (r.Next()) {
        if (r.FieldNo == 1 && r.Wire == 0) {
            c.HasId = true;
            c.Id = r.Value;
        } else if (r.FieldNo == 2 && r.Wire == 2) {
            c.HasName = true;
            c.Name = std::string(r.Bytes);
        }
    }
    return c;
}

class TProgramBuilder {
public:
    TProgramBuilder(const IColumnResolver& resolver, const TParameters& params, bool directOperations, const std::vector<uint32_t>* kernelOps)
        : Resolver(resolver), Params(params), DirectOperations(directOperations), KernelOps(kernelOps) {}

    std::string Error;
    std::vector<std::pair<uint32_t, TColumnInfo>> Sources;

    TColumnInfo GetColumnInfo(const PbColumn& column) {
        if (column.HasId && column.Id) {
            const uint32_t columnId = uint32_t(column.Id);
            const std::string name = Resolver.GetColumnName(columnId, false);
            if (name.empty()) return TColumnInfo::Generated(columnId, GenerateName(column));
            AddSource(columnId, name);
            return TColumnInfo::Original(columnId, name);
        }
        if (column.HasName && !column.Name.empty()) {
            auto columnId = Resolver.GetColumnIdOptional(column.Name);
            if (columnId) {
                AddSource(*columnId, column.Name);
                return TColumnInfo::Original(*columnId, column.Name);
            }
            return TColumnInfo::Generated(0, GenerateName(column));
        }
        return TColumnInfo::Generated(0, GenerateName(column));
    }

    bool ExtractAssign(TProgramStep& step, std::string_view bytes) {
        PbColumn column;
        enum { NONE, FUNC, EXT, CONST, NUL, PARAM } kind = NONE;
        std::string_view body;
        PbReader r(bytes);
        while (r.Next()) {
            switch (r.FieldNo) {
                case 1: column = ReadColumn(r.Bytes); break;
                case 2: kind = FUNC; body = r.Bytes; break;
                case 3: kind = EXT; break;
                case 4: kind = CONST; body = r.Bytes; break;
                case 5: kind = NUL; break;
                case 6: kind = PARAM; body = r.Bytes; break;
                default: break;
            }
        }
        if (!r.Ok()) return false;
        const TColumnInfo columnName = GetColumnInfo(column);
        switch (kind) {
            case FUNC: {
                TAssign func = MakeFunction(columnName, body);
                if (!func.IsOk()) return false;
                step.AddAssigne(func);
                return true;
            }
            case CONST: {
                TAssign cnst = MakeConstant(columnName, body);
                if (!cnst.IsConstant()) return false;
                Constants[columnName.ColumnName] = *cnst.Constant;
                step.AddAssigne(cnst);
                return true;
            }
            case PARAM: {
                std::string paramName;
                PbReader p(body);
                while (p.Next()) {
                    if (p.FieldNo == 1 && p.Wire == 2) paramName = std::string(p.Bytes);
                }
                auto it = Params.find(paramName);
                if (it == Params.end()) return false;  // no such parameter / not exactly one value
                TAssign a;
                a.Column = columnName;
                a.Operation = EOperation::Constant;
                a.Constant = it->second;
                step.AddAssigne(a);
                return true;
            }
            default:
                return false;  // ExternalFunction, Null, not set
        }
    }

    bool ExtractFilter(TProgramStep& step, std::string_view bytes) {
        PbColumn column;
        bool has = false;
        PbReader r(bytes);
        while (r.Next()) {
            if (r.FieldNo == 1 && r.Wire == 2) {
                column = ReadColumn(r.Bytes);
                has = true;
            }
        }
        if (!has || (!column.HasId && !column.HasName)) return false;
        step.AddFilter(GetColumnInfo(column));
        return true;
    }

    bool ExtractProjection(TProgramStep& step, std::string_view bytes) {
        PbReader r(bytes);
        while (r.Next()) {
            if (r.FieldNo == 1 && r.Wire == 2) step.Projection.push_back(GetColumnInfo(ReadColumn(r.Bytes)));
        }
        return r.Ok();
    }

    bool ExtractGroupBy(TProgramStep& step, std::string_view bytes) {
        std::vector<std::string_view> aggs, keys;
        PbReader r(bytes);
        while (r.Next()) {
            if (r.FieldNo == 1 && r.Wire == 2) aggs.push_back(r.Bytes);
            if (r.FieldNo == 2 && r.Wire == 2) keys.push_back(r.Bytes);
        }
        if (!r.Ok() || aggs.empty()) return false;
        for (auto agg : aggs) {
            PbColumn column;
            std::string_view fn;
            PbReader a(agg);
            while (a.Next()) {
                if (a.FieldNo == 1 && a.Wire == 2) column = ReadColumn(a.Bytes);
                if (a.FieldNo == 2 && a.Wire == 2) fn = a.Bytes;
            }
            const TColumnInfo columnName = GetColumnInfo(column);
            TAggregateAssign func = MakeAggregate(columnName, fn);
            if (!func.IsOk()) return false;
            step.GroupBy.push_back(func);
        }
        for (auto key : keys) step.GroupByKeys.push_back(GetColumnInfo(ReadColumn(key)));
        return true;
    }

private:
    void AddSource(uint32_t id, const std::string& name) {
        for (auto& kv : Sources) {
            if (kv.first == id) return;
        }
        Sources.emplace_back(id, TColumnInfo::Original(id, name));
    }

    static std::string GenerateName(const PbColumn& column) {
        return column.HasName ? column.Name : std::to_string(column.Id);
    }

    TAssign MakeFunction(const TColumnInfo& name, std::string_view bytes) {
        uint32_t id = 0, functionType = 1, kernelIdx = 0;
        std::optional<uint32_t> yqlOp;
        TAssign res;
        res.Column = name;
        PbReader r(bytes);
        while (r.Next()) {
            switch (r.FieldNo) {
                case 1: id = uint32_t(r.Value); break;
                case 2: res.Arguments.push_back(GetColumnInfo(ReadColumn(r.Bytes))); break;
                case 3: functionType = uint32_t(r.Value); break;
                case 4: kernelIdx = uint32_t(r.Value); break;
                case 5: yqlOp = uint32_t(r.Value); break;
                default: break;
            }
        }
        if (!r.Ok()) return res;

        // mkLikeOptions: the pattern is a previously assigned binary-like constant; it becomes
        // MatchSubstringOptions and the argument is dropped.
        auto likeOptions = [&](bool ignoreCase) {
            if (res.Arguments.size() != 2) return false;
            auto it = Constants.find(res.Arguments[1].ColumnName);
            if (it == Constants.end() || !IsStringLike(it->second.Type)) return false;
            res.Arguments.pop_back();
            res.HasMatchOptions = true;
            res.Pattern = it->second.Bytes;
            res.IgnoreCase = ignoreCase;
            return true;
        };
        auto finish = [&](EOperation op) {
            res.Operation = ValidateOperation(op, uint32_t(res.Arguments.size()));
            return res;
        };
        auto cast = [&](EOperation op, int32_t to) {
            res.CastTo = to;
            return finish(op);
        };

        if (functionType == 2) {  // YQL_KERNEL
            // The reference resolves KernelIdx inside the serialized MiniKQL kernel set; this build
            // identifies the kernel by YqlOperationId (kqp_olap_compiler.cpp:680-738 always sets it for
            // runtime version >= 4) or, failing that, by a SIMPLE function id naming the same operation.
            res.IsYqlKernel = true;
            res.KernelIdx = kernelIdx;
            // (what the caller knows about the kernel set comes first: it is the only way a unary kernel is told)
            if (!yqlOp && KernelOps && kernelIdx < KernelOps->size() && (*KernelOps)[kernelIdx] != kYqlKernelUnknown) yqlOp = (*KernelOps)[kernelIdx];
            if (yqlOp) {
                res.YqlOperationId = yqlOp;
            } else if (res.Arguments.size() == 3) {
                res.YqlOperationId = uint32_t(EYqlOp::If);
            } else {
                switch (id) {
                    case 33: res.YqlOperationId = uint32_t(EYqlOp::StartsWith); break;
                    case 34: res.YqlOperationId = uint32_t(EYqlOp::EndsWith); break;
                    case 9: res.YqlOperationId = uint32_t(EYqlOp::StringContains); break;
                    case 1: res.YqlOperationId = uint32_t(EYqlOp::Equals); break;
                    case 2: res.YqlOperationId = uint32_t(EYqlOp::NotEquals); break;
                    case 3: res.YqlOperationId = uint32_t(EYqlOp::Less); break;
                    case 4: res.YqlOperationId = uint32_t(EYqlOp::LessOrEqual); break;
                    case 5: res.YqlOperationId = uint32_t(EYqlOp::Greater); break;
                    case 6: res.YqlOperationId = uint32_t(EYqlOp::GreaterOrEqual); break;
                    case 11: res.YqlOperationId = uint32_t(EYqlOp::And); break;
                    case 12: res.YqlOperationId = uint32_t(EYqlOp::Or); break;
                    case 13: res.YqlOperationId = uint32_t(EYqlOp::Xor); break;
                    case 14: res.YqlOperationId = uint32_t(EYqlOp::Add); break;
                    case 15: res.YqlOperationId = uint32_t(EYqlOp::Sub); break;
                    case 16: res.YqlOperationId = uint32_t(EYqlOp::Mul); break;
                    case 17: res.YqlOperationId = uint32_t(EYqlOp::Div); break;
                    default:
                        Error = "Unknown kernel for " + name.ColumnName + ";kernel_idx=" + std::to_string(kernelIdx);
                        res.IsYqlKernel = false;
                        break;
                }
            }
            return res;
        }

        if (DirectOperations && id >= kDirectOperationBase) {
            // a TAssign built in code: TAssign(column, EOperation, arguments) (ydb/core/formats/arrow/program.h:139-147)
            const uint32_t op = id - kDirectOperationBase;
            if (op > uint32_t(EOperation::RoundToExp2)) return res;
            return finish(EOperation(op));
        }
        switch (id) {
            case 1: return finish(EOperation::Equal);
            case 2: return finish(EOperation::NotEqual);
            case 3: return finish(EOperation::Less);
            case 4: return finish(EOperation::LessEqual);
            case 5: return finish(EOperation::Greater);
            case 6: return finish(EOperation::GreaterEqual);
            case 7: return finish(EOperation::IsNull);
            case 8: return finish(EOperation::BinaryLength);
            case 9: if (likeOptions(false)) return finish(EOperation::MatchSubstring); break;
            case 32: if (likeOptions(false)) return finish(EOperation::MatchLike); break;
            case 33: if (likeOptions(false)) return finish(EOperation::StartsWith); break;
            case 34: if (likeOptions(false)) return finish(EOperation::EndsWith); break;
            case 35: if (likeOptions(true)) return finish(EOperation::MatchSubstring); break;
            case 36: if (likeOptions(true)) return finish(EOperation::StartsWith); break;
            case 37: if (likeOptions(true)) return finish(EOperation::EndsWith); break;
            case 10: return finish(EOperation::Invert);
            case 11: return finish(EOperation::And);
            case 12: return finish(EOperation::Or);
            case 13: return finish(EOperation::Xor);
            case 14: return finish(EOperation::Add);
            case 15: return finish(EOperation::Subtract);
            case 16: return finish(EOperation::Multiply);
            case 17: return finish(EOperation::Divide);
            case 18: return cast(EOperation::CastBoolean, T_BOOL);
            case 19: return cast(EOperation::CastInt8, T_I8);
            case 20: return cast(EOperation::CastInt16, T_I16);
            case 21: return cast(EOperation::CastInt32, T_I32);
            case 22: return cast(EOperation::CastInt64, T_I64);
            case 23: return cast(EOperation::CastUInt8, T_U8);
            case 24: return cast(EOperation::CastUInt16, T_U16);
            case 25: return cast(EOperation::CastUInt32, T_U32);
            case 26: return cast(EOperation::CastUInt64, T_U64);
            case 27: return cast(EOperation::CastFloat, T_F32);
            case 28: return cast(EOperation::CastDouble, T_F64);
            case 31: return cast(EOperation::CastTimestamp, T_TS);
            default: break;  // CAST_TO_BINARY, CAST_TO_FIXED_SIZE_BINARY, UNSPECIFIED
        }
        res.Operation = EOperation::Unspecified;
        return res;
    }

    TAssign MakeConstant(const TColumnInfo& name, std::string_view bytes) {
        TAssign res;
        res.Column = name;
        TScalar s;
        bool set = false;
        PbReader r(bytes);
        while (r.Next()) {  // a oneof: the last field on the wire wins
            set = true;
            switch (r.FieldNo) {
                case 1: s = {T_BOOL, true, uint64_t(r.Value != 0), {}}; break;
                case 2: s = {T_I32, true, uint64_t(int64_t(r.AsInt32())), {}}; break;
                case 3: s = {T_U32, true, uint64_t(uint32_t(r.Value)), {}}; break;
                case 4: s = {T_I64, true, r.Value, {}}; break;
                case 5: s = {T_U64, true, r.Value, {}}; break;
                case 6: s = {T_F32, true, F64ToBits(double(r.AsFloat())), {}}; break;
                case 7: s = {T_F64, true, F64ToBits(r.AsDouble()), {}}; break;
                case 8: s = {T_BINARY, true, 0, std::string(r.Bytes)}; break;
                case 9: s = {T_STRING, true, 0, std::string(r.Bytes)}; break;
                case 10: s = {T_I8, true, uint64_t(int64_t(int8_t(r.AsInt32()))), {}}; break;
                case 11: s = {T_U8, true, uint64_t(uint8_t(r.Value)), {}}; break;
                case 12: s = {T_I16, true, uint64_t(int64_t(int16_t(r.AsInt32()))), {}}; break;
                case 13: s = {T_U16, true, uint64_t(uint16_t(r.Value)), {}}; break;
                case 14: s = {T_TS, true, r.Value, {}}; break;
                default: set = false; break;
            }
        }
        if (!r.Ok() || !set) return res;
        res.Operation = EOperation::Constant;
        res.Constant = s;
        return res;
    }

    TAggregateAssign MakeAggregate(const TColumnInfo& name, std::string_view bytes) {
        uint32_t id = 0, functionType = 1;
        std::vector<TColumnInfo> args;
        PbReader r(bytes);
        while (r.Next()) {
            switch (r.FieldNo) {
                case 1: id = uint32_t(r.Value); break;
                case 2: args.push_back(GetColumnInfo(ReadColumn(r.Bytes))); break;
                case 4: functionType = uint32_t(r.Value); break;
                default: break;
            }
        }
        TAggregateAssign res;
        res.Column = name;
        if (functionType == 2) {
            Error = "Unknown kernel for " + std::to_string(id) + ";aggregate YQL kernels are not supported";
            return res;
        }
        if (args.size() == 1) {
            if (id >= 1 && id <= 5) {
                res.Operation = EAggregate(id);
                res.Arguments = args;
            }
        } else if (args.empty() && id == uint32_t(EAggregate::Count)) {
            res.Operation = EAggregate::Count;  // COUNT(*)
        }
        return res;
    }

    const IColumnResolver& Resolver;
    const TParameters& Params;
    bool DirectOperations;
    const std::vector<uint32_t>* KernelOps;  // may be null
    std::unordered_map<std::string, TScalar> Constants;
};

}  // namespace

bool TProgramContainer::Init(const IColumnResolver& columnResolver, const uint8_t* serialized, size_t size,
                             bool olapWrapper, const TParameters& parameters, std::string& error, bool directOperations,
                             const std::vector<uint32_t>* kernelOps) {
    std::string_view program(reinterpret_cast<const char*>(serialized), size);
    if (olapWrapper) {
        bool found = false;
        PbReader r(serialized, size);
        while (r.Next()) {
            if (r.FieldNo == 1 && r.Wire == 2) {
                program = r.Bytes;
                found = true;
            }
        }
        if (!r.Ok()) {
            error = "Can't parse TOlapProgram";
            return false;
        }
        if (!found) program = std::string_view();
    }
    if (!ParseProgram(columnResolver, program, parameters, error, directOperations, kernelOps)) {
        if (error.empty()) error = "Wrong olap program";
        return false;
    }
    return true;
}

bool TProgramContainer::ParseProgram(const IColumnResolver& columnResolver, std::string_view program,
                                     const TParameters& parameters, std::string& error, bool directOperations,
                                     const std::vector<uint32_t>* kernelOps) {
    auto ssaProgram = std::make_shared<TProgram>();
    TProgramBuilder builder(columnResolver, parameters, directOperations, kernelOps);
    auto step = std::make_shared<TProgramStep>();

    PbReader r(program);
    while (r.Next()) {
        if (r.FieldNo != 1 || r.Wire != 2) continue;  // Version, Kernels
        // TCommand: oneof line
        PbReader c(r.Bytes);
        int line = 0;
        std::string_view body;
        while (c.Next()) {
            if (c.Wire == 2 && c.FieldNo >= 1 && c.FieldNo <= 4) {
                line = int(c.FieldNo);
                body = c.Bytes;
            }
        }
        if (!c.Ok()) {
            error = "Can't parse TProgram";
            return false;
        }
        bool ok = false;
        switch (line) {
            case 1:
                ok = builder.ExtractAssign(*step, body);
                break;
            case 3:
                ok = builder.ExtractFilter(*step, body);
                break;
            case 2:
                ok = builder.ExtractProjection(*step, body);
                if (ok) {
                    ssaProgram->Steps.push_back(step);
                    step = std::make_shared<TProgramStep>();
                }
                break;
            case 4:
                ok = builder.ExtractGroupBy(*step, body);
                if (ok) {
                    ssaProgram->Steps.push_back(step);
                    step = std::make_shared<TProgramStep>();
                }
                break;
            default:
                return false;  // LINE_NOT_SET
        }
        if (!ok) {
            error = builder.Error;
            return false;
        }
    }
    if (!r.Ok()) {
        error = "Can't parse TProgram";
        return false;
    }

    if (!step->Empty()) ssaProgram->Steps.push_back(step);  // final step without final projection
    ssaProgram->SourceColumns = std::move(builder.Sources);

    // 'SELECT count(*) FROM table' needs a column.
    if (ssaProgram->SourceColumns.empty()) {
        const TColumnInfo useless = columnResolver.GetDefaultColumn();
        ssaProgram->SourceColumns.emplace_back(useless.ColumnId, useless);
    }
    if (!ssaProgram->Steps.empty()) OptimizeProgram(*ssaProgram);
    Program = ssaProgram;
    return true;
}

}  // namespace ssa
