#include "plan.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <unordered_map>

#include "vm_ops.h"

namespace ssa {

namespace {

const char* TypeName(int t) {
    static const char* names[] = {"none", "bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32",
                                  "uint64", "float", "double", "timestamp[us]", "duration[us]", "string", "binary",
                                  "decimal128"};
    return (t >= 0 && t <= T_DEC128) ? names[t] : "?";
}

int BitWidth(int t) { return TypeWidth(t) * 8; }

// Arrow's CommonNumeric (compute/kernels/codegen_internal.cc:226-278): the narrowest numeric type
// both operands can be implicitly cast to; T_NONE if either is not a number.
int CommonNumeric(int t0, int t1) {
    if (!IsNumeric(t0) || !IsNumeric(t1)) return T_NONE;
    if (t0 == T_F64 || t1 == T_F64) return T_F64;
    if (t0 == T_F32 || t1 == T_F32) return T_F32;
    int maxSigned = 0, maxUnsigned = 0;
    for (int t : {t0, t1}) {
        int& w = IsSignedInt(t) ? maxSigned : maxUnsigned;
        w = std::max(w, BitWidth(t));
    }
    if (maxSigned == 0) {
        return maxUnsigned >= 64 ? T_U64 : maxUnsigned == 32 ? T_U32 : maxUnsigned == 16 ? T_U16 : T_U8;
    }
    if (maxSigned <= maxUnsigned) {
        int w = maxUnsigned + 1;  // next power of two that can hold the unsigned range
        int p = 8;
        while (p < w) p <<= 1;
        maxSigned = p;
    }
    return maxSigned >= 64 ? T_I64 : maxSigned == 32 ? T_I32 : maxSigned == 16 ? T_I16 : T_I8;
}

// Result type of the YQL integral / real binary kernels
// (ydb/library/yql/minikql/invoke_builtins/mkql_builtins_impl.h.txt:1415-1535).
int YqlArithType(int t0, int t1) {
    if (!IsNumeric(t0) || !IsNumeric(t1)) return T_NONE;
    if (t0 == T_F64 || t1 == T_F64) return T_F64;
    if (t0 == T_F32 || t1 == T_F32) return T_F32;
    int w0 = BitWidth(t0), w1 = BitWidth(t1);
    if (w0 != w1) return w0 > w1 ? t0 : t1;
    if (IsSignedInt(t0)) return t0;
    return t1;
}

struct Expr;
using ExprPtr = std::shared_ptr<Expr>;

struct Expr {
    enum EKind { COL, CONST, ALIAS, UNARY, BINARY, STROP, SELECT } Kind = COL;
    int32_t Type = T_NONE;
    bool Nullable = false;
    int ColSlot = -1;     // COL: VM column slot
    int InputIndex = -1;  // COL: index in the step's input batch
    TScalar Scalar;       // CONST
    int Op = 0, A = 0, B = 0;
    ExprPtr L, R;
    ExprPtr C;  // SELECT: L = predicate, R = then, C = else
    // codegen state
    int Uses = 0;
    int Reg = -1;
    bool Done = false;
    bool IsLeaf() const { return Kind == COL || Kind == CONST; }
};

ExprPtr MakeConst(const TScalar& s) {
    auto e = std::make_shared<Expr>();
    e->Kind = Expr::CONST;
    e->Type = s.Type;
    e->Nullable = !s.Valid;
    e->Scalar = s;
    return e;
}

TScalar NumScalar(int type, uint64_t bits, bool valid = true) {
    TScalar s;
    s.Type = type;
    s.Valid = valid;
    s.Bits = bits;
    return s;
}

struct Compiler {
    const TProgramStep& Step;
    const std::vector<FieldDesc>& Input;
    StepPlan& Plan;
    std::vector<std::pair<std::string, ExprPtr>> Env;  // TDatumBatch columns in schema order
    std::vector<ExprPtr> ColExprs;                     // per input column (lazy)
    int NextTemp = 0;
    int NumOuts = 0;
    bool SerializedKeys = false;

    Compiler(const TProgramStep& step, const std::vector<FieldDesc>& input, StepPlan& plan)
        : Step(step), Input(input), Plan(plan) {}

    ExprPtr Lookup(const std::string& name) const {
        ExprPtr found;
        int count = 0;
        for (auto& kv : Env) {
            if (kv.first == name) {
                found = kv.second;
                ++count;
            }
        }
        return count == 1 ? found : nullptr;  // arrow::Schema::GetFieldIndex: -1 if absent or duplicated
    }

    ExprPtr InputColumn(int idx) {
        if (ColExprs[idx]) return ColExprs[idx];
        const FieldDesc& f = Input[idx];
        ExprPtr e;
        if (f.IsScalar) {
            e = MakeConst(f.Scalar);
        } else {
            e = std::make_shared<Expr>();
            e->Kind = Expr::COL;
            e->Type = f.Type;
            e->Nullable = f.Nullable;
            e->InputIndex = idx;
        }
        ColExprs[idx] = e;
        return e;
    }

    // ---- casts -----------------------------------------------------------------------------
    static bool LosslessWiden(int from, int to) {
        if (from == to) return true;
        if (from == T_BOOL) return IsInt(to);
        if (IsSignedInt(from) && IsSignedInt(to)) return BitWidth(to) >= BitWidth(from);
        if (IsUnsignedInt(from) && IsUnsignedInt(to)) return BitWidth(to) >= BitWidth(from);
        if (IsUnsignedInt(from) && IsSignedInt(to)) return BitWidth(to) > BitWidth(from);
        if (from == T_F32 && to == T_F64) return true;
        if (IsInt(from) && to == T_F64) return BitWidth(from) <= 32;  // every value is exactly representable
        if (IsInt(from) && to == T_F32) return BitWidth(from) <= 16;
        if (IsTemporal(from) && to == T_I64) return true;
        if (from == T_I64 && IsTemporal(to)) return true;
        if (from == T_U32 && to == T_TS) return true;
        return false;
    }

    Status Unary(int op, int a, int b, ExprPtr arg, int type, bool nullable, ExprPtr& out) {
        if (arg->Kind == Expr::CONST) {  // fold
            TScalar s = arg->Scalar;
            uint64_t v = s.Bits;
            bool valid = s.Valid;
            switch (op) {
                case OP_WRAP: v = WrapTo(v, a); break;
                case OP_CAST: v = CastVal(v, a, b); break;
                case OP_CHECK:
                    if (valid && !FitsType(v, a, b))
                        return Status::Invalid(std::string("Integer value not in range of ") + TypeName(b));
                    break;
                case OP_NOT: v = v ? 0 : 1; break;
                case OP_ISNULL: v = valid ? 0 : 1; valid = true; break;
                case OP_ISVALID: v = valid ? 1 : 0; valid = true; break;
                case OP_FUNC: v = ApplyFunc(b, a, v, 0).v; break;
                default: return Status::Unsupported("cannot fold unary op");
            }
            out = MakeConst(NumScalar(type, v, valid));
            if (IsArrayShaped(arg)) ArrayShapedConsts.push_back(out.get());  // a folded array stays an array
            return Status::OK();
        }
        auto e = std::make_shared<Expr>();
        e->Kind = Expr::UNARY;
        e->Op = op;
        e->A = a;
        e->B = b;
        e->L = arg;
        e->Type = type;
        e->Nullable = nullable;
        out = e;
        return Status::OK();
    }

    // Numeric cast with Arrow semantics; `safe` adds the range check of CastOptions::Safe.
    Status Cast(ExprPtr arg, int to, bool safe, ExprPtr& out) {
        const int from = arg->Type;
        if (from == to) {
            out = arg;
            return Status::OK();
        }
        const bool fromOk = IsNumeric(from) || from == T_BOOL || IsTemporal(from);
        const bool toOk = IsNumeric(to) || to == T_BOOL || IsTemporal(to);
        if (!fromOk || !toOk) {
            return Status::Unsupported(std::string("cast from ") + TypeName(from) + " to " + TypeName(to));
        }
        const int cls = ClassOfType(from);
        ExprPtr cur = arg;
        if (safe && !LosslessWiden(from, to) && (cls == CLS_I || cls == CLS_U)) {
            Status st = Unary(OP_CHECK, cls, to, cur, from, cur->Nullable, cur);
            if (!st.ok()) return st;
        }
        if (LosslessWiden(from, to) && !(IsFloat(to) && !IsFloat(from))) {
            if (cur->Kind == Expr::CONST) {
                TScalar s = cur->Scalar;
                s.Type = to;
                out = MakeConst(s);
                if (IsArrayShaped(cur)) ArrayShapedConsts.push_back(out.get());
                return Status::OK();
            }
            auto e = std::make_shared<Expr>();  // payload unchanged, only the static type changes
            e->Kind = Expr::ALIAS;
            e->L = cur;
            e->Type = to;
            e->Nullable = cur->Nullable;
            out = e;
            return Status::OK();
        }
        return Unary(OP_CAST, cls, to, cur, to, cur->Nullable, out);
    }

    // ---- binary nodes ----------------------------------------------------------------------
    Status Binary(int op, int a, int b, ExprPtr l, ExprPtr r, int type, bool nullable, int wrapType, ExprPtr& out) {
        if (l->Kind == Expr::CONST && r->Kind == Expr::CONST) {
            const TScalar& x = l->Scalar;
            const TScalar& y = r->Scalar;
            uint64_t v = 0;
            bool valid = x.Valid && y.Valid;
            switch (op) {
                case OP_ADD: case OP_SUB: case OP_MUL: v = ArithBasic(op, a, x.Bits, y.Bits); break;
                case OP_DIV: case OP_MOD: {
                    if ((b & 1) == DIALECT_ARROW && !valid) {
                        v = 0;
                        break;
                    }
                    ArithResult res = ArithDiv(op == OP_MOD, a, b & 1, b >> 1, x.Bits, y.Bits);
                    if (res.error) return Status::Invalid("divide by zero");
                    if (res.null) valid = false;
                    v = res.v;
                    break;
                }
                case OP_CMP: v = CompareVals(b, a, x.Bits, y.Bits); break;
                case OP_FUNC: {
                    if (IsNotNullFunc(b) && !valid) break;  // v = 0
                    ArithResult res = ApplyFunc(b, a, x.Bits, y.Bits);
                    if (res.error) return Status::Invalid("divide by zero");
                    if (res.overflow) return Status::Invalid("overflow");
                    v = res.v;
                    break;
                }
                case OP_AND:
                    v = (x.Bits && y.Bits);
                    if (b == DIALECT_YQL) valid = (x.Valid && y.Valid) || (x.Valid && !x.Bits) || (y.Valid && !y.Bits);
                    break;
                case OP_OR:
                    v = (x.Bits || y.Bits);
                    if (b == DIALECT_YQL) valid = (x.Valid && y.Valid) || (x.Valid && x.Bits) || (y.Valid && y.Bits);
                    break;
                case OP_XOR: v = (x.Bits != 0) != (y.Bits != 0); break;
                case OP_COALESCE:
                    v = x.Valid ? x.Bits : y.Bits;
                    valid = x.Valid || y.Valid;
                    break;
                default: return Status::Unsupported("cannot fold binary op");
            }
            if (wrapType) v = WrapTo(v, wrapType);
            out = MakeConst(NumScalar(type, v, valid));
            if (IsArrayShaped(l) || IsArrayShaped(r)) ArrayShapedConsts.push_back(out.get());
            return Status::OK();
        }
        auto e = std::make_shared<Expr>();
        e->Kind = Expr::BINARY;
        e->Op = op;
        e->A = a;
        e->B = b;
        e->L = l;
        e->R = r;
        e->Type = type;
        e->Nullable = nullable;
        out = e;
        if (wrapType && BitWidth(wrapType) < 64) {
            return Unary(OP_WRAP, wrapType, 0, e, type, nullable, out);
        }
        return Status::OK();
    }

    static std::string NoKernel(const char* fn, int t0, int t1) {
        return std::string("Function '") + fn + "' has no kernel matching input types (" + TypeName(t0) + ", " +
               TypeName(t1) + ")";
    }

    // String-valued operand: a string column or a string constant.
    static bool IsStrLeaf(const ExprPtr& e) { return IsStringLike(e->Type) && e->IsLeaf(); }

    Status StrCompareNode(int cmp, ExprPtr l, ExprPtr r, int outType, ExprPtr& out) {
        if (!IsStrLeaf(l) || !IsStrLeaf(r)) return Status::Unsupported("string compare on computed strings");
        if (l->Kind == Expr::CONST && r->Kind == Expr::CONST) {
            const std::string& x = l->Scalar.Bytes;
            const std::string& y = r->Scalar.Bytes;
            int ord = ssa::StrCompare(reinterpret_cast<const uint8_t*>(x.data()), int32_t(x.size()),
                                 reinterpret_cast<const uint8_t*>(y.data()), int32_t(y.size()));
            out = MakeConst(NumScalar(outType, CmpFromOrder(cmp, ord), l->Scalar.Valid && r->Scalar.Valid));
            return Status::OK();
        }
        if (l->Kind == Expr::CONST) {  // flip so that the column is on the left
            std::swap(l, r);
            cmp = cmp == CMP_LT ? CMP_GT : cmp == CMP_GT ? CMP_LT : cmp == CMP_LE ? CMP_GE : cmp == CMP_GE ? CMP_LE : cmp;
        }
        auto e = std::make_shared<Expr>();
        e->Kind = Expr::STROP;
        e->Op = OP_STRCMP;
        e->B = cmp;
        e->L = l;
        e->R = r;
        e->Type = outType;
        e->Nullable = l->Nullable || r->Nullable;
        out = e;
        return Status::OK();
    }

    Status StrMatchNode(int kind, bool icase, ExprPtr col, const std::string& pattern, bool patternValid, int outType,
                        ExprPtr& out) {
        if (!IsStrLeaf(col)) return Status::Unsupported("string match on a computed string");
        if (col->Kind == Expr::CONST) {
            const std::string& s = col->Scalar.Bytes;
            bool m = ssa::StrMatch(kind, icase, reinterpret_cast<const uint8_t*>(s.data()), int32_t(s.size()),
                              reinterpret_cast<const uint8_t*>(pattern.data()), int32_t(pattern.size()));
            out = MakeConst(NumScalar(outType, m, col->Scalar.Valid && patternValid));
            return Status::OK();
        }
        TScalar p;
        p.Type = T_BINARY;
        p.Valid = patternValid;
        p.Bytes = pattern;
        auto e = std::make_shared<Expr>();
        e->Kind = Expr::STROP;
        e->Op = OP_STRMATCH;
        e->B = kind | (icase ? 16 : 0);
        e->L = col;
        e->R = MakeConst(p);
        e->Type = outType;
        e->Nullable = col->Nullable || !patternValid;
        out = e;
        return Status::OK();
    }

    // ---- SIMPLE_ARROW functions (Arrow 5.0 semantics) -----------------------------------------
    Status ArrowCompare(int cmp, const char* fn, ExprPtr l, ExprPtr r, ExprPtr& out) {
        int t0 = l->Type, t1 = r->Type;
        if (IsStringLike(t0) && IsStringLike(t1)) return StrCompareNode(cmp, l, r, T_BOOL, out);
        int common = T_NONE;
        if (t0 == t1 && (IsNumeric(t0) || t0 == T_BOOL || IsTemporal(t0))) {
            common = t0;
        } else {
            common = CommonNumeric(t0, t1);
        }
        if (common == T_NONE) return Status::Invalid(NoKernel(fn, t0, t1));
        Status st = Cast(l, common, true, l);
        if (!st.ok()) return st;
        st = Cast(r, common, true, r);
        if (!st.ok()) return st;
        int cls = ClassOfType(common);
        if (cls == CLS_F32) cls = CLS_F64;
        return Binary(OP_CMP, cls, cmp, l, r, T_BOOL, l->Nullable || r->Nullable, 0, out);
    }

    Status ArrowArith(int op, const char* fn, ExprPtr l, ExprPtr r, ExprPtr& out) {
        int t0 = l->Type, t1 = r->Type;
        int common = (t0 == t1 && IsNumeric(t0)) ? t0 : CommonNumeric(t0, t1);
        if (common == T_NONE) return Status::Invalid(NoKernel(fn, t0, t1));
        Status st = Cast(l, common, true, l);
        if (!st.ok()) return st;
        st = Cast(r, common, true, r);
        if (!st.ok()) return st;
        int cls = ClassOfType(common);
        bool nullable = l->Nullable || r->Nullable;
        int wrap = IsInt(common) ? common : 0;
        int b = (op == OP_DIV) ? (DIALECT_ARROW | (common << 1)) : DIALECT_ARROW;  // DIV/MOD carry the result type
        return Binary(op, cls, b, l, r, common, nullable, wrap, out);
    }

    Status ArrowLogic(int op, const char* fn, ExprPtr l, ExprPtr r, ExprPtr& out) {
        if (l->Type != T_BOOL || r->Type != T_BOOL) return Status::Invalid(NoKernel(fn, l->Type, r->Type));
        return Binary(op, CLS_U, DIALECT_ARROW, l, r, T_BOOL, l->Nullable || r->Nullable, 0, out);
    }

    // gcd / lcm / mod / modOrZero (ydb/library/arrow_kernels/func_common.h:81-143): both operands implicitly cast to the
    // common numeric type, which is also the result type; gcd and lcm have integer kernels only.
    Status CustomBinary(int fn, const char* name, ExprPtr l, ExprPtr r, ExprPtr& out) {
        const int t0 = l->Type, t1 = r->Type;
        const bool intsOnly = (fn == FN_GCD || fn == FN_LCM);
        auto okType = [&](int t) { return intsOnly ? IsInt(t) : IsNumeric(t); };
        int common = (t0 == t1 && okType(t0)) ? t0 : CommonNumeric(t0, t1);
        if (common == T_NONE || !okType(common)) return Status::Invalid(NoKernel(name, t0, t1));
        Status st = Cast(l, common, true, l);
        if (!st.ok()) return st;
        st = Cast(r, common, true, r);
        if (!st.ok()) return st;
        return Binary(OP_FUNC, common, fn, l, r, common, l->Nullable || r->Nullable, 0, out);
    }

    // abs / negate / round / roundBankers / roundToExp2: one kernel per numeric type, result of the same type
    // (func_common.h:146-154; compute/kernels/scalar_arithmetic.cc:1205-1214).
    Status SameTypeUnary(int fn, const char* name, ExprPtr arg, ExprPtr& out) {
        const int t = arg->Type;
        if (!IsNumeric(t)) return Status::Invalid(NoKernel(name, t, T_NONE));
        return Unary(OP_FUNC, t, fn, arg, t, arg->Nullable, out);
    }

    // floor / ceil / trunc: floating kernels only, integers are implicitly cast to double
    // (compute/kernels/scalar_arithmetic.cc:1151-1179,1281-1292).
    Status FloatingUnary(int fn, const char* name, ExprPtr arg, ExprPtr& out) {
        if (!IsNumeric(arg->Type)) return Status::Invalid(NoKernel(name, arg->Type, T_NONE));
        if (IsInt(arg->Type)) {
            Status st = Cast(arg, T_F64, true, arg);
            if (!st.ok()) return st;
        }
        return Unary(OP_FUNC, arg->Type, fn, arg, arg->Type, arg->Nullable, out);
    }

    // sqrt: a kernel per numeric type with a double result; a float argument is computed in float
    // (ydb/library/arrow_kernels/func_common.h:156-164, execs.h:113-143).
    Status MathUnary(int fn, const char* name, ExprPtr arg, ExprPtr& out) {
        if (!IsNumeric(arg->Type)) return Status::Invalid(NoKernel(name, arg->Type, T_NONE));
        if (!IsFloat(arg->Type)) {
            Status st = Cast(arg, T_F64, false, arg);
            if (!st.ok()) return st;
        }
        return Unary(OP_FUNC, arg->Type, fn, arg, T_F64, arg->Nullable, out);
    }

    // add_checked / subtract_checked / multiply_checked / divide_checked: typed like their unchecked forms
    // (compute/kernels/scalar_arithmetic.cc:1193-1203).
    Status CheckedArith(int fn, const char* name, ExprPtr l, ExprPtr r, ExprPtr& out) {
        const int t0 = l->Type, t1 = r->Type;
        int common = (t0 == t1 && IsNumeric(t0)) ? t0 : CommonNumeric(t0, t1);
        if (common == T_NONE) return Status::Invalid(NoKernel(name, t0, t1));
        Status st = Cast(l, common, true, l);
        if (!st.ok()) return st;
        st = Cast(r, common, true, r);
        if (!st.ok()) return st;
        return Binary(OP_FUNC, common, fn, l, r, common, l->Nullable || r->Nullable, 0, out);
    }

    // hypot: both operands implicitly cast to the common numeric type, double result (func_common.h:166-174).
    Status MathBinary(int fn, const char* name, ExprPtr l, ExprPtr r, ExprPtr& out) {
        const int t0 = l->Type, t1 = r->Type;
        int common = (t0 == t1 && IsNumeric(t0)) ? t0 : CommonNumeric(t0, t1);
        if (common == T_NONE) return Status::Invalid(NoKernel(name, t0, t1));
        Status st = Cast(l, common, true, l);
        if (!st.ok()) return st;
        st = Cast(r, common, true, r);
        if (!st.ok()) return st;
        if (!IsFloat(common)) {
            st = Cast(l, T_F64, false, l);
            if (!st.ok()) return st;
            st = Cast(r, T_F64, false, r);
            if (!st.ok()) return st;
        }
        return Binary(OP_FUNC, l->Type, fn, l, r, T_F64, l->Nullable || r->Nullable, 0, out);
    }

    // ydb.cast with CastOptions(safe=false): ydb/library/arrow_kernels/func_cast.cpp:101-149.
    Status YdbCast(ExprPtr arg, int to, ExprPtr& out) {
        const int from = arg->Type;
        if (from == to) {
            out = arg;
            return Status::OK();
        }
        if (to == T_TS && from != T_U32) {
            return Status::Invalid(std::string("Function 'ydb_cast_timestamp' has no kernel matching input types (") +
                                   TypeName(from) + ")");
        }
        if (IsTemporal(from) && to != T_I64) {
            return Status::Unsupported(std::string("cast from ") + TypeName(from) + " to " + TypeName(to));
        }
        return Cast(arg, to, false, out);
    }

    // match_like: only patterns that Arrow itself rewrites to substring / prefix / suffix search.
    static bool SimplifyLike(const std::string& p, int& kind, std::string& lit) {
        auto plain = [](const std::string& s) {
            return s.find('%') == std::string::npos && s.find('_') == std::string::npos &&
                   s.find('\\') == std::string::npos;
        };
        if (p.size() >= 2 && p.front() == '%' && p.back() == '%' && plain(p.substr(1, p.size() - 2))) {
            kind = STR_CONTAINS;
            lit = p.substr(1, p.size() - 2);
            return true;
        }
        if (p.size() >= 1 && p.back() == '%' && plain(p.substr(0, p.size() - 1))) {
            kind = STR_STARTS;
            lit = p.substr(0, p.size() - 1);
            return true;
        }
        if (p.size() >= 1 && p.front() == '%' && plain(p.substr(1))) {
            kind = STR_ENDS;
            lit = p.substr(1);
            return true;
        }
        return false;
    }

    Status SimpleFunction(const TAssign& a, std::vector<ExprPtr>& args, ExprPtr& out) {
        const char* fn = GetFunctionName(a.Operation);
        switch (a.Operation) {
            case EOperation::Equal: return ArrowCompare(CMP_EQ, fn, args[0], args[1], out);
            case EOperation::NotEqual: return ArrowCompare(CMP_NE, fn, args[0], args[1], out);
            case EOperation::Less: return ArrowCompare(CMP_LT, fn, args[0], args[1], out);
            case EOperation::LessEqual: return ArrowCompare(CMP_LE, fn, args[0], args[1], out);
            case EOperation::Greater: return ArrowCompare(CMP_GT, fn, args[0], args[1], out);
            case EOperation::GreaterEqual: return ArrowCompare(CMP_GE, fn, args[0], args[1], out);
            case EOperation::Add: return ArrowArith(OP_ADD, fn, args[0], args[1], out);
            case EOperation::Subtract: return ArrowArith(OP_SUB, fn, args[0], args[1], out);
            case EOperation::Multiply: return ArrowArith(OP_MUL, fn, args[0], args[1], out);
            case EOperation::Divide: return ArrowArith(OP_DIV, fn, args[0], args[1], out);
            case EOperation::And: return ArrowLogic(OP_AND, fn, args[0], args[1], out);
            case EOperation::Or: return ArrowLogic(OP_OR, fn, args[0], args[1], out);
            case EOperation::Xor: return ArrowLogic(OP_XOR, fn, args[0], args[1], out);
            case EOperation::Invert:
                if (args[0]->Type != T_BOOL) return Status::Invalid(NoKernel(fn, args[0]->Type, T_NONE));
                return Unary(OP_NOT, 0, 0, args[0], T_BOOL, args[0]->Nullable, out);
            case EOperation::IsNull:
                if (!args[0]->Nullable && args[0]->Kind != Expr::CONST) {
                    out = MakeConstArrayFalse();
                    return Status::OK();
                }
                return Unary(OP_ISNULL, 0, 0, args[0], T_BOOL, false, out);
            case EOperation::IsValid:
                return Unary(OP_ISVALID, 0, 0, args[0], T_BOOL, false, out);
            case EOperation::Gcd: return CustomBinary(FN_GCD, fn, args[0], args[1], out);
            case EOperation::Lcm: return CustomBinary(FN_LCM, fn, args[0], args[1], out);
            case EOperation::Modulo: return CustomBinary(FN_MOD, fn, args[0], args[1], out);
            case EOperation::ModuloOrZero: return CustomBinary(FN_MODZ, fn, args[0], args[1], out);
            case EOperation::AddNotNull: return CheckedArith(FN_ADD_CHECKED, fn, args[0], args[1], out);
            case EOperation::SubtractNotNull: return CheckedArith(FN_SUB_CHECKED, fn, args[0], args[1], out);
            case EOperation::MultiplyNotNull: return CheckedArith(FN_MUL_CHECKED, fn, args[0], args[1], out);
            case EOperation::DivideNotNull: return CheckedArith(FN_DIV_CHECKED, fn, args[0], args[1], out);
            case EOperation::Abs: return SameTypeUnary(FN_ABS, fn, args[0], out);
            case EOperation::Negate: return SameTypeUnary(FN_NEG, fn, args[0], out);
            case EOperation::Round: case EOperation::RoundBankers: case EOperation::RoundToExp2:
                if (args.size() != 1) return Status::Invalid(std::string("Function '") + fn + "' accepts 1 arguments");
                return SameTypeUnary(a.Operation == EOperation::Round ? FN_ROUND
                                     : a.Operation == EOperation::RoundBankers ? FN_ROUNDBANKERS : FN_ROUNDEXP2, fn, args[0], out);
            case EOperation::Floor: case EOperation::Ceil: case EOperation::Trunc:
                if (args.size() != 1) return Status::Invalid(std::string("Function '") + fn + "' accepts 1 arguments");
                return FloatingUnary(a.Operation == EOperation::Floor ? FN_FLOOR : a.Operation == EOperation::Ceil ? FN_CEIL : FN_TRUNC,
                                     fn, args[0], out);
            case EOperation::Sqrt: case EOperation::Acosh: case EOperation::Atanh: case EOperation::Cbrt: case EOperation::Cosh:
            case EOperation::Erf: case EOperation::Erfc: case EOperation::Exp: case EOperation::Exp2: case EOperation::Lgamma:
            case EOperation::Sinh: case EOperation::Tgamma: {
                if (args.size() != 1) return Status::Invalid(std::string("Function '") + fn + "' accepts 1 arguments");
                int id = FN_SQRT;
                switch (a.Operation) {
                    case EOperation::Acosh: id = FN_ACOSH; break;
                    case EOperation::Atanh: id = FN_ATANH; break;
                    case EOperation::Cbrt: id = FN_CBRT; break;
                    case EOperation::Cosh: id = FN_COSH; break;
                    case EOperation::Erf: id = FN_ERF; break;
                    case EOperation::Erfc: id = FN_ERFC; break;
                    case EOperation::Exp: id = FN_EXP; break;
                    case EOperation::Exp2: id = FN_EXP2; break;
                    case EOperation::Lgamma: id = FN_LGAMMA; break;
                    case EOperation::Sinh: id = FN_SINH; break;
                    case EOperation::Tgamma: id = FN_TGAMMA; break;
                    default: break;
                }
                return MathUnary(id, fn, args[0], out);
            }
            case EOperation::Hypot:
                if (args.size() != 2) return Status::Invalid(std::string("Function '") + fn + "' accepts 2 arguments");
                return MathBinary(FN_HYPOT, fn, args[0], args[1], out);
            case EOperation::Exp10:  // not in the registry (ydb/core/formats/arrow/custom_registry.cpp:35-38)
                return Status::Invalid("Key error: No function registered with name: exp10");
            case EOperation::E: case EOperation::Pi: {
                // nullary kernels that return a double scalar (func_common.h:116-122, func_math.h:47-56,131-140)
                if (!args.empty()) return Status::Invalid(std::string("Function '") + fn + "' accepts 0 arguments");
                const double v = a.Operation == EOperation::E ? 2.7182818284590452353602874713526624977572470
                                                              : 3.1415926535897932384626433832795028841971693;
                out = MakeConst(NumScalar(T_F64, F64ToBits(v), true));
                return Status::OK();
            }
            case EOperation::CastBoolean: case EOperation::CastInt8: case EOperation::CastInt16:
            case EOperation::CastInt32: case EOperation::CastInt64: case EOperation::CastUInt8:
            case EOperation::CastUInt16: case EOperation::CastUInt32: case EOperation::CastUInt64:
            case EOperation::CastFloat: case EOperation::CastDouble: case EOperation::CastTimestamp:
                return YdbCast(args[0], a.CastTo, out);
            case EOperation::BinaryLength: {
                if (!IsStrLeaf(args[0])) return Status::Invalid(NoKernel(fn, args[0]->Type, T_NONE));
                if (args[0]->Kind == Expr::CONST) {
                    out = MakeConst(NumScalar(T_I32, args[0]->Scalar.Bytes.size(), args[0]->Scalar.Valid));
                    return Status::OK();
                }
                auto e = std::make_shared<Expr>();
                e->Kind = Expr::STROP;
                e->Op = OP_STRLEN;
                e->L = args[0];
                e->Type = T_I32;
                e->Nullable = args[0]->Nullable;
                out = e;
                return Status::OK();
            }
            case EOperation::MatchSubstring:
                if (!a.HasMatchOptions) return Status::Invalid("MatchSubstringOptions required");
                if (!IsStringLike(args[0]->Type)) return Status::Invalid(NoKernel(fn, args[0]->Type, T_NONE));
                if (a.IgnoreCase && NeedsUnicodeFold(a.Pattern)) return Status::Unsupported(kUnicodeFold);
                return StrMatchNode(STR_CONTAINS, a.IgnoreCase, args[0], a.Pattern, true, T_BOOL, out);
            case EOperation::StartsWith:
                if (!a.HasMatchOptions) return Status::Invalid("MatchSubstringOptions required");
                if (!IsStringLike(args[0]->Type)) return Status::Invalid(NoKernel(fn, args[0]->Type, T_NONE));
                if (a.IgnoreCase && NeedsUnicodeFold(a.Pattern)) return Status::Unsupported(kUnicodeFold);
                return StrMatchNode(STR_STARTS, a.IgnoreCase, args[0], a.Pattern, true, T_BOOL, out);
            case EOperation::EndsWith:
                if (!a.HasMatchOptions) return Status::Invalid("MatchSubstringOptions required");
                if (!IsStringLike(args[0]->Type)) return Status::Invalid(NoKernel(fn, args[0]->Type, T_NONE));
                if (a.IgnoreCase && NeedsUnicodeFold(a.Pattern)) return Status::Unsupported(kUnicodeFold);
                return StrMatchNode(STR_ENDS, a.IgnoreCase, args[0], a.Pattern, true, T_BOOL, out);
            case EOperation::MatchLike: {
                if (!a.HasMatchOptions) return Status::Invalid("MatchSubstringOptions required");
                if (!IsStringLike(args[0]->Type)) return Status::Invalid(NoKernel(fn, args[0]->Type, T_NONE));
                int kind;
                std::string lit;
                if (!SimplifyLike(a.Pattern, kind, lit)) {
                    // Arrow hands every other pattern to RE2 (MakeLikeRegex); case-insensitive matching there folds
                    // Unicode, which this path does not do (the program builder never asks for it with LIKE)
                    if (a.IgnoreCase) return Status::Unsupported("case-insensitive general LIKE pattern");
                    return StrMatchNode(STR_LIKE, false, args[0], a.Pattern, true, T_BOOL, out);
                }
                if (a.IgnoreCase && NeedsUnicodeFold(lit)) return Status::Unsupported(kUnicodeFold);
                return StrMatchNode(kind, a.IgnoreCase, args[0], lit, true, T_BOOL, out);
            }
            default:
                return Status::Unsupported("operation is not available on the GPU path");
        }
    }

    // The reference matches ignore_case patterns with RE2, which folds Unicode: a pattern byte >= 0x80 starts a character
    // with a case orbit of its own, and 'k' / 's' also match U+212A KELVIN SIGN / U+017F LONG S.  The device folds ASCII
    // letters only -- the same thing for every other pattern -- so those patterns go back to the CPU.
    static constexpr const char* kUnicodeFold = "case-insensitive match whose pattern needs Unicode case folding";
    static bool NeedsUnicodeFold(const std::string& pattern) {
        for (unsigned char ch : pattern) {
            if (ch >= 0x80 || ch == 'k' || ch == 'K' || ch == 's' || ch == 'S') return true;
        }
        return false;
    }

    static ExprPtr DecimalPart(const ExprPtr& col, int part) {
        auto e = std::make_shared<Expr>();
        e->Kind = Expr::STROP;
        e->Op = OP_DECPART;
        e->A = part;
        e->L = col;
        e->Type = T_U64;
        e->Nullable = col->Nullable;
        return e;
    }

    // A non-null input to is_null yields an all-false array; keep it array-shaped (not a scalar).
    ExprPtr MakeConstArrayFalse() {
        auto e = MakeConst(NumScalar(T_BOOL, 0, true));
        ArrayShapedConsts.push_back(e.get());
        return e;
    }
    std::vector<Expr*> ArrayShapedConsts;
    bool IsArrayShaped(const ExprPtr& e) const {
        if (e->Kind != Expr::CONST) return true;
        return std::find(ArrayShapedConsts.begin(), ArrayShapedConsts.end(), e.get()) != ArrayShapedConsts.end();
    }

    // ---- YQL_KERNEL functions (block kernels selected by YqlOperationId) ----------------------
    // BlockIf (ydb/library/yql/minikql/comp_nodes/mkql_block_if.cpp:120-250): a Bool predicate that is not optional,
    // branches of one type; each row takes the value and the nullness of the branch its predicate byte selects.
    Status YqlIf(ExprPtr pred, ExprPtr th, ExprPtr el, ExprPtr& out) {
        if (!(pred->Type == T_U8 || pred->Type == T_BOOL) || pred->Nullable) return Status::Invalid("Expected bool as first argument");
        if (th->Type != el->Type) return Status::Invalid("Different return types in branches.");
        if (IsStringLike(th->Type) || th->Type == T_DEC128) return Status::Unsupported("If over string or decimal branches");
        if (pred->Kind == Expr::CONST) {  // TBlockIfScalarWrapper: the chosen branch as it is
            out = pred->Scalar.Bits ? th : el;
            if (IsArrayShaped(pred) && out->Kind == Expr::CONST && !IsArrayShaped(out)) {
                out = MakeConst(out->Scalar);
                ArrayShapedConsts.push_back(out.get());
            }
            return Status::OK();
        }
        auto e = std::make_shared<Expr>();
        e->Kind = Expr::SELECT;
        e->L = pred;
        e->R = th;
        e->C = el;
        e->Type = th->Type;
        e->Nullable = th->Nullable || el->Nullable;
        out = e;
        return Status::OK();
    }

    // TKernelRequestBuilder::AddUnaryOp (ydb/library/yql/core/arrow_kernels/request/request.cpp:25-46):
    //   Not   BlockNot (minikql/comp_nodes/mkql_block_logical.cpp:379-417): a Bool block (UInt8 0 / 1), nulls stay nulls
    //   Just  BlockJust (mkql_block_just.cpp): the values as they are, now of an optional type -- the same Arrow array
    //   Size  BlockFunc("Size"): the byte length of a string as Uint32 (kqp_olap_compiler.cpp:663-665)
    //   Minus, Abs  BlockFunc of the argument's own type: two's-complement wrap for integers (Abs of an unsigned is the
    //         value itself), sign operations for floats (mkql_builtins_minus.cpp, mkql_builtins_abs.cpp)
    Status YqlUnary(EYqlOp op, ExprPtr arg, ExprPtr& out) {
        const int t = arg->Type;
        switch (op) {
            case EYqlOp::Not: {
                if (t == T_BOOL) return Unary(OP_NOT, 0, 0, arg, T_BOOL, arg->Nullable, out);
                if (t != T_U8) return Status::Unsupported(std::string("YQL Not of ") + TypeName(t));
                // (a Bool block holds 0 or 1: !x is x == 0)
                return Binary(OP_CMP, CLS_U, CMP_EQ, arg, MakeConst(NumScalar(T_U8, 0, true)), T_U8, arg->Nullable, 0, out);
            }
            case EYqlOp::Just:
                // (Just of an optional nests optionals: Optional<Optional<T>> blocks are structs, not this array)
                if (arg->Nullable) return Status::Unsupported("YQL Just of an optional value");
                out = arg;
                return Status::OK();
            case EYqlOp::Size: {
                if (!IsStrLeaf(arg)) return Status::Unsupported(std::string("YQL Size of ") + TypeName(t));
                if (arg->Kind == Expr::CONST) {
                    out = MakeConst(NumScalar(T_U32, arg->Scalar.Bytes.size(), arg->Scalar.Valid));
                    return Status::OK();
                }
                auto e = std::make_shared<Expr>();
                e->Kind = Expr::STROP;
                e->Op = OP_STRLEN;
                e->L = arg;
                e->Type = T_U32;
                e->Nullable = arg->Nullable;
                out = e;
                return Status::OK();
            }
            case EYqlOp::Minus: case EYqlOp::Abs:
                if (!IsNumeric(t)) return Status::Unsupported(std::string("YQL ") + (op == EYqlOp::Abs ? "Abs" : "Minus") + " of " + TypeName(t));
                return Unary(OP_FUNC, t, op == EYqlOp::Abs ? FN_ABS : FN_NEG, arg, t, arg->Nullable, out);
            default:
                return Status::Unsupported("unknown unary YQL kernel");
        }
    }

    Status YqlFunction(const TAssign& a, std::vector<ExprPtr>& args, ExprPtr& out) {
        if (!a.YqlOperationId) return Status::Unsupported("YQL kernel without YqlOperationId");
        if (EYqlOp(*a.YqlOperationId) == EYqlOp::If) {
            if (args.size() != 3) return Status::Invalid("Expected 3 args");
            return YqlIf(args[0], args[1], args[2], out);
        }
        if (*a.YqlOperationId >= kYqlUnaryBase && *a.YqlOperationId <= uint32_t(EYqlOp::Just)) {
            if (args.size() != 1) return Status::Invalid("Expected 1 arg");
            return YqlUnary(EYqlOp(*a.YqlOperationId), args[0], out);
        }
        if (args.size() != 2) return Status::Unsupported("YQL kernel arity");
        ExprPtr l = args[0], r = args[1];
        const int t0 = l->Type, t1 = r->Type;
        const bool nullable = l->Nullable || r->Nullable;
        const EYqlOp op = EYqlOp(*a.YqlOperationId);
        auto asBool = [&](ExprPtr& e) {  // YQL Bool is a UInt8 0/1 block; accept Arrow booleans too
            return e->Type == T_U8 || e->Type == T_BOOL;
        };
        switch (op) {
            case EYqlOp::Equals: case EYqlOp::NotEquals: case EYqlOp::Less: case EYqlOp::LessOrEqual:
            case EYqlOp::Greater: case EYqlOp::GreaterOrEqual: {
                static const int cmps[] = {CMP_EQ, CMP_NE, CMP_LT, CMP_LE, CMP_GT, CMP_GE};
                const int cmp = cmps[int(op) - int(EYqlOp::Equals)];
                if (IsStringLike(t0) && IsStringLike(t1)) return StrCompareNode(cmp, l, r, T_U8, out);
                int cls;
                if (IsNumeric(t0) && IsNumeric(t1)) {
                    if (IsFloat(t0) || IsFloat(t1)) {
                        // compared in the wider floating type present (mkql_builtins_equals.cpp:33-44)
                        int ft = (t0 == T_F64 || t1 == T_F64) ? T_F64 : T_F32;
                        Status st = Cast(l, ft, false, l);
                        if (!st.ok()) return st;
                        st = Cast(r, ft, false, r);
                        if (!st.ok()) return st;
                        cls = CLS_F64;
                    } else if (IsSignedInt(t0) == IsSignedInt(t1)) {
                        cls = IsSignedInt(t0) ? CLS_I : CLS_U;
                    } else {
                        cls = IsSignedInt(t0) ? CLS_IU : CLS_UI;
                    }
                } else if (t0 == t1 && (t0 == T_BOOL || IsTemporal(t0))) {
                    cls = ClassOfType(t0);
                } else if (asBool(l) && asBool(r)) {
                    cls = CLS_U;
                } else {
                    return Status::Unsupported(std::string("YQL compare of ") + TypeName(t0) + " and " + TypeName(t1));
                }
                return Binary(OP_CMP, cls, cmp, l, r, T_U8, nullable, 0, out);
            }
            case EYqlOp::Add: case EYqlOp::Sub: case EYqlOp::Mul: case EYqlOp::Div: case EYqlOp::Mod: {
                int rt = YqlArithType(t0, t1);
                if (rt == T_NONE) return Status::Unsupported(std::string("YQL arithmetic on ") + TypeName(t0) + ", " + TypeName(t1));
                Status st = Cast(l, rt, false, l);
                if (!st.ok()) return st;
                st = Cast(r, rt, false, r);
                if (!st.ok()) return st;
                int vmop = op == EYqlOp::Add ? OP_ADD : op == EYqlOp::Sub ? OP_SUB : op == EYqlOp::Mul ? OP_MUL
                         : op == EYqlOp::Div ? OP_DIV : OP_MOD;
                bool canNull = nullable || ((vmop == OP_DIV || vmop == OP_MOD) && IsInt(rt));
                int b = (vmop == OP_DIV || vmop == OP_MOD) ? (DIALECT_YQL | (rt << 1)) : DIALECT_YQL;
                return Binary(vmop, ClassOfType(rt), b, l, r, rt, canNull, IsInt(rt) ? rt : 0, out);
            }
            case EYqlOp::And: case EYqlOp::Or: case EYqlOp::Xor: {
                if (!asBool(l) || !asBool(r)) return Status::Unsupported("YQL logic on non-bool");
                int vmop = op == EYqlOp::And ? OP_AND : op == EYqlOp::Or ? OP_OR : OP_XOR;
                return Binary(vmop, CLS_U, DIALECT_YQL, l, r, T_U8, nullable, 0, out);
            }
            case EYqlOp::StartsWith: case EYqlOp::EndsWith: case EYqlOp::StringContains: {
                if (!IsStringLike(t0) || !IsStringLike(t1)) return Status::Unsupported("YQL string op on non-strings");
                if (r->Kind != Expr::CONST) return Status::Unsupported("YQL string op with a column pattern");
                int kind = op == EYqlOp::StartsWith ? STR_STARTS : op == EYqlOp::EndsWith ? STR_ENDS : STR_CONTAINS;
                return StrMatchNode(kind, false, l, r->Scalar.Bytes, r->Scalar.Valid, T_U8, out);
            }
            case EYqlOp::Coalesce: {
                if (t0 != t1 || IsStringLike(t0)) return Status::Unsupported("YQL Coalesce with mixed / string types");
                return Binary(OP_COALESCE, 0, 0, l, r, t0, l->Nullable && r->Nullable, 0, out);
            }
        }
        return Status::Unsupported("unknown YqlOperationId");
    }

    // ---- code generation ---------------------------------------------------------------------
    int ConstIndex(const TScalar& s, Status& st) {
        DevConst c{};
        c.valid = s.Valid;
        c.type = s.Type;
        if (IsStringLike(s.Type)) {
            // reuse an identical pattern
            size_t off = Plan.PatternBytes.find(s.Bytes);
            if (s.Bytes.empty()) off = 0;
            if (off == std::string::npos) {
                off = Plan.PatternBytes.size();
                Plan.PatternBytes += s.Bytes;
            }
            if (Plan.PatternBytes.size() > size_t(kMaxPatternBytes)) {
                st = Status::Unsupported("string constants exceed the pattern buffer");
                return 0;
            }
            c.bits = (uint64_t(off) << 32) | uint64_t(s.Bytes.size());
        } else {
            c.bits = s.Bits;
        }
        for (size_t i = 0; i < Plan.Consts.size(); ++i) {
            if (Plan.Consts[i].bits == c.bits && Plan.Consts[i].valid == c.valid && Plan.Consts[i].type == c.type) return int(i);
        }
        if (Plan.Consts.size() >= size_t(kMaxConsts)) {
            st = Status::Unsupported("too many constants in one step");
            return 0;
        }
        Plan.Consts.push_back(c);
        return int(Plan.Consts.size() - 1);
    }

    int ColSlot(Expr* e, Status& st) {
        if (e->ColSlot >= 0) return e->ColSlot;
        for (size_t i = 0; i < Plan.SourceColumns.size(); ++i) {
            if (Plan.SourceColumns[i] == e->InputIndex) return e->ColSlot = int(i);
        }
        if (Plan.SourceColumns.size() >= size_t(kMaxCols)) {
            st = Status::Unsupported("too many source columns in one step");
            return 0;
        }
        Plan.SourceColumns.push_back(e->InputIndex);
        if (e->Nullable) Plan.Nullable = true;
        return e->ColSlot = int(Plan.SourceColumns.size() - 1);
    }

    void Emit(uint32_t insn, Status& st) {
        if (Plan.Code.size() >= size_t(kMaxCode) - 1) {  // one lane of a wave per instruction
            st = Status::Unsupported("step too long for one scan program");
            return;
        }
        Plan.Code.push_back(insn);
    }

    int AllocTemp(Status& st) {
        int r = NumOuts + NextTemp++;
        if (r >= kMaxRegs) {
            st = Status::Unsupported("expression needs too many registers");
            return 0;
        }
        Plan.NumRegsUsed = std::max(Plan.NumRegsUsed, r + 1);
        return r;
    }

    static Expr* Strip(Expr* e) {
        while (e->Kind == Expr::ALIAS) e = e->L.get();
        return e;
    }

    void CountUses(Expr* e) {
        e = Strip(e);
        if (++e->Uses > 1) return;
        if (e->Kind == Expr::UNARY) CountUses(e->L.get());
        if (e->Kind == Expr::BINARY) {
            CountUses(e->L.get());
            CountUses(e->R.get());
        }
        if (e->Kind == Expr::SELECT) {
            CountUses(e->L.get());
            CountUses(e->R.get());
            CountUses(e->C.get());
        }
    }

    // Operand that an instruction can take directly as its B source.
    bool DirectSource(Expr* e, int& sk, int& si, Status& st) {
        e = Strip(e);
        if (e->Kind == Expr::COL && !IsStringLike(e->Type)) {
            sk = SRC_COL;
            si = ColSlot(e, st);
            return true;
        }
        if (e->Kind == Expr::CONST) {
            sk = SRC_CONST;
            si = ConstIndex(e->Scalar, st);
            if (!e->Scalar.Valid) Plan.Nullable = true;
            return true;
        }
        if (e->Done && e->Reg >= 0) {
            sk = SRC_REG;
            si = e->Reg;
            return true;
        }
        return false;
    }

    static int ReverseOp(int op, int& a, int& b) {
        switch (op) {
            case OP_ADD: case OP_MUL: case OP_AND: case OP_OR: case OP_XOR: return op;
            case OP_SUB: return OP_RSUB;
            case OP_DIV: return OP_RDIV;
            case OP_MOD: return OP_RMOD;
            case OP_FUNC: return OP_RFUNC;
            case OP_CMP:
                b = b == CMP_LT ? CMP_GT : b == CMP_GT ? CMP_LT : b == CMP_LE ? CMP_GE : b == CMP_GE ? CMP_LE : b;
                a = a == CLS_IU ? CLS_UI : a == CLS_UI ? CLS_IU : a;
                return OP_CMP;
            default: return -1;
        }
    }

    // Can evaluating `e` end the program with an error (a failing implicit cast, an Arrow division by zero, a custom
    // kernel's error)?
    static bool MayFail(Expr* e) {
        e = Strip(e);
        if (e->IsLeaf() || e->Kind == Expr::STROP) return false;
        if (e->Kind == Expr::UNARY && e->Op == OP_CHECK) return true;
        if (e->Kind == Expr::BINARY && (e->Op == OP_DIV || e->Op == OP_MOD) && (e->B & 1) == DIALECT_ARROW && (e->A == CLS_I || e->A == CLS_U)) return true;
        if ((e->Kind == Expr::BINARY || e->Kind == Expr::UNARY) && e->Op == OP_FUNC &&
            (e->B == FN_LCM || e->B == FN_MOD || IsNotNullFunc(e->B))) return true;
        if (e->L && MayFail(e->L.get())) return true;
        if (e->R && MayFail(e->R.get())) return true;
        if (e->C && MayFail(e->C.get())) return true;
        return false;
    }

    // Leaves the value of `e` in the accumulator.
    void Gen(Expr* e, Status& st) {
        if (!st.ok()) return;
        e = Strip(e);
        int sk, si;
        if (e->Kind == Expr::COL && IsStringLike(e->Type)) {
            Emit(MakeInsn(OP_LD, SRC_COL, ColSlot(e, st)), st);  // payload 0, validity of the string column
            return;
        }
        if (DirectSource(e, sk, si, st)) {
            Emit(MakeInsn(OP_LD, sk, si), st);
            return;
        }
        if (e->Nullable) Plan.Nullable = true;
        switch (e->Kind) {
            case Expr::UNARY:
                Gen(e->L.get(), st);
                Emit(MakeInsn(e->Op, 0, 0, e->A, e->B), st);
                break;
            case Expr::BINARY: {
                Expr* l = Strip(e->L.get());
                Expr* r = Strip(e->R.get());
                int a = e->A, b = e->B;
                if (DirectSource(r, sk, si, st)) {
                    Gen(l, st);
                    Emit(MakeInsn(e->Op, sk, si, a, b), st);
                } else if (int rop; DirectSource(l, sk, si, st) && (rop = ReverseOp(e->Op, a, b)) >= 0) {
                    Gen(r, st);
                    Emit(MakeInsn(rop, sk, si, a, b), st);
                } else {
                    Gen(r, st);
                    int t = r->Reg;
                    if (!(r->Done && t >= 0)) {
                        t = AllocTemp(st);
                        Emit(MakeInsn(OP_ST, 0, t), st);
                    }
                    Gen(l, st);
                    Emit(MakeInsn(e->Op, SRC_REG, t, e->A, e->B), st);
                }
                break;
            }
            case Expr::SELECT: {
                // else into a register, then as the instruction's operand, the predicate in the accumulator
                Expr* c = Strip(e->C.get());
                Gen(c, st);
                int t = c->Reg;
                if (!(c->Done && t >= 0)) {
                    t = AllocTemp(st);
                    Emit(MakeInsn(OP_ST, 0, t), st);
                }
                Expr* th = Strip(e->R.get());
                if (!DirectSource(th, sk, si, st)) {
                    Gen(th, st);
                    sk = SRC_REG;
                    si = th->Reg;
                    if (!(th->Done && si >= 0)) {
                        si = AllocTemp(st);
                        Emit(MakeInsn(OP_ST, 0, si), st);
                    }
                }
                Gen(Strip(e->L.get()), st);
                Emit(MakeInsn(OP_SELECT, sk, si, t, 0), st);
                break;
            }
            case Expr::STROP: {
                Expr* col = Strip(e->L.get());
                int slot = ColSlot(col, st);
                if (e->Op == OP_STRLEN) {
                    Emit(MakeInsn(OP_STRLEN, SRC_NONE, slot), st);
                } else if (e->Op == OP_DECPART) {
                    Emit(MakeInsn(OP_DECPART, SRC_NONE, slot, e->A), st);
                } else if (e->Op == OP_STRMATCH) {
                    int k = ConstIndex(e->R->Scalar, st);
                    if (!e->R->Scalar.Valid) Plan.Nullable = true;
                    Emit(MakeInsn(OP_STRMATCH, SRC_NONE, slot, k, e->B), st);
                } else {
                    Expr* rhs = Strip(e->R.get());
                    if (rhs->Kind == Expr::CONST) {
                        int k = ConstIndex(rhs->Scalar, st);
                        if (!rhs->Scalar.Valid) Plan.Nullable = true;
                        Emit(MakeInsn(OP_STRCMP, SRC_NONE, slot, k, e->B), st);
                    } else {
                        int slot2 = ColSlot(rhs, st);
                        Emit(MakeInsn(OP_STRCMP, SRC_NONE, slot, slot2, e->B | 16), st);
                    }
                }
                break;
            }
            default:
                st = Status::Invalid("internal: unexpected expression kind");
                return;
        }
        if (e->Uses > 1 && e->Reg < 0) {  // shared sub-expression: keep it in a temporary
            e->Reg = AllocTemp(st);
            Emit(MakeInsn(OP_ST, 0, e->Reg), st);
        } else if (e->Reg >= 0) {
            Emit(MakeInsn(OP_ST, 0, e->Reg), st);
        }
        e->Done = true;
    }

    // Filter of the form <column> cmp <constant> that nothing else reads: one fused instruction.
    bool TryPredCmp(Expr* f, Status& st) {
        f = Strip(f);
        if (f->Kind != Expr::BINARY || f->Op != OP_CMP || f->Uses != 1 || f->Reg >= 0 || f->Done) return false;
        Expr* l = Strip(f->L.get());
        Expr* r = Strip(f->R.get());
        int cls = f->A, cmp = f->B;
        if (l->Kind == Expr::CONST && r->Kind == Expr::COL) {
            std::swap(l, r);
            int op = ReverseOp(OP_CMP, cls, cmp);
            (void)op;
        }
        if (l->Kind != Expr::COL || IsStringLike(l->Type) || r->Kind != Expr::CONST) return false;
        const int slot = ColSlot(l, st);
        const int k = ConstIndex(r->Scalar, st);
        Emit(MakeInsn(OP_PREDCMP, SRC_COL, slot, k, cmp | (cls << 4)), st);
        f->Done = true;
        return true;
    }

    // Evaluate `e` and make sure its value sits in output register `reg`.
    void GenToReg(Expr* e, int reg, Status& st) {
        Expr* s = Strip(e);
        if (s->Done && s->Reg == reg) return;
        if (!s->IsLeaf() && !s->Done && (s->Reg < 0 || s->Reg == reg)) {
            s->Reg = reg;
            Gen(s, st);
            return;
        }
        Gen(s, st);
        if (s->Reg != reg) Emit(MakeInsn(OP_ST, 0, reg), st);
    }

    // ---- the step ----------------------------------------------------------------------------
    Status Run(ECompileMode mode) {
        const bool filterOnly = (mode == CM_FILTER_ONLY);
        ColExprs.assign(Input.size(), nullptr);
        for (size_t i = 0; i < Input.size(); ++i) Env.emplace_back(Input[i].Name, InputColumn(int(i)));

        // ApplyAssignes
        for (auto& a : Step.Assignes) {
            for (auto& kv : Env) {
                if (kv.first == a.GetName()) return Status::Invalid("Assign to existing column '" + a.GetName() + "'.");
            }
            ExprPtr res;
            if (a.IsConstant()) {
                res = MakeConst(*a.Constant);
            } else {
                std::vector<ExprPtr> args;
                for (auto& arg : a.Arguments) {
                    ExprPtr e = Lookup(arg.ColumnName);
                    if (!e) return Status::Invalid("Not found column '" + arg.ColumnName + "' or duplicate");
                    args.push_back(e);
                }
                Status st;
                if (a.DictTable >= 0) {
                    if (args.size() != 1 || !IsInt(args[0]->Type)) return Status::Invalid("internal: dictionary lookup over a column that is not integer codes");
                    st = Unary(OP_LOOKUP, a.DictTable, 0, args[0], a.DictResultType, a.DictResultNullable, res);
                } else {
                    st = a.IsYqlKernel ? YqlFunction(a, args, res) : SimpleFunction(a, args, res);
                }
                if (!st.ok()) return st;
            }
            Env.emplace_back(a.GetName(), res);
        }

        // MakeCombinedFilter: Boolean / Int8 / UInt8 arrays only.
        std::vector<ExprPtr> filters;
        for (auto& f : Step.Filters) {
            ExprPtr e = Lookup(f.ColumnName);
            if (!e) return Status::Invalid("Not found column '" + f.ColumnName + "' or duplicate");
            if (e->Kind == Expr::CONST && !IsArrayShaped(e)) return Status::Invalid("Filter predicate '" + f.ColumnName + "' is a scalar");
            if (e->Type != T_BOOL && e->Type != T_I8 && e->Type != T_U8)
                return Status::Invalid(std::string("Filter predicate '") + f.ColumnName + "' has type " + TypeName(e->Type));
            filters.push_back(e);
        }
        Plan.HasFilter = !filters.empty();

        struct RegOut {
            ExprPtr E;
            int Reg;
        };
        std::vector<RegOut> regOuts;
        Status st;
        auto addRegOut = [&](ExprPtr e) {
            int r = NumOuts++;
            if (r >= kMaxRegs) st = Status::Unsupported("too many computed outputs in one step");
            Plan.NumRegsUsed = std::max(Plan.NumRegsUsed, r + 1);
            regOuts.push_back({e, r});
            return r;
        };

        if (filterOnly) {
            Plan.Kind = StepPlan::FILTER_ONLY;
        } else if (!Step.GroupBy.empty()) {
            Plan.Kind = Step.GroupByKeys.empty() ? StepPlan::AGG_NOKEY : StepPlan::AGG_KEYED;
            if (Step.GroupBy.size() > size_t(kMaxAggs)) return Status::Unsupported("too many aggregates in one step");
            if (Step.GroupByKeys.size() > size_t(kMaxKeys)) return Status::Unsupported("too many GROUP BY keys");
            // keys first: method selection depends on them
            bool hasNullableKey = false, allFixed = true;
            int keyBytes = 0;
            std::vector<ExprPtr> keyExprs;
            for (auto& k : Step.GroupByKeys) {
                ExprPtr e = Lookup(k.ColumnName);
                if (!e) return Status::Invalid("No such key for GROUP BY.");
                if (e->Kind == Expr::CONST) return Status::Invalid("Unexpected GROUP BY key type.");
                if (e->Type == T_BOOL) return Status::Invalid("serializeValueIntoArena unexpected type bool");
                if (e->Type == T_DEC128) return Status::Unsupported("decimal GROUP BY key");
                hasNullableKey |= e->Nullable;
                if (IsStringLike(e->Type)) allFixed = false; else keyBytes += TypeWidth(e->Type);
                keyExprs.push_back(e);
            }
            // Aggregator::chooseAggregationMethod (ydb/library/arrow_clickhouse/Aggregator.cpp:132-226): null keys form
            // their own groups only in the nullable_keys{64,128,256} methods; the `serialized` fallback reads the raw
            // value under a null slot.
            const bool nullAware = hasNullableKey && allFixed && (keyBytes <= 28) && !SerializedKeys;
            // key8 lookup-table path does not look at argument validity (IAggregateFunction.h:258-296,384-459).
            const bool key8Path = !hasNullableKey && keyExprs.size() == 1 && TypeWidth(keyExprs[0]->Type) == 1 && !SerializedKeys;

            std::vector<int> decimalSums, nanTagged, zeroTagged;
            for (size_t i = 0; i < Step.GroupBy.size(); ++i) {
                const TAggregateAssign& g = Step.GroupBy[i];
                AggPlan ap{};
                ap.Spec.reg = -1;
                ExprPtr arg;
                if (!g.Arguments.empty()) {
                    arg = Lookup(g.Arguments[0].ColumnName);
                    if (!arg) {
                        if (g.Operation != EAggregate::Count) {
                            return Status::Invalid("Not found column '" + g.Arguments[0].ColumnName + "' or duplicate");
                        }
                        // COUNT(*) rewritten to COUNT(<default column>) while the batch does not carry that
                        // column: count rows.
                    } else if (arg->Kind == Expr::CONST) {
                        return Status::Unsupported("aggregate over a scalar");
                    }
                }
                const int at = arg ? arg->Type : T_NONE;
                ap.ArgType = at;
                ap.ArgNullable = arg ? (arg->Nullable && !(key8Path)) : false;
                if (arg && Strip(arg.get())->Kind == Expr::COL) ap.ArgInputColumn = Strip(arg.get())->InputIndex;
                switch (g.Operation) {
                    case EAggregate::Count:
                        ap.Spec.kind = AGG_COUNT;
                        ap.OutType = T_U64;
                        break;
                    case EAggregate::Sum:
                        if (IsSignedInt(at)) { ap.Spec.kind = AGG_SUM_I; ap.OutType = T_I64; }
                        else if (IsUnsignedInt(at)) { ap.Spec.kind = AGG_SUM_I; ap.OutType = T_U64; }
                        else if (IsFloat(at)) { ap.Spec.kind = AGG_SUM_F; ap.OutType = T_F64; }
                        else if (at == T_DEC128) { ap.Spec.kind = AGG_SUM_I; ap.OutType = T_DEC128; }
                        else return Status::Invalid(std::string("Function 'sum' has no kernel matching input types (") + TypeName(at) + ")");
                        break;
                    case EAggregate::Min: case EAggregate::Max: {
                        const bool mn = g.Operation == EAggregate::Min;
                        if (IsSignedInt(at) || IsTemporal(at)) ap.Spec.kind = mn ? AGG_MIN_I : AGG_MAX_I;
                        else if (IsUnsignedInt(at)) ap.Spec.kind = mn ? AGG_MIN_U : AGG_MAX_U;
                        else if (IsFloat(at)) ap.Spec.kind = mn ? AGG_MIN_F : AGG_MAX_F;
                        else if (IsStringLike(at)) ap.Spec.kind = mn ? AGG_MIN_S : AGG_MAX_S;
                        else return Status::Unsupported(std::string("min/max over ") + TypeName(at));
                        ap.OutType = at;
                        break;
                    }
                    case EAggregate::Some:
                        if (!(IsNumeric(at) || IsTemporal(at) || IsStringLike(at))) return Status::Unsupported(std::string("some() over ") + TypeName(at));
                        ap.Spec.kind = AGG_SOME;
                        ap.OutType = at;
                        break;
                    default:
                        return Status::Invalid("Unexpected agregate function in GROUP BY");
                }
                ap.Spec.arg_class = arg ? ClassOfType(at) : CLS_U;
                ap.Spec.check_valid = ap.ArgNullable ? 1 : 0;
                if (ap.Spec.kind == AGG_SOME || ap.Spec.kind == AGG_MIN_S || ap.Spec.kind == AGG_MAX_S) {
                    // first non-null in row order == the value at the smallest row index whose argument is
                    // valid: aggregate MIN(row index) here, the executor gathers the value afterwards.  The smallest /
                    // largest string likewise: the sink keeps the row index of the winner (SingleValueDataString,
                    // AggregateFunctionMinMaxAny.h:160-341: the first value, then bytewise `<` / `>`).
                    if (ap.ArgInputColumn < 0 || Input[size_t(ap.ArgInputColumn)].Type != at) {
                        // (a widening cast of a column is still that column to the VM, but not to the gather)
                        return Status::Unsupported("some() / string min / max over a computed expression");
                    }
                    auto rowIdx = std::make_shared<Expr>();
                    rowIdx->Kind = Expr::UNARY;
                    rowIdx->Op = OP_ROWIDX;
                    rowIdx->L = arg;
                    rowIdx->Type = T_U64;
                    rowIdx->Nullable = arg->Nullable;
                    // (string states take the plain row-order loop of the lookup-table path: IAggregateFunction.h:396-402)
                    if (key8Path && !Step.GroupByKeys.empty() && !IsStringLike(at)) {
                        rowIdx->A = ROWIDX_LOOKUP_ORDER;
                        Plan.NeedsFilteredPositions = true;
                    }
                    if (ap.Spec.kind == AGG_SOME) {
                        ap.Spec.kind = AGG_MIN_U;
                        if (IsStringLike(at)) Plan.HasStringMinMax = true;  // (the key-less sink fetches fixed-width values only)
                    } else {
                        ap.Spec.str_col = ColSlot(Strip(arg.get()), st);
                        Plan.HasStringMinMax = true;
                    }
                    ap.Spec.arg_class = CLS_U;
                    ap.SomeInputColumn = ap.ArgInputColumn;
                    ap.Spec.reg = addRegOut(rowIdx);
                } else if (arg && ap.OutType == T_DEC128) {
                    // Extension (the reference has no decimal SUM): wrapping 128-bit sum as three 64-bit sums.
                    Expr* s = Strip(arg.get());
                    if (s->Kind != Expr::COL) return Status::Unsupported("decimal128 SUM over a computed expression");
                    ap.Precision = Input[s->InputIndex].Precision;
                    ap.Scale = Input[s->InputIndex].Scale;
                    ap.Spec.arg_class = CLS_U;
                    ap.Spec.reg = addRegOut(DecimalPart(arg, 0));
                    decimalSums.push_back(int(i));
                } else if (arg) {
                    const bool needValue = ap.Spec.kind != AGG_COUNT;
                    if (needValue || ap.ArgNullable) {
                        Expr* s = Strip(arg.get());
                        if (s->Kind == Expr::COL && !IsStringLike(s->Type)) {
                            // plain input column: the sink reads it from the column file, no register copy
                            ap.Spec.reg = -2 - ColSlot(s, st);
                        } else {
                            // share a register between aggregates over the same expression
                            int reg = -1;
                            for (auto& ro : regOuts) {
                                if (Strip(ro.E.get()) == s) reg = ro.Reg;
                            }
                            if (reg < 0) reg = addRegOut(arg);
                            ap.Spec.reg = reg;
                        }
                    }
                }
                if (!st.ok()) return st;
                if ((ap.Spec.kind == AGG_MIN_F || ap.Spec.kind == AGG_MAX_F) && !Step.GroupByKeys.empty()) {
                    if (key8Path) {
                        // Four interleaved streams, each with the NaN-first rule, merged afterwards: values behind a
                        // stream's leading NaN are lost.  Not reproduced: a NaN argument sends the batch to the CPU path.
                        ap.Spec.flags |= AGG_NAN_UNSUPPORTED;
                        Plan.NeedsFilteredPositions = true;
                    } else {
                        nanTagged.push_back(int(i));
                    }
                    zeroTagged.push_back(int(i));
                }
                Plan.Aggs.push_back(ap);
                OutColumn oc;
                oc.Source = OutColumn::FROM_AGG;
                oc.Name = g.GetName();
                oc.Type = ap.OutType;
                oc.Index = int(i);
                oc.Nullable = ap.Spec.kind != AGG_COUNT;
                oc.Precision = ap.Precision;
                oc.Scale = ap.Scale;
                Plan.Outs.push_back(oc);
            }
            for (int user : decimalSums) {
                for (int part = 1; part <= 2; ++part) {
                    AggPlan hidden = Plan.Aggs[size_t(user)];
                    hidden.Hidden = true;
                    hidden.OutType = T_U64;
                    hidden.DecMid = hidden.DecHi = -1;
                    ExprPtr arg = Lookup(Step.GroupBy[size_t(user)].Arguments[0].ColumnName);
                    hidden.Spec.reg = addRegOut(DecimalPart(arg, part));
                    (part == 1 ? Plan.Aggs[size_t(user)].DecMid : Plan.Aggs[size_t(user)].DecHi) = int(Plan.Aggs.size());
                    Plan.Aggs.push_back(hidden);
                }
            }
            for (int pass = 0; pass < 2; ++pass) {
              std::vector<std::pair<Expr*, int>> made;  // MIN and MAX of one argument share their tags
              for (int user : (pass == 0 ? nanTagged : zeroTagged)) {
                Expr* of = Strip(Lookup(Step.GroupBy[size_t(user)].Arguments[0].ColumnName).get());
                int shared = -1;
                for (auto& m : made) {
                    if (m.first == of) shared = m.second;
                }
                if (shared >= 0) {
                    (pass == 0 ? Plan.Aggs[size_t(user)].NanTag : Plan.Aggs[size_t(user)].ZeroTag) = shared;
                    continue;
                }
                made.emplace_back(of, int(Plan.Aggs.size()));
                AggPlan hidden{};
                hidden.Hidden = true;
                hidden.OutType = T_U64;
                hidden.ArgType = T_U64;
                hidden.ArgNullable = Plan.Aggs[size_t(user)].ArgNullable;
                hidden.Spec.kind = AGG_MIN_U;
                hidden.Spec.arg_class = CLS_U;
                hidden.Spec.check_valid = hidden.ArgNullable ? 1 : 0;
                ExprPtr arg = Lookup(Step.GroupBy[size_t(user)].Arguments[0].ColumnName);
                auto tag = std::make_shared<Expr>();
                tag->Kind = Expr::UNARY;
                tag->Op = OP_ROWIDX;
                tag->A = pass == 0 ? ROWIDX_NAN_TAG : (ROWIDX_ZERO_TAG | (key8Path ? ROWIDX_LOOKUP_ORDER : 0));
                tag->L = arg;
                tag->Type = T_U64;
                tag->Nullable = arg->Nullable;
                hidden.Spec.reg = addRegOut(tag);
                (pass == 0 ? Plan.Aggs[size_t(user)].NanTag : Plan.Aggs[size_t(user)].ZeroTag) = int(Plan.Aggs.size());
                Plan.Aggs.push_back(hidden);
              }
            }
            if (Plan.Aggs.size() > size_t(kMaxAggs)) return Status::Unsupported("too many aggregates in one step");
            if (!st.ok()) return st;
            for (size_t i = 0; i < keyExprs.size(); ++i) {
                ExprPtr e = keyExprs[i];
                KeyPlan kp{};
                kp.Spec.type = e->Type;
                kp.Spec.nullable = (nullAware && e->Nullable) ? 1 : 0;
                kp.Spec.reg = -1;
                kp.Spec.col = -1;
                Expr* s = Strip(e.get());
                if (s->Kind == Expr::COL) kp.InputColumn = s->InputIndex;
                if (IsStringLike(e->Type)) {
                    if (s->Kind != Expr::COL) return Status::Unsupported("computed string GROUP BY key");
                    kp.Spec.col = ColSlot(s, st);
                } else if (s->Kind == Expr::COL) {
                    kp.Spec.reg = -2 - ColSlot(s, st);
                } else {
                    int reg = -1;
                    for (auto& ro : regOuts) {
                        if (Strip(ro.E.get()) == s) reg = ro.Reg;
                    }
                    if (reg < 0) reg = addRegOut(e);
                    kp.Spec.reg = reg;
                }
                if (!st.ok()) return st;
                Plan.Keys.push_back(kp);
                OutColumn oc;
                oc.Source = OutColumn::FROM_KEY;
                oc.Name = Step.GroupByKeys[i].ColumnName;
                oc.Type = e->Type;
                oc.Index = int(i);
                oc.Nullable = kp.Spec.nullable;
                Plan.Outs.push_back(oc);
            }
            // ApplyProjection after the aggregation selects among the aggregate/key columns.
            if (!Step.Projection.empty()) {
                std::vector<OutColumn> projected;
                for (auto& p : Step.Projection) {
                    int found = -1, count = 0;
                    for (size_t i = 0; i < Plan.Outs.size(); ++i) {
                        if (Plan.Outs[i].Name == p.ColumnName) {
                            found = int(i);
                            ++count;
                        }
                    }
                    if (count != 1) return Status::Invalid("Could not find column " + p.ColumnName + " in record batch schema.");
                    projected.push_back(Plan.Outs[found]);
                }
                Plan.Outs = projected;
            }
        } else {
            // Projection (or every column of the working batch when there is none).
            std::vector<std::pair<std::string, ExprPtr>> outs;
            if (Step.Projection.empty()) {
                outs = Env;
            } else {
                for (auto& p : Step.Projection) {
                    ExprPtr e = Lookup(p.ColumnName);
                    if (!e) return Status::Invalid("Could not find column " + p.ColumnName + " in record batch schema.");
                    outs.emplace_back(p.ColumnName, e);
                }
            }
            bool anyComputed = false;
            for (auto& kv : outs) {
                Expr* s = Strip(kv.second.get());
                OutColumn oc;
                oc.Name = kv.first;
                oc.Type = kv.second->Type;
                oc.Nullable = kv.second->Nullable;
                if (s->Kind == Expr::COL && s->Type == kv.second->Type) {
                    oc.Source = OutColumn::FROM_INPUT;
                    oc.Index = s->InputIndex;
                    oc.Precision = Input[s->InputIndex].Precision;
                    oc.Scale = Input[s->InputIndex].Scale;
                } else if (s->Kind == Expr::CONST && !IsArrayShaped(kv.second)) {
                    oc.Source = OutColumn::FROM_SCALAR;
                    oc.Scalar = s->Scalar;
                    oc.Scalar.Type = kv.second->Type;
                } else {
                    if (IsStringLike(oc.Type)) return Status::Unsupported("computed string output");
                    oc.Source = OutColumn::FROM_REG;
                    int reg = -1;
                    for (auto& ro : regOuts) {
                        if (Strip(ro.E.get()) == s) reg = ro.Reg;
                    }
                    if (reg < 0) reg = addRegOut(kv.second);
                    oc.Index = reg;
                    anyComputed = true;
                }
                if (!st.ok()) return st;
                Plan.Outs.push_back(oc);
            }
            Plan.Kind = (anyComputed || Plan.HasFilter) ? StepPlan::PROJECT : StepPlan::PASS_THROUGH;
        }

        // pin computed outputs to their registers before any code is emitted
        for (auto& ro : regOuts) {
            Expr* s = Strip(ro.E.get());
            if (!s->IsLeaf() && s->Reg < 0) s->Reg = ro.Reg;
        }
        // use counts over everything that will be evaluated
        if (mode != CM_OUTPUTS_ONLY) {
            for (auto& f : filters) CountUses(f.get());
        }
        for (auto& ro : regOuts) CountUses(ro.E.get());
        // The reference evaluates every Assign, used or not (TProgramStep::ApplyAssignes, program.cpp:869-878), so an
        // assign nothing reads can still fail the program: those that can are evaluated for their errors, in the one
        // pass of the step that sees every row.
        std::vector<Expr*> unread;
        if (mode != CM_OUTPUTS_ONLY || filters.empty()) {
            for (auto& kv : Env) {
                Expr* s = Strip(kv.second.get());
                if (!s->IsLeaf() && s->Uses == 0 && MayFail(s)) unread.push_back(s);
            }
            for (Expr* s : unread) CountUses(s);
            Plan.UnreadFallible = !unread.empty();
        }
        // code: filters first, then outputs
        if (mode != CM_OUTPUTS_ONLY) {
            for (auto& f : filters) {
                if (TryPredCmp(f.get(), st)) continue;
                Gen(f.get(), st);
                Emit(MakeInsn(OP_PRED), st);
            }
        }
        for (auto& ro : regOuts) {
            if (ro.E->Nullable) Plan.Nullable = true;
            GenToReg(ro.E.get(), ro.Reg, st);
        }
        for (Expr* s : unread) {
            if (!s->Done) Gen(s, st);
        }
        if (!st.ok()) return st;
        Plan.Code.push_back(MakeInsn(OP_END));
        return Status::OK();
    }
};

}  // namespace

std::string StepPlan::DebugString() const {
    static const char* kinds[] = {"PASS_THROUGH", "PROJECT", "AGG_NOKEY", "AGG_KEYED", "FILTER_ONLY"};
    std::string s = std::string("{kind=") + kinds[Kind] + ";cols=[";
    for (int c : SourceColumns) s += std::to_string(c) + ",";
    s += "];code=[";
    for (uint32_t i : Code) {
        s += std::to_string(InsnOp(i)) + ":" + std::to_string(InsnSk(i)) + ":" + std::to_string(InsnSi(i)) + ":" +
             std::to_string(InsnA(i)) + ":" + std::to_string(InsnB(i)) + " ";
    }
    s += "];nullable=" + std::to_string(Nullable) + ";outs=[";
    for (auto& o : Outs) s += o.Name + ":" + TypeName(o.Type) + "/" + std::to_string(int(o.Source)) + ",";
    return s + "]}";
}

Status CompileStep(const TProgramStep& step, const std::vector<FieldDesc>& input, ECompileMode mode, StepPlan& out, bool serializedKeys) {
    out = StepPlan();
    Compiler c(step, input, out);
    c.SerializedKeys = serializedKeys;
    return c.Run(mode);
}

}  // namespace ssa
