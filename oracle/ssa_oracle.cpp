// TEST INFRASTRUCTURE -- not part of the product.
//
// CPU restatement of the reference's SSA program execution (NKikimrSSA::TProgram over Arrow record
// batches), written op-at-a-time the way the reference runs it: every Assign materialises a full
// column, Filters build one boolean mask that compacts the needed columns, aggregates run after that,
// Projection last (ydb/core/formats/arrow/program.cpp:869-902).  Each function cites the reference
// code it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
// library; the GPU product never links or calls it.
//
// Pinning: checked against the known-answer vectors of the reference's own unit tests
// (ut_program_step.cpp, ut_program.cpp, ut_aggregator.cpp, aggregations_ut.cpp; transcribed under
// tests/golden/).  decimal128 SUM has no counterpart in the reference: parity unpinned for that one
// aggregate (defined here as the wrapping two's-complement 128-bit sum).
//
// Storage is deliberately simple: one widened vector per column (int64 / uint64 / double / string).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

enum Ty : int { NONE = 0, BOOL, I8, I16, I32, I64, U8, U16, U32, U64, F32, F64, TS, DUR, STR, BIN, DEC };

static const char* TyName(int t) {
    static const char* n[] = {"none", "bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64",
                              "float", "double", "timestamp", "duration", "string", "binary", "decimal128"};
    return n[t];
}
static bool IsSigned(int t) { return t >= I8 && t <= I64; }
static bool IsUnsigned(int t) { return t >= U8 && t <= U64; }
static bool IsInt(int t) { return IsSigned(t) || IsUnsigned(t); }
static bool IsFloat(int t) { return t == F32 || t == F64; }
static bool IsNum(int t) { return IsInt(t) || IsFloat(t); }
static bool IsTemporal(int t) { return t == TS || t == DUR; }
static bool IsStr(int t) { return t == STR || t == BIN; }
static int Bits(int t) {
    switch (t) {
        case BOOL: case I8: case U8: return 8;
        case I16: case U16: return 16;
        case I32: case U32: case F32: return 32;
        case DEC: return 128;
        default: return 64;
    }
}
// which widened vector holds the values
enum Store { S_I, S_U, S_F, S_S };
static Store StoreOf(int t) {
    if (IsSigned(t) || IsTemporal(t)) return S_I;
    if (IsUnsigned(t) || t == BOOL) return S_U;
    if (IsFloat(t)) return S_F;
    return S_S;
}

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// arrow::Datum: an array, or a scalar that broadcasts.
struct Datum {
    int ty = NONE;
    bool scalar = false;
    int64_t n = 0;
    std::vector<int64_t> i;
    std::vector<uint64_t> u;
    std::vector<double> f;
    std::vector<std::string> s;
    std::vector<int64_t> hi;   // DEC: high words (low words in u)
    std::vector<uint8_t> ok;   // empty = no nulls
    bool Valid(int64_t r) const { return ok.empty() || ok[scalar ? 0 : r]; }
    bool MayHaveNulls() const {
        for (uint8_t v : ok) if (!v) return true;
        return false;
    }
    int64_t I(int64_t r) const { return i[scalar ? 0 : r]; }
    uint64_t U(int64_t r) const { return u[scalar ? 0 : r]; }
    double F(int64_t r) const { return f[scalar ? 0 : r]; }
    const std::string& S(int64_t r) const { return s[scalar ? 0 : r]; }
};
using DatumPtr = std::shared_ptr<Datum>;

static DatumPtr NewArray(int ty, int64_t n) {
    auto d = std::make_shared<Datum>();
    d->ty = ty;
    d->n = n;
    switch (StoreOf(ty)) {
        case S_I: d->i.assign(n, 0); break;
        case S_U: d->u.assign(n, 0); break;
        case S_F: d->f.assign(n, 0); break;
        case S_S: d->s.assign(n, std::string()); break;
    }
    if (ty == DEC) { d->u.assign(n, 0); d->hi.assign(n, 0); d->i.clear(); }
    return d;
}

// ---------------------------------------------------------------------------------------------------
// protobuf wire reader (ydb/core/formats/arrow/protos/ssa.proto:19-209)
// ---------------------------------------------------------------------------------------------------
struct Wire {
    const uint8_t* p;
    const uint8_t* e;
    Wire(const uint8_t* b, size_t n) : p(b), e(b + n) {}
    bool More() const { return p < e; }
    uint64_t Varint() {
        uint64_t v = 0;
        for (int sh = 0; sh < 70; sh += 7) {
            if (p >= e) throw Error("truncated varint");
            uint8_t b = *p++;
            v |= uint64_t(b & 0x7F) << sh;
            if (!(b & 0x80)) return v;
        }
        throw Error("bad varint");
    }
    // returns field number, fills wire type
    uint32_t Tag(int& wt) {
        uint64_t t = Varint();
        wt = int(t & 7);
        return uint32_t(t >> 3);
    }
    Wire Sub() {
        uint64_t n = Varint();
        if (uint64_t(e - p) < n) throw Error("truncated field");
        Wire w(p, n);
        p += n;
        return w;
    }
    std::string Str() {
        Wire w = Sub();
        return std::string(reinterpret_cast<const char*>(w.p), size_t(w.e - w.p));
    }
    uint32_t Fixed32() { uint32_t v; if (e - p < 4) throw Error("truncated"); std::memcpy(&v, p, 4); p += 4; return v; }
    uint64_t Fixed64() { uint64_t v; if (e - p < 8) throw Error("truncated"); std::memcpy(&v, p, 8); p += 8; return v; }
    void Skip(int wt) {
        switch (wt) {
            case 0: Varint(); break;
            case 1: Fixed64(); break;
            case 2: Sub(); break;
            case 5: Fixed32(); break;
            default: throw Error("unsupported wire type");
        }
    }
};

struct PColumn { bool hasId = false; uint64_t id = 0; bool hasName = false; std::string name; };
static PColumn ReadColumn(Wire w) {
    PColumn c;
    while (w.More()) {
        int wt; uint32_t f = w.Tag(wt);
        if (f == 1 && wt == 0) { c.hasId = true; c.id = w.Varint(); }
        else if (f == 2 && wt == 2) { c.hasName = true; c.name = w.Str(); }
        else w.Skip(wt);
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------
// program model (ydb/core/formats/arrow/program.h:60-447)
// ---------------------------------------------------------------------------------------------------
enum Op {
    OP_NONE, OP_CONST, OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_ISNULL, OP_STRLEN, OP_MATCH, OP_MATCH_LIKE,
    OP_STARTS, OP_ENDS, OP_NOT, OP_AND, OP_OR, OP_XOR, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_CAST, OP_YQL,
    // operations without a protobuf id (TAssign objects built in code)
    OP_ISVALID, OP_ABS, OP_NEGATE, OP_GCD, OP_LCM, OP_MODULO, OP_MODZ, OP_ROUND, OP_ROUNDBANKERS, OP_ROUNDEXP2,
    OP_FLOOR, OP_CEIL, OP_TRUNC, OP_SQRT, OP_E, OP_PI,
    OP_ACOSH, OP_ATANH, OP_CBRT, OP_COSH, OP_ERF, OP_ERFC, OP_EXP, OP_EXP2, OP_LGAMMA, OP_SINH, OP_TGAMMA, OP_HYPOT,
    OP_ADD_CHECKED, OP_SUB_CHECKED, OP_MUL_CHECKED, OP_DIV_CHECKED
};
enum Agg { AG_NONE = 0, AG_SOME = 1, AG_COUNT = 2, AG_MIN = 3, AG_MAX = 4, AG_SUM = 5 };

struct Assign {
    std::string name;
    Op op = OP_NONE;
    std::vector<std::string> args;
    DatumPtr constant;
    int castTo = NONE;
    bool hasPattern = false, needsPattern = false;
    std::string pattern;
    bool ignoreCase = false;
    bool yqlHasOp = false;
    uint32_t yqlOp = 0;
};
struct AggAssign { std::string name; Agg fn = AG_NONE; std::vector<std::string> args; };
struct Step {
    std::vector<Assign> assigns;
    std::vector<std::string> filters;
    std::vector<AggAssign> groupBy;
    std::vector<std::string> keys;
    std::vector<std::string> projection;
    bool Empty() const { return assigns.empty() && filters.empty() && groupBy.empty() && keys.empty() && projection.empty(); }
};

struct Program {
    std::vector<Step> steps;
    bool directOperations = false;  // TFunction ids of 1000 + NKernels::EOperation are accepted
    std::vector<uint32_t> kernelOps;  // per KernelIdx what the caller says the YQL kernel is (0xFFFFFFFF: nothing)
    std::map<uint32_t, std::string> resolver;       // column id -> table column name
    std::map<std::string, uint32_t> byName;
    std::vector<std::pair<uint32_t, std::string>> sources;  // in first-use order
    std::map<std::string, DatumPtr> constants;      // TProgramBuilder::Constants
    std::map<std::string, DatumPtr> parameters;     // row 0 of the parameters batch, by name
    uint32_t defaultId = 0;                         // IColumnResolver::GetDefaultColumn(): the resolver's first column
    std::string defaultName;

    // TProgramBuilder::GetColumnInfo (ydb/core/tx/program/program.cpp:36-68): a non-zero Id wins and resolves to
    // the table column of that id, else the generated name is the explicit Name or the decimal Id.
    std::string ColumnName(const PColumn& c) {
        if (c.hasId && c.id) {
            auto it = resolver.find(uint32_t(c.id));
            if (it != resolver.end()) { AddSource(it->first, it->second); return it->second; }
            return c.hasName ? c.name : std::to_string(c.id);
        }
        if (c.hasName && !c.name.empty()) {
            auto it = byName.find(c.name);
            if (it != byName.end()) AddSource(it->second, c.name);
            return c.name;
        }
        return c.hasName ? c.name : std::to_string(c.id);
    }
    void AddSource(uint32_t id, const std::string& name) {
        for (auto& s : sources) if (s.first == id) return;
        sources.emplace_back(id, name);
    }
};

static DatumPtr ScalarI(int ty, int64_t v) { auto d = NewArray(ty, 1); d->scalar = true; d->i[0] = v; return d; }
static DatumPtr ScalarU(int ty, uint64_t v) { auto d = NewArray(ty, 1); d->scalar = true; d->u[0] = v; return d; }
static DatumPtr ScalarF(int ty, double v) { auto d = NewArray(ty, 1); d->scalar = true; d->f[0] = v; return d; }
static DatumPtr ScalarS(int ty, const std::string& v) { auto d = NewArray(ty, 1); d->scalar = true; d->s[0] = v; return d; }

// TProgramBuilder::MakeConstant (ydb/core/tx/program/program.cpp:253-295)
static DatumPtr ReadConstant(Wire w) {
    DatumPtr out;
    while (w.More()) {
        int wt; uint32_t f = w.Tag(wt);
        switch (f) {
            case 1: out = ScalarU(BOOL, w.Varint() ? 1 : 0); break;
            case 2: out = ScalarI(I32, int32_t(w.Varint())); break;
            case 3: out = ScalarU(U32, uint32_t(w.Varint())); break;
            case 4: out = ScalarI(I64, int64_t(w.Varint())); break;
            case 5: out = ScalarU(U64, w.Varint()); break;
            case 6: { uint32_t b = w.Fixed32(); float v; std::memcpy(&v, &b, 4); out = ScalarF(F32, v); break; }
            case 7: { uint64_t b = w.Fixed64(); double v; std::memcpy(&v, &b, 8); out = ScalarF(F64, v); break; }
            case 8: out = ScalarS(BIN, w.Str()); break;
            case 9: out = ScalarS(STR, w.Str()); break;
            case 10: out = ScalarI(I8, int8_t(int32_t(w.Varint()))); break;
            case 11: out = ScalarU(U8, uint8_t(w.Varint())); break;
            case 12: out = ScalarI(I16, int16_t(int32_t(w.Varint()))); break;
            case 13: out = ScalarU(U16, uint16_t(w.Varint())); break;
            case 14: out = ScalarI(TS, int64_t(w.Varint())); break;
            default: w.Skip(wt);
        }
    }
    if (!out) throw Error("Unsupported constant");
    return out;
}

// TProgramBuilder::MakeFunction (ydb/core/tx/program/program.cpp:84-251)
static void ReadFunction(Program& prog, Wire w, Assign& a) {
    uint32_t id = 0, ftype = 1, kernelIdx = 0;
    std::vector<PColumn> args;
    while (w.More()) {
        int wt; uint32_t f = w.Tag(wt);
        if (f == 1 && wt == 0) id = uint32_t(w.Varint());
        else if (f == 2 && wt == 2) args.push_back(ReadColumn(w.Sub()));
        else if (f == 3 && wt == 0) ftype = uint32_t(w.Varint());
        else if (f == 4 && wt == 0) kernelIdx = uint32_t(w.Varint());
        else if (f == 5 && wt == 0) { a.yqlHasOp = true; a.yqlOp = uint32_t(w.Varint()); }
        else w.Skip(wt);
    }
    for (auto& c : args) a.args.push_back(prog.ColumnName(c));
    if (ftype == 2) {  // YQL_KERNEL
        a.op = OP_YQL;
        // a kernel sent with its KernelIdx alone (unary kernels, If: kqp_olap_compiler.cpp:655-676, 740-759) is what the
        // caller's table of the kernel set says it is
        if (!a.yqlHasOp && kernelIdx < prog.kernelOps.size() && prog.kernelOps[kernelIdx] != 0xFFFFFFFFu) {
            a.yqlHasOp = true;
            a.yqlOp = prog.kernelOps[kernelIdx];
        }
        return;
    }
    auto like = [&](bool icase) {  // mkLikeOptions: the pattern is a previously assigned binary-like constant
        a.needsPattern = true;
        a.ignoreCase = icase;
        if (a.args.size() != 2) return;
        auto it = prog.constants.find(a.args[1]);
        if (it == prog.constants.end() || !IsStr(it->second->ty)) return;
        a.pattern = it->second->s[0];
        a.hasPattern = true;
        a.args.pop_back();
    };
    static const int castTypes[] = {BOOL, I8, I16, I32, I64, U8, U16, U32, U64, F32, F64};
    if (prog.directOperations && id >= 1000) {
        // TAssign(column, EOperation, arguments): the enumerators of ydb/library/arrow_kernels/operations.h:5-84 by
        // position; ValidateOperation (ydb/core/formats/arrow/program.cpp:303-389) checks the arity of the first group only
        size_t want = 1;
        switch (id - 1000) {
            case 17: a.op = OP_ISVALID; break;
            case 33: a.op = OP_ABS; break;
            case 34: a.op = OP_NEGATE; break;
            case 35: a.op = OP_GCD; want = 2; break;
            case 36: a.op = OP_LCM; want = 2; break;
            case 37: a.op = OP_MODULO; want = 2; break;
            case 38: a.op = OP_MODZ; want = 2; break;
            case 39: a.op = OP_ADD_CHECKED; want = 2; break;
            case 40: a.op = OP_SUB_CHECKED; want = 2; break;
            case 41: a.op = OP_MUL_CHECKED; want = 2; break;
            case 42: a.op = OP_DIV_CHECKED; want = 2; break;
            case 48: a.op = OP_ACOSH; break;
            case 49: a.op = OP_ATANH; break;
            case 50: a.op = OP_CBRT; break;
            case 51: a.op = OP_COSH; break;
            case 53: a.op = OP_ERF; break;
            case 54: a.op = OP_ERFC; break;
            case 55: a.op = OP_EXP; break;
            case 56: a.op = OP_EXP2; break;
            case 57: throw Error("Key error: No function registered with name: exp10");  // custom_registry.cpp:35-38
            case 58: a.op = OP_HYPOT; want = 2; break;
            case 59: a.op = OP_LGAMMA; break;
            case 61: a.op = OP_SINH; break;
            case 63: a.op = OP_TGAMMA; break;
            case 52: a.op = OP_E; want = 0; break;
            case 60: a.op = OP_PI; want = 0; break;
            case 62: a.op = OP_SQRT; break;
            case 64: a.op = OP_FLOOR; break;
            case 65: a.op = OP_CEIL; break;
            case 66: a.op = OP_TRUNC; break;
            case 67: a.op = OP_ROUND; break;
            case 68: a.op = OP_ROUNDBANKERS; break;
            case 69: a.op = OP_ROUNDEXP2; break;
            default: throw Error("Unsupported function");
        }
        const bool checked = (id - 1000) <= 47;
        if (a.args.size() != want) throw Error(checked ? "Wrong arguments count for function" : "Function accepts another number of arguments");
        return;
    }
    switch (id) {
        case 1: a.op = OP_EQ; break;
        case 2: a.op = OP_NE; break;
        case 3: a.op = OP_LT; break;
        case 4: a.op = OP_LE; break;
        case 5: a.op = OP_GT; break;
        case 6: a.op = OP_GE; break;
        case 7: a.op = OP_ISNULL; break;
        case 8: a.op = OP_STRLEN; break;
        case 9: a.op = OP_MATCH; like(false); break;
        case 10: a.op = OP_NOT; break;
        case 11: a.op = OP_AND; break;
        case 12: a.op = OP_OR; break;
        case 13: a.op = OP_XOR; break;
        case 14: a.op = OP_ADD; break;
        case 15: a.op = OP_SUB; break;
        case 16: a.op = OP_MUL; break;
        case 17: a.op = OP_DIV; break;
        case 31: a.op = OP_CAST; a.castTo = TS; break;
        case 32: a.op = OP_MATCH_LIKE; like(false); break;
        case 33: a.op = OP_STARTS; like(false); break;
        case 34: a.op = OP_ENDS; like(false); break;
        case 35: a.op = OP_MATCH; like(true); break;
        case 36: a.op = OP_STARTS; like(true); break;
        case 37: a.op = OP_ENDS; like(true); break;
        default:
            if (id >= 18 && id <= 28) { a.op = OP_CAST; a.castTo = castTypes[id - 18]; break; }
            throw Error("Unsupported function");  // incl. CAST_TO_BINARY / CAST_TO_FIXED_SIZE_BINARY
    }
    // ValidateOperation (ydb/core/formats/arrow/program.cpp:303-389): arity per operation
    size_t want = 2;
    switch (a.op) {
        case OP_ISNULL: case OP_STRLEN: case OP_NOT: case OP_CAST: want = 1; break;
        case OP_MATCH: case OP_MATCH_LIKE: case OP_STARTS: case OP_ENDS: want = 1; break;
        default: break;
    }
    if (a.args.size() != want) throw Error("Wrong arguments count for function");
}

static void ParseProgramBody(Program& prog, Wire w) {
    Step step;
    auto closeStep = [&]() { prog.steps.push_back(step); step = Step(); };
    while (w.More()) {
        int wt; uint32_t f = w.Tag(wt);
        if (!(f == 1 && wt == 2)) { w.Skip(wt); continue; }
        Wire cmd = w.Sub();
        while (cmd.More()) {
            int cwt; uint32_t cf = cmd.Tag(cwt);
            if (cwt != 2) { cmd.Skip(cwt); continue; }
            Wire body = cmd.Sub();
            if (cf == 1) {  // Assign (ExtractAssign, ydb/core/tx/program/program.cpp:358-399)
                Assign a;
                PColumn target;
                bool haveExpr = false;
                Wire b = body;
                // the target column is read first so that its name exists when the expression is parsed
                for (Wire t = body; t.More();) {
                    int twt; uint32_t tf = t.Tag(twt);
                    if (tf == 1 && twt == 2) target = ReadColumn(t.Sub()); else t.Skip(twt);
                }
                a.name = target.hasName ? target.name : std::to_string(target.id);
                while (b.More()) {
                    int bwt; uint32_t bf = b.Tag(bwt);
                    if (bf == 2 && bwt == 2) { ReadFunction(prog, b.Sub(), a); haveExpr = true; }
                    else if (bf == 4 && bwt == 2) {
                        a.op = OP_CONST;
                        a.constant = ReadConstant(b.Sub());
                        prog.constants[a.name] = a.constant;
                        haveExpr = true;
                    } else if (bf == 6 && bwt == 2) {  // Parameter: row 0 of the parameters batch
                        Wire pw = b.Sub();
                        std::string pname;
                        while (pw.More()) { int pwt; uint32_t pf = pw.Tag(pwt); if (pf == 1 && pwt == 2) pname = pw.Str(); else pw.Skip(pwt); }
                        auto it = prog.parameters.find(pname);
                        if (it == prog.parameters.end()) throw Error("Parameter not found: " + pname);
                        a.op = OP_CONST;
                        a.constant = it->second;
                        prog.constants[a.name] = a.constant;
                        haveExpr = true;
                    } else if (bf == 3 || bf == 5) {
                        throw Error("Unsupported assignment (ExternalFunction / Null)");
                    } else b.Skip(bwt);
                }
                if (!haveExpr) throw Error("Empty assignment");
                step.assigns.push_back(a);
            } else if (cf == 2) {  // Projection
                while (body.More()) {
                    int bwt; uint32_t bf = body.Tag(bwt);
                    if (bf == 1 && bwt == 2) step.projection.push_back(prog.ColumnName(ReadColumn(body.Sub()))); else body.Skip(bwt);
                }
                closeStep();
            } else if (cf == 3) {  // Filter
                bool any = false;
                while (body.More()) {
                    int bwt; uint32_t bf = body.Tag(bwt);
                    if (bf == 1 && bwt == 2) {
                        PColumn c = ReadColumn(body.Sub());
                        if (!c.hasId && !c.hasName) throw Error("Filter without predicate");
                        step.filters.push_back(prog.ColumnName(c));
                        any = true;
                    } else body.Skip(bwt);
                }
                if (!any) throw Error("Filter without predicate");
            } else if (cf == 4) {  // GroupBy (ExtractGroupBy, ydb/core/tx/program/program.cpp:421-441)
                size_t naggs = 0;
                while (body.More()) {
                    int bwt; uint32_t bf = body.Tag(bwt);
                    if (bf == 1 && bwt == 2) {
                        Wire aw = body.Sub();
                        AggAssign ag;
                        PColumn target;
                        while (aw.More()) {
                            int awt; uint32_t af = aw.Tag(awt);
                            if (af == 1 && awt == 2) target = ReadColumn(aw.Sub());
                            else if (af == 2 && awt == 2) {
                                Wire fw = aw.Sub();
                                while (fw.More()) {
                                    int fwt; uint32_t ff = fw.Tag(fwt);
                                    if (ff == 1 && fwt == 0) ag.fn = Agg(fw.Varint());
                                    else if (ff == 2 && fwt == 2) ag.args.push_back(prog.ColumnName(ReadColumn(fw.Sub())));
                                    else fw.Skip(fwt);
                                }
                            } else aw.Skip(awt);
                        }
                        ag.name = prog.ColumnName(target);
                        // MakeAggregate (ydb/core/tx/program/program.cpp:297-336): one argument, COUNT may have none
                        if (ag.fn < AG_SOME || ag.fn > AG_SUM) throw Error("Unknown aggregate function");
                        if (ag.args.size() > 1 || (ag.args.empty() && ag.fn != AG_COUNT)) throw Error("Wrong aggregate arguments");
                        step.groupBy.push_back(ag);
                        ++naggs;
                    } else if (bf == 2 && bwt == 2) {
                        step.keys.push_back(prog.ColumnName(ReadColumn(body.Sub())));
                    } else body.Skip(bwt);
                }
                if (!naggs) throw Error("GroupBy without aggregates");
                closeStep();
            }
        }
    }
    if (!step.Empty()) prog.steps.push_back(step);
    // Query 'SELECT count(*) FROM table' needs a column (ydb/core/tx/program/program.cpp:601-605)
    if (prog.sources.empty() && !prog.defaultName.empty()) prog.AddSource(prog.defaultId, prog.defaultName);
    // OptimizeProgram (ydb/core/formats/arrow/ssa_program_optimizer.cpp:7-30): COUNT(*) counts the first GROUP BY
    // key of its step, else any source column of the program (the first one in use order here; the scan's
    // default column when the program reads none).
    for (auto& s : prog.steps) {
        for (auto& g : s.groupBy) {
            if (g.fn == AG_COUNT && g.args.empty()) {
                if (!s.keys.empty()) g.args.push_back(s.keys[0]);
                else if (!prog.sources.empty()) g.args.push_back(prog.sources[0].second);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// scalar kernels
// ---------------------------------------------------------------------------------------------------
// CommonNumeric (contrib/libs/apache/arrow/cpp/src/arrow/compute/kernels/codegen_internal.cc:226-278)
static int CommonNumeric(int a, int b) {
    if (!IsNum(a) || !IsNum(b)) return NONE;
    if (a == F64 || b == F64) return F64;
    if (a == F32 || b == F32) return F32;
    int ms = 0, mu = 0;
    for (int t : {a, b}) { if (IsSigned(t)) ms = std::max(ms, Bits(t)); else mu = std::max(mu, Bits(t)); }
    auto pick = [](bool sgn, int w) {
        if (sgn) return w >= 64 ? I64 : w == 32 ? I32 : w == 16 ? I16 : I8;
        return w >= 64 ? U64 : w == 32 ? U32 : w == 16 ? U16 : U8;
    };
    if (ms == 0) return pick(false, mu);
    if (ms <= mu) { int w = 8; while (w < mu + 1) w <<= 1; ms = w; }
    return pick(true, ms);
}

static int64_t WrapI(int64_t v, int ty) {
    switch (ty) {
        case I8: return int8_t(v);
        case I16: return int16_t(v);
        case I32: return int32_t(v);
        default: return v;
    }
}
static uint64_t WrapU(uint64_t v, int ty) {
    switch (ty) {
        case U8: case BOOL: return uint8_t(v);
        case U16: return uint16_t(v);
        case U32: return uint32_t(v);
        default: return v;
    }
}
static bool IntFits(bool srcSigned, int64_t si, uint64_t ui, int to) {
    if (IsSigned(to) || IsTemporal(to)) {
        int w = IsTemporal(to) ? 64 : Bits(to);
        int64_t lo = w == 64 ? INT64_MIN : -(int64_t(1) << (w - 1)), hi = w == 64 ? INT64_MAX : (int64_t(1) << (w - 1)) - 1;
        if (srcSigned) return si >= lo && si <= hi;
        return ui <= uint64_t(hi);
    }
    uint64_t hi = Bits(to) == 64 ? UINT64_MAX : (uint64_t(1) << Bits(to)) - 1;
    if (srcSigned) return si >= 0 && uint64_t(si) <= hi;
    return ui <= hi;
}

// static_cast<integer>(double) is undefined for NaN and out-of-range values; these follow what the reference's x86-64
// build (clang, no AVX-512) computes: cvttsd2si returns the "integer indefinite" value INT64_MIN / INT32_MIN for such
// inputs; targets of up to 32 bits convert through int32 (uint32 through int64) and are then truncated; uint64 is
// cvttsd2si(x) | (cvttsd2si(x - 2^63) & sign(cvttsd2si(x))).
static int64_t CvtI64(double d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return INT64_MIN;
    return int64_t(d);
}
static int32_t CvtI32(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return INT32_MIN;
    return int32_t(d);
}
static uint64_t UnsafeFloatToInt(double v, int to) {
    switch (to) {
        case I64: case TS: case DUR: return uint64_t(CvtI64(v));
        case U64: {
            const int64_t a = CvtI64(v), b = CvtI64(v - 9223372036854775808.0);
            return uint64_t(a) | (uint64_t(b) & uint64_t(a >> 63));
        }
        case U32: return uint64_t(CvtI64(v));
        default: return uint64_t(int64_t(CvtI32(v)));
    }
}

// Numeric cast.  safe = CastOptions::Safe() of Arrow's implicit casts (out-of-range integers and integers
// a float cannot hold are errors: compute/kernels/scalar_cast_numeric.cc:70-290); unsafe = the
// CastOptions(false) of ydb.cast (ydb/library/arrow_kernels/func_cast.cpp:101-149): C++ conversions.
static DatumPtr CastTo(const DatumPtr& in, int to, bool safe) {
    if (in->ty == to) return in;
    const int from = in->ty;
    const bool okFrom = IsNum(from) || from == BOOL || IsTemporal(from);
    const bool okTo = IsNum(to) || to == BOOL || IsTemporal(to);
    if (!okFrom || !okTo) throw Error(std::string("Unsupported cast from ") + TyName(from) + " to " + TyName(to));
    const int64_t n = in->scalar ? 1 : in->n;
    auto out = NewArray(to, n);
    out->scalar = in->scalar;
    out->n = in->n;
    out->ok = in->ok;
    const Store sf = StoreOf(from), st = StoreOf(to);
    for (int64_t r = 0; r < n; ++r) {
        const bool valid = in->ok.empty() || in->ok[r];
        if (sf == S_F) {
            const double v = in->f[r];
            if (st == S_F) { out->f[r] = to == F32 ? double(float(v)) : v; continue; }
            if (safe && valid) {
                if (!(v == std::trunc(v))) throw Error("Float value was truncated converting to " + std::string(TyName(to)));
            }
            if (to == BOOL) { out->u[r] = v != 0; continue; }
            if (st == S_I) {
                if (safe && valid && (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0) || !IntFits(true, int64_t(v), 0, to)))
                    throw Error("Float value out of range");
                out->i[r] = WrapI(int64_t(UnsafeFloatToInt(v, to)), to);
            } else {
                if (safe && valid && (!(v >= 0 && v < 18446744073709551616.0) || !IntFits(false, 0, uint64_t(v), to)))
                    throw Error("Float value out of range");
                out->u[r] = WrapU(UnsafeFloatToInt(v, to), to);
            }
            continue;
        }
        const bool sgn = sf == S_I;
        const int64_t si = sgn ? in->i[r] : 0;
        const uint64_t ui = sgn ? uint64_t(si) : in->u[r];
        if (st == S_F) {
            if (safe && valid) {  // integers beyond the float's mantissa are rejected
                const int64_t lim = to == F32 ? (int64_t(1) << 24) : (int64_t(1) << 53);
                if (sgn ? (si < -lim || si > lim) : (ui > uint64_t(lim)))
                    throw Error("Integer value not in range: " + std::to_string(sgn ? si : int64_t(ui)));
            }
            double v = sgn ? double(si) : double(ui);
            if (to == F32) v = double(sgn ? float(si) : float(ui));
            out->f[r] = v;
            continue;
        }
        if (to == BOOL) { out->u[r] = ui != 0; continue; }
        if (safe && valid && from != BOOL && !IntFits(sgn, si, ui, to))
            throw Error("Integer value " + std::to_string(sgn ? si : int64_t(ui)) + " not in range of " + TyName(to));
        if (st == S_I) out->i[r] = WrapI(int64_t(ui), to); else out->u[r] = WrapU(ui, to);
    }
    return out;
}

static void MergeValidity(const Datum& a, const Datum& b, int64_t n, Datum& out) {
    if (a.ok.empty() && b.ok.empty()) return;
    out.ok.assign(n, 1);
    for (int64_t r = 0; r < n; ++r) out.ok[r] = a.Valid(r) && b.Valid(r);
}

static int CompareOrder3(int cmp, int ord /* -1, 0, 1, 2 = unordered */) {
    switch (cmp) {
        case OP_EQ: return ord == 0;
        case OP_NE: return ord != 0;
        case OP_LT: return ord == -1;
        case OP_LE: return ord == -1 || ord == 0;
        case OP_GT: return ord == 1;
        default: return ord == 1 || ord == 0;
    }
}
template <class T> static int Ord(T a, T b) { return a < b ? -1 : a > b ? 1 : a == b ? 0 : 2; }
static int OrdStr(const std::string& a, const std::string& b) { int c = a.compare(b); return c < 0 ? -1 : c > 0 ? 1 : 0; }

static int64_t OutLen(const Datum& a, const Datum& b) { return a.scalar ? (b.scalar ? 1 : b.n) : a.n; }

// equal / not_equal / less / ... (compute/kernels/scalar_compare.cc:36-200): operands implicitly cast to the
// common numeric type; strings compare bytewise; output is a boolean, null where either side is null.
static DatumPtr ArrowCompare(int cmp, DatumPtr a, DatumPtr b) {
    int common;
    if (IsStr(a->ty) && IsStr(b->ty)) common = STR;
    else if (a->ty == b->ty && (IsNum(a->ty) || a->ty == BOOL || IsTemporal(a->ty))) common = a->ty;
    else common = CommonNumeric(a->ty, b->ty);
    if (common == NONE) throw Error(std::string("Function has no kernel matching input types (") + TyName(a->ty) + ", " + TyName(b->ty) + ")");
    if (common != STR) { a = CastTo(a, common, true); b = CastTo(b, common, true); }
    const int64_t n = OutLen(*a, *b);
    auto out = NewArray(BOOL, n);
    out->scalar = a->scalar && b->scalar;
    out->n = out->scalar ? 1 : n;
    MergeValidity(*a, *b, n, *out);
    for (int64_t r = 0; r < n; ++r) {
        int ord;
        switch (StoreOf(common)) {
            case S_I: ord = Ord(a->I(r), b->I(r)); break;
            case S_U: ord = Ord(a->U(r), b->U(r)); break;
            case S_F: ord = Ord(a->F(r), b->F(r)); break;
            default: ord = OrdStr(a->S(r), b->S(r)); break;
        }
        out->u[r] = CompareOrder3(cmp, ord);
    }
    return out;
}

// add / subtract / multiply / divide (compute/kernels/scalar_arithmetic.cc:122-327): unchecked variants, integer
// results wrap in the common type; integer division by zero is an error, MIN / -1 yields 0; the divide kernel
// is applied to non-null slots only (null slots hold 0).
static DatumPtr ArrowArith(int op, DatumPtr a, DatumPtr b) {
    int common = (a->ty == b->ty && IsNum(a->ty)) ? a->ty : CommonNumeric(a->ty, b->ty);
    if (common == NONE) throw Error(std::string("Function has no kernel matching input types (") + TyName(a->ty) + ", " + TyName(b->ty) + ")");
    a = CastTo(a, common, true);
    b = CastTo(b, common, true);
    const int64_t n = OutLen(*a, *b);
    auto out = NewArray(common, n);
    out->scalar = a->scalar && b->scalar;
    out->n = out->scalar ? 1 : n;
    MergeValidity(*a, *b, n, *out);
    for (int64_t r = 0; r < n; ++r) {
        const bool valid = out->ok.empty() || out->ok[r];
        switch (StoreOf(common)) {
            case S_I: {
                const int64_t x = a->I(r), y = b->I(r);
                int64_t v = 0;
                if (op == OP_ADD) v = int64_t(uint64_t(x) + uint64_t(y));
                else if (op == OP_SUB) v = int64_t(uint64_t(x) - uint64_t(y));
                else if (op == OP_MUL) v = int64_t(uint64_t(x) * uint64_t(y));
                else if (valid) {
                    if (y == 0) throw Error("divide by zero");
                    const int64_t mn = Bits(common) == 64 ? INT64_MIN : -(int64_t(1) << (Bits(common) - 1));
                    v = (x == mn && y == -1) ? 0 : x / y;
                }
                out->i[r] = WrapI(v, common);
                break;
            }
            case S_U: {
                const uint64_t x = a->U(r), y = b->U(r);
                uint64_t v = 0;
                if (op == OP_ADD) v = x + y;
                else if (op == OP_SUB) v = x - y;
                else if (op == OP_MUL) v = x * y;
                else if (valid) {
                    if (y == 0) throw Error("divide by zero");
                    v = x / y;
                }
                out->u[r] = WrapU(v, common);
                break;
            }
            default: {
                double v;
                if (common == F32) {
                    const float x = float(a->F(r)), y = float(b->F(r));
                    v = double(op == OP_ADD ? x + y : op == OP_SUB ? x - y : op == OP_MUL ? x * y : x / y);
                    if (op == OP_DIV && !valid) v = 0;
                } else {
                    const double x = a->F(r), y = b->F(r);
                    v = op == OP_ADD ? x + y : op == OP_SUB ? x - y : op == OP_MUL ? x * y : x / y;
                    if (op == OP_DIV && !valid) v = 0;
                }
                out->f[r] = v;
            }
        }
    }
    return out;
}

// and / or / xor / invert (compute/kernels/scalar_boolean.cc:118-160,548-560): the plain variants -- null if
// either input is null.
static DatumPtr ArrowLogic(int op, const DatumPtr& a, const DatumPtr& b) {
    if (a->ty != BOOL || b->ty != BOOL) throw Error("Function has no kernel matching input types");
    const int64_t n = OutLen(*a, *b);
    auto out = NewArray(BOOL, n);
    out->scalar = a->scalar && b->scalar;
    out->n = out->scalar ? 1 : n;
    MergeValidity(*a, *b, n, *out);
    for (int64_t r = 0; r < n; ++r) {
        const bool x = a->U(r), y = b->U(r);
        out->u[r] = op == OP_AND ? (x && y) : op == OP_OR ? (x || y) : (x != y);
    }
    return out;
}

static uint8_t LowerAscii(uint8_t c) { return (c >= 'A' && c <= 'Z') ? uint8_t(c + 32) : c; }
static std::string Lower(const std::string& s) { std::string o = s; for (auto& c : o) c = char(LowerAscii(uint8_t(c))); return o; }

// match_substring / starts_with / ends_with with MatchSubstringOptions{pattern, ignore_case}
// (compute/kernels/scalar_string.cc: PlainSubstringMatcher / PlainStartsWithMatcher / PlainEndsWithMatcher).
// ignore_case lower-cases ASCII on both sides.  kind: 0 contains, 1 prefix, 2 suffix.
static bool MatchOne(int kind, const std::string& s, const std::string& p) {
    if (p.size() > s.size()) return false;
    if (kind == 1) return std::memcmp(s.data(), p.data(), p.size()) == 0;
    if (kind == 2) return std::memcmp(s.data() + s.size() - p.size(), p.data(), p.size()) == 0;
    return s.find(p) != std::string::npos;
}
static DatumPtr StringMatch(int kind, bool icase, const DatumPtr& a, std::string pattern, bool patternValid, int outTy) {
    if (!IsStr(a->ty)) throw Error("Function has no kernel matching input types");
    if (icase) pattern = Lower(pattern);
    const int64_t n = a->scalar ? 1 : a->n;
    auto out = NewArray(outTy, n);
    out->scalar = a->scalar;
    out->n = a->n;
    out->ok = a->ok;
    if (!patternValid) out->ok.assign(n, 0);
    for (int64_t r = 0; r < n; ++r) out->u[r] = MatchOne(kind, icase ? Lower(a->s[r]) : a->s[r], pattern);
    return out;
}

// match_like: Arrow rewrites '%lit%' / 'lit%' / '%lit' (no other wildcard, no escape) to the plain matchers
// (compute/kernels/scalar_string.cc: MatchLike); other patterns go through RE2: LikeRegexMatch below.
static bool SimplifyLike(const std::string& p, int& kind, std::string& lit) {
    auto plain = [](const std::string& s) { return s.find_first_of("%_\\") == std::string::npos; };
    if (p.size() >= 2 && p.front() == '%' && p.back() == '%' && plain(p.substr(1, p.size() - 2))) { kind = 0; lit = p.substr(1, p.size() - 2); return true; }
    if (!p.empty() && p.back() == '%' && plain(p.substr(0, p.size() - 1))) { kind = 1; lit = p.substr(0, p.size() - 1); return true; }
    if (!p.empty() && p.front() == '%' && plain(p.substr(1))) { kind = 2; lit = p.substr(1); return true; }
    return false;
}

// The patterns Arrow does not rewrite become the regular expression "(?s:^...$)" with '%' -> ".*", '_' -> ".", a
// backslash making the next character literal (MakeLikeRegex, compute/kernels/scalar_string.cc:739-778), matched by
// RE2 in UTF-8 mode: '.' is one UTF-8 encoded code point up to U+10FFFF.  Restated without a regex engine: the
// pattern as a token list, and the set of text positions reachable after each token.
static size_t Utf8Len(const std::string& s, size_t i) {
    const auto b = [&](size_t k) { return uint8_t(s[k]); };
    const auto cont = [&](size_t k) { return k < s.size() && (b(k) & 0xC0) == 0x80; };
    if (i >= s.size()) return 0;
    const uint8_t c = b(i);
    if (c < 0x80) return 1;
    if (c >= 0xC2 && c <= 0xDF) return cont(i + 1) ? 2 : 0;
    if (c == 0xE0) return (cont(i + 1) && b(i + 1) >= 0xA0 && cont(i + 2)) ? 3 : 0;
    if (c >= 0xE1 && c <= 0xEF) return (cont(i + 1) && cont(i + 2)) ? 3 : 0;
    if (c == 0xF0) return (cont(i + 1) && b(i + 1) >= 0x90 && cont(i + 2) && cont(i + 3)) ? 4 : 0;
    if (c >= 0xF1 && c <= 0xF3) return (cont(i + 1) && cont(i + 2) && cont(i + 3)) ? 4 : 0;
    if (c == 0xF4) return (cont(i + 1) && b(i + 1) <= 0x8F && cont(i + 2) && cont(i + 3)) ? 4 : 0;
    return 0;
}
static bool LikeRegexMatch(const std::string& s, const std::string& pattern) {
    struct Token { int kind; char lit; };  // 0 literal, 1 any one character, 2 any run of characters
    std::vector<Token> tokens;
    bool escaped = false;
    for (const char c : pattern) {
        if (!escaped && c == '%') tokens.push_back({2, 0});
        else if (!escaped && c == '_') tokens.push_back({1, 0});
        else if (!escaped && c == '\\') escaped = true;
        else { tokens.push_back({0, c}); escaped = false; }
    }
    std::vector<char> at(s.size() + 1, 0), next;
    at[0] = 1;
    for (const Token& t : tokens) {
        next.assign(s.size() + 1, 0);
        for (size_t i = 0; i <= s.size(); ++i) {
            if (!at[i]) continue;
            if (t.kind == 0) {
                if (i < s.size() && s[i] == t.lit) next[i + 1] = 1;
            } else if (t.kind == 1) {
                if (const size_t l = Utf8Len(s, i)) next[i + l] = 1;
            } else {
                for (size_t j = i;;) {
                    next[j] = 1;
                    const size_t l = Utf8Len(s, j);
                    if (!l) break;
                    j += l;
                }
            }
        }
        at.swap(next);
    }
    return at[s.size()] != 0;
}

// ---- YQL block kernels (selected by YqlOperationId, ydb/library/yql/core/arrow_kernels/request/request.h:19-42) ----
enum YqlOp { Y_AND = 0, Y_OR, Y_XOR, Y_ADD, Y_SUB, Y_MUL, Y_DIV, Y_MOD, Y_STARTS, Y_ENDS, Y_CONTAINS, Y_EQ, Y_NE, Y_LT, Y_LE, Y_GT, Y_GE, Y_COALESCE };

// Result type of integral binary kernels (ydb/library/yql/minikql/invoke_builtins/mkql_builtins_impl.h.txt:1415-1535):
// the wider operand wins; equal widths -> the signed one.
static int YqlResultType(int a, int b) {
    if (!IsNum(a) || !IsNum(b)) return NONE;
    if (a == F64 || b == F64) return F64;
    if (a == F32 || b == F32) return F32;
    if (Bits(a) != Bits(b)) return Bits(a) > Bits(b) ? a : b;
    return IsSigned(a) ? a : b;
}

static DatumPtr YqlKernel(uint32_t op, DatumPtr a, DatumPtr b) {
    const int64_t n = OutLen(*a, *b);
    const bool scalar = a->scalar && b->scalar;
    auto isBool = [](const DatumPtr& d) { return d->ty == U8 || d->ty == BOOL; };
    if (op >= Y_EQ && op <= Y_GE) {
        static const int cmps[] = {OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE};
        const int cmp = cmps[op - Y_EQ];
        auto out = NewArray(U8, n);
        out->scalar = scalar;
        out->n = scalar ? 1 : n;
        MergeValidity(*a, *b, n, *out);
        const int ta = a->ty, tb = b->ty;
        if (IsStr(ta) && IsStr(tb)) {
            for (int64_t r = 0; r < n; ++r) out->u[r] = CompareOrder3(cmp, OrdStr(a->S(r), b->S(r)));
        } else if (IsNum(ta) && IsNum(tb) && (IsFloat(ta) || IsFloat(tb))) {
            // mkql_builtins_equals.cpp:33-44 (and siblings): compared in the wider floating type present
            const int ft = (ta == F64 || tb == F64) ? F64 : F32;
            a = CastTo(a, ft, false);
            b = CastTo(b, ft, false);
            for (int64_t r = 0; r < n; ++r) out->u[r] = CompareOrder3(cmp, Ord(a->F(r), b->F(r)));
        } else if (IsNum(ta) && IsNum(tb)) {
            // mkql_builtins_less.cpp:13-44: mixed signedness compares by value
            for (int64_t r = 0; r < n; ++r) {
                int ord;
                if (IsSigned(ta) && IsSigned(tb)) ord = Ord(a->I(r), b->I(r));
                else if (IsUnsigned(ta) && IsUnsigned(tb)) ord = Ord(a->U(r), b->U(r));
                else if (IsSigned(ta)) ord = a->I(r) < 0 ? -1 : Ord(uint64_t(a->I(r)), b->U(r));
                else ord = b->I(r) < 0 ? 1 : Ord(a->U(r), uint64_t(b->I(r)));
                out->u[r] = CompareOrder3(cmp, ord);
            }
        } else if (ta == tb && IsTemporal(ta)) {
            for (int64_t r = 0; r < n; ++r) out->u[r] = CompareOrder3(cmp, Ord(a->I(r), b->I(r)));
        } else if (isBool(a) && isBool(b)) {
            for (int64_t r = 0; r < n; ++r) out->u[r] = CompareOrder3(cmp, Ord(a->U(r), b->U(r)));
        } else {
            throw Error("YQL compare: unsupported types");
        }
        return out;
    }
    if (op >= Y_ADD && op <= Y_MOD) {
        const int rt = YqlResultType(a->ty, b->ty);
        if (rt == NONE) throw Error("YQL arithmetic: unsupported types");
        a = CastTo(a, rt, false);
        b = CastTo(b, rt, false);
        auto out = NewArray(rt, n);
        out->scalar = scalar;
        out->n = scalar ? 1 : n;
        MergeValidity(*a, *b, n, *out);
        const bool divLike = op == Y_DIV || op == Y_MOD;
        if (divLike && IsInt(rt) && out->ok.empty()) out->ok.assign(n, 1);
        for (int64_t r = 0; r < n; ++r) {
            switch (StoreOf(rt)) {
                case S_I: {
                    const int64_t x = a->I(r), y = b->I(r);
                    int64_t v = 0;
                    if (op == Y_ADD) v = int64_t(uint64_t(x) + uint64_t(y));
                    else if (op == Y_SUB) v = int64_t(uint64_t(x) - uint64_t(y));
                    else if (op == Y_MUL) v = int64_t(uint64_t(x) * uint64_t(y));
                    else {
                        // mkql_builtins_div.cpp:30-48, mkql_builtins_mod.cpp: 0 divisor or MIN / -1 -> NULL
                        const int64_t mn = Bits(rt) == 64 ? INT64_MIN : -(int64_t(1) << (Bits(rt) - 1));
                        if (y == 0 || (x == mn && y == -1)) out->ok[r] = 0;
                        else v = op == Y_DIV ? x / y : x % y;
                    }
                    out->i[r] = WrapI(v, rt);
                    break;
                }
                case S_U: {
                    const uint64_t x = a->U(r), y = b->U(r);
                    uint64_t v = 0;
                    if (op == Y_ADD) v = x + y;
                    else if (op == Y_SUB) v = x - y;
                    else if (op == Y_MUL) v = x * y;
                    else if (y == 0) out->ok[r] = 0;
                    else v = op == Y_DIV ? x / y : x % y;
                    out->u[r] = WrapU(v, rt);
                    break;
                }
                default: {
                    double v;
                    if (rt == F32) {
                        const float x = float(a->F(r)), y = float(b->F(r));
                        v = double(op == Y_ADD ? x + y : op == Y_SUB ? x - y : op == Y_MUL ? x * y : op == Y_DIV ? x / y : std::fmod(x, y));
                    } else {
                        const double x = a->F(r), y = b->F(r);
                        v = op == Y_ADD ? x + y : op == Y_SUB ? x - y : op == Y_MUL ? x * y : op == Y_DIV ? x / y : std::fmod(x, y);
                    }
                    out->f[r] = v;
                }
            }
        }
        return out;
    }
    if (op == Y_AND || op == Y_OR || op == Y_XOR) {
        // mkql_block_logical.cpp:130-193 (And), 266-329 (Or): three-valued; 379-417 (Xor): null if either is null.
        if (!isBool(a) || !isBool(b)) throw Error("YQL logic: unsupported types");
        auto out = NewArray(U8, n);
        out->scalar = scalar;
        out->n = scalar ? 1 : n;
        const bool anyNull = !a->ok.empty() || !b->ok.empty();
        if (anyNull) out->ok.assign(n, 1);
        for (int64_t r = 0; r < n; ++r) {
            const bool x = a->U(r) != 0, y = b->U(r) != 0, vx = a->Valid(r), vy = b->Valid(r);
            if (op == Y_AND) {
                out->u[r] = x && y;
                if (anyNull) out->ok[r] = (vx && vy) || (vx && !x) || (vy && !y);
            } else if (op == Y_OR) {
                out->u[r] = x || y;
                if (anyNull) out->ok[r] = (vx && vy) || (vx && x) || (vy && y);
            } else {
                out->u[r] = x != y;
                if (anyNull) out->ok[r] = vx && vy;
            }
        }
        return out;
    }
    if (op == Y_STARTS || op == Y_ENDS || op == Y_CONTAINS) {
        // mkql_builtins_string_kernels.cpp:249-290: bytewise, ui8 result, validity = intersection
        if (!IsStr(a->ty) || !IsStr(b->ty) || !b->scalar) throw Error("YQL string kernel: unsupported arguments");
        return StringMatch(op == Y_STARTS ? 1 : op == Y_ENDS ? 2 : 0, false, a, b->s[0], b->Valid(0), U8);
    }
    if (op == Y_COALESCE) {
        if (a->ty != b->ty || IsStr(a->ty)) throw Error("YQL Coalesce: unsupported types");
        auto out = NewArray(a->ty, n);
        out->scalar = scalar;
        out->n = scalar ? 1 : n;
        if (!a->ok.empty() && !b->ok.empty()) out->ok.assign(n, 1);
        for (int64_t r = 0; r < n; ++r) {
            const bool va = a->Valid(r);
            switch (StoreOf(a->ty)) {
                case S_I: out->i[r] = va ? a->I(r) : b->I(r); break;
                case S_U: out->u[r] = va ? a->U(r) : b->U(r); break;
                default: out->f[r] = va ? a->F(r) : b->F(r); break;
            }
            if (!out->ok.empty()) out->ok[r] = va || b->Valid(r);
        }
        return out;
    }
    throw Error("unknown YqlOperationId");
}

// ---------------------------------------------------------------------------------------------------
// TDatumBatch and the step (ydb/core/formats/arrow/program.cpp:533-902)
// ---------------------------------------------------------------------------------------------------
struct Batch {
    std::vector<std::string> names;
    std::vector<DatumPtr> cols;
    int64_t rows = 0;
    int Find(const std::string& name) const {  // arrow::Schema::GetFieldIndex: -1 when absent or duplicated
        int found = -1, count = 0;
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == name) { found = int(i); ++count; }
        return count == 1 ? found : -1;
    }
    DatumPtr Get(const std::string& name) const {
        int i = Find(name);
        if (i < 0) throw Error("Not found column '" + name + "' or duplicate");
        return cols[size_t(i)];
    }
    void Add(const std::string& name, const DatumPtr& d) {  // TDatumBatch::AddColumn (program.cpp:533-549)
        for (auto& nm : names) if (nm == name) throw Error("Assign to existing column '" + name + "'.");
        if (!d->scalar && d->n != rows) throw Error("Wrong column length");
        names.push_back(name);
        cols.push_back(d);
    }
};


// ---------------------------------------------------------------------------------------------------
// custom-registry kernels (ydb/core/formats/arrow/custom_registry.cpp:47-58) and Arrow's abs / negate / floor / ceil /
// trunc, each on the C type of the column like the reference's templates
// ---------------------------------------------------------------------------------------------------
template <class T> static T GetAs(const Datum& d, int64_t r) {
    switch (StoreOf(d.ty)) {
        case S_I: return T(d.I(r));
        case S_U: return T(d.U(r));
        default: return T(d.F(r));
    }
}
template <class T> static void PutAs(Datum& d, int64_t r, T v) {
    switch (StoreOf(d.ty)) {
        case S_I: d.i[r] = int64_t(v); break;
        case S_U: d.u[r] = uint64_t(v); break;
        default: d.f[r] = double(v); break;
    }
}
template <class F> static void WithCType(int ty, F&& f) {
    switch (ty) {
        case I8: f(int8_t()); break;
        case I16: f(int16_t()); break;
        case I32: f(int32_t()); break;
        case I64: f(int64_t()); break;
        case U8: f(uint8_t()); break;
        case U16: f(uint16_t()); break;
        case U32: f(uint32_t()); break;
        case U64: f(uint64_t()); break;
        case F32: f(float()); break;
        case F64: f(double()); break;
        default: throw Error("Function has no kernel matching input types");
    }
}

// lhs % rhs in T; the one undefined case (MIN % -1) is given its mathematical value
template <class T> static T RemT(T lhs, T rhs) {
    if constexpr (std::is_signed_v<T>) { if (rhs == T(-1)) return 0; }
    return T(lhs % rhs);
}
// func_gcd.h:20-36
template <class T> static T GcdT(T lhs, T rhs) {
    while (rhs != 0) {
        lhs = RemT(lhs, rhs);
        std::swap(lhs, rhs);
    }
    return lhs;
}
// func_mul.h:16-53: the product wraps in T
template <class T> static T MulWrapT(T a, T b) {
    using UT = std::make_unsigned_t<T>;
    return T(UT(uint64_t(UT(a)) * uint64_t(UT(b))));
}
// func_lcm.h:6-22
template <class T> static T LcmT(T lhs, T rhs) {
    const T g = GcdT(lhs, rhs);
    if (g == 0) throw Error("divide by zero");
    const T m = MulWrapT(lhs, rhs);
    if constexpr (std::is_signed_v<T>) {
        if (g == T(-1)) { using UT = std::make_unsigned_t<T>; return T(UT(0) - UT(m)); }
    }
    return T(m / g);
}
// func_modulo.h:6-30, func_modulo_or_zero.h:6-28: floats go through int64
// static_cast<int64_t>(double) of a NaN or an out-of-range value is undefined; the reference's x86-64 build gets
// cvttsd2si's "integer indefinite", INT64_MIN, which is pinned here
static int64_t ToI64LikeX86(double d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return INT64_MIN;
    return int64_t(d);
}
template <class T> static T ModT(T lhs, T rhs, bool orZero) {
    if constexpr (std::is_floating_point_v<T>) {
        const int64_t x = ToI64LikeX86(double(lhs)), y = ToI64LikeX86(double(rhs));
        if (y == 0) { if (orZero) return 0; throw Error("divide by zero"); }
        return T(RemT<int64_t>(x, y));
    } else {
        if (rhs == 0) { if (orZero) return 0; throw Error("divide by zero"); }
        return RemT(lhs, rhs);
    }
}

static DatumPtr CustomBinary(int op, const char* name, DatumPtr a, DatumPtr b) {
    const bool intsOnly = (op == OP_GCD || op == OP_LCM);
    int common = (a->ty == b->ty && IsNum(a->ty)) ? a->ty : CommonNumeric(a->ty, b->ty);
    if (common == NONE || (intsOnly && !IsInt(common)))
        throw Error(std::string("Function '") + name + "' has no kernel matching input types (" + TyName(a->ty) + ", " + TyName(b->ty) + ")");
    a = CastTo(a, common, true);
    b = CastTo(b, common, true);
    const int64_t n = OutLen(*a, *b);
    auto out = NewArray(common, n);
    out->scalar = a->scalar && b->scalar;
    out->n = out->scalar ? 1 : n;
    MergeValidity(*a, *b, n, *out);
    // the applicator visits every slot, null ones too (compute/kernels/codegen_internal.h ScalarBinary::ArrayArray)
    WithCType(common, [&](auto tag) {
        using T = decltype(tag);
        for (int64_t r = 0; r < n; ++r) {
            const T x = GetAs<T>(*a, r), y = GetAs<T>(*b, r);
            T v;
            if constexpr (std::is_floating_point_v<T>) {
                v = ModT(x, y, op == OP_MODZ);
            } else {
                v = op == OP_GCD ? GcdT(x, y) : op == OP_LCM ? LcmT(x, y) : ModT(x, y, op == OP_MODZ);
            }
            PutAs<T>(*out, r, v);
        }
    });
    return out;
}

// func_round.h:23-76 and scalar_arithmetic.cc:81-96,362-378,820-838
template <class T> static T UnaryT(int op, T x) {
    using UT = std::conditional_t<std::is_integral_v<T>, std::make_unsigned<T>, std::enable_if<true, T>>;
    switch (op) {
        case OP_ABS:
            if constexpr (std::is_floating_point_v<T>) return std::fabs(x);
            else if constexpr (std::is_signed_v<T>) return x < 0 ? T(typename UT::type(0) - typename UT::type(x)) : x;
            else return x;
        case OP_NEGATE:
            if constexpr (std::is_floating_point_v<T>) return -x;
            else return T(typename UT::type(0) - typename UT::type(x));
        case OP_ROUND: case OP_ROUNDBANKERS: {
            if constexpr (std::is_floating_point_v<T>) return op == OP_ROUND ? std::round(x) : std::rint(x);
            else {
                const double d = op == OP_ROUND ? std::round(double(x)) : std::rint(double(x));
                // conversions beyond the type's range are undefined in the reference; pinned here to what x86-64 gives
                if constexpr (std::is_signed_v<T>) return T(d >= 9223372036854775808.0 ? INT64_MIN : int64_t(d));
                else return T(d >= 18446744073709551616.0 ? uint64_t(0) : uint64_t(d));
            }
        }
        case OP_ROUNDEXP2:
            if constexpr (std::is_same_v<T, double>) {
                uint64_t bits; std::memcpy(&bits, &x, 8); bits &= ~((uint64_t(1) << 52) - 1); double r; std::memcpy(&r, &bits, 8); return r;
            } else if constexpr (std::is_same_v<T, float>) {
                uint32_t bits; std::memcpy(&bits, &x, 4); bits &= ~((uint32_t(1) << 23) - 1); float r; std::memcpy(&r, &bits, 4); return r;
            } else if constexpr (sizeof(T) <= 4) {
                return x <= 0 ? T(0) : T(T(1) << (31 - __builtin_clz(uint32_t(x))));
            } else {
                return x <= 0 ? T(0) : T(T(1) << (63 - __builtin_clzll(uint64_t(x))));
            }
        default: return x;
    }
}

static DatumPtr UnaryLike(const DatumPtr& in, int outTy) {
    const int64_t n = in->scalar ? 1 : in->n;
    auto out = NewArray(outTy, n);
    out->scalar = in->scalar;
    out->n = in->n;
    out->ok = in->ok;
    return out;
}

// add_checked / subtract_checked / multiply_checked / divide_checked (compute/kernels/scalar_arithmetic.cc:147-360):
// overflow of the column's C type is an error; only non-null slots are visited, the others hold 0.
template <class T> static T CheckedT(int op, T x, T y) {
    if constexpr (std::is_floating_point_v<T>) {
        if (op == OP_DIV_CHECKED) {
            if (y == 0) throw Error("divide by zero");
            return x / y;
        }
        return op == OP_ADD_CHECKED ? x + y : op == OP_SUB_CHECKED ? x - y : x * y;
    } else {
        if (op == OP_DIV_CHECKED) {
            if (y == 0) throw Error("divide by zero");
            if constexpr (std::is_signed_v<T>) {
                if (x == std::numeric_limits<T>::min() && y == T(-1)) throw Error("overflow");
            }
            return T(x / y);
        }
        T r = 0;
        const bool over = op == OP_ADD_CHECKED ? __builtin_add_overflow(x, y, &r)
                          : op == OP_SUB_CHECKED ? __builtin_sub_overflow(x, y, &r) : __builtin_mul_overflow(x, y, &r);
        if (over) throw Error("overflow");
        return r;
    }
}

static DatumPtr CheckedArith(int op, const char* name, DatumPtr a, DatumPtr b) {
    int common = (a->ty == b->ty && IsNum(a->ty)) ? a->ty : CommonNumeric(a->ty, b->ty);
    if (common == NONE)
        throw Error(std::string("Function '") + name + "' has no kernel matching input types (" + TyName(a->ty) + ", " + TyName(b->ty) + ")");
    a = CastTo(a, common, true);
    b = CastTo(b, common, true);
    const int64_t n = OutLen(*a, *b);
    auto out = NewArray(common, n);
    out->scalar = a->scalar && b->scalar;
    out->n = out->scalar ? 1 : n;
    MergeValidity(*a, *b, n, *out);
    WithCType(common, [&](auto tag) {
        using T = decltype(tag);
        for (int64_t r = 0; r < n; ++r) {
            if (!out->ok.empty() && !out->ok[r]) continue;
            PutAs<T>(*out, r, CheckedT<T>(op, GetAs<T>(*a, r), GetAs<T>(*b, r)));
        }
    });
    return out;
}

// hypot (func_math.h:111-119 through MathBinaryExec): operands cast to the common type, double out
static DatumPtr MathBinary(const char* name, DatumPtr a, DatumPtr b) {
    int common = (a->ty == b->ty && IsNum(a->ty)) ? a->ty : CommonNumeric(a->ty, b->ty);
    if (common == NONE)
        throw Error(std::string("Function '") + name + "' has no kernel matching input types (" + TyName(a->ty) + ", " + TyName(b->ty) + ")");
    a = CastTo(a, common, true);
    b = CastTo(b, common, true);
    const int64_t n = OutLen(*a, *b);
    auto out = NewArray(F64, n);
    out->scalar = a->scalar && b->scalar;
    out->n = out->scalar ? 1 : n;
    MergeValidity(*a, *b, n, *out);
    WithCType(common, [&](auto tag) {
        using T = decltype(tag);
        for (int64_t r = 0; r < n; ++r) out->f[r] = double(std::hypot(GetAs<T>(*a, r), GetAs<T>(*b, r)));
    });
    return out;
}

static DatumPtr CustomUnary(int op, const char* name, DatumPtr a) {
    if (!IsNum(a->ty)) throw Error(std::string("Function '") + name + "' has no kernel matching input types (" + TyName(a->ty) + ")");
    if (op == OP_FLOOR || op == OP_CEIL || op == OP_TRUNC) {
        if (IsInt(a->ty)) a = CastTo(a, F64, true);  // DispatchBest of the floating-point functions
        auto out = UnaryLike(a, a->ty);
        for (size_t r = 0; r < out->f.size(); ++r) {
            const double x = a->f[r];
            out->f[r] = op == OP_FLOOR ? std::floor(x) : op == OP_CEIL ? std::ceil(x) : std::trunc(x);
        }
        return out;
    }
    if (op == OP_SQRT || (op >= OP_ACOSH && op <= OP_TGAMMA)) {
        // func_math.h:7-170 through MathUnaryExec: double out; std:: picks the float overload for a float argument and
        // the double one for integers
        auto out = UnaryLike(a, F64);
        WithCType(a->ty, [&](auto tag) {
            using T = decltype(tag);
            for (size_t r = 0; r < out->f.size(); ++r) {
                const T x = GetAs<T>(*a, int64_t(r));
                double v;
                switch (op) {
                    case OP_SQRT: v = double(std::sqrt(x)); break;
                    case OP_ACOSH: v = double(std::acosh(x)); break;
                    case OP_ATANH: v = double(std::atanh(x)); break;
                    case OP_CBRT: v = double(std::cbrt(x)); break;
                    case OP_COSH: v = double(std::cosh(x)); break;
                    case OP_ERF: v = double(std::erf(x)); break;
                    case OP_ERFC: v = double(std::erfc(x)); break;
                    case OP_EXP: v = double(std::exp(x)); break;
                    case OP_EXP2: v = double(std::exp2(x)); break;
                    case OP_LGAMMA: v = double(std::lgamma(x)); break;
                    case OP_SINH: v = double(std::sinh(x)); break;
                    default: v = double(std::tgamma(x)); break;
                }
                out->f[r] = v;
            }
        });
        return out;
    }
    auto out = UnaryLike(a, a->ty);
    const int64_t n = a->scalar ? 1 : a->n;
    WithCType(a->ty, [&](auto tag) {
        using T = decltype(tag);
        for (int64_t r = 0; r < n; ++r) PutAs<T>(*out, r, UnaryT<T>(op, GetAs<T>(*a, r)));
    });
    return out;
}

// BlockIf (ydb/library/yql/minikql/comp_nodes/mkql_block_if.cpp:66-250): row i takes then[i] or else[i], value and
// nullness, by the predicate's value byte; a scalar predicate returns the chosen branch as it is.
static DatumPtr YqlIf(const DatumPtr& pred, const DatumPtr& th, const DatumPtr& el) {
    if (!(pred->ty == U8 || pred->ty == BOOL) || pred->MayHaveNulls() || (pred->scalar && !pred->Valid(0)))
        throw Error("Expected bool as first argument");
    if (th->ty != el->ty) throw Error("Different return types in branches.");
    if (IsStr(th->ty) || th->ty == DEC) throw Error("Unsupported: If over string or decimal branches");
    if (pred->scalar) return pred->u[0] ? th : el;
    const int64_t n = pred->n;
    auto out = NewArray(th->ty, n);
    const bool nulls = !th->ok.empty() || !el->ok.empty();
    if (nulls) out->ok.assign(size_t(n), 1);
    for (int64_t r = 0; r < n; ++r) {
        const Datum& src = pred->u[r] ? *th : *el;
        switch (StoreOf(th->ty)) {
            case S_I: out->i[r] = src.I(r); break;
            case S_U: out->u[r] = src.U(r); break;
            default: out->f[r] = src.F(r); break;
        }
        if (nulls) out->ok[r] = src.Valid(r);
    }
    return out;
}

// The unary kernels of NYql::TKernelRequestBuilder::AddUnaryOp (ydb/library/yql/core/arrow_kernels/request/request.cpp:25-46),
// by EUnaryOp (request.h:11-17).  No fixture of the reference holds their results: restated from the kernels' sources.
static DatumPtr YqlUnary(uint32_t op, const DatumPtr& a) {
    const int64_t n = a->scalar ? 1 : a->n;
    switch (op) {
        case 0: {  // Not -- BlockNot (minikql/comp_nodes/mkql_block_logical.cpp:379-417): a Bool block is UInt8 0 / 1; nulls stay
            if (a->ty != U8 && a->ty != BOOL) throw Error("Unsupported: YQL Not of a non-bool");
            auto out = std::make_shared<Datum>(*a);
            for (auto& v : out->u) v = v ? 0 : 1;
            return out;
        }
        case 1: {  // Size -- BlockFunc("Size"): bytes of the string, Uint32 (kqp_olap_compiler.cpp:663-665)
            if (!IsStr(a->ty)) throw Error("Unsupported: YQL Size of a non-string");
            auto out = NewArray(U32, n);
            out->scalar = a->scalar;
            out->n = a->n;
            out->ok = a->ok;
            for (int64_t r = 0; r < n; ++r) out->u[r] = (out->ok.empty() || out->ok[r]) ? uint64_t(a->s[r].size()) : 0;
            return out;
        }
        case 2:  // Minus -- mkql_builtins_minus.cpp: -x in the argument's type (integers wrap)
            if (!IsNum(a->ty)) throw Error("Unsupported: YQL Minus of a non-number");
            return CustomUnary(OP_NEGATE, "Minus", a);
        case 3:  // Abs -- mkql_builtins_abs.cpp: the argument's type; the smallest integer stays itself, unsigned values too
            if (!IsNum(a->ty)) throw Error("Unsupported: YQL Abs of a non-number");
            return CustomUnary(OP_ABS, "Abs", a);
        case 4:  // Just -- mkql_block_just.cpp: the same values under an optional type (the same Arrow array)
            if (!a->ok.empty()) throw Error("Unsupported: YQL Just of an optional value");
            return a;
        default:
            throw Error("unknown unary YQL kernel");
    }
}

static DatumPtr EvalAssign(const Assign& a, const Batch& b) {
    if (a.op == OP_CONST) return a.constant;
    std::vector<DatumPtr> args;
    for (auto& nm : a.args) args.push_back(b.Get(nm));
    switch (a.op) {
        case OP_EQ: case OP_NE: case OP_LT: case OP_LE: case OP_GT: case OP_GE: return ArrowCompare(a.op, args[0], args[1]);
        case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: return ArrowArith(a.op, args[0], args[1]);
        case OP_AND: case OP_OR: case OP_XOR: return ArrowLogic(a.op, args[0], args[1]);
        case OP_NOT: {
            if (args[0]->ty != BOOL) throw Error("Function 'invert' has no kernel matching input types");
            auto out = std::make_shared<Datum>(*args[0]);
            for (auto& v : out->u) v = !v;
            return out;
        }
        case OP_ISNULL: {  // is_null: never null itself
            const int64_t n = args[0]->scalar ? 1 : args[0]->n;
            auto out = NewArray(BOOL, n);
            out->scalar = args[0]->scalar;
            out->n = args[0]->n;
            for (int64_t r = 0; r < n; ++r) out->u[r] = !args[0]->Valid(r);
            return out;
        }
        case OP_ISVALID: {
            const int64_t n = args[0]->scalar ? 1 : args[0]->n;
            auto out = NewArray(BOOL, n);
            out->scalar = args[0]->scalar;
            out->n = args[0]->n;
            for (int64_t r = 0; r < n; ++r) out->u[r] = args[0]->Valid(r);
            return out;
        }
        case OP_GCD: return CustomBinary(a.op, "gcd", args[0], args[1]);
        case OP_LCM: return CustomBinary(a.op, "lcm", args[0], args[1]);
        case OP_MODULO: return CustomBinary(a.op, "mod", args[0], args[1]);
        case OP_MODZ: return CustomBinary(a.op, "modOrZero", args[0], args[1]);
        case OP_ABS: return CustomUnary(a.op, "abs", args[0]);
        case OP_NEGATE: return CustomUnary(a.op, "negate", args[0]);
        case OP_ROUND: return CustomUnary(a.op, "round", args[0]);
        case OP_ROUNDBANKERS: return CustomUnary(a.op, "roundBankers", args[0]);
        case OP_ROUNDEXP2: return CustomUnary(a.op, "roundToExp2", args[0]);
        case OP_FLOOR: return CustomUnary(a.op, "floor", args[0]);
        case OP_CEIL: return CustomUnary(a.op, "ceil", args[0]);
        case OP_TRUNC: return CustomUnary(a.op, "trunc", args[0]);
        case OP_SQRT: return CustomUnary(a.op, "sqrt", args[0]);
        case OP_ACOSH: return CustomUnary(a.op, "acosh", args[0]);
        case OP_ATANH: return CustomUnary(a.op, "atanh", args[0]);
        case OP_CBRT: return CustomUnary(a.op, "cbrt", args[0]);
        case OP_COSH: return CustomUnary(a.op, "cosh", args[0]);
        case OP_ERF: return CustomUnary(a.op, "erf", args[0]);
        case OP_ERFC: return CustomUnary(a.op, "erfc", args[0]);
        case OP_EXP: return CustomUnary(a.op, "exp", args[0]);
        case OP_EXP2: return CustomUnary(a.op, "exp2", args[0]);
        case OP_LGAMMA: return CustomUnary(a.op, "lgamma", args[0]);
        case OP_SINH: return CustomUnary(a.op, "sinh", args[0]);
        case OP_TGAMMA: return CustomUnary(a.op, "tgamma", args[0]);
        case OP_HYPOT: return MathBinary("hypot", args[0], args[1]);
        case OP_ADD_CHECKED: return CheckedArith(a.op, "add_checked", args[0], args[1]);
        case OP_SUB_CHECKED: return CheckedArith(a.op, "subtract_checked", args[0], args[1]);
        case OP_MUL_CHECKED: return CheckedArith(a.op, "multiply_checked", args[0], args[1]);
        case OP_DIV_CHECKED: return CheckedArith(a.op, "divide_checked", args[0], args[1]);
        case OP_E: return ScalarF(F64, 2.7182818284590452353602874713526624977572470);   // func_math.h:47-56
        case OP_PI: return ScalarF(F64, 3.1415926535897932384626433832795028841971693);  // func_math.h:131-140
        case OP_STRLEN: {  // binary_length -> int32 for utf8/binary (compute/kernels/scalar_string.cc:3285-3299)
            if (!IsStr(args[0]->ty)) throw Error("Function 'binary_length' has no kernel matching input types");
            const int64_t n = args[0]->scalar ? 1 : args[0]->n;
            auto out = NewArray(I32, n);
            out->scalar = args[0]->scalar;
            out->n = args[0]->n;
            out->ok = args[0]->ok;
            for (int64_t r = 0; r < n; ++r) out->i[r] = (out->ok.empty() || out->ok[r]) ? int64_t(args[0]->s[r].size()) : 0;
            return out;
        }
        case OP_MATCH: case OP_STARTS: case OP_ENDS: case OP_MATCH_LIKE: {
            if (!a.hasPattern) throw Error("MatchSubstringOptions required");
            int kind = a.op == OP_MATCH ? 0 : a.op == OP_STARTS ? 1 : 2;
            std::string pat = a.pattern;
            if (a.op == OP_MATCH_LIKE && !SimplifyLike(a.pattern, kind, pat)) {
                if (a.ignoreCase) throw Error("Unsupported: case-insensitive general LIKE pattern");
                if (!IsStr(args[0]->ty)) throw Error("Function has no kernel matching input types");
                const int64_t n = args[0]->scalar ? 1 : args[0]->n;
                auto out = NewArray(BOOL, n);
                out->scalar = args[0]->scalar;
                out->n = args[0]->n;
                out->ok = args[0]->ok;
                for (int64_t r = 0; r < n; ++r) out->u[r] = LikeRegexMatch(args[0]->s[r], a.pattern);
                return out;
            }
            // The reference matches ignore_case patterns with RE2, which folds Unicode: a pattern byte >= 0x80 starts a
            // character with its own case orbit, and 'k' / 's' also match U+212A KELVIN SIGN / U+017F LONG S
            // (arrow/compute/kernels/scalar_string.cc, MatchSubstring with ignore_case -> RE2::Options::set_case_sensitive(false)).
            // Folding ASCII letters only is the same thing for every other pattern; those are declined.
            if (a.ignoreCase) {
                for (unsigned char ch : pat) {
                    if (ch >= 0x80 || ch == 'k' || ch == 'K' || ch == 's' || ch == 'S') throw Error("Unsupported: case-insensitive match whose pattern needs Unicode case folding");
                }
            }
            return StringMatch(kind, a.ignoreCase, args[0], pat, true, BOOL);
        }
        case OP_CAST: {
            // ydb.cast: timestamp only from uint32 seconds?  (func_cast.cpp:101-119 registers uint32 -> timestamp)
            const int from = args[0]->ty;
            if (a.castTo == TS && from != U32 && from != TS) throw Error("Function 'ydb_cast_timestamp' has no kernel matching input types");
            if (IsTemporal(from) && a.castTo != I64 && a.castTo != from) throw Error("Unsupported cast");
            return CastTo(args[0], a.castTo, false);
        }
        case OP_YQL:
            // the only kernel of three arguments is If, sent with a KernelIdx alone
            // (ydb/core/kqp/query_compiler/kqp_olap_compiler.cpp:740-759)
            if ((!a.yqlHasOp || a.yqlOp == 0x10000u) && args.size() == 3) return YqlIf(args[0], args[1], args[2]);
            if (!a.yqlHasOp) throw Error("Unsupported: YQL kernel without YqlOperationId");
            if (a.yqlOp >= 0x100u && a.yqlOp <= 0x104u) {
                if (args.size() != 1) throw Error("Expected 1 arg");
                return YqlUnary(a.yqlOp - 0x100u, args[0]);
            }
            if (args.size() != 2) throw Error("Unsupported: YQL kernel arity");
            return YqlKernel(a.yqlOp, args[0], args[1]);
        default:
            throw Error("Unsupported operation");
    }
}

// arrow::compute::Filter over one column with a boolean mask (order-preserving compaction)
static DatumPtr TakeMask(const DatumPtr& d, const std::vector<uint8_t>& mask, int64_t selected) {
    if (d->scalar) return d;
    auto out = NewArray(d->ty, selected);
    if (!d->ok.empty()) out->ok.assign(selected, 1);
    int64_t o = 0;
    const Store st = StoreOf(d->ty);
    for (int64_t r = 0; r < d->n; ++r) {
        if (!mask[r]) continue;
        if (d->ty == DEC) { out->u[o] = d->u[r]; out->hi[o] = d->hi[r]; }
        else if (st == S_I) out->i[o] = d->i[r];
        else if (st == S_U) out->u[o] = d->u[r];
        else if (st == S_F) out->f[o] = d->f[r];
        else out->s[o] = d->s[r];
        if (!d->ok.empty()) out->ok[o] = d->ok[r];
        ++o;
    }
    return out;
}

// TDatumBatch::ToRecordBatch -> MakeArrayFromScalar (program.cpp:582-598)
static DatumPtr Materialize(const DatumPtr& d, int64_t rows) {
    if (!d->scalar) return d;
    auto out = NewArray(d->ty, rows);
    const Store st = StoreOf(d->ty);
    for (int64_t r = 0; r < rows; ++r) {
        if (d->ty == DEC) { out->u[r] = d->u[0]; out->hi[r] = d->hi[0]; }
        else if (st == S_I) out->i[r] = d->i[0]; else if (st == S_U) out->u[r] = d->u[0]; else if (st == S_F) out->f[r] = d->f[0]; else out->s[r] = d->s[0];
    }
    if (!d->ok.empty() && !d->ok[0]) out->ok.assign(rows, 0);
    return out;
}

// ---- aggregation without keys: Arrow's scalar aggregates ----------------------------------------------
// sum (compute/kernels/aggregate_internal.h:71-161): floats are summed pairwise over blocks of 16, separately for
// every run of valid values; integers wrap in int64 / uint64.
struct PairwiseState {
    std::vector<double> sum = std::vector<double>(64, 0.0);
    uint64_t mask = 0;
    int rootLevel = 0;
};
// The reference's SumArray keeps one levels array across all valid runs of the array.
static double PairwiseSum(const Datum& d) {
    constexpr int kBlock = 16;
    PairwiseState st;
    auto reduce = [&](double blockSum) {
        int cur = 0;
        uint64_t curMask = 1;
        st.sum[cur] += blockSum;
        st.mask ^= curMask;
        while ((st.mask & curMask) == 0) {
            blockSum = st.sum[cur];
            st.sum[cur] = 0;
            ++cur;
            curMask <<= 1;
            st.sum[cur] += blockSum;
            st.mask ^= curMask;
        }
        st.rootLevel = std::max(st.rootLevel, cur);
    };
    auto run = [&](int64_t pos, int64_t len) {
        const double* v = d.f.data() + pos;
        const int64_t blocks = len / kBlock, rem = len % kBlock;
        for (int64_t b = 0; b < blocks; ++b) {
            double s = 0;
            for (int k = 0; k < kBlock; ++k) s += v[b * kBlock + k];
            reduce(s);
        }
        if (rem) {
            double s = 0;
            for (int64_t k = 0; k < rem; ++k) s += v[blocks * kBlock + k];
            reduce(s);
        }
    };
    if (d.ok.empty()) {
        run(0, d.n);
    } else {
        int64_t r = 0;
        while (r < d.n) {
            while (r < d.n && !d.ok[r]) ++r;
            int64_t b = r;
            while (r < d.n && d.ok[r]) ++r;
            if (r > b) run(b, r - b);
        }
    }
    for (int i = 1; i <= st.rootLevel; ++i) st.sum[i] += st.sum[i - 1];
    return st.sum[st.rootLevel];
}

static int SumOutType(int t) {
    if (IsSigned(t)) return I64;
    if (IsUnsigned(t)) return U64;
    if (IsFloat(t)) return F64;
    if (t == DEC) return DEC;
    return NONE;
}

static void AddDec(uint64_t& lo, int64_t& hi, uint64_t alo, int64_t ahi) {
    const uint64_t nlo = lo + alo;
    hi = int64_t(uint64_t(hi) + uint64_t(ahi) + (nlo < lo ? 1 : 0));
    lo = nlo;
}

// One aggregate over a whole (materialised) column -> scalar (ApplyAggregates, program.cpp:712-721; TAggregateFunction
// program.cpp:97-126; ScalarAggregateOptions(skip_nulls = true, min_count = 1)).
static DatumPtr AggregateNoKeys(Agg fn, const DatumPtr& argIn, int64_t rows) {
    DatumPtr arg = argIn ? Materialize(argIn, rows) : nullptr;
    if (fn == AG_COUNT) {
        uint64_t c = 0;
        if (!arg) c = uint64_t(rows);
        else for (int64_t r = 0; r < arg->n; ++r) c += arg->Valid(r);
        return ScalarU(U64, c);  // count's int64 is cast to uint64 (program.cpp:70-72)
    }
    if (!arg) throw Error("aggregate needs an argument");
    int64_t nvalid = 0;
    for (int64_t r = 0; r < arg->n; ++r) nvalid += arg->Valid(r);
    const int t = arg->ty;
    if (fn == AG_SUM) {
        const int ot = SumOutType(t);
        if (ot == NONE) throw Error(std::string("Function 'sum' has no kernel matching input types (") + TyName(t) + ")");
        DatumPtr out = NewArray(ot, 1);
        out->scalar = true;
        if (ot == I64) { uint64_t s = 0; for (int64_t r = 0; r < arg->n; ++r) if (arg->Valid(r)) s += uint64_t(arg->i[r]); out->i[0] = int64_t(s); }
        else if (ot == U64) { uint64_t s = 0; for (int64_t r = 0; r < arg->n; ++r) if (arg->Valid(r)) s += arg->u[r]; out->u[0] = s; }
        else if (ot == F64) out->f[0] = PairwiseSum(*arg);
        else { uint64_t lo = 0; int64_t hi = 0; for (int64_t r = 0; r < arg->n; ++r) if (arg->Valid(r)) AddDec(lo, hi, arg->u[r], arg->hi[r]); out->u[0] = lo; out->hi[0] = hi; }
        if (nvalid == 0) out->ok.assign(1, 0);
        return out;
    }
    if (fn == AG_MIN || fn == AG_MAX) {
        // min_max (compute/kernels/aggregate_basic_internal.h:207-228): floats fold with fmin / fmax from +-inf
        if (IsStr(t)) {
            // Arrow 5.0's min_max has no string kernel: the call falls through to ch.min / ch.max (program.cpp:97-103),
            // a SingleValueDataString state: the first value, then bytewise `<` / `>`
            // (AggregateFunctionMinMaxAny.h:160-341)
            DatumPtr out = NewArray(t, 1);
            out->scalar = true;
            out->ok.assign(1, 0);
            for (int64_t r = 0; r < arg->n; ++r) {
                if (!arg->Valid(r)) continue;
                if (!out->ok.empty()) { out->s[0] = arg->s[r]; out->ok.clear(); continue; }
                if (fn == AG_MIN ? arg->s[r] < out->s[0] : arg->s[r] > out->s[0]) out->s[0] = arg->s[r];
            }
            return out;
        }
        if (!(IsNum(t) || IsTemporal(t))) throw Error(std::string("min/max over ") + TyName(t) + " is unsupported");
        DatumPtr out = NewArray(t, 1);
        out->scalar = true;
        const bool mn = fn == AG_MIN;
        if (StoreOf(t) == S_I) {
            int64_t v = mn ? INT64_MAX : INT64_MIN;
            for (int64_t r = 0; r < arg->n; ++r) if (arg->Valid(r)) v = mn ? std::min(v, arg->i[r]) : std::max(v, arg->i[r]);
            out->i[0] = nvalid ? v : 0;
        } else if (StoreOf(t) == S_U) {
            uint64_t v = mn ? UINT64_MAX : 0;
            for (int64_t r = 0; r < arg->n; ++r) if (arg->Valid(r)) v = mn ? std::min(v, arg->u[r]) : std::max(v, arg->u[r]);
            out->u[0] = nvalid ? v : 0;
        } else {
            double v = mn ? std::numeric_limits<double>::infinity() : -std::numeric_limits<double>::infinity();
            for (int64_t r = 0; r < arg->n; ++r) if (arg->Valid(r)) v = mn ? std::fmin(v, arg->f[r]) : std::fmax(v, arg->f[r]);
            out->f[0] = nvalid ? v : 0;
        }
        if (nvalid == 0) out->ok.assign(1, 0);
        return out;
    }
    if (fn == AG_SOME) {
        // no Arrow function of that name: ch.any = the first non-null value
        // (ydb/library/arrow_clickhouse/AggregateFunctions/AggregateFunctionMinMaxAny.h:526-534)
        if (!(IsNum(t) || IsTemporal(t) || IsStr(t))) throw Error(std::string("some() over ") + TyName(t) + " is unsupported");
        DatumPtr out = NewArray(t, 1);
        out->scalar = true;
        out->ok.assign(1, 0);
        for (int64_t r = 0; r < arg->n; ++r) {
            if (!arg->Valid(r)) continue;
            if (IsStr(t)) out->s[0] = arg->s[r];
            else if (StoreOf(t) == S_I) out->i[0] = arg->i[r]; else if (StoreOf(t) == S_U) out->u[0] = arg->u[r]; else out->f[0] = arg->f[r];
            out->ok.clear();
            break;
        }
        return out;
    }
    throw Error("Unexpected agregate function in GROUP BY");
}

// ---- aggregation with keys: the ClickHouse-derived aggregator ---------------------------------------------
// Method choice (ydb/library/arrow_clickhouse/Aggregator.cpp:132-226): with a nullable key and only fixed-width
// keys of <= 32 bytes in total (incl. the null bitmap) the nullable_keys methods make NULL a key value of its own;
// every other combination with a nullable key falls to `serialized`, which reads the raw slot (a null string is the
// empty string, ColumnsCommon.cpp:308-365).  The single one-byte key uses the lookup-table path that does not look at
// the validity of the aggregate arguments (IAggregateFunction.h:258-296,384-459).
struct GroupState {
    std::vector<uint8_t> has;       // per aggregate: saw a value
    std::vector<int64_t> i;
    std::vector<uint64_t> u;
    std::vector<double> f;
    std::vector<int64_t> hi;
    std::vector<std::string> s;
    int64_t firstRow = 0;
};

static void AppendRaw(std::string& key, const Datum& d, int64_t r) {
    switch (d.ty) {
        case STR: case BIN: {
            const std::string& s = d.S(r);
            uint64_t len = s.size();
            key.append(reinterpret_cast<const char*>(&len), 8);
            key.append(s);
            break;
        }
        default: {
            const Store st = StoreOf(d.ty);
            uint64_t bits;
            if (d.ty == F32) { float fv = float(d.F(r)); uint32_t b; std::memcpy(&b, &fv, 4); bits = b; }
            else if (st == S_F) { double dv = d.F(r); std::memcpy(&bits, &dv, 8); }
            else bits = st == S_I ? uint64_t(d.I(r)) : d.U(r);
            key.append(reinterpret_cast<const char*>(&bits), size_t(Bits(d.ty) / 8));
        }
    }
}

static Batch AggregateKeyed(const Step& step, const Batch& in) {
    const int64_t rows = in.rows;
    std::vector<DatumPtr> keys, args;
    for (auto& k : step.keys) {
        int idx = in.Find(k);
        if (idx < 0) throw Error("No such key for GROUP BY.");
        DatumPtr d = in.cols[size_t(idx)];
        if (d->scalar) throw Error("Unexpected GROUP BY key type.");
        if (d->ty == BOOL) throw Error("serializeValueIntoArena unexpected type bool");
        if (d->ty == DEC) throw Error("Unsupported: decimal GROUP BY key");
        keys.push_back(d);
    }
    bool hasNullableKey = false, allFixed = true;
    int keyBytes = 0;
    for (auto& k : keys) {
        hasNullableKey |= !k->ok.empty();  // MayHaveNulls(): a validity buffer is present
        if (IsStr(k->ty)) allFixed = false; else keyBytes += Bits(k->ty) / 8;
    }
    const bool nullAware = hasNullableKey && allFixed && keyBytes <= 28;
    const bool key8 = !hasNullableKey && keys.size() == 1 && !IsStr(keys[0]->ty) && Bits(keys[0]->ty) == 8;
    for (auto& g : step.groupBy) {
        DatumPtr a;
        if (!g.args.empty()) {
            int idx = in.Find(g.args[0]);
            if (idx < 0) {
                if (g.fn != AG_COUNT) throw Error("Not found column '" + g.args[0] + "' or duplicate");
            } else {
                a = in.cols[size_t(idx)];
                if (a->scalar) throw Error("Unsupported: aggregate over a scalar");
            }
        }
        if (g.fn != AG_COUNT) {
            const int t = a ? a->ty : NONE;
            if (g.fn == AG_SUM && SumOutType(t) == NONE) throw Error(std::string("Function 'sum' has no kernel matching input types (") + TyName(t) + ")");
            if (g.fn != AG_SUM && !(IsNum(t) || IsTemporal(t) || IsStr(t))) throw Error("Unsupported aggregate argument type");
            if (g.fn == AG_SUM && t == DEC && key8) throw Error("Unsupported: decimal128 SUM on the lookup-table path");
        }
        args.push_back(a);
    }
    const size_t na = step.groupBy.size();
    std::unordered_map<std::string, size_t> index;
    std::vector<GroupState> groups;
    std::vector<int64_t> repRow;
    std::string key;
    auto newState = [&](int64_t r) {
        GroupState st;
        st.has.assign(na, 0);
        st.i.assign(na, 0); st.u.assign(na, 0); st.f.assign(na, 0); st.hi.assign(na, 0);
        st.s.assign(na, std::string());
        st.firstRow = r;
        return st;
    };
    // IAggregateFunction::add of aggregate k for row r
    auto addRow = [&](GroupState& st, size_t k, int64_t r) {
        const Agg fn = step.groupBy[k].fn;
        const Datum* a = args[k].get();
        // addBatch skips rows whose argument is null (IAggregateFunction.h:199-221) -- except on the key8 path
        if (a && !key8 && !a->Valid(r)) return;
        if (fn == AG_COUNT) { ++st.u[k]; st.has[k] = 1; return; }
        if (IsStr(a->ty)) {  // SingleValueDataString (AggregateFunctionMinMaxAny.h:160-341): bytewise compares
            const std::string& v = a->s[r];
            if (!st.has[k]) st.s[k] = v;
            else if (fn == AG_MIN ? v < st.s[k] : fn == AG_MAX ? v > st.s[k] : false) st.s[k] = v;
            st.has[k] = 1;
            return;
        }
        const Store s = StoreOf(a->ty);
        switch (fn) {
            case AG_SUM:  // AggregateFunctionSum.h:31-47: sequential adds in row order, integers wrap
                if (a->ty == DEC) AddDec(st.u[k], st.hi[k], a->u[r], a->hi[r]);  // extension: wrapping 128-bit sum
                else if (s == S_I) st.i[k] = int64_t(uint64_t(st.i[k]) + uint64_t(a->i[r]));
                else if (s == S_U) st.u[k] += a->u[r];
                else st.f[k] += a->f[r];
                break;
            case AG_MIN: case AG_MAX: {  // AggregateFunctionMinMaxAny.h:96-138: first value, then `<` / `>`
                const bool mn = fn == AG_MIN;
                if (!st.has[k]) { st.i[k] = s == S_I ? a->i[r] : 0; st.u[k] = s == S_U ? a->u[r] : 0; st.f[k] = s == S_F ? a->f[r] : 0; }
                else if (s == S_I) { if (mn ? a->i[r] < st.i[k] : a->i[r] > st.i[k]) st.i[k] = a->i[r]; }
                else if (s == S_U) { if (mn ? a->u[r] < st.u[k] : a->u[r] > st.u[k]) st.u[k] = a->u[r]; }
                else { if (mn ? a->f[r] < st.f[k] : a->f[r] > st.f[k]) st.f[k] = a->f[r]; }
                break;
            }
            case AG_SOME:
                if (!st.has[k]) { st.i[k] = s == S_I ? a->i[r] : 0; st.u[k] = s == S_U ? a->u[r] : 0; st.f[k] = s == S_F ? a->f[r] : 0; }
                break;
            default: break;
        }
        st.has[k] = 1;
    };
    // IAggregateFunction::merge of aggregate k: `from` into `st` (Sum: sum += rhs; Count: count += rhs; Min / Max:
    // changeIfLess / changeIfGreater(to); Any: changeFirstTime(to) -- AggregateFunctionMinMaxAny.h:68-138)
    auto mergeState = [&](GroupState& st, const GroupState& from, size_t k) {
        const Agg fn = step.groupBy[k].fn;
        if (!from.has[k]) return;
        if (fn == AG_COUNT) { st.u[k] += from.u[k]; st.has[k] = 1; return; }
        if (IsStr(args[k]->ty)) {
            if (!st.has[k] || (fn == AG_MIN ? from.s[k] < st.s[k] : fn == AG_MAX ? from.s[k] > st.s[k] : false)) st.s[k] = from.s[k];
            st.has[k] = 1;
            return;
        }
        const Store s = StoreOf(args[k]->ty);
        switch (fn) {
            case AG_SUM:
                if (s == S_I) st.i[k] = int64_t(uint64_t(st.i[k]) + uint64_t(from.i[k]));
                else if (s == S_U) st.u[k] += from.u[k];
                else st.f[k] += from.f[k];
                break;
            case AG_MIN: case AG_MAX: {
                const bool mn = fn == AG_MIN;
                bool take = !st.has[k];
                if (!take) {
                    if (s == S_I) take = mn ? from.i[k] < st.i[k] : from.i[k] > st.i[k];
                    else if (s == S_U) take = mn ? from.u[k] < st.u[k] : from.u[k] > st.u[k];
                    else take = mn ? from.f[k] < st.f[k] : from.f[k] > st.f[k];
                }
                if (take) { st.i[k] = from.i[k]; st.u[k] = from.u[k]; st.f[k] = from.f[k]; }
                break;
            }
            case AG_SOME:
                if (!st.has[k]) { st.i[k] = from.i[k]; st.u[k] = from.u[k]; st.f[k] = from.f[k]; }
                break;
            default: break;
        }
        st.has[k] = 1;
    };
    auto groupOf = [&](int64_t r) {
        key.clear();
        for (auto& k : keys) {
            if (nullAware) {
                const bool isNull = !k->Valid(r);
                key.push_back(isNull ? 1 : 0);
                if (isNull) { key.append(size_t(Bits(k->ty) / 8), '\0'); continue; }
            }
            AppendRaw(key, *k, r);
        }
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        const size_t g = groups.size();
        index.emplace(key, g);
        groups.push_back(newState(r));
        return g;
    };
    if (key8) {
        // The single one-byte key takes the lookup-table path (Aggregator.cpp:302-325): every aggregate goes over the
        // batch in four interleaved streams with states of their own (row i of the first rows / 4 * 4 rows belongs to
        // stream i % 4), the streams are merged into the group's state in stream order, and the last rows % 4 rows are
        // added to it one by one (IAggregateFunction.h:384-459).
        const int64_t unrolled = rows / 4 * 4;
        std::vector<std::vector<GroupState>> streams(4);
        for (int64_t r = 0; r < rows; ++r) {
            const size_t g = groupOf(r);
            if (r >= unrolled) continue;
            auto& sub = streams[size_t(r % 4)];
            while (sub.size() <= g) sub.push_back(newState(r));
            for (size_t k = 0; k < na; ++k) {
                // string states allocate from the arena: those aggregates take the plain loop, in row order
                // (IAggregateFunction.h:396-402, 258-296)
                const bool plain = step.groupBy[k].fn != AG_COUNT && args[k] && IsStr(args[k]->ty);
                addRow(plain ? groups[g] : sub[g], k, r);
            }
        }
        for (size_t g = 0; g < groups.size(); ++g) {
            for (int j = 0; j < 4; ++j) {
                if (streams[size_t(j)].size() <= g) continue;
                for (size_t k = 0; k < na; ++k) {
                    const bool plain = step.groupBy[k].fn != AG_COUNT && args[k] && IsStr(args[k]->ty);
                    if (!plain) mergeState(groups[g], streams[size_t(j)][g], k);
                }
            }
        }
        for (int64_t r = unrolled; r < rows; ++r) {
            const size_t g = groupOf(r);
            for (size_t k = 0; k < na; ++k) addRow(groups[g], k, r);
        }
    } else {
        for (int64_t r = 0; r < rows; ++r) {
            const size_t g = groupOf(r);
            for (size_t k = 0; k < na; ++k) addRow(groups[g], k, r);
        }
    }
    // Output: aggregates in GroupBy order, then keys (program.cpp:728-763); rows in first-appearance order here
    // (the reference's order is its hash table's iteration order, i.e. unspecified).
    const int64_t ng = int64_t(groups.size());
    Batch out;
    out.rows = ng;
    for (size_t k = 0; k < na; ++k) {
        const Agg fn = step.groupBy[k].fn;
        const int at = args[k] ? args[k]->ty : NONE;
        const int ot = fn == AG_COUNT ? U64 : fn == AG_SUM ? SumOutType(at) : at;
        DatumPtr col = NewArray(ot, ng);
        bool anyNull = false;
        for (int64_t g = 0; g < ng; ++g) {
            const GroupState& st = groups[size_t(g)];
            const bool valid = fn == AG_COUNT || st.has[k];
            if (!valid) { anyNull = true; continue; }
            if (ot == DEC) { col->u[g] = st.u[k]; col->hi[g] = st.hi[k]; continue; }
            if (IsStr(ot)) { col->s[g] = st.s[k]; continue; }
            switch (StoreOf(ot)) {
                case S_I: col->i[g] = st.i[k]; break;
                case S_U: col->u[g] = st.u[k]; break;
                default: col->f[g] = ot == F32 ? double(float(st.f[k])) : st.f[k]; break;
            }
        }
        if (anyNull) {
            col->ok.assign(ng, 1);
            for (int64_t g = 0; g < ng; ++g) col->ok[g] = groups[size_t(g)].has[k];
        }
        out.names.push_back(step.groupBy[k].name);
        out.cols.push_back(col);
    }
    for (size_t k = 0; k < keys.size(); ++k) {
        const Datum& src = *keys[k];
        DatumPtr col = NewArray(src.ty, ng);
        if (nullAware && !src.ok.empty()) col->ok.assign(ng, 1);
        for (int64_t g = 0; g < ng; ++g) {
            const int64_t r = groups[size_t(g)].firstRow;
            switch (StoreOf(src.ty)) {
                case S_I: col->i[g] = src.i[r]; break;
                case S_U: col->u[g] = src.u[r]; break;
                case S_F: col->f[g] = src.f[r]; break;
                default: col->s[g] = src.s[r]; break;
            }
            if (!col->ok.empty()) col->ok[g] = src.Valid(r);
        }
        out.names.push_back(step.keys[k]);
        out.cols.push_back(col);
    }
    return out;
}

// TProgramStep::Apply (program.cpp:869-902): Assigns -> Filters -> Aggregates -> Projection
static void ApplyStep(const Step& step, Batch& b, std::vector<uint8_t>* filterOut = nullptr) {
    for (auto& a : step.assigns) b.Add(a.name, EvalAssign(a, b));

    // MakeCombinedFilter (program.cpp:469-528,773-796): AND of the predicate columns; Boolean / Int8 / UInt8 arrays,
    // `(bool)Value(i)` -- the validity bitmap is not consulted.
    if (!step.filters.empty()) {
        std::vector<uint8_t> mask(size_t(b.rows), 1);
        for (auto& f : step.filters) {
            DatumPtr d = b.Get(f);
            if (d->scalar) throw Error("Filter predicate '" + f + "' is a scalar");
            if (d->ty != BOOL && d->ty != I8 && d->ty != U8) throw Error("Filter predicate '" + f + "' has type " + TyName(d->ty));
            for (int64_t r = 0; r < b.rows; ++r) {
                const bool v = d->ty == I8 ? d->i[r] != 0 : d->u[r] != 0;
                mask[r] = mask[r] && v;
            }
        }
        int64_t selected = 0;
        for (uint8_t m : mask) selected += m;
        if (filterOut) { *filterOut = mask; return; }
        if (selected != b.rows) {
            // ApplyFilters (program.cpp:798-832): only the columns the rest of the step reads are kept
            std::vector<std::string> needed;
            for (auto& g : step.groupBy) for (auto& a : g.args) needed.push_back(a);
            for (auto& k : step.keys) needed.push_back(k);
            for (auto& p : step.projection) needed.push_back(p);
            Batch nb;
            nb.rows = selected;
            for (size_t i = 0; i < b.cols.size(); ++i) {
                const bool keep = needed.empty() || std::find(needed.begin(), needed.end(), b.names[i]) != needed.end();
                if (!keep && !b.cols[i]->scalar) continue;
                nb.names.push_back(b.names[i]);
                nb.cols.push_back(TakeMask(b.cols[i], mask, selected));
            }
            b = std::move(nb);
        }
    } else if (filterOut) {
        filterOut->assign(size_t(b.rows), 1);
        return;
    }

    if (!step.groupBy.empty()) {
        if (step.keys.empty()) {
            Batch out;
            out.rows = 1;
            for (auto& g : step.groupBy) {
                DatumPtr arg;
                if (!g.args.empty()) {
                    int idx = b.Find(g.args[0]);
                    if (idx >= 0) arg = b.cols[size_t(idx)];
                    else if (g.fn != AG_COUNT) throw Error("Not found column '" + g.args[0] + "' or duplicate");
                }
                if (arg && arg->scalar) throw Error("Unsupported: aggregate over a scalar");
                DatumPtr s = AggregateNoKeys(g.fn, arg, b.rows);
                out.names.push_back(g.name);
                out.cols.push_back(Materialize(s, 1));
            }
            b = std::move(out);
        } else {
            b = AggregateKeyed(step, b);
        }
    }

    if (!step.projection.empty()) {  // ApplyProjection (program.cpp:834-851)
        Batch out;
        out.rows = b.rows;
        for (auto& p : step.projection) {
            int idx = b.Find(p);
            if (idx < 0) throw Error("Could not find column " + p + " in record batch schema.");
            out.names.push_back(p);
            out.cols.push_back(b.cols[size_t(idx)]);
        }
        b = std::move(out);
    }
    for (auto& c : b.cols) c = Materialize(c, b.rows);
}

}  // namespace orc

// ---------------------------------------------------------------------------------------------------
// C API for ctypes
// ---------------------------------------------------------------------------------------------------
using namespace orc;

extern "C" {

struct orc_col {
    const char* name;
    int32_t type;            // orc::Ty
    int64_t n;
    const void* values;      // natural width (bool: one byte per row; decimal: lo, hi uint64 pairs)
    const uint8_t* valid;    // one byte per row or NULL
    const int32_t* offsets;  // strings: n + 1
    const uint8_t* chars;
};

static char* DupStr(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

void orc_free_string(char* s) { std::free(s); }

static DatumPtr ImportCol(const orc_col& c) {
    auto d = NewArray(c.type, c.n);
    for (int64_t r = 0; r < c.n; ++r) {
        switch (c.type) {
            case BOOL: case U8: d->u[r] = static_cast<const uint8_t*>(c.values)[r]; break;
            case I8: d->i[r] = static_cast<const int8_t*>(c.values)[r]; break;
            case I16: d->i[r] = static_cast<const int16_t*>(c.values)[r]; break;
            case I32: d->i[r] = static_cast<const int32_t*>(c.values)[r]; break;
            case I64: case TS: case DUR: d->i[r] = static_cast<const int64_t*>(c.values)[r]; break;
            case U16: d->u[r] = static_cast<const uint16_t*>(c.values)[r]; break;
            case U32: d->u[r] = static_cast<const uint32_t*>(c.values)[r]; break;
            case U64: d->u[r] = static_cast<const uint64_t*>(c.values)[r]; break;
            case F32: d->f[r] = static_cast<const float*>(c.values)[r]; break;
            case F64: d->f[r] = static_cast<const double*>(c.values)[r]; break;
            case STR: case BIN: d->s[r].assign(reinterpret_cast<const char*>(c.chars) + c.offsets[r], size_t(c.offsets[r + 1] - c.offsets[r])); break;
            case DEC: d->u[r] = static_cast<const uint64_t*>(c.values)[2 * r]; d->hi[r] = int64_t(static_cast<const uint64_t*>(c.values)[2 * r + 1]); break;
            default: throw Error("unsupported column type");
        }
    }
    if (c.valid) d->ok.assign(c.valid, c.valid + c.n);
    return d;
}

void* orc_program_create_with_kernels(const uint8_t* pb, size_t n, int olap_wrapper, const uint32_t* ids, const char* const* names,
                                      size_t ncols, const orc_col* params, size_t nparams, const uint32_t* kernelOps, size_t nkernels, char** err);
void* orc_program_create(const uint8_t* pb, size_t n, int olap_wrapper, const uint32_t* ids, const char* const* names,
                         size_t ncols, const orc_col* params, size_t nparams, char** err) {
    return orc_program_create_with_kernels(pb, n, olap_wrapper, ids, names, ncols, params, nparams, nullptr, 0, err);
}

void* orc_program_create_with_kernels(const uint8_t* pb, size_t n, int olap_wrapper, const uint32_t* ids, const char* const* names,
                                      size_t ncols, const orc_col* params, size_t nparams, const uint32_t* kernelOps, size_t nkernels, char** err) {
    try {
        auto prog = std::make_unique<Program>();
        if (kernelOps) prog->kernelOps.assign(kernelOps, kernelOps + nkernels);
        prog->directOperations = (olap_wrapper & 2) != 0;
        olap_wrapper &= 1;
        for (size_t i = 0; i < ncols; ++i) { prog->resolver[ids[i]] = names[i]; prog->byName[names[i]] = ids[i]; }
        if (ncols) { prog->defaultId = ids[0]; prog->defaultName = names[0]; }
        for (size_t i = 0; i < nparams; ++i) {
            DatumPtr d = ImportCol(params[i]);
            if (d->n < 1) throw Error("empty parameter");
            std::vector<uint8_t> first(size_t(d->n), 0);
            first[0] = 1;
            DatumPtr s = TakeMask(d, first, 1);
            s->scalar = true;
            s->n = 1;
            prog->parameters[params[i].name] = s;
        }
        Wire w(pb, n);
        if (olap_wrapper) {  // TOlapProgram { bytes Program = 1; ... }
            bool found = false;
            while (w.More()) {
                int wt; uint32_t f = w.Tag(wt);
                if (f == 1 && wt == 2) { ParseProgramBody(*prog, w.Sub()); found = true; } else w.Skip(wt);
            }
            if (!found) throw Error("Can't parse TOlapProgram");
        } else {
            ParseProgramBody(*prog, w);
        }
        return prog.release();
    } catch (const std::exception& e) {
        if (err) *err = DupStr(e.what());
        return nullptr;
    }
}

void orc_program_free(void* p) { delete static_cast<Program*>(p); }
int orc_program_steps(void* p) { return int(static_cast<Program*>(p)->steps.size()); }
// '\n'-joined source column names in first-use order
char* orc_program_sources(void* p) {
    std::string s;
    for (auto& kv : static_cast<Program*>(p)->sources) s += kv.second + "\n";
    return DupStr(s);
}

void* orc_batch_import(const orc_col* cols, size_t ncols, int64_t rows, char** err) {
    try {
        auto b = std::make_unique<Batch>();
        b->rows = rows;
        for (size_t i = 0; i < ncols; ++i) {
            if (cols[i].n != rows) throw Error("column length mismatch");
            b->names.push_back(cols[i].name);
            b->cols.push_back(ImportCol(cols[i]));
        }
        return b.release();
    } catch (const std::exception& e) {
        if (err) *err = DupStr(e.what());
        return nullptr;
    }
}
void orc_batch_free(void* b) { delete static_cast<Batch*>(b); }

// TProgram::ApplyTo: every step in order.  Returns a new batch.
void* orc_apply(void* prog, const void* batch, char** err) {
    try {
        auto b = std::make_unique<Batch>(*static_cast<const Batch*>(batch));
        for (auto& s : static_cast<Program*>(prog)->steps) ApplyStep(s, *b);
        for (auto& c : b->cols) c = Materialize(c, b->rows);
        return b.release();
    } catch (const std::exception& e) {
        if (err) *err = DupStr(e.what());
        return nullptr;
    }
}

// TProgramStep::BuildFilter (program.cpp:921-946) of step `step`: one byte per row.
int orc_build_filter(void* prog, int step, const void* batch, uint8_t* mask, char** err) {
    try {
        Batch b(*static_cast<const Batch*>(batch));
        std::vector<uint8_t> m;
        auto& steps = static_cast<Program*>(prog)->steps;
        if (step < 0 || size_t(step) >= steps.size()) throw Error("no such step");
        ApplyStep(steps[size_t(step)], b, &m);
        std::memcpy(mask, m.data(), m.size());
        return 0;
    } catch (const std::exception& e) {
        if (err) *err = DupStr(e.what());
        return 1;
    }
}

int64_t orc_batch_rows(const void* b) { return static_cast<const Batch*>(b)->rows; }
int orc_batch_ncols(const void* b) { return int(static_cast<const Batch*>(b)->cols.size()); }
const char* orc_batch_col_name(const void* b, int i) { return static_cast<const Batch*>(b)->names[size_t(i)].c_str(); }
int orc_batch_col_type(const void* b, int i) { return static_cast<const Batch*>(b)->cols[size_t(i)]->ty; }
int orc_batch_col_has_nulls(const void* b, int i) { return !static_cast<const Batch*>(b)->cols[size_t(i)]->ok.empty(); }
int64_t orc_batch_col_chars(const void* b, int i) {
    int64_t n = 0;
    for (auto& s : static_cast<const Batch*>(b)->cols[size_t(i)]->s) n += int64_t(s.size());
    return n;
}
// Copies column i out in its natural width (strings: offsets + chars); valid: one byte per row.
void orc_batch_col_read(const void* b, int i, void* values, uint8_t* valid, int32_t* offsets, uint8_t* chars) {
    const Datum& d = *static_cast<const Batch*>(b)->cols[size_t(i)];
    const int64_t n = d.n;
    for (int64_t r = 0; r < n; ++r) {
        if (valid) valid[r] = d.ok.empty() ? 1 : d.ok[r];
        switch (d.ty) {
            case BOOL: case U8: static_cast<uint8_t*>(values)[r] = uint8_t(d.u[r]); break;
            case I8: static_cast<int8_t*>(values)[r] = int8_t(d.i[r]); break;
            case I16: static_cast<int16_t*>(values)[r] = int16_t(d.i[r]); break;
            case I32: static_cast<int32_t*>(values)[r] = int32_t(d.i[r]); break;
            case I64: case TS: case DUR: static_cast<int64_t*>(values)[r] = d.i[r]; break;
            case U16: static_cast<uint16_t*>(values)[r] = uint16_t(d.u[r]); break;
            case U32: static_cast<uint32_t*>(values)[r] = uint32_t(d.u[r]); break;
            case U64: static_cast<uint64_t*>(values)[r] = d.u[r]; break;
            case F32: static_cast<float*>(values)[r] = float(d.f[r]); break;
            case F64: static_cast<double*>(values)[r] = d.f[r]; break;
            case DEC: static_cast<uint64_t*>(values)[2 * r] = d.u[r]; static_cast<uint64_t*>(values)[2 * r + 1] = uint64_t(d.hi[r]); break;
            default: break;
        }
    }
    if (IsStr(d.ty)) {
        int32_t pos = 0;
        for (int64_t r = 0; r < n; ++r) {
            offsets[r] = pos;
            std::memcpy(chars + pos, d.s[r].data(), d.s[r].size());
            pos += int32_t(d.s[r].size());
        }
        offsets[n] = pos;
    }
}

}  // extern "C"
