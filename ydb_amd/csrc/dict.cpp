// Program steps over dictionary-encoded columns.
//
// The reference stores dictionary-encoded columns and decodes them when a chunk is loaded
// (ydb/core/formats/arrow/transformer/dictionary.cpp:9-11, dictionary/conversion.h:13-15): its executor only ever sees
// the plain column.  The contract here is "the result of the decoded column"; the work is done on the codes where that
// is the same thing:
//   * a function of the column and constants (a LIKE, a compare with a literal, a length) is evaluated once per
//     dictionary entry -- by this same executor, over the dictionary as a batch -- and the step reads the result through
//     the row's code (OP_LOOKUP);
//   * GROUP BY the column groups by code.  Two codes with equal bytes are one group (the step then keys on the smallest
//     code of the string, read through a lookup table), unused entries never appear, and a null row groups with the
//     empty string, as the reference's `serialized` method does with a decoded null slot;
//   * COUNT(column) and validity tests need the codes' validity only.
// Any other use (the column as an aggregate's value, in a projection, next to another column in one function) decodes
// the column on the device first -- a gather through the codes -- and runs the step as written.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <set>
#include <unordered_map>

#include "exec.h"

namespace ssa {

std::atomic<int64_t> gDictStepsOnCodes{0};     // steps / filters that ran over codes
std::atomic<int64_t> gDictColumnsDecoded{0};   // whole columns decoded through a gather

#define SSA_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
        }                                                                                              \
    } while (0)

#define SSA_RETURN_NOT_OK(expr)     \
    do {                            \
        Status s_ = (expr);         \
        if (!s_.ok()) return s_;    \
    } while (0)

// Equal entries, the empty entry, and the dictionary extended by one null row.
Status Executor::AnalyseDictionary(DictInfo& d) {
    std::lock_guard<std::mutex> g(d.Lock);
    if (d.Analysed) return Status::OK();
    const int64_t m = d.Entries.Length;
    std::vector<int32_t> offs(size_t(m) + 1, 0);
    if (m > 0) SSA_HIP(hipMemcpyAsync(offs.data(), d.Entries.ValuesPtr(), size_t(m + 1) * 4, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    const int32_t first = offs[0], last = offs[size_t(m)];
    if (last < first || (d.Entries.Chars && size_t(last) > d.Entries.Chars->Size)) return Status::Invalid("dictionary offsets run past its data");
    std::string chars(size_t(last - first), '\0');
    if (last > first) {
        SSA_HIP(hipMemcpyAsync(&chars[0], static_cast<const char*>(d.Entries.Chars->Ptr) + first, chars.size(), hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    d.Canon.assign(size_t(m), 0);
    d.HasDuplicates = false;
    d.EmptyCode = -1;
    std::unordered_map<std::string, int32_t> seen;
    for (int64_t e = 0; e < m; ++e) {
        if (offs[size_t(e) + 1] < offs[size_t(e)]) return Status::Invalid("dictionary offsets decrease");
        const std::string v = chars.substr(size_t(offs[size_t(e)] - first), size_t(offs[size_t(e) + 1] - offs[size_t(e)]));
        auto it = seen.emplace(v, int32_t(e));
        d.Canon[size_t(e)] = it.first->second;
        if (!it.second) d.HasDuplicates = true;
        if (v.empty() && d.EmptyCode < 0) d.EmptyCode = e;
    }
    // entries + one null row, as a batch of one column (named per use)
    auto b = std::make_shared<Batch>();
    b->Rows = m + 1;
    Column col;
    col.Field = d.Entries.Field;
    col.Field.Nullable = true;
    col.Length = m + 1;
    col.NullCount = 1;
    std::vector<int32_t> ext(size_t(m) + 2);
    for (int64_t e = 0; e <= m; ++e) ext[size_t(e)] = offs[size_t(e)];
    ext[size_t(m) + 1] = offs[size_t(m)];
    std::vector<uint8_t> bits(size_t((m + 1 + 7) / 8) + 16, 0);
    for (int64_t e = 0; e < m; ++e) bits[size_t(e >> 3)] |= uint8_t(1u << (e & 7));
    col.Values = Buffer::Alloc(ext.size() * 4 + 16);
    col.Validity = Buffer::Alloc(bits.size());
    if (!col.Values || !col.Validity) return Status::Hip("out of device memory");
    SSA_HIP(hipMemcpyAsync(col.Values->Ptr, ext.data(), ext.size() * 4, hipMemcpyHostToDevice, StreamHandle));
    SSA_HIP(hipMemcpyAsync(col.Validity->Ptr, bits.data(), bits.size(), hipMemcpyHostToDevice, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    col.Chars = d.Entries.Chars;
    if (!col.Chars) {
        col.Chars = Buffer::Alloc(32);
        if (!col.Chars) return Status::Hip("out of device memory");
    }
    b->Cols.push_back(col);
    d.EntriesAndNull = b;
    d.Analysed = true;
    return Status::OK();
}

// The plain column a dictionary column stands for: a gather of the entries through the codes; a null (or out-of-range)
// code is an empty string under the codes' validity.
Status Executor::DecodeColumn(const Column& src, int64_t rows, Column& dst) {
    if (!src.Dict) {
        dst = src;
        return Status::OK();
    }
    gDictColumnsDecoded.fetch_add(1);
    BufferPtr idx = Buffer::Alloc(size_t(rows) * 8 + 16);
    if (!idx) return Status::Hip("out of device memory");
    SSA_HIP(LaunchWidenCodes(src.ValuesPtr(), src.Dict->IndexType, rows, uint64_t(src.Dict->Entries.Length),
                             src.Validity ? static_cast<const uint8_t*>(src.Validity->Ptr) : nullptr, src.ValidBitOff,
                             static_cast<uint64_t*>(idx->Ptr), StreamHandle));
    Column plain;
    SSA_RETURN_NOT_OK(GatherRows(src.Dict->Entries, idx, rows, false, plain));
    plain.Field = src.Field;
    plain.Validity = src.Validity;
    plain.ValidBitOff = src.ValidBitOff;
    plain.NullCount = src.NullCount;
    plain.Length = rows;
    dst = plain;
    return Status::OK();
}

// Group keys that came out as codes -> the strings.  A code past the entries (the stand-in for "null / empty string
// that is no entry") is the empty string.
Status Executor::DecodeCodes(const Column& codes, DictInfo& d, Column& dst) {
    const int64_t n = codes.Length;
    BufferPtr idx = Buffer::Alloc(size_t(n) * 8 + 16);
    if (!idx) return Status::Hip("out of device memory");
    SSA_HIP(LaunchWidenCodes(codes.ValuesPtr(), codes.Field.Type, n, uint64_t(d.Entries.Length), nullptr, 0, static_cast<uint64_t*>(idx->Ptr), StreamHandle));
    SSA_RETURN_NOT_OK(GatherRows(d.Entries, idx, n, false, dst));
    dst.Field = d.Entries.Field;
    dst.Field.Nullable = false;
    dst.Length = n;
    return Status::OK();
}

Status Executor::DecodedBatch(const BatchPtr& in, BatchPtr& out) {
    bool any = false;
    for (const Column& c : in->Cols) any = any || bool(c.Dict);
    if (!any) {
        out = in;
        return Status::OK();
    }
    (void)hipSetDevice(Device);
    auto b = std::make_shared<Batch>(*in);
    for (Column& c : b->Cols) {
        if (!c.Dict) continue;
        Column plain;
        SSA_RETURN_NOT_OK(DecodeColumn(c, in->Rows, plain));
        c = plain;
    }
    out = b;
    return Status::OK();
}

// The results of `fn` -- an assign of `step` whose only non-constant argument is the dictionary column `column` -- for
// every entry of the dictionary and for a null: computed by running the assign over the dictionary as a batch, kept with
// the dictionary for the next batch that carries it.
Status Executor::EntryTable(DictInfo& d, const std::string& column, const TProgramStep& step, const TAssign& fn, DictInfo::Table& out) {
    TProgramStep mini;
    std::string print = fn.DebugString();
    for (const TAssign& a : step.Assignes) {
        if (!a.IsConstant()) continue;
        bool used = false;
        for (auto& arg : fn.Arguments) used = used || arg.ColumnName == a.GetName();
        if (!used) continue;
        mini.Assignes.push_back(a);
        print += "|" + a.GetName() + "=" + a.Constant->ToString() + ":" + std::to_string(a.Constant->Type);
    }
    {
        std::lock_guard<std::mutex> g(d.Lock);
        auto it = d.Tables.find(print);
        if (it != d.Tables.end()) {
            out = it->second;
            return Status::OK();
        }
    }
    mini.Assignes.push_back(fn);
    mini.Projection.push_back(fn.Column);
    auto batch = std::make_shared<Batch>(*d.EntriesAndNull);
    batch->Cols[0].Field.Name = column;
    BatchPtr res;
    SSA_RETURN_NOT_OK(ApplyStepPlain(mini, batch, res));
    if (res->Cols.size() != 1 || res->Rows != batch->Rows) return Status::Unsupported("per-entry evaluation changed the row count");
    const Column& r = res->Cols[0];
    if (IsStringLike(r.Field.Type) || r.Field.Type == T_DEC128) return Status::Unsupported("per-entry function with a string result");
    DictInfo::Table t;
    t.Type = r.Field.Type;
    t.Nullable = r.Validity != nullptr;
    t.Payload = Buffer::Alloc(size_t(res->Rows) * 8 + 16);
    t.Valid = Buffer::Alloc(size_t(res->Rows) + 16);
    if (!t.Payload || !t.Valid) return Status::Hip("out of device memory");
    SSA_HIP(LaunchColumnToPayload(r.Field.Type == T_BOOL ? r.Values->Ptr : r.ValuesPtr(), r.Field.Type, r.ValBitOff,
                                  r.Validity ? static_cast<const uint8_t*>(r.Validity->Ptr) : nullptr, r.ValidBitOff, res->Rows,
                                  static_cast<uint64_t*>(t.Payload->Ptr), static_cast<uint8_t*>(t.Valid->Ptr), StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    {
        std::lock_guard<std::mutex> g(d.Lock);
        d.Tables[print] = t;
    }
    out = t;
    return Status::OK();
}

// A step rewritten for the plain executor: dictionary columns as their codes, per-entry functions as lookups.
struct DictLowering {
    struct KeyFix {
        std::string OutName, Name;  // the key's column in the step's output; the name it has to leave with
        std::shared_ptr<DictInfo> Dict;
    };
    TProgramStep Low;
    BatchPtr Lowered;
    std::vector<LookupTable> Tables;
    std::vector<DictInfo::Table> KeepTables;  // owners of the tables' memory
    std::vector<KeyFix> Fixes;
    bool Native = false;  // something is left to gain over decoding every dictionary column
};

Status Executor::LowerDictStep(const TProgramStep& stepIn, const BatchPtr& in, bool filterOnly, DictLowering& L) {
    TProgramStep stripped;
    if (filterOnly) {  // BuildFilter looks at the assigns and the filters only
        stripped.Assignes = stepIn.Assignes;
        stripped.Filters = stepIn.Filters;
    }
    const TProgramStep& step = filterOnly ? stripped : stepIn;
    std::map<std::string, size_t> dictCols;
    for (size_t i = 0; i < in->Cols.size(); ++i) {
        if (in->Cols[i].Dict && in->FindColumn(in->Cols[i].Field.Name) == int(i)) dictCols[in->Cols[i].Field.Name] = i;
    }
    std::set<std::string> constNames, assigned;
    for (const TAssign& a : step.Assignes) {
        assigned.insert(a.GetName());
        if (a.IsConstant()) constNames.insert(a.GetName());
    }
    auto isDict = [&](const std::string& name) { return dictCols.count(name) != 0 && assigned.count(name) == 0; };

    // how the step uses each dictionary column
    std::set<std::string> decode;
    std::vector<std::pair<size_t, std::string>> fns;  // (assign index, its dictionary column)
    for (size_t ai = 0; ai < step.Assignes.size(); ++ai) {
        const TAssign& a = step.Assignes[ai];
        if (a.IsConstant()) continue;
        std::set<std::string> dictArgs;
        bool othersConstant = true;
        for (auto& arg : a.Arguments) {
            if (isDict(arg.ColumnName)) dictArgs.insert(arg.ColumnName);
            else if (!constNames.count(arg.ColumnName)) othersConstant = false;
        }
        if (dictArgs.empty()) continue;
        if (dictArgs.size() == 1 && othersConstant) fns.emplace_back(ai, *dictArgs.begin());
        else decode.insert(dictArgs.begin(), dictArgs.end());
    }
    for (const TAggregateAssign& g : step.GroupBy) {
        for (auto& arg : g.Arguments) {
            if (isDict(arg.ColumnName) && g.Operation != EAggregate::Count) decode.insert(arg.ColumnName);
        }
    }
    for (auto& f : step.Filters) {
        if (isDict(f.ColumnName)) decode.insert(f.ColumnName);
    }
    if (step.GroupBy.empty() && !filterOnly) {  // the step's output carries columns themselves
        for (auto& p : step.Projection) {
            if (isDict(p.ColumnName)) decode.insert(p.ColumnName);
        }
        if (step.Projection.empty()) {
            for (auto& kv : dictCols) decode.insert(kv.first);
        }
    }
    std::vector<size_t> keyCols;  // indices into GroupByKeys
    bool plainStringKey = false;
    for (size_t k = 0; k < step.GroupByKeys.size(); ++k) {
        const std::string& name = step.GroupByKeys[k].ColumnName;
        if (isDict(name)) {
            keyCols.push_back(k);
        } else {
            const int ci = in->FindColumn(name);
            if (ci >= 0 && IsStringLike(in->Cols[size_t(ci)].Field.Type) && assigned.count(name) == 0) plainStringKey = true;
        }
    }

    // the batch and the step as the plain executor sees them
    using KeyFix = DictLowering::KeyFix;
    auto lowered = std::make_shared<Batch>(*in);
    L.Low = step;
    TProgramStep& low = L.Low;
    std::vector<LookupTable>& tables = L.Tables;
    std::vector<DictInfo::Table>& keepTables = L.KeepTables;
    std::vector<KeyFix>& fixes = L.Fixes;
    bool& native = L.Native;
    native = false;
    Status lowering = Status::OK();
    for (auto& kv : dictCols) {
        Column& col = lowered->Cols[kv.second];
        if (decode.count(kv.first)) continue;
        lowering = AnalyseDictionary(*col.Dict);
        if (!lowering.ok()) break;
    }
    if (lowering.ok()) {
        for (auto& f : fns) {
            if (decode.count(f.second)) continue;
            if (tables.size() >= size_t(kMaxLookupTables)) {
                decode.insert(f.second);
                continue;
            }
            DictInfo& d = *in->Cols[dictCols[f.second]].Dict;
            DictInfo::Table t;
            Status st = EntryTable(d, f.second, step, step.Assignes[f.first], t);
            if (st.Code == ST_HIP_ERROR) return st;
            if (!st.ok()) {  // not a function this executor runs, or one that fails on some entry (which no row may use)
                decode.insert(f.second);
                continue;
            }
            TAssign& a = low.Assignes[f.first];
            a.DictTable = int(tables.size());
            a.DictResultType = t.Type;
            a.DictResultNullable = t.Nullable;
            a.Arguments.clear();
            a.Arguments.push_back(TColumnInfo::Original(0, f.second));
            tables.push_back(LookupTable{static_cast<const uint64_t*>(t.Payload->Ptr), static_cast<const uint8_t*>(t.Valid->Ptr), uint64_t(d.Entries.Length)});
            keepTables.push_back(t);
            native = true;
        }
        // (a function dropped to "decode" above may have left earlier functions of the same column rewritten: undo them)
        for (auto& f : fns) {
            if (decode.count(f.second) && low.Assignes[f.first].DictTable >= 0) low.Assignes[f.first] = step.Assignes[f.first];
        }
        for (size_t k : keyCols) {
            const std::string name = step.GroupByKeys[k].ColumnName;
            if (decode.count(name)) continue;
            const Column& src = in->Cols[dictCols[name]];
            DictInfo& d = *src.Dict;
            if (plainStringKey && (d.HasDuplicates || src.Validity)) {  // a computed key next to a hashed one: not in the plan compiler
                decode.insert(name);
                continue;
            }
            if (!d.HasDuplicates && !src.Validity) {
                fixes.push_back(KeyFix{name, name, src.Dict});  // the codes themselves are the key
                native = true;
                continue;
            }
            if (tables.size() >= size_t(kMaxLookupTables)) {
                decode.insert(name);
                continue;
            }
            // code -> the smallest code of the same string; a null row -> the empty string's code (or one past the entries)
            const int64_t m = d.Entries.Length;
            std::vector<uint64_t> payload(size_t(m) + 1);
            for (int64_t e = 0; e < m; ++e) payload[size_t(e)] = uint64_t(d.Canon[size_t(e)]);
            payload[size_t(m)] = uint64_t(d.EmptyCode >= 0 ? d.EmptyCode : m);
            DictInfo::Table t;
            t.Type = T_I32;
            t.Payload = Buffer::Alloc(payload.size() * 8 + 16);
            t.Valid = Buffer::Alloc(payload.size() + 16);
            if (!t.Payload || !t.Valid) return Status::Hip("out of device memory");
            SSA_HIP(hipMemcpyAsync(t.Payload->Ptr, payload.data(), payload.size() * 8, hipMemcpyHostToDevice, StreamHandle));
            SSA_HIP(hipMemsetAsync(t.Valid->Ptr, 1, payload.size(), StreamHandle));
            SSA_HIP(hipStreamSynchronize(StreamHandle));
            const std::string shadow = std::string("\x01key:") + name;
            TAssign a;
            a.Column = TColumnInfo::Generated(0, shadow);
            a.Operation = EOperation::CastInt32;  // (any operation: DictTable decides)
            a.Arguments.push_back(TColumnInfo::Original(0, name));
            a.DictTable = int(tables.size());
            a.DictResultType = T_I32;
            a.DictResultNullable = false;
            low.Assignes.push_back(a);
            low.GroupByKeys[k].ColumnName = shadow;
            for (auto& p : low.Projection) {
                if (p.ColumnName == name) p.ColumnName = shadow;
            }
            tables.push_back(LookupTable{static_cast<const uint64_t*>(t.Payload->Ptr), static_cast<const uint8_t*>(t.Valid->Ptr), uint64_t(m)});
            keepTables.push_back(t);
            fixes.push_back(KeyFix{shadow, name, src.Dict});
            native = true;
        }
    }

    if (!lowering.ok()) native = false;
    if (!native) return Status::OK();

    // keys decoded where the loop above gave up, codes in place of every other dictionary column
    for (auto& kv : dictCols) {
        Column& col = lowered->Cols[kv.second];
        if (decode.count(kv.first)) {
            Column plain;
            SSA_RETURN_NOT_OK(DecodeColumn(in->Cols[kv.second], in->Rows, plain));
            col = plain;
            continue;
        }
        col.Field.Type = col.Dict->IndexType;
        col.Field.Precision = col.Field.Scale = 0;
        col.Dict = nullptr;
        col.Chars = nullptr;
    }
    for (size_t i = 0; i < fixes.size();) {
        if (decode.count(fixes[i].Name)) fixes.erase(fixes.begin() + long(i)); else ++i;
    }
    L.Lowered = lowered;
    return Status::OK();
}

Status Executor::ApplyStepDict(const TProgramStep& step, const BatchPtr& in, BatchPtr& out) {
    (void)hipSetDevice(Device);
    auto runDecoded = [&]() -> Status {
        BatchPtr plain;
        SSA_RETURN_NOT_OK(DecodedBatch(in, plain));
        return ApplyStepPlain(step, plain, out);
    };
    DictLowering L;
    SSA_RETURN_NOT_OK(LowerDictStep(step, in, false, L));
    if (!L.Native) return runDecoded();

    StepTables = L.Tables;
    ForceSerializedKeys = !L.Fixes.empty();
    HintOverride = &step;
    BatchPtr res;
    Status st = ApplyStepPlain(L.Low, L.Lowered, res);
    StepTables.clear();
    ForceSerializedKeys = false;
    HintOverride = nullptr;
    if (st.Code == ST_UNSUPPORTED) return runDecoded();
    SSA_RETURN_NOT_OK(st);
    gDictStepsOnCodes.fetch_add(1);
    for (Column& col : res->Cols) {
        for (const DictLowering::KeyFix& f : L.Fixes) {
            if (col.Field.Name != f.OutName) continue;
            Column strings;
            SSA_RETURN_NOT_OK(DecodeCodes(col, *f.Dict, strings));
            strings.Field.Name = f.Name;
            col = strings;
            break;
        }
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = res;
    return Status::OK();
}

// TProgramStep::BuildFilter over a batch with dictionary columns: predicates on them are read through lookup tables.
Status Executor::FilterFromStepDict(const TProgramStep& step, const BatchPtr& in, FilterPtr& out) {
    (void)hipSetDevice(Device);
    DictLowering L;
    SSA_RETURN_NOT_OK(LowerDictStep(step, in, true, L));
    Status st = Status::Unsupported("");
    if (L.Native) {
        StepTables = L.Tables;
        st = FilterFromStepPlain(L.Low, L.Lowered, out);
        StepTables.clear();
        if (st.ok()) gDictStepsOnCodes.fetch_add(1);
    }
    if (st.Code != ST_UNSUPPORTED) return st;
    BatchPtr plain;
    SSA_RETURN_NOT_OK(DecodedBatch(in, plain));
    return FilterFromStepPlain(step, plain, out);
}

}  // namespace ssa
