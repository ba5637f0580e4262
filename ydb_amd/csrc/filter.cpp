// TColumnFilter on the device: construction, run-length form, And / Or / CombineSequentialAnd, MakePredicateFilter,
// GetFilteredCount and Apply (ydb/core/formats/arrow/arrow_filter.h:20-211, arrow_filter.cpp:232-616 of the reference;
// callers engines/reader/plain_reader/iterator/fetching.cpp:74-109, fetched_data.h:70-80, engines/predicate/container.h:74).
// The reference stores run lengths; here the filter is an HBM bitmap and every operation is a streaming pass over
// 64-row words.  The observable state is kept identical: Filter.empty() + DefaultFilterValue for filters without a
// size, else Count rows whose maximal runs of equal bits are exactly the reference's `Filter` vector.
#include <hip/hip_runtime.h>

#include <cstring>

#include "exec.h"

namespace ssa {

#define SSA_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { \
            (void)hipGetLastError(); /* the runtime keeps the error for the next caller otherwise */   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));        \
        } \
    } while (0)
#define SSA_RETURN_NOT_OK(expr) \
    do {                        \
        Status s_ = (expr);     \
        if (!s_.ok()) return s_; \
    } while (0)

namespace {

constexpr int64_t kMaxFilterRows = (int64_t(1) << 31) - 1;  // the reference counts rows in ui32

int64_t WordsOf(int64_t rows) { return (rows + 63) >> 6; }

// BuildAllowFilter / BuildDenyFilter (arrow_filter.h:188-194)
FilterPtr SimpleFilter(bool value) {
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = true;
    f->DefaultValue = value;
    return f;
}

FilterPtr SameFilter(const DeviceFilter& f) { return std::make_shared<DeviceFilter>(f); }  // the bits are immutable: shared

// Scratch of LaunchTileScan over `n` counters: n + 1 offsets followed by the block sums.
size_t TileScanBytes(int64_t n) { return size_t(n + 1 + (n + 4095) / 4096 + 2) * 8 + 64; }

}  // namespace

Status Executor::FilterFromRuns(bool firstValue, const std::vector<uint32_t>& runs, FilterPtr& out) {
    (void)hipSetDevice(Device);
    // TColumnFilter::Add (arrow_filter.cpp:239-250): empty runs vanish, equal neighbours join
    std::vector<uint32_t> starts;
    bool first = firstValue, last = firstValue, have = false;
    uint64_t pos = 0;
    for (size_t j = 0; j < runs.size(); ++j) {
        const bool v = firstValue ^ bool(j & 1);
        if (runs[j] == 0) continue;
        if (!have || v != last) {
            starts.push_back(uint32_t(pos));
            if (!have) first = v;
            have = true;
            last = v;
        }
        pos += runs[j];
        if (pos > uint64_t(kMaxFilterRows)) return Status::Unsupported("a filter of 2^31 rows or more");
    }
    if (!have) {  // TColumnFilter(std::vector<bool>{}) keeps Filter empty with the default of the constructor
        out = SimpleFilter(true);
        return Status::OK();
    }
    starts.push_back(uint32_t(pos));
    const int64_t nruns = int64_t(starts.size()) - 1;
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = false;
    f->DefaultValue = true;
    f->Count = int64_t(pos);
    f->Bits = Buffer::Alloc(size_t(WordsOf(f->Count)) * 8 + 64);
    BufferPtr dev = Buffer::Alloc(starts.size() * 4 + 16);
    if (!f->Bits || !dev) return Status::Hip("out of device memory");
    SSA_HIP(hipMemcpyAsync(dev->Ptr, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, StreamHandle));
    SSA_HIP(LaunchExpandRuns(static_cast<const uint32_t*>(dev->Ptr), nruns, first ? 1 : 0, f->Count, static_cast<uint64_t*>(f->Bits->Ptr),
                             StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));  // `starts` is read by the copy above
    out = f;
    return Status::OK();
}

Status Executor::FilterToRuns(const DeviceFilter& f, std::vector<uint32_t>& runs, bool& lastValue) {
    (void)hipSetDevice(Device);
    runs.clear();
    lastValue = true;  // LastValue of a filter nothing was added to (arrow_filter.h:23)
    if (f.Empty) return Status::OK();
    const int64_t nwords = WordsOf(f.Count);
    BufferPtr counts = Buffer::Alloc(size_t(nwords) * 4 + 64);
    BufferPtr offsets = Buffer::Alloc(TileScanBytes(nwords));
    BufferPtr total = Buffer::Alloc(256);
    if (!counts || !offsets || !total) return Status::Hip("out of device memory");
    const uint64_t* bits = static_cast<const uint64_t*>(f.Bits->Ptr);
    SSA_HIP(LaunchTransitionCount(bits, f.Count, static_cast<uint32_t*>(counts->Ptr), StreamHandle));
    SSA_HIP(LaunchTileScan(static_cast<const uint32_t*>(counts->Ptr), nwords, static_cast<int64_t*>(offsets->Ptr),
                           static_cast<int64_t*>(total->Ptr), StreamHandle));
    int64_t ntrans = 0;
    uint64_t word0 = 0;
    SSA_HIP(hipMemcpyAsync(&ntrans, total->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipMemcpyAsync(&word0, f.Bits->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    std::vector<uint32_t> starts(size_t(ntrans) + 2, 0);
    if (ntrans > 0) {
        BufferPtr dev = Buffer::Alloc(size_t(ntrans) * 4 + 16);
        if (!dev) return Status::Hip("out of device memory");
        SSA_HIP(LaunchTransitionScatter(bits, f.Count, static_cast<const int64_t*>(offsets->Ptr), static_cast<uint32_t*>(dev->Ptr),
                                        StreamHandle));
        SSA_HIP(hipMemcpyAsync(starts.data() + 1, dev->Ptr, size_t(ntrans) * 4, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    starts[size_t(ntrans) + 1] = uint32_t(f.Count);
    runs.resize(size_t(ntrans) + 1);
    for (size_t j = 0; j < runs.size(); ++j) runs[j] = starts[j + 1] - starts[j];
    lastValue = bool(word0 & 1ull) ^ bool(ntrans & 1);
    return Status::OK();
}

Status Executor::FilterFromStep(const TProgramStep& step, const BatchPtr& in, FilterPtr& out) {
    for (const Column& c : in->Cols) {
        if (c.Dict) return FilterFromStepDict(step, in, out);
    }
    return FilterFromStepPlain(step, in, out);
}

Status Executor::FilterFromStepPlain(const TProgramStep& step, const BatchPtr& in, FilterPtr& out) {
    (void)hipSetDevice(Device);
    if (in->Rows > kMaxFilterRows) return Status::Unsupported("a filter of 2^31 rows or more");
    StepPlan plan;
    SSA_RETURN_NOT_OK(CompileStep(step, in->Schema(), CM_FILTER_ONLY, plan));
    if (in->Rows == 0) {  // fullLocal.Append(local) of a filter without rows adds nothing (program.cpp:930-943)
        out = SimpleFilter(true);
        return Status::OK();
    }
    BufferPtr bitmap, tileOffsets;
    int64_t selected = 0;
    SSA_RETURN_NOT_OK(RunFilterBitmap(plan, *in, bitmap, tileOffsets, selected));
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = false;
    f->DefaultValue = true;
    f->Count = in->Rows;
    f->Bits = bitmap;
    out = f;
    return Status::OK();
}

// TMergerImpl::Merge (arrow_filter.cpp:444-501) with TMergePolicyAnd / TMergePolicyOr (:402-430).
Status Executor::FilterAndOr(const DeviceFilter& a, const DeviceFilter& b, bool isOr, FilterPtr& out) {
    (void)hipSetDevice(Device);
    if (a.Empty && b.Empty) {
        out = SimpleFilter(isOr ? (a.DefaultValue || b.DefaultValue) : (a.DefaultValue && b.DefaultValue));
        return Status::OK();
    }
    if (a.Empty || b.Empty) {
        const DeviceFilter& sized = a.Empty ? b : a;
        const bool simple = a.Empty ? a.DefaultValue : b.DefaultValue;
        if (isOr) out = simple ? SimpleFilter(true) : SameFilter(sized);
        else out = simple ? SameFilter(sized) : SimpleFilter(false);
        return Status::OK();
    }
    if (a.Count != b.Count) return Status::Invalid("filters of different sizes cannot be merged");  // Y_ABORT_UNLESS in the reference
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = false;
    f->DefaultValue = true;
    f->Count = a.Count;
    const int64_t nwords = WordsOf(a.Count);
    f->Bits = Buffer::Alloc(size_t(nwords) * 8 + 64);
    if (!f->Bits) return Status::Hip("out of device memory");
    SSA_HIP(LaunchBitmapAndOr(static_cast<const uint64_t*>(a.Bits->Ptr), static_cast<const uint64_t*>(b.Bits->Ptr), nwords, isOr ? 1 : 0,
                              static_cast<uint64_t*>(f->Bits->Ptr), StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = f;
    return Status::OK();
}

// arrow_filter.cpp:515-575: `ext` has one row per row that `self` keeps.
Status Executor::FilterCombineSequentialAnd(const DeviceFilter& self, const DeviceFilter& ext, FilterPtr& out) {
    (void)hipSetDevice(Device);
    if (self.Empty) {
        out = self.DefaultValue ? SameFilter(ext) : SimpleFilter(false);
        return Status::OK();
    }
    if (ext.Empty) {
        out = ext.DefaultValue ? SameFilter(self) : SimpleFilter(false);
        return Status::OK();
    }
    const int64_t nwords = WordsOf(self.Count);
    BufferPtr counts = Buffer::Alloc(size_t(nwords) * 4 + 64);
    BufferPtr ranks = Buffer::Alloc(TileScanBytes(nwords));
    BufferPtr total = Buffer::Alloc(256);
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = false;
    f->DefaultValue = true;
    f->Count = self.Count;
    f->Bits = Buffer::Alloc(size_t(nwords) * 8 + 64);
    if (!counts || !ranks || !total || !f->Bits) return Status::Hip("out of device memory");
    const uint64_t* bits = static_cast<const uint64_t*>(self.Bits->Ptr);
    SSA_HIP(LaunchWordPopcount(bits, nwords, static_cast<uint32_t*>(counts->Ptr), StreamHandle));
    SSA_HIP(LaunchTileScan(static_cast<const uint32_t*>(counts->Ptr), nwords, static_cast<int64_t*>(ranks->Ptr),
                           static_cast<int64_t*>(total->Ptr), StreamHandle));
    int64_t kept = 0;
    SSA_HIP(hipMemcpyAsync(&kept, total->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    // the reference walks both run lists to their ends and aborts otherwise (:535, :568)
    if (kept != ext.Count) return Status::Invalid("the second filter must have one row per row the first filter keeps");
    SSA_HIP(LaunchSequentialAnd(bits, nwords, static_cast<const int64_t*>(ranks->Ptr), static_cast<const uint64_t*>(ext.Bits->Ptr),
                                WordsOf(ext.Count), static_cast<uint64_t*>(f->Bits->Ptr), StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = f;
    return Status::OK();
}

// MakePredicateFilter over a record batch and a one-row border batch (arrow_filter.cpp:118-136, 259-311).
Status Executor::FilterPredicate(const Batch& in, const std::vector<BorderValue>& border, int compareType, FilterPtr& out) {
    (void)hipSetDevice(Device);
    if (compareType < 1 || compareType > 4) return Status::Invalid("compare type must be 1..4 (LESS, LESS_OR_EQUAL, GREATER, GREATER_OR_EQUAL)");
    if (border.empty()) return Status::Invalid("the border has no key columns");  // Y_ABORT_UNLESS(key.size())
    if (border.size() > size_t(kMaxShardKeys)) return Status::Unsupported("more than 8 border key columns");
    if (in.Rows > kMaxFilterRows) return Status::Unsupported("a filter of 2^31 rows or more");
    PredicateKeys keys{};
    std::vector<BufferPtr> keep;
    for (const BorderValue& b : border) {
        const int ci = in.FindColumn(b.Name);
        if (ci < 0) return Status::Invalid("border key column '" + b.Name + "' is not in the batch");
        const Column& col = in.Cols[size_t(ci)];
        const int t = col.Field.Type;
        if (t != b.Value.Type) return Status::Invalid("border key column '" + b.Name + "' differs in type from the batch column");
        PredicateKey& k = keys.cols[keys.nkeys++];
        k.values = col.ValuesPtr();
        k.chars = nullptr;
        k.border_chars = nullptr;
        k.border = b.Value.Bits;
        k.width = TypeWidth(t);
        k.val_bit_off = int32_t(col.ValBitOff);
        if (IsSignedInt(t) || IsTemporal(t)) {
            k.kind = PK_SIGNED;
        } else if (IsUnsignedInt(t)) {
            k.kind = PK_UNSIGNED;
        } else if (t == T_F64) {
            k.kind = PK_F64;
        } else if (t == T_F32) {
            k.kind = PK_F32;
            double d;
            std::memcpy(&d, &b.Value.Bits, 8);
            const float fl = float(d);
            uint32_t fb;
            std::memcpy(&fb, &fl, 4);
            k.border = fb;
        } else if (t == T_BOOL) {
            k.kind = PK_BOOL;
            k.values = col.Values->Ptr;
        } else if (IsStringLike(t)) {
            k.kind = PK_STRING;
            k.border = b.Value.Bytes.size();
            BufferPtr dev = Buffer::Alloc(b.Value.Bytes.size() + 16);
            if (!dev) return Status::Hip("out of device memory");
            if (!b.Value.Bytes.empty()) {
                SSA_HIP(hipMemcpyAsync(dev->Ptr, b.Value.Bytes.data(), b.Value.Bytes.size(), hipMemcpyHostToDevice, StreamHandle));
            }
            k.border_chars = static_cast<const uint8_t*>(dev->Ptr);
            k.chars = col.Chars ? static_cast<const uint8_t*>(col.Chars->Ptr) : static_cast<const uint8_t*>(dev->Ptr);
            keep.push_back(dev);
        } else {
            return Status::Unsupported("border key type");
        }
    }
    if (in.Rows == 0) {  // TColumnFilter(std::vector<bool>{})
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        out = SimpleFilter(true);
        return Status::OK();
    }
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = false;
    f->DefaultValue = true;
    f->Count = in.Rows;
    f->Bits = Buffer::Alloc(size_t(WordsOf(in.Rows)) * 8 + 64);
    if (!f->Bits) return Status::Hip("out of device memory");
    SSA_HIP(LaunchPredicate(keys, in.Rows, compareType, static_cast<uint64_t*>(f->Bits->Ptr), StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = f;
    return Status::OK();
}

// GetFilteredCount (arrow_filter.cpp:586-608)
Status Executor::FilterCountSelected(const DeviceFilter& f, int64_t& selected) {
    (void)hipSetDevice(Device);
    if (f.Empty) {
        selected = f.DefaultValue ? -1 : 0;
        return Status::OK();
    }
    const int64_t nwords = WordsOf(f.Count);
    BufferPtr counts = Buffer::Alloc(size_t(nwords) * 4 + 64);
    BufferPtr ranks = Buffer::Alloc(TileScanBytes(nwords));
    BufferPtr total = Buffer::Alloc(256);
    if (!counts || !ranks || !total) return Status::Hip("out of device memory");
    SSA_HIP(LaunchWordPopcount(static_cast<const uint64_t*>(f.Bits->Ptr), nwords, static_cast<uint32_t*>(counts->Ptr), StreamHandle));
    SSA_HIP(LaunchTileScan(static_cast<const uint32_t*>(counts->Ptr), nwords, static_cast<int64_t*>(ranks->Ptr),
                           static_cast<int64_t*>(total->Ptr), StreamHandle));
    int64_t kept = 0;
    SSA_HIP(hipMemcpyAsync(&kept, total->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    selected = kept;
    return Status::OK();
}

Status Executor::FilterBits(const DeviceFilter& f, std::vector<uint8_t>& bits) {
    (void)hipSetDevice(Device);
    bits.clear();
    if (f.Empty) return Status::OK();  // BuildSimpleFilter aborts without a size: nothing to give
    bits.assign(size_t((f.Count + 7) / 8), 0);
    SSA_HIP(hipMemcpyAsync(bits.data(), f.Bits->Ptr, bits.size(), hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

// ApplyImpl (arrow_filter.cpp:313-336): a batch without rows is left alone; a sized filter must match the batch;
// deny-all gives the empty batch of the same schema, allow-all the batch itself, anything else the kept rows in order.
Status Executor::FilterApply(const DeviceFilter& f, const BatchPtr& in, BatchPtr& out) {
    (void)hipSetDevice(Device);
    if (in->Rows == 0) {
        out = in;
        return Status::OK();
    }
    if (!f.Empty && f.Count != in->Rows) return Status::Invalid("filter size differs from the batch's row count");
    if (f.Empty && f.DefaultValue) {
        out = in;
        return Status::OK();
    }
    const int64_t rows = in->Rows;
    int64_t selected = 0;
    BufferPtr tileOffsets;
    if (!f.Empty) {
        const int64_t nwords = WordsOf(rows);
        const int64_t ntiles = (rows + kTileRows - 1) / kTileRows;
        BufferPtr tileCounts = Buffer::Alloc(size_t(ntiles) * 4 + 64);
        tileOffsets = Buffer::Alloc(TileScanBytes(ntiles));
        BufferPtr total = Buffer::Alloc(256);
        if (!tileCounts || !tileOffsets || !total) return Status::Hip("out of device memory");
        SSA_HIP(LaunchBitmapTileCount(static_cast<const uint64_t*>(f.Bits->Ptr), nwords, ntiles, static_cast<uint32_t*>(tileCounts->Ptr),
                                      StreamHandle));
        SSA_HIP(LaunchTileScan(static_cast<const uint32_t*>(tileCounts->Ptr), ntiles, static_cast<int64_t*>(tileOffsets->Ptr),
                               static_cast<int64_t*>(total->Ptr), StreamHandle));
        SSA_HIP(hipMemcpyAsync(&selected, total->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        if (selected == rows) {  // IsTotalAllowFilter: one run of kept rows
            out = in;
            return Status::OK();
        }
    }
    auto res = std::make_shared<Batch>();
    res->Rows = selected;
    for (const Column& col : in->Cols) {
        Column dst;
        if (selected == 0) {  // GetEmptySame: the schema with no rows
            dst.Field = col.Field;
            dst.Length = 0;
            dst.Values = Buffer::Alloc(64);
            if (!dst.Values) return Status::Hip("out of device memory");
            SSA_HIP(hipMemsetAsync(dst.Values->Ptr, 0, 64, StreamHandle));
            if (IsStringLike(col.Field.Type)) dst.Chars = Buffer::Alloc(16);
        } else {
            SSA_RETURN_NOT_OK(CompactColumn(col, rows, f.Bits, tileOffsets, selected, dst));
        }
        dst.Field.Nullable = dst.Validity != nullptr || col.Field.Nullable;
        res->Cols.push_back(std::move(dst));
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = res;
    return Status::OK();
}

}  // namespace ssa
