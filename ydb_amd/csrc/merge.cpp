// NArrow::NMerger::TMergePartialStream::DrainAll on the device (ydb/core/formats/arrow/reader/merger.cpp:145-186,
// merger.h:78-89): sources sorted by the key columns, merged into one sorted batch in which every key appears once,
// with the row of its newest version -- unless that row is deleted by its source's filter, in which case the key does
// not appear at all (merger.cpp:152-163: only the top record of a key is looked at, the others are skipped).
//
// Between rows of one key and one version the reference's heap does not define an order; here the row of the earlier
// source wins, and inside a source the earlier row.
#include <hip/hip_runtime.h>

#include <cstring>

#include "exec.h"

namespace ssa {

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

namespace {

Status DescribeKey(const Batch& all, const std::string& name, MergeKeyColumn& out) {
    const int ci = all.FindColumn(name);
    if (ci < 0) return Status::Invalid("merge column '" + name + "' is not in the sources");
    const Column& col = all.Cols[size_t(ci)];
    const int t = col.Field.Type;
    out.values = col.ValuesPtr();
    out.chars = col.Chars ? static_cast<const uint8_t*>(col.Chars->Ptr) : nullptr;
    out.width = TypeWidth(t);
    if (IsStringLike(t)) out.kind = PK_STRING;
    else if (t == T_F32) out.kind = PK_F32;
    else if (t == T_F64) out.kind = PK_F64;
    else if (t == T_DEC128) out.kind = PK_DEC128;
    else if (IsSignedInt(t) || IsTemporal(t)) out.kind = PK_SIGNED;
    else if (t == T_BOOL) out.kind = PK_BOOL;  // (never tells two rows apart: compare.h:12-14)
    else out.kind = PK_UNSIGNED;
    if (col.Dict) return Status::Unsupported("merge by a dictionary-encoded column");
    return Status::OK();
}

}  // namespace

Status Executor::MergeSorted(const std::vector<BatchPtr>& sourcesIn, const std::vector<FilterPtr>& filtersIn, const std::vector<std::string>& keyNames,
                             const std::vector<std::string>& versionNames, bool reverse, BatchPtr& out) {
    (void)hipSetDevice(Device);
    if (keyNames.empty()) return Status::Invalid("the sort schema has no columns");  // Y_ABORT_UNLESS(SortSchema->num_fields())
    if (keyNames.size() + versionNames.size() > size_t(kMaxMergeKeys)) return Status::Unsupported("more than 12 key and version columns");
    if (!filtersIn.empty() && filtersIn.size() != sourcesIn.size()) return Status::Invalid("one filter (or none) per source is needed");
    if (sourcesIn.empty()) return Status::Invalid("no sources");
    // AddSource: sources without rows and sources whose filter denies every row take no part
    std::vector<BatchPtr> sources;
    std::vector<FilterPtr> filters;
    bool anyFilter = false;
    for (size_t i = 0; i < sourcesIn.size(); ++i) {
        if (!sourcesIn[i]) return Status::Invalid("null source");
        FilterPtr f = filtersIn.empty() ? nullptr : filtersIn[i];
        if (f && !f->Empty && f->Count != sourcesIn[i]->Rows) return Status::Invalid("filter size differs from its source's row count");
        if (sourcesIn[i]->Rows == 0) continue;
        if (f && f->Empty) {
            if (!f->DefaultValue) continue;  // deny all
            f = nullptr;                     // allow all
        }
        BatchPtr plain;
        SSA_RETURN_NOT_OK(DecodedBatch(sourcesIn[i], plain));
        sources.push_back(plain);
        filters.push_back(f);
        anyFilter = anyFilter || bool(f);
    }
    if (sources.empty()) {  // the schema with no rows
        BatchPtr plain;
        SSA_RETURN_NOT_OK(DecodedBatch(sourcesIn[0], plain));
        auto empty = std::make_shared<Batch>();
        for (const Column& col : plain->Cols) {
            Column dst;
            dst.Field = col.Field;
            dst.Field.Nullable = false;
            dst.Values = Buffer::Alloc(64);
            if (!dst.Values) return Status::Hip("out of device memory");
            SSA_HIP(hipMemsetAsync(dst.Values->Ptr, 0, 64, StreamHandle));
            if (IsStringLike(col.Field.Type)) dst.Chars = Buffer::Alloc(16);
            empty->Cols.push_back(std::move(dst));
        }
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        out = empty;
        return Status::OK();
    }
    BatchPtr all;
    SSA_RETURN_NOT_OK(ConcatBatches(sources, all));
    const int64_t n = all->Rows;
    if (n >= (int64_t(1) << 31)) return Status::Unsupported("a merge of 2^31 rows or more");

    MergeKeys keys{};
    keys.nkeys = int32_t(keyNames.size());
    keys.nversions = int32_t(versionNames.size());
    for (size_t i = 0; i < keyNames.size(); ++i) SSA_RETURN_NOT_OK(DescribeKey(*all, keyNames[i], keys.cols[i]));
    for (size_t i = 0; i < versionNames.size(); ++i) SSA_RETURN_NOT_OK(DescribeKey(*all, versionNames[i], keys.cols[keyNames.size() + i]));

    // deleted rows: one byte per row of the concatenated sources
    BufferPtr keep;
    if (anyFilter) {
        keep = Buffer::Alloc(size_t(n) + 64);
        if (!keep) return Status::Hip("out of device memory");
        std::vector<CopySegment> segs;
        int64_t row = 0;
        for (size_t i = 0; i < sources.size(); ++i) {
            uint8_t* dst = static_cast<uint8_t*>(keep->Ptr) + row;
            if (filters[i]) segs.push_back(CopySegment{static_cast<const uint8_t*>(filters[i]->Bits->Ptr), dst, uint64_t(sources[i]->Rows), SEG_BITS_TO_BYTES, 0});
            else segs.push_back(CopySegment{nullptr, dst, uint64_t(sources[i]->Rows), SEG_FILL, 1});
            row += sources[i]->Rows;
        }
        SSA_RETURN_NOT_OK(RunSegments(segs));
    }

    // The rows a source's iterator shows the heap -- the first of every run of equal keys inside the source -- are the
    // rows that enter the merge; the others are skipped with them whatever their version (merger.cpp:169-184).
    std::vector<uint64_t> runs;  // run r = ids [runs[r], runs[r + 1]); the sources are the first runs
    runs.push_back(0);
    for (const BatchPtr& s : sources) runs.push_back(runs.back() + uint64_t(s->Rows));
    BufferPtr idsA = Buffer::Alloc(size_t(n) * 8 + 64), idsB = Buffer::Alloc(size_t(n) * 8 + 64);
    if (!idsA || !idsB) return Status::Hip("out of device memory");
    SSA_HIP(LaunchIotaU64(static_cast<uint64_t*>(idsA->Ptr), n, StreamHandle));
    int64_t nids = n;
    {
        BufferPtr boundary = Buffer::Alloc(size_t(n) + 64);
        auto candidates = std::make_shared<DeviceFilter>();
        candidates->Empty = false;
        candidates->Count = n;
        candidates->Bits = Buffer::Alloc(size_t((n + 63) / 64) * 8 + 64);
        if (!boundary || !candidates->Bits) return Status::Hip("out of device memory");
        SSA_HIP(hipMemsetAsync(boundary->Ptr, 0, size_t(n), StreamHandle));
        std::vector<CopySegment> marks;
        for (size_t r = 0; r + 1 < runs.size(); ++r) marks.push_back(CopySegment{nullptr, static_cast<uint8_t*>(boundary->Ptr) + runs[r], 1, SEG_FILL, 1});
        SSA_RETURN_NOT_OK(RunSegments(marks));
        SSA_HIP(LaunchMergeCandidates(keys, n, static_cast<const uint8_t*>(boundary->Ptr), reverse, static_cast<uint64_t*>(candidates->Bits->Ptr), StreamHandle));
        auto iota = std::make_shared<Batch>();
        iota->Rows = n;
        Column ids;
        ids.Field.Name = "row";
        ids.Field.Type = T_U64;
        ids.Length = n;
        ids.Values = idsA;
        iota->Cols.push_back(ids);
        BatchPtr kept;
        SSA_RETURN_NOT_OK(FilterApply(*candidates, iota, kept));
        if (kept.get() != iota.get()) {  // some source repeats a key
            nids = kept->Rows;
            idsA = kept->Cols[0].Values;
            BufferPtr bounds = Buffer::Alloc(runs.size() * 8 + 64), counts = Buffer::Alloc(runs.size() * 8 + 64);
            if (!bounds || !counts) return Status::Hip("out of device memory");
            std::vector<uint64_t> below(runs.size());
            SSA_HIP(hipMemcpyAsync(bounds->Ptr, runs.data(), runs.size() * 8, hipMemcpyHostToDevice, StreamHandle));
            SSA_HIP(LaunchCountBelow(static_cast<const uint64_t*>(idsA->Ptr), nids, static_cast<const uint64_t*>(bounds->Ptr), int(runs.size()),
                                     static_cast<uint64_t*>(counts->Ptr), StreamHandle));
            SSA_HIP(hipMemcpyAsync(below.data(), counts->Ptr, runs.size() * 8, hipMemcpyDeviceToHost, StreamHandle));
            SSA_HIP(hipStreamSynchronize(StreamHandle));
            runs = below;
        }
    }

    // the ids sorted by rounds of pairwise merges
    const uint64_t tileRows = uint64_t(MergeTileRows());
    BufferPtr tilesDev, splits;
    std::vector<MergeTile> tiles;
    uint64_t* in = static_cast<uint64_t*>(idsA->Ptr);
    uint64_t* other = static_cast<uint64_t*>(idsB->Ptr);
    while (runs.size() > 2) {
        tiles.clear();
        std::vector<uint64_t> next;
        next.push_back(0);
        for (size_t r = 0; r + 1 < runs.size(); r += 2) {
            const uint64_t s = runs[r], m = runs[r + 1], e = r + 2 < runs.size() ? runs[r + 2] : runs[r + 1];
            for (uint64_t d = 0; d < e - s; d += tileRows) tiles.push_back(MergeTile{s, m, e, d});
            next.push_back(e);
        }
        if (!tilesDev || tilesDev->Size < tiles.size() * sizeof(MergeTile)) {
            tilesDev = Buffer::Alloc(tiles.size() * sizeof(MergeTile) + 64);
            splits = Buffer::Alloc(tiles.size() * 16 + 64);
            if (!tilesDev || !splits) return Status::Hip("out of device memory");
        }
        SSA_HIP(hipMemcpyAsync(tilesDev->Ptr, tiles.data(), tiles.size() * sizeof(MergeTile), hipMemcpyHostToDevice, StreamHandle));
        SSA_HIP(LaunchMergeRound(keys, in, other, static_cast<const MergeTile*>(tilesDev->Ptr), int64_t(tiles.size()), static_cast<uint64_t*>(splits->Ptr), StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));  // (the host array of tiles is read until here)
        std::swap(in, other);
        runs = next;
    }

    // the first row of every key, unless deleted
    auto winners = std::make_shared<DeviceFilter>();
    winners->Empty = false;
    winners->Count = nids;
    winners->Bits = Buffer::Alloc(size_t((nids + 63) / 64) * 8 + 64);
    if (!winners->Bits) return Status::Hip("out of device memory");
    SSA_HIP(LaunchMergeWinners(keys, in, nids, keep ? static_cast<const uint8_t*>(keep->Ptr) : nullptr, static_cast<uint64_t*>(winners->Bits->Ptr), StreamHandle));
    auto idBatch = std::make_shared<Batch>();
    idBatch->Rows = nids;
    {
        Column ids;
        ids.Field.Name = "row";
        ids.Field.Type = T_U64;
        ids.Length = nids;
        ids.Values = Buffer::Wrap(in, size_t(nids) * 8);
        idBatch->Cols.push_back(ids);
    }
    BatchPtr kept;
    SSA_RETURN_NOT_OK(FilterApply(*winners, idBatch, kept));
    const int64_t m = kept->Rows;
    BufferPtr rowIds = kept->Cols[0].Values;
    if (kept->Cols[0].ElemOffset != 0 || kept.get() == idBatch.get()) {  // (every row kept: the sorted ids themselves)
        rowIds = Buffer::Alloc(size_t(m) * 8 + 64);
        if (!rowIds) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(rowIds->Ptr, kept->Cols[0].ValuesPtr(), size_t(m) * 8, hipMemcpyDeviceToDevice, StreamHandle));
    }
    if (reverse && m > 1) {  // the sources are read from their ends: the same rows, the other way round
        BufferPtr rev = Buffer::Alloc(size_t(m) * 8 + 64);
        if (!rev) return Status::Hip("out of device memory");
        SSA_HIP(LaunchReverseU64(static_cast<const uint64_t*>(rowIds->Ptr), m, static_cast<uint64_t*>(rev->Ptr), StreamHandle));
        rowIds = rev;
    }
    auto res = std::make_shared<Batch>();
    res->Rows = m;
    for (const Column& col : all->Cols) {
        Column dst;
        if (m == 0) {
            dst.Field = col.Field;
            dst.Field.Nullable = false;
            dst.Values = Buffer::Alloc(64);
            if (!dst.Values) return Status::Hip("out of device memory");
            SSA_HIP(hipMemsetAsync(dst.Values->Ptr, 0, 64, StreamHandle));
            if (IsStringLike(col.Field.Type)) dst.Chars = Buffer::Alloc(16);
        } else {
            SSA_RETURN_NOT_OK(GatherRows(col, rowIds, m, col.Validity != nullptr, dst));
            dst.Field.Name = col.Field.Name;
        }
        res->Cols.push_back(std::move(dst));
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = res;
    return Status::OK();
}

}  // namespace ssa
