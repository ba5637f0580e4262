// Device-resident record batches and the step executor.
//
// The executor is the GPU counterpart of NSsa::TProgram::ApplyTo / TProgramStep::Apply
// (ydb/core/formats/arrow/program.h:435-447, program.cpp:869-902 of the reference): it runs the steps
// of a parsed program in order, each as a fused scan over an HBM-resident batch, and produces a new
// HBM-resident batch.  Host <-> device crossings happen only in Import/Export (the Arrow C Data
// Interface boundary) and for the handful of scalars a step needs on the host (row counts).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "arrow_abi.h"
#include "kernels.h"
#include "plan.h"
#include "program.h"

namespace ssa {

// Size-bucketed cache of device allocations (hipMalloc / hipFree are synchronising and slow).
class DevPool {
public:
    static DevPool& Instance();
    void* Alloc(size_t bytes, int* device);
    void Free(void* p, size_t bytes, int device);
    void Trim();
    size_t BytesInUse() const { return InUse; }

private:
    std::mutex Lock;
    std::map<std::pair<int, size_t>, std::vector<void*>> FreeLists;  // (device, block size) -> free blocks
    size_t InUse = 0;
    size_t Cached = 0;  // bytes in FreeLists
    static constexpr size_t kMaxCachedBytes = size_t(64) << 30;  // a scan of an SF100 shard frees and takes again a few GB
};

struct Buffer {
    void* Ptr = nullptr;
    size_t Size = 0;
    bool Owned = false;  // allocated from DevPool (else: caller-owned device memory that outlives the batch)
    int Device = 0;      // owned blocks: the device they live on
    ~Buffer();
    static std::shared_ptr<Buffer> Alloc(size_t bytes);
    static std::shared_ptr<Buffer> Wrap(void* p, size_t bytes);
};
using BufferPtr = std::shared_ptr<Buffer>;

struct DictInfo;
struct DictLowering;
// Process-wide counters: steps / filters that ran over dictionary codes; whole dictionary columns decoded by a gather.
extern std::atomic<int64_t> gDictStepsOnCodes, gDictColumnsDecoded;
// ... sets of slices (of one schema, under one step) that ApplyProgramMany scanned with one launch; sets it ran slice by slice.
extern std::atomic<int64_t> gManyStepsOneLaunch, gManyStepsPerSlice;
extern std::atomic<int64_t> gKeyedStepsCompacted;  // keyed scans run the third way: the rows behind the filter dealt out over the lanes, taken one by one
extern std::atomic<int64_t> gRadixCursorOverflows;  // sorts by region whose places ran out of room and were done again with counts
extern std::atomic<int64_t> gKeyedStepsDeferred;  // keyed scans whose level-2 updates were applied region by region

struct Column {
    FieldDesc Field;         // name, type, nullable (validity present), scalar flag + value
    int64_t Length = 0;
    int64_t NullCount = 0;   // -1: unknown
    BufferPtr Values;        // fixed width: values; BOOL: bitmap; strings: int32 offsets (Length + 1)
    BufferPtr Validity;      // may be null
    BufferPtr Chars;         // strings
    int64_t ElemOffset = 0;  // element offset of row 0 inside Values (fixed width / offsets)
    int32_t ValBitOff = 0;   // BOOL: bit offset of row 0 inside Values
    int32_t ValidBitOff = 0;
    // Dictionary-encoded string / binary column: Values holds the codes (of Dict->IndexType), Field.Type stays the
    // logical type, Chars is unused.
    std::shared_ptr<DictInfo> Dict;
    const void* ValuesPtr() const;  // address of row 0
    int StorageType() const;        // type of the elements of Values: the code type of a dictionary column
};

// The dictionary of an encoded column, shared by the batches that carry the column, and what has been worked out
// about it: equal entries (a GROUP BY on codes needs one code per distinct string), and the results of functions
// evaluated once per entry.
struct DictInfo {
    Column Entries;              // the dictionary's values: a plain string / binary column
    int32_t IndexType = T_I32;
    std::mutex Lock;             // everything below
    bool Analysed = false;
    bool HasDuplicates = false;
    int64_t EmptyCode = -1;      // smallest code of the empty string, -1 when no entry is empty
    std::vector<int32_t> Canon;  // per code: the smallest code with the same bytes
    std::shared_ptr<struct Batch> EntriesAndNull;  // one column: the entries and one null row (input of per-entry evaluation)
    struct Table {
        BufferPtr Payload, Valid;  // entries + 1 values (see LookupTable)
        int32_t Type = T_NONE;
        bool Nullable = false;
    };
    std::map<std::string, Table> Tables;  // function fingerprint -> its per-entry results
};

struct Batch {
    std::vector<Column> Cols;
    int64_t Rows = 0;
    std::vector<FieldDesc> Schema() const;
    int FindColumn(const std::string& name) const;  // -1 if absent or duplicated
};
using BatchPtr = std::shared_ptr<Batch>;

// Timing of the dominant scan kernel of the most recent Apply calls (HIP events on the executor's stream).
struct KernelProfile {
    bool Enabled = false;
    double ScanMs = 0;       // accumulated elapsed time of the fused scan kernels
    int64_t ScanLaunches = 0;
    int64_t ScanBytes = 0;   // algorithmic bytes of those launches (source columns touched once)
    std::string LastKernel;
};

// Diagnostics (SSA_HOST_PHASES=1): where the host's time of an apply goes.  Mark() notes the time at a named point of the
// calling thread's apply, Report() prints the intervals between the marks to stderr and forgets them.
struct HostPhases {
    static bool On();
    static void Mark(const char* name);
    static void Report(const char* what);
};

// Column descriptions of a struct ArrowSchema (no data): used to explain / type-check a program on the host.
Status SchemaFields(const ArrowSchema* schema, std::vector<FieldDesc>& out);

// TColumnFilter (ydb/core/formats/arrow/arrow_filter.h:20-211) with its bits in HBM.  `Empty` mirrors Filter.empty():
// such a filter has no size and stands for "every row" (DefaultValue) or "no row".
struct DeviceFilter {
    bool Empty = true;
    bool DefaultValue = true;
    int64_t Count = 0;
    BufferPtr Bits;  // ceil(Count / 64) words, bit i of word i / 64 (Arrow order), zero past Count
};
using FilterPtr = std::shared_ptr<DeviceFilter>;

// One key column of a predicate border (MakePredicateFilter): the border's value as the column's type.
struct BorderValue {
    std::string Name;
    TScalar Value;  // numbers as the VM payload (floats widened to double), strings as bytes
};

// Rows [start, start + count) of a column as a view of the same buffers.
Column ColumnView(const Column& col, int64_t start, int64_t count);
// A batch of views of `data`, a flat buffer filled by Executor::PackData and described by `header` (Executor::PackedLayout).
Status UnpackBatch(const uint8_t* header, size_t headerBytes, void* data, size_t dataBytes, BatchPtr& out);

// Several slices of one schema scanned by one launch (Executor::ApplyProgramMany).
struct ManySlices {
    std::vector<const Batch*> In;                  // every slice; In[0] is the batch the plan was made for
    std::vector<std::vector<OffsetCheck>> Checks;  // per slice, or empty
    std::vector<int64_t> GroupStart;               // out: the groups of slice i are rows [GroupStart[i], GroupStart[i + 1]) of the result ...
    std::vector<int64_t> GroupCount;               // ... or, when this is not empty, the first GroupCount[i] of them
    std::vector<SliceDesc> HostDescs;              // what the launch reads, kept until it has been copied
    std::vector<int32_t> HostMap;
};

// Where the buffers of one column lie in the executor's pinned staging memory after Executor::PackToHost (valid until
// the executor's next call).
struct HostColumnView {
    const uint8_t* Values = nullptr;  // fixed width: the rows' values; BOOL: the bytes that hold the rows' bits; strings: rows + 1 offsets from 0
    size_t ValuesBytes = 0;
    int ValBitOff = 0;                // BOOL: bit of row 0 inside the first byte
    const uint8_t* Valid = nullptr;   // null: no validity
    size_t ValidBytes = 0;
    int ValidBitOff = 0;
    const uint8_t* Chars = nullptr;   // strings: the bytes in use
    size_t CharsBytes = 0;
};

class Executor {
public:
    // A few words of pinned host memory: where small results (a scan's status, a group count, probed offsets) land
    // without the staging copy an asynchronous copy into pageable memory takes.
    int32_t* HostWords = nullptr;
    static constexpr int kHostWords = 128;
    uint8_t* HostStaging = nullptr;  // pinned, kHostStagingBytes, allocated at the first PackToHost
    static constexpr size_t kHostStagingBytes = size_t(1) << 20;
    bool EnsureHostStaging();  // false: no pinned memory to be had
    // Stored blobs on their way to the device: pageable host memory is copied by a few threads into slots of pinned
    // memory and leaves them by DMA (one thread's memcpy, which is what a pageable hipMemcpy is, feeds half the link).
    static constexpr size_t kBlobSlotBytes = size_t(8) << 20;
    static constexpr int kBlobSlots = 6;
    uint8_t* BlobStaging = nullptr;  // pinned, kBlobSlots * kBlobSlotBytes, allocated at the first large import
    hipEvent_t BlobSlotDone[kBlobSlots] = {};
    bool EnsureBlobStaging();
    // Copies the blobs to deviceBase + base[i]; false: no pinned memory, nothing queued.
    Status UploadBlobs(const std::vector<std::pair<const uint8_t*, size_t>>& blobs, const std::vector<size_t>& base, size_t total, char* deviceBase,
                       bool& done);
    explicit Executor(int device);
    ~Executor();

    Status ImportBatch(const ArrowArray* array, const ArrowSchema* schema, BatchPtr& out);
    Status ExportBatch(const Batch& batch, ArrowArray* outArray, ArrowSchema* outSchema);

    // TNativeSerializer::DoDeserialize on the device: IPC blobs (a whole stream, or a lone RecordBatch message read with
    // `schema`) -> HBM-resident batches, one per blob; all compressed buffers are decoded by one kernel launch.
    Status ImportBlobs(const std::vector<std::pair<const uint8_t*, size_t>>& blobs, const std::vector<FieldDesc>& schema,
                       std::vector<BatchPtr>& out);
    Status ImportBlobsQueued(const std::vector<std::pair<const uint8_t*, size_t>>& blobs, const std::vector<FieldDesc>& schema,
                             std::vector<BatchPtr>& out);  // (ImportBlobs without the wait on an error's way out)

    // TProgram::ApplyTo: every step in order.
    Status ApplyProgram(const TProgram& program, const BatchPtr& in, BatchPtr& out);
    // The program applied to each of `in` (slices of a table: program.h:423-433 runs them one by one), one result per
    // slice.  Slices of one schema (the same columns carry nulls) whose step is an aggregation are scanned by one launch.
    Status ApplyProgramMany(const TProgram& program, const std::vector<BatchPtr>& in, std::vector<BatchPtr>& out);
    // TProgramStep::BuildFilter for the first step: Arrow-order bitmap (LSB first) of rows*1 bits, on the host.
    Status BuildFilter(const TProgramStep& step, const BatchPtr& in, std::vector<uint8_t>& bits, int64_t& selected);

    // THashConstructor::BuildHashUI64's XXH64 branch: hashes[row] = XXH64(seed) over the row's non-null key fields.
    Status ShardingHash(const Batch& in, const std::vector<std::string>& keyNames, uint64_t seed, BufferPtr& hashes);
    // BuildHashUI64 + TShardingSplitIndex::Apply: `out` holds the rows shard after shard (ascending row order inside a
    // shard), shardRows[s] of them in shard s.  A batch that the reference leaves unsharded comes back unchanged with all
    // rows counted in shard 0.
    Status ShardingSplit(const Batch& in, const std::vector<std::string>& keyNames, uint32_t shards, BatchPtr& out,
                         std::vector<uint32_t>& shardRows, bool balance = true);

    // -- TColumnFilter ------------------------------------------------------------------------------------
    // TColumnFilter(std::vector<bool>) given as runs: run j has `runs[j]` rows of value firstValue ^ (j & 1).
    Status FilterFromRuns(bool firstValue, const std::vector<uint32_t>& runs, FilterPtr& out);
    // The reference's representation back: run lengths, LastValue, Count (arrow_filter.h:22-25).
    Status FilterToRuns(const DeviceFilter& f, std::vector<uint32_t>& runs, bool& lastValue);
    // TProgramStep::BuildFilter kept in HBM.
    Status FilterFromStep(const TProgramStep& step, const BatchPtr& in, FilterPtr& out);
    // The same without allocating device memory: the bitmap goes to caller-provided device memory (ceil(rows / 64)
    // words), the per-tile counts to the library's static scratch buffer; up to kNoAllocMaxRows rows.
    Status FilterFromStepNoAlloc(const TProgramStep& step, const Batch& in, uint64_t* outBits, int64_t& selected);
    Status FilterAndOr(const DeviceFilter& a, const DeviceFilter& b, bool isOr, FilterPtr& out);
    Status FilterCombineSequentialAnd(const DeviceFilter& self, const DeviceFilter& ext, FilterPtr& out);
    Status FilterPredicate(const Batch& in, const std::vector<BorderValue>& border, int compareType, FilterPtr& out);
    // GetFilteredCount: -1 stands for the reference's empty optional (an allow-all filter that has no size).
    Status FilterCountSelected(const DeviceFilter& f, int64_t& selected);
    // BuildSimpleFilter as an Arrow-order bitmap on the host: ceil(Count / 8) bytes.
    Status FilterBits(const DeviceFilter& f, std::vector<uint8_t>& bits);
    // TColumnFilter::Apply: the rows the filter keeps, in order.
    Status FilterApply(const DeviceFilter& f, const BatchPtr& in, BatchPtr& out);

    // `in` itself when it has no dictionary column, else a copy with those columns decoded (a gather through the codes)
    Status DecodedBatch(const BatchPtr& in, BatchPtr& out);

    // -- result batches between GPUs (exchange.cpp) -----------------------------------------------------------
    // A batch as one flat buffer: the layout's description (host bytes: row count, per column its type, name and the place
    // of its buffers) and the size of the data; PackData fills the data, UnpackBatch makes a batch of views of it.
    Status PackedLayout(const Batch& in, std::vector<uint8_t>& header, uint64_t& dataBytes);
    Status PackData(const Batch& in, void* dst, size_t capacity);
    // A small batch (a result of few groups) to the host in one copy: packed on the device, copied into pinned memory.
    // `done` stays false for batches this does not take (large ones, dictionary or scalar columns): the caller copies
    // those buffer by buffer.
    Status PackToHost(const Batch& in, std::vector<HostColumnView>& views, bool& done);
    Status ConcatBatches(const std::vector<BatchPtr>& parts, BatchPtr& out);

    // TMergePartialStream::AddSource + DrainAll (ydb/core/formats/arrow/reader/merger.h:78-95): sources sorted by
    // `keyNames` (ascending), merged into one batch sorted by them (descending when `reverse`) where every key appears
    // once, with the row of its greatest `versionNames`; a key whose newest row is denied by its source's filter does not
    // appear.  filters: empty, or one per source (null = none).
    Status MergeSorted(const std::vector<BatchPtr>& sources, const std::vector<FilterPtr>& filters, const std::vector<std::string>& keyNames,
                       const std::vector<std::string>& versionNames, bool reverse, BatchPtr& out);

    Status Sync();
    hipStream_t Stream() const { return StreamHandle; }
    KernelProfile Profile;
    void ProfileCollect();  // the time of the last timed launch into Profile (waits for it)

private:
    Status ApplyStep(const TProgramStep& step, const BatchPtr& in, BatchPtr& out);
    // `stable`: `step` is one of the program's own steps (it lives as long as the executor does and never changes), so
    // its plan for this input schema may be kept (PlanCache) -- planning is a tenth of an apply of a shard-sized batch.
    Status ApplyStepPlain(const TProgramStep& step, const BatchPtr& in, BatchPtr& out, bool stable = false);
    Status ApplyStepWhole(const TProgramStep& step, const BatchPtr& in, BatchPtr& out, bool stable);
    // A step too wide for one scan (more aggregates, computed outputs, temporaries, source columns or constants than the
    // kernels hold): the key-less aggregates, or the projected columns, are dealt out over several steps of the same
    // assigns and filters, each run as a scan of its own, and their results put side by side.  Kept per step.
    std::map<const TProgramStep*, std::vector<std::shared_ptr<TProgramStep>>> SplitSteps;
    std::vector<std::shared_ptr<TProgramStep>> RetiredSteps;
    Status ApplySplit(const std::vector<std::shared_ptr<TProgramStep>>& parts, const BatchPtr& in, BatchPtr& out);
    std::map<std::string, std::shared_ptr<const StepPlan>> PlanCache;  // (step address, input schema) -> plan
    Status PlanFor(const TProgramStep& step, const Batch& in, bool stable, std::shared_ptr<const StepPlan>& kept);
    bool StableStep = false;  // the step being run is one of the program's own (plans made for it may be kept)
    Status FixedKeysPlan(const TProgramStep& step, const Batch& shadow, const std::vector<std::pair<size_t, std::string>>& renames,
                         const std::vector<uint32_t>& lens, std::shared_ptr<const StepPlan>& kept);
    // -- dictionary-encoded columns (dict.cpp) ---------------------------------------------------------------
    Status ApplyStepDict(const TProgramStep& step, const BatchPtr& in, BatchPtr& out);
    Status LowerDictStep(const TProgramStep& step, const BatchPtr& in, bool filterOnly, DictLowering& lowering);
    Status FilterFromStepDict(const TProgramStep& step, const BatchPtr& in, FilterPtr& out);
    Status FilterFromStepPlain(const TProgramStep& step, const BatchPtr& in, FilterPtr& out);
    Status AnalyseDictionary(DictInfo& d);
    Status DecodeColumn(const Column& src, int64_t rows, Column& dst);
    Status DecodeCodes(const Column& codes, DictInfo& d, Column& dst);
    Status EntryTable(DictInfo& d, const std::string& column, const TProgramStep& step, const TAssign& fn, DictInfo::Table& out);
    Status RunNoKey(const StepPlan& plan, const Batch& in, BatchPtr& out);
    Status RunKeyed(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out);
    Status RunNoKeyAsOneGroup(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out);
    Status RunKeyedUniform(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out, bool& done);
    Status RunKeyedScan(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out,
                        const std::vector<OffsetCheck>& checks, bool* notUniform, ManySlices* many = nullptr);
    // The records of a scan in region order (region.cpp); `keep` holds the buffers behind the pointers.
    struct SortedRecords {
        RecordSet set{};
        const uint32_t* region_start = nullptr;
        // the first round went without its counting pass (RadixArgs.cursor): *overflow says that some digit's place ran out
        // of room and the sort is void; the digits' cursors and the room of a place for diagnostics
        const uint32_t* overflow = nullptr;
        const uint32_t* cursors = nullptr;
        size_t ncursors = 0;
        uint32_t cursor_cap = 0;
        int64_t stride = 0;  // elements between two operand arrays of `set`
        std::vector<BufferPtr> keep;
    };
    // `byCursors`: the first of two rounds without its counting pass -- every digit's place in the sorted arrays gets room to
    // spare; the caller looks at *overflow once the stream is synchronised and sorts again without if it is set.
    Status SortDeferredRecords(const KeyedSink& sink, int64_t rows, int nvals, bool needMask, int regionShift, uint64_t regions,
                               SortedRecords& out, bool byCursors = false);
    Status ApplyDeferredRecords(const KeyedSink& sink, const ScanArgs& args, int64_t rows, int nvals, bool needMask, int regionSlots,
                                uint64_t cap);
    // The groups of a region pass in its temporary table, and what the kernel that moves them to the output needs.
    struct RegionGroups {
        RegionEmitArgs args{};
        std::vector<BufferPtr> keep;
    };
    Status AggregateDeferredRecords(const KeyedSink& sink, const ScanArgs& args, const SortedRecords& sorted, int64_t rows, int nvals,
                                    int ldsSlots, int smallSlots, uint64_t regions, RegionGroups& out, int64_t* groupsAt);
    Status EmitRegionGroups(RegionGroups& groups, const KeyedFinalizeArgs& fa);
    Status DescribeSlices(const StepPlan& plan, ManySlices& many, ScanArgs& args, int64_t tileRows, uint64_t cap, LaunchCfg& cfg,
                          BufferPtr& descs, BufferPtr& map);
    Status RunSegments(const std::vector<CopySegment>& segments);
    Status StringBounds(const Batch& in, std::vector<int32_t>& bounds);
    Status OffsetRamp(int len, int64_t n, BufferPtr& out);
    std::map<int, BufferPtr> OffsetRamps;  // string length -> offsets of 65536 strings of that length
    Status ApplyStepMany(const TProgramStep& step, const std::vector<BatchPtr>& in, std::vector<BatchPtr>& out);
    Status RunProject(const TProgramStep& step, const StepPlan& plan, const BatchPtr& in, BatchPtr& out);
    Status RunFilterBitmap(const StepPlan& filterPlan, const Batch& in, BufferPtr& bitmap, BufferPtr& tileOffsets,
                           int64_t& selected);
    Status FillScanArgs(const StepPlan& plan, const Batch& in, int sinkLdsBytes, int kernelId, ScanArgs& args,
                        LaunchCfg& cfg, int64_t& algoBytes);
    int WideRowsPerThread(const StepPlan& plan, const Batch& in) const;
    hipFunction_t SpecializedKernel(const char* header, const char* entry, const std::string& defines, const ScanArgs& args,
                                    const void* sinkImage, size_t sinkBytes, int64_t rows, LaunchCfg& cfg);
    hipError_t LaunchSpecialized(hipFunction_t fn, const ScanArgs& args, const void* sink, size_t sinkBytes, const LaunchCfg& cfg);
    Status CheckScanStatus(bool* tableFull = nullptr, bool* notUniform = nullptr);
    Status DecodeScanStatus(uint32_t st, bool* tableFull, bool* notUniform);
    Status CompactColumn(const Column& src, int64_t rows, const BufferPtr& bitmap, const BufferPtr& tileOffsets,
                         int64_t selected, Column& dst);
    Status MaterializeScalar(const OutColumn& oc, int64_t rows, Column& dst);
    Status PackValidity(const BufferPtr& validBytes, int64_t rows, Column& col);
    Status GatherRows(const Column& src, const BufferPtr& rowIdx, int64_t n, bool keepValidity, Column& dst);
    void ProfileBegin();
    void ProfileEnd(const char* kernel, int64_t algoBytes);
    bool ProfilePending = false;  // a timed launch whose events have not been read yet
    int64_t ProfilePendingBytes = 0;

    static constexpr int64_t kUniformKeysMinRows = 65536;  // below, the probe's host round trip outweighs the gain
    int Device;
    int ScanRowsPerThread = kRowsPerThread;  // rows per thread of the kernel FillScanArgs / SpecializedKernel plan for
    hipStream_t StreamHandle = nullptr;
    hipEvent_t EvStart = nullptr, EvStop = nullptr;
    hipEvent_t EvTuneA = nullptr, EvTuneB = nullptr;
    BufferPtr StatusWord;   // device uint32
    std::vector<LookupTable> StepTables;      // lookup tables of the step being run (FillScanArgs copies them)
    bool ForceSerializedKeys = false;         // the step's string keys have been replaced by codes: still `serialized` semantics
    const TProgramStep* HintOverride = nullptr;  // the step whose hints a rewritten copy uses
    void* NoAllocScratch = nullptr;  // this device's copy of the static scratch buffer (resolved at construction)
    // Sizing of the two hash-table levels of a keyed step, learnt from its previous run (performance only).
    struct KeyedHint {
        int LdsCap = 16;
        bool ManyGroups = false;
        int64_t Groups = 0;      // groups of the step's last scan
        uint64_t TableCap = 4096;
        // The register-held groups pay when a few groups take most rows (TPC-H Q1: 2.7x) and cost registers and per-row
        // work when they do not (a selective filter in front of thirty groups: 0.6x).  A kernel built for the step is
        // timed once each way on its first large scans and the faster way is kept.
        // Behind a selective filter a third way wins: no register-held groups, the rows that passed dealt out evenly over
        // the lanes of the wave and taken one by one (a 3 % filter in front of thirty groups: 2x over the plain way);
        // timed like the others when the step qualifies (KeyedSink.compact_n).
        int HotChoice = -1;      // -1 undecided, 0 off, 1 on, 2 off with the rows behind the filter one by one
        bool UniformKeysFailed = false;  // a batch of this step had string keys of differing lengths: no further attempts
        // Milliseconds and rows of the scans that timed each way: one scan of four million rows settles a way, shard-sized
        // batches (64 K rows and more) add up to that over their first applies.
        double TunedMs[3] = {0, 0, 0};
        int64_t TunedRows[3] = {0, 0, 0};
        // Groupings of a million groups and more: records sorted by table region and applied inside LDS, or every row
        // updating the table in HBM itself.  The sorted way wins where the key is in the record (fixed-width keys:
        // ClickBench UserID 1.6x), the direct way where every update compares key bytes in memory anyway (long string
        // keys).  Both ways are timed once on the step's large scans, scan to finished table.
        int DeferChoice = -1;    // -1 undecided, 0 row by row, 1 by region
        double MsDeferOn = 0, MsDeferOff = 0;
        bool NoRadixCursors = false;  // a sort by region without counting passes ran out of room once: with counts from then on
        // The level-2 tables of the step's small scans (a few thousand slots each), kept from scan to scan: the launch that
        // emits their groups leaves them free again.  Clean: nothing has touched them since.
        struct SmallTable {
            BufferPtr Lo, Hi, Acc, Seen, Rep;
            uint64_t TotalCap = 0;
            int Naggs = 0;
            bool TwoWords = false;
            bool Clean = false;
        };
        SmallTable Small;
    };
    std::map<const TProgramStep*, KeyedHint> KeyedHints;
};

}  // namespace ssa
