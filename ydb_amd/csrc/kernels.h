// Host-visible launch interface of the HIP kernels.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#endif

#include "vm.h"

namespace ssa {

constexpr int kBlock = 256;           // threads per workgroup = 4 wavefronts of 64
constexpr int kRowsPerThread = 4;     // U: rows each thread evaluates per tile
constexpr int kTileRows = kBlock * kRowsPerThread;
constexpr int kMaxGrid = 256 * 8;     // persistent grid: 256 CUs x up to 8 resident workgroups

// ---- scan sinks ---------------------------------------------------------------------------------
struct NoKeySink {
    int32_t naggs;
    AggSpec aggs[kMaxAggs];
    uint64_t* partial;       // [grid][kMaxAggs] per-workgroup partial states
    uint32_t* partial_seen;  // [grid] bit k: aggregate k saw a non-null argument
};

struct FilterSink {
    uint64_t* bitmap;        // ceil(nrows / 64) words, bit i of word w = row 64 w + i passes the filter
    uint32_t* tile_counts;   // [ntiles] selected rows per tile of kTileRows rows
};

struct ProjectOut {
    void* values;            // typed output values (BOOL outputs: one byte per row, packed afterwards)
    uint8_t* valid_bytes;    // one byte per row, packed afterwards; nullptr when the output cannot be null
    int32_t type;
    int32_t reg;             // >= 0 R register, <= -2 column slot
    uint32_t fetch_kind;     // pre-decoded operand for `reg`
    uint32_t fetch_payload;
};

struct ProjectSink {
    const uint64_t* bitmap;      // nullptr: every row is selected (no compaction)
    const int64_t* tile_offsets; // exclusive prefix sum of tile_counts
    int32_t nouts;
    ProjectOut outs[kMaxOuts];
};

// Group-by: open-addressing table in global memory (the "spill table"), fronted by a small
// per-workgroup table in LDS that absorbs the hot groups.
// Aggregates the keyed sink keeps in registers for the first groups of a workgroup.
constexpr int kHotAggs = 6;
enum HotKind : int32_t { HOT_NONE = 0, HOT_SUM_F64 = 1, HOT_SUM_I64 = 2, HOT_ROWS = 3 };

constexpr int kMaxCompact = 8;
struct KeyedSink {
    int32_t naggs;           // distinct accumulators (identical aggregates of the step share one)
    int32_t nkeys;
    AggSpec aggs[kMaxAggs];
    KeySpec keys[kMaxKeys];
    int32_t two_words;       // keys use the hi word
    int32_t hashed;          // keys are 126-bit hashes (strings / more than 119 key bits)
    int32_t lds_cap;         // level-1 (LDS) table slots, power of two, 0 = none
    int32_t lds_rep;         // accumulator replicas per level-1 slot, power of two
    uint64_t table_mask;     // level-2 capacity - 1
    // Level 2 is one array per field, on purpose.  With the fields of a group next to each other (one 64-byte line per
    // row instead of two) ClickBench's GROUP BY UserID ran 1.8 times slower: the key words are read with loads, which the
    // XCD's L2 serves, the accumulators are updated by atomics, which run at the memory side and drop their line from L2
    // -- on a shared line the two keep evicting each other.
    uint64_t* t_key_lo;      // [cap] ~0 = free
    uint64_t* t_key_hi;      // [cap] ~0 = not yet published
    uint64_t* t_acc;         // [naggs][cap]
    uint32_t* t_seen;        // [cap] bit k: accumulator k saw a non-null argument
    uint64_t* t_rep_row;     // [cap] hashed keys: smallest representative row index seen for the group (~0 = none)
    int32_t hot_kind[kHotAggs];  // register path of accumulator k (HotKind)
    int32_t hot_off[kHotAggs];   // LDS byte offset of its 64-bit operand tile
    int32_t use_hot;             // keep the first groups of a workgroup in registers (else every row goes through the tables)
    int32_t defer;               // rows that miss level 1 are written out as records (rec_*) instead of updating level 2
    uint64_t region_mask;        // a probe sequence stays among the slots that share its start's bits above this mask
                                 // (table_mask: the whole table)
    uint64_t acc_stride;         // slots between the accumulators k and k + 1 of one group in t_acc: the capacity, or the
                                 // capacities of all slices' tables together (SliceDesc)
    // Deferred level-2 updates (defer != 0), one record per input row, at the row's index:
    uint64_t* rec_lo;            // [nrows] key lo word; ~0: the row left no record
    uint64_t* rec_hi;            // [nrows] key hi word (two-word keys)
    uint64_t* rec_val;           // [nvals][nrows] operand of every accumulator that is no counter, in accumulator order
    uint32_t* rec_mask;          // [nrows] bit k: the row updates accumulator k (nullptr: every record updates all of them)
    // Keys of exactly 64 bits (one int64 column: ClickBench's UserID) need the second key word for one bit.  Their records
    // carry the 64 packed bits in rec_lo alone (rec_raw64; every value is a key then, so the rows that left a record are
    // marked in a bitmap of their own, bit row & 63 of word row >> 6); the passes of the sort move half the bytes, and
    // the table's two words are made from the 64 bits where they are needed (RawKeyWords).
    int32_t rec_raw64;
    int32_t pad2_;
    uint64_t* rec_present;
    // A scan whose number of groups is not known yet (or is large) may find level 2 too small: with this set, the first
    // lane that finds no slot raises SCAN_TABLE_FULL in the status word at once, probe loops and the tile loop look at
    // the word and give up, and the host repeats the step with a larger table after microseconds instead of a whole scan.
    // (Launch state like the addresses above: zero in the image a kernel is built for.)
    int32_t abort_check;
    // Behind a selective filter (kernels built with SSA_SPEC_COMPACT) the sink takes the rows that passed one per lane
    // and trip and reads their keys and arguments from the staged tiles at the row's index.  A step qualifies when all of
    // those are staged columns without a validity bitmap (or constants): compact_n > 0, with the tiles (LDS offset, bytes
    // per value) listed for reference.
    int32_t compact_n;
    uint32_t compact_off[kMaxCompact];
    int32_t compact_w[kMaxCompact];
    // Such a kernel may be given, behind the sink's own LDS (this many bytes from lds_sink_off; 0: none), two bytes per row
    // of a tile: the lanes list their rows that passed there, and the lane that takes a row reads which one it is.
    int32_t compact_slots;
    int32_t pad3_;
};

// Kernel arguments of a scan kernel: passed by value, read in place from the kernel-argument segment.
template <class Sink>
struct KArgs {
    ScanArgs a;
    Sink s;
};

#ifndef __HIPCC_RTC__  // the rest is the host's launch interface
struct LaunchCfg {
    int grid;
    int max_grid;         // workgroups the device holds at once for this kernel (the grid of a scan with tiles to spare)
    uint32_t lds_bytes;   // dynamic LDS per workgroup: staged column tiles + R registers + sink area
    hipStream_t stream;
};

hipError_t LaunchScanNoKey(const ScanArgs& a, const NoKeySink& s, LaunchCfg cfg);
hipError_t LaunchScanFilter(const ScanArgs& a, const FilterSink& s, LaunchCfg cfg);
hipError_t LaunchScanProject(const ScanArgs& a, const ProjectSink& s, LaunchCfg cfg);
hipError_t LaunchScanKeyed(const ScanArgs& a, const KeyedSink& s, LaunchCfg cfg);
// Workgroups of a scan kernel that fit on one CU with `ldsBytes` of dynamic LDS (registers and LDS both count).
enum ScanKernelId : int { SCAN_NOKEY = 0, SCAN_FILTER = 1, SCAN_PROJECT = 2, SCAN_KEYED_EXACT = 3, SCAN_KEYED_HASHED = 4 };
int ScanKernelBlocksPerCU(int kernelId, uint32_t ldsBytes);
int KeyedKernelBlocksPerCU(bool hashed, uint32_t ldsBytes);
// Raise the dynamic-LDS limit of the scan kernels (call once per device).
hipError_t ConfigureScanKernels(int maxLdsBytes);

// ---- small follow-up kernels ----------------------------------------------------------------------
// Reduce per-workgroup partials; out_vals[k], out_valid[k] (1 = non-null).
// One output column of a key-less aggregation: which aggregate it shows and how its 64-bit state becomes the value.
enum NoKeyOutMode : int32_t { NOKEY_OUT_PLAIN = 0, NOKEY_OUT_DECIMAL = 1, NOKEY_OUT_ROW = 2, NOKEY_OUT_F32 = 3 };
struct NoKeyOutSpec {
    int32_t agg;
    int32_t mode;
    int32_t mid, hi;   // NOKEY_OUT_DECIMAL: the aggregates that hold the sums of the middle and high parts
    const void* src;   // NOKEY_OUT_ROW: row 0 of the column the value comes from, and the width of its values
    int32_t width;
    int32_t pad_;
};
// `result`: 4 * kMaxOuts + 1 words of device memory (per output: value, high half, validity word, padding; then, when
// `host` is null, the status and the not-null bits); `host`: two words of pinned memory or nullptr.
hipError_t LaunchNoKeyFinalize(const NoKeySink& s, int grid, const NoKeyOutSpec* outs, int nouts, uint64_t* result, const uint32_t* status,
                               uint32_t* host, hipStream_t st);
// tile_offsets[i] = sum(tile_counts[0..i)), total[0] = sum of all.
hipError_t LaunchTileScan(const uint32_t* tile_counts, int64_t ntiles, int64_t* tile_offsets, int64_t* total, hipStream_t st);
// Order-preserving compaction of one fixed-width column (width 1, 2, 4, 8 or 16 bytes).
hipError_t LaunchCompactFixed(const void* src, void* dst, int width, int64_t nrows, const uint64_t* bitmap,
                              const int64_t* tile_offsets, hipStream_t st);
// Compaction of a bit-packed column (validity bitmap or BOOL values) into one byte per selected row.
hipError_t LaunchCompactBits(const uint8_t* src_bits, int64_t src_bit_off, uint8_t* dst_bytes, int64_t nrows,
                             const uint64_t* bitmap, const int64_t* tile_offsets, hipStream_t st);
// dst bit i = (src_bytes[i] != 0); returns the number of zero bytes through *null_count (device counter, pre-zeroed).
hipError_t LaunchPackBytesToBits(const uint8_t* src_bytes, int64_t n, uint8_t* dst_bits, unsigned long long* zero_count,
                                 hipStream_t st);
// String compaction: lengths of selected rows -> new offsets (exclusive scan) -> byte copy.
hipError_t LaunchCompactStringLens(const int32_t* offsets, int32_t* out_lens, int64_t nrows, const uint64_t* bitmap,
                                   const int64_t* tile_offsets, hipStream_t st);
hipError_t LaunchExclusiveScanI32(const int32_t* in, int32_t* out, int64_t n, int64_t* block_tmp, hipStream_t st);
hipError_t LaunchCompactStringBytes(const int32_t* offsets, const uint8_t* chars, const int32_t* new_offsets,
                                    uint8_t* out_chars, int64_t nrows, const uint64_t* bitmap,
                                    const int64_t* tile_offsets, hipStream_t st);
// Fill n elements of `width` bytes with the low bytes of `value`.
hipError_t LaunchFill(void* dst, int width, uint64_t value, int64_t n, hipStream_t st);
// Popcount of the first nbits of a bitmap starting at bit offset.
hipError_t LaunchCountBits(const uint8_t* bits, int64_t bit_off, int64_t nbits, unsigned long long* out, hipStream_t st);

// ---- group-by finalize ----------------------------------------------------------------------------
struct KeyedFinalizeArgs {
    uint64_t capacity;
    const uint64_t* t_key_lo;
    const uint64_t* t_key_hi;
    const uint64_t* t_acc;
    const uint32_t* t_seen;
    const uint64_t* t_rep_row;
    const int64_t* tile_offsets;      // [capacity / kTileRows + 1] groups in the tiles before each tile of slots
    int32_t naggs;                    // output aggregates (the step's GroupBy list)
    int32_t nkeys;
    int32_t two_words;
    int32_t hashed;
    AggSpec aggs[kMaxAggs];
    int32_t agg_src[kMaxAggs];        // accumulator each output aggregate reads
    int32_t agg_out_type[kMaxAggs];
    int32_t agg_row_index[kMaxAggs];  // the accumulator is a row index (some()): raw uint64 out
    int32_t agg_nan_src[kMaxAggs];    // float MIN / MAX: accumulator holding MIN((row << 1) | is-NaN), or -1
    int32_t agg_zero_src[kMaxAggs];   // float MIN / MAX: accumulator holding MIN((row << 1) | sign) over zero arguments, or -1
    KeySpec keys[kMaxKeys];
    // outputs (one row per group)
    uint64_t* group_rep_row;          // hashed keys: representative input row of each group
    void* agg_vals[kMaxAggs];
    uint8_t* agg_valid_bytes[kMaxAggs];
    void* key_vals[kMaxKeys];         // exact keys only: decoded from the table
    uint8_t* key_valid_bytes[kMaxKeys];
};
// Tables of up to kSmallTableSlots slots each, counted and emitted by one launch (SmallTablesEmitKernel).
constexpr int kSmallTableSlots = 8192;
constexpr int kSmallEmitCounts = 32;   // host word of the first table's group count
constexpr int kSmallEmitTables = 64;   // tables of one launch (their counts fit the executor's pinned words)
struct SmallEmitArgs {
    KeyedFinalizeArgs f;      // capacity: the slots of all tables (the stride of the accumulator arrays); tile_offsets unused
    uint32_t slice_cap;       // slots of one table
    int32_t reset;            // leave the table as KeyedTableInitKernel does
    int32_t nacc;             // reset: the table's accumulators and their identities
    int32_t pad_;
    uint64_t ident[kMaxAggs];
    const uint32_t* status;   // the scan's status word
    uint32_t* host;           // pinned: [0] the status, [kSmallEmitCounts + b] the groups of table b
};
hipError_t LaunchSmallTablesEmit(const SmallEmitArgs& a, int tables, hipStream_t st);
uint64_t KeyedTableIdentity(int aggKind);  // initial value of a level-2 accumulator
hipError_t ConfigureKeyedKernels(int maxLdsBytes);
// Free keys, identity accumulators, clear flags: every array of the level-2 table `sink` points at, in one launch.
hipError_t LaunchKeyedTableInit(const KeyedSink& sink, bool two_words, bool hashed, hipStream_t st);
// Deferred level-2 updates of a keyed scan (KeyedSink.defer).  The table is cut into regions of `region_slots` slots
// (probe sequences wrap inside their region); the records the scan left are sorted by region with one or two rounds of
// a counting sort on the region number, and every region is then brought up to date by one workgroup, in LDS.
struct RecordSet {
    uint64_t* lo;
    uint64_t* hi;             // two-word keys
    uint64_t* val;            // [nvals][stride]
    uint32_t* mask;           // nullptr: every record updates all accumulators
    uint32_t* row;            // the record's row (nullptr where it is not carried: in the scan's output the record's index)
};
// The table's key words of a 64-bit key carried as one word (KeyedSink.rec_raw64): what KeyBuilder::Finish makes of it.
__device__ __forceinline__ void RawKeyWords(uint64_t raw, uint64_t& lo, uint64_t& hi) {
    lo = raw & 0x7FFFFFFFFFFFFFFFull;
    hi = raw >> 63;
}
// One round of the sort: the records of each segment of `src` (seg_start) to `dst`, ordered by digit inside the segment.
struct RadixArgs {
    RecordSet src, dst;
    int64_t stride;               // elements between two operand arrays of a record set (the batch's row count)
    int32_t nvals;
    int32_t two_words;
    int32_t raw64;                // the records carry 64-bit keys in one word (KeyedSink.rec_raw64): no hi arrays
    const uint64_t* present;      // raw64, first round: bit i says whether index i of src holds a record (nullptr: all do)
    int32_t skip_free;            // src.lo may hold ~0: no record at that index (not with raw64)
    uint64_t table_mask;
    int32_t region_shift;         // region of a slot = slot >> region_shift
    int32_t digit_shift;          // digit = (region >> digit_shift) & (ndigits - 1)
    int32_t ndigits;              // 64 .. 256, a power of two
    int32_t nseg;
    int32_t chunks;               // workgroups per segment
    const uint32_t* seg_start;    // [nseg + 1]; nullptr: one segment, the `stride` records of src
    uint32_t* counts;             // [nseg][chunks][ndigits] records per chunk and digit; then those of the earlier chunks
    uint32_t* out_start;          // [nseg * ndigits + 1] first record of each (segment, digit) in dst
    // Without a counting pass: (segment, digit) number b of dst has room for bucket_cap records from b * bucket_cap on
    // (out_start is not read), a tile's records of a digit take their places by one atomic add on the digit's cursor, and
    // a digit that runs out of room raises *overflow (the caller sorts again, with counts).  The segments of src may have
    // been left that way by the round before: segment seg then is the seg_count[seg] (at most seg_cap) records from
    // seg * seg_cap on, and seg_start is not read.
    uint32_t* cursor;             // [nseg * ndigits], zero before the launch; nullptr: counts / out_start as above
    uint32_t bucket_cap;
    uint32_t* overflow;
    const uint32_t* seg_count;    // [nseg] or nullptr: seg_start[seg + 1] ends segment seg
    uint32_t seg_cap;
    int64_t dst_stride;           // elements between two operand arrays of dst (0: stride)
};
hipError_t LaunchRadixCount(const RadixArgs& a, hipStream_t st);
hipError_t LaunchRadixOffsets(const RadixArgs& a, hipStream_t st);
hipError_t LaunchRadixScatter(const RadixArgs& a, hipStream_t st);
struct RegionApplyArgs {
    KeyedSink s;                  // the scan's sink: table, accumulators, keys
    const DevColumn* cols;        // the batch's columns on the device (hashed keys: rows are compared by their key values)
    RecordSet rec;                // the records in region order
    int64_t stride;
    int32_t nvals;
    int32_t region_slots;
    int32_t nregions;
    const uint32_t* region_start; // [nregions + 1]
    uint32_t* status;
};
// LDS bytes the apply pass needs for regions of `region_slots` slots.
size_t RegionApplyLdsBytes(const KeyedSink& s, int region_slots);
hipError_t LaunchRegionApply(const RegionApplyArgs& a, hipStream_t st);
// The same records aggregated region by region in LDS, with no table of the whole grouping in HBM.  `s` carries the
// scan's own level-2 table, which in this mode holds only what the scan's workgroups merged into it as they ended:
// nregions regions of small_slots slots, chosen by the same key bits as the records' regions.  The pass leaves the groups
// of region r as rows [r * region_slots, + region_groups[r]) of a temporary table (tmp_*: key words as the tables hold
// them, accumulators, seen flags, representative rows); LaunchRegionCompactEmit then writes them to the output columns
// at region_offsets[r] (the exclusive scan of region_groups), formatted like KeyedEmitKernel's.
struct RegionEmitArgs {
    KeyedSink s;
    const DevColumn* cols;        // hashed keys: rows are compared by their key values
    RecordSet rec;                // the records in region order
    int64_t stride;
    int32_t nvals;
    int32_t region_slots;         // slots of the LDS table a region's groups are built in: a power of two, 64 .. 4096
    int32_t small_slots;          // slots per region of the scan's table (a power of two)
    int32_t nregions;
    const uint32_t* region_start; // [nregions + 1]
    uint32_t* status;
    uint32_t* region_groups;      // [nregions]
    const int64_t* region_offsets; // [nregions + 1] (the compacting kernel)
    uint64_t* tmp_lo;             // [nregions * region_slots]
    uint64_t* tmp_hi;             // two-word keys
    uint64_t* tmp_rep;            // hashed keys
    uint64_t* tmp_acc;            // [naggs][tmp_stride]
    uint32_t* tmp_seen;
    uint64_t tmp_stride;
    KeyedFinalizeArgs f;          // the compacting kernel: the output columns; f.capacity: the rows they hold (tile_offsets and t_* unused)
};
static_assert(sizeof(RegionEmitArgs) <= 4096, "kernel arguments are limited to 4 KB");
size_t RegionEmitLdsBytes(const KeyedSink& s, int region_slots);
hipError_t ConfigureRegionKernels(int maxLdsBytes);
hipError_t LaunchRegionEmit(const RegionEmitArgs& a, hipStream_t st);
hipError_t LaunchRegionCompactEmit(const RegionEmitArgs& a, hipStream_t st);
// tile_counts[t] = occupied slots among slots [t * kTileRows, (t + 1) * kTileRows); capacity is a multiple of kTileRows
hipError_t LaunchKeyedTileCount(const uint64_t* t_key_lo, uint64_t capacity, uint32_t* tile_counts, hipStream_t st);
hipError_t LaunchKeyedEmit(const KeyedFinalizeArgs& a, hipStream_t st);
// out[i] = (lo32[i] + (mid32[i] << 32) + (hi[i] << 64)) mod 2^128 as two little-endian 64-bit words.
hipError_t LaunchCombineDecimal(const uint64_t* lo32, const uint64_t* mid32, const uint64_t* hi, int64_t n, uint64_t* out,
                                hipStream_t st);
// Gather rows of a fixed-width column by 64-bit row index (idx == ~0 -> zero + invalid).
hipError_t LaunchGatherFixed(const void* src, const uint8_t* src_valid, int64_t src_valid_off, int width,
                             const uint64_t* rows, int64_t n, void* dst, uint8_t* dst_valid_bytes, hipStream_t st);
// Validity bits of the gathered rows as one byte per row (no bitmap: all valid).
hipError_t LaunchGatherValidity(const uint8_t* src_valid, int64_t src_valid_off, const uint64_t* rows, int64_t n,
                                uint8_t* dst_valid_bytes, hipStream_t st);
// out[i] = i * step
hipError_t LaunchIotaI32(int32_t* out, int64_t n, int32_t step, hipStream_t st);
// out[i] = *addrs[i] (addrs: device array of device addresses)
hipError_t LaunchGatherI32At(const int32_t* const* addrs, int64_t n, int32_t* out, hipStream_t st);
// The same for a handful of addresses, passed with the launch: out[i] = *addrs[i], i < n <= kMaxProbeWords.  `out` may be
// pinned host memory (the caller synchronises the stream and reads it: no copy of its own).
constexpr int kMaxProbeWords = 24;
struct ProbeWords {
    const int32_t* addr[kMaxProbeWords];
    int32_t n;
    int32_t* out;
};
hipError_t LaunchProbeI32(const ProbeWords& p, hipStream_t st);
hipError_t LaunchGatherStringLens(const int32_t* offsets, const uint64_t* rows, int64_t n, int32_t* out_lens, hipStream_t st);
hipError_t LaunchGatherStringBytes(const int32_t* offsets, const uint8_t* chars, const uint64_t* rows, int64_t n,
                                   const int32_t* new_offsets, uint8_t* out_chars, hipStream_t st);

// ---- the entry point that allocates nothing -----------------------------------------------------------
constexpr int64_t kNoAllocMaxRows = int64_t(1) << 26;  // 64 Mi rows: 65 536 tiles
// [tile counts: one uint32 per tile of kTileRows rows][total: 8 bytes, 256-byte aligned]
constexpr size_t kNoAllocTotalOffset = size_t(kNoAllocMaxRows / kTileRows) * 4;
constexpr size_t kNoAllocScratchBytes = kNoAllocTotalOffset + 256;
void* NoAllocScratchFor(int device, hipError_t* error);
hipError_t LaunchSumTileCounts(const uint32_t* counts, int64_t ntiles, unsigned long long* total, hipStream_t st);

// ---- dictionary-encoded columns -----------------------------------------------------------------------------
// rows[i] = code i as a row index of the dictionary, ~0 for a null or out-of-range code
hipError_t LaunchWidenCodes(const void* idx, int type, int64_t n, uint64_t entries, const uint8_t* validity, int64_t validBitOff, uint64_t* rows,
                            hipStream_t st);
// a column of n values (any fixed-width type, or bit-packed booleans) as VM payloads + validity bytes: a lookup table
hipError_t LaunchColumnToPayload(const void* values, int type, int64_t valBitOff, const uint8_t* validity, int64_t validBitOff, int64_t n,
                                 uint64_t* payload, uint8_t* valid, hipStream_t st);

// ---- batches between GPUs: pack / concat --------------------------------------------------------------------
enum CopySegmentKind : int32_t {
    SEG_BYTES = 0,          // dst[i] = src[i], count bytes
    SEG_I32_ADD = 1,        // int32 dst[i] = src[i] + arg, count values
    SEG_BITS_TO_BYTES = 2,  // dst[i] = bit (arg + i) of src, count rows
    SEG_FILL = 3,           // dst[i] = arg, count bytes
};
struct CopySegment {
    const uint8_t* src;
    uint8_t* dst;
    uint64_t count;
    int32_t kind;
    int64_t arg;
};
// Runs every segment of the device array (one grid row per segment).  maxCount: the largest `count` (sizes the grid).
hipError_t LaunchCopySegments(const CopySegment* deviceSegments, int nsegments, uint64_t maxCount, hipStream_t st);

// ---- compressed IPC bodies ---------------------------------------------------------------------------------
// One block of an LZ4 frame (https://github.com/lz4/lz4/blob/dev/doc/lz4_Frame_format.md).
struct Lz4BlockRef {
    uint64_t Src;        // offset of the block's data inside the blob
    uint32_t SrcBytes;
    uint32_t Stored;     // 1: the block is stored uncompressed
};
// One frame = one buffer of a column.
struct Lz4FrameTask {
    uint64_t first_block;  // index of the frame's first block in the block array
    uint32_t nblocks;
    uint32_t linked;       // matches may reach into the previous block
    uint8_t* dst;
    uint64_t dst_bytes;    // the length the buffer announces
};
// Decodes every frame (one wavefront each) from the blob in device memory (256-byte aligned, 64 bytes of padding
// behind it).  A frame that does not decode to its announced length raises SCAN_BAD_BLOB in *status.
hipError_t LaunchLz4Decode(const uint8_t* blob, const Lz4FrameTask* tasks, int64_t ntasks, const Lz4BlockRef* blocks, uint32_t* status, hipStream_t st);
// One sequence of a block, as Lz4ParseKernel leaves it for Lz4BatchCopyKernel.
struct Lz4Seq {
    uint32_t lit;    // literal count (low 24 bits) and the flags of lz4_kernels.hip
    uint32_t match;  // match length (low 16 bits; 0: none -- a block's last sequence) and offset (high 16)
    uint32_t lo, hi; // up to eight literals, or (literals outside the record) lo = where they start in the block
};
// What the parse found out about a block.
struct Lz4BlockSummary {
    uint32_t Sequences;    // records written (0: a stored block, or a block that is damaged)
    uint32_t Bytes;        // the bytes they decode to
    uint32_t ReachesBack;  // how far before the block's own first byte its matches read
    uint32_t pad_;
};
// The same decode in two launches: every block parsed into records (block g's at recs[recStart[g]...], room for
// SrcBytes / 3 + 2 of them) and a summary, then the copies frame by frame (a workgroup each, batches of records).
hipError_t LaunchLz4DecodeSplit(const uint8_t* blob, const Lz4FrameTask* tasks, int64_t ntasks, const Lz4BlockRef* blocks, int64_t nblocks,
                                const uint64_t* recStart, Lz4Seq* recs, Lz4BlockSummary* summaries, uint32_t* status, hipStream_t st);

constexpr int kMaxShardKeys = 8;  // key columns of a sharding hash / a predicate border

// ---- TColumnFilter on the device ----------------------------------------------------------------------
enum PredicateKind : int32_t { PK_SIGNED = 0, PK_UNSIGNED = 1, PK_F32 = 2, PK_F64 = 3, PK_BOOL = 4, PK_STRING = 5 };
struct PredicateKey {
    const void* values;          // fixed-width values, bit-packed booleans, or int32 offsets of a string column
    const uint8_t* chars;        // string columns only
    const uint8_t* border_chars; // string border (device memory)
    uint64_t border;             // border value bits; for strings its length
    int32_t kind;                // PredicateKind
    int32_t width;
    int32_t val_bit_off;
};
struct PredicateKeys {
    int32_t nkeys;
    PredicateKey cols[kMaxShardKeys];
};
// ---- k-way merge of sorted sources (merge_kernels.hip) ----------------------------------------------------
constexpr int32_t PK_DEC128 = 6;  // (a PredicateKind for the merge's comparisons only)
constexpr int kMaxMergeKeys = 12;  // key columns + version columns
struct MergeKeyColumn {
    const void* values;    // fixed-width values, or int32 offsets of a string column
    const uint8_t* chars;  // string columns only
    int32_t kind;          // PredicateKind / PK_DEC128
    int32_t width;
};
struct MergeKeys {
    int32_t nkeys, nversions;  // cols[0 .. nkeys): the sort key; cols[nkeys .. nkeys + nversions): the version, newest first
    MergeKeyColumn cols[kMaxMergeKeys];
};
// One tile of one round's output: the rows of runs in[s, m) and in[m, e) merged, from diagonal d0 on.
struct MergeTile {
    uint64_t s, m, e, d0;
};
int MergeTileRows();
// One round: every tile's part of the merge of its two runs (row ids in `in`) into `out`; splits: 2 * ntiles words of scratch.
hipError_t LaunchMergeRound(const MergeKeys& keys, const uint64_t* in, uint64_t* out, const MergeTile* deviceTiles, int64_t ntiles, uint64_t* splits,
                            hipStream_t st);
// bits: bit i set when sorted[i] is the first row of its key and keep[sorted[i]] != 0 (keep may be nullptr: all kept)
hipError_t LaunchMergeWinners(const MergeKeys& keys, const uint64_t* sorted, int64_t n, const uint8_t* keep, uint64_t* bits, hipStream_t st);
// bit r: row r is the first of a run of equal keys inside its source (the last when `reverse`); boundary[r] = 1 where a source begins
hipError_t LaunchMergeCandidates(const MergeKeys& keys, int64_t n, const uint8_t* boundary, bool reverse, uint64_t* bits, hipStream_t st);
// out[j] = number of ids (ascending, n of them) below bounds[j]
hipError_t LaunchCountBelow(const uint64_t* ids, int64_t n, const uint64_t* bounds, int nbounds, uint64_t* out, hipStream_t st);
hipError_t LaunchIotaU64(uint64_t* out, int64_t n, hipStream_t st);
hipError_t LaunchReverseU64(const uint64_t* in, int64_t n, uint64_t* out, hipStream_t st);

hipError_t LaunchExpandRuns(const uint32_t* starts, int64_t nruns, int firstValue, int64_t nbits, uint64_t* bits, hipStream_t st);
hipError_t LaunchTransitionCount(const uint64_t* bits, int64_t nbits, uint32_t* counts, hipStream_t st);
hipError_t LaunchTransitionScatter(const uint64_t* bits, int64_t nbits, const int64_t* offsets, uint32_t* starts, hipStream_t st);
hipError_t LaunchBitmapAndOr(const uint64_t* a, const uint64_t* b, int64_t nwords, int isOr, uint64_t* out, hipStream_t st);
hipError_t LaunchWordPopcount(const uint64_t* bits, int64_t nwords, uint32_t* counts, hipStream_t st);
hipError_t LaunchSequentialAnd(const uint64_t* self, int64_t nwords, const int64_t* ranks, const uint64_t* ext, int64_t extWords,
                               uint64_t* out, hipStream_t st);
// compareType: 1 LESS, 2 LESS_OR_EQUAL, 3 GREATER, 4 GREATER_OR_EQUAL (ECompareType of the reference)
hipError_t LaunchPredicate(const PredicateKeys& keys, int64_t nrows, int compareType, uint64_t* bits, hipStream_t st);
hipError_t LaunchBitmapTileCount(const uint64_t* bits, int64_t nwords, int64_t ntiles, uint32_t* counts, hipStream_t st);

// ---- compute sharding of a result batch -------------------------------------------------------------
constexpr uint32_t kMaxShards = 512;  // the partition kernels keep 16 x shards counters in LDS
struct ShardKeyColumn {
    const void* values;       // fixed-width values, bit-packed booleans, or int32 offsets of a string column
    const uint8_t* validity;  // nullptr: no nulls
    const uint8_t* chars;     // string columns only
    int32_t width;            // bytes per value; 0 = boolean
    int32_t val_bit_off;
    int32_t valid_bit_off;
};
struct ShardKeyColumns {
    int32_t nkeys;
    ShardKeyColumn cols[kMaxShardKeys];
};
// Size correction of the per-shard lists as tables in device memory (see ShardPlaceKernel).
struct ShardCorrection {
    const uint32_t* keep;       // [shards]
    const uint32_t* seg_begin;  // [shards + 1]
    const uint32_t* seg_start;  // [segments]
    const uint32_t* seg_to;     // [segments]
};
// out[row] = XXH64(seed) of the row's non-null key fields.
hipError_t LaunchXX64Rows(const ShardKeyColumns& keys, int64_t nrows, uint64_t seed, uint64_t* out, hipStream_t st);
int64_t ShardBlocks(int64_t nrows);
// out[i] = in[i * stride] for i < count (first scanned offset of every shard).
hipError_t LaunchGatherStrided(const int32_t* in, int64_t stride, int count, int32_t* out, hipStream_t st);
// counts: [shards][ShardBlocks(nrows)]; shard ids from the hash column (type T_U64 / T_I64 / T_U32 / T_I32) or from shard_ids.
hipError_t LaunchShardCount(const void* hashes, int hashType, const uint32_t* shard_ids, int64_t nrows, uint32_t shards,
                            uint32_t* shard_out, int32_t* counts, unsigned long long* totals, hipStream_t st);
// offsets: exclusive scan of counts; writes the permutation and / or the corrected shard ids.
hipError_t LaunchShardPlace(const uint32_t* shard_ids, int64_t nrows, uint32_t shards, const int32_t* offsets, uint64_t* perm,
                            uint32_t* final_shard, const ShardCorrection& corr, hipStream_t st);

// ---- synthetic data (benchmark / parity inputs generated in HBM) -----------------------------------
struct LineitemColumns {
    int32_t* l_shipdate;
    double* l_quantity;
    double* l_extendedprice;
    double* l_discount;
    double* l_tax;
    int32_t* rf_offsets;     // l_returnflag utf8 (rows + 1 offsets), 1 char per row
    uint8_t* rf_chars;
    int32_t* ls_offsets;     // l_linestatus utf8
    uint8_t* ls_chars;
    int8_t* rf_code;         // dictionary codes (0 = "A", 1 = "N", 2 = "R"), may be nullptr
    int8_t* ls_code;         // (0 = "F", 1 = "O")
};
hipError_t LaunchGenLineitem(const LineitemColumns& c, int64_t row0, int64_t nrows, uint64_t seed, hipStream_t st);
#endif  // !__HIPCC_RTC__

}  // namespace ssa
