// Keyed aggregation of many groups: what follows a scan that left its level-2 updates as records (KeyedSink.defer).
// Row-by-row updates of a table far larger than the caches cost a random access to HBM per array and row; sorted by
// region of the table, the same updates are applied by one workgroup per region inside LDS, and HBM sees streams.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "exec.h"

#define SSA_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
        }                                                                                              \
    } while (0)

namespace ssa {

namespace {
int Log2Of(uint64_t x) {
    int n = 0;
    while ((uint64_t(1) << n) < x) ++n;
    return n;
}
}  // namespace

// The records a scan left, sorted by region.  `sink`: the sink the scan ran with (record arrays; its table_mask and
// `regionShift` say which bits of a key's table position are its region number); `regions`: how many there are.
Status Executor::SortDeferredRecords(const KeyedSink& sink, int64_t rows, int nvals, bool needMask, int regionShift, uint64_t regions,
                                     SortedRecords& out, bool byCursors) {
    const bool twoWords = sink.two_words != 0, hashed = sink.hashed != 0;
    const size_t n = size_t(rows);
    const int regionBits = Log2Of(regions);
    // one round of the sort for up to 256 regions, else two of about as many bits each (the second of 6 bits at least)
    const int bitsB = regionBits <= 8 ? 0 : std::max(6, regionBits / 2);  // (fewer digits a round: longer runs per tile in the scatter)
    const int bitsA = regionBits - bitsB;
    const int digitsA = std::max(64, 1 << bitsA);
    // The first round without its counting pass: every digit's place in the sorted arrays gets room for half as many
    // records again as the mean.  The digit is a slice of a hash, but the records are not spread like balls over bins: a key
    // of many rows puts them all into one place (ClickBench UserID: the fullest of 128 digits holds 1.28 times the mean --
    // and the fullest of the 16384 regions 30 times, which is why the second round counts).  SSA_RADIX_ROOM_PERMILLE
    // (tests) gives a part of the room only.
    uint64_t permille = 1000;
    if (const char* env = std::getenv("SSA_RADIX_ROOM_PERMILLE")) permille = uint64_t(std::max(1, std::atoi(env)));
    const uint64_t meanA = uint64_t(n >> bitsA);  // (2^bitsA digits have records: digitsA may be more)
    const uint64_t capA = ((meanA + meanA / 2 + 4096) * permille / 1000 + 63) & ~uint64_t(63);
    const uint64_t roomA = capA << bitsA;
    if (bitsB == 0 || roomA >= (uint64_t(1) << 32)) byCursors = false;  // (places are 32-bit numbers)
    if (std::getenv("SSA_DEBUG_PLAN")) {
        std::fprintf(stderr, "regions: %llu regions (table mask %llx, shift %d), sort rounds of %d and %d bits%s\n", (unsigned long long)regions,
                     (unsigned long long)sink.table_mask, regionShift, bitsA, bitsB, byCursors ? ", the first without counts" : "");
    }

    const bool raw64 = sink.rec_raw64 != 0;  // (64-bit keys in one word: no hi arrays)
    const size_t partN = byCursors ? size_t(roomA) : n;
    const size_t backN = n;
    BufferPtr partLo = Buffer::Alloc(partN * 8), partHi = Buffer::Alloc(twoWords && !raw64 ? partN * 8 : 64);
    BufferPtr partVal = Buffer::Alloc(size_t(nvals) * partN * 8 + 64), partMask = Buffer::Alloc(needMask ? partN * 4 : 64);
    BufferPtr partRow = Buffer::Alloc(hashed ? partN * 4 : 64), recRow = Buffer::Alloc(hashed && bitsB > 0 ? backN * 4 : 64);
    if (!partLo || !partHi || !partVal || !partMask || !partRow || !recRow) return Status::Hip("out of device memory");
    RecordSet scanned{sink.rec_lo, sink.rec_hi, sink.rec_val, needMask ? sink.rec_mask : nullptr, nullptr};
    RecordSet part{static_cast<uint64_t*>(partLo->Ptr), static_cast<uint64_t*>(partHi->Ptr), static_cast<uint64_t*>(partVal->Ptr),
                   needMask ? static_cast<uint32_t*>(partMask->Ptr) : nullptr, hashed ? static_cast<uint32_t*>(partRow->Ptr) : nullptr};
    RecordSet back = scanned;  // the second round sorts into the scan's arrays again
    back.row = hashed ? static_cast<uint32_t*>(recRow->Ptr) : nullptr;
    out.keep = {partLo, partHi, partVal, partMask, partRow, recRow};
    if (byCursors) {  // (... unless the first went without counts: arrays of their own, and the scan's records stay as they are)
        BufferPtr backLo = Buffer::Alloc(backN * 8), backHi = Buffer::Alloc(twoWords && !raw64 ? backN * 8 : 64);
        BufferPtr backVal = Buffer::Alloc(size_t(nvals) * backN * 8 + 64), backMask = Buffer::Alloc(needMask ? backN * 4 : 64);
        if (!backLo || !backHi || !backVal || !backMask) return Status::Hip("out of device memory");
        back = RecordSet{static_cast<uint64_t*>(backLo->Ptr), static_cast<uint64_t*>(backHi->Ptr), static_cast<uint64_t*>(backVal->Ptr),
                         needMask ? static_cast<uint32_t*>(backMask->Ptr) : nullptr, hashed ? static_cast<uint32_t*>(recRow->Ptr) : nullptr};
        out.keep.insert(out.keep.end(), {backLo, backHi, backVal, backMask});
    }

    RadixArgs ra{};
    ra.src = scanned;
    ra.dst = part;
    ra.stride = rows;
    ra.dst_stride = int64_t(partN);
    ra.nvals = nvals;
    ra.two_words = twoWords;
    ra.raw64 = raw64;
    ra.present = raw64 ? sink.rec_present : nullptr;
    ra.skip_free = 1;
    ra.table_mask = sink.table_mask;
    ra.region_shift = regionShift;
    ra.digit_shift = bitsB;
    ra.ndigits = digitsA;
    ra.nseg = 1;
    // (two workgroups a CU, of the three that a tile's 47 KB of LDS allow: 256 / 384 / 512 / 640 / 768 / 1024 workgroups
    // measured 3.20 / 2.72 / 2.48 / 2.57 / 2.53 / 2.55 ms per step of a hundred million 64-bit keys)
    int64_t workgroups = 512;
    if (const char* env = std::getenv("SSA_RADIX_WORKGROUPS")) workgroups = std::max(64, std::atoi(env));  // experiments
    ra.chunks = int(std::min<int64_t>(workgroups, std::max<int64_t>(1, (rows + 16383) / 16384)));
    BufferPtr countsA = Buffer::Alloc(size_t(ra.chunks) * ra.ndigits * 4), startA = Buffer::Alloc(size_t(ra.ndigits + 1) * 4 + 64);
    if (!countsA || !startA) return Status::Hip("out of device memory");
    out.keep.insert(out.keep.end(), {countsA, startA});
    ra.seg_start = nullptr;  // (one segment: the scan's `rows` records)
    ra.counts = static_cast<uint32_t*>(countsA->Ptr);
    ra.out_start = static_cast<uint32_t*>(startA->Ptr);
    // the cursors and the overflow word: one buffer, zeroed once (digit d's place starts at d * cap: no table of starts)
    BufferPtr cursors;
    uint32_t* cursorA = nullptr;
    uint32_t* overflow = nullptr;
    if (byCursors) {
        const size_t words = size_t(digitsA) + 16;
        cursors = Buffer::Alloc(words * 4);
        if (!cursors) return Status::Hip("out of device memory");
        out.keep.push_back(cursors);
        SSA_HIP(hipMemsetAsync(cursors->Ptr, 0, words * 4, StreamHandle));
        cursorA = static_cast<uint32_t*>(cursors->Ptr);
        overflow = cursorA + digitsA;
        ra.cursor = cursorA;
        ra.bucket_cap = uint32_t(capA);
        ra.overflow = overflow;
    } else {
        SSA_HIP(LaunchRadixCount(ra, StreamHandle));
        SSA_HIP(LaunchRadixOffsets(ra, StreamHandle));
    }
    SSA_HIP(LaunchRadixScatter(ra, StreamHandle));

    out.set = part;
    out.stride = int64_t(partN);
    out.region_start = ra.out_start;
    if (byCursors) {
        out.overflow = overflow;
        out.cursors = cursorA;
        out.ncursors = size_t(digitsA);
        out.cursor_cap = uint32_t(capA);
    }
    if (bitsB > 0) {
        RadixArgs rb = ra;
        rb.src = part;
        rb.dst = back;
        rb.stride = int64_t(partN);
        rb.dst_stride = rows;
        rb.skip_free = 0;
        rb.present = nullptr;  // (the first round kept the records only)
        rb.digit_shift = 0;
        rb.ndigits = 1 << bitsB;
        rb.nseg = 1 << bitsA;
        rb.chunks = std::max(1, std::min(16, int(workgroups) / rb.nseg));
        rb.seg_start = ra.out_start;
        BufferPtr countsB = Buffer::Alloc(size_t(rb.nseg) * rb.chunks * rb.ndigits * 4);
        BufferPtr startB = Buffer::Alloc((size_t(rb.nseg) * rb.ndigits + 1) * 4 + 64);
        if (!countsB || !startB) return Status::Hip("out of device memory");
        rb.counts = static_cast<uint32_t*>(countsB->Ptr);
        rb.out_start = static_cast<uint32_t*>(startB->Ptr);
        rb.cursor = nullptr;
        rb.overflow = nullptr;
        if (byCursors) {  // (segment seg: the seg_count[seg] records from seg * capA on)
            rb.seg_count = cursorA;
            rb.seg_cap = uint32_t(capA);
        }
        SSA_HIP(LaunchRadixCount(rb, StreamHandle));
        SSA_HIP(LaunchRadixOffsets(rb, StreamHandle));
        SSA_HIP(LaunchRadixScatter(rb, StreamHandle));
        out.set = back;
        out.stride = rb.dst_stride;
        out.region_start = rb.out_start;
        out.keep.push_back(countsB);
        out.keep.push_back(startB);
    }
    return Status::OK();
}

// `cap`: slots of the table in HBM; its regions have min(cap, regionSlots) slots.  Errors of the apply pass arrive in the
// status word, like the scan's.
Status Executor::ApplyDeferredRecords(const KeyedSink& sink, const ScanArgs& args, int64_t rows, int nvals, bool needMask,
                                      int regionSlots, uint64_t cap) {
    const bool hashed = sink.hashed != 0;
    const uint64_t R = std::min<uint64_t>(cap, uint64_t(regionSlots));
    const uint64_t regions = cap / R;
    SortedRecords sorted;
    const Status st = SortDeferredRecords(sink, rows, nvals, needMask, Log2Of(R), regions, sorted);
    if (!st.ok()) {
        (void)hipStreamSynchronize(StreamHandle);
        return st;
    }
    RegionApplyArgs aa{};
    aa.s = sink;
    BufferPtr devCols;
    if (hashed) {
        devCols = Buffer::Alloc(sizeof(args.cols));
        if (!devCols) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(devCols->Ptr, args.cols, sizeof(args.cols), hipMemcpyHostToDevice, StreamHandle));
        aa.cols = static_cast<const DevColumn*>(devCols->Ptr);
    }
    aa.rec = sorted.set;
    aa.stride = rows;
    aa.nvals = nvals;
    aa.region_slots = int32_t(R);
    aa.nregions = int32_t(regions);
    aa.region_start = sorted.region_start;
    aa.status = static_cast<uint32_t*>(StatusWord->Ptr);
    SSA_HIP(LaunchRegionApply(aa, StreamHandle));
    // (the buffers of this function are freed when it returns: the stream has to be past their last use)
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

// The sorted records aggregated region by region in LDS (RegionEmitKernel / RegionEmitExactKernel); the groups of the
// regions are left in a temporary table, counted and their places in the output scanned.  `sink`: the scan's sink, whose
// table holds `regions` regions of `smallSlots` slots; `ldsSlots`: slots of the LDS table of a region.  The number of
// groups is on its way into *groupsAt (pinned host memory, or read after the next synchronisation of the stream).
Status Executor::AggregateDeferredRecords(const KeyedSink& sink, const ScanArgs& args, const SortedRecords& sorted, int64_t rows, int nvals,
                                          int ldsSlots, int smallSlots, uint64_t regions, RegionGroups& out, int64_t* groupsAt) {
    RegionEmitArgs& ea = out.args;
    ea = RegionEmitArgs{};
    ea.s = sink;
    if (const char* env = std::getenv("SSA_REGION_DEBUG")) ea.s.pad2_ = std::atoi(env);  // experiments: parts of the pass left out
    const bool twoWords = sink.two_words != 0, hashed = sink.hashed != 0;
    if (hashed) {
        BufferPtr devCols = Buffer::Alloc(sizeof(args.cols));
        if (!devCols) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(devCols->Ptr, args.cols, sizeof(args.cols), hipMemcpyHostToDevice, StreamHandle));
        ea.cols = static_cast<const DevColumn*>(devCols->Ptr);
        out.keep.push_back(devCols);
    }
    const size_t slots = size_t(regions) * size_t(ldsSlots);
    BufferPtr tmpLo = Buffer::Alloc(slots * 8), tmpHi = Buffer::Alloc(twoWords ? slots * 8 : 64), tmpRep = Buffer::Alloc(hashed ? slots * 8 : 64);
    BufferPtr tmpAcc = Buffer::Alloc(size_t(sink.naggs) * slots * 8 + 64), tmpSeen = Buffer::Alloc(slots * 4);
    BufferPtr counts = Buffer::Alloc(size_t(regions) * 4 + 64);
    BufferPtr offsets = Buffer::Alloc((size_t(regions) + 1 + (size_t(regions) + 4095) / 4096 + 2) * 8 + 64), total = Buffer::Alloc(256);
    if (!tmpLo || !tmpHi || !tmpRep || !tmpAcc || !tmpSeen || !counts || !offsets || !total) return Status::Hip("out of device memory");
    out.keep.insert(out.keep.end(), {tmpLo, tmpHi, tmpRep, tmpAcc, tmpSeen, counts, offsets, total});
    SSA_HIP(hipMemsetAsync(counts->Ptr, 0, size_t(regions) * 4, StreamHandle));  // (a region without records leaves no count)
    ea.rec = sorted.set;
    ea.stride = sorted.stride != 0 ? sorted.stride : rows;
    ea.nvals = nvals;
    ea.region_slots = ldsSlots;
    ea.small_slots = smallSlots;
    ea.nregions = int32_t(regions);
    ea.region_start = sorted.region_start;
    ea.status = static_cast<uint32_t*>(StatusWord->Ptr);
    ea.region_groups = static_cast<uint32_t*>(counts->Ptr);
    ea.region_offsets = static_cast<const int64_t*>(offsets->Ptr);
    ea.tmp_lo = static_cast<uint64_t*>(tmpLo->Ptr);
    ea.tmp_hi = static_cast<uint64_t*>(tmpHi->Ptr);
    ea.tmp_rep = static_cast<uint64_t*>(tmpRep->Ptr);
    ea.tmp_acc = static_cast<uint64_t*>(tmpAcc->Ptr);
    ea.tmp_seen = static_cast<uint32_t*>(tmpSeen->Ptr);
    ea.tmp_stride = slots;
    SSA_HIP(LaunchRegionEmit(ea, StreamHandle));
    SSA_HIP(LaunchTileScan(ea.region_groups, int64_t(regions), static_cast<int64_t*>(offsets->Ptr), static_cast<int64_t*>(total->Ptr), StreamHandle));
    SSA_HIP(hipMemcpyAsync(groupsAt, total->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    if (std::getenv("SSA_REGION_TIMING")) {  // diagnostics build only (-DSSA_REGION_TIMING): ticks per phase, summed over workgroups
        unsigned long long t[8];
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        SSA_HIP(hipMemcpy(t, static_cast<char*>(StatusWord->Ptr) + 128, 64, hipMemcpyDeviceToHost));
        SSA_HIP(hipMemset(static_cast<char*>(StatusWord->Ptr) + 128, 0, 64));
        std::fprintf(stderr, "region pass ticks: first loads %llu, vote %llu, set-up %llu, seeds %llu, records %llu, vote %llu, count %llu, rows out %llu\n", t[0], t[1],
                     t[2], t[3], t[4], t[5], t[6], t[7]);
    }
    return Status::OK();
}

// The groups the region pass left, to the output columns `fa` describes (room for all of them).
Status Executor::EmitRegionGroups(RegionGroups& groups, const KeyedFinalizeArgs& fa) {
    groups.args.f = fa;
    SSA_HIP(LaunchRegionCompactEmit(groups.args, StreamHandle));
    return Status::OK();
}

}  // namespace ssa
