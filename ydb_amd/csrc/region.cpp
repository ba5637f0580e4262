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
                                     SortedRecords& out) {
    const bool twoWords = sink.two_words != 0, hashed = sink.hashed != 0;
    const size_t n = size_t(rows);
    const int regionBits = Log2Of(regions);
    // one round of the sort for up to 256 regions, else two of about as many bits each (the second of 6 bits at least)
    const int bitsB = regionBits <= 8 ? 0 : std::max(6, regionBits / 2);  // (fewer digits a round: longer runs per tile in the scatter)
    const int bitsA = regionBits - bitsB;
    if (std::getenv("SSA_DEBUG_PLAN")) {
        std::fprintf(stderr, "regions: %llu regions (table mask %llx, shift %d), sort rounds of %d and %d bits\n", (unsigned long long)regions,
                     (unsigned long long)sink.table_mask, regionShift, bitsA, bitsB);
    }

    const bool raw64 = sink.rec_raw64 != 0;  // (64-bit keys in one word: no hi arrays)
    BufferPtr partLo = Buffer::Alloc(n * 8), partHi = Buffer::Alloc(twoWords && !raw64 ? n * 8 : 64);
    BufferPtr partVal = Buffer::Alloc(size_t(nvals) * n * 8 + 64), partMask = Buffer::Alloc(needMask ? n * 4 : 64);
    BufferPtr partRow = Buffer::Alloc(hashed ? n * 4 : 64), recRow = Buffer::Alloc(hashed && bitsB > 0 ? n * 4 : 64);
    if (!partLo || !partHi || !partVal || !partMask || !partRow || !recRow) return Status::Hip("out of device memory");
    RecordSet scanned{sink.rec_lo, sink.rec_hi, sink.rec_val, needMask ? sink.rec_mask : nullptr, nullptr};
    RecordSet part{static_cast<uint64_t*>(partLo->Ptr), static_cast<uint64_t*>(partHi->Ptr), static_cast<uint64_t*>(partVal->Ptr),
                   needMask ? static_cast<uint32_t*>(partMask->Ptr) : nullptr, hashed ? static_cast<uint32_t*>(partRow->Ptr) : nullptr};
    RecordSet back = scanned;  // the second round sorts into the scan's arrays again
    back.row = hashed ? static_cast<uint32_t*>(recRow->Ptr) : nullptr;

    RadixArgs ra{};
    ra.src = scanned;
    ra.dst = part;
    ra.stride = rows;
    ra.nvals = nvals;
    ra.two_words = twoWords;
    ra.raw64 = raw64;
    ra.present = raw64 ? sink.rec_present : nullptr;
    ra.skip_free = 1;
    ra.table_mask = sink.table_mask;
    ra.region_shift = regionShift;
    ra.digit_shift = bitsB;
    ra.ndigits = std::max(64, 1 << bitsA);
    ra.nseg = 1;
    ra.chunks = int(std::min<int64_t>(768, std::max<int64_t>(1, (rows + 16383) / 16384)));  // (three workgroups of the scatter fit a CU)
    BufferPtr segA = Buffer::Alloc(64), countsA = Buffer::Alloc(size_t(ra.chunks) * ra.ndigits * 4), startA = Buffer::Alloc(size_t(ra.ndigits + 1) * 4 + 64);
    if (!segA || !countsA || !startA) return Status::Hip("out of device memory");
    const uint32_t whole[2] = {0u, uint32_t(rows)};
    SSA_HIP(hipMemcpyAsync(segA->Ptr, whole, sizeof(whole), hipMemcpyHostToDevice, StreamHandle));
    ra.seg_start = static_cast<const uint32_t*>(segA->Ptr);
    ra.counts = static_cast<uint32_t*>(countsA->Ptr);
    ra.out_start = static_cast<uint32_t*>(startA->Ptr);
    SSA_HIP(LaunchRadixCount(ra, StreamHandle));
    SSA_HIP(LaunchRadixOffsets(ra, StreamHandle));
    SSA_HIP(LaunchRadixScatter(ra, StreamHandle));

    out.set = part;
    out.region_start = ra.out_start;
    out.keep = {partLo, partHi, partVal, partMask, partRow, recRow, segA, countsA, startA};
    if (bitsB > 0) {
        RadixArgs rb = ra;
        rb.src = part;
        rb.dst = back;
        rb.skip_free = 0;
        rb.present = nullptr;  // (the first round kept the records only)
        rb.digit_shift = 0;
        rb.ndigits = 1 << bitsB;
        rb.nseg = 1 << bitsA;
        rb.chunks = std::max(1, std::min(16, 768 / rb.nseg));
        rb.seg_start = ra.out_start;
        BufferPtr countsB = Buffer::Alloc(size_t(rb.nseg) * rb.chunks * rb.ndigits * 4);
        BufferPtr startB = Buffer::Alloc((size_t(rb.nseg) * rb.ndigits + 1) * 4 + 64);
        if (!countsB || !startB) return Status::Hip("out of device memory");
        rb.counts = static_cast<uint32_t*>(countsB->Ptr);
        rb.out_start = static_cast<uint32_t*>(startB->Ptr);
        SSA_HIP(LaunchRadixCount(rb, StreamHandle));
        SSA_HIP(LaunchRadixOffsets(rb, StreamHandle));
        SSA_HIP(LaunchRadixScatter(rb, StreamHandle));
        out.set = back;
        out.region_start = rb.out_start;
        out.keep.push_back(countsB);
        out.keep.push_back(startB);
    }
    // (the whole stack of whole[] is read by the copy above: the caller synchronises the stream before it returns)
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

// The sorted records aggregated region by region in LDS and written straight into the output columns `fa` describes
// (RegionEmitKernel).  `sink`: the scan's sink, whose table holds `regions` regions of `smallSlots` slots; `ldsSlots`: slots of
// the LDS table of a region.  *groups: the number of groups -- more than fa.capacity when the columns were too short (nothing
// was written then, and the call can be repeated with longer columns).  Ends with the stream synchronised.
Status Executor::EmitDeferredRecords(const KeyedSink& sink, const ScanArgs& args, const SortedRecords& sorted, int64_t rows, int nvals,
                                     int ldsSlots, int smallSlots, uint64_t regions, const KeyedFinalizeArgs& fa, int64_t* groups) {
    RegionEmitArgs ea{};
    ea.s = sink;
    BufferPtr devCols;
    if (sink.hashed) {
        devCols = Buffer::Alloc(sizeof(args.cols));
        if (!devCols) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(devCols->Ptr, args.cols, sizeof(args.cols), hipMemcpyHostToDevice, StreamHandle));
        ea.cols = static_cast<const DevColumn*>(devCols->Ptr);
    }
    BufferPtr counter = Buffer::Alloc(256);
    if (!counter) return Status::Hip("out of device memory");
    SSA_HIP(hipMemsetAsync(counter->Ptr, 0, 8, StreamHandle));
    ea.rec = sorted.set;
    ea.stride = rows;
    ea.nvals = nvals;
    ea.region_slots = ldsSlots;
    ea.small_slots = smallSlots;
    ea.nregions = int32_t(regions);
    ea.region_start = sorted.region_start;
    ea.status = static_cast<uint32_t*>(StatusWord->Ptr);
    ea.f = fa;
    ea.out_count = static_cast<unsigned long long*>(counter->Ptr);
    SSA_HIP(LaunchRegionEmit(ea, StreamHandle));
    int64_t* at = HostWords ? reinterpret_cast<int64_t*>(HostWords + 8) : groups;
    SSA_HIP(hipMemcpyAsync(at, counter->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    *groups = *at;
    return Status::OK();
}

}  // namespace ssa
