// Group-by kernels of the library build: the keyed scan (keyed_scan.h) in its generic form, and the kernels that turn
// the hash table into output columns.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"
#include "keyed_scan.h"

namespace ssa {

namespace {

// ---- level-2 table set-up ----------------------------------------------------------------------------
// Every array of the table in one launch: keys free, hi words unpublished, accumulators at their identities, seen
// flags clear, representative rows none.
struct KeyedTableInitArgs {
    uint64_t capacity;
    uint64_t* t_key_lo;
    uint64_t* t_key_hi;   // nullptr: one-word keys
    uint64_t* t_acc;
    uint32_t* t_seen;
    uint64_t* t_rep_row;  // nullptr: exact keys
    int32_t naggs;
    uint64_t ident[kMaxAggs];
};
__global__ __launch_bounds__(kBlock) void KeyedTableInitKernel(const KeyedTableInitArgs a) {
    for (uint64_t i = uint64_t(blockIdx.x) * kBlock + threadIdx.x; i < a.capacity; i += uint64_t(gridDim.x) * kBlock) {
        a.t_key_lo[i] = kEmpty;
        if (a.t_key_hi != nullptr) a.t_key_hi[i] = kUnpublished;
        if (a.t_rep_row != nullptr) a.t_rep_row[i] = ~0ull;
        a.t_seen[i] = 0;
        for (int k = 0; k < a.naggs; ++k) a.t_acc[uint64_t(k) * a.capacity + i] = a.ident[k];
    }
}

// ---- deferred level-2 updates ------------------------------------------------------------------------
// Digit of a record in one round of the sort: bits of the number of the region its probe sequence starts in.
__device__ __forceinline__ uint32_t DigitOf(const RadixArgs& a, uint64_t lo, uint64_t hi) {
    const uint64_t start = TablePos(lo, hi) & a.table_mask;
    return uint32_t(start >> a.region_shift >> a.digit_shift) & uint32_t(a.ndigits - 1);
}

constexpr int kRadixPerThread = 16;
constexpr int kRadixTile = kBlock * kRadixPerThread;

// The records a workgroup takes: chunk `blockIdx.x % chunks` of segment `blockIdx.x / chunks`, whole tiles.
__device__ __forceinline__ void RadixChunk(const RadixArgs& a, uint32_t& begin, uint32_t& end) {
    const int seg = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
    // (a segment left by a round with cursors: at seg * seg_cap, seg_count[seg] records of it)
    // (no seg_start at all: one segment, the `stride` records of src)
    const bool whole = a.seg_count == nullptr && a.seg_start == nullptr;
    const uint32_t s0 = whole ? 0u : (a.seg_count != nullptr ? uint32_t(seg) * a.seg_cap : a.seg_start[seg]);
    const uint32_t s1 = whole ? uint32_t(a.stride)
                              : (a.seg_count != nullptr ? s0 + (a.seg_count[seg] < a.seg_cap ? a.seg_count[seg] : a.seg_cap) : a.seg_start[seg + 1]);
    uint32_t per = (s1 - s0 + uint32_t(a.chunks) - 1) / uint32_t(a.chunks);
    per = (per + kRadixTile - 1) / kRadixTile * kRadixTile;
    const uint64_t b = uint64_t(s0) + uint64_t(chunk) * per;
    begin = b < s1 ? uint32_t(b) : s1;
    end = b + per < s1 ? uint32_t(b + per) : s1;
}

// (four times the scatter's threads on the same chunk, every 256 of them a tile at a time: the pass only reads, and a chunk's
// one workgroup -- two a CU -- had too few loads in flight to keep the memory busy: 232 us for 0.8 GB)
constexpr int kCountThreads = 4 * kBlock;
__global__ __launch_bounds__(kCountThreads) void RadixCountKernel(const RadixArgs) {
    const RadixArgs& a = KernArgsAs<RadixArgs>();
    __shared__ uint32_t lHist[256];
    const uint32_t tid = threadIdx.x & uint32_t(kBlock - 1), sub = threadIdx.x / uint32_t(kBlock);
    if (threadIdx.x < 256) lHist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t begin, end;
    RadixChunk(a, begin, end);
    const bool two = a.two_words != 0, raw = a.raw64 != 0;
    for (uint64_t t0 = uint64_t(begin) + uint64_t(sub) * kRadixTile; t0 < end; t0 += uint64_t(kCountThreads / kBlock) * kRadixTile) {
        const uint32_t r0 = uint32_t(t0);
        // (loads without conditions, so that all of them are in flight together: a row past the end reads the last one)
        uint64_t lo[kRadixPerThread], hi[kRadixPerThread];
#pragma unroll
        for (int u = 0; u < kRadixPerThread; ++u) {
            const uint32_t r = r0 + u * kBlock + tid;
            lo[u] = a.src.lo[r < end ? r : end - 1];
            hi[u] = 0;
        }
        if (two && !raw) {
#pragma unroll
            for (int u = 0; u < kRadixPerThread; ++u) {
                const uint32_t r = r0 + u * kBlock + tid;
                hi[u] = a.src.hi[r < end ? r : end - 1];
            }
        }
        if (raw) {
            // 64-bit keys in one word: the table's two words from it; which indices hold a record at all, the first
            // round reads from the bitmap (64 consecutive records are one wave's load: one word of it)
            // (the words of the bitmap like the keys: loaded without conditions, all of them in flight together -- a load
            // and a wait per row made a tile sixteen round trips to memory)
            uint64_t pres[kRadixPerThread];
#pragma unroll
            for (int u = 0; u < kRadixPerThread; ++u) pres[u] = ~0ull;
            if (a.present != nullptr) {
#pragma unroll
                for (int u = 0; u < kRadixPerThread; ++u) {
                    const uint32_t r = r0 + u * kBlock + tid;
                    pres[u] = a.present[(r < end ? r : end - 1) >> 6];
                }
            }
#pragma unroll
            for (int u = 0; u < kRadixPerThread; ++u) {
                const uint32_t r = r0 + u * kBlock + tid;
                const bool there = r < end && ((pres[u] >> (r & 63u)) & 1ull) != 0;
                uint64_t l, h;
                RawKeyWords(lo[u], l, h);
                lo[u] = there ? l : kEmpty;
                hi[u] = h;
            }
        }
#pragma unroll
        for (int u = 0; u < kRadixPerThread; ++u) {
            if (r0 + u * kBlock + tid >= end) lo[u] = kEmpty;
        }
#pragma unroll
        for (int u = 0; u < kRadixPerThread; ++u) {
            if (lo[u] != kEmpty) atomicAdd(lHist + DigitOf(a, lo[u], hi[u]), 1u);
        }
    }
    __syncthreads();
    if (int(threadIdx.x) < a.ndigits) a.counts[size_t(blockIdx.x) * a.ndigits + threadIdx.x] = lHist[threadIdx.x];
}

// counts[seg][c][d] -> records of (seg, d) in the chunks before c; out_start[seg * ndigits + d + 1] <- records of
// (seg, d), scanned by the next kernel.  A workgroup takes 64 digits of one segment; its waves share the chunks.
constexpr int kOffsetWaves = 16;
__global__ __launch_bounds__(64 * kOffsetWaves) void RadixColumnScanKernel(const RadixArgs) {
    const RadixArgs& a = KernArgsAs<RadixArgs>();
    __shared__ uint32_t lSum[kOffsetWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = a.ndigits / 64;
    const int seg = blockIdx.x / groups, d = (blockIdx.x % groups) * 64 + lane;
    uint32_t* col = a.counts + size_t(seg) * a.chunks * a.ndigits + d;
    const int per = (a.chunks + kOffsetWaves - 1) / kOffsetWaves;
    const int c0 = wave * per, c1 = (c0 + per < a.chunks) ? c0 + per : a.chunks;
    uint32_t sum = 0;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) sum += col[size_t(c) * a.ndigits];
    lSum[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (int q = 0; q < kOffsetWaves; ++q) {
        if (q < wave) run += lSum[q][lane];
        total += lSum[q][lane];
    }
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        const uint32_t n = col[size_t(c) * a.ndigits];
        col[size_t(c) * a.ndigits] = run;
        run += n;
    }
    if (wave == 0) a.out_start[size_t(seg) * a.ndigits + d + 1] = total;
}

// out_start[1 ..]: counts -> positions (inclusive scan in place; out_start[0] = first record of the source).  One workgroup.
__global__ __launch_bounds__(1024) void RadixStartScanKernel(const RadixArgs) {
    const RadixArgs& a = KernArgsAs<RadixArgs>();
    __shared__ uint32_t lPart[1024];
    const int n = a.nseg * a.ndigits;
    const int per = (n + 1023) / 1024;
    const int i0 = int(threadIdx.x) * per;
    uint32_t sum = 0;
    for (int i = i0; i < i0 + per && i < n; ++i) sum += a.out_start[i + 1];
    lPart[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = int(threadIdx.x) >= off ? lPart[threadIdx.x - off] : 0u;
        __syncthreads();
        lPart[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = lPart[threadIdx.x] - sum;
    for (int i = i0; i < i0 + per && i < n; ++i) {
        run += a.out_start[i + 1];
        a.out_start[i + 1] = run;
    }
    if (threadIdx.x == 0) a.out_start[0] = 0;
}

// The records to their places.  A tile of the chunk is first ordered by digit inside LDS (the first key words and the
// records' indices in the tile), then written out in that order: neighbouring lanes then write neighbouring records of
// one (segment, digit), whole lines at a time.  kRaw: 64-bit keys in one word (RadixArgs.raw64).
// (Raw keys were once kept out of LDS and read a second time where they were written out -- from the cache, mostly, but
// the counters saw 1.35 GB read per round for 0.8 GB of records, and the second round trip to memory in every tile cost
// more than the 32 KB of LDS: 615 -> 553 us a round for a hundred million keys.)
template <bool kRaw, int kPer>
__global__ __launch_bounds__(kBlock) void RadixScatterKernel(const RadixArgs) {
    constexpr int kTile = kBlock * kPer;  // (a chunk is whole tiles of kTile records: its last tile here may be half of one)
    const RadixArgs& a = KernArgsAs<RadixArgs>();
    __shared__ uint32_t lCursor[256], lCount[256], lBase[256 + 1];
    __shared__ uint32_t lWave[kBlock / 64];
    __shared__ uint64_t lSortedLo[kTile];
    __shared__ uint16_t lOrder[kTile];
    __shared__ uint8_t lDigit[kTile];
    const int seg = blockIdx.x / a.chunks;
    const bool more = (a.two_words != 0 && a.raw64 == 0) || a.dst.mask != nullptr || a.dst.row != nullptr || a.nvals > 0;  // more than the lo word to move
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lCount[threadIdx.x] = 0;
    const bool byCursor = a.cursor != nullptr;
    const int64_t dstStride = a.dst_stride != 0 ? a.dst_stride : a.stride;
    if (int(threadIdx.x) < a.ndigits && !byCursor) {
        lCursor[threadIdx.x] = a.out_start[size_t(seg) * a.ndigits + threadIdx.x] + a.counts[size_t(blockIdx.x) * a.ndigits + threadIdx.x];
    }
    __syncthreads();
    uint32_t begin, end;
    RadixChunk(a, begin, end);
    const bool two = a.two_words != 0;
    constexpr bool raw = kRaw;
    for (uint32_t r0 = begin; r0 < end; r0 += kTile) {
        // (loads without conditions, so that all of them are in flight together: a row past the end reads the last one)
        uint64_t lo[kPer], hi[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t r = r0 + u * kBlock + threadIdx.x;
            lo[u] = a.src.lo[r < end ? r : end - 1];
            hi[u] = 0;
        }
        if (two && !raw) {
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const uint32_t r = r0 + u * kBlock + threadIdx.x;
                hi[u] = a.src.hi[r < end ? r : end - 1];
            }
        }
        if (raw) {
            // 64-bit keys in one word: the table's two words from it; which indices hold a record at all, the first
            // round reads from the bitmap (64 consecutive records are one wave's load: one word of it)
            // (the words of the bitmap like the keys: loaded without conditions, all of them in flight together)
            uint64_t pres[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) pres[u] = ~0ull;
            if (a.present != nullptr) {
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const uint32_t r = r0 + u * kBlock + threadIdx.x;
                    pres[u] = a.present[(r < end ? r : end - 1) >> 6];
                }
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const uint32_t r = r0 + u * kBlock + threadIdx.x;
                const bool there = r < end && ((pres[u] >> (r & 63u)) & 1ull) != 0;
                uint64_t l, h;
                RawKeyWords(lo[u], l, h);
                lo[u] = there ? l : kEmpty;
                hi[u] = h;
            }
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            if (r0 + u * kBlock + threadIdx.x >= end) lo[u] = kEmpty;
        }
        uint32_t digit[kPer], rank[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            digit[u] = lo[u] != kEmpty ? DigitOf(a, lo[u], hi[u]) : 0xFFFFFFFFu;
            rank[u] = lo[u] != kEmpty ? atomicAdd(lCount + digit[u], 1u) : 0u;
        }
        __syncthreads();
        // first position of each digit inside the ordered tile: exclusive scan of lCount (256 entries, one wave's worth each)
        {
            const uint32_t c = lCount[threadIdx.x];
            if (byCursor && int(threadIdx.x) < a.ndigits) {
                // the tile's records of this digit take the next c places of the digit's room in dst; those past the room's
                // end are not written (the sort is void then, but every place the cursors count holds a record)
                const size_t bucket = size_t(seg) * a.ndigits + threadIdx.x;
                const uint32_t before = c != 0 ? atomicAdd(a.cursor + bucket, c) : 0u;
                if (before + c > a.bucket_cap) atomicOr(a.overflow, 1u);
                lCursor[threadIdx.x] = uint32_t(bucket) * a.bucket_cap + (before < a.bucket_cap ? before : a.bucket_cap);
            }
            uint32_t inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (lane == 63) lWave[wave] = inc;
            __syncthreads();
            uint32_t before = 0;
            for (int w = 0; w < wave; ++w) before += lWave[w];
            lBase[threadIdx.x] = before + inc - c;
            if (threadIdx.x == kBlock - 1) lBase[256] = before + inc;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            if (digit[u] == 0xFFFFFFFFu) continue;
            const uint32_t at = lBase[digit[u]] + rank[u];
            lSortedLo[at] = raw ? (lo[u] | (hi[u] << 63)) : lo[u];  // (a raw key: the word as it came)
            lOrder[at] = uint16_t(u * kBlock + threadIdx.x);
            lDigit[at] = uint8_t(digit[u]);
        }
        __syncthreads();
        const uint32_t n = lBase[256];
        // first place past the room of digit d (places by cursors)
        const uint32_t room0 = (uint32_t(seg) * uint32_t(a.ndigits) + 1u) * a.bucket_cap;
        auto fits = [&](uint32_t d, uint32_t at) { return !byCursor || at < room0 + d * a.bucket_cap; };
        if (raw && !more) {
            // One word a record: the places and words of a thread's kPer records first, then the writes
            uint64_t word[kPer];
            uint32_t to[kPer];
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const uint32_t p = uint32_t(j) * kBlock + threadIdx.x;
                const bool there = p < n;
                const uint32_t d = there ? lDigit[p] : 0u;
                to[j] = there ? lCursor[d] + (p - lBase[d]) : 0xFFFFFFFFu;
                if (!fits(d, to[j])) to[j] = 0xFFFFFFFFu;
                word[j] = lSortedLo[there ? p : 0u];
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                if (to[j] != 0xFFFFFFFFu) a.dst.lo[to[j]] = word[j];
            }
        } else {
            // Records of several words: four of a thread's records at a time, and of each array the four reads first, in
            // flight together, then the four writes (a read and a write per record and array is a round trip to memory
            // each: the compiler keeps a read behind the write before it, the arrays might overlap).
            constexpr int kGroup = 4;
            for (int j0 = 0; j0 < kPer && uint32_t(j0) * kBlock < n; j0 += kGroup) {
                uint32_t to[kGroup], r[kGroup];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) {
                    const uint32_t p = uint32_t(j0 + j) * kBlock + threadIdx.x;
                    const bool there = p < n;
                    const uint32_t d = there ? lDigit[p] : 0u;
                    to[j] = there ? lCursor[d] + (p - lBase[d]) : 0xFFFFFFFFu;
                    if (!fits(d, to[j])) to[j] = 0xFFFFFFFFu;
                    r[j] = r0 + (there ? uint32_t(lOrder[p]) : 0u);  // (a record of the tile in any case: reads need no condition)
                    if (to[j] != 0xFFFFFFFFu) a.dst.lo[to[j]] = lSortedLo[p];
                }
                if (two && !raw) {
                    uint64_t w[kGroup];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) w[j] = a.src.hi[r[j]];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        if (to[j] != 0xFFFFFFFFu) a.dst.hi[to[j]] = w[j];
                    }
                }
                if (a.dst.mask != nullptr) {
                    uint32_t w[kGroup];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) w[j] = a.src.mask[r[j]];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        if (to[j] != 0xFFFFFFFFu) a.dst.mask[to[j]] = w[j];
                    }
                }
                if (a.dst.row != nullptr) {
                    uint32_t w[kGroup];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) w[j] = a.src.row != nullptr ? a.src.row[r[j]] : r[j];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        if (to[j] != 0xFFFFFFFFu) a.dst.row[to[j]] = w[j];
                    }
                }
                for (int v = 0; v < a.nvals; ++v) {
                    const uint64_t* from = a.src.val + size_t(v) * size_t(a.stride);
                    uint64_t* into = a.dst.val + size_t(v) * size_t(dstStride);
                    uint64_t w[kGroup];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) w[j] = from[r[j]];
#pragma unroll
                    for (int j = 0; j < kGroup; ++j) {
                        if (to[j] != 0xFFFFFFFFu) into[to[j]] = w[j];
                    }
                }
            }
        }
        __syncthreads();
        if (!byCursor) lCursor[threadIdx.x] += lCount[threadIdx.x];
        lCount[threadIdx.x] = 0;
        __syncthreads();
    }
}

// One region of the table brought up to date by one workgroup: the region's slots are read into LDS, the region's
// records are applied there, the slots are written back.  No other workgroup touches the region meanwhile.
__global__ __launch_bounds__(kBlock) void RegionApplyKernel(const RegionApplyArgs) {
    const RegionApplyArgs& a = KernArgsAs<RegionApplyArgs>();
    const KeyedSink& s = a.s;
    extern __shared__ uint64_t lRegion[];
    const int R = a.region_slots;
    const bool two = s.two_words != 0, hashed = s.hashed != 0;
    uint64_t* lLo = lRegion;
    uint64_t* lHi = lLo + R;
    uint64_t* lRep = lHi + (two ? R : 0);
    uint64_t* lAcc = lRep + (hashed ? R : 0);
    uint32_t* lSeen = reinterpret_cast<uint32_t*>(lAcc + size_t(s.naggs) * R);
    __shared__ uint32_t lStop;
    uint32_t err = 0;
    for (int region = blockIdx.x; region < a.nregions; region += gridDim.x) {
        const uint32_t first = a.region_start[region], last = a.region_start[region + 1];
        if (first == last) continue;  // (the same for the whole workgroup)
        // a region that ran full somewhere: the host repeats the step with a larger table, the rest is wasted work
        if (threadIdx.x == 0) lStop = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SCAN_TABLE_FULL;
        __syncthreads();
        if (lStop != 0) break;
        const uint64_t slot0 = uint64_t(region) * R;
        // (R is a multiple of the workgroup size; four loads of an array in flight per thread)
        auto load = [&](uint64_t* dst, const uint64_t* src) {
            for (int i0 = threadIdx.x; i0 < R; i0 += 4 * kBlock) {
                uint64_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = src[i0 + j * kBlock < R ? i0 + j * kBlock : i0];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (i0 + j * kBlock < R) dst[i0 + j * kBlock] = w[j];
                }
            }
        };
        // The keys first: nearly all slots are still free (what the table holds before this pass is what the workgroups'
        // own tables held at the end of the scan), and a free slot's other words are known without reading them.
        load(lLo, s.t_key_lo + slot0);
        __syncthreads();
        for (int i = threadIdx.x; i < R; i += kBlock) {
            const bool taken = lLo[i] != kEmpty;
            if (two) lHi[i] = taken ? s.t_key_hi[slot0 + i] : kUnpublished;
            if (hashed) lRep[i] = taken ? s.t_rep_row[slot0 + i] : ~0ull;
            lSeen[i] = taken ? s.t_seen[slot0 + i] : 0u;
            for (int k = 0; k < s.naggs; ++k) {
                lAcc[size_t(k) * R + i] = taken ? s.t_acc[size_t(k) * s.acc_stride + slot0 + i] : TableIdentity(s.aggs[k].kind);
            }
        }
        __syncthreads();
        for (uint32_t i0 = first + threadIdx.x; i0 < last; i0 += 4 * kBlock) {
          uint64_t los[4], his[4];
          uint32_t masks[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
              const uint32_t i = i0 + j * kBlock < last ? i0 + j * kBlock : i0;
              los[j] = a.rec.lo[i];
              his[j] = 0;
              masks[j] = ~0u;
          }
          if (two && !s.rec_raw64) {
#pragma unroll
              for (int j = 0; j < 4; ++j) his[j] = a.rec.hi[i0 + j * kBlock < last ? i0 + j * kBlock : i0];
          }
          if (s.rec_raw64) {
#pragma unroll
              for (int j = 0; j < 4; ++j) RawKeyWords(los[j], los[j], his[j]);
          }
          if (a.rec.mask != nullptr) {
#pragma unroll
              for (int j = 0; j < 4; ++j) masks[j] = a.rec.mask[i0 + j * kBlock < last ? i0 + j * kBlock : i0];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t i = i0 + j * kBlock;
            if (i >= last || err != 0) continue;
            const uint64_t lo = los[j], hi = his[j];
            const uint32_t mask = masks[j];
            const uint64_t start = TablePos(lo, hi);
            int64_t g;
            if (hashed && (hi >> 63) != 0) {
                const uint64_t row = a.rec.row[i];
                g = TableUpsertVerified(lLo, lHi, lRep, uint64_t(R - 1), uint64_t(R - 1), start, lo, hi, row, uint32_t(R),
                                        [&](uint64_t rep) { return SameKeyRows<4>(a.cols, s, row, rep); });
            } else {
                g = TableUpsert(lLo, lHi, uint64_t(R - 1), uint64_t(R - 1), start, lo, hi, two, uint32_t(R));
                if (hashed && g >= 0) atomicMin(reinterpret_cast<unsigned long long*>(lRep + g), static_cast<unsigned long long>(a.rec.row[i]));
            }
            if (g < 0) {
                err |= SCAN_TABLE_FULL;
                atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
                continue;
            }
            uint32_t seen = 0;
            int v = 0;
            for (int k = 0; k < s.naggs; ++k) {
                const int kind = s.aggs[k].kind;
                const int slotV = v;
                if (kind != AGG_COUNT) ++v;
                if (!((mask >> k) & 1u)) continue;
                if (s.aggs[k].check_valid) seen |= 1u << k;
                AccUpdate(kind, lAcc + size_t(k) * R + g, kind == AGG_COUNT ? 1ull : a.rec.val[size_t(slotV) * size_t(a.stride) + i]);
            }
            if (seen != 0 && (lSeen[g] & seen) != seen) atomicOr(lSeen + g, seen);
          }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < R; i += kBlock) {
            s.t_key_lo[slot0 + i] = lLo[i];
            if (two) s.t_key_hi[slot0 + i] = lHi[i];
            if (hashed) s.t_rep_row[slot0 + i] = lRep[i];
            s.t_seen[slot0 + i] = lSeen[i];
            for (int k = 0; k < s.naggs; ++k) s.t_acc[size_t(k) * s.acc_stride + slot0 + i] = lAcc[size_t(k) * R + i];
        }
        __syncthreads();
    }
    if (err != 0) atomicOr(a.status, err);
}

// ---- table -> output columns -------------------------------------------------------------------------
// The groups leave the table in slot order: occupied slots are counted per tile of kTileRows slots, the counts are
// scanned, and the emit pass places every group at tile offset + rank inside the tile.  (One shared output counter
// would be hit once per wave -- a million same-address updates for a table of 64 M slots.)  The capacity is a power of
// two of at least kTileRows slots.
__global__ __launch_bounds__(kBlock) void KeyedTileCountKernel(const uint64_t* tLo, uint64_t capacity, uint32_t* tileCounts) {
    __shared__ uint32_t waveCount[kBlock / 64];
    const uint64_t ntiles = capacity / kTileRows;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) c += uint32_t(__popcll(__ballot(tLo[tile * kTileRows + u * kBlock + threadIdx.x] != kEmpty)));
        if ((threadIdx.x & 63) == 0) waveCount[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) tileCounts[tile] = waveCount[0] + waveCount[1] + waveCount[2] + waveCount[3];
        __syncthreads();
    }
}

// One group's row of the output: aggregate results (type conversions, nulls, the order-dependent MIN / MAX rules) and,
// for keys that are packed exactly, the key columns decoded from the table's key words.  accAt(j): accumulator j of the group.
template <class AccAt>
__device__ __forceinline__ void EmitAggregate(const KeyedFinalizeArgs& f, int k, int64_t g, uint32_t seen, AccAt accAt) {
    const int kind = f.aggs[k].kind;
    uint64_t v = accAt(f.agg_src[k]);
    bool valid = true;
    if (f.agg_row_index[k]) {
        valid = v != ~0ull;  // smallest row with a non-null argument, none -> null
        if (valid) v &= kRowIndexMask;  // (the order of the single-byte-key path sits above the row index)
    } else {
        if (kind == AGG_MIN_F || kind == AGG_MAX_F) {
            v = F64FromOrdered(v);
            // the group's first non-null argument was a NaN: the reference's state never leaves it
            if (f.agg_nan_src[k] >= 0) {
                const uint64_t first = accAt(f.agg_nan_src[k]);
                if (first != ~0ull && (first & 1ull)) v = 0x7FF8000000000000ull;
            }
            // a zero result: the first zero that arrived stays, `<` / `>` do not tell the two zeros apart
            if (f.agg_zero_src[k] >= 0 && (v << 1) == 0) {
                const uint64_t first = accAt(f.agg_zero_src[k]);
                if (first != ~0ull) v = (first & 1ull) << 63;
            }
        }
        if (kind != AGG_COUNT && f.aggs[k].check_valid) valid = (seen >> f.agg_src[k]) & 1u;
        if (!valid) v = 0;
        v = PayloadToStored(v, f.agg_out_type[k]);
    }
    if (f.agg_vals[k] != nullptr) {
        StoreTyped(f.agg_vals[k], f.agg_row_index[k] ? 8 : TypeWidth(f.agg_out_type[k]), g, v);
        if (f.agg_valid_bytes[k] != nullptr) f.agg_valid_bytes[k][g] = valid ? 1 : 0;
    }
}

// Key k of a group from the 127 packed bits (lo, hi: Finish() undone), which start at bit `pos`; returns the next key's.
__device__ __forceinline__ int EmitKey(const KeyedFinalizeArgs& f, int k, int pos, int64_t g, uint64_t lo, uint64_t hi) {
    const int type = f.keys[k].type;
    const int nbits = 8 * TypeWidth(type);
    uint64_t bits;
    if (pos < 64) {
        bits = lo >> pos;
        if (pos > 0 && pos + nbits > 64) bits |= hi << (64 - pos);
    } else {
        bits = hi >> (pos - 64);
    }
    bits &= WidthMask(nbits);
    pos += nbits;
    StoreTyped(f.key_vals[k], TypeWidth(type), g, bits);
    if (f.keys[k].nullable) {
        const uint64_t isNull = pos < 64 ? (lo >> pos) & 1ull : (hi >> (pos - 64)) & 1ull;
        pos += 1;
        if (f.key_valid_bytes[k] != nullptr) f.key_valid_bytes[k][g] = isNull ? 0 : 1;
    }
    return pos;
}

template <class AccAt>
__device__ __forceinline__ void EmitGroup(const KeyedFinalizeArgs& f, int64_t g, uint64_t klo, uint64_t khi, uint32_t seen, uint64_t rep,
                                          AccAt accAt) {
    if (f.group_rep_row != nullptr) f.group_rep_row[g] = rep;
    for (int k = 0; k < f.naggs; ++k) EmitAggregate(f, k, g, seen, accAt);
    if (!f.hashed) {
        const uint64_t lo = klo | (khi << 63), hi = khi >> 1;
        int pos = 0;
        for (int k = 0; k < f.nkeys; ++k) pos = EmitKey(f, k, pos, g, lo, hi);
    }
}

__global__ __launch_bounds__(kBlock) void KeyedEmitKernel(const KeyedFinalizeArgs) {
    const KeyedFinalizeArgs& f = KernArgsAs<KeyedFinalizeArgs>();
    __shared__ uint32_t wordCount[kTileRows / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t ntiles = f.capacity / kTileRows;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tileFirst = f.tile_offsets[tile];
        if (f.tile_offsets[tile + 1] == tileFirst) continue;  // (the same for the whole workgroup)
        // word w of the tile = slots 64 w .. 64 w + 63; this wave owns words u * 4 + wave
        uint64_t klos[kRowsPerThread], taken[kRowsPerThread];
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            klos[u] = f.t_key_lo[tile * kTileRows + u * kBlock + threadIdx.x];
            taken[u] = __ballot(klos[u] != kEmpty);
            if (lane == 0) wordCount[u * (kBlock / 64) + wave] = uint32_t(__popcll(taken[u]));
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const uint64_t klo = klos[u];
            const uint64_t i = tile * kTileRows + u * kBlock + threadIdx.x;
            uint32_t before = 0;
            for (int w = 0; w < u * (kBlock / 64) + wave; ++w) before += wordCount[w];
            if (klo == kEmpty) continue;
            const int64_t g = tileFirst + before + __popcll(taken[u] & ((1ull << lane) - 1ull));
            const uint64_t khi = (!f.hashed && f.two_words) ? f.t_key_hi[i] : 0ull;
            const uint64_t rep = f.group_rep_row != nullptr ? f.t_rep_row[i] : 0ull;
            EmitGroup(f, g, klo, khi, f.t_seen[i], rep, [&](int j) { return f.t_acc[size_t(j) * f.capacity + i]; });
        }
        __syncthreads();  // wordCount is rewritten for the next tile
    }
}

// Small tables (a shard-sized batch of a few groups; the tables of the slices of one launch), counted and emitted by one
// launch: workgroup b takes table b of `slice_cap` slots (64 .. kSmallTableSlots) and writes its groups, in slot order, to
// rows b * slice_cap and on of the output columns -- every table has its own rows, so no workgroup waits for another's
// count.  The number of groups of table b goes to host[kSmallEmitCounts + b], the scan's status word to host[0] (`host`:
// pinned memory the host reads after its one wait on the stream).  With `reset` the slots that held a group are set
// back to what KeyedTableInitKernel leaves, and the table can take the next scan as it is.
// A table of few groups is a chain of round trips to memory and little else: every wave reads its eight words of slots at
// once and keeps them, and up to 256 groups are emitted one aggregate per thread -- all accumulators in flight together.
constexpr int kSmallEmitBlock = 1024;
__global__ __launch_bounds__(kSmallEmitBlock) void SmallTablesEmitKernel(const SmallEmitArgs) {
    const SmallEmitArgs& a = KernArgsAs<SmallEmitArgs>();
    const KeyedFinalizeArgs& f = a.f;
    constexpr int kWords = kSmallTableSlots / 64;
    constexpr int kPerWave = kWords / (kSmallEmitBlock / 64);
    static_assert(kWords == 128 && kPerWave * (kSmallEmitBlock / 64) == kWords, "the scan below takes two words a lane");
    __shared__ uint32_t wordCount[kWords];
    __shared__ uint32_t wordStart[kWords];
    __shared__ uint16_t list[kSmallTableSlots];
    __shared__ uint32_t total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cap = a.slice_cap, nwords = cap / 64;
    const uint64_t slot0 = uint64_t(blockIdx.x) * cap;
    uint64_t klo[kPerWave], taken[kPerWave];
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        const uint32_t w = uint32_t(wave * kPerWave + j);
        klo[j] = w < nwords ? f.t_key_lo[slot0 + uint64_t(w) * 64 + lane] : kEmpty;
    }
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        taken[j] = __ballot(klo[j] != kEmpty);
        if (lane == 0) wordCount[wave * kPerWave + j] = uint32_t(__popcll(taken[j]));
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the word counts: two words a lane
        const uint32_t c0 = wordCount[2 * lane], c1 = wordCount[2 * lane + 1];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        wordStart[2 * lane] = incl - c0 - c1;
        wordStart[2 * lane + 1] = incl - c1;
        if (lane == 63) total = incl;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        const uint32_t w = uint32_t(wave * kPerWave + j);
        if (klo[j] != kEmpty) list[wordStart[w] + __popcll(taken[j] & ((1ull << lane) - 1ull))] = uint16_t(w * 64 + lane);
    }
    __syncthreads();
    const uint32_t n = total;
    if (n * kMaxAggs <= 4 * kSmallEmitBlock) {
        for (uint32_t idx = threadIdx.x; idx < n * kMaxAggs; idx += kSmallEmitBlock) {
            const uint32_t r = idx / kMaxAggs;
            const int k = int(idx % kMaxAggs);
            if (k >= f.naggs) continue;
            const uint64_t i = slot0 + list[r];
            EmitAggregate(f, k, int64_t(slot0 + r), f.t_seen[i], [&](int j) { return f.t_acc[size_t(j) * f.capacity + i]; });
        }
        for (uint32_t r = threadIdx.x; r < n; r += kSmallEmitBlock) {
            const uint64_t i = slot0 + list[r];
            if (f.group_rep_row != nullptr) f.group_rep_row[slot0 + r] = f.t_rep_row[i];
            if (!f.hashed) {
                const uint64_t tlo = f.t_key_lo[i], thi = f.two_words ? f.t_key_hi[i] : 0ull;
                const uint64_t lo = tlo | (thi << 63), hi = thi >> 1;
                int pos = 0;
                for (int k = 0; k < f.nkeys; ++k) pos = EmitKey(f, k, pos, int64_t(slot0 + r), lo, hi);
            }
        }
    } else {
        for (uint32_t r = threadIdx.x; r < n; r += kSmallEmitBlock) {
            const uint64_t i = slot0 + list[r];
            const uint64_t khi = (!f.hashed && f.two_words) ? f.t_key_hi[i] : 0ull;
            const uint64_t rep = f.group_rep_row != nullptr ? f.t_rep_row[i] : 0ull;
            EmitGroup(f, int64_t(slot0 + r), f.t_key_lo[i], khi, f.t_seen[i], rep, [&](int j) { return f.t_acc[size_t(j) * f.capacity + i]; });
        }
    }
    if (a.reset) {
        __syncthreads();  // (an aggregate may share its accumulator with another: every read of the table is over)
        for (uint32_t r = threadIdx.x; r < n; r += kSmallEmitBlock) {
            const uint64_t i = slot0 + list[r];
            const_cast<uint64_t*>(f.t_key_lo)[i] = kEmpty;
            if (f.two_words) const_cast<uint64_t*>(f.t_key_hi)[i] = kUnpublished;
            if (f.t_rep_row != nullptr && f.hashed) const_cast<uint64_t*>(f.t_rep_row)[i] = ~0ull;
            const_cast<uint32_t*>(f.t_seen)[i] = 0;
            for (int k = 0; k < a.nacc; ++k) const_cast<uint64_t*>(f.t_acc)[size_t(k) * f.capacity + i] = a.ident[k];
        }
    }
    if (threadIdx.x == 0) {
        a.host[kSmallEmitCounts + blockIdx.x] = n;
        if (blockIdx.x == 0) a.host[0] = *a.status;
    }
}

// Position of a key inside the LDS table of its region.  The region itself is chosen by other bits of the same 32-bit
// fold (DigitOf), so this takes them through another multiplier.
__device__ __forceinline__ uint32_t RegionLdsStart(uint32_t fold, int shift) { return (fold * 0xC2B2AE35u) >> shift; }

// One region of the grouping finished by one workgroup, without a table in HBM: the region's groups are built in an LDS
// table -- first what the scan's workgroups left in their region of the scan's small level-2 table (their level-1 tables,
// merged there as they ended), then the region's records -- and leave straight into the output columns, at positions
// the workgroup claims from one counter (the order of the groups in the output is as unspecified as the reference's
// hash-table order).  A region whose groups outnumber the LDS slots raises SCAN_TABLE_FULL (the host repeats the step with
// more regions); when the output columns are too short nothing is written, the counter still ends as the number of
// groups, and the host runs this pass again with longer ones.
__global__ __launch_bounds__(kBlock) void RegionEmitKernel(const RegionEmitArgs) {
    const RegionEmitArgs& a = KernArgsAs<RegionEmitArgs>();
    const KeyedSink& s = a.s;
    const KeyedFinalizeArgs& f = a.f;
    extern __shared__ uint64_t lRegion[];
    const int R = a.region_slots, Rs = a.small_slots;
    const int ldsShift = __builtin_clz(uint32_t(R)) + 1;  // 32 - log2(R)
    const bool two = s.two_words != 0, hashed = s.hashed != 0;
    uint64_t* lLo = lRegion;
    uint64_t* lHi = lLo + R;
    uint64_t* lRep = lHi + (two ? R : 0);
    uint64_t* lAcc = lRep + (hashed ? R : 0);
    uint32_t* lSeen = reinterpret_cast<uint32_t*>(lAcc + size_t(s.naggs) * R);
    __shared__ uint32_t lWordStart[64 + 1];  // groups in the 64-slot words before each word of the table; [nwords]: all
    __shared__ uint32_t lStop;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwords = R / 64;
    uint32_t err = 0;
    for (int region = blockIdx.x; region < a.nregions; region += gridDim.x) {
        const uint32_t first = a.region_start[region], last = a.region_start[region + 1];
        const uint64_t small0 = uint64_t(region) * uint64_t(Rs);
        bool any = first != last;
        for (int i = threadIdx.x; i < Rs && !any; i += kBlock) any = s.t_key_lo[small0 + i] != kEmpty;
        // (also keeps the previous region's emit pass and this region's set-up apart)
        if (!__syncthreads_or(any ? 1 : 0)) continue;
        // a region that ran full somewhere: the host repeats the step with more regions, the rest is wasted work
        if (threadIdx.x == 0) lStop = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SCAN_TABLE_FULL;
        __syncthreads();
        if (lStop != 0) break;  // (one value for the whole workgroup: the next store to it is barriers away)
        for (int i = threadIdx.x; i < R; i += kBlock) {
            lLo[i] = kEmpty;
            if (two) lHi[i] = kUnpublished;
            if (hashed) lRep[i] = ~0ull;
            lSeen[i] = 0;
            for (int k = 0; k < s.naggs; ++k) lAcc[size_t(k) * R + i] = TableIdentity(s.aggs[k].kind);
        }
        __syncthreads();
        // what the scan's workgroups merged into this region of their table
        for (int i = threadIdx.x; i < Rs; i += kBlock) {
            const uint64_t lo = s.t_key_lo[small0 + i];
            if (lo == kEmpty || err != 0) continue;
            const uint64_t hi = two ? s.t_key_hi[small0 + i] : 0ull;
            const uint32_t start = RegionLdsStart(Fold32(lo, hi), ldsShift);
            int64_t g;
            if (hashed && (hi >> 63) != 0) {
                const uint64_t row = s.t_rep_row[small0 + i];
                g = TableUpsertVerified(lLo, lHi, lRep, uint64_t(R - 1), uint64_t(R - 1), start, lo, hi, row, uint32_t(R),
                                        [&](uint64_t rep) { return SameKeyRows<4>(a.cols, s, row, rep); });
            } else {
                g = TableUpsert(lLo, lHi, uint64_t(R - 1), uint64_t(R - 1), start, lo, hi, two, uint32_t(R));
                if (hashed && g >= 0) atomicMin(reinterpret_cast<unsigned long long*>(lRep + g), static_cast<unsigned long long>(s.t_rep_row[small0 + i]));
            }
            if (g < 0) {
                err |= SCAN_TABLE_FULL;
                continue;
            }
            for (int k = 0; k < s.naggs; ++k) {
                const int kind = s.aggs[k].kind;
                const uint64_t v = s.t_acc[size_t(k) * s.acc_stride + small0 + i];
                if (v != TableIdentity(kind)) AccUpdate(kind, lAcc + size_t(k) * R + g, v);
            }
            const uint32_t seen = s.t_seen[small0 + i];
            if (seen != 0) atomicOr(lSeen + g, seen);
        }
        // the region's records
        for (uint32_t i0 = first + threadIdx.x; i0 < last; i0 += 4 * kBlock) {
            uint64_t los[4], his[4];
            uint32_t masks[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i = i0 + j * kBlock < last ? i0 + j * kBlock : i0;
                los[j] = a.rec.lo[i];
                his[j] = 0;
                masks[j] = ~0u;
            }
            if (two && !s.rec_raw64) {
#pragma unroll
                for (int j = 0; j < 4; ++j) his[j] = a.rec.hi[i0 + j * kBlock < last ? i0 + j * kBlock : i0];
            }
            if (s.rec_raw64) {
#pragma unroll
                for (int j = 0; j < 4; ++j) RawKeyWords(los[j], los[j], his[j]);
            }
            if (a.rec.mask != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) masks[j] = a.rec.mask[i0 + j * kBlock < last ? i0 + j * kBlock : i0];
            }
            uint32_t rows[4] = {0, 0, 0, 0};  // (with the other words of the records: not a load and a wait per record)
            if (hashed) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rows[j] = a.rec.row[i0 + j * kBlock < last ? i0 + j * kBlock : i0];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t i = i0 + j * kBlock;
                if (i >= last || err != 0) continue;
                const uint64_t lo = los[j], hi = his[j];
                const uint32_t mask = masks[j];
                const uint32_t start = RegionLdsStart(Fold32(lo, hi), ldsShift);
                int64_t g;
                if (hashed && (hi >> 63) != 0) {
                    const uint64_t row = rows[j];
                    g = TableUpsertVerified(lLo, lHi, lRep, uint64_t(R - 1), uint64_t(R - 1), start, lo, hi, row, uint32_t(R),
                                            [&](uint64_t rep) { return SameKeyRows<4>(a.cols, s, row, rep); });
                } else {
                    g = TableUpsert(lLo, lHi, uint64_t(R - 1), uint64_t(R - 1), start, lo, hi, two, uint32_t(R));
                    if (hashed && g >= 0) atomicMin(reinterpret_cast<unsigned long long*>(lRep + g), static_cast<unsigned long long>(rows[j]));
                }
                if (g < 0) {
                    err |= SCAN_TABLE_FULL;
                    continue;
                }
                uint32_t seen = 0;
                int v = 0;
                for (int k = 0; k < s.naggs; ++k) {
                    const int kind = s.aggs[k].kind;
                    const int slotV = v;
                    if (kind != AGG_COUNT) ++v;
                    if (!((mask >> k) & 1u)) continue;
                    if (s.aggs[k].check_valid) seen |= 1u << k;
                    AccUpdate(kind, lAcc + size_t(k) * R + g, kind == AGG_COUNT ? 1ull : a.rec.val[size_t(slotV) * size_t(a.stride) + i]);
                }
                if (seen != 0 && (lSeen[g] & seen) != seen) atomicOr(lSeen + g, seen);
            }
        }
        if (__syncthreads_or(err != 0 ? 1 : 0)) {
            if (threadIdx.x == 0) atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
            break;
        }
        // The groups leave in slot order.  Word w of the table = slots 64 w .. 64 w + 63, counted by the wave that will emit
        // them (thread t takes the slots t, t + 256, ..: word 4 j + wave in its trip j).
        for (int j = 0; j * kBlock < R; ++j) {
            const int i = j * kBlock + threadIdx.x;
            const uint64_t taken = __ballot(i < R && lLo[i] != kEmpty);
            if (lane == 0 && i < R) lWordStart[(i >> 6) + 1] = uint32_t(__popcll(taken));
        }
        __syncthreads();
        if (wave == 0) {  // (64 words at most: one lane each)
            uint32_t c = lane < nwords ? lWordStart[lane + 1] : 0u;
            uint32_t inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (lane < nwords) lWordStart[lane + 1] = inc;
            if (lane == 0) lWordStart[0] = 0;
        }
        __syncthreads();
        const uint32_t total = lWordStart[nwords];
        if (threadIdx.x == 0) a.region_groups[region] = total;
        const uint64_t row0 = uint64_t(region) * uint64_t(R);
        for (int j = 0; j * kBlock < R; ++j) {
            const int i = j * kBlock + threadIdx.x;
            const uint64_t klo = i < R ? lLo[i] : kEmpty;
            const uint64_t taken = __ballot(klo != kEmpty);
            if (klo == kEmpty) continue;
            const uint64_t at = row0 + lWordStart[i >> 6] + __popcll(taken & ((1ull << lane) - 1ull));
            a.tmp_lo[at] = klo;
            if (two) a.tmp_hi[at] = lHi[i];
            if (hashed) a.tmp_rep[at] = lRep[i];
            a.tmp_seen[at] = lSeen[i];
            for (int k = 0; k < s.naggs; ++k) a.tmp_acc[size_t(k) * a.tmp_stride + at] = lAcc[size_t(k) * R + i];
        }
    }
}

// The same pass for keys that are packed exactly (nothing to verify in memory), built around what it is bound by: the
// instructions of the probes.  A lane takes kEmitC records at a time, hashes them in straight-line code, and looks at the
// first slot of each with plain reads, all of them in flight together: at the load the host sizes these tables for, most
// records find their key there.  The rest -- keys not yet in the table, collisions -- walk on in a loop that takes one
// step for every record still looking, claims free slots by compare-and-swap, and skips the records that are done.
// Double hashing: inside LDS a probe sequence gains nothing from being contiguous, and strides that differ from key to
// key keep the sequences short.  First slot and stride come from the bits of the key's table position above those that
// chose the region.  The accumulators are then updated from registers, the next records are on their way from HBM
// meanwhile, and nothing in the loop over the records waits for another wave.  Keys of exactly 64 bits
// (KeyedSink.rec_raw64) live in the table as the one word their records carry; the one value that equals the free
// marker has a slot of its own behind the table.
constexpr int kEmitC = 4;                  // records per lane and trip
constexpr int kEmitWave = kEmitC * 64;
// kRaw: 64-bit keys carried as one word; kTwo: the table keeps a hi word; kCountOnly: one aggregate, a count of rows (no
// operands, no masks, no seen flags).  (What the launch knows is compiled in: the pass is bound by its instructions.)
// kThreads: threads of a workgroup -- a region is a chain of LDS round trips per record, and what hides them is waves.
template <bool kRaw, bool kTwo, bool kCountOnly, int kThreads>
__global__ __launch_bounds__(kThreads) void RegionEmitExactKernel(const RegionEmitArgs) {
    const RegionEmitArgs& a = KernArgsAs<RegionEmitArgs>();
    const KeyedSink& s = a.s;
    extern __shared__ uint64_t lRegion[];
    const int R = a.region_slots, Rs = a.small_slots;
    const uint32_t slotMask = uint32_t(R - 1);
    const int posShift = 64 - __builtin_clzll(s.table_mask);  // bits of a table position that choose slot and region in the scan's table
    constexpr bool raw = kRaw, two = kTwo;
    const int naggs = kCountOnly ? 1 : s.naggs;
    const int R1 = R + 1;  // (slot R: the raw key ~0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool anyCheck = false;
    if (!kCountOnly) {
        for (int k = 0; k < naggs; ++k) anyCheck = anyCheck || s.aggs[k].check_valid != 0;
    }
    uint64_t* lLo = lRegion;
    uint64_t* lHi = lLo + R1;
    uint64_t* lAcc = lHi + (two ? R : 0);
    uint32_t* lSeen = reinterpret_cast<uint32_t*>(lAcc + size_t(naggs) * R1);
    uint16_t* lList = reinterpret_cast<uint16_t*>(lSeen + (anyCheck ? R1 : 0));  // the slots in use, in slot order
    __shared__ uint32_t lWordStart[64 + 1];  // groups in the 64-slot words before each word of the table; [nwords]: all
    const int nwords = R / 64;
    const bool needMask = !kCountOnly && a.rec.mask != nullptr;
    uint32_t err = 0;
#ifdef SSA_REGION_TIMING
    uint64_t tphase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = __builtin_readcyclecounter();
#define SSA_RT(i) do { const uint64_t now_ = __builtin_readcyclecounter(); tphase[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define SSA_RT(i) do {} while (0)
#endif

    // first slot and stride of a key's walk
    auto startOf = [&](uint64_t lo, uint64_t hi, uint32_t& slot, uint32_t& step) {
        const uint64_t pos = TablePos(lo, hi) >> posShift;
        slot = uint32_t(pos) & slotMask;
        step = (uint32_t(pos >> 10) | 1u) & slotMask;
    };
    // one step of a walk: the key's slot, or -1 (go on: `slot` has moved unless the slot's owner is between its two stores)
    auto visit = [&](uint64_t lo, uint64_t hi, uint32_t& slot, uint32_t step) -> int {
        uint64_t seen = __hip_atomic_load(lLo + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (seen == kEmpty) {
            seen = atomicCAS(reinterpret_cast<unsigned long long*>(lLo + slot), static_cast<unsigned long long>(kEmpty), static_cast<unsigned long long>(lo));
            if (seen == kEmpty) {
                if (two) __hip_atomic_store(lHi + slot, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return int(slot);
            }
        }
        if (seen == lo) {
            if (!two) return int(slot);
            const uint64_t ch = __hip_atomic_load(lHi + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ch == hi) return int(slot);
            if (ch == kUnpublished) return -1;  // the same slot again
        }
        slot = (slot + step) & slotMask;
        return -1;
    };
    // the records [i0, i0 + kEmitWave) of this wave, as far as they exist, from HBM into registers
    auto loadRecords = [&](uint32_t i0, uint32_t last, uint64_t (&lo)[kEmitC], uint64_t (&hi)[kEmitC]) {
#pragma unroll
        for (int c = 0; c < kEmitC; ++c) {
            const uint32_t i = i0 + c * 64 + lane;
            lo[c] = i < last ? a.rec.lo[i] : 0ull;
            hi[c] = 0;
        }
        if (two) {
#pragma unroll
            for (int c = 0; c < kEmitC; ++c) {
                const uint32_t i = i0 + c * 64 + lane;
                hi[c] = i < last ? a.rec.hi[i] : 0ull;
            }
        }
    };

    for (int region = blockIdx.x; region < a.nregions; region += gridDim.x) {
        const uint32_t first = a.region_start[region], last = a.region_start[region + 1];
        const uint64_t small0 = uint64_t(region) * uint64_t(Rs);
        // on their way while the table is set up: the first slots of the scan's table, the wave's first records
        const uint64_t seed0 = int(threadIdx.x) < Rs ? s.t_key_lo[small0 + threadIdx.x] : kEmpty;
        uint64_t nlo[kEmitC], nhi[kEmitC];
        loadRecords(first + wave * kEmitWave, last, nlo, nhi);
        SSA_RT(0);
        // (also keeps the previous region's last pass over the table and this region's set-up apart)
        if (!__syncthreads_or((first != last || seed0 != kEmpty || Rs > kThreads) ? 1 : 0)) continue;
        SSA_RT(1);
        for (int i = threadIdx.x; i < R1; i += kThreads) {
            lLo[i] = kEmpty;
            if (two && i < R) lHi[i] = kUnpublished;
            if (anyCheck) lSeen[i] = 0;
            if (kCountOnly) lAcc[i] = 0;
            else for (int k = 0; k < naggs; ++k) lAcc[size_t(k) * R1 + i] = TableIdentity(s.aggs[k].kind);
        }
        __syncthreads();
        SSA_RT(2);
        // what the scan's workgroups merged into this region of their table
        for (int i = threadIdx.x; i < Rs && !(s.pad2_ & 4); i += kThreads) {
            const uint64_t slo = i == int(threadIdx.x) ? seed0 : s.t_key_lo[small0 + i];
            if (slo == kEmpty || err != 0) continue;
            const uint64_t shi = s.two_words ? s.t_key_hi[small0 + i] : 0ull;
            const uint64_t lo = raw ? (slo | (shi << 63)) : slo, hi = raw ? 0ull : shi;
            int g = -1;
            if (raw && lo == kEmpty) {
                g = R;
                lLo[R] = 0;
            } else {
                uint32_t slot, step;
                startOf(slo, shi, slot, step);
                for (uint32_t tries = 0; g < 0 && tries < 3u * uint32_t(R) + 64u; ++tries) g = visit(lo, hi, slot, step);
            }
            if (g < 0) {
                err |= SCAN_TABLE_FULL;
                continue;
            }
            for (int k = 0; k < naggs; ++k) {
                const int kind = s.aggs[k].kind;
                const uint64_t v = s.t_acc[size_t(k) * s.acc_stride + small0 + i];
                if (v != TableIdentity(kind)) AccUpdate(kind, lAcc + size_t(k) * R1 + g, v);
            }
            if (anyCheck) {
                const uint32_t seen = s.t_seen[small0 + i];
                if (seen != 0) atomicOr(lSeen + g, seen);
            }
        }
        SSA_RT(3);
        // the region's records, kEmitWave per wave at a time
        for (uint32_t w0 = first + wave * kEmitWave; w0 < last && !(s.pad2_ & 1); w0 += (kThreads / 64) * kEmitWave) {
            uint64_t lo[kEmitC], hi[kEmitC];
            uint32_t slot[kEmitC], step[kEmitC];
            uint32_t live = 0, pending = 0;
#pragma unroll
            for (int c = 0; c < kEmitC; ++c) {
                lo[c] = nlo[c];
                hi[c] = nhi[c];
            }
            if (w0 + (kThreads / 64) * kEmitWave < last) loadRecords(w0 + (kThreads / 64) * kEmitWave, last, nlo, nhi);
            uint64_t seen[kEmitC];
#pragma unroll
            for (int c = 0; c < kEmitC; ++c) {
                // (table positions are those of the key words the scan's tables hold: TablePosRaw says what they are for a raw word)
                const uint64_t pos = (raw ? TablePosRaw(lo[c]) : TablePos(lo[c], hi[c])) >> posShift;
                slot[c] = uint32_t(pos) & slotMask;
                step[c] = (uint32_t(pos >> 10) | 1u) & slotMask;
                seen[c] = lLo[slot[c]];
            }
#pragma unroll
            for (int c = 0; c < kEmitC; ++c) {
                if (w0 + c * 64 + lane >= last) continue;
                live |= 1u << c;
                if (raw && lo[c] == kEmpty) {
                    slot[c] = uint32_t(R);
                    lLo[R] = 0;
                } else if (two || seen[c] != lo[c]) {
                    pending |= 1u << c;
                }
            }
            if (err != 0) live = pending = 0;
            // a first slot that read as free is claimed at once (one-word keys: nothing else to store); what is then
            // still pending are collisions and lost races
            if (!two) {
#pragma unroll
                for (int c = 0; c < kEmitC; ++c) {
                    if (((pending >> c) & 1u) && seen[c] == kEmpty) {
                        const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(lLo + slot[c]), static_cast<unsigned long long>(kEmpty),
                                                        static_cast<unsigned long long>(lo[c]));
                        seen[c] = prev == kEmpty ? lo[c] : prev;
                    }
                }
#pragma unroll
                for (int c = 0; c < kEmitC; ++c) {
                    if (!((pending >> c) & 1u)) continue;
                    if (seen[c] == lo[c]) pending &= ~(1u << c); else slot[c] = (slot[c] + step[c]) & slotMask;
                }
            }
            uint32_t tries = 0;
            while (__any(pending != 0)) {
#pragma unroll
                for (int c = 0; c < kEmitC; ++c) {
                    if (!((pending >> c) & 1u)) continue;
                    if (visit(lo[c], hi[c], slot[c], step[c]) >= 0) pending &= ~(1u << c);
                }
                if (++tries > 3u * uint32_t(R) + 64u) {  // (an odd stride visits every slot: the table is full)
                    if (pending != 0) err |= SCAN_TABLE_FULL;
                    live = pending = 0;
                }
            }
            if (kCountOnly) {
                // (the records of a key of many rows fill whole waves: 64 additions to one address would go one after the
                // other, so the lanes that share the first lane's slot add their number once)
#pragma unroll
                for (int c = 0; c < kEmitC; ++c) {
                    const bool on = ((live >> c) & 1u) != 0;
                    const uint64_t lanesOn = __ballot(on);
                    if (lanesOn == 0) continue;
                    const int leader = __ffsll(static_cast<unsigned long long>(lanesOn)) - 1;
                    const uint32_t leaderSlot = uint32_t(__builtin_amdgcn_readlane(int(slot[c]), leader));
                    const bool shares = on && slot[c] == leaderSlot;
                    const uint64_t sharing = __ballot(shares);
                    const uint64_t add = shares ? (lane == leader ? uint64_t(__popcll(sharing)) : 0ull) : 1ull;
                    if (on && add != 0) atomicAdd(reinterpret_cast<unsigned long long*>(lAcc + slot[c]), static_cast<unsigned long long>(add));
                }
                continue;
            }
            // the accumulators, one aggregate after the other (what describes an aggregate is the same for every record)
            uint32_t masks[kEmitC];
#pragma unroll
            for (int c = 0; c < kEmitC; ++c) masks[c] = ((live >> c) & 1u) ? ~0u : 0u;
            if (needMask) {
#pragma unroll
                for (int c = 0; c < kEmitC; ++c) {
                    if ((live >> c) & 1u) masks[c] = a.rec.mask[w0 + uint32_t(c * 64 + lane)];
                }
            }
            int valSlot = 0;
            for (int k = 0; k < naggs; ++k) {
                const int kind = s.aggs[k].kind;
                uint64_t* acc = lAcc + size_t(k) * R1;
                if (kind == AGG_COUNT) {
#pragma unroll
                    for (int c = 0; c < kEmitC; ++c) {
                        if ((masks[c] >> k) & 1u) atomicAdd(reinterpret_cast<unsigned long long*>(acc + slot[c]), 1ull);
                    }
                } else {
                    const uint64_t* vals = a.rec.val + size_t(valSlot) * size_t(a.stride);
                    ++valSlot;
                    uint64_t v[kEmitC];
#pragma unroll
                    for (int c = 0; c < kEmitC; ++c) v[c] = ((masks[c] >> k) & 1u) ? vals[w0 + uint32_t(c * 64 + lane)] : 0ull;
#pragma unroll
                    for (int c = 0; c < kEmitC; ++c) {
                        if ((masks[c] >> k) & 1u) AccUpdate(kind, acc + slot[c], v[c]);
                    }
                }
                if (s.aggs[k].check_valid) {
#pragma unroll
                    for (int c = 0; c < kEmitC; ++c) {
                        if (((masks[c] >> k) & 1u) && !((lSeen[slot[c]] >> k) & 1u)) atomicOr(lSeen + slot[c], 1u << k);
                    }
                }
            }
        }
        SSA_RT(4);
        if (__syncthreads_or(err != 0 ? 1 : 0)) {
            if (threadIdx.x == 0) atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
            break;
        }
        SSA_RT(5);
        if (s.pad2_ & 2) continue;
        // The groups leave in slot order.  Word w of the table = slots 64 w .. 64 w + 63.
        for (int j = 0; j * kThreads < R; ++j) {
            const int i = j * kThreads + threadIdx.x;
            const uint64_t taken = __ballot(i < R && lLo[i] != kEmpty);
            if (lane == 0 && i < R) lWordStart[(i >> 6) + 1] = uint32_t(__popcll(taken));
        }
        __syncthreads();
        if (wave == 0) {  // (64 words at most: one lane each)
            uint32_t c = lane < nwords ? lWordStart[lane + 1] : 0u;
            uint32_t inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            if (lane < nwords) lWordStart[lane + 1] = inc;
            if (lane == 0) lWordStart[0] = 0;
        }
        __syncthreads();
        SSA_RT(6);
        const uint32_t inTable = lWordStart[nwords];
        const bool extra = lLo[R] != kEmpty;
        const uint32_t ngroups = inTable + (extra ? 1u : 0u);
        if (ngroups > uint32_t(R)) {  // (a full table and the key behind it: one row more than the region has in the temporary table)
            if (threadIdx.x == 0) atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
            break;
        }
        if (threadIdx.x == 0) a.region_groups[region] = ngroups;
        // the slots in use, listed, then array after array over the list: every lane has a group, and neighbouring lanes
        // write neighbouring rows
        for (int j = 0; j * kThreads < R; ++j) {
            const int i = j * kThreads + threadIdx.x;
            const bool there = i < R && lLo[i] != kEmpty;
            const uint64_t taken = __ballot(there);
            if (there) lList[lWordStart[i >> 6] + __popcll(taken & ((1ull << lane) - 1ull))] = uint16_t(i);
        }
        if (threadIdx.x == 0 && extra) lList[inTable] = uint16_t(R);
        __syncthreads();
        const uint64_t row0 = uint64_t(region) * uint64_t(R);
        for (uint32_t idx = threadIdx.x; idx < ngroups; idx += kThreads) {
            const int i = lList[idx];
            // (a raw 64-bit key leaves as the one word it is; seen flags only where an aggregate looks at them)
            const uint64_t elo = lLo[i < R ? i : 0];
            a.tmp_lo[row0 + idx] = raw && i == R ? kEmpty : elo;
            if (kTwo) a.tmp_hi[row0 + idx] = lHi[i < R ? i : 0];
            if (anyCheck) a.tmp_seen[row0 + idx] = lSeen[i];
        }
        for (int k = 0; k < naggs; ++k) {
            for (uint32_t idx = threadIdx.x; idx < ngroups; idx += kThreads) a.tmp_acc[size_t(k) * a.tmp_stride + row0 + idx] = lAcc[size_t(k) * R1 + lList[idx]];
        }
        SSA_RT(7);
    }
#ifdef SSA_REGION_TIMING
    if (threadIdx.x == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.status) + 16;
        for (int i = 0; i < 8; ++i) atomicAdd(dbg + i, static_cast<unsigned long long>(tphase[i]));
    }
#endif
#undef SSA_RT
}

// The groups of all regions, which the region pass left as rows [region * R, region * R + groups of the region) of the
// temporary table, to their places in the output columns: region after region, so the order of the result is the same
// from run to run.  offsets: the exclusive scan of the regions' group counts.  Rows beyond the output's capacity are not
// written (the host looks at the total and runs this kernel again with longer columns).
__global__ __launch_bounds__(kBlock) void RegionCompactEmitKernel(const RegionEmitArgs) {
    const RegionEmitArgs& a = KernArgsAs<RegionEmitArgs>();
    const KeyedFinalizeArgs& f = a.f;
    const int R = a.region_slots;
    // what the pass over exact keys leaves out: the second word of a raw 64-bit key (tmp_lo holds the key as one word), the
    // seen flags when no aggregate looks at them
    const bool raw = !f.hashed && a.s.rec_raw64 != 0;
    bool seenKept = f.hashed != 0;
    for (int k = 0; k < a.s.naggs; ++k) seenKept = seenKept || a.s.aggs[k].check_valid != 0;
    for (int region = blockIdx.x; region < a.nregions; region += gridDim.x) {
        const uint32_t n = a.region_groups[region];
        const int64_t base = a.region_offsets[region];
        if (uint64_t(base) + n > f.capacity) continue;
        const uint64_t row0 = uint64_t(region) * uint64_t(R);
        for (uint32_t idx = threadIdx.x; idx < n; idx += kBlock) {
            const uint64_t at = row0 + idx;
            uint64_t klo = a.tmp_lo[at];
            uint64_t khi = (!f.hashed && f.two_words && !raw) ? a.tmp_hi[at] : 0ull;
            if (raw) RawKeyWords(klo, klo, khi);
            const uint64_t rep = f.group_rep_row != nullptr ? a.tmp_rep[at] : 0ull;
            EmitGroup(f, base + idx, klo, khi, seenKept ? a.tmp_seen[at] : 0u, rep, [&](int j) { return a.tmp_acc[size_t(j) * a.tmp_stride + at]; });
        }
    }
}

// ---- gathers (representative rows -> key columns, row indices -> some() values) --------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void GatherFixedKernel(const T* src, const uint8_t* srcValid, int64_t srcValidOff,
                                                          const uint64_t* rows, int64_t n, T* dst, uint8_t* dstValidBytes) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        const uint64_t r = rows[i];
        if (r == ~0ull) {
            dst[i] = T{};
            if (dstValidBytes != nullptr) dstValidBytes[i] = 0;
        } else {
            dst[i] = src[r];
            if (dstValidBytes != nullptr) dstValidBytes[i] = srcValid ? uint8_t(LoadBit(srcValid, int64_t(r) + srcValidOff)) : 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void GatherValidityKernel(const uint8_t* srcValid, int64_t srcValidOff, const uint64_t* rows,
                                                             int64_t n, uint8_t* dstValidBytes) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        const uint64_t r = rows[i];
        dstValidBytes[i] = r == ~0ull ? 0 : (srcValid ? uint8_t(LoadBit(srcValid, int64_t(r) + srcValidOff)) : 1);
    }
}

__global__ __launch_bounds__(kBlock) void GatherStringLensKernel(const int32_t* offsets, const uint64_t* rows, int64_t n,
                                                               int32_t* outLens) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        const uint64_t r = rows[i];
        outLens[i] = r == ~0ull ? 0 : offsets[r + 1] - offsets[r];
    }
}

// Eight lanes share a string and move it in 8-byte words (a thread per string, byte by byte, took 1.65 ms for the 5.3 M
// phrases of ClickBench's GROUP BY SearchPhrase).
__global__ __launch_bounds__(kBlock) void GatherStringBytesKernel(const int32_t* offsets, const uint8_t* chars,
                                                                const uint64_t* rows, int64_t n, const int32_t* newOffsets,
                                                                uint8_t* outChars) {
    const int part = threadIdx.x & 7;
    for (int64_t i = (int64_t(blockIdx.x) * kBlock + threadIdx.x) >> 3; i < n; i += (int64_t(gridDim.x) * kBlock) >> 3) {
        const uint64_t r = rows[i];
        if (r == ~0ull) continue;
        const int32_t b = offsets[r], len = offsets[r + 1] - b, o = newOffsets[i];
        for (int32_t j = part * 8; j < len; j += 64) {
            if (j + 8 <= len) {
                uint64_t w;
                __builtin_memcpy(&w, chars + b + j, 8);
                __builtin_memcpy(outChars + o + j, &w, 8);
            } else {
                for (int32_t t = j; t < len; ++t) outChars[o + t] = chars[b + t];
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void CombineDecimalKernel(const uint64_t* lo32, const uint64_t* mid32, const uint64_t* hi,
                                                             int64_t n, uint64_t* out) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        const unsigned __int128 total = (unsigned __int128)lo32[i] + ((unsigned __int128)mid32[i] << 32) + ((unsigned __int128)hi[i] << 64);
        out[2 * i] = uint64_t(total);
        out[2 * i + 1] = uint64_t(total >> 64);
    }
}

__global__ __launch_bounds__(kBlock) void IotaI32Kernel(int32_t* out, int64_t n, int32_t step) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) out[i] = int32_t(i) * step;
}

// out[i] = *addrs[i]: a few words from many places (the first and last offsets of the string columns of many slices)
__global__ __launch_bounds__(64) void ProbeI32Kernel(const ProbeWords p) {
    if (int(threadIdx.x) < p.n) p.out[threadIdx.x] = *p.addr[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void GatherI32AtKernel(const int32_t* const* addrs, int64_t n, int32_t* out) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) out[i] = *addrs[i];
}

struct G128 {
    uint64_t lo, hi;
};

int GridFor(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return int(g);
}

}  // namespace

uint64_t KeyedTableIdentity(int aggKind) { return TableIdentity(aggKind); }

hipError_t LaunchScanKeyed(const ScanArgs& a, const KeyedSink& s, LaunchCfg cfg) {
    KArgs<KeyedSink> k{a, s};
    if (s.hashed) {
        hipLaunchKernelGGL((ScanKeyedKernel<kRowsPerThread, true>), dim3(cfg.grid), dim3(kBlock), cfg.lds_bytes, cfg.stream, k);
    } else {
        hipLaunchKernelGGL((ScanKeyedKernel<kRowsPerThread, false>), dim3(cfg.grid), dim3(kBlock), cfg.lds_bytes, cfg.stream, k);
    }
    return hipGetLastError();
}

int KeyedKernelBlocksPerCU(bool hashed, uint32_t ldsBytes) {
    int n = 0;
    const hipError_t e = hashed ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ScanKeyedKernel<kRowsPerThread, true>, kBlock, ldsBytes)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ScanKeyedKernel<kRowsPerThread, false>, kBlock, ldsBytes);
    return (e == hipSuccess && n > 0) ? n : 1;
}

hipError_t ConfigureKeyedKernels(int maxLdsBytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ScanKeyedKernel<kRowsPerThread, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ScanKeyedKernel<kRowsPerThread, false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    return e;
}

hipError_t LaunchKeyedTableInit(const KeyedSink& sink, bool twoWords, bool hashed, hipStream_t st) {
    KeyedTableInitArgs a{};
    a.capacity = sink.table_mask + 1;
    a.t_key_lo = sink.t_key_lo;
    a.t_key_hi = twoWords ? sink.t_key_hi : nullptr;
    a.t_acc = sink.t_acc;
    a.t_seen = sink.t_seen;
    a.t_rep_row = hashed ? sink.t_rep_row : nullptr;
    a.naggs = sink.naggs;
    for (int k = 0; k < sink.naggs; ++k) a.ident[k] = TableIdentity(sink.aggs[k].kind);
    hipLaunchKernelGGL(KeyedTableInitKernel, dim3(GridFor(int64_t(a.capacity))), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t LaunchRadixCount(const RadixArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(RadixCountKernel, dim3(uint32_t(a.nseg * a.chunks)), dim3(kCountThreads), 0, st, a);
    return hipGetLastError();
}

hipError_t LaunchRadixOffsets(const RadixArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(RadixColumnScanKernel, dim3(uint32_t(a.nseg * (a.ndigits / 64))), dim3(64 * kOffsetWaves), 0, st, a);
    hipLaunchKernelGGL(RadixStartScanKernel, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t LaunchRadixScatter(const RadixArgs& a, hipStream_t st) {
    // (tiles of 32 records a thread -- runs of 256 bytes a digit -- measured slower, 0.93 ms against 0.68 ms for a hundred
    // million 64-bit keys: the kernel is bound by its waves in flight, not by the length of the runs it writes)
    if (a.raw64) {
        hipLaunchKernelGGL((RadixScatterKernel<true, kRadixPerThread>), dim3(uint32_t(a.nseg * a.chunks)), dim3(kBlock), 0, st, a);
    } else {
        hipLaunchKernelGGL((RadixScatterKernel<false, kRadixPerThread>), dim3(uint32_t(a.nseg * a.chunks)), dim3(kBlock), 0, st, a);
    }
    return hipGetLastError();
}

size_t RegionApplyLdsBytes(const KeyedSink& s, int regionSlots) {
    return size_t(regionSlots) * (8 + (s.two_words ? 8 : 0) + (s.hashed ? 8 : 0) + 8 * size_t(s.naggs) + 4);
}

hipError_t LaunchRegionApply(const RegionApplyArgs& a, hipStream_t st) {
    const int grid = a.nregions < kMaxGrid * 16 ? a.nregions : kMaxGrid * 16;
    hipLaunchKernelGGL(RegionApplyKernel, dim3(uint32_t(grid)), dim3(kBlock), RegionApplyLdsBytes(a.s, a.region_slots), st, a);
    return hipGetLastError();
}

size_t RegionEmitLdsBytes(const KeyedSink& s, int regionSlots) {
    if (s.hashed) return RegionApplyLdsBytes(s, regionSlots);
    const bool two = s.two_words != 0 && s.rec_raw64 == 0;
    bool anyCheck = false;
    for (int k = 0; k < s.naggs; ++k) anyCheck = anyCheck || s.aggs[k].check_valid != 0;
    const size_t r1 = size_t(regionSlots) + 1;
    return r1 * (8 + 8 * size_t(s.naggs)) + (two ? size_t(regionSlots) * 8 : 0) + (anyCheck ? r1 * 4 : 0) + r1 * 2 + 16;
}

namespace {
bool CountOnlySink(const KeyedSink& s) { return s.naggs == 1 && s.aggs[0].kind == AGG_COUNT && s.aggs[0].check_valid == 0; }

using RegionKernel = void (*)(const RegionEmitArgs);
template <int kThreads>
RegionKernel ExactRegionKernelOf(bool raw, bool two, bool countOnly) {
    if (raw) return countOnly ? RegionEmitExactKernel<true, false, true, kThreads> : RegionEmitExactKernel<true, false, false, kThreads>;
    if (two) return countOnly ? RegionEmitExactKernel<false, true, true, kThreads> : RegionEmitExactKernel<false, true, false, kThreads>;
    return countOnly ? RegionEmitExactKernel<false, false, true, kThreads> : RegionEmitExactKernel<false, false, false, kThreads>;
}
constexpr int kExactThreadChoices[] = {256, 512, 1024};
RegionKernel ExactRegionKernel(bool raw, bool two, bool countOnly, int threads) {
    if (threads == 1024) return ExactRegionKernelOf<1024>(raw, two, countOnly);
    if (threads == 512) return ExactRegionKernelOf<512>(raw, two, countOnly);
    return ExactRegionKernelOf<256>(raw, two, countOnly);
}
// threads of a workgroup of the exact-key region pass (SSA_REGION_THREADS: experiments)
int ExactRegionThreads() {
    static const int chosen = [] {
        const char* env = std::getenv("SSA_REGION_THREADS");
        const int t = env ? std::atoi(env) : 512;
        return (t == 256 || t == 512 || t == 1024) ? t : 512;
    }();
    return chosen;
}
}  // namespace

hipError_t ConfigureRegionKernels(int maxLdsBytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&RegionEmitKernel), hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    for (int threads : kExactThreadChoices) {
        for (int v = 0; v < 6 && e == hipSuccess; ++v) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(ExactRegionKernel(v / 2 == 0, v / 2 == 1, v % 2 != 0, threads)),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
        }
    }
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&RegionApplyKernel), hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    return e;
}

hipError_t LaunchRegionEmit(const RegionEmitArgs& a, hipStream_t st) {
    const int grid = a.nregions < kMaxGrid * 16 ? a.nregions : kMaxGrid * 16;
    if (a.s.hashed) {
        hipLaunchKernelGGL(RegionEmitKernel, dim3(uint32_t(grid)), dim3(kBlock), RegionEmitLdsBytes(a.s, a.region_slots), st, a);
    } else {
        const bool raw = a.s.rec_raw64 != 0, two = a.s.two_words != 0 && !raw;
        const bool countOnly = CountOnlySink(a.s) && a.rec.mask == nullptr;
        const int threads = ExactRegionThreads();
        hipLaunchKernelGGL(ExactRegionKernel(raw, two, countOnly, threads), dim3(uint32_t(grid)), dim3(uint32_t(threads)), RegionEmitLdsBytes(a.s, a.region_slots), st, a);
    }
    return hipGetLastError();
}

hipError_t LaunchRegionCompactEmit(const RegionEmitArgs& a, hipStream_t st) {
    const int grid = a.nregions < kMaxGrid * 16 ? a.nregions : kMaxGrid * 16;
    hipLaunchKernelGGL(RegionCompactEmitKernel, dim3(uint32_t(grid)), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t LaunchKeyedTileCount(const uint64_t* tKeyLo, uint64_t capacity, uint32_t* tileCounts, hipStream_t st) {
    const uint64_t ntiles = capacity / kTileRows;
    hipLaunchKernelGGL(KeyedTileCountKernel, dim3(uint32_t(ntiles < uint64_t(kMaxGrid) ? ntiles : uint64_t(kMaxGrid))), dim3(kBlock), 0, st,
                       tKeyLo, capacity, tileCounts);
    return hipGetLastError();
}

hipError_t LaunchKeyedEmit(const KeyedFinalizeArgs& a, hipStream_t st) {
    const uint64_t ntiles = a.capacity / kTileRows;
    hipLaunchKernelGGL(KeyedEmitKernel, dim3(uint32_t(ntiles < uint64_t(kMaxGrid) ? ntiles : uint64_t(kMaxGrid))), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t LaunchSmallTablesEmit(const SmallEmitArgs& a, int tables, hipStream_t st) {
    if (tables < 1 || a.slice_cap < 64 || a.slice_cap > uint32_t(kSmallTableSlots) || (a.slice_cap & (a.slice_cap - 1)) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(SmallTablesEmitKernel, dim3(uint32_t(tables)), dim3(kSmallEmitBlock), 0, st, a);
    return hipGetLastError();
}

hipError_t LaunchCombineDecimal(const uint64_t* lo32, const uint64_t* mid32, const uint64_t* hi, int64_t n, uint64_t* out,
                                hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(CombineDecimalKernel, dim3(GridFor(n)), dim3(kBlock), 0, st, lo32, mid32, hi, n, out);
    return hipGetLastError();
}

hipError_t LaunchGatherFixed(const void* src, const uint8_t* srcValid, int64_t srcValidOff, int width, const uint64_t* rows,
                             int64_t n, void* dst, uint8_t* dstValidBytes, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int grid = GridFor(n);
#define SSA_GATHER(T) \
    hipLaunchKernelGGL((GatherFixedKernel<T>), dim3(grid), dim3(kBlock), 0, st, (const T*)src, srcValid, srcValidOff, rows, n, (T*)dst, dstValidBytes)
    switch (width) {
        case 1: SSA_GATHER(uint8_t); break;
        case 2: SSA_GATHER(uint16_t); break;
        case 4: SSA_GATHER(uint32_t); break;
        case 8: SSA_GATHER(uint64_t); break;
        case 16: SSA_GATHER(G128); break;
        default: return hipErrorInvalidValue;
    }
#undef SSA_GATHER
    return hipGetLastError();
}

hipError_t LaunchProbeI32(const ProbeWords& p, hipStream_t st) {
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ProbeI32Kernel, dim3(1), dim3(64), 0, st, p);
    return hipGetLastError();
}

hipError_t LaunchGatherI32At(const int32_t* const* addrs, int64_t n, int32_t* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(GatherI32AtKernel, dim3(GridFor(n)), dim3(kBlock), 0, st, addrs, n, out);
    return hipGetLastError();
}

hipError_t LaunchIotaI32(int32_t* out, int64_t n, int32_t step, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(IotaI32Kernel, dim3(GridFor(n)), dim3(kBlock), 0, st, out, n, step);
    return hipGetLastError();
}

hipError_t LaunchGatherValidity(const uint8_t* srcValid, int64_t srcValidOff, const uint64_t* rows, int64_t n,
                                uint8_t* dstValidBytes, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(GatherValidityKernel, dim3(GridFor(n)), dim3(kBlock), 0, st, srcValid, srcValidOff, rows, n, dstValidBytes);
    return hipGetLastError();
}

hipError_t LaunchGatherStringLens(const int32_t* offsets, const uint64_t* rows, int64_t n, int32_t* outLens, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(GatherStringLensKernel, dim3(GridFor(n)), dim3(kBlock), 0, st, offsets, rows, n, outLens);
    return hipGetLastError();
}

hipError_t LaunchGatherStringBytes(const int32_t* offsets, const uint8_t* chars, const uint64_t* rows, int64_t n,
                                   const int32_t* newOffsets, uint8_t* outChars, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(GatherStringBytesKernel, dim3(GridFor(n * 8)), dim3(kBlock), 0, st, offsets, chars, rows, n, newOffsets, outChars);
    return hipGetLastError();
}

}  // namespace ssa
