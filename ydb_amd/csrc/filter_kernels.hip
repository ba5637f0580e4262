// TColumnFilter on the device: the reference keeps a filter as run lengths of alternating values
// (ydb/core/formats/arrow/arrow_filter.h:20-211); here a filter lives in HBM as an Arrow-order bitmap (bit i of word
// i / 64, LSB first, zero past the end) and these kernels are the data-parallel forms of its operations:
//
//   runs -> bitmap, bitmap -> run starts        the two directions of the RLE representation
//   and / or                                     TColumnFilter::And / Or (arrow_filter.cpp:430-507)
//   sequential and                               TColumnFilter::CombineSequentialAnd (arrow_filter.cpp:509-569): the
//                                                second filter speaks about the rows the first one keeps
//   key rows against a border tuple              TColumnFilter::MakePredicateFilter (arrow_filter.cpp:17-135, 259-311)
//   selected rows per 1024-row tile              the input of the order-preserving compaction that applies a filter
//
// Everything streams bitmaps or key columns once: HBM-bound, tiny next to the scans they gate.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ssa {

namespace {

__device__ __forceinline__ uint64_t TailMask(int64_t nbits, int64_t word) {
    const int64_t left = nbits - word * 64;
    return left >= 64 ? ~0ull : (left <= 0 ? 0ull : ((1ull << left) - 1ull));
}

// starts[j] = first row of run j (starts[nruns] = nbits); run j has value firstValue ^ (j & 1).
__global__ __launch_bounds__(kBlock) void ExpandRunsKernel(const uint32_t* starts, int64_t nruns, int firstValue, int64_t nbits,
                                                          uint64_t* bits) {
    const int64_t nwords = (nbits + 63) >> 6;
    for (int64_t w = int64_t(blockIdx.x) * kBlock + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kBlock) {
        const int64_t lo = w * 64, hi = lo + 64 < nbits ? lo + 64 : nbits;
        // last run that starts at or before `lo`
        int64_t a = 0, b = nruns - 1;
        while (a < b) {
            const int64_t m = (a + b + 1) >> 1;
            if (int64_t(starts[m]) <= lo) a = m; else b = m - 1;
        }
        uint64_t word = 0;
        int64_t pos = lo, j = a;
        while (pos < hi) {
            const int64_t end = int64_t(starts[j + 1]) < hi ? int64_t(starts[j + 1]) : hi;
            if ((firstValue ^ int(j & 1)) && end > pos) {
                const int64_t n = end - pos;
                word |= (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << (pos - lo);
            }
            pos = end;
            ++j;
        }
        bits[w] = word;
    }
}

// Bit i of the result: row i starts a new run (its value differs from row i - 1); row 0 never does.
__device__ __forceinline__ uint64_t Transitions(const uint64_t* bits, int64_t w, int64_t nbits) {
    const uint64_t v = bits[w];
    const uint64_t prev = w == 0 ? (v & 1ull) : (bits[w - 1] >> 63);
    return (v ^ ((v << 1) | prev)) & TailMask(nbits, w);
}

__global__ __launch_bounds__(kBlock) void TransitionCountKernel(const uint64_t* bits, int64_t nbits, uint32_t* counts) {
    const int64_t nwords = (nbits + 63) >> 6;
    for (int64_t w = int64_t(blockIdx.x) * kBlock + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kBlock) {
        counts[w] = uint32_t(__popcll(Transitions(bits, w, nbits)));
    }
}

__global__ __launch_bounds__(kBlock) void TransitionScatterKernel(const uint64_t* bits, int64_t nbits, const int64_t* offsets,
                                                                 uint32_t* starts) {
    const int64_t nwords = (nbits + 63) >> 6;
    for (int64_t w = int64_t(blockIdx.x) * kBlock + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kBlock) {
        uint64_t t = Transitions(bits, w, nbits);
        int64_t at = offsets[w];
        while (t != 0) {
            const int i = __ffsll(static_cast<unsigned long long>(t)) - 1;
            starts[at++] = uint32_t(w * 64 + i);
            t &= t - 1;
        }
    }
}

__global__ __launch_bounds__(kBlock) void AndOrKernel(const uint64_t* a, const uint64_t* b, int64_t nwords, int isOr, uint64_t* out) {
    for (int64_t w = int64_t(blockIdx.x) * kBlock + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kBlock) {
        out[w] = isOr ? (a[w] | b[w]) : (a[w] & b[w]);
    }
}

__global__ __launch_bounds__(kBlock) void WordPopcountKernel(const uint64_t* bits, int64_t nwords, uint32_t* counts) {
    for (int64_t w = int64_t(blockIdx.x) * kBlock + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kBlock) {
        counts[w] = uint32_t(__popcll(bits[w]));
    }
}

// out = self with its k-th set bit replaced by bit k of ext (ranks[w] = set bits of self before word w).
__global__ __launch_bounds__(kBlock) void SequentialAndKernel(const uint64_t* self, int64_t nwords, const int64_t* ranks,
                                                             const uint64_t* ext, int64_t extWords, uint64_t* out) {
    for (int64_t w = int64_t(blockIdx.x) * kBlock + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kBlock) {
        uint64_t m = self[w];
        uint64_t r = 0;
        if (m != 0) {
            const int64_t base = ranks[w];
            const int64_t ew = base >> 6;
            const int sh = int(base & 63);
            const uint64_t w0 = ew < extWords ? ext[ew] : 0ull;
            const uint64_t w1 = ew + 1 < extWords ? ext[ew + 1] : 0ull;
            const uint64_t window = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
            int k = 0;
            while (m != 0) {
                const int i = __ffsll(static_cast<unsigned long long>(m)) - 1;
                r |= ((window >> k) & 1ull) << i;
                m &= m - 1;
                ++k;
            }
        }
        out[w] = r;
    }
}

// Rows against a border tuple, key after key while the row still equals the border (arrow_filter.cpp:30-50): numbers
// by < and > (a NaN is neither: the next key decides), strings by memcmp then length, validity never looked at.
__device__ __forceinline__ int CompareKey(const PredicateKey& c, int64_t row) {
    switch (c.kind) {
        case PK_SIGNED: {
            int64_t v;
            switch (c.width) {
                case 1: v = static_cast<const int8_t*>(c.values)[row]; break;
                case 2: v = static_cast<const int16_t*>(c.values)[row]; break;
                case 4: v = static_cast<const int32_t*>(c.values)[row]; break;
                default: v = static_cast<const int64_t*>(c.values)[row]; break;
            }
            const int64_t b = int64_t(c.border);
            return v < b ? -1 : (v > b ? 1 : 0);
        }
        case PK_UNSIGNED: {
            uint64_t v;
            switch (c.width) {
                case 1: v = static_cast<const uint8_t*>(c.values)[row]; break;
                case 2: v = static_cast<const uint16_t*>(c.values)[row]; break;
                case 4: v = static_cast<const uint32_t*>(c.values)[row]; break;
                default: v = static_cast<const uint64_t*>(c.values)[row]; break;
            }
            return v < c.border ? -1 : (v > c.border ? 1 : 0);
        }
        case PK_F32: {
            const float v = static_cast<const float*>(c.values)[row];
            const float b = __uint_as_float(uint32_t(c.border));
            return v < b ? -1 : (v > b ? 1 : 0);
        }
        case PK_F64: {
            const double v = static_cast<const double*>(c.values)[row];
            const double b = __longlong_as_double(static_cast<long long>(c.border));
            return v < b ? -1 : (v > b ? 1 : 0);
        }
        case PK_BOOL: {
            const int64_t bit = row + c.val_bit_off;
            const uint32_t v = (static_cast<const uint8_t*>(c.values)[bit >> 3] >> (bit & 7)) & 1u;
            const uint32_t b = uint32_t(c.border) & 1u;
            return v < b ? -1 : (v > b ? 1 : 0);
        }
        default: {  // PK_STRING
            const int32_t* offs = static_cast<const int32_t*>(c.values);
            const int32_t s = offs[row], n = offs[row + 1] - s;
            const int32_t bn = int32_t(c.border);
            const int32_t m = n < bn ? n : bn;
            const uint8_t* p = c.chars + s;
            int res = 0;
            for (int32_t i = 0; i < m && res == 0; ++i) {
                const uint8_t x = p[i], y = c.border_chars[i];
                res = x < y ? -1 : (x > y ? 1 : 0);
            }
            if (res == 0) res = n < bn ? -1 : (n > bn ? 1 : 0);
            return res;
        }
    }
}

// A wave takes kPredRows words of 64 rows per trip: the first key of all of them is compared before any later key is
// looked at, so that a wave has that many independent loads in flight (the later keys only matter for the rows that
// equal the border so far -- few, and usually none).
constexpr int kPredRows = 4;
__global__ __launch_bounds__(kBlock) void PredicateKernel(const PredicateKeys keys, int64_t nrows, int compareType, uint64_t* bits) {
    const int lane = threadIdx.x & 63;
    const int64_t nwords = (nrows + 63) >> 6;
    const int64_t ngroups = (nwords + kPredRows - 1) / kPredRows;
    const int64_t wavesTotal = int64_t(gridDim.x) * (kBlock / 64);
    for (int64_t g = int64_t(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6); g < ngroups; g += wavesTotal) {
        int res[kPredRows];
#pragma unroll
        for (int u = 0; u < kPredRows; ++u) {
            const int64_t row = (g * kPredRows + u) * 64 + lane;
            res[u] = row < nrows ? CompareKey(keys.cols[0], row) : 0;
        }
#pragma unroll
        for (int u = 0; u < kPredRows; ++u) {
            const int64_t w = g * kPredRows + u;
            const int64_t row = w * 64 + lane;
            if (row < nrows) {
                for (int k = 1; k < keys.nkeys && res[u] == 0; ++k) res[u] = CompareKey(keys.cols[k], row);
            }
            bool keep;
            switch (compareType) {
                case 1: keep = res[u] < 0; break;   // ECompareType::LESS
                case 2: keep = res[u] <= 0; break;  // LESS_OR_EQUAL
                case 3: keep = res[u] > 0; break;   // GREATER
                default: keep = res[u] >= 0; break; // GREATER_OR_EQUAL
            }
            const uint64_t word = __ballot(keep && row < nrows);
            if (lane == 0 && w < nwords) bits[w] = word;
        }
    }
}

// Selected rows of every tile of kTileRows rows (16 bitmap words).
__global__ __launch_bounds__(kBlock) void BitmapTileCountKernel(const uint64_t* bits, int64_t nwords, int64_t ntiles, uint32_t* counts) {
    constexpr int kWordsPerTile = kTileRows / 64;
    for (int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x; t < ntiles; t += int64_t(gridDim.x) * kBlock) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < kWordsPerTile; ++i) {
            const int64_t w = t * kWordsPerTile + i;
            if (w < nwords) c += uint32_t(__popcll(bits[w]));
        }
        counts[t] = c;
    }
}

int GridForItems(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return int(g);
}

}  // namespace

hipError_t LaunchExpandRuns(const uint32_t* starts, int64_t nruns, int firstValue, int64_t nbits, uint64_t* bits, hipStream_t st) {
    hipLaunchKernelGGL(ExpandRunsKernel, dim3(GridForItems((nbits + 63) >> 6)), dim3(kBlock), 0, st, starts, nruns, firstValue, nbits, bits);
    return hipGetLastError();
}
hipError_t LaunchTransitionCount(const uint64_t* bits, int64_t nbits, uint32_t* counts, hipStream_t st) {
    hipLaunchKernelGGL(TransitionCountKernel, dim3(GridForItems((nbits + 63) >> 6)), dim3(kBlock), 0, st, bits, nbits, counts);
    return hipGetLastError();
}
hipError_t LaunchTransitionScatter(const uint64_t* bits, int64_t nbits, const int64_t* offsets, uint32_t* starts, hipStream_t st) {
    hipLaunchKernelGGL(TransitionScatterKernel, dim3(GridForItems((nbits + 63) >> 6)), dim3(kBlock), 0, st, bits, nbits, offsets, starts);
    return hipGetLastError();
}
hipError_t LaunchBitmapAndOr(const uint64_t* a, const uint64_t* b, int64_t nwords, int isOr, uint64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(AndOrKernel, dim3(GridForItems(nwords)), dim3(kBlock), 0, st, a, b, nwords, isOr, out);
    return hipGetLastError();
}
hipError_t LaunchWordPopcount(const uint64_t* bits, int64_t nwords, uint32_t* counts, hipStream_t st) {
    hipLaunchKernelGGL(WordPopcountKernel, dim3(GridForItems(nwords)), dim3(kBlock), 0, st, bits, nwords, counts);
    return hipGetLastError();
}
hipError_t LaunchSequentialAnd(const uint64_t* self, int64_t nwords, const int64_t* ranks, const uint64_t* ext, int64_t extWords,
                               uint64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(SequentialAndKernel, dim3(GridForItems(nwords)), dim3(kBlock), 0, st, self, nwords, ranks, ext, extWords, out);
    return hipGetLastError();
}
hipError_t LaunchPredicate(const PredicateKeys& keys, int64_t nrows, int compareType, uint64_t* bits, hipStream_t st) {
    const int64_t nwords = (nrows + 63) >> 6;
    hipLaunchKernelGGL(PredicateKernel, dim3(GridForItems((nwords + kPredRows - 1) / kPredRows * 64)), dim3(kBlock), 0, st, keys, nrows, compareType,
                       bits);
    return hipGetLastError();
}
hipError_t LaunchBitmapTileCount(const uint64_t* bits, int64_t nwords, int64_t ntiles, uint32_t* counts, hipStream_t st) {
    hipLaunchKernelGGL(BitmapTileCountKernel, dim3(GridForItems(ntiles)), dim3(kBlock), 0, st, bits, nwords, ntiles, counts);
    return hipGetLastError();
}

}  // namespace ssa
