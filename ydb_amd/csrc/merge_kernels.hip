// K-way merge of sorted sources with replacement of older versions: the device side of
// NArrow::NMerger::TMergePartialStream::DrainAll (ydb/core/formats/arrow/reader/merger.cpp:145-186).
//
// The reference keeps the sources in a heap ordered by (key ascending, version descending) (batch_iterator.cpp:33-49),
// emits the top record of every distinct key and skips the other records of that key.  Here the rows of all sources
// are one array of row ids, sorted by merging pairs of runs round after round (the sources are the first runs): every
// round cuts its output into tiles along the merge path (one binary search per tile border), and inside a tile every
// element finds its place by a binary search in the tile's part of the other run.  The order is total -- key columns,
// then version columns descending, then the row id -- so ranks never tie.  A source's iterator only ever shows the heap
// the first row of a run of equal keys (the others are skipped with it, merger.cpp:169-184): only those rows enter the
// merge.  Rows compare as the reference compares them
// (switch/compare.h:82-119, the not-null form: validity is not looked at; strings by memcmp then length).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ssa {

namespace {

constexpr int kMergeTile = 2048;  // output rows per workgroup

__device__ __forceinline__ int CompareColumn(const MergeKeyColumn& c, uint64_t a, uint64_t b) {
    switch (c.kind) {
        case PK_SIGNED: {
            int64_t x, y;
            switch (c.width) {
                case 1: x = static_cast<const int8_t*>(c.values)[a], y = static_cast<const int8_t*>(c.values)[b]; break;
                case 2: x = static_cast<const int16_t*>(c.values)[a], y = static_cast<const int16_t*>(c.values)[b]; break;
                case 4: x = static_cast<const int32_t*>(c.values)[a], y = static_cast<const int32_t*>(c.values)[b]; break;
                default: x = static_cast<const int64_t*>(c.values)[a], y = static_cast<const int64_t*>(c.values)[b]; break;
            }
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        case PK_UNSIGNED: {
            uint64_t x, y;
            switch (c.width) {
                case 1: x = static_cast<const uint8_t*>(c.values)[a], y = static_cast<const uint8_t*>(c.values)[b]; break;
                case 2: x = static_cast<const uint16_t*>(c.values)[a], y = static_cast<const uint16_t*>(c.values)[b]; break;
                case 4: x = static_cast<const uint32_t*>(c.values)[a], y = static_cast<const uint32_t*>(c.values)[b]; break;
                default: x = static_cast<const uint64_t*>(c.values)[a], y = static_cast<const uint64_t*>(c.values)[b]; break;
            }
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        case PK_F32: {
            const float x = static_cast<const float*>(c.values)[a], y = static_cast<const float*>(c.values)[b];
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        case PK_F64: {
            const double x = static_cast<const double*>(c.values)[a], y = static_cast<const double*>(c.values)[b];
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        case PK_DEC128: {  // 16-byte little-endian two's complement
            const uint64_t* p = static_cast<const uint64_t*>(c.values);
            const int64_t xh = int64_t(p[2 * a + 1]), yh = int64_t(p[2 * b + 1]);
            if (xh != yh) return xh < yh ? -1 : 1;
            const uint64_t xl = p[2 * a], yl = p[2 * b];
            return xl < yl ? -1 : (xl > yl ? 1 : 0);
        }
        case PK_STRING: {
            const int32_t* offs = static_cast<const int32_t*>(c.values);
            const int32_t sa = offs[a], na = offs[a + 1] - sa, sb = offs[b], nb = offs[b + 1] - sb;
            const int32_t m = na < nb ? na : nb;
            const uint8_t* p = c.chars + sa;
            const uint8_t* q = c.chars + sb;
            for (int32_t i = 0; i < m; ++i) {
                if (p[i] != q[i]) return p[i] < q[i] ? -1 : 1;
            }
            return na < nb ? -1 : (na > nb ? 1 : 0);
        }
        default: return 0;  // (booleans compare as equal in the reference: compare.h:12-14)
    }
}

__device__ __forceinline__ int CompareKeys(const MergeKeys& k, uint64_t a, uint64_t b) {
    for (int c = 0; c < k.nkeys; ++c) {
        const int r = CompareColumn(k.cols[c], a, b);
        if (r != 0) return r;
    }
    return 0;
}

// The total order of the merge: key ascending, version descending, row id ascending.
__device__ __forceinline__ bool Before(const MergeKeys& k, uint64_t a, uint64_t b) {
    int r = CompareKeys(k, a, b);
    if (r != 0) return r < 0;
    for (int c = k.nkeys; c < k.nkeys + k.nversions; ++c) {
        r = CompareColumn(k.cols[c], a, b);
        if (r != 0) return r > 0;
    }
    return a < b;
}

// How many elements of run A = in[s, m) come before diagonal `d` of the merge of A and B = in[m, e).
__device__ uint64_t PathSplit(const MergeKeys& k, const uint64_t* in, uint64_t s, uint64_t m, uint64_t e, uint64_t d) {
    const uint64_t na = m - s, nb = e - m;
    uint64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;  // A[mid] against B[d - 1 - mid]
        if (Before(k, in[m + (d - 1 - mid)], in[s + mid])) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(kBlock) void MergePartitionKernel(const MergeKeys k, const uint64_t* in, const MergeTile* tiles, int64_t ntiles, uint64_t* splits) {
    const int64_t t = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (t >= ntiles) return;
    const MergeTile tile = tiles[t];
    const uint64_t n = tile.e - tile.s;
    const uint64_t d1 = tile.d0 + kMergeTile < n ? tile.d0 + kMergeTile : n;
    splits[2 * t] = PathSplit(k, in, tile.s, tile.m, tile.e, tile.d0);
    splits[2 * t + 1] = PathSplit(k, in, tile.s, tile.m, tile.e, d1);
}

__global__ __launch_bounds__(kBlock) void MergeTilesKernel(const MergeKeys k, const uint64_t* in, uint64_t* out, const MergeTile* tiles, const uint64_t* splits) {
    __shared__ uint64_t ids[kMergeTile];
    const MergeTile tile = tiles[blockIdx.x];
    const uint64_t n = tile.e - tile.s;
    const uint64_t d1 = tile.d0 + kMergeTile < n ? tile.d0 + kMergeTile : n;
    const uint64_t i0 = splits[2 * blockIdx.x], i1 = splits[2 * blockIdx.x + 1];
    const uint64_t j0 = tile.d0 - i0, j1 = d1 - i1;
    const int na = int(i1 - i0), nb = int(j1 - j0);
    for (int t = threadIdx.x; t < na + nb; t += kBlock) ids[t] = t < na ? in[tile.s + i0 + t] : in[tile.m + j0 + (t - na)];
    __syncthreads();
    uint64_t* dst = out + tile.s + tile.d0;
    for (int t = threadIdx.x; t < na + nb; t += kBlock) {
        const uint64_t x = ids[t];
        int lo = 0, hi, base;
        if (t < na) {  // elements of B's part that come before x
            hi = nb, base = na;
        } else {
            hi = na, base = 0;
        }
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (Before(k, ids[base + mid], x)) lo = mid + 1;
            else hi = mid;
        }
        dst[(t < na ? t : t - na) + lo] = x;
    }
}

// bit i: the row at place i of the sorted order is the first of its key (the newest version) and is not deleted
__global__ __launch_bounds__(kBlock) void MergeWinnersKernel(const MergeKeys k, const uint64_t* sorted, int64_t n, const uint8_t* keep, uint64_t* bits) {
    const int64_t nwords = (n + 63) >> 6;
    const int64_t wavesTotal = int64_t(gridDim.x) * (kBlock / 64);
    for (int64_t w = int64_t(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6); w < nwords; w += wavesTotal) {
        const int64_t i = w * 64 + (threadIdx.x & 63);
        bool win = false;
        if (i < n) {
            const uint64_t row = sorted[i];
            win = (i == 0 || CompareKeys(k, sorted[i - 1], row) != 0) && (keep == nullptr || keep[row] != 0);
        }
        const uint64_t word = __ballot(win);
        if ((threadIdx.x & 63) == 0) bits[w] = word;
    }
}

// bit r: row r is the row of its key that its source's iterator meets first -- the first of a run of equal keys inside
// the source (the last when the sources are read from their ends).  boundary[r] = 1: a source begins at row r.
__global__ __launch_bounds__(kBlock) void MergeCandidatesKernel(const MergeKeys k, int64_t n, const uint8_t* boundary, int reverse, uint64_t* bits) {
    const int64_t nwords = (n + 63) >> 6;
    const int64_t wavesTotal = int64_t(gridDim.x) * (kBlock / 64);
    for (int64_t w = int64_t(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6); w < nwords; w += wavesTotal) {
        const int64_t r = w * 64 + (threadIdx.x & 63);
        bool first = false;
        if (r < n) {
            if (!reverse) first = boundary[r] != 0 || CompareKeys(k, uint64_t(r - 1), uint64_t(r)) != 0;
            else first = r + 1 == n || boundary[r + 1] != 0 || CompareKeys(k, uint64_t(r), uint64_t(r + 1)) != 0;
        }
        const uint64_t word = __ballot(first);
        if ((threadIdx.x & 63) == 0) bits[w] = word;
    }
}

// out[j] = number of ids (ascending) below bounds[j]
__global__ void CountBelowKernel(const uint64_t* ids, int64_t n, const uint64_t* bounds, int nbounds, uint64_t* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nbounds) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids[mid] < bounds[j]) lo = mid + 1;
        else hi = mid;
    }
    out[j] = uint64_t(lo);
}

__global__ __launch_bounds__(kBlock) void IotaU64Kernel(uint64_t* out, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) out[i] = uint64_t(i);
}

__global__ __launch_bounds__(kBlock) void ReverseU64Kernel(const uint64_t* in, int64_t n, uint64_t* out) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) out[i] = in[n - 1 - i];
}

int GridOf(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return int(g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g));
}

}  // namespace

int MergeTileRows() { return kMergeTile; }

hipError_t LaunchMergeRound(const MergeKeys& keys, const uint64_t* in, uint64_t* out, const MergeTile* deviceTiles, int64_t ntiles, uint64_t* splits,
                            hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(MergePartitionKernel, dim3(unsigned((ntiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, keys, in, deviceTiles, ntiles, splits);
    hipLaunchKernelGGL(MergeTilesKernel, dim3(unsigned(ntiles)), dim3(kBlock), 0, st, keys, in, out, deviceTiles, splits);
    return hipGetLastError();
}

hipError_t LaunchMergeWinners(const MergeKeys& keys, const uint64_t* sorted, int64_t n, const uint8_t* keep, uint64_t* bits, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(MergeWinnersKernel, dim3(GridOf((n + 63) / 64 * 64)), dim3(kBlock), 0, st, keys, sorted, n, keep, bits);
    return hipGetLastError();
}

hipError_t LaunchMergeCandidates(const MergeKeys& keys, int64_t n, const uint8_t* boundary, bool reverse, uint64_t* bits, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(MergeCandidatesKernel, dim3(GridOf(n)), dim3(kBlock), 0, st, keys, n, boundary, reverse ? 1 : 0, bits);
    return hipGetLastError();
}

hipError_t LaunchCountBelow(const uint64_t* ids, int64_t n, const uint64_t* bounds, int nbounds, uint64_t* out, hipStream_t st) {
    if (nbounds <= 0) return hipSuccess;
    hipLaunchKernelGGL(CountBelowKernel, dim3(unsigned((nbounds + 63) / 64)), dim3(64), 0, st, ids, n, bounds, nbounds, out);
    return hipGetLastError();
}

hipError_t LaunchIotaU64(uint64_t* out, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(IotaU64Kernel, dim3(GridOf(n)), dim3(kBlock), 0, st, out, n);
    return hipGetLastError();
}

hipError_t LaunchReverseU64(const uint64_t* in, int64_t n, uint64_t* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ReverseU64Kernel, dim3(GridOf(n)), dim3(kBlock), 0, st, in, n, out);
    return hipGetLastError();
}

}  // namespace ssa
