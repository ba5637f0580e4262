// Fused scan kernels: expression VM + {key-less aggregation | filter bitmap | projection} sinks, and
// the small follow-up kernels of stream compaction (tile-count scan, column compaction, bit packing).
//
// All of these are HBM-streaming kernels: the algorithmic traffic is "each referenced source column
// once" and the roofline that bounds them is HBM bandwidth.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "filter_project_scan.h"
#include "nokey_scan.h"

namespace ssa {

namespace {


// One workgroup folds the per-workgroup partials in a fixed order (bit-reproducible for a fixed grid), then writes the
// result row as the output columns hold it: per output column four words of `result` (value, its high half for
// decimals, a validity word, padding).  What the host needs to finish the batch -- the scan's status word and which
// aggregates are null -- goes to `host` (pinned memory, when the executor has it), so that the apply ends with one wait
// on the stream and no copy.
struct FinalizeArgs {
    NoKeySink s;
    int grid;
    int nouts;
    NoKeyOutSpec outs[kMaxOuts];
    uint64_t* result;
    const uint32_t* status;
    uint32_t* host;  // [0] status, [1] bit k: output k is not null; may be nullptr (then `result` word 4 * kMaxOuts and on)
};
__global__ __launch_bounds__(kBlock) void NoKeyFinalizeKernel(const FinalizeArgs) {
    const FinalizeArgs& fa = KernArgsAs<FinalizeArgs>();
    const NoKeySink& s = fa.s;
    const int grid = fa.grid;
    __shared__ uint64_t lds[kWaves];
    __shared__ uint32_t ldsSeen[kWaves];
    __shared__ uint64_t vals[kMaxAggs];
    uint32_t seen = 0;
    for (int b = threadIdx.x; b < grid; b += kBlock) seen |= s.partial_seen[b];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if (LaneId() == 0) ldsSeen[WaveId()] = seen;
    __syncthreads();
    seen = ldsSeen[0] | ldsSeen[1] | ldsSeen[2] | ldsSeen[3];
    for (int k = 0; k < s.naggs; ++k) {
        const int kind = s.aggs[k].kind;
        uint64_t v = AggIdentity(kind);
        for (int b = threadIdx.x; b < grid; b += kBlock) v = AggCombine(kind, v, s.partial[size_t(b) * kMaxAggs + k]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = AggCombine(kind, v, ShflXor64(v, off));
        __syncthreads();
        if (LaneId() == 0) lds[WaveId()] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t r = lds[0];
            for (int w = 1; w < kWaves; ++w) r = AggCombine(kind, r, lds[w]);
            vals[k] = r;
            if (kind == AGG_COUNT) seen |= 1u << k;
        }
    }
    if (threadIdx.x != 0) return;
    uint32_t validOuts = 0;
    for (int o = 0; o < fa.nouts; ++o) {
        const NoKeyOutSpec& os = fa.outs[o];
        const bool valid = (seen >> os.agg) & 1u;
        uint64_t lo = vals[os.agg], hi = 0;
        if (os.mode == NOKEY_OUT_DECIMAL) {  // 128-bit sum from the sums of its three parts
            const unsigned __int128 total = (unsigned __int128)lo + ((unsigned __int128)vals[os.mid] << 32) + ((unsigned __int128)vals[os.hi] << 64);
            lo = uint64_t(total);
            hi = uint64_t(total >> 64);
        } else if (os.mode == NOKEY_OUT_ROW) {  // the value is a row index: that row of the source column
            const uint64_t row = lo;
            lo = 0;
            if (valid) {
                const char* at = static_cast<const char*>(os.src) + row * uint64_t(os.width);
                for (int b = 0; b < os.width; ++b) lo |= uint64_t(uint8_t(at[b])) << (8 * b);
            }
        } else if (os.mode == NOKEY_OUT_F32) {
            lo = __float_as_uint(float(__longlong_as_double((long long)lo)));
        }
        if (!valid) lo = hi = 0;
        fa.result[4 * o] = lo;
        fa.result[4 * o + 1] = hi;
        fa.result[4 * o + 2] = valid ? 1 : 0;
        if (valid) validOuts |= 1u << o;
    }
    uint32_t* host = fa.host ? fa.host : reinterpret_cast<uint32_t*>(fa.result + 4 * kMaxOuts);
    host[0] = *fa.status;
    host[1] = validOuts;
}

// ---- exclusive scans ------------------------------------------------------------------------------
constexpr int kScanItems = 16;
constexpr int kScanChunk = kBlock * kScanItems;  // elements per workgroup

template <typename TIn>
__global__ __launch_bounds__(kBlock) void ScanBlockSumsKernel(const TIn* in, int64_t n, int64_t* blockSums) {
    __shared__ int64_t lds[kWaves];
    const int64_t base = int64_t(blockIdx.x) * kScanChunk;
    int64_t sum = 0;
    for (int i = threadIdx.x; i < kScanChunk; i += kBlock) {
        const int64_t idx = base + i;
        if (idx < n) sum += int64_t(in[idx]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += int64_t(ShflXor64(uint64_t(sum), off));
    if (LaneId() == 0) lds[WaveId()] = sum;
    __syncthreads();
    if (threadIdx.x == 0) blockSums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// Single workgroup: exclusive scan of blockSums in place; total -> *total (may be nullptr).
__global__ __launch_bounds__(kBlock) void ScanOfBlockSumsKernel(int64_t* blockSums, int64_t nblocks, int64_t* total) {
    __shared__ int64_t lds[kBlock];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += kBlock) {
        const int64_t idx = base + threadIdx.x;
        const int64_t v = idx < nblocks ? blockSums[idx] : 0;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kBlock; off <<= 1) {  // Hillis-Steele inclusive scan
            int64_t t = threadIdx.x >= off ? lds[threadIdx.x - off] : 0;
            __syncthreads();
            lds[threadIdx.x] += t;
            __syncthreads();
        }
        const int64_t incl = lds[threadIdx.x];
        const int64_t c = carry;
        if (idx < nblocks) blockSums[idx] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total != nullptr) *total = carry;
}

// Short inputs (one chunk): sums, scan and total in a single launch.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock) void ScanSingleChunkKernel(const TIn* in, int64_t n, TOut* out, int64_t* total, int appendTotal) {
    __shared__ int64_t waveSums[kWaves];
    const int64_t base = int64_t(threadIdx.x) * kScanItems;
    int64_t vals[kScanItems];
    int64_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        vals[i] = base + i < n ? int64_t(in[base + i]) : 0;
        sum += vals[i];
    }
    int64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t t = int64_t(__shfl_up(static_cast<unsigned long long>(incl), off, 64));
        if (LaneId() >= off) incl += t;
    }
    if (LaneId() == 63) waveSums[WaveId()] = incl;
    __syncthreads();
    int64_t waveOff = 0, all = 0;
    for (int w = 0; w < kWaves; ++w) {
        if (w < WaveId()) waveOff += waveSums[w];
        all += waveSums[w];
    }
    int64_t run = waveOff + incl - sum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = TOut(run);
        run += vals[i];
    }
    if (threadIdx.x == 0) {
        if (total != nullptr) *total = all;
        if (appendTotal) out[n] = TOut(all);
    }
}

// out[i] = blockOffset + exclusive prefix inside the chunk.  When `appendTotal` is set out[n] = total.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(kBlock) void ScanApplyKernel(const TIn* in, int64_t n, const int64_t* blockOffsets, TOut* out,
                                                        const int64_t* total, int appendTotal) {
    __shared__ int64_t waveSums[kWaves];
    const int64_t base = int64_t(blockIdx.x) * kScanChunk + int64_t(threadIdx.x) * kScanItems;
    int64_t vals[kScanItems];
    int64_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + i;
        vals[i] = idx < n ? int64_t(in[idx]) : 0;
        sum += vals[i];
    }
    // exclusive scan of per-thread sums inside the wave, then across waves
    int64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int64_t t = int64_t(__shfl_up(static_cast<unsigned long long>(incl), off, 64));
        if (LaneId() >= off) incl += t;
    }
    if (LaneId() == 63) waveSums[WaveId()] = incl;
    __syncthreads();
    int64_t waveOff = 0;
    for (int w = 0; w < WaveId(); ++w) waveOff += waveSums[w];
    int64_t run = blockOffsets[blockIdx.x] + waveOff + incl - sum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + i;
        if (idx < n) out[idx] = TOut(run);
        run += vals[i];
    }
    if (appendTotal && blockIdx.x == 0 && threadIdx.x == 0) out[n] = TOut(*total);
}

// ---- stream compaction of source columns --------------------------------------------------------------
// One lane per row; a wave covers exactly one 64-row bitmap word.
__device__ __forceinline__ bool RowPosition(const uint64_t* bitmap, const int64_t* tileOffsets, int64_t row, int64_t nrows,
                                            int64_t& pos) {
    const int64_t nwords = (nrows + 63) >> 6;
    const int64_t wordIdx = row >> 6;  // wave-uniform
    const int64_t tile = row / kTileRows;
    const int64_t tileFirstWord = tile * (kTileRows / 64);
    const int wordInTile = int(wordIdx - tileFirstWord);
    const uint64_t word = wordIdx < nwords ? bitmap[wordIdx] : 0;
    const uint32_t before = SelectedBeforeWord(bitmap, tileFirstWord, wordInTile, nwords);
    const int lane = LaneId();
    pos = tileOffsets[tile < (nrows + kTileRows - 1) / kTileRows ? tile : 0] + before + __popcll(word & ((1ull << lane) - 1ull));
    return row < nrows && ((word >> lane) & 1ull);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void CompactFixedKernel(const T* src, T* dst, int64_t nrows, const uint64_t* bitmap,
                                                           const int64_t* tileOffsets) {
    const int64_t nloops = (nrows + int64_t(gridDim.x) * kBlock - 1) / (int64_t(gridDim.x) * kBlock);
    for (int64_t it = 0; it < nloops; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        if ((row & ~63ll) >= nrows) continue;  // wave-uniform exit
        int64_t pos;
        if (RowPosition(bitmap, tileOffsets, row, nrows, pos)) dst[pos] = src[row];
    }
}

__global__ __launch_bounds__(kBlock) void CompactBitsKernel(const uint8_t* srcBits, int64_t srcBitOff, uint8_t* dstBytes,
                                                          int64_t nrows, const uint64_t* bitmap, const int64_t* tileOffsets) {
    const int64_t nloops = (nrows + int64_t(gridDim.x) * kBlock - 1) / (int64_t(gridDim.x) * kBlock);
    for (int64_t it = 0; it < nloops; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        if ((row & ~63ll) >= nrows) continue;
        int64_t pos;
        if (RowPosition(bitmap, tileOffsets, row, nrows, pos)) dstBytes[pos] = uint8_t(LoadBit(srcBits, row + srcBitOff));
    }
}

__global__ __launch_bounds__(kBlock) void CompactStringLensKernel(const int32_t* offsets, int32_t* outLens, int64_t nrows,
                                                                const uint64_t* bitmap, const int64_t* tileOffsets) {
    const int64_t nloops = (nrows + int64_t(gridDim.x) * kBlock - 1) / (int64_t(gridDim.x) * kBlock);
    for (int64_t it = 0; it < nloops; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        if ((row & ~63ll) >= nrows) continue;
        int64_t pos;
        if (RowPosition(bitmap, tileOffsets, row, nrows, pos)) outLens[pos] = offsets[row + 1] - offsets[row];
    }
}

__global__ __launch_bounds__(kBlock) void CompactStringBytesKernel(const int32_t* offsets, const uint8_t* chars,
                                                                 const int32_t* newOffsets, uint8_t* outChars, int64_t nrows,
                                                                 const uint64_t* bitmap, const int64_t* tileOffsets) {
    const int64_t nloops = (nrows + int64_t(gridDim.x) * kBlock - 1) / (int64_t(gridDim.x) * kBlock);
    for (int64_t it = 0; it < nloops; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
        if ((row & ~63ll) >= nrows) continue;
        int64_t pos;
        if (RowPosition(bitmap, tileOffsets, row, nrows, pos)) {
            const int32_t b = offsets[row], n = offsets[row + 1] - b, o = newOffsets[pos];
            for (int32_t i = 0; i < n; ++i) outChars[o + i] = chars[b + i];
        }
    }
}

// 8 bytes -> 1 bitmap byte; counts zero bytes (nulls / false values).
__global__ __launch_bounds__(kBlock) void PackBytesToBitsKernel(const uint8_t* src, int64_t n, uint8_t* dstBits,
                                                              unsigned long long* zeroCount) {
    const int64_t nbytes = (n + 7) >> 3;
    unsigned long long zeros = 0;
    for (int64_t ob = int64_t(blockIdx.x) * kBlock + threadIdx.x; ob < nbytes; ob += int64_t(gridDim.x) * kBlock) {
        uint32_t b = 0;
        for (int i = 0; i < 8; ++i) {
            const int64_t idx = ob * 8 + i;
            if (idx < n) {
                if (src[idx]) b |= 1u << i; else ++zeros;
            }
        }
        dstBits[ob] = uint8_t(b);
    }
    if (zeroCount != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) zeros += ShflXor64(zeros, off);
        if (LaneId() == 0 && zeros) atomicAdd(zeroCount, zeros);
    }
}

__global__ __launch_bounds__(kBlock) void CountBitsKernel(const uint8_t* bits, int64_t bitOff, int64_t nbits,
                                                        unsigned long long* out) {
    unsigned long long c = 0;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < nbits; i += int64_t(gridDim.x) * kBlock) {
        c += LoadBit(bits, i + bitOff);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += ShflXor64(c, off);
    if (LaneId() == 0 && c) atomicAdd(out, c);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void FillKernel(T* dst, T value, int64_t n) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) dst[i] = value;
}

struct U128 {
    uint64_t lo, hi;
};

int GridFor(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return int(g);
}

}  // namespace

// ---- launchers ------------------------------------------------------------------------------------
#define SSA_LAUNCH_SCAN(KERNEL, SINK, A, S, CFG)                                                        \
    do {                                                                                                \
        KArgs<SINK> k{A, S};                                                                            \
        hipLaunchKernelGGL((KERNEL<kRowsPerThread>), dim3((CFG).grid), dim3(kBlock), (CFG).lds_bytes, (CFG).stream, k); \
        return hipGetLastError();                                                                       \
    } while (0)

hipError_t LaunchScanNoKey(const ScanArgs& a, const NoKeySink& s, LaunchCfg cfg) {
    SSA_LAUNCH_SCAN(ScanNoKeyKernel, NoKeySink, a, s, cfg);
}
hipError_t LaunchScanFilter(const ScanArgs& a, const FilterSink& s, LaunchCfg cfg) {
    SSA_LAUNCH_SCAN(ScanFilterKernel, FilterSink, a, s, cfg);
}
hipError_t LaunchScanProject(const ScanArgs& a, const ProjectSink& s, LaunchCfg cfg) {
    SSA_LAUNCH_SCAN(ScanProjectKernel, ProjectSink, a, s, cfg);
}

int ScanKernelBlocksPerCU(int kernelId, uint32_t ldsBytes) {
    int n = 0;
    hipError_t e = hipSuccess;
    switch (kernelId) {
        case SCAN_NOKEY: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ScanNoKeyKernel<kRowsPerThread>, kBlock, ldsBytes); break;
        case SCAN_FILTER: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ScanFilterKernel<kRowsPerThread>, kBlock, ldsBytes); break;
        case SCAN_PROJECT: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ScanProjectKernel<kRowsPerThread>, kBlock, ldsBytes); break;
        case SCAN_KEYED_EXACT: return KeyedKernelBlocksPerCU(false, ldsBytes);
        default: return KeyedKernelBlocksPerCU(true, ldsBytes);
    }
    return (e == hipSuccess && n > 0) ? n : 1;
}

hipError_t ConfigureScanKernels(int maxLdsBytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ScanNoKeyKernel<kRowsPerThread>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ScanFilterKernel<kRowsPerThread>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ScanProjectKernel<kRowsPerThread>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, maxLdsBytes);
    if (e == hipSuccess) e = ConfigureKeyedKernels(maxLdsBytes);
    return e;
}

hipError_t LaunchNoKeyFinalize(const NoKeySink& s, int grid, const NoKeyOutSpec* outs, int nouts, uint64_t* result, const uint32_t* status,
                               uint32_t* host, hipStream_t st) {
    if (nouts < 0 || nouts > kMaxOuts) return hipErrorInvalidValue;
    FinalizeArgs fa{};
    fa.s = s;
    fa.grid = grid;
    fa.nouts = nouts;
    for (int o = 0; o < nouts; ++o) fa.outs[o] = outs[o];
    fa.result = result;
    fa.status = status;
    fa.host = host;
    hipLaunchKernelGGL(NoKeyFinalizeKernel, dim3(1), dim3(kBlock), 0, st, fa);
    return hipGetLastError();
}

// block_tmp must hold ceil(n / kScanChunk) int64 values.
template <typename TIn, typename TOut>
static hipError_t ExclusiveScan(const TIn* in, TOut* out, int64_t n, int64_t* blockTmp, int64_t* total, bool appendTotal,
                                hipStream_t st) {
    if (n <= 0) {
        if (total != nullptr) (void)hipMemsetAsync(total, 0, sizeof(int64_t), st);
        if (appendTotal) (void)hipMemsetAsync(out, 0, sizeof(TOut), st);
        return hipGetLastError();
    }
    const int64_t nblocks = (n + kScanChunk - 1) / kScanChunk;
    if (nblocks == 1) {
        hipLaunchKernelGGL((ScanSingleChunkKernel<TIn, TOut>), dim3(1), dim3(kBlock), 0, st, in, n, out, total, appendTotal ? 1 : 0);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((ScanBlockSumsKernel<TIn>), dim3(uint32_t(nblocks)), dim3(kBlock), 0, st, in, n, blockTmp);
    hipLaunchKernelGGL(ScanOfBlockSumsKernel, dim3(1), dim3(kBlock), 0, st, blockTmp, nblocks, total);
    hipLaunchKernelGGL((ScanApplyKernel<TIn, TOut>), dim3(uint32_t(nblocks)), dim3(kBlock), 0, st, in, n, blockTmp, out, total,
                       appendTotal ? 1 : 0);
    return hipGetLastError();
}

// tile_offsets has ntiles + 1 entries (the last one = total); the tail of the same buffer is the block scratch.
hipError_t LaunchTileScan(const uint32_t* tileCounts, int64_t ntiles, int64_t* tileOffsets, int64_t* total, hipStream_t st) {
    int64_t* blockTmp = tileOffsets + ntiles + 1;
    return ExclusiveScan<uint32_t, int64_t>(tileCounts, tileOffsets, ntiles, blockTmp, total, true, st);
}

hipError_t LaunchExclusiveScanI32(const int32_t* in, int32_t* out, int64_t n, int64_t* blockTmp, hipStream_t st) {
    // blockTmp: ceil(n / kScanChunk) + 1 values, the last one receives the total
    const int64_t nblocks = (n + kScanChunk - 1) / kScanChunk;
    return ExclusiveScan<int32_t, int32_t>(in, out, n, blockTmp, blockTmp + nblocks, true, st);
}

hipError_t LaunchCompactFixed(const void* src, void* dst, int width, int64_t nrows, const uint64_t* bitmap,
                              const int64_t* tileOffsets, hipStream_t st) {
    const int grid = GridFor(nrows);
    switch (width) {
        case 1: hipLaunchKernelGGL((CompactFixedKernel<uint8_t>), dim3(grid), dim3(kBlock), 0, st, (const uint8_t*)src, (uint8_t*)dst, nrows, bitmap, tileOffsets); break;
        case 2: hipLaunchKernelGGL((CompactFixedKernel<uint16_t>), dim3(grid), dim3(kBlock), 0, st, (const uint16_t*)src, (uint16_t*)dst, nrows, bitmap, tileOffsets); break;
        case 4: hipLaunchKernelGGL((CompactFixedKernel<uint32_t>), dim3(grid), dim3(kBlock), 0, st, (const uint32_t*)src, (uint32_t*)dst, nrows, bitmap, tileOffsets); break;
        case 8: hipLaunchKernelGGL((CompactFixedKernel<uint64_t>), dim3(grid), dim3(kBlock), 0, st, (const uint64_t*)src, (uint64_t*)dst, nrows, bitmap, tileOffsets); break;
        case 16: hipLaunchKernelGGL((CompactFixedKernel<U128>), dim3(grid), dim3(kBlock), 0, st, (const U128*)src, (U128*)dst, nrows, bitmap, tileOffsets); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t LaunchCompactBits(const uint8_t* srcBits, int64_t srcBitOff, uint8_t* dstBytes, int64_t nrows,
                             const uint64_t* bitmap, const int64_t* tileOffsets, hipStream_t st) {
    hipLaunchKernelGGL(CompactBitsKernel, dim3(GridFor(nrows)), dim3(kBlock), 0, st, srcBits, srcBitOff, dstBytes, nrows, bitmap, tileOffsets);
    return hipGetLastError();
}

hipError_t LaunchPackBytesToBits(const uint8_t* srcBytes, int64_t n, uint8_t* dstBits, unsigned long long* zeroCount,
                                 hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(PackBytesToBitsKernel, dim3(GridFor((n + 7) / 8)), dim3(kBlock), 0, st, srcBytes, n, dstBits, zeroCount);
    return hipGetLastError();
}

hipError_t LaunchCompactStringLens(const int32_t* offsets, int32_t* outLens, int64_t nrows, const uint64_t* bitmap,
                                   const int64_t* tileOffsets, hipStream_t st) {
    hipLaunchKernelGGL(CompactStringLensKernel, dim3(GridFor(nrows)), dim3(kBlock), 0, st, offsets, outLens, nrows, bitmap, tileOffsets);
    return hipGetLastError();
}

hipError_t LaunchCompactStringBytes(const int32_t* offsets, const uint8_t* chars, const int32_t* newOffsets,
                                    uint8_t* outChars, int64_t nrows, const uint64_t* bitmap,
                                    const int64_t* tileOffsets, hipStream_t st) {
    hipLaunchKernelGGL(CompactStringBytesKernel, dim3(GridFor(nrows)), dim3(kBlock), 0, st, offsets, chars, newOffsets, outChars, nrows, bitmap, tileOffsets);
    return hipGetLastError();
}

hipError_t LaunchFill(void* dst, int width, uint64_t value, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int grid = GridFor(n);
    switch (width) {
        case 1: hipLaunchKernelGGL((FillKernel<uint8_t>), dim3(grid), dim3(kBlock), 0, st, (uint8_t*)dst, uint8_t(value), n); break;
        case 2: hipLaunchKernelGGL((FillKernel<uint16_t>), dim3(grid), dim3(kBlock), 0, st, (uint16_t*)dst, uint16_t(value), n); break;
        case 4: hipLaunchKernelGGL((FillKernel<uint32_t>), dim3(grid), dim3(kBlock), 0, st, (uint32_t*)dst, uint32_t(value), n); break;
        case 8: hipLaunchKernelGGL((FillKernel<uint64_t>), dim3(grid), dim3(kBlock), 0, st, (uint64_t*)dst, value, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t LaunchCountBits(const uint8_t* bits, int64_t bitOff, int64_t nbits, unsigned long long* out, hipStream_t st) {
    if (nbits <= 0) return hipSuccess;
    hipLaunchKernelGGL(CountBitsKernel, dim3(GridFor(nbits)), dim3(kBlock), 0, st, bits, bitOff, nbits, out);
    return hipGetLastError();
}

}  // namespace ssa
