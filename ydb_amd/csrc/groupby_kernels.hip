// Group-by kernels of the library build: the keyed scan (keyed_scan.h) in its generic form, and the kernels that turn
// the hash table into output columns.
#include <hip/hip_runtime.h>

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
        const uint32_t seen = f.t_seen[i];
        if (f.group_rep_row != nullptr) f.group_rep_row[g] = f.t_rep_row[i];
        for (int k = 0; k < f.naggs; ++k) {
            const int kind = f.aggs[k].kind;
            uint64_t v = f.t_acc[size_t(f.agg_src[k]) * f.capacity + i];
            bool valid = true;
            if (f.agg_row_index[k]) {
                valid = v != ~0ull;  // smallest row with a non-null argument, none -> null
                if (valid) v &= kRowIndexMask;  // (the order of the single-byte-key path sits above the row index)
            } else {
                if (kind == AGG_MIN_F || kind == AGG_MAX_F) {
                    v = F64FromOrdered(v);
                    // the group's first non-null argument was a NaN: the reference's state never leaves it
                    if (f.agg_nan_src[k] >= 0) {
                        const uint64_t first = f.t_acc[size_t(f.agg_nan_src[k]) * f.capacity + i];
                        if (first != ~0ull && (first & 1ull)) v = 0x7FF8000000000000ull;
                    }
                    // a zero result: the first zero that arrived stays, `<` / `>` do not tell the two zeros apart
                    if (f.agg_zero_src[k] >= 0 && (v << 1) == 0) {
                        const uint64_t first = f.t_acc[size_t(f.agg_zero_src[k]) * f.capacity + i];
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
        if (!f.hashed) {
            // undo Finish(): 127 packed bits
            const uint64_t khi = f.two_words ? f.t_key_hi[i] : 0ull;
            const uint64_t lo = klo | (khi << 63), hi = khi >> 1;
            int pos = 0;
            for (int k = 0; k < f.nkeys; ++k) {
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
            }
        }
        }  // slots of this thread
        __syncthreads();  // wordCount is rewritten for the next tile
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

__global__ __launch_bounds__(kBlock) void GatherStringBytesKernel(const int32_t* offsets, const uint8_t* chars,
                                                                const uint64_t* rows, int64_t n, const int32_t* newOffsets,
                                                                uint8_t* outChars) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        const uint64_t r = rows[i];
        if (r == ~0ull) continue;
        const int32_t b = offsets[r], len = offsets[r + 1] - b, o = newOffsets[i];
        for (int32_t j = 0; j < len; ++j) outChars[o + j] = chars[b + j];
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
    hipLaunchKernelGGL(GatherStringBytesKernel, dim3(GridFor(n)), dim3(kBlock), 0, st, offsets, chars, rows, n, newOffsets, outChars);
    return hipGetLastError();
}

}  // namespace ssa
