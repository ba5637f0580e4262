// Compute-sharding of a result batch: the hash of the sharding key per row and the split of the rows into per-shard
// lists -- the step the reference runs right after the program (engines/reader/plain_reader/iterator/merge.cpp:38-45).
//
//   * XX64RowsKernel: XXH64 (xxHash 0.8.2 as vendored under contrib/libs/xxhash, algorithm as published in the xxHash
//     specification) of the concatenated bytes of the row's non-null key fields -- what
//     ydb/core/formats/arrow/hash/calcer.cpp:31-70 computes with a streaming XXH64 state.  One lane per row; the lane
//     keeps the 32-byte stripe buffer and the four lanes of the hash state in registers.
//   * ShardCountKernel / ShardPlaceKernel: a stable counting partition of the rows by shard = |hash| % shards
//     (ydb/core/formats/arrow/permutations.h:84-100): per-block histograms, one exclusive scan over the
//     [shard][block] counts, then every row's position = scanned offset + its rank among the same-shard rows of its
//     block.  Within a shard the row indices stay ascending, as in the reference's per-shard vectors.
//
// All of it streams the key columns once; the bound is HBM bandwidth.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ssa {

namespace {

constexpr uint64_t kP1 = 0x9E3779B185EBCA87ull;
constexpr uint64_t kP2 = 0xC2B2AE3D27D4EB4Full;
constexpr uint64_t kP3 = 0x165667B19E3779F9ull;
constexpr uint64_t kP4 = 0x85EBCA77C2B2AE63ull;
constexpr uint64_t kP5 = 0x27D4EB2F165667C5ull;

__device__ __forceinline__ uint64_t Rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t Round(uint64_t acc, uint64_t input) {
    acc += input * kP2;
    acc = Rotl(acc, 31);
    return acc * kP1;
}
__device__ __forceinline__ uint64_t MergeRound(uint64_t acc, uint64_t val) {
    acc ^= Round(0, val);
    return acc * kP1 + kP4;
}

// Streaming XXH64 state of one row.  `mem` holds the bytes of the unfinished 32-byte stripe, little-endian in four words.
struct XX64 {
    uint64_t v1, v2, v3, v4;
    uint64_t mem[4];
    uint32_t memBytes;
    uint64_t total;
    uint64_t seed;

    __device__ __forceinline__ void Start(uint64_t s) {
        seed = s;
        v1 = s + kP1 + kP2;
        v2 = s + kP2;
        v3 = s;
        v4 = s - kP1;
        mem[0] = mem[1] = mem[2] = mem[3] = 0;
        memBytes = 0;
        total = 0;
    }
    __device__ __forceinline__ void Stripe() {
        v1 = Round(v1, mem[0]);
        v2 = Round(v2, mem[1]);
        v3 = Round(v3, mem[2]);
        v4 = Round(v4, mem[3]);
        mem[0] = mem[1] = mem[2] = mem[3] = 0;
        memBytes = 0;
    }
    // the low `n` bytes of `word` (1 <= n <= 8; bytes above them are zero)
    __device__ __forceinline__ void Put(uint64_t word, uint32_t n) {
        total += n;
        const uint32_t at = memBytes;           // 0..31
        const uint32_t w = at >> 3, sh = (at & 7u) * 8u;
        const uint32_t room = 32u - at;
        // part that lands in word w, part that spills into word w + 1 (or the next stripe)
        const uint64_t lowPart = word << sh;
        const uint64_t highPart = sh ? word >> (64u - sh) : 0ull;
        const uint32_t inFirst = 8u - (at & 7u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (uint32_t(i) == w) mem[i] |= lowPart;
        }
        if (n <= room) {
            if (n > inFirst) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (uint32_t(i) == w + 1) mem[i] |= highPart;
                }
            }
            memBytes = at + n;
            if (memBytes == 32u) Stripe();
        } else {
            // the stripe fills up inside this word (room < n <= 8, so the first word was the stripe's last)
            Stripe();
            mem[0] = highPart;
            memBytes = n - room;
        }
    }
    __device__ __forceinline__ void PutBytes(const uint8_t* p, uint32_t n) {
        uint32_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) w |= uint64_t(p[i + j]) << (8 * j);
            Put(w, 8);
        }
        if (i < n) {
            uint64_t w = 0;
            for (uint32_t j = 0; i + j < n; ++j) w |= uint64_t(p[i + j]) << (8 * j);
            Put(w, n - i);
        }
    }
    __device__ __forceinline__ uint64_t Finish() const {
        uint64_t h;
        if (total >= 32) {
            h = Rotl(v1, 1) + Rotl(v2, 7) + Rotl(v3, 12) + Rotl(v4, 18);
            h = MergeRound(h, v1);
            h = MergeRound(h, v2);
            h = MergeRound(h, v3);
            h = MergeRound(h, v4);
        } else {
            h = seed + kP5;
        }
        h += total;
        uint32_t left = memBytes, at = 0;
        while (left >= 8) {
            uint64_t k = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (uint32_t(i) == (at >> 3)) k = mem[i];
            }
            h ^= Round(0, k);
            h = Rotl(h, 27) * kP1 + kP4;
            at += 8;
            left -= 8;
        }
        uint64_t tail = 0;  // the word that holds the remaining 0..7 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (uint32_t(i) == (at >> 3)) tail = mem[i];
        }
        if (left >= 4) {
            h ^= (tail & 0xFFFFFFFFull) * kP1;
            h = Rotl(h, 23) * kP2 + kP3;
            tail >>= 32;
            left -= 4;
        }
        while (left > 0) {
            h ^= (tail & 0xFFull) * kP5;
            h = Rotl(h, 11) * kP1;
            tail >>= 8;
            --left;
        }
        h ^= h >> 33;
        h *= kP2;
        h ^= h >> 29;
        h *= kP3;
        h ^= h >> 32;
        return h;
    }
};

__global__ __launch_bounds__(kBlock) void XX64RowsKernel(const ShardKeyColumns keys, int64_t nrows, uint64_t seed, uint64_t* out) {
    for (int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x; row < nrows; row += int64_t(gridDim.x) * kBlock) {
        XX64 h;
        h.Start(seed);
        for (int k = 0; k < keys.nkeys; ++k) {
            const ShardKeyColumn& c = keys.cols[k];
            if (c.validity != nullptr) {
                const int64_t bit = row + c.valid_bit_off;
                if (!((c.validity[bit >> 3] >> (bit & 7)) & 1u)) continue;  // a null adds nothing (calcer.cpp:37)
            }
            if (c.chars != nullptr) {
                const int32_t* offs = static_cast<const int32_t*>(c.values);
                const int32_t b = offs[row], e = offs[row + 1];
                h.PutBytes(c.chars + b, uint32_t(e - b));
            } else if (c.width == 0) {  // boolean: one byte, 0 or 1 (sizeof(bool))
                const int64_t bit = row + c.val_bit_off;
                h.Put((static_cast<const uint8_t*>(c.values)[bit >> 3] >> (bit & 7)) & 1u, 1);
            } else {
                uint64_t v;
                switch (c.width) {
                    case 1: v = static_cast<const uint8_t*>(c.values)[row]; break;
                    case 2: v = static_cast<const uint16_t*>(c.values)[row]; break;
                    case 4: v = static_cast<const uint32_t*>(c.values)[row]; break;
                    default: v = static_cast<const uint64_t*>(c.values)[row]; break;
                }
                h.Put(v, uint32_t(c.width));
            }
        }
        out[row] = h.Finish();
    }
}

// |v| % shards for the hash column read as a signed 64-bit value (permutations.h:88-90).  The magnitude of the most
// negative value is taken as 2^63 (negating it overflows in the reference).
__device__ __forceinline__ uint32_t ShardOf(const void* hashes, int hashType, int64_t row, uint32_t shards) {
    int64_t v;
    switch (hashType) {
        case T_U32: v = int64_t(static_cast<const uint32_t*>(hashes)[row]); break;
        case T_I32: v = int64_t(static_cast<const int32_t*>(hashes)[row]); break;
        default: v = static_cast<const int64_t*>(hashes)[row]; break;  // T_I64, T_U64 (reinterpreted)
    }
    const uint64_t mag = v < 0 ? 0ull - uint64_t(v) : uint64_t(v);
    return uint32_t(mag % shards);
}

constexpr int kShardBlockRows = 1024;  // rows per workgroup of the partition kernels: 16 groups of 64

// counts[shard * nblocks + block] = rows of the block in the shard; totals[shard] += the same.  Shard ids come from
// `shardIds` when given (second pass over corrected ids), else from the hash column, and are stored to `shardOut`.
__global__ __launch_bounds__(kBlock) void ShardCountKernel(const void* hashes, int hashType, const uint32_t* shardIds, int64_t nrows,
                                                          uint32_t shards, int64_t nblocks, uint32_t* shardOut, int32_t* counts,
                                                          unsigned long long* totals) {
    extern __shared__ uint32_t hist[];
    for (int64_t block = blockIdx.x; block < nblocks; block += gridDim.x) {
        for (uint32_t s = threadIdx.x; s < shards; s += kBlock) hist[s] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kShardBlockRows / kBlock; ++u) {
            const int64_t row = block * kShardBlockRows + u * kBlock + threadIdx.x;
            const bool live = row < nrows;
            uint32_t s = 0xFFFFFFFFu;
            if (live) {
                s = shardIds ? shardIds[row] : ShardOf(hashes, hashType, row, shards);
                if (shardOut) shardOut[row] = s;
            }
            // one LDS atomic per distinct shard among the wave's 64 rows, not one per row
            uint64_t todo = __ballot(live);
            while (todo != 0) {
                const int leader = __ffsll(static_cast<unsigned long long>(todo)) - 1;
                const uint32_t s0 = uint32_t(__shfl(int(s), leader, 64));
                const uint64_t same = __ballot(live && s == s0);
                if (int(threadIdx.x & 63) == leader) atomicAdd(hist + s0, uint32_t(__popcll(same)));
                todo &= ~same;
            }
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < shards; s += kBlock) {
            const uint32_t c = hist[s];
            counts[int64_t(s) * nblocks + block] = int32_t(c);
            if (c && totals) atomicAdd(totals + s, static_cast<unsigned long long>(c));
        }
        __syncthreads();
    }
}

// Places every row: position = offsets[shard][block] + rank of the row among the block's earlier rows of the same
// shard.  With `perm` the row index is written at that position (the final lists, shard after shard).  With
// `finalShard` the row's rank inside its shard decides where the size correction sends it: ranks below keep[shard]
// stay, the rest belong to the segments seg*[segBegin[shard] .. segBegin[shard + 1]) -- segment j covers ranks from
// segStart[j] up to the next segment's start and moves to shard segTo[j] -- and the corrected shard id is written.
__global__ __launch_bounds__(kBlock) void ShardPlaceKernel(const uint32_t* shardIds, int64_t nrows, uint32_t shards, int64_t nblocks,
                                                          const int32_t* offsets, uint64_t* perm, uint32_t* finalShard,
                                                          const ShardCorrection corr) {
    extern __shared__ uint32_t groupCount[];  // [16 groups][shards], turned into exclusive prefixes over the groups
    constexpr int kGroups = kShardBlockRows / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t block = blockIdx.x; block < nblocks; block += gridDim.x) {
        for (uint32_t i = threadIdx.x; i < kGroups * shards; i += kBlock) groupCount[i] = 0;
        __syncthreads();
        uint32_t myShard[kGroups / 4], myRank[kGroups / 4];
#pragma unroll
        for (int it = 0; it < kGroups / 4; ++it) {
            const int g = it * 4 + wave;  // groups in row order: group g holds rows block * 1024 + g * 64 ...
            const int64_t row = block * kShardBlockRows + g * 64 + lane;
            const bool live = row < nrows;
            const uint32_t s = live ? shardIds[row] : 0xFFFFFFFFu;
            uint32_t rank = 0;
            uint64_t todo = __ballot(live);
            while (todo != 0) {
                const int leader = __ffsll(static_cast<unsigned long long>(todo)) - 1;
                const uint32_t s0 = uint32_t(__shfl(int(s), leader, 64));
                const uint64_t same = __ballot(live && s == s0);
                if (s == s0) rank = uint32_t(__popcll(same & ((1ull << lane) - 1ull)));
                if (lane == leader) groupCount[g * shards + s0] = uint32_t(__popcll(same));
                todo &= ~same;
            }
            myShard[it] = s;
            myRank[it] = rank;
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < shards; s += kBlock) {
            uint32_t run = 0;
            for (int g = 0; g < kGroups; ++g) {
                const uint32_t c = groupCount[g * shards + s];
                groupCount[g * shards + s] = run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kGroups / 4; ++it) {
            const int g = it * 4 + wave;
            const int64_t row = block * kShardBlockRows + g * 64 + lane;
            if (row >= nrows) continue;
            const uint32_t s = myShard[it];
            const int64_t pos = int64_t(offsets[int64_t(s) * nblocks + block]) + groupCount[g * shards + s] + myRank[it];
            if (perm) perm[pos] = uint64_t(row);
            if (finalShard) {
                const uint32_t rankInShard = uint32_t(pos - int64_t(offsets[int64_t(s) * nblocks]));
                uint32_t to = s;
                if (rankInShard >= corr.keep[s]) {
                    for (uint32_t j = corr.seg_begin[s]; j < corr.seg_begin[s + 1]; ++j) {
                        if (rankInShard >= corr.seg_start[j]) to = corr.seg_to[j];
                    }
                }
                finalShard[row] = to;
            }
        }
        __syncthreads();
    }
}

// out[i] = in[i * stride]
__global__ __launch_bounds__(kBlock) void GatherStridedKernel(const int32_t* in, int64_t stride, int count, int32_t* out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) out[i] = in[int64_t(i) * stride];
}

int GridForRows(int64_t n, int rowsPerBlock) {
    int64_t g = (n + rowsPerBlock - 1) / rowsPerBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return int(g);
}

}  // namespace

hipError_t LaunchXX64Rows(const ShardKeyColumns& keys, int64_t nrows, uint64_t seed, uint64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(XX64RowsKernel, dim3(GridForRows(nrows, kBlock)), dim3(kBlock), 0, st, keys, nrows, seed, out);
    return hipGetLastError();
}

hipError_t LaunchGatherStrided(const int32_t* in, int64_t stride, int count, int32_t* out, hipStream_t st) {
    hipLaunchKernelGGL(GatherStridedKernel, dim3(GridForRows(count, kBlock)), dim3(kBlock), 0, st, in, stride, count, out);
    return hipGetLastError();
}

int64_t ShardBlocks(int64_t nrows) { return (nrows + kShardBlockRows - 1) / kShardBlockRows; }

hipError_t LaunchShardCount(const void* hashes, int hashType, const uint32_t* shardIds, int64_t nrows, uint32_t shards,
                            uint32_t* shardOut, int32_t* counts, unsigned long long* totals, hipStream_t st) {
    const int64_t nblocks = ShardBlocks(nrows);
    hipLaunchKernelGGL(ShardCountKernel, dim3(GridForRows(nrows, kShardBlockRows)), dim3(kBlock), shards * sizeof(uint32_t), st, hashes,
                       hashType, shardIds, nrows, shards, nblocks, shardOut, counts, totals);
    return hipGetLastError();
}

hipError_t LaunchShardPlace(const uint32_t* shardIds, int64_t nrows, uint32_t shards, const int32_t* offsets, uint64_t* perm,
                            uint32_t* finalShard, const ShardCorrection& corr, hipStream_t st) {
    const int64_t nblocks = ShardBlocks(nrows);
    hipLaunchKernelGGL(ShardPlaceKernel, dim3(GridForRows(nrows, kShardBlockRows)), dim3(kBlock),
                       (kShardBlockRows / 64) * shards * sizeof(uint32_t), st, shardIds, nrows, shards, nblocks, offsets, perm, finalShard,
                       corr);
    return hipGetLastError();
}

}  // namespace ssa
