// Keyed GROUP BY: the scan VM feeding a hash aggregation sink, and the kernels that turn the hash table
// into output columns.
//
// The reference aggregates through a ClickHouse-derived hash aggregator, one batch at a time
// (ydb/library/arrow_clickhouse/Aggregator.cpp:132-226 picks the hash method from the key columns,
// Aggregator.cpp:228-330 executes a batch).  Here one fused pass evaluates the step's expressions and
// feeds every passing row into a two-level table:
//
//   * level 1 lives in the workgroup's LDS and absorbs low-cardinality groupings entirely: each slot
//     carries R replicas of every accumulator so that the lanes of a wave that hit the same group
//     spread over R addresses instead of serialising on one;
//   * level 2 is an open-addressing table in HBM.  Rows whose key does not fit level 1 go there
//     directly; level 1 is merged into it once, when the workgroup has consumed its last tile.
//
// Keys are 128 bits (lo, hi).  Fixed-width keys of up to 119 bits in total are packed exactly (value
// bits and, for the null-aware methods, one null flag per key), so the table compares keys, not
// hashes, and the output key columns are decoded from the table.  Anything wider -- and every string
// key of more than 8 bytes -- is reduced to a 126-bit hash that finds the slot; the slot is the row's group
// only if the key columns of its representative row equal the row's (SameKeyRows), so the grouping is
// exact.  Those tables keep the smallest row index of each group and the output keys are gathered from
// that row of the input.
//
// HBM traffic of the scan is the referenced source columns once plus the table, which is tiny for the
// groupings this path is for; the bound is HBM bandwidth.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "kernels.h"
#include "scan_vm.h"

namespace ssa {

namespace {

constexpr uint64_t kEmpty = ~0ull;        // lo word of a free slot (real keys keep bit 63 of lo clear)
constexpr uint64_t kUnpublished = ~0ull;  // hi word of a slot whose owner has not stored it yet

__device__ __forceinline__ uint64_t Mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33;
    x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33;
    return x;
}

// Table position of a key.  Only the spread matters (equality is decided on the key words), so 32 bits of
// multiply-xorshift are enough; both table levels take their index from it.
__device__ __forceinline__ uint32_t Fold32(uint64_t lo, uint64_t hi) {
    uint32_t x = uint32_t(lo) ^ (uint32_t(lo >> 32) * 0x9E3779B1u) ^ (uint32_t(hi) * 0x85EBCA6Bu) ^ (uint32_t(hi >> 32) * 0xC2B2AE35u);
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// Position of a key in the table in HBM (40 bits; the table's mask cuts it down), and with it the region of the table a
// deferred update of the key belongs to.  One 64-bit multiply (Fibonacci hashing: the top bits of the product depend on
// every bit of the key): this is computed for every record in every pass of the sort and again in the region pass.
// The key words are those the tables hold (KeyBuilder::Finish); for a 64-bit key carried as one raw word the first line
// is that word again, so that TablePosRaw(raw) is the same position.
__device__ __forceinline__ uint64_t TablePosRaw(uint64_t raw) { return (raw * 0x9E3779B97F4A7C15ull) >> 24; }
__device__ __forceinline__ uint64_t TablePos(uint64_t lo, uint64_t hi) {
    return TablePosRaw((lo | (hi << 63)) ^ ((hi >> 1) * 0xC2B2AE3D27D4EB4Full));
}

// Order-preserving map double -> uint64 (so float min/max become integer atomics) and back.
__device__ __forceinline__ uint64_t OrderedFromF64(uint64_t bits) {
    return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
}
__device__ __forceinline__ uint64_t F64FromOrdered(uint64_t enc) {
    return (enc >> 63) ? (enc & 0x7FFFFFFFFFFFFFFFull) : ~enc;
}

// What the tables hold for an untouched accumulator.
__device__ __host__ inline uint64_t TableIdentity(int kind) {
    switch (kind) {
        case AGG_MIN_I: return uint64_t(INT64_MAX);
        case AGG_MAX_I: return uint64_t(INT64_MIN);
        case AGG_MIN_U: return ~0ull;
        case AGG_MIN_F: return 0xFFF0000000000000ull;  // ordered(+inf)
        case AGG_MAX_F: return 0x000FFFFFFFFFFFFFull;  // ordered(-inf)
        case AGG_MIN_S: case AGG_MAX_S: return ~0ull;  // no row
        default: return 0;
    }
}

// MIN / MAX over a string column: the accumulator holds the row of the smallest / largest string seen, replaced by a
// compare-and-swap whenever `row`'s string beats it (bytewise, then by length: std::string_view's order; equal strings
// keep the smaller row).  The strings are read from the column in memory.  Works on LDS and HBM accumulators alike.
__device__ __noinline__ void StringBestUpdate(const DevColumn& col, bool isMin, uint64_t* p, uint64_t row) {
    const int32_t* offs = static_cast<const int32_t*>(col.values);
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(p);
    const int32_t b1 = offs[row], n1 = offs[row + 1] - b1;
    unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (cur == row) return;
        if (cur != ~0ull) {
            const int32_t b2 = offs[cur], n2 = offs[cur + 1] - b2;
            const int c = StrCompare(col.chars + b1, n1, col.chars + b2, n2);
            const bool better = c == 0 ? row < cur : (isMin ? c < 0 : c > 0);
            if (!better) return;
        }
        const unsigned long long prev = atomicCAS(slot, cur, static_cast<unsigned long long>(row));
        if (prev == cur) return;
        cur = prev;
    }
}

// Accumulator update.  The same code serves LDS and HBM accumulators: after inlining the compiler
// knows the address space and picks ds_* or global_* atomics.
__device__ __forceinline__ void AccUpdate(int kind, uint64_t* p, uint64_t v) {
    switch (kind) {
        case AGG_COUNT:
        case AGG_SUM_I:
            atomicAdd(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
            break;
        case AGG_SUM_F:
            atomicAdd(reinterpret_cast<double*>(p), BitsToF64(v));
            break;
        case AGG_MIN_I:
            atomicMin(reinterpret_cast<long long*>(p), static_cast<long long>(v));
            break;
        case AGG_MAX_I:
            atomicMax(reinterpret_cast<long long*>(p), static_cast<long long>(v));
            break;
        case AGG_MIN_U:
        case AGG_MIN_F:
            atomicMin(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
            break;
        case AGG_MAX_U:
        case AGG_MAX_F:
            atomicMax(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
            break;
        default:
            break;
    }
}

// Merge of two table states of one aggregate (level-1 replicas; level 1 into level 2 goes through
// AccUpdate because other workgroups update level 2 concurrently).
__device__ __forceinline__ uint64_t TableCombine(int kind, uint64_t a, uint64_t b) {
    switch (kind) {
        case AGG_COUNT: case AGG_SUM_I: return a + b;
        case AGG_SUM_F: return F64ToBits(BitsToF64(a) + BitsToF64(b));
        case AGG_MIN_I: return int64_t(b) < int64_t(a) ? b : a;
        case AGG_MAX_I: return int64_t(b) > int64_t(a) ? b : a;
        case AGG_MIN_U: case AGG_MIN_F: return b < a ? b : a;
        case AGG_MAX_U: case AGG_MAX_F: return b > a ? b : a;
        default: return a;
    }
}

// Find or claim the slot of key (lo, hi) in an open-addressing table of `mask + 1` slots, starting at
// `start`.  A slot is claimed by a compare-and-swap on its lo word; the owner then publishes hi.  A
// lane that finds its own lo with hi not yet published looks at the same slot again on the next trip
// of the loop: the lanes of a wave take every trip together, so the owner's store has been issued by
// then, and owners in other waves make progress on their own.  Returns -1 after `maxProbes` slots.
// The probe sequence stays inside the region of its first slot: the slots that agree with it above `regionMask`
// (the table's mask: the whole table is one region).
// `abortWord`: a status word other lanes raise SCAN_TABLE_FULL in (nullptr: none); looked at every 64 slots.
__device__ __forceinline__ int64_t TableUpsert(uint64_t* tLo, uint64_t* tHi, uint64_t mask, uint64_t regionMask, uint64_t start,
                                               uint64_t lo, uint64_t hi, bool twoWords, uint32_t maxProbes, const uint32_t* abortWord = nullptr) {
    uint64_t s = start & mask;
    uint32_t probes = 0;
    while (probes < maxProbes) {
        if (abortWord != nullptr && (probes & 63u) == 63u &&
            (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SCAN_TABLE_FULL)) {
            return -1;
        }
        uint64_t cur = __hip_atomic_load(tLo + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmpty) {
            cur = atomicCAS(reinterpret_cast<unsigned long long*>(tLo + s), static_cast<unsigned long long>(kEmpty),
                            static_cast<unsigned long long>(lo));
            if (cur == kEmpty) {
                if (twoWords) __hip_atomic_store(tHi + s, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return int64_t(s);
            }
        }
        if (cur == lo) {
            if (!twoWords) return int64_t(s);
            const uint64_t ch = __hip_atomic_load(tHi + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ch == hi) return int64_t(s);
            if (ch == kUnpublished) continue;
        }
        s = (s & ~regionMask) | ((s + 1) & regionMask);
        ++probes;
    }
    return -1;
}

// Grouping keys of rows r1 and r2 compared value by value, from the key columns in memory.  Used for the rows whose
// table key is a hash (a string of more than 8 bytes, more than 119 key bits): the hash finds a slot, this decides
// whether the slot's group is the row's -- equal hashes of different keys end up in different slots.  Such steps take
// all their keys from input columns.  A string whose offsets do not describe the chars buffer counts as empty.
// kWords: 8-byte words of both strings that are loaded together (a load, a compare and a branch per word is a round trip
// to memory per 8 bytes): 4 in the region passes, which wait for these loads and little else; 1 in the scan, whose
// registers are better spent on rows in flight.  The last bytes of a string are compared as masked words where both
// buffers have the bytes to read, not byte by byte.
template <int kWords = 1>
__device__ __noinline__ bool SameKeyRows(const DevColumn* cols, const KeyedSink& s, uint64_t r1, uint64_t r2) {
    if (r1 == r2) return true;
    for (int k = 0; k < s.nkeys; ++k) {
        const KeySpec& ks = s.keys[k];
        const DevColumn& col = cols[ks.col >= 0 ? ks.col : -2 - ks.reg];
        if (ks.col >= 0) {
            const int32_t* offs = static_cast<const int32_t*>(col.values);
            int32_t b1 = offs[r1], n1 = offs[r1 + 1] - b1, b2 = offs[r2], n2 = offs[r2 + 1] - b2;
            if (b1 < 0 || n1 < 0 || int64_t(b1) + n1 > int64_t(col.chars_bytes)) n1 = 0;
            if (b2 < 0 || n2 < 0 || int64_t(b2) + n2 > int64_t(col.chars_bytes)) n2 = 0;
            if (n1 != n2) return false;
            const uint8_t* p = col.chars + b1;
            const uint8_t* q = col.chars + b2;
            constexpr int32_t kStep = 8 * kWords;
            int32_t i = 0;
            for (; i + kStep <= n1; i += kStep) {
                uint64_t x[kWords], y[kWords], diff = 0;
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    __builtin_memcpy(&x[w], p + i + 8 * w, 8);
                    __builtin_memcpy(&y[w], q + i + 8 * w, 8);
                }
#pragma unroll
                for (int w = 0; w < kWords; ++w) diff |= x[w] ^ y[w];
                if (diff != 0) return false;
            }
            if (i == n1) continue;
            // the last 1 .. kStep - 1 bytes: the same loads where both buffers have kStep readable bytes from here on (no
            // sectors fetched for words behind the strings' ends), the bytes past the ends masked away
            if (int64_t(b1) + i + kStep <= int64_t(col.chars_bytes) && int64_t(b2) + i + kStep <= int64_t(col.chars_bytes)) {
                const int32_t rest = n1 - i;
                uint64_t x[kWords], y[kWords], diff = 0;
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    x[w] = y[w] = 0;
                    if (8 * w < rest) {
                        __builtin_memcpy(&x[w], p + i + 8 * w, 8);
                        __builtin_memcpy(&y[w], q + i + 8 * w, 8);
                    }
                }
#pragma unroll
                for (int w = 0; w < kWords; ++w) {
                    const int32_t m = rest - 8 * w;  // bytes of this word that belong to the strings
                    const uint64_t keep = m >= 8 ? ~0ull : (m <= 0 ? 0ull : (1ull << (8 * m)) - 1ull);
                    diff |= (x[w] ^ y[w]) & keep;
                }
                if (diff != 0) return false;
                continue;
            }
            for (; i + 8 <= n1; i += 8) {
                uint64_t x, y;
                __builtin_memcpy(&x, p + i, 8);
                __builtin_memcpy(&y, q + i, 8);
                if (x != y) return false;
            }
            for (; i < n1; ++i) {
                if (p[i] != q[i]) return false;
            }
        } else {
            if (ks.nullable && col.validity != nullptr) {
                const uint32_t v1 = LoadBit(col.validity, int64_t(r1) + col.valid_bit_off);
                const uint32_t v2 = LoadBit(col.validity, int64_t(r2) + col.valid_bit_off);
                if (v1 != v2) return false;
                if (!v1) continue;
            }
            const int w = TypeWidth(col.type);
            if (LoadRaw(col.values, w, int64_t(r1)) != LoadRaw(col.values, w, int64_t(r2))) return false;
        }
    }
    return true;
}

// TableUpsert for a row whose key is a hash: a slot with the row's hash is the row's group only if the keys of the
// slot's representative row equal the row's (`same(rep)`); otherwise the probe goes on, so two keys with one hash
// take two slots.  The owner of a slot stores its row as the representative before it publishes hi; a reader that
// still sees no representative looks again on its next trip.  Keeps the smallest row of the group as representative.
template <class Same>
__device__ __forceinline__ int64_t TableUpsertVerified(uint64_t* tLo, uint64_t* tHi, uint64_t* tRep, uint64_t mask, uint64_t regionMask,
                                                       uint64_t start, uint64_t lo, uint64_t hi, uint64_t row, uint32_t maxProbes, Same same,
                                                       const uint32_t* abortWord = nullptr) {
    uint64_t s = start & mask;
    uint32_t probes = 0;
    while (probes < maxProbes) {
        if (abortWord != nullptr && (probes & 63u) == 63u &&
            (__hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SCAN_TABLE_FULL)) {
            return -1;
        }
        uint64_t cur = __hip_atomic_load(tLo + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmpty) {
            cur = atomicCAS(reinterpret_cast<unsigned long long*>(tLo + s), static_cast<unsigned long long>(kEmpty),
                            static_cast<unsigned long long>(lo));
            if (cur == kEmpty) {
                __hip_atomic_store(tRep + s, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(tHi + s, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return int64_t(s);
            }
        }
        if (cur == lo) {
            const uint64_t ch = __hip_atomic_load(tHi + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ch == kUnpublished) continue;
            if (ch == hi) {
                const uint64_t rep = __hip_atomic_load(tRep + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rep == ~0ull) continue;  // the owner's stores arrive in any order
                if (same(rep)) {
                    if (row < rep) atomicMin(reinterpret_cast<unsigned long long*>(tRep + s), static_cast<unsigned long long>(row));
                    return int64_t(s);
                }
            }
        }
        s = (s & ~regionMask) | ((s + 1) & regionMask);
        ++probes;
    }
    return -1;
}

// Hash of a string key: its length, then its bytes in 8-byte little-endian words.  Takes and returns
// the two hash lanes by value so that the caller's key state stays in registers.
struct HashPair {
    uint64_t lo, hi;
};
__device__ __forceinline__ void AbsorbWord(HashPair& h, uint64_t w) {
    h.lo = (h.lo ^ w) * 0x9E3779B97F4A7C15ull;
    h.lo ^= h.lo >> 29;
    h.hi = (h.hi + w) * 0xC2B2AE3D27D4EB4Full;
    h.hi ^= h.hi >> 31;
}
// `avail`: bytes of the column's chars buffer from `p` on (8-byte loads stay inside it; a string whose offsets point
// outside the buffer hashes as empty and is never dereferenced).
__device__ __noinline__ HashPair AbsorbString(HashPair h, const uint8_t* p, int32_t n, int64_t avail) {
    if (n < 0 || int64_t(n) > avail) n = 0;
    AbsorbWord(h, uint64_t(uint32_t(n)));
    for (int32_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        const int32_t m = (n - i) < 8 ? (n - i) : 8;
        if (int64_t(i) + 8 <= avail) {
            __builtin_memcpy(&w, p + i, 8);  // unaligned 8-byte load
            if (m < 8) w &= (1ull << (8 * m)) - 1ull;
        } else {
            for (int32_t j = 0; j < m; ++j) w |= uint64_t(p[i + j]) << (8 * j);
        }
        AbsorbWord(h, w);
    }
    return h;
}

// ---- key construction ---------------------------------------------------------------------------------
template <int U>
struct KeyBuilder {
    uint64_t lo[U], hi[U];
    int pos;  // exact packing: bits used so far (wave-uniform)

    __device__ __forceinline__ void Begin(bool hashed) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            lo[u] = hashed ? 0x243F6A8885A308D3ull : 0ull;
            hi[u] = hashed ? 0x13198A2E03707344ull : 0ull;
        }
        pos = 0;
    }
    // exact: append the low `nbits` of `bits`
    __device__ __forceinline__ void Append(int u, uint64_t bits, int nbits) {
        (void)nbits;
        if (pos < 64) {
            lo[u] |= bits << pos;
            if (pos > 0 && pos + nbits > 64) hi[u] |= bits >> (64 - pos);
        } else {
            hi[u] |= bits << (pos - 64);
        }
    }
    // exact packing at a per-row position (rows of a batch with string keys differ in length)
    // (`bits` has nothing above its low `nbits` bits; written without branches: `at` differs from lane to lane)
    __device__ __forceinline__ void AppendAt(int u, uint64_t bits, uint32_t nbits, uint32_t& at) {
        const uint32_t sh = at & 63u;
        const uint64_t low = bits << sh;
        const uint64_t carry = sh ? bits >> (64u - sh) : 0ull;
        lo[u] |= at < 64 ? low : 0ull;
        hi[u] |= at < 64 ? carry : (at < 128 ? low : 0ull);
        at += nbits;
    }
    // hashed: absorb one 64-bit word into both lanes of the hash
    __device__ __forceinline__ void Absorb(int u, uint64_t w) {
        HashPair h{lo[u], hi[u]};
        AbsorbWord(h, w);
        lo[u] = h.lo;
        hi[u] = h.hi;
    }
    // Final table key: bit 63 of lo stays clear (so lo never equals the free marker) and hi never
    // equals the unpublished marker: exact keys keep its top byte zero, hashed keys force 0b10 on top.
    __device__ __forceinline__ void Finish(int u, bool hashed) {
        if (hashed) {
            lo[u] = Mix64(lo[u]) & 0x7FFFFFFFFFFFFFFFull;
            hi[u] = (Mix64(hi[u]) | 0x8000000000000000ull) & ~0x4000000000000000ull;
        } else {
            const uint64_t l = lo[u], h = hi[u];
            lo[u] = l & 0x7FFFFFFFFFFFFFFFull;
            hi[u] = (h << 1) | (l >> 63);
        }
    }
};

__device__ __forceinline__ uint64_t WidthMask(int nbits) { return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull); }

}  // namespace

// ---- sink: keyed aggregation ------------------------------------------------------------------------
namespace {

constexpr uint32_t kNoDense = 0xFFFFFFFFu;  // dense id of a level-1 slot not yet published by its owner
constexpr int kHotGroups = 4;               // groups whose SUM / COUNT accumulators live in registers

// Level-1 upsert.  Hash positions are claimed like level 2; the owner then numbers the group with the next dense id
// (order of first appearance in this workgroup), records a representative row and publishes the id.  Accumulators are
// indexed by dense id, so the first kHotGroups groups of a workgroup are the ones kept in registers.
// `verify`: the key is a hash -- a slot with it is the row's group only if `same(representative row)`.
template <class Same>
__device__ __forceinline__ int LdsUpsertDense(uint64_t* lLo, uint64_t* lHi, uint64_t* lRep, uint32_t* lDense, uint32_t* lCount,
                                              uint64_t* lHotKey, uint32_t mask, uint64_t start, uint64_t lo, uint64_t hi, bool twoWords,
                                              uint64_t row, uint32_t maxProbes, bool verify, Same sameRows) {
    uint32_t s = uint32_t(start) & mask;
    uint32_t probes = 0;
    while (probes < maxProbes) {
        uint64_t cur = __hip_atomic_load(lLo + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == kEmpty) {
            cur = atomicCAS(reinterpret_cast<unsigned long long*>(lLo + s), static_cast<unsigned long long>(kEmpty),
                            static_cast<unsigned long long>(lo));
            if (cur == kEmpty) {
                const uint32_t id = atomicAdd(lCount, 1u);
                lRep[s] = row;
                if (twoWords) __hip_atomic_store(lHi + s, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (id < uint32_t(kHotGroups)) {  // key of a register-held group: waves pick it up at their next tile
                    lHotKey[2 * id + 1] = hi;
                    __hip_atomic_store(lHotKey + 2 * id, lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __hip_atomic_store(lDense + s, id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                return int(id);
            }
        }
        if (cur == lo) {
            bool same = true;
            if (twoWords) {
                const uint64_t ch = __hip_atomic_load(lHi + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (ch == kUnpublished) continue;  // owner is between its two stores: look again next trip
                same = ch == hi;
            }
            if (same) {
                const uint32_t id = __hip_atomic_load(lDense + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (id == kNoDense) continue;
                if (!verify || sameRows(lRep[s])) return int(id);
            }
        }
        s = (s + 1) & mask;
        ++probes;
    }
    return -1;
}

}  // namespace

// kHashed: the batch has string keys or more than 119 key bits (per-row packing with a hash for rows that do not fit);
// otherwise all keys pack exactly at wave-uniform positions.
#ifdef SSA_PHASE_TIMING
#define SSA_T(i) do { const uint64_t now_ = __builtin_readcyclecounter(); tphase[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define SSA_T(i) do {} while (0)
#endif

// `a` and `rs` are the kernel arguments of this launch (addresses, row count, table size); `s` describes the sink and
// is the constant image in a kernel built for one step program.
template <int U, bool kHashed>
__device__ __forceinline__ void ScanKeyedBody() {
    const ScanArgs& a = KernArgs<KeyedSink>().a;
    const KeyedSink& rs = KernArgs<KeyedSink>().s;
#ifdef SSA_SPEC
    const KeyedSink& s = reinterpret_cast<const KArgs<KeyedSink>*>(kSpecImage)->s;
    constexpr bool kStatic = true;
#else
    const KeyedSink& s = rs;
    constexpr bool kStatic = false;
#endif
#define SSA_ROW(table, expr, i) (kStatic ? uint32_t(expr) : ReadLane32(table, i))
    // a launch over several slices: this workgroup's slice, its rows and its part of the level-2 table.  (A kernel built
    // for a step program is built for one kind of launch: in the one for single batches none of this is left.)
#if defined(SSA_SPEC) && !SSA_SPEC_SLICES
    const SliceDesc* slice = nullptr;
#else
    const SliceDesc* slice = a.slices != nullptr ? a.slices + __builtin_amdgcn_readfirstlane(a.block_slice[blockIdx.x]) : nullptr;
#endif
    // Deferred level-2 updates: a row that finds no place in level 1 leaves a record (key, operands) at its row index
    // instead of touching the table in HBM; the host then sorts the records by table region and applies them region by
    // region.  Only in kernels built for a step program that asked for it.
#if defined(SSA_SPEC) && defined(SSA_SPEC_DEFER)
    constexpr bool kDefer = SSA_SPEC_DEFER != 0;
#else
    constexpr bool kDefer = false;
#endif
#if defined(SSA_SPEC) && defined(SSA_SPEC_COMPACT)
    constexpr bool kCompact = SSA_SPEC_COMPACT != 0;
#else
    constexpr bool kCompact = false;
#endif
    constexpr int64_t kDeferredSlot = INT64_MIN;
    Machine<U> m(a, slice);
    const uint64_t tableOff = slice != nullptr ? Uniform64(slice->table_off) : 0ull;
    const int64_t firstTile = slice != nullptr ? int64_t(blockIdx.x) - __builtin_amdgcn_readfirstlane(slice->first_block) : int64_t(blockIdx.x);
    const int64_t tileStride = slice != nullptr ? int64_t(__builtin_amdgcn_readfirstlane(slice->nblocks)) : int64_t(gridDim.x);
    const int naggs = s.naggs;
    const int L = s.lds_cap, R = s.lds_rep;
    const bool twoWords = kHashed || s.two_words != 0;
    constexpr bool hashed = kHashed;
    uint64_t* const tKeyLo = rs.t_key_lo + tableOff;
    uint64_t* const tKeyHi = twoWords ? rs.t_key_hi + tableOff : rs.t_key_hi;
    uint64_t* const tRepRow = hashed ? rs.t_rep_row + tableOff : rs.t_rep_row;
    uint32_t* const tSeen = rs.t_seen + tableOff;
    uint64_t* const tAcc = rs.t_acc + tableOff;
    const uint64_t accStride = rs.acc_stride;
    const uint64_t tableMask = rs.table_mask;
    const uint64_t regionMask = rs.region_mask;
    // level-1 table in the sink area of LDS:
    //   by hash position: lo[L], hi[L], rep[L], dense[L];  by dense id: acc[naggs][L][R], seen[L];  count;
    //   keys of the register-held groups: hot[kHotGroups][2]
    uint64_t* lLo = reinterpret_cast<uint64_t*>(gLds + a.lds_sink_off);
    uint64_t* lHi = lLo + L;
    uint64_t* lRep = lHi + L;
    uint64_t* lHotKey = lRep + L;
    uint64_t* lAcc = lHotKey + 2 * kHotGroups;
    uint32_t* lSeen = reinterpret_cast<uint32_t*>(lAcc + size_t(naggs) * L * R);
    uint32_t* lDense = lSeen + L;
    uint32_t* lCount = lDense + L;
    for (int i = threadIdx.x; i < L; i += kBlock) {
        lLo[i] = kEmpty;
        lHi[i] = kUnpublished;
        lRep[i] = ~0ull;
        lSeen[i] = 0;
        lDense[i] = kNoDense;
    }
    if (threadIdx.x == 0) *lCount = 0;
    if (threadIdx.x < 2 * kHotGroups) lHotKey[threadIdx.x] = kEmpty;
    for (int k = 0; k < naggs; ++k) {
        const uint64_t ident = TableIdentity(s.aggs[k].kind);
        for (int i = threadIdx.x; i < L * R; i += kBlock) lAcc[size_t(k) * L * R + i] = ident;
    }
    __syncthreads();  // the table is shared by the workgroup's waves
    const int rep = threadIdx.x & (R - 1);
    const uint32_t ldsProbes = L < 8 ? uint32_t(L) : 8u;

    // Register accumulators of the first kHotGroups groups x first kHotAggs aggregates (SUM / COUNT only; their
    // identity is 0 in every representation).  Lane-private for the whole kernel, folded into level 1 at the end.
    uint64_t hot[kHotGroups][kHotAggs];
#pragma unroll
    for (int g = 0; g < kHotGroups; ++g) {
#pragma unroll
        for (int k = 0; k < kHotAggs; ++k) hot[g][k] = 0;
    }
    // (The generic kernel for hashed keys goes without them: it has no registers left, and a spilled lane table is not
    // restored in the lanes that are inactive at the reload -- the v_readlane decode below would read stale values.)
    constexpr bool kHotPossible = kStatic || !kHashed;
    const bool useHot = kHotPossible && L >= kHotGroups && s.use_hot != 0 && !(m.S().debug_flags & 8);
    // keys of the register-held groups as this wave knows them (a free marker never equals a key)
    uint64_t hotLo[kHotGroups], hotHi[kHotGroups];
#pragma unroll
    for (int g = 0; g < kHotGroups; ++g) hotLo[g] = kEmpty, hotHi[g] = 0;
    int hotKnown = 0;

    // Sink description in VGPR lanes (lane i: key i / accumulator i), decoded with v_readlane inside the tile loop:
    // re-reading the kernel-argument segment there costs a scalar-memory round trip per field and tile.
    const int lane = threadIdx.x & 63;
    const int ki = lane < kMaxKeys ? lane : 0;
    const uint32_t tKeyCol = uint32_t(s.keys[ki].col), tKeyType = uint32_t(s.keys[ki].type), tKeyNullable = uint32_t(s.keys[ki].nullable);
    const uint32_t tKeyFk = s.keys[ki].fetch_kind, tKeyPayload = s.keys[ki].fetch_payload;
    const int ai = lane < kMaxAggs ? lane : 0;
    const uint32_t tAggKind = uint32_t(s.aggs[ai].kind), tAggReg = uint32_t(s.aggs[ai].reg), tAggCheck = uint32_t(s.aggs[ai].check_valid);
    const uint32_t tAggFk = s.aggs[ai].fetch_kind, tAggPayload = s.aggs[ai].fetch_payload, tAggFlags = uint32_t(s.aggs[ai].flags),
                   tAggStrCol = uint32_t(s.aggs[ai].str_col);
    const int hi_ = lane < kHotAggs ? lane : 0;
    const uint32_t tHotKind = (lane < kHotAggs && !(m.S().debug_flags & 32)) ? uint32_t(s.hot_kind[hi_]) : 0u;
    const uint32_t tHotOff = uint32_t(s.hot_off[hi_]);
    const int nkeys = s.nkeys;

#ifdef SSA_PHASE_TIMING
    uint64_t tphase[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = __builtin_readcyclecounter();
#endif
    const int64_t ntiles = (m.nrows + kBlock * U - 1) / (kBlock * U);
    const uint32_t* const abortWord = rs.abort_check != 0 ? a.status : nullptr;  // (wave-uniform)
    uint32_t l1Tried = 0, l1Found = 0;  // rows of this wave looked up in level 1 / found there, since the last verdict
    bool l1Bypass = false;
    for (int64_t tile = firstTile; tile < ntiles; tile += tileStride) {
        SSA_T(7);
        // (the load is issued ahead of the tile's own and has arrived when they have)
        const uint32_t stopNow = abortWord != nullptr ? __hip_atomic_load(abortWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        m.BeginTile(tile * (kBlock * U), (tile + tileStride) * (kBlock * U));
        SSA_T(0);
        m.Run();
        SSA_T(1);
        if (__builtin_amdgcn_readfirstlane(stopNow) & SCAN_TABLE_FULL) break;  // the step is being repeated with a larger table
        if constexpr (kCompact) {
            // Behind a selective filter most lanes have no row that passed and the rest one or two: the sink below would
            // still build the keys of all U rows of every lane.  This one takes the rows one per lane and trip, and
            // reads what it needs of a row -- key columns, arguments -- from the staged tiles at the row's index.  Only
            // steps whose keys pack exactly and whose operands are staged columns without validity bitmaps (or
            // constants) are built this way (KeyedSink.compact_n); the semantics are those of the sink below.
            auto fetchAt = [&](uint32_t fk, uint32_t payload, int idx) -> uint64_t {
                const unsigned char* base = gLds + (payload & 0xFFFFFFu);
                switch (fk & 0x7Fu) {
                    case FK_LDS_64: return reinterpret_cast<const uint64_t*>(base)[idx];
                    case FK_LDS_I32: return uint64_t(int64_t(reinterpret_cast<const int32_t*>(base)[idx]));
                    case FK_LDS_U32: return reinterpret_cast<const uint32_t*>(base)[idx];
                    case FK_LDS_I16: return uint64_t(int64_t(reinterpret_cast<const int16_t*>(base)[idx]));
                    case FK_LDS_U16: return reinterpret_cast<const uint16_t*>(base)[idx];
                    case FK_LDS_I8: return uint64_t(int64_t(reinterpret_cast<const int8_t*>(base)[idx]));
                    case FK_LDS_U8: return reinterpret_cast<const uint8_t*>(base)[idx];
                    case FK_LDS_F32: return F64ToBits(double(reinterpret_cast<const float*>(base)[idx]));
                    case FK_CONST: return m.ConstBits(int(payload & 0xFFFFFFu));
                    default: return 0;
                }
            };
            // The rows that passed are dealt out evenly: they are numbered lane after lane (a lane's rows in order), and
            // lane l takes the rows l, l + 64, ...  It finds the lane that owns a row by a binary search over the running
            // counts (six shuffles) -- thirty rows of a thousand are one trip of the loop instead of three.
            const uint32_t mine = uint32_t(__popc(m.P));
            // rows that passed in this lane and the lanes before it: a running sum across the wave by row shifts and row
            // broadcasts (no round trips through the LDS crossbar)
            uint32_t upTo = mine;
            upTo += uint32_t(__builtin_amdgcn_update_dpp(0, int(upTo), 0x111, 0xf, 0xf, false));  // row_shr:1
            upTo += uint32_t(__builtin_amdgcn_update_dpp(0, int(upTo), 0x112, 0xf, 0xf, false));  // row_shr:2
            upTo += uint32_t(__builtin_amdgcn_update_dpp(0, int(upTo), 0x114, 0xf, 0xf, false));  // row_shr:4
            upTo += uint32_t(__builtin_amdgcn_update_dpp(0, int(upTo), 0x118, 0xf, 0xf, false));  // row_shr:8
            upTo += uint32_t(__builtin_amdgcn_update_dpp(0, int(upTo), 0x142, 0xa, 0xf, false));  // row_bcast:15
            upTo += uint32_t(__builtin_amdgcn_update_dpp(0, int(upTo), 0x143, 0xc, 0xf, false));  // row_bcast:31
            const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(upTo), 63));
            // With room in LDS every lane lists its rows at their numbers (row slot and lane in two bytes), and the lane that
            // takes a row reads which one it is: one write per row that passed and one read per row taken.  (LDS accesses
            // of a wave complete in order: the reads below see the list.)
            uint16_t* const slots = s.compact_slots != 0
                ? reinterpret_cast<uint16_t*>(gLds + a.lds_sink_off + s.compact_slots) + (threadIdx.x >> 6) * (64 * U) : nullptr;
            if (slots != nullptr) {
                uint32_t left = m.P, at = upTo - mine;
                while (left != 0) {
                    slots[at++] = uint16_t((__ffs(int(left)) - 1) * 64 + lane);
                    left &= left - 1u;
                }
            }
            for (uint32_t first = 0; first < total; first += 64) {
                const uint32_t want = first + uint32_t(lane);
                const bool has = want < total;
                int owner = 0;  // the first lane whose running count exceeds `want`
                int pu = 0;
                if (slots != nullptr) {
                    const uint32_t v = has ? slots[want] : 0u;
                    owner = int(v & 63u);
                    pu = int(v >> 6);
                } else {
                    // without it: a binary search over the running counts (six shuffles), then the owner's rows counted off
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1) {
                        if (uint32_t(__shfl(int(upTo), owner + step - 1, 64)) <= want) owner += step;
                    }
                    owner = owner < 64 ? owner : 63;
                    uint32_t ownerRows = uint32_t(__shfl(int(m.P), owner, 64));
                    uint32_t skip = want - uint32_t(__shfl(int(upTo - mine), owner, 64));  // the owner's rows before this one
                    if (has) {
                        while (skip-- != 0) ownerRows &= ownerRows - 1u;
                        pu = __ffs(int(ownerRows)) - 1;
                    }
                }
                if (has) {
                const int idx = m.waveBase + pu * 64 + owner;
                uint64_t klo = 0, khi = 0;
                int pos = 0;
#pragma unroll
                for (int k = 0; k < kMaxKeys; ++k) {
                    if (k >= nkeys) break;
                    const KeySpec& ks = s.keys[k];
                    const int nbits = 8 * TypeWidth(ks.type);
                    const uint64_t bits = PayloadToStored(fetchAt(ks.fetch_kind, ks.fetch_payload, idx), ks.type) & WidthMask(nbits);
                    if (pos < 64) {
                        klo |= bits << pos;
                        if (pos > 0 && pos + nbits > 64) khi |= bits >> (64 - pos);
                    } else {
                        khi |= bits << (pos - 64);
                    }
                    pos += nbits + (ks.nullable ? 1 : 0);  // (the null flag of a key that is never null stays 0)
                }
                {
                    const uint64_t l = klo, h = khi;
                    klo = l & 0x7FFFFFFFFFFFFFFFull;
                    khi = (h << 1) | (l >> 63);
                }
                const uint32_t h32 = Fold32(klo, khi);
                int64_t sl = -1;
                if (L > 0) {
                    sl = LdsUpsertDense(lLo, lHi, lRep, lDense, lCount, lHotKey, uint32_t(L - 1), h32, klo, khi, twoWords,
                                        uint64_t(m.Row(0) + (owner - lane) + int64_t(pu) * 64), ldsProbes, false, [](uint64_t) { return true; });
                }
                bool placed = true;
                if (sl < 0) {
                    const int64_t g = TableUpsert(tKeyLo, tKeyHi, tableMask, regionMask, TablePos(klo, khi), klo, khi, twoWords, 4096u, abortWord);
                    if (g < 0) {
                        if (abortWord != nullptr && !(m.err & SCAN_TABLE_FULL)) atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
                        m.err |= SCAN_TABLE_FULL;
                        placed = false;
                    }
                    sl = -2 - g;
                }
#pragma unroll
                for (int k = 0; k < kMaxAggs; ++k) {
                    if (k >= naggs || !placed) break;
                    const AggSpec& spec = s.aggs[k];
                    const int kind = spec.kind;
                    uint64_t x = kind == AGG_COUNT ? 1ull : fetchAt(spec.fetch_kind, spec.fetch_payload, idx);
                    if (spec.check_valid) {  // (the argument has no nulls: every row marks the accumulator as seen)
                        if (sl >= 0) {
                            if (!((lSeen[sl] >> k) & 1u)) atomicOr(lSeen + sl, 1u << k);
                        } else {
                            uint32_t* g = tSeen + (-2 - sl);
                            if (!((__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> k) & 1u)) atomicOr(g, 1u << k);
                        }
                    }
                    if (kind == AGG_MIN_F || kind == AGG_MAX_F) {
                        const double d = BitsToF64(x);
                        if (d != d) {  // fmin / fmax semantics: a NaN never replaces a number
                            if (spec.flags & AGG_NAN_UNSUPPORTED) m.err |= SCAN_NAN_ORDER;
                            continue;
                        }
                        x = OrderedFromF64(x);
                    }
                    if (sl >= 0) {  // (two calls: each knows its address space)
                        AccUpdate(kind, lAcc + size_t(k) * L * R + sl * R + rep, x);
                    } else {
                        AccUpdate(kind, tAcc + size_t(k) * accStride + (-2 - sl), x);
                    }
                }
                }
            }
            continue;
        }
        if (kDefer && s.rec_raw64) {
            // (the bitmap of the rows that leave records is written by the wave as a whole: no lane leaves the tile early)
            if (!__any(m.P != 0)) {
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (m.Row(u) < m.nrows) rs.rec_present[uint64_t(m.Row(u)) >> 6] = 0ull;
                    }
                }
                continue;
            }
        } else if (m.P == 0) {
            if (kDefer) {  // no row of the tile leaves a record
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (m.Row(u) < m.nrows) rs.rec_lo[uint64_t(m.Row(u))] = kEmpty;
                }
            }
            continue;
        }

        // Keys of the lane's rows.  Batches whose keys all have a fixed width of at most 119 bits pack them at
        // wave-uniform positions.  Otherwise every row first tries the same exact packing at its own positions, a
        // string of up to 8 bytes as (length byte, bytes); rows that do not fit in 119 bits -- and only those -- are
        // reduced to a 126-bit hash.  Exact and hashed keys cannot meet: they differ in the top bits of `hi`.
        KeyBuilder<U> kb;
        kb.Begin(false);
        uint32_t at[U];
        uint32_t toHash = 0;  // bit u: row u needs the hash
#pragma unroll
        for (int u = 0; u < U; ++u) at[u] = 0;
        // While every key so far has the same width in all rows of the wave (fixed-width keys, string columns whose
        // strings of this slice all have one length) the packing positions are wave-uniform: kb.pos, shifts by a scalar.
        // The first key that differs from row to row moves the positions into at[] (same bits, per-row shifts).
        bool uniformPos = true;
        SSA_UNROLL_STATIC
        for (int k = 0; k < nkeys; ++k) {
            KeySpec ks;
            ks.col = int32_t(SSA_ROW(tKeyCol, s.keys[k].col, k));
            ks.type = int32_t(SSA_ROW(tKeyType, s.keys[k].type, k));
            ks.nullable = int32_t(SSA_ROW(tKeyNullable, s.keys[k].nullable, k));
            ks.fetch_kind = SSA_ROW(tKeyFk, s.keys[k].fetch_kind, k);
            ks.fetch_payload = SSA_ROW(tKeyPayload, s.keys[k].fetch_payload, k);
            if (kHashed && ks.col >= 0) {
                // string key; the validity of the column is not part of these keys.  The loads of the lane's rows are
                // issued together, then consumed.
                int32_t n[U];
                uint64_t w[U];
                m.WaitStringHeads();
                const int len = m.UniformStringKeyWords(ks.col, w);
                if (len > 0 && uniformPos) {
                    if (kb.pos + 8 + 8 * len <= 128) {
#pragma unroll
                        for (int u = 0; u < U; ++u) kb.Append(u, uint64_t(len), 8);
                        kb.pos += 8;
#pragma unroll
                        for (int u = 0; u < U; ++u) kb.Append(u, w[u], 8 * len);
                        kb.pos += 8 * len;
                    } else {
                        kb.pos = 128;  // does not fit: every row of the wave takes the hash
                    }
                    continue;
                }
                if (len > 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) n[u] = len;
                } else {
                    m.StringKeyWords(ks.col, m.P, n, w);
                }
                if (uniformPos) {
#pragma unroll
                    for (int u = 0; u < U; ++u) at[u] = uint32_t(kb.pos);
                    uniformPos = false;
                }
                // (no branches: the key words of rows that fail the predicate are never used, those of rows with a
                // longer string are replaced by the hash below)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    toHash |= (n[u] > 8 ? 1u : 0u) << u;
                    kb.AppendAt(u, uint64_t(uint32_t(n[u]) & 0xFFu), 8, at[u]);
                    kb.AppendAt(u, w[u], uint32_t(8 * n[u]), at[u]);
                }
                continue;
            }
            Val<U> x;
            m.FetchOperand(ks.fetch_kind, ks.fetch_payload, x);
            const int nbits = 8 * TypeWidth(ks.type);
            const uint64_t wmask = WidthMask(nbits);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool isNull = ks.nullable && !((x.ok >> u) & 1u);
                const uint64_t bits = isNull ? 0ull : (PayloadToStored(x.v[u], ks.type) & wmask);
                if constexpr (hashed) {
                    if (uniformPos) {
                        if (kb.pos + nbits <= 128) kb.Append(u, bits, nbits);
                    } else {
                        kb.AppendAt(u, bits, uint32_t(nbits), at[u]);
                        if (ks.nullable) kb.AppendAt(u, isNull ? 1ull : 0ull, 1, at[u]);
                    }
                } else {
                    kb.Append(u, bits, nbits);
                }
            }
            if constexpr (hashed) {
                if (uniformPos) {
                    kb.pos += nbits;
                    if (ks.nullable) {
                        if (kb.pos < 128) {
#pragma unroll
                            for (int u = 0; u < U; ++u) kb.Append(u, ((x.ok >> u) & 1u) ? 0ull : 1ull, 1);
                        }
                        kb.pos += 1;
                    }
                }
            }
            if constexpr (!hashed) {
                kb.pos += nbits;
                if (ks.nullable) {
#pragma unroll
                    for (int u = 0; u < U; ++u) kb.Append(u, ((x.ok >> u) & 1u) ? 0ull : 1ull, 1);
                    kb.pos += 1;
                }
            }
        }
        SSA_T(6);
        if constexpr (hashed) {
            if (uniformPos) {
                if (kb.pos > 119) toHash = AllMask<U>();
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (at[u] > 119) toHash |= 1u << u;
                }
            }
            toHash &= m.P;
            if (__any(toHash != 0)) {
                // the rare rows: long strings or more than 119 key bits
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!((toHash >> u) & 1u)) continue;
                    HashPair h{0x243F6A8885A308D3ull, 0x13198A2E03707344ull};
                    SSA_UNROLL_STATIC
                    for (int k = 0; k < nkeys; ++k) {
                        const int col = int32_t(SSA_ROW(tKeyCol, s.keys[k].col, k));
                        if (col >= 0) {
                            const uint8_t* p;
                            int32_t n;
                            m.StringAt(col, u, p, n);
                            h = AbsorbString(h, p, n, m.CharsAvailable(col, p));
                        } else {
                            Val<U> x;
                            m.FetchOperand(SSA_ROW(tKeyFk, s.keys[k].fetch_kind, k), SSA_ROW(tKeyPayload, s.keys[k].fetch_payload, k), x);
                            const int type = int32_t(SSA_ROW(tKeyType, s.keys[k].type, k));
                            const bool nullable = SSA_ROW(tKeyNullable, s.keys[k].nullable, k) != 0;
                            const bool isNull = nullable && !((x.ok >> u) & 1u);
                            AbsorbWord(h, isNull ? 0ull : (PayloadToStored(x.v[u], type) & WidthMask(8 * TypeWidth(type))));
                            if (nullable) AbsorbWord(h, isNull ? 1ull : 0ull);
                        }
                    }
                    kb.lo[u] = h.lo;
                    kb.hi[u] = h.hi;
                }
            }
        }

        SSA_T(2);
        // group of each row: >= 0 dense id in level 1, <= -2 level-2 slot g as -2 - g, -1 no row
        int64_t slot[U];
        // Groups held in registers are recognised by comparing with their keys: no hashing, no LDS round trip.
        if (useHot && hotKnown < kHotGroups) {
#pragma unroll
            for (int g = 0; g < kHotGroups; ++g) {
                if (g == hotKnown) {
                    const uint64_t lo = __hip_atomic_load(lHotKey + 2 * g, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__builtin_amdgcn_readfirstlane(uint32_t(lo >> 32)) != 0xFFFFFFFFu) {
                        hotLo[g] = lo;
                        hotHi[g] = lHotKey[2 * g + 1];
                        hotKnown = g + 1;
                    }
                }
            }
        }
        uint32_t pending = 0;  // rows whose group still has to be looked up in the tables
#pragma unroll
        for (int u = 0; u < U; ++u) {
            slot[u] = -1;
            if (!((m.P >> u) & 1u)) continue;
            const bool byHash = ((toHash >> u) & 1u) != 0;
            kb.Finish(u, byHash);
            if (byHash && (m.S().debug_flags & 64)) {  // tests: eight hash values for all such rows
                kb.lo[u] &= 7ull;
                kb.hi[u] = 0x8000000000000001ull;
            }
            int id = -1;
#pragma unroll
            for (int g = 0; g < kHotGroups; ++g) {
                if (kb.lo[u] == hotLo[g] && (!twoWords || kb.hi[u] == hotHi[g])) id = g;
            }
            // (a row whose key is a hash goes through the tables, where its key is compared with the group's)
            if (id >= 0 && useHot && !byHash) slot[u] = id; else pending |= 1u << u;
        }
        // Rows of other groups (and of the first groups until their keys are known) go through the tables.  Every trip
        // of the loop takes each lane's next pending row, whichever of its U rows that is: behind a selective filter
        // most lanes have none and the rest one, so the wave makes one or two trips instead of U.
        if (kDefer && l1Bypass) {
            // Level 1 holds few of this wave's rows any more (see below).  What it does hold are the heavy hitters that
            // came early, and those sit in the slot their hash points at: that one slot is read, nothing is claimed,
            // and every other row leaves a record.  (A key that is in the table elsewhere leaves records too: level 1
            // and the records meet in level 2.)
            uint64_t seenLo[U], seenHi[U];
            uint32_t seenId[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t at = L > 0 ? Fold32(kb.lo[u], kb.hi[u]) & uint32_t(L - 1) : 0u;
                seenLo[u] = L > 0 ? lLo[at] : kEmpty;
                seenHi[u] = (L > 0 && twoWords) ? lHi[at] : 0ull;
                seenId[u] = L > 0 ? lDense[at] : kNoDense;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!((pending >> u) & 1u)) continue;
                const bool byHash = hashed && (kb.hi[u] >> 63) != 0;  // (such a row would have to be compared with the group's)
                if (seenLo[u] == kb.lo[u] && (!twoWords || seenHi[u] == kb.hi[u]) && seenId[u] != kNoDense && !byHash) {
                    slot[u] = int64_t(seenId[u]);
                    continue;
                }
                if (twoWords && !s.rec_raw64) rs.rec_hi[uint64_t(m.Row(u))] = kb.hi[u];
                slot[u] = kDeferredSlot;
            }
            pending = 0;
        }
        while (__any(pending != 0)) {
            const bool has = pending != 0;
            const int pu = has ? __ffs(int(pending)) - 1 : 0;
            pending &= pending - 1u;
            uint64_t klo = kb.lo[0], khi = kb.hi[0];
#pragma unroll
            for (int u = 1; u < U; ++u) {
                klo = pu == u ? kb.lo[u] : klo;
                khi = pu == u ? kb.hi[u] : khi;
            }
            int64_t sl = -1;
            if (has) {
                const uint64_t row = uint64_t(m.Row(0)) + uint64_t(pu) * 64u;
                const uint32_t h = Fold32(klo, khi);
                const bool byHash = hashed && (khi >> 63) != 0;  // (exact keys keep the top of hi clear)
                auto sameRows = [&](uint64_t rep) { return SameKeyRows(m.cols, s, row, rep); };
                if (L > 0) sl = LdsUpsertDense(lLo, lHi, lRep, lDense, lCount, lHotKey, uint32_t(L - 1), h, klo, khi, twoWords, row, ldsProbes, byHash, sameRows);
                if (kDefer && sl < 0) {
                    if (twoWords && !s.rec_raw64) rs.rec_hi[row] = khi;  // (the lo word: at the end of the tile, for every row)
                    sl = kDeferredSlot;
                } else if (sl < 0) {
                    const uint64_t start = TablePos(klo, khi);
                    int64_t g;
                    if (byHash) {
                        g = TableUpsertVerified(tKeyLo, tKeyHi, tRepRow, tableMask, regionMask, start, klo, khi, row, 4096u, sameRows, abortWord);
                    } else {
                        g = TableUpsert(tKeyLo, tKeyHi, tableMask, regionMask, start, klo, khi, twoWords, 4096u, abortWord);
                        if (hashed && g >= 0) atomicMin(reinterpret_cast<unsigned long long*>(tRepRow + g), static_cast<unsigned long long>(row));
                    }
                    if (g < 0) {
                        if (abortWord != nullptr && !(m.err & SCAN_TABLE_FULL)) atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
                        m.err |= SCAN_TABLE_FULL;
                        m.P &= ~(1u << pu);
                    } else {
                        sl = -2 - g;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) slot[u] = (has && pu == u) ? sl : slot[u];
            if (kDefer) {
                l1Tried += uint32_t(__popcll(__ballot(has)));
                l1Found += uint32_t(__popcll(__ballot(has && sl >= 0)));
            }
        }
        // Level 1 keeps the first groups a workgroup meets.  Among millions of groups it is soon full of keys that
        // never come again, and every row pays eight probes to learn that: once the table is full, a wave that finds
        // fewer than one row in eight there stops claiming slots and reads one slot per row instead (ClickBench
        // UserID; SearchPhrase, where the empty phrase is most rows, goes on as before).
        if (kDefer && !l1Bypass && l1Tried >= 2048u) {
            if (L == 0 || (l1Found * 8u < l1Tried && __hip_atomic_load(lCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= uint32_t(L))) l1Bypass = true;
            l1Tried = l1Found = 0;
        }

        SSA_T(3);
        // membership of the lane's rows in the register-held groups: bit u of hit[g]; as 0.0 / 1.0 factors for the
        // floating-point sums (one FMA per row and group)
        uint32_t hit[kHotGroups];
        double one[kHotGroups][U];
        uint32_t hotRows = 0;
#pragma unroll
        for (int g = 0; g < kHotGroups; ++g) {
            hit[g] = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = useHot && slot[u] == g;
                hit[g] |= (in ? 1u : 0u) << u;
                one[g][u] = in ? 1.0 : 0.0;
            }
            hotRows |= hit[g];
        }

        // Register path: aggregates the host marked as plain 64-bit SUMs over an LDS-resident operand (a staged
        // column or an R register) and the row counter.  Kept tiny on purpose: it is unrolled per aggregate.
        uint32_t hotDone = 0;
        if (useHot && !(m.S().debug_flags & 2)) {
#pragma unroll
            for (int k = 0; k < kHotAggs; ++k) {
                const int hk = int(SSA_ROW(tHotKind, (m.S().debug_flags & 32) ? 0 : s.hot_kind[k], k));
                if (hk == HOT_SUM_F64) {
                    const double* src = reinterpret_cast<const double*>(gLds + SSA_ROW(tHotOff, s.hot_off[k], k));
                    double x[U];
                    bool finite = true;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        x[u] = src[m.Idx(u)];
                        finite = finite && (x[u] - x[u] == 0.0);
                    }
                    // x * 0.0 must be 0: an infinity / NaN anywhere in the wave's rows sends this aggregate through
                    // the tables for this tile
                    if (__all(finite)) {
#pragma unroll
                        for (int g = 0; g < kHotGroups; ++g) {
                            double acc = BitsToF64(hot[g][k]);
#pragma unroll
                            for (int u = 0; u < U; ++u) acc = __builtin_fma(x[u], one[g][u], acc);
                            hot[g][k] = F64ToBits(acc);
                        }
                        hotDone |= 1u << k;
                    }
                } else if (hk == HOT_SUM_I64) {
                    const uint64_t* src = reinterpret_cast<const uint64_t*>(gLds + SSA_ROW(tHotOff, s.hot_off[k], k));
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint64_t x = src[m.Idx(u)];
#pragma unroll
                        for (int g = 0; g < kHotGroups; ++g) hot[g][k] += ((hit[g] >> u) & 1u) ? x : 0ull;
                    }
                    hotDone |= 1u << k;
                } else if (hk == HOT_ROWS) {
#pragma unroll
                    for (int g = 0; g < kHotGroups; ++g) hot[g][k] += uint64_t(__popc(hit[g]));
                    hotDone |= 1u << k;
                }
            }
        }

        SSA_T(4);
        // Table path: everything else, and the rows of groups that are not register-held.
        const bool tablesIdle = hotDone == ((1u << naggs) - 1u) && !__any((m.P & ~hotRows) != 0);
        const int tableAggs = (tablesIdle || (m.S().debug_flags & (2 | 16))) ? 0 : naggs;
        uint32_t recMask[U];  // deferred rows: the accumulators the row updates
#pragma unroll
        for (int u = 0; u < U; ++u) recMask[u] = 0;
        int valSlot = 0;  // operands of the accumulators before k that records carry
        SSA_UNROLL_STATIC
        for (int k = 0; k < naggs; ++k) {
            if (k >= tableAggs) break;
            const int recSlot = valSlot;
            if (kDefer && int32_t(SSA_ROW(tAggKind, s.aggs[k].kind, k)) != AGG_COUNT) ++valSlot;
            uint32_t live = m.P & (((hotDone >> k) & 1u) ? ~hotRows : ~0u);
            if (!__any(live != 0)) continue;
            AggSpec spec;
            spec.kind = int32_t(SSA_ROW(tAggKind, s.aggs[k].kind, k));
            spec.reg = int32_t(SSA_ROW(tAggReg, s.aggs[k].reg, k));
            spec.check_valid = int32_t(SSA_ROW(tAggCheck, s.aggs[k].check_valid, k));
            spec.fetch_kind = SSA_ROW(tAggFk, s.aggs[k].fetch_kind, k);
            spec.fetch_payload = SSA_ROW(tAggPayload, s.aggs[k].fetch_payload, k);
            spec.flags = int32_t(SSA_ROW(tAggFlags, s.aggs[k].flags, k));
            Val<U> x;
            if (spec.reg != -1) {
                m.FetchOperand(spec.fetch_kind, spec.fetch_payload, x);
                if (spec.check_valid) live &= x.ok;
            }
            if (live == 0) continue;
            const int kind = spec.kind;
            if (spec.check_valid) {
                // a nullable argument was seen for the group: set once
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!((live >> u) & 1u)) continue;
                    if (slot[u] >= 0) {
                        if (!((lSeen[slot[u]] >> k) & 1u)) atomicOr(lSeen + slot[u], 1u << k);
                    } else if (kDefer && slot[u] == kDeferredSlot) {
                        // (the record's mask says so)
                    } else {
                        uint32_t* g = tSeen + (-2 - slot[u]);
                        if (!((__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> k) & 1u)) atomicOr(g, 1u << k);
                    }
                }
            }
            const uint32_t counted = live;  // the rows the group has seen an argument of, NaNs included
            if (kind == AGG_COUNT) {
#pragma unroll
                for (int u = 0; u < U; ++u) x.v[u] = 1;
            } else if (kind == AGG_MIN_F || kind == AGG_MAX_F) {
                // fmin / fmax semantics: a NaN never replaces a number (and the identity is an infinity)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double d = BitsToF64(x.v[u]);
                    if (d != d) {
                        if ((spec.flags & AGG_NAN_UNSUPPORTED) && ((live >> u) & 1u)) m.err |= SCAN_NAN_ORDER;
                        live &= ~(1u << u);
                    }
                    x.v[u] = OrderedFromF64(x.v[u]);
                }
            }
            uint64_t* accK = lAcc + size_t(k) * L * R;
            uint64_t* tabK = tAcc + size_t(k) * accStride;
            if (kind == AGG_MIN_S || kind == AGG_MAX_S) {
                const DevColumn& strCol = m.cols[SSA_ROW(tAggStrCol, s.aggs[k].str_col, k)];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!((live >> u) & 1u)) continue;
                    StringBestUpdate(strCol, kind == AGG_MIN_S, slot[u] >= 0 ? accK + slot[u] * R + rep : tabK + (-2 - slot[u]), x.v[u]);
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (kDefer && slot[u] == kDeferredSlot) {
                    // (a NaN the MIN / MAX passes over still marks the accumulator as seen: the record carries the identity)
                    if (!((counted >> u) & 1u)) continue;
                    recMask[u] |= 1u << k;
                    if (kind != AGG_COUNT) {
                        rs.rec_val[uint64_t(recSlot) * uint64_t(m.nrows) + uint64_t(m.Row(u))] = ((live >> u) & 1u) ? x.v[u] : TableIdentity(kind);
                    }
                    continue;
                }
                if (!((live >> u) & 1u)) continue;
                if (slot[u] >= 0) {
                    AccUpdate(kind, accK + slot[u] * R + rep, x.v[u]);
                } else {
                    AccUpdate(kind, tabK + (-2 - slot[u]), x.v[u]);
                }
            }
        }
        if (kDefer && s.rec_raw64) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool leaves = slot[u] == kDeferredSlot;
                const uint64_t present = __ballot(leaves);
                if (m.Row(u) < m.nrows) rs.rec_lo[uint64_t(m.Row(u))] = kb.lo[u] | (kb.hi[u] << 63);  // (Finish() undone)
                if (lane == 0 && m.Row(u) < m.nrows) rs.rec_present[uint64_t(m.Row(u)) >> 6] = present;
                if (rs.rec_mask != nullptr && leaves) rs.rec_mask[uint64_t(m.Row(u))] = recMask[u];
            }
        } else if (kDefer) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (m.Row(u) < m.nrows) rs.rec_lo[uint64_t(m.Row(u))] = slot[u] == kDeferredSlot ? kb.lo[u] : kEmpty;
                if (rs.rec_mask != nullptr && slot[u] == kDeferredSlot) rs.rec_mask[uint64_t(m.Row(u))] = recMask[u];
            }
        }
        SSA_T(5);
    }

#ifdef SSA_PHASE_TIMING
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.status) + 8;
        for (int i = 0; i < 8; ++i) atomicAdd(dbg + i, static_cast<unsigned long long>(tphase[i]));
    }
#endif
    // fold the register accumulators into level 1 (dense ids 0 .. kHotGroups-1): wave butterfly, then one lane adds
    if (useHot) {
#pragma unroll
        for (int k = 0; k < kHotAggs; ++k) {
            const int hk = int(SSA_ROW(tHotKind, (m.S().debug_flags & 32) ? 0 : s.hot_kind[k], k));
            if (hk == HOT_NONE) continue;
#pragma unroll
            for (int g = 0; g < kHotGroups; ++g) {
                uint64_t v = hot[g][k];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const uint64_t o = uint64_t(__shfl_xor(static_cast<unsigned long long>(v), off, 64));
                    v = hk == HOT_SUM_F64 ? F64ToBits(BitsToF64(v) + BitsToF64(o)) : v + o;
                }
                if ((threadIdx.x & 63) == 0 && v != 0) AccUpdate(hk == HOT_SUM_F64 ? AGG_SUM_F : AGG_SUM_I, lAcc + (size_t(k) * L + g) * R, v);
            }
        }
    }

    // merge level 1 into level 2
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += kBlock) {
        const uint64_t lo = lLo[i];
        if (lo == kEmpty) continue;
        const uint64_t hi = twoWords ? lHi[i] : 0ull;
        const uint32_t d = lDense[i];
        const uint32_t h = Fold32(lo, twoWords ? hi : 0ull);
        const uint64_t start = TablePos(lo, twoWords ? hi : 0ull);
        int64_t g;
        if (hashed && (hi >> 63) != 0) {
            const uint64_t row = lRep[i];
            g = TableUpsertVerified(tKeyLo, tKeyHi, tRepRow, tableMask, regionMask, start, lo, hi, row, 4096u,
                                    [&](uint64_t rep) { return SameKeyRows(m.cols, s, row, rep); }, abortWord);
        } else {
            g = TableUpsert(tKeyLo, tKeyHi, tableMask, regionMask, start, lo, hi, twoWords, 4096u, abortWord);
            if (hashed && g >= 0) atomicMin(reinterpret_cast<unsigned long long*>(tRepRow + g), static_cast<unsigned long long>(lRep[i]));
        }
        if (g < 0) {
            if (abortWord != nullptr && !(m.err & SCAN_TABLE_FULL)) atomicOr(a.status, uint32_t(SCAN_TABLE_FULL));
            m.err |= SCAN_TABLE_FULL;
            continue;
        }
        for (int k = 0; k < naggs; ++k) {
            const int kind = s.aggs[k].kind;
            const uint64_t* src = lAcc + (size_t(k) * L + d) * R;
            if (kind == AGG_MIN_S || kind == AGG_MAX_S) {
                for (int r = 0; r < R; ++r) {
                    if (src[r] != ~0ull) StringBestUpdate(m.cols[s.aggs[k].str_col], kind == AGG_MIN_S, tAcc + size_t(k) * accStride + g, src[r]);
                }
                continue;
            }
            uint64_t v = src[0];
            for (int r = 1; r < R; ++r) v = TableCombine(kind, v, src[r]);
            if (v != TableIdentity(kind)) AccUpdate(kind, tAcc + size_t(k) * accStride + g, v);
        }
        const uint32_t seen = lSeen[d];
        if (seen) atomicOr(tSeen + g, seen);
    }
    m.ReportErrors();
#undef SSA_ROW
}

#ifdef SSA_SPEC
}  // namespace ssa
extern "C" __global__ __launch_bounds__(ssa::kBlock) void ssa_spec_keyed(const ssa::KArgs<ssa::KeyedSink>) {
    ssa::ScanKeyedBody<SSA_SPEC_U, SSA_SPEC_HASHED != 0>();
}
// The same kernel under the name the variant without register-held groups is launched by (a step's first large scans
// time both variants: in a kernel trace the two then show as two kernels, not as one with a bimodal duration).
extern "C" __global__ __launch_bounds__(ssa::kBlock) void ssa_spec_keyed_plain(const ssa::KArgs<ssa::KeyedSink>) {
    ssa::ScanKeyedBody<SSA_SPEC_U, SSA_SPEC_HASHED != 0>();
}
// ... and the variant that compacts every tile to the rows that passed the filter (built with SSA_SPEC_COMPACT).
extern "C" __global__ __launch_bounds__(ssa::kBlock) void ssa_spec_keyed_compact(const ssa::KArgs<ssa::KeyedSink>) {
    ssa::ScanKeyedBody<SSA_SPEC_U, SSA_SPEC_HASHED != 0>();
}
namespace ssa {
#else
// Two waves per SIMD at least: the register allocator has to stay within 256 VGPRs.
template <int U, bool kHashed>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void ScanKeyedKernel(const KArgs<KeyedSink>) {
    ScanKeyedBody<U, kHashed>();
}
#endif

}  // namespace ssa
