// CPU restatement of the reference's compute-sharding step.  TEST INFRASTRUCTURE: only tests/, the smoke check and the
// benchmark's CPU baseline may use it; the GPU product never does.
//
//  * orc_xxh64: XXH64 as published in the xxHash specification (the reference vendors xxHash 0.8.2 under
//    contrib/libs/xxhash, which is not compiled here; ydb/core/formats/arrow/hash/xx_hash.cpp:5-15 calls its streaming
//    XXH64_reset / XXH64_update / XXH64_digest, whose result equals the one-shot hash of the concatenated input --
//    the property ydb/core/tx/sharding/ut/ut_sharding.cpp:14-31 asserts).  Pinned by the specification's known answers
//    and against the independent `xxhash` Python package on random inputs (tests/test_oracle_sharding.py).
//  * orc_sharding_split: TShardingSplitIndex::Initialize (ydb/core/formats/arrow/permutations.h:84-125) -- shard of a
//    row = |hash as int64| % shards, per-shard lists in row order, then the size correction that moves the tail of the
//    longest list into the shortest one.  PARITY UNPINNED: the reference holds no known-answer test for this function;
//    where its std::sort leaves the order of equally long lists open, a stable sort (shard number breaks ties) is used.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <vector>

namespace {

constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;

uint64_t Rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
uint64_t Read64(const uint8_t* p) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}
uint32_t Read32(const uint8_t* p) {
    uint32_t v;
    std::memcpy(&v, p, 4);
    return v;
}
uint64_t Round(uint64_t acc, uint64_t in) {
    acc += in * P2;
    acc = Rotl(acc, 31);
    return acc * P1;
}
uint64_t Merge(uint64_t acc, uint64_t v) {
    acc ^= Round(0, v);
    return acc * P1 + P4;
}

}  // namespace

extern "C" {

uint64_t orc_xxh64(const uint8_t* p, uint64_t len, uint64_t seed) {
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* limit = end - 32;
        do {
            v1 = Round(v1, Read64(p));
            v2 = Round(v2, Read64(p + 8));
            v3 = Round(v3, Read64(p + 16));
            v4 = Round(v4, Read64(p + 24));
            p += 32;
        } while (p <= limit);
        h = Rotl(v1, 1) + Rotl(v2, 7) + Rotl(v3, 12) + Rotl(v4, 18);
        h = Merge(h, v1);
        h = Merge(h, v2);
        h = Merge(h, v3);
        h = Merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += len;
    while (p + 8 <= end) {
        h ^= Round(0, Read64(p));
        h = Rotl(h, 27) * P1 + P4;
        p += 8;
    }
    if (p + 4 <= end) {
        h ^= uint64_t(Read32(p)) * P1;
        h = Rotl(h, 23) * P2 + P3;
        p += 4;
    }
    while (p < end) {
        h ^= uint64_t(*p) * P5;
        h = Rotl(h, 11) * P1;
        ++p;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

// Hashes of n rows whose concatenated key bytes are rows[offsets[i] .. offsets[i + 1]).
void orc_xxh64_rows(const uint8_t* bytes, const int64_t* offsets, int64_t n, uint64_t seed, uint64_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = orc_xxh64(bytes + offsets[i], uint64_t(offsets[i + 1] - offsets[i]), seed);
}

// permutation[n]: row indices shard after shard; sizes[shards].  `hashes` are the hash column's values as the
// reference reads them into an i64 (permutations.h:88).
void orc_sharding_split(const int64_t* hashes, int64_t n, uint32_t shards, uint32_t* permutation, uint32_t* sizes) {
    std::vector<std::vector<uint32_t>> remap(shards);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t v = hashes[i];
        const uint64_t mag = v < 0 ? 0ull - uint64_t(v) : uint64_t(v);  // (-v of the most negative value overflows in the reference)
        remap[mag % shards].push_back(uint32_t(i));
    }
    const uint64_t expectation = uint64_t(n) / shards + 1;
    std::deque<std::vector<uint32_t>*> order;
    for (auto& l : remap) order.push_back(&l);
    std::stable_sort(order.begin(), order.end(), [](const std::vector<uint32_t>* l, const std::vector<uint32_t>* r) { return l->size() < r->size(); });
    while (order.size() > 1 && order.back()->size() > expectation && order.front()->size() < expectation) {
        const uint64_t useless = order.back()->size() - expectation;
        const uint64_t need = expectation - order.front()->size();
        const uint64_t move = std::min(need, useless);
        if (move == 0) break;
        std::vector<uint32_t> merged(order.front()->size() + move);
        std::merge(order.front()->begin(), order.front()->end(), order.back()->end() - move, order.back()->end(), merged.begin());
        *order.front() = std::move(merged);
        order.back()->resize(order.back()->size() - move);
        if (order.back()->size() <= expectation) order.pop_back();
        if (order.front()->size() >= expectation) order.pop_front();
    }
    int64_t at = 0;
    for (uint32_t s = 0; s < shards; ++s) {
        sizes[s] = uint32_t(remap[s].size());
        for (uint32_t r : remap[s]) permutation[at++] = r;
    }
}

}  // extern "C"
