// Compute sharding of a result batch on the device: THashConstructor::BuildHashUI64 + TShardingSplitIndex
// (ydb/core/formats/arrow/permutations.cpp:187-207, permutations.h:49-144; called from
// ydb/core/tx/columnshard/engines/reader/plain_reader/iterator/merge.cpp:38-45).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <deque>

#include "exec.h"

namespace ssa {

#define SSA_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { \
            (void)hipGetLastError(); /* the runtime keeps the error for the next caller otherwise */   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));        \
        } \
    } while (0)
#define SSA_RETURN_NOT_OK(expr) \
    do {                        \
        Status s_ = (expr);     \
        if (!s_.ok()) return s_; \
    } while (0)

namespace {

// The reference's seed for the compute-sharding hash (permutations.cpp:204).
constexpr uint64_t kComputeShardingSeed = 34323543;

bool IsPassThroughHashType(int t) { return t == T_U64 || t == T_U32 || t == T_I64 || t == T_I32; }

}  // namespace

Status Executor::ShardingHash(const Batch& in, const std::vector<std::string>& keyNames, uint64_t seed, BufferPtr& hashes) {
    (void)hipSetDevice(Device);
    if (keyNames.empty()) return Status::Invalid("sharding hash needs at least one column");
    if (keyNames.size() > size_t(kMaxShardKeys)) return Status::Unsupported("more than 8 sharding key columns");
    ShardKeyColumns keys{};
    keys.nkeys = 0;
    for (const std::string& name : keyNames) {
        const int ci = in.FindColumn(name);
        if (ci < 0) return Status::Invalid("sharding key column '" + name + "' is not in the batch");  // ENoColumnPolicy::Verify
        const Column& col = in.Cols[size_t(ci)];
        if (col.Field.Type == T_DEC128) continue;  // decimals add nothing to the hash (calcer.cpp:44-46)
        ShardKeyColumn& k = keys.cols[keys.nkeys++];
        k.values = col.ValuesPtr();
        k.validity = col.Validity ? static_cast<const uint8_t*>(col.Validity->Ptr) : nullptr;
        k.chars = nullptr;
        k.val_bit_off = int32_t(col.ValBitOff);
        k.valid_bit_off = int32_t(col.ValidBitOff);
        if (IsStringLike(col.Field.Type)) {
            // an all-empty string column may have no chars buffer: any non-null address will do, nothing is read
            k.chars = col.Chars ? static_cast<const uint8_t*>(col.Chars->Ptr) : reinterpret_cast<const uint8_t*>(col.ValuesPtr());
            k.width = 0;
        } else if (col.Field.Type == T_BOOL) {
            k.width = 0;
        } else {
            k.width = TypeWidth(col.Field.Type);
            if (k.width != 1 && k.width != 2 && k.width != 4 && k.width != 8) return Status::Unsupported("sharding key type");
        }
    }
    hashes = Buffer::Alloc(size_t(std::max<int64_t>(in.Rows, 1)) * 8);
    if (!hashes) return Status::Hip("out of device memory");
    if (in.Rows > 0) SSA_HIP(LaunchXX64Rows(keys, in.Rows, seed, static_cast<uint64_t*>(hashes->Ptr), StreamHandle));
    return Status::OK();
}

// The reference balances the lists after the split (permutations.h:101-125): with expectation = rows / shards + 1 it
// repeatedly moves the last records of the longest list into the shortest one (merging by row index) until no list is
// both above and another below the expectation.  Only the list sizes decide what moves, so the plan is made here from
// the per-shard counts; `keep[s]` rows of shard s stay, its later rows leave in segments (last ranks first).
// Lists of equal size are ordered by shard index (the reference's std::sort leaves that order unspecified).
void PlanSizeCorrection(const std::vector<uint64_t>& sizes, std::vector<uint32_t>& keep,
                        std::vector<std::vector<std::pair<uint32_t, uint32_t>>>& segments /* per donor: (first rank, receiver) */,
                        std::vector<uint64_t>& finalSizes) {
    const uint32_t shards = uint32_t(sizes.size());
    uint64_t rows = 0;
    for (uint64_t s : sizes) rows += s;
    const uint64_t expectation = rows / shards + 1;
    finalSizes = sizes;
    keep.assign(shards, 0);
    for (uint32_t s = 0; s < shards; ++s) keep[s] = uint32_t(sizes[s]);
    segments.assign(shards, {});
    std::deque<uint32_t> order(shards);
    for (uint32_t s = 0; s < shards; ++s) order[s] = s;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t l, uint32_t r) { return sizes[l] < sizes[r]; });
    while (order.size() > 1 && finalSizes[order.back()] > expectation && finalSizes[order.front()] < expectation) {
        const uint32_t big = order.back(), small = order.front();
        const uint64_t useless = finalSizes[big] - expectation;
        const uint64_t need = expectation - finalSizes[small];
        const uint64_t move = std::min(need, useless);
        if (move == 0) break;
        // the donor's current last `move` records: ranks [keep - move, keep)
        keep[big] -= uint32_t(move);
        segments[big].push_back({keep[big], small});
        finalSizes[big] -= move;
        finalSizes[small] += move;
        if (finalSizes[big] <= expectation) order.pop_back();
        if (finalSizes[small] >= expectation) order.pop_front();
    }
}

// `balance`: the reference's size correction.  Without it a row's shard depends on its key alone, which is what an exchange
// between GPUs needs (equal keys of all senders end in one place); with it the last rows of long lists move elsewhere.
Status Executor::ShardingSplit(const Batch& in, const std::vector<std::string>& keyNames, uint32_t shards, BatchPtr& out,
                               std::vector<uint32_t>& shardRows, bool balance) {
    (void)hipSetDevice(Device);
    if (shards == 0) return Status::Invalid("shards count is zero");
    if (shards > kMaxShards) return Status::Unsupported("more than 512 shards");
    if (in.Rows >= (int64_t(1) << 31)) return Status::Unsupported("more than 2^31 rows in one batch");
    shardRows.assign(shards, 0);
    const int64_t n = in.Rows;
    // one shard, or no key columns: the batch as it is (permutations.cpp:261-263, :188-190)
    if (shards == 1 || keyNames.empty() || n == 0) {
        out = std::make_shared<Batch>(in);
        shardRows[0] = uint32_t(n);
        return Status::OK();
    }
    // BuildHashUI64: a single 32/64-bit integer key is its own hash, everything else goes through XXH64
    const void* hashPtr = nullptr;
    int hashType = T_U64;
    BufferPtr hashes;
    if (keyNames.size() == 1) {
        const int ci = in.FindColumn(keyNames[0]);
        if (ci < 0) {  // "cannot_build_hash": not sharded
            out = std::make_shared<Batch>(in);
            shardRows[0] = uint32_t(n);
            return Status::OK();
        }
        if (IsPassThroughHashType(in.Cols[size_t(ci)].Field.Type)) {
            hashPtr = in.Cols[size_t(ci)].ValuesPtr();
            hashType = in.Cols[size_t(ci)].Field.Type;
        }
    }
    if (!hashPtr) {
        SSA_RETURN_NOT_OK(ShardingHash(in, keyNames, kComputeShardingSeed, hashes));
        hashPtr = hashes->Ptr;
        hashType = T_U64;
    }

    const int64_t nblocks = ShardBlocks(n);
    const int64_t ncounts = int64_t(shards) * nblocks;
    BufferPtr shardIds = Buffer::Alloc(size_t(n) * 4);
    BufferPtr counts = Buffer::Alloc(size_t(ncounts + 1) * 4 + 16);
    BufferPtr offsets = Buffer::Alloc(size_t(ncounts + 1) * 4 + 16);
    BufferPtr scanTmp = Buffer::Alloc(size_t((ncounts + 4095) / 4096 + 2) * 8 + 16);
    BufferPtr perm = Buffer::Alloc(size_t(n) * 8);
    BufferPtr firstsDev = Buffer::Alloc(size_t(shards + 1) * 4 + 16);
    if (!shardIds || !counts || !offsets || !scanTmp || !perm || !firstsDev) return Status::Hip("out of device memory");
    uint32_t* ids = static_cast<uint32_t*>(shardIds->Ptr);

    auto countAndScan = [&](const void* h, const uint32_t* given, uint32_t* idsOut, std::vector<uint64_t>& sizes) -> Status {
        SSA_HIP(LaunchShardCount(h, hashType, given, n, shards, idsOut, static_cast<int32_t*>(counts->Ptr), nullptr, StreamHandle));
        SSA_HIP(LaunchExclusiveScanI32(static_cast<const int32_t*>(counts->Ptr), static_cast<int32_t*>(offsets->Ptr), ncounts,
                                       static_cast<int64_t*>(scanTmp->Ptr), StreamHandle));
        // the first offset of every shard (and the grand total behind the last one)
        std::vector<int32_t> firsts(shards + 1, 0);
        SSA_HIP(LaunchGatherStrided(static_cast<const int32_t*>(offsets->Ptr), nblocks, int(shards + 1), static_cast<int32_t*>(firstsDev->Ptr),
                                    StreamHandle));
        SSA_HIP(hipMemcpyAsync(firsts.data(), firstsDev->Ptr, size_t(shards + 1) * 4, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        sizes.assign(shards, 0);
        for (uint32_t s = 0; s < shards; ++s) sizes[s] = uint64_t(firsts[s + 1] - firsts[s]);
        return Status::OK();
    };

    std::vector<uint64_t> sizes;
    SSA_RETURN_NOT_OK(countAndScan(hashPtr, nullptr, ids, sizes));
    std::vector<uint32_t> keep;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> segments;
    std::vector<uint64_t> finalSizes;
    if (balance) {
        PlanSizeCorrection(sizes, keep, segments, finalSizes);
    } else {
        finalSizes = sizes;
    }
    bool corrected = false;
    for (const auto& s : segments) corrected = corrected || !s.empty();

    ShardCorrection none{nullptr, nullptr, nullptr, nullptr};
    if (corrected) {
        // tables of the plan; a donor's segments in ascending order of their first rank
        std::vector<uint32_t> segBegin(shards + 1, 0), segStart, segTo;
        for (uint32_t s = 0; s < shards; ++s) {
            segBegin[s] = uint32_t(segStart.size());
            auto segs = segments[s];
            std::sort(segs.begin(), segs.end());
            for (const auto& sg : segs) {
                segStart.push_back(sg.first);
                segTo.push_back(sg.second);
            }
        }
        segBegin[shards] = uint32_t(segStart.size());
        const size_t words = keep.size() + segBegin.size() + segStart.size() + segTo.size();
        BufferPtr tables = Buffer::Alloc(words * 4 + 16);
        BufferPtr ids2 = Buffer::Alloc(size_t(n) * 4);
        if (!tables || !ids2) return Status::Hip("out of device memory");
        std::vector<uint32_t> host;
        host.insert(host.end(), keep.begin(), keep.end());
        host.insert(host.end(), segBegin.begin(), segBegin.end());
        host.insert(host.end(), segStart.begin(), segStart.end());
        host.insert(host.end(), segTo.begin(), segTo.end());
        SSA_HIP(hipMemcpyAsync(tables->Ptr, host.data(), host.size() * 4, hipMemcpyHostToDevice, StreamHandle));
        const uint32_t* t = static_cast<const uint32_t*>(tables->Ptr);
        ShardCorrection corr{t, t + shards, t + shards + (shards + 1), t + shards + (shards + 1) + segStart.size()};
        SSA_HIP(LaunchShardPlace(ids, n, shards, static_cast<const int32_t*>(offsets->Ptr), nullptr,
                                 static_cast<uint32_t*>(ids2->Ptr), corr, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));  // `host` is read by the copy above
        std::vector<uint64_t> sizes2;
        SSA_RETURN_NOT_OK(countAndScan(nullptr, static_cast<const uint32_t*>(ids2->Ptr), nullptr, sizes2));
        if (sizes2 != finalSizes) return Status::Invalid("internal: sharding size correction mismatch");
        SSA_HIP(LaunchShardPlace(static_cast<const uint32_t*>(ids2->Ptr), n, shards, static_cast<const int32_t*>(offsets->Ptr),
                                 static_cast<uint64_t*>(perm->Ptr), nullptr, none, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    } else {
        SSA_HIP(LaunchShardPlace(ids, n, shards, static_cast<const int32_t*>(offsets->Ptr), static_cast<uint64_t*>(perm->Ptr), nullptr,
                                 none, StreamHandle));
    }
    for (uint32_t s = 0; s < shards; ++s) shardRows[s] = uint32_t(finalSizes[s]);

    // TShardingSplitIndex::Apply: Take(input, permutation); the shards are consecutive slices of the result
    out = std::make_shared<Batch>();
    out->Rows = n;
    for (const Column& col : in.Cols) {
        Column dst;
        SSA_RETURN_NOT_OK(GatherRows(col, perm, n, col.Validity != nullptr, dst));
        dst.Field = col.Field;
        dst.Field.Nullable = dst.Validity != nullptr || col.Field.Nullable;
        out->Cols.push_back(std::move(dst));
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

}  // namespace ssa
