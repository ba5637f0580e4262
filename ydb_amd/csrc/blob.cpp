// Column blobs -> HBM-resident record batches.
//
// The GPU counterpart of NArrow::NSerialization::TNativeSerializer::DoDeserialize
// (ydb/core/formats/arrow/serializer/native.cpp:18-81): the blob crosses PCIe as stored (compressed), its buffers are
// decoded on the device straight into the column buffers of an ssa_batch_t.  Codecs: none and LZ4_FRAME (the
// reference's default, native.h:54-57); ZSTD is declined with SSA_UNSUPPORTED.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "exec.h"
#include "ipc.h"

namespace ssa {

#define SSA_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
        }                                                                                              \
    } while (0)

#define SSA_RETURN_NOT_OK(expr)     \
    do {                            \
        Status s_ = (expr);         \
        if (!s_.ok()) return s_;    \
    } while (0)

namespace {

struct PendingBuffer {
    size_t Blob;       // which blob
    IpcBuffer Where;   // inside the body of one of that blob's messages ...
    size_t Body;       // ... which starts here in the blob ...
    int32_t Codec;     // ... and is compressed this way
    size_t MinBytes;   // bytes the column needs from it (0: whatever it holds)
    BufferPtr* Target; // receives the decoded buffer
    int64_t* Bytes;    // receives its length (may be null)
};

}  // namespace

bool Executor::EnsureBlobStaging() {
    if (BlobStaging) return true;
    void* pinned = nullptr;
    if (hipHostMalloc(&pinned, kBlobSlots * kBlobSlotBytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    for (hipEvent_t& e : BlobSlotDone) {
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(pinned);
            return false;
        }
    }
    BlobStaging = static_cast<uint8_t*>(pinned);
    return true;
}

// The device copy of the blobs is cut into pieces of a slot's size; a piece is put together in a slot (the parts of the
// blobs that fall into it, at the places they have on the device) and sent by one DMA copy.  Threads take the pieces in
// order; a slot is written again when the copy that last left it has completed.
Status Executor::UploadBlobs(const std::vector<std::pair<const uint8_t*, size_t>>& blobs, const std::vector<size_t>& base, size_t total,
                             char* deviceBase, bool& done) {
    done = false;
    if (total < 2 * kBlobSlotBytes || !EnsureBlobStaging()) return Status::OK();
    const size_t npieces = (total + kBlobSlotBytes - 1) / kBlobSlotBytes;
    const int nthreads = int(std::min<size_t>(4, npieces));
    std::atomic<size_t> next{0};
    std::atomic<int64_t> queued[kBlobSlots];  // the last piece whose copy out of the slot is queued, its event recorded
    for (auto& q : queued) q.store(-1);
    std::atomic<int> failed{0};
    std::mutex errLock;
    std::string errText;
    auto fail = [&](hipError_t e, const char* what) {
        std::lock_guard<std::mutex> g(errLock);
        if (!failed.exchange(1)) errText = std::string(what) + ": " + hipGetErrorString(e);
    };
    auto work = [&]() {
        (void)hipSetDevice(Device);
        size_t first = 0;  // the first blob that may reach into the piece (pieces come in rising order per thread)
        for (;;) {
            const size_t piece = next.fetch_add(1);
            if (piece >= npieces) return;
            const int slot = int(piece % kBlobSlots);
            const int64_t before = int64_t(piece) - kBlobSlots;
            while (queued[slot].load(std::memory_order_acquire) != (before >= 0 ? before : -1)) {
                if (failed.load()) return;
                std::this_thread::yield();
            }
            if (before >= 0) {
                const hipError_t e = hipEventSynchronize(BlobSlotDone[slot]);
                if (e != hipSuccess) {
                    fail(e, "hipEventSynchronize");
                    queued[slot].store(int64_t(piece), std::memory_order_release);
                    return;
                }
            }
            const size_t lo = piece * kBlobSlotBytes, hi = std::min(total, lo + kBlobSlotBytes);
            uint8_t* at = BlobStaging + size_t(slot) * kBlobSlotBytes;
            while (first + 1 < blobs.size() && base[first + 1] <= lo) ++first;
            for (size_t i = first; i < blobs.size() && base[i] < hi; ++i) {
                const size_t b0 = std::max(lo, base[i]), b1 = std::min(hi, base[i] + blobs[i].second);
                if (b0 < b1) std::memcpy(at + (b0 - lo), blobs[i].first + (b0 - base[i]), b1 - b0);
            }
            hipError_t e = failed.load() ? hipSuccess : hipMemcpyAsync(deviceBase + lo, at, hi - lo, hipMemcpyHostToDevice, StreamHandle);
            if (e == hipSuccess) e = hipEventRecord(BlobSlotDone[slot], StreamHandle);
            if (e != hipSuccess) fail(e, "hipMemcpyAsync");
            queued[slot].store(int64_t(piece), std::memory_order_release);
        }
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < nthreads; ++t) threads.emplace_back(work);
    work();
    for (auto& t : threads) t.join();
    if (failed.load()) {
        (void)hipGetLastError();
        return Status::Hip(errText);
    }
    done = true;
    return Status::OK();
}

Status Executor::ImportBlobs(const std::vector<std::pair<const uint8_t*, size_t>>& blobs, const std::vector<FieldDesc>& schema,
                             std::vector<BatchPtr>& out) {
    const Status st = ImportBlobsQueued(blobs, schema, out);
    // An error may leave copies queued on this executor's stream: out of the caller's blobs, and into buffers that go back
    // to the pool all executors share when this call's vectors die.  Nothing is handed back before the stream is past them.
    if (!st.ok()) {
        (void)hipStreamSynchronize(StreamHandle);
        (void)hipGetLastError();
    }
    return st;
}

Status Executor::ImportBlobsQueued(const std::vector<std::pair<const uint8_t*, size_t>>& blobs, const std::vector<FieldDesc>& schema,
                                   std::vector<BatchPtr>& out) {
    (void)hipSetDevice(Device);
    out.clear();
    const size_t nb = blobs.size();
    const bool timing = std::getenv("SSA_DEBUG_PLAN") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(StreamHandle);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "blobs: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    std::vector<IpcBatch> parsed(nb);
    std::vector<size_t> base(nb, 0);   // offset of each blob in the device copy
    size_t total = 0;
    for (size_t i = 0; i < nb; ++i) {
        if (!blobs[i].first && blobs[i].second) return Status::Invalid("null blob");
        SSA_RETURN_NOT_OK(ParseIpcBlob(blobs[i].first, blobs[i].second, schema, parsed[i]));
        // Row counts come from the blob's metadata, unchecked so far: the buffer sizes below are products of them, and a count
        // near 2^61 would wrap those products to a few bytes (Arrow's Validate() refuses such a batch; so does every kernel
        // here that counts rows in 32 bits).
        constexpr int64_t kMaxRows = INT32_MAX;
        bool rowsOk = parsed[i].Rows >= 0 && parsed[i].Rows <= kMaxRows;
        for (const IpcColumn& c : parsed[i].Columns) rowsOk = rowsOk && c.Length >= 0 && c.Length <= kMaxRows && c.NullCount <= c.Length;
        for (const IpcDictionary& d : parsed[i].Dicts) rowsOk = rowsOk && d.Rows >= 0 && d.Rows <= kMaxRows;
        if (!rowsOk) return Status::Invalid("a row count of the blob is negative or beyond 2^31 - 1");
        bool zstd = parsed[i].Codec == IPC_ZSTD;
        for (const IpcDictionary& d : parsed[i].Dicts) zstd = zstd || d.Codec == IPC_ZSTD;
        if (zstd) return Status::Unsupported("ZSTD body compression of an IPC blob (LZ4_FRAME and uncompressed bodies are decoded)");
        base[i] = total;
        total += (blobs[i].second + 255) & ~size_t(255);
    }
    // the blobs as stored, one copy each over PCIe
    BufferPtr dblob = Buffer::Alloc(total + 512);  // (the parse reads a window of 256 bytes ahead of where it stands)
    if (!dblob) return Status::Hip("out of device memory");
    bool uploaded = false;
    SSA_RETURN_NOT_OK(UploadBlobs(blobs, base, total, static_cast<char*>(dblob->Ptr), uploaded));
    for (size_t i = 0; i < nb && !uploaded; ++i) {
        if (blobs[i].second) {
            SSA_HIP(hipMemcpyAsync(static_cast<char*>(dblob->Ptr) + base[i], blobs[i].first, blobs[i].second, hipMemcpyHostToDevice, StreamHandle));
        }
    }

    mark("parse + copy to the device");
    std::vector<BatchPtr> batches(nb);
    std::vector<PendingBuffer> pending;
    // string columns (and dictionaries: their values are one) whose offsets are checked against their data buffers below
    struct StringCheck {
        Column* Col;
        int64_t Bytes;
    };
    std::vector<StringCheck> strings;
    size_t nstrings = 0;
    for (size_t i = 0; i < nb; ++i) {
        for (auto& c : parsed[i].Columns) nstrings += IsStringLike(c.Field.Type) && c.DictId < 0;
        nstrings += parsed[i].Dicts.size();
    }
    strings.reserve(nstrings);  // (the pending buffers keep addresses of its elements)
    std::vector<std::shared_ptr<DictInfo>> dictKeep;
    for (size_t i = 0; i < nb; ++i) {
        // the blob's dictionaries: plain string columns the encoded columns share by id
        std::vector<std::shared_ptr<DictInfo>> dicts;
        for (const IpcDictionary& d : parsed[i].Dicts) {
            if (d.NullCount > 0) return Status::Unsupported("dictionary with null entries");
            auto info = std::make_shared<DictInfo>();
            Column& e = info->Entries;
            e.Length = d.Rows;
            e.NullCount = 0;
            const size_t rows = size_t(d.Rows);
            if (rows == 0 && d.Values.Length == 0) {
                e.Values = Buffer::Alloc(16);  // a single zero offset
                if (!e.Values) return Status::Hip("out of device memory");
                SSA_HIP(hipMemsetAsync(e.Values->Ptr, 0, 16, StreamHandle));
            } else {
                pending.push_back({i, d.Values, d.BodyOffset, d.Codec, (rows + 1) * 4, &e.Values, nullptr});
            }
            strings.push_back({&e, 0});
            pending.push_back({i, d.Data, d.BodyOffset, d.Codec, 0, &e.Chars, &strings.back().Bytes});
            dicts.push_back(info);
            dictKeep.push_back(info);
        }
        auto b = std::make_shared<Batch>();
        b->Rows = parsed[i].Rows;
        b->Cols.resize(parsed[i].Columns.size());
        const size_t body = parsed[i].BodyOffset;
        const int32_t codec = parsed[i].Codec;
        for (size_t c = 0; c < parsed[i].Columns.size(); ++c) {
            const IpcColumn& ic = parsed[i].Columns[c];
            Column& col = b->Cols[c];
            col.Field = ic.Field;
            col.Field.Nullable = false;
            col.Length = ic.Length;
            col.NullCount = ic.NullCount;
            const size_t rows = size_t(ic.Length);
            if (ic.NullCount > 0) {
                if (ic.Validity.Length == 0) return Status::Invalid("column '" + ic.Field.Name + "' counts nulls but has no validity buffer");
                pending.push_back({i, ic.Validity, body, codec, (rows + 7) / 8, &col.Validity, nullptr});
                col.Field.Nullable = true;
            }
            if (ic.DictId >= 0) {
                // stays encoded in HBM (Column::Dict): the codes here, the values with the dictionary; a column of a type
                // of its own per use, sharing the buffers (DictInfo holds what is worked out per dictionary and use)
                for (size_t k = 0; k < parsed[i].Dicts.size(); ++k) {
                    if (parsed[i].Dicts[k].Id == ic.DictId) col.Dict = dicts[k];
                }
                if (!col.Dict) return Status::Invalid("column '" + ic.Field.Name + "' refers to a dictionary the blob does not carry");
                if (col.Dict->Entries.Field.Type != T_NONE && (col.Dict->Entries.Field.Type != ic.Field.Type || col.Dict->IndexType != ic.IndexType)) {
                    return Status::Unsupported("one dictionary shared by columns of differing types");
                }
                col.Dict->Entries.Field.Type = ic.Field.Type;
                col.Dict->Entries.Field.Name = ic.Field.Name;
                col.Dict->IndexType = ic.IndexType;
                pending.push_back({i, ic.Values, body, codec, rows * size_t(TypeWidth(ic.IndexType)), &col.Values, nullptr});
                continue;
            }
            size_t need;
            if (IsStringLike(ic.Field.Type)) need = (rows + 1) * 4;
            else if (ic.Field.Type == T_BOOL) need = (rows + 7) / 8;
            else need = rows * size_t(TypeWidth(ic.Field.Type));
            if (rows == 0 && IsStringLike(ic.Field.Type) && ic.Values.Length == 0) {
                col.Values = Buffer::Alloc(16);  // a single zero offset
                if (!col.Values) return Status::Hip("out of device memory");
                SSA_HIP(hipMemsetAsync(col.Values->Ptr, 0, 16, StreamHandle));
            } else {
                pending.push_back({i, ic.Values, body, codec, need, &col.Values, nullptr});
            }
            if (IsStringLike(ic.Field.Type)) {
                strings.push_back({&col, 0});
                pending.push_back({i, ic.Data, body, codec, 0, &col.Chars, &strings.back().Bytes});
            }
        }
        batches[i] = b;
    }

    // every buffer: a plain copy, or an LZ4 frame for the decode kernel
    std::vector<Lz4FrameTask> tasks;
    std::vector<Lz4BlockRef> blocks;
    uint32_t maxBlock = 0;
    for (PendingBuffer& pb : pending) {
        const uint8_t* host = blobs[pb.Blob].first;
        const size_t bufPos = pb.Body + size_t(pb.Where.Offset);  // inside the blob
        int64_t rawLen = pb.Where.Length;     // bytes to copy as they are
        size_t rawPos = bufPos;
        int64_t outLen = rawLen;
        bool frame = false;
        if (pb.Codec == IPC_LZ4_FRAME && pb.Where.Length > 0) {
            if (pb.Where.Length < 8) return Status::Invalid("compressed buffer without its length prefix");
            int64_t ulen;
            std::memcpy(&ulen, host + bufPos, 8);
            if (ulen == -1) {  // stored as is
                rawPos = bufPos + 8;
                rawLen = pb.Where.Length - 8;
                outLen = rawLen;
            } else {
                if (ulen < 0 || ulen > (int64_t(1) << 40)) return Status::Invalid("compressed buffer announces an impossible length");
                if (ulen >= (int64_t(1) << 32)) return Status::Unsupported("compressed buffer of 4 GB or more");  // (the decode kernel counts in 32 bits)
                frame = true;
                outLen = ulen;
            }
        }
        if (size_t(outLen) < pb.MinBytes) return Status::Invalid("a buffer of the blob is shorter than its column needs");
        BufferPtr dst = Buffer::Alloc(size_t(outLen) + 32);  // (strings are read in 16-byte pieces: padding behind the bytes)
        if (!dst) return Status::Hip("out of device memory");
        *pb.Target = dst;
        if (pb.Bytes) *pb.Bytes = outLen;
        if (outLen == 0) continue;
        if (!frame) {
            SSA_HIP(hipMemcpyAsync(dst->Ptr, static_cast<const char*>(dblob->Ptr) + base[pb.Blob] + rawPos, size_t(rawLen), hipMemcpyDeviceToDevice,
                                   StreamHandle));
            continue;
        }
        std::vector<Lz4BlockRef> frameBlocks;
        uint32_t blockMax = 0;
        bool linked = true;
        SSA_RETURN_NOT_OK(ParseLz4Frame(host, blobs[pb.Blob].second, bufPos + 8, size_t(pb.Where.Length - 8), frameBlocks, blockMax, linked));
        Lz4FrameTask t{};
        t.first_block = blocks.size();
        t.nblocks = uint32_t(frameBlocks.size());
        t.linked = linked ? 1u : 0u;
        t.dst = static_cast<uint8_t*>(dst->Ptr);
        t.dst_bytes = uint64_t(outLen);
        for (Lz4BlockRef& r : frameBlocks) {
            r.Src += base[pb.Blob];
            if (!r.Stored) maxBlock = std::max(maxBlock, r.SrcBytes);
            blocks.push_back(r);
        }
        tasks.push_back(t);
    }
    mark("buffers allocated, frames parsed, stored buffers copied");
    if (maxBlock > 90u * 1024u) return Status::Unsupported("LZ4 frame with blocks of more than 64 KB");
    if (!tasks.empty()) {
        BufferPtr dtasks = Buffer::Alloc(tasks.size() * sizeof(Lz4FrameTask));
        BufferPtr dblocks = Buffer::Alloc(std::max<size_t>(1, blocks.size()) * sizeof(Lz4BlockRef));
        if (!dtasks || !dblocks) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(dtasks->Ptr, tasks.data(), tasks.size() * sizeof(Lz4FrameTask), hipMemcpyHostToDevice, StreamHandle));
        if (!blocks.empty()) {
            SSA_HIP(hipMemcpyAsync(dblocks->Ptr, blocks.data(), blocks.size() * sizeof(Lz4BlockRef), hipMemcpyHostToDevice, StreamHandle));
        }
        // Room for the records of every block (a sequence per three compressed bytes at most): the parse then runs block by
        // block side by side and the frames' waves are left with the copies.  Without that room (the records of highly
        // compressible columns take twice the columns' own bytes) one wave does both, frame by frame.
        const char* splitEnv = std::getenv("SSA_LZ4_SPLIT");  // (read at every import: a test runs both ways in one process)
        const bool splitWanted = !(splitEnv && splitEnv[0] == '0');
        std::vector<uint64_t> recStart(blocks.size() + 1, 0);
        for (size_t g = 0; g < blocks.size(); ++g) recStart[g + 1] = recStart[g] + (blocks[g].Stored ? 0u : blocks[g].SrcBytes / 3u + 2u);
        BufferPtr drecStart, drecs, dcounts;
        if (splitWanted && !blocks.empty() && recStart.back() > 0 && recStart.back() * sizeof(Lz4Seq) <= (size_t(32) << 30)) {
            drecStart = Buffer::Alloc(recStart.size() * 8);
            dcounts = Buffer::Alloc(blocks.size() * sizeof(Lz4BlockSummary) + 64);
            drecs = Buffer::Alloc(size_t(recStart.back()) * sizeof(Lz4Seq) + 64);
            if (!drecStart || !dcounts || !drecs) {
                (void)hipGetLastError();
                drecs = nullptr;
            }
        }
        ProfileBegin();
        if (drecs) {
            SSA_HIP(hipMemcpyAsync(drecStart->Ptr, recStart.data(), recStart.size() * 8, hipMemcpyHostToDevice, StreamHandle));
            SSA_HIP(LaunchLz4DecodeSplit(static_cast<const uint8_t*>(dblob->Ptr), static_cast<const Lz4FrameTask*>(dtasks->Ptr), int64_t(tasks.size()),
                                         static_cast<const Lz4BlockRef*>(dblocks->Ptr), int64_t(blocks.size()), static_cast<const uint64_t*>(drecStart->Ptr),
                                         static_cast<Lz4Seq*>(drecs->Ptr), static_cast<Lz4BlockSummary*>(dcounts->Ptr), static_cast<uint32_t*>(StatusWord->Ptr),
                                         StreamHandle));
        } else {
            SSA_HIP(LaunchLz4Decode(static_cast<const uint8_t*>(dblob->Ptr), static_cast<const Lz4FrameTask*>(dtasks->Ptr), int64_t(tasks.size()),
                                    static_cast<const Lz4BlockRef*>(dblocks->Ptr), static_cast<uint32_t*>(StatusWord->Ptr), StreamHandle));
        }
        int64_t produced = 0;
        for (auto& t : tasks) produced += int64_t(t.dst_bytes);
        ProfileEnd("Lz4DecodeKernel", produced);
        // (the host vectors above are read by the copies: wait before they go out of scope)
        uint32_t st = 0;
        SSA_HIP(hipMemcpyAsync(&st, StatusWord->Ptr, 4, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        if (st != 0) {
            SSA_HIP(hipMemsetAsync(StatusWord->Ptr, 0, 4, StreamHandle));
            return Status::Invalid("corrupt LZ4 data in an IPC blob");
        }
    } else {
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    mark("decode");
    // RecordBatch::Validate of the reference: offsets of a string column must describe its data buffer.  The first and
    // the last offset of every string column, all copies queued before the one wait.
    {
        std::vector<int32_t> ends(2 * strings.size(), 0);
        for (size_t i = 0; i < strings.size(); ++i) {
            const Column& col = *strings[i].Col;
            if (col.Length == 0) continue;
            SSA_HIP(hipMemcpyAsync(&ends[2 * i], col.Values->Ptr, 4, hipMemcpyDeviceToHost, StreamHandle));
            SSA_HIP(hipMemcpyAsync(&ends[2 * i + 1], static_cast<const int32_t*>(col.Values->Ptr) + col.Length, 4, hipMemcpyDeviceToHost, StreamHandle));
        }
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        for (size_t i = 0; i < strings.size(); ++i) {
            if (strings[i].Col->Length == 0) continue;
            if (ends[2 * i] < 0 || ends[2 * i + 1] < ends[2 * i] || int64_t(ends[2 * i + 1]) > strings[i].Bytes) {
                return Status::Invalid("validation error: string offsets run past the data buffer");
            }
        }
    }
    mark("validation");
    out = batches;
    return Status::OK();
}

}  // namespace ssa
