// Result batches between the GPUs of a sharded scan: pack into one flat buffer (the payload of a collective), unpack
// as views of the received buffer, concatenate the parts of all ranks into the input of the merge program.
//
// The reference ships every shard's partial result to the compute node that merges them
// (ydb/core/tx/columnshard/columnshard__scan.cpp:300-340, the rows of TEvScanData); here the parts never leave device
// memory: each is laid out as [column buffers], described by a small header that travels separately (on the host side
// of the collective), and the concatenation is one kernel launch over a list of copy segments.
#include <hip/hip_runtime.h>

#include <cstring>

#include "exec.h"

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

constexpr uint32_t kPackMagic = 0x42415353u;  // "SSAB"
constexpr uint64_t kNone = ~0ull;

struct PackedHeader {
    uint32_t magic, ncols;
    int64_t rows;
    uint64_t data_bytes;
};

struct PackedColumn {
    int32_t type, precision, scale, name_len;
    uint64_t values_off, values_bytes;  // fixed width: the rows' values; BOOL: the bytes that hold the rows' bits; strings: rows + 1 offsets
    uint64_t valid_off, valid_bytes;    // kNone: no validity
    int32_t val_bit_off, valid_bit_off; // bit of row 0 inside the first byte
    uint64_t chars_off, chars_bytes;    // strings: the bytes in use (the packed offsets start at 0)
};

uint64_t Align(uint64_t x) { return (x + 255) & ~uint64_t(255); }

struct Region {
    const void* src;
    uint64_t off, bytes;
    int64_t shift;  // string offsets: added to every offset (they start at 0 in the packed form); 0 = plain bytes
    bool offsets;
};

// The layout of `in` as a flat buffer, and where every region comes from.  `bounds`: first and last offset of every
// string column, in column order (the bytes in use; StringBounds).
Status Layout(const Batch& in, const std::vector<int32_t>& bounds, std::vector<uint8_t>* header, std::vector<Region>* regions, uint64_t& dataBytes) {
    std::vector<PackedColumn> cols;
    std::string names;
    uint64_t at = 0;
    size_t boundAt = 0;
    auto place = [&](const void* src, uint64_t bytes, bool offsets = false, int64_t shift = 0) {
        const uint64_t off = at;
        if (regions && bytes) regions->push_back(Region{src, off, bytes, shift, offsets});
        at = Align(at + bytes);
        return off;
    };
    for (const Column& c : in.Cols) {
        if (c.Dict || c.Field.IsScalar) return Status::Unsupported("packing a dictionary-encoded or scalar column");
        PackedColumn pc{};
        pc.type = c.Field.Type;
        pc.precision = c.Field.Precision;
        pc.scale = c.Field.Scale;
        pc.name_len = int32_t(c.Field.Name.size());
        names += c.Field.Name;
        const int64_t n = in.Rows;
        if (IsStringLike(c.Field.Type)) {
            int64_t lo = 0, hi = 0;
            if (c.Values && n > 0) {
                if (boundAt + 2 > bounds.size()) return Status::Invalid("internal: string bounds");
                lo = bounds[boundAt], hi = bounds[boundAt + 1];
                boundAt += 2;
                if (lo < 0 || hi < lo || (hi > lo && (!c.Chars || size_t(hi) > c.Chars->Size))) return Status::Invalid("string offsets run past the column's bytes");
            }
            pc.values_bytes = uint64_t(n + 1) * 4;
            pc.values_off = place(c.Values ? c.ValuesPtr() : nullptr, c.Values ? pc.values_bytes : 0, true, -lo);
            if (!c.Values) pc.values_bytes = 0;
            pc.chars_bytes = uint64_t(hi - lo);
            pc.chars_off = place(c.Chars ? static_cast<const char*>(c.Chars->Ptr) + lo : nullptr, pc.chars_bytes);
            at = Align(at + 16);  // (strings are read in 16-byte pieces: every bytes buffer keeps a readable tail)
        } else if (c.Field.Type == T_BOOL) {
            pc.val_bit_off = c.ValBitOff & 7;
            pc.values_bytes = uint64_t((pc.val_bit_off + n + 7) / 8);
            pc.values_off = place(c.Values ? static_cast<const char*>(c.Values->Ptr) + (c.ValBitOff >> 3) : nullptr, c.Values ? pc.values_bytes : 0);
        } else {
            pc.values_bytes = uint64_t(n) * uint64_t(TypeWidth(c.Field.Type));
            pc.values_off = place(c.Values ? c.ValuesPtr() : nullptr, c.Values ? pc.values_bytes : 0);
        }
        pc.valid_off = kNone;
        if (c.Validity) {
            pc.valid_bit_off = c.ValidBitOff & 7;
            pc.valid_bytes = uint64_t((pc.valid_bit_off + n + 7) / 8);
            pc.valid_off = place(static_cast<const char*>(c.Validity->Ptr) + (c.ValidBitOff >> 3), pc.valid_bytes);
        }
        cols.push_back(pc);
    }
    dataBytes = Align(at + 16);
    if (header) {
        PackedHeader h{kPackMagic, uint32_t(cols.size()), in.Rows, dataBytes};
        header->resize(sizeof(h) + cols.size() * sizeof(PackedColumn) + names.size());
        std::memcpy(header->data(), &h, sizeof(h));
        if (!cols.empty()) std::memcpy(header->data() + sizeof(h), cols.data(), cols.size() * sizeof(PackedColumn));
        if (!names.empty()) std::memcpy(header->data() + sizeof(h) + cols.size() * sizeof(PackedColumn), names.data(), names.size());
    }
    return Status::OK();
}

}  // namespace

Column ColumnView(const Column& col, int64_t start, int64_t count) {
    Column v = col;
    v.Length = count;
    if (col.Field.Type == T_BOOL) v.ValBitOff = int32_t(col.ValBitOff + start);
    else v.ElemOffset = col.ElemOffset + start;
    if (col.Validity) {
        v.ValidBitOff = int32_t(col.ValidBitOff + start);
        v.NullCount = -1;
    }
    return v;
}

// First and last offset of the string columns of `in` (columns without rows or offsets are left out), in column order.
Status Executor::StringBounds(const Batch& in, std::vector<int32_t>& bounds) {
    std::vector<const int32_t*> addrs;
    for (const Column& c : in.Cols) {
        if (!IsStringLike(c.Field.Type) || c.Dict || !c.Values || in.Rows == 0) continue;
        const int32_t* offs = static_cast<const int32_t*>(c.ValuesPtr());
        addrs.push_back(offs);
        addrs.push_back(offs + in.Rows);
    }
    bounds.assign(addrs.size(), 0);
    if (addrs.empty()) return Status::OK();
    (void)hipSetDevice(Device);
    BufferPtr dAddrs = Buffer::Alloc(addrs.size() * 8 + 64), dBounds = Buffer::Alloc(addrs.size() * 4 + 64);
    if (!dAddrs || !dBounds) return Status::Hip("out of device memory");
    SSA_HIP(hipMemcpyAsync(dAddrs->Ptr, addrs.data(), addrs.size() * 8, hipMemcpyHostToDevice, StreamHandle));
    SSA_HIP(LaunchGatherI32At(static_cast<const int32_t* const*>(dAddrs->Ptr), int64_t(addrs.size()), static_cast<int32_t*>(dBounds->Ptr), StreamHandle));
    SSA_HIP(hipMemcpyAsync(bounds.data(), dBounds->Ptr, addrs.size() * 4, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

Status Executor::PackedLayout(const Batch& in, std::vector<uint8_t>& header, uint64_t& dataBytes) {
    std::vector<int32_t> bounds;
    SSA_RETURN_NOT_OK(StringBounds(in, bounds));
    return Layout(in, bounds, &header, nullptr, dataBytes);
}

Status Executor::RunSegments(const std::vector<CopySegment>& segments) {
    if (segments.empty()) return Status::OK();
    uint64_t maxCount = 0;
    for (const CopySegment& s : segments) maxCount = std::max(maxCount, s.kind == SEG_I32_ADD ? s.count * 4 : s.count);
    BufferPtr dev = Buffer::Alloc(segments.size() * sizeof(CopySegment) + 64);
    if (!dev) return Status::Hip("out of device memory");
    SSA_HIP(hipMemcpyAsync(dev->Ptr, segments.data(), segments.size() * sizeof(CopySegment), hipMemcpyHostToDevice, StreamHandle));
    SSA_HIP(LaunchCopySegments(static_cast<const CopySegment*>(dev->Ptr), int(segments.size()), maxCount, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));  // (the host array and `dev` are read until here)
    return Status::OK();
}

// The column buffers of `in` into `dst` (device memory of at least the layout's data size): one launch.
Status Executor::PackData(const Batch& in, void* dst, size_t capacity) {
    (void)hipSetDevice(Device);
    std::vector<Region> regions;
    uint64_t dataBytes = 0;
    std::vector<int32_t> bounds;
    SSA_RETURN_NOT_OK(StringBounds(in, bounds));
    SSA_RETURN_NOT_OK(Layout(in, bounds, nullptr, &regions, dataBytes));
    if (dataBytes > capacity) return Status::Invalid("the packed batch does not fit the buffer");
    if (!dst) return Status::Invalid("null buffer");
    std::vector<CopySegment> segs;
    for (const Region& r : regions) {
        if (!r.src) continue;
        if (r.offsets) segs.push_back(CopySegment{static_cast<const uint8_t*>(r.src), static_cast<uint8_t*>(dst) + r.off, r.bytes / 4, SEG_I32_ADD, r.shift});
        else segs.push_back(CopySegment{static_cast<const uint8_t*>(r.src), static_cast<uint8_t*>(dst) + r.off, r.bytes, SEG_BYTES, 0});
    }
    return RunSegments(segs);
}

Status Executor::PackToHost(const Batch& in, std::vector<HostColumnView>& views, bool& done) {
    done = false;
    if (in.Rows <= 0 || in.Rows > 16384 || in.Cols.empty()) return Status::OK();
    for (const Column& c : in.Cols) {
        if (c.Dict || c.Field.IsScalar || !c.Values || c.Length != in.Rows) return Status::OK();
        if (IsStringLike(c.Field.Type) && !c.Chars) return Status::OK();
    }
    (void)hipSetDevice(Device);
    if (!EnsureHostStaging()) return Status::OK();  // (the copies buffer by buffer still work)
    std::vector<int32_t> bounds;
    SSA_RETURN_NOT_OK(StringBounds(in, bounds));
    std::vector<uint8_t> header;
    std::vector<Region> regions;
    uint64_t dataBytes = 0;
    SSA_RETURN_NOT_OK(Layout(in, bounds, &header, &regions, dataBytes));
    if (dataBytes > kHostStagingBytes) return Status::OK();
    std::vector<CopySegment> segs;
    BufferPtr staged = Buffer::Alloc(size_t(dataBytes) + 64);
    if (!staged) return Status::Hip("out of device memory");
    uint8_t* dst = static_cast<uint8_t*>(staged->Ptr);
    uint64_t maxCount = 0;
    for (const Region& r : regions) {
        if (!r.src) continue;
        if (r.offsets) segs.push_back(CopySegment{static_cast<const uint8_t*>(r.src), dst + r.off, r.bytes / 4, SEG_I32_ADD, r.shift});
        else segs.push_back(CopySegment{static_cast<const uint8_t*>(r.src), dst + r.off, r.bytes, SEG_BYTES, 0});
        maxCount = std::max(maxCount, r.bytes);
    }
    BufferPtr dsegs = Buffer::Alloc(segs.size() * sizeof(CopySegment) + 64);
    if (!dsegs) return Status::Hip("out of device memory");
    if (!segs.empty()) {
        SSA_HIP(hipMemcpyAsync(dsegs->Ptr, segs.data(), segs.size() * sizeof(CopySegment), hipMemcpyHostToDevice, StreamHandle));
        SSA_HIP(LaunchCopySegments(static_cast<const CopySegment*>(dsegs->Ptr), int(segs.size()), maxCount, StreamHandle));
    }
    SSA_HIP(hipMemcpyAsync(HostStaging, dst, size_t(dataBytes), hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));  // (the host array and the device buffers are read until here)
    views.assign(in.Cols.size(), HostColumnView{});
    for (size_t i = 0; i < in.Cols.size(); ++i) {
        PackedColumn pc;
        std::memcpy(&pc, header.data() + sizeof(PackedHeader) + i * sizeof(PackedColumn), sizeof(pc));
        HostColumnView& v = views[i];
        v.Values = HostStaging + pc.values_off;
        v.ValuesBytes = size_t(pc.values_bytes);
        v.ValBitOff = pc.val_bit_off;
        if (pc.valid_off != kNone) {
            v.Valid = HostStaging + pc.valid_off;
            v.ValidBytes = size_t(pc.valid_bytes);
            v.ValidBitOff = pc.valid_bit_off;
        }
        if (IsStringLike(pc.type)) {
            v.Chars = HostStaging + pc.chars_off;
            v.CharsBytes = size_t(pc.chars_bytes);
        }
    }
    done = true;
    return Status::OK();
}

// A batch whose columns are views of `data` (device memory that the caller keeps alive as long as the batch lives).
Status UnpackBatch(const uint8_t* header, size_t headerBytes, void* data, size_t dataBytes, BatchPtr& out) {
    PackedHeader h;
    if (!header || headerBytes < sizeof(h)) return Status::Invalid("packed batch: short header");
    std::memcpy(&h, header, sizeof(h));
    if (h.magic != kPackMagic || h.rows < 0 || h.ncols > 4096) return Status::Invalid("packed batch: not a header");
    if (headerBytes < sizeof(h) + size_t(h.ncols) * sizeof(PackedColumn)) return Status::Invalid("packed batch: short header");
    if (h.data_bytes > dataBytes) return Status::Invalid("packed batch: the data buffer is smaller than the header says");
    if (h.data_bytes > 0 && !data) return Status::Invalid("packed batch: null data");
    auto batch = std::make_shared<Batch>();
    batch->Rows = h.rows;
    size_t nameAt = sizeof(h) + size_t(h.ncols) * sizeof(PackedColumn);
    char* base = static_cast<char*>(data);
    auto inside = [&](uint64_t off, uint64_t bytes) { return off <= h.data_bytes && bytes <= h.data_bytes - off; };
    for (uint32_t i = 0; i < h.ncols; ++i) {
        PackedColumn pc;
        std::memcpy(&pc, header + sizeof(h) + size_t(i) * sizeof(PackedColumn), sizeof(pc));
        if (pc.name_len < 0 || nameAt + size_t(pc.name_len) > headerBytes) return Status::Invalid("packed batch: column name");
        if (pc.type < T_BOOL || pc.type > T_DEC128) return Status::Invalid("packed batch: column type");
        Column c;
        c.Field.Name.assign(reinterpret_cast<const char*>(header) + nameAt, size_t(pc.name_len));
        nameAt += size_t(pc.name_len);
        c.Field.Type = pc.type;
        c.Field.Precision = pc.precision;
        c.Field.Scale = pc.scale;
        c.Length = h.rows;
        if (!inside(pc.values_off, pc.values_bytes)) return Status::Invalid("packed batch: values outside the data");
        const uint64_t want = IsStringLike(pc.type) ? uint64_t(h.rows + 1) * 4
                              : pc.type == T_BOOL   ? uint64_t((pc.val_bit_off + h.rows + 7) / 8)
                                                    : uint64_t(h.rows) * uint64_t(TypeWidth(pc.type));
        if (pc.values_bytes != want && !(pc.values_bytes == 0 && h.rows == 0)) return Status::Invalid("packed batch: values size");
        c.Values = Buffer::Wrap(base + pc.values_off, size_t(pc.values_bytes));
        c.ValBitOff = pc.val_bit_off;
        if (IsStringLike(pc.type)) {
            if (!inside(pc.chars_off, pc.chars_bytes) || pc.chars_off + pc.chars_bytes + 16 > h.data_bytes) return Status::Invalid("packed batch: string bytes outside the data");
            c.Chars = Buffer::Wrap(base + pc.chars_off, size_t(pc.chars_bytes) + 16);
            if (pc.values_bytes == 0) c.Values = nullptr;
        }
        if (pc.valid_off != kNone) {
            if (!inside(pc.valid_off, pc.valid_bytes) || pc.valid_bytes != uint64_t((pc.valid_bit_off + h.rows + 7) / 8)) return Status::Invalid("packed batch: validity outside the data");
            c.Validity = Buffer::Wrap(base + pc.valid_off, size_t(pc.valid_bytes));
            c.ValidBitOff = pc.valid_bit_off;
            c.Field.Nullable = true;
            c.NullCount = -1;
        }
        batch->Cols.push_back(std::move(c));
    }
    out = batch;
    return Status::OK();
}

// All rows of `parts` (batches of one schema: same names and types; a column may carry nulls in some parts only) as one
// batch, part after part.  One launch copies every buffer; validity goes through bytes and is packed per column.
Status Executor::ConcatBatches(const std::vector<BatchPtr>& parts, BatchPtr& out) {
    (void)hipSetDevice(Device);
    if (parts.empty()) return Status::Invalid("no batches to concatenate");
    const Batch& first = *parts[0];
    int64_t total = 0;
    for (const BatchPtr& p : parts) {
        if (!p || p->Cols.size() != first.Cols.size()) return Status::Invalid("batches of differing schemas");
        for (size_t c = 0; c < first.Cols.size(); ++c) {
            const Column& col = p->Cols[c];
            if (col.Field.Name != first.Cols[c].Field.Name || col.Field.Type != first.Cols[c].Field.Type) return Status::Invalid("batches of differing schemas");
            if (col.Dict || col.Field.IsScalar) return Status::Unsupported("concatenating a dictionary-encoded or scalar column");
        }
        total += p->Rows;
    }
    if (parts.size() == 1) {
        out = parts[0];
        return Status::OK();
    }
    // string columns: the first and the last offset of every part (its bytes in use), fetched together
    std::vector<const int32_t*> addrs;
    for (size_t c = 0; c < first.Cols.size(); ++c) {
        if (!IsStringLike(first.Cols[c].Field.Type)) continue;
        for (const BatchPtr& p : parts) {
            if (p->Rows == 0) continue;
            const Column& src = p->Cols[c];
            if (!src.Values) return Status::Invalid("string column without offsets");
            const int32_t* offs = static_cast<const int32_t*>(src.ValuesPtr());
            addrs.push_back(offs);
            addrs.push_back(offs + p->Rows);
        }
    }
    std::vector<int32_t> bounds(addrs.size());
    if (!addrs.empty()) {
        BufferPtr dAddrs = Buffer::Alloc(addrs.size() * 8 + 64), dBounds = Buffer::Alloc(addrs.size() * 4 + 64);
        if (!dAddrs || !dBounds) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(dAddrs->Ptr, addrs.data(), addrs.size() * 8, hipMemcpyHostToDevice, StreamHandle));
        SSA_HIP(LaunchGatherI32At(static_cast<const int32_t* const*>(dAddrs->Ptr), int64_t(addrs.size()), static_cast<int32_t*>(dBounds->Ptr), StreamHandle));
        SSA_HIP(hipMemcpyAsync(bounds.data(), dBounds->Ptr, addrs.size() * 4, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    size_t boundAt = 0;
    auto res = std::make_shared<Batch>();
    res->Rows = total;
    std::vector<CopySegment> segs;
    std::vector<BufferPtr> validBytes(first.Cols.size());
    std::vector<BufferPtr> boolBytes(first.Cols.size());  // BOOL columns: the parts' bits as a byte per row, packed at the end
    for (size_t c = 0; c < first.Cols.size(); ++c) {
        Column col;
        col.Field = first.Cols[c].Field;
        col.Field.Nullable = false;
        col.Length = total;
        const int type = col.Field.Type;
        bool anyValidity = false;
        for (const BatchPtr& p : parts) anyValidity = anyValidity || (p->Cols[c].Validity && p->Rows > 0);
        if (anyValidity) {
            validBytes[c] = Buffer::Alloc(size_t(total) + 16);
            if (!validBytes[c]) return Status::Hip("out of device memory");
        }
        if (IsStringLike(type)) {
            uint64_t chars = 0;
            size_t b = boundAt;
            for (const BatchPtr& p : parts) {
                if (p->Rows == 0) continue;
                const int64_t lo = bounds[b], hi = bounds[b + 1];
                b += 2;
                const Column& src = p->Cols[c];
                if (lo < 0 || hi < lo || (hi > lo && (!src.Chars || size_t(hi) > src.Chars->Size))) return Status::Invalid("string offsets run past the column's bytes");
                chars += uint64_t(hi - lo);
            }
            if (chars + 16 > uint64_t(INT32_MAX)) return Status::Unsupported("more than 2 GiB of string bytes in one column");
            col.Values = Buffer::Alloc(size_t(total + 1) * 4 + 16);
            col.Chars = Buffer::Alloc(size_t(chars) + 16);
            if (!col.Values || !col.Chars) return Status::Hip("out of device memory");
        } else if (type == T_BOOL) {
            // (the parts' bitmaps start at any bit: their rows go through a byte each and are packed behind the copies)
            boolBytes[c] = Buffer::Alloc(size_t(total) + 16);
            col.Values = Buffer::Alloc(size_t((total + 7) / 8) + 16);
            if (!boolBytes[c] || !col.Values) return Status::Hip("out of device memory");
        } else {
            col.Values = Buffer::Alloc(size_t(total) * size_t(TypeWidth(type)) + 16);
            if (!col.Values) return Status::Hip("out of device memory");
        }
        int64_t row = 0;
        uint64_t charsAt = 0;
        for (size_t pi = 0; pi < parts.size(); ++pi) {
            const Batch& p = *parts[pi];
            const Column& src = p.Cols[c];
            const int64_t n = p.Rows;
            if (n == 0) continue;
            if (IsStringLike(type)) {
                // the part's bytes in use follow those of the parts before, its offsets move by the difference; every part
                // writes its end offset too (the next part's first offset is the same number)
                const int64_t lo = bounds[boundAt], hi = bounds[boundAt + 1];
                boundAt += 2;
                segs.push_back(CopySegment{static_cast<const uint8_t*>(src.ValuesPtr()), static_cast<uint8_t*>(col.Values->Ptr) + size_t(row) * 4,
                                           uint64_t(n + 1), SEG_I32_ADD, int64_t(charsAt) - lo});
                if (hi > lo) {
                    segs.push_back(CopySegment{static_cast<const uint8_t*>(src.Chars->Ptr) + lo, static_cast<uint8_t*>(col.Chars->Ptr) + charsAt, uint64_t(hi - lo), SEG_BYTES, 0});
                    charsAt += uint64_t(hi - lo);
                }
            } else if (type == T_BOOL) {
                if (!src.Values) return Status::Invalid("boolean column without values");
                segs.push_back(CopySegment{static_cast<const uint8_t*>(src.Values->Ptr), static_cast<uint8_t*>(boolBytes[c]->Ptr) + row, uint64_t(n),
                                           SEG_BITS_TO_BYTES, int64_t(src.ValBitOff)});
            } else {
                const size_t w = size_t(TypeWidth(type));
                segs.push_back(CopySegment{static_cast<const uint8_t*>(src.ValuesPtr()), static_cast<uint8_t*>(col.Values->Ptr) + size_t(row) * w, uint64_t(n) * w, SEG_BYTES, 0});
            }
            if (validBytes[c]) {
                uint8_t* dst = static_cast<uint8_t*>(validBytes[c]->Ptr) + row;
                if (src.Validity) segs.push_back(CopySegment{static_cast<const uint8_t*>(src.Validity->Ptr), dst, uint64_t(n), SEG_BITS_TO_BYTES, int64_t(src.ValidBitOff)});
                else segs.push_back(CopySegment{nullptr, dst, uint64_t(n), SEG_FILL, 1});
            }
            row += n;
        }
        if (IsStringLike(type) && total == 0) SSA_HIP(hipMemsetAsync(col.Values->Ptr, 0, 4, StreamHandle));
        res->Cols.push_back(std::move(col));
    }
    SSA_RETURN_NOT_OK(RunSegments(segs));
    BufferPtr unused;
    for (size_t c = 0; c < res->Cols.size(); ++c) {
        if (boolBytes[c] && total > 0) {
            if (!unused) {
                unused = Buffer::Alloc(256);
                if (!unused) return Status::Hip("out of device memory");
            }
            SSA_HIP(LaunchPackBytesToBits(static_cast<const uint8_t*>(boolBytes[c]->Ptr), total, static_cast<uint8_t*>(res->Cols[c].Values->Ptr),
                                          static_cast<unsigned long long*>(unused->Ptr), StreamHandle));
        }
        if (validBytes[c]) SSA_RETURN_NOT_OK(PackValidity(validBytes[c], total, res->Cols[c]));
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));  // (boolBytes are read until here)
    out = res;
    return Status::OK();
}

}  // namespace ssa
