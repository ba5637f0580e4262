#include "ipc.h"

#include <climits>
#include <cstring>

namespace ssa {

#define SSA_RETURN_NOT_OK_IPC(expr) \
    do {                            \
        Status s_ = (expr);         \
        if (!s_.ok()) return s_;    \
    } while (0)

namespace {

// Bounds-checked reader of a flatbuffer (https://flatbuffers.dev/internals): tables find their fields through a
// vtable of 16-bit offsets, strings / vectors / sub-tables hang off 32-bit forward offsets.  Every access is checked
// against the buffer: blobs come from storage and may be damaged.
class Fb {
public:
    Fb(const uint8_t* p, size_t n) : P(p), N(n) {}
    bool Ok() const { return !Bad; }

    template <class T>
    T Read(size_t pos) {
        T v{};
        if (Bad || pos > N || N - pos < sizeof(T)) {
            Bad = true;
            return v;
        }
        std::memcpy(&v, P + pos, sizeof(T));
        return v;
    }
    size_t Root() { return Indirect(0); }
    // position of field `id` of the table at `table`, 0 when the field is absent
    size_t Field(size_t table, int id) {
        const int32_t soff = Read<int32_t>(table);
        const int64_t vt = int64_t(table) - soff;
        if (vt < 0 || size_t(vt) > N) {
            Bad = true;
            return 0;
        }
        const uint16_t vtSize = Read<uint16_t>(size_t(vt));
        const size_t slot = 4 + 2 * size_t(id);
        if (slot + 2 > vtSize) return 0;
        const uint16_t off = Read<uint16_t>(size_t(vt) + slot);
        return off ? table + off : 0;
    }
    template <class T>
    T Scalar(size_t table, int id, T def) {
        const size_t pos = Field(table, id);
        return pos ? Read<T>(pos) : def;
    }
    // target of the offset stored at `pos`
    size_t Indirect(size_t pos) {
        const uint32_t off = Read<uint32_t>(pos);
        const size_t target = pos + off;
        if (target > N || off == 0) {
            Bad = true;
            return 0;
        }
        return target;
    }
    size_t Table(size_t table, int id) {
        const size_t pos = Field(table, id);
        return pos ? Indirect(pos) : 0;
    }
    // vector field: position of element 0 and the element count
    size_t Vector(size_t table, int id, uint32_t& count) {
        count = 0;
        const size_t pos = Field(table, id);
        if (!pos) return 0;
        const size_t v = Indirect(pos);
        if (!v) return 0;
        count = Read<uint32_t>(v);
        return v + 4;
    }
    std::string String(size_t table, int id) {
        uint32_t n = 0;
        const size_t pos = Vector(table, id, n);
        if (!pos || Bad || n > N - pos) {
            if (pos) Bad = true;
            return std::string();
        }
        return std::string(reinterpret_cast<const char*>(P + pos), n);
    }
    void Fail() { Bad = true; }

private:
    const uint8_t* P;
    size_t N;
    bool Bad = false;
};

// org.apache.arrow.flatbuf.Type (format/Schema.fbs)
enum FbType : uint8_t { FB_Int = 2, FB_FloatingPoint = 3, FB_Binary = 4, FB_Utf8 = 5, FB_Bool = 6, FB_Decimal = 7, FB_Timestamp = 10, FB_Duration = 18 };
enum FbHeader : uint8_t { FB_Schema = 1, FB_DictionaryBatch = 2, FB_RecordBatch = 3 };

Status IntType(Fb& fb, size_t type, int32_t& out) {
    const int32_t bits = type ? fb.Scalar<int32_t>(type, 0, 0) : 0;
    const bool sgn = type ? fb.Scalar<uint8_t>(type, 1, 0) != 0 : false;
    switch (bits) {
        case 8: out = sgn ? T_I8 : T_U8; break;
        case 16: out = sgn ? T_I16 : T_U16; break;
        case 32: out = sgn ? T_I32 : T_U32; break;
        case 64: out = sgn ? T_I64 : T_U64; break;
        default: return Status::Unsupported("integer column of " + std::to_string(bits) + " bits");
    }
    return Status::OK();
}

// `dictId` / `indexType`: the field's DictionaryEncoding {id, indexType (absent: int32), isOrdered, dictionaryKind}, -1
// for a plain column.  The field's own type is then the type of the dictionary's values.
Status ReadField(Fb& fb, size_t field, FieldDesc& out, int64_t& dictId, int32_t& indexType) {
    out = FieldDesc();
    dictId = -1;
    indexType = T_NONE;
    out.Name = fb.String(field, 0);
    out.Nullable = fb.Scalar<uint8_t>(field, 1, 0) != 0;
    const uint8_t tt = fb.Scalar<uint8_t>(field, 2, 0);
    const size_t type = fb.Table(field, 3);
    if (const size_t enc = fb.Table(field, 4)) {
        if (tt != FB_Utf8 && tt != FB_Binary) return Status::Unsupported("dictionary of non-string values in column '" + out.Name + "'");
        if (fb.Scalar<int16_t>(enc, 3, 0) != 0) return Status::Unsupported("dictionary kind of column '" + out.Name + "'");
        dictId = fb.Scalar<int64_t>(enc, 0, 0);
        indexType = T_I32;
        if (const size_t it = fb.Table(enc, 1)) SSA_RETURN_NOT_OK_IPC(IntType(fb, it, indexType));
    }
    uint32_t nchildren = 0;
    (void)fb.Vector(field, 5, nchildren);
    if (nchildren != 0) return Status::Unsupported("nested column '" + out.Name + "'");
    switch (tt) {
        case FB_Int:
            SSA_RETURN_NOT_OK_IPC(IntType(fb, type, out.Type));
            break;
        case FB_FloatingPoint: {
            const int16_t prec = type ? fb.Scalar<int16_t>(type, 0, 0) : 0;
            if (prec == 1) out.Type = T_F32;
            else if (prec == 2) out.Type = T_F64;
            else return Status::Unsupported("half-precision column");
            break;
        }
        case FB_Binary: out.Type = T_BINARY; break;
        case FB_Utf8: out.Type = T_STRING; break;
        case FB_Bool: out.Type = T_BOOL; break;
        case FB_Decimal: {
            out.Type = T_DEC128;
            out.Precision = type ? fb.Scalar<int32_t>(type, 0, 0) : 0;
            out.Scale = type ? fb.Scalar<int32_t>(type, 1, 0) : 0;
            if (type && fb.Scalar<int32_t>(type, 2, 128) != 128) return Status::Unsupported("decimal column that is not 128 bits wide");
            break;
        }
        case FB_Timestamp:
            if (!type || fb.Scalar<int16_t>(type, 0, 0) != 2) return Status::Unsupported("timestamp column whose unit is not microseconds");
            out.Type = T_TS;
            break;
        case FB_Duration:
            if (!type || fb.Scalar<int16_t>(type, 0, 1) != 2) return Status::Unsupported("duration column whose unit is not microseconds");
            out.Type = T_DUR;
            break;
        default:
            return Status::Unsupported("Arrow type " + std::to_string(int(tt)) + " of column '" + out.Name + "'");
    }
    return Status::OK();
}

}  // namespace

Status ParseIpcBlob(const uint8_t* blob, size_t size, const std::vector<FieldDesc>& schema, IpcBatch& out) {
    out = IpcBatch();
    std::vector<FieldDesc> fields = schema;
    std::vector<std::pair<int64_t, int32_t>> encoding(schema.size(), {-1, T_NONE});  // per field: dictionary id, index type
    bool haveSchema = !schema.empty();
    size_t pos = 0;
    for (int guard = 0; guard < 64; ++guard) {
        if (size - pos < 8) return Status::Invalid("IPC blob ends before a RecordBatch message");
        uint32_t marker, metaLen;
        std::memcpy(&marker, blob + pos, 4);
        if (marker == 0xFFFFFFFFu) {
            std::memcpy(&metaLen, blob + pos + 4, 4);
            pos += 8;
        } else {  // the format before the continuation marker was introduced: the length comes first
            metaLen = marker;
            pos += 4;
        }
        if (metaLen == 0) return Status::Invalid("IPC stream ends without a RecordBatch message");
        if (metaLen > size - pos) return Status::Invalid("IPC message metadata runs past the end of the blob");
        Fb fb(blob + pos, metaLen);
        const size_t msg = fb.Root();
        const uint8_t headerType = fb.Scalar<uint8_t>(msg, 1, 0);
        const size_t header = fb.Table(msg, 2);
        const int64_t bodyLength = fb.Scalar<int64_t>(msg, 3, 0);
        if (!fb.Ok() || !header || bodyLength < 0) return Status::Invalid("damaged IPC message metadata");
        const size_t bodyPos = pos + metaLen;
        if (uint64_t(bodyLength) > size - bodyPos) return Status::Invalid("IPC message body runs past the end of the blob");
        if (headerType == FB_Schema) {
            uint32_t n = 0;
            const size_t vec = fb.Vector(header, 1, n);
            if (fb.Scalar<int16_t>(header, 0, 0) != 0) return Status::Unsupported("big-endian IPC stream");
            std::vector<FieldDesc> own;
            std::vector<std::pair<int64_t, int32_t>> ownEncoding;
            for (uint32_t i = 0; i < n; ++i) {
                FieldDesc f;
                int64_t dictId;
                int32_t indexType;
                const size_t field = fb.Indirect(vec + 4 * size_t(i));
                if (!fb.Ok()) return Status::Invalid("damaged IPC schema");
                SSA_RETURN_NOT_OK_IPC(ReadField(fb, field, f, dictId, indexType));
                own.push_back(f);
                ownEncoding.push_back({dictId, indexType});
            }
            if (!fb.Ok()) return Status::Invalid("damaged IPC schema");
            if (haveSchema) {
                if (own.size() != fields.size()) return Status::Invalid("the blob's schema has another number of columns than the one given");
                for (size_t i = 0; i < own.size(); ++i) {
                    if (own[i].Type != fields[i].Type) return Status::Invalid("column '" + own[i].Name + "' has another type than the schema given");
                }
            }
            fields = own;
            encoding = ownEncoding;
            haveSchema = true;
            pos = bodyPos + size_t(bodyLength);
            continue;
        }
        if (headerType == FB_DictionaryBatch) {
            // {id, data: RecordBatch of one column, isDelta}
            IpcDictionary d;
            d.Id = fb.Scalar<int64_t>(header, 0, 0);
            const size_t data = fb.Table(header, 1);
            if (fb.Scalar<uint8_t>(header, 2, 0) != 0) return Status::Unsupported("delta dictionary batch in an IPC blob");
            if (!fb.Ok() || !data) return Status::Invalid("damaged DictionaryBatch metadata");
            d.Rows = fb.Scalar<int64_t>(data, 0, 0);
            uint32_t nnodes = 0, nbuffers = 0;
            const size_t nodes = fb.Vector(data, 1, nnodes);
            const size_t buffers = fb.Vector(data, 2, nbuffers);
            if (const size_t compression = fb.Table(data, 3)) {
                d.Codec = fb.Scalar<int8_t>(compression, 0, 0);
                if (fb.Scalar<int8_t>(compression, 1, 0) != 0) return Status::Unsupported("IPC body compression method other than BUFFER");
                if (d.Codec != IPC_LZ4_FRAME && d.Codec != IPC_ZSTD) return Status::Unsupported("unknown IPC body compression codec");
            }
            if (fb.Ok() && d.Rows > INT32_MAX) return Status::Invalid("a row count of the blob is beyond 2^31 - 1");
            if (!fb.Ok() || d.Rows < 0 || nnodes != 1 || nbuffers != 3) return Status::Unsupported("a dictionary that is not one column of strings");
            d.NullCount = fb.Read<int64_t>(nodes + 8);
            if (fb.Read<int64_t>(nodes) != d.Rows || d.NullCount < 0 || d.NullCount > d.Rows) return Status::Invalid("damaged DictionaryBatch field node");
            d.BodyOffset = bodyPos;
            d.BodyLength = bodyLength;
            IpcBuffer* dst[3] = {&d.Validity, &d.Values, &d.Data};
            for (int k = 0; k < 3; ++k) {
                dst[k]->Offset = fb.Read<int64_t>(buffers + 16 * size_t(k));
                dst[k]->Length = fb.Read<int64_t>(buffers + 16 * size_t(k) + 8);
                if (!fb.Ok() || dst[k]->Offset < 0 || dst[k]->Length < 0 || dst[k]->Offset > bodyLength || dst[k]->Length > bodyLength - dst[k]->Offset) {
                    return Status::Invalid("a DictionaryBatch buffer runs past its body");
                }
            }
            bool replaced = false;
            for (IpcDictionary& have : out.Dicts) {
                if (have.Id == d.Id) {
                    have = d;  // (a later dictionary of the same id replaces the earlier one)
                    replaced = true;
                }
            }
            if (!replaced) out.Dicts.push_back(d);
            pos = bodyPos + size_t(bodyLength);
            continue;
        }
        if (headerType != FB_RecordBatch) return Status::Invalid("unexpected IPC message type " + std::to_string(int(headerType)));
        if (!haveSchema) return Status::Invalid("a RecordBatch message without a schema: give the schema of the column chunk");
        out.Rows = fb.Scalar<int64_t>(header, 0, 0);
        uint32_t nnodes = 0, nbuffers = 0;
        const size_t nodes = fb.Vector(header, 1, nnodes);
        const size_t buffers = fb.Vector(header, 2, nbuffers);
        const size_t compression = fb.Table(header, 3);
        if (compression) {
            out.Codec = fb.Scalar<int8_t>(compression, 0, 0);
            if (fb.Scalar<int8_t>(compression, 1, 0) != 0) return Status::Unsupported("IPC body compression method other than BUFFER");
            if (out.Codec != IPC_LZ4_FRAME && out.Codec != IPC_ZSTD) return Status::Unsupported("unknown IPC body compression codec");
        }
        if (!fb.Ok() || out.Rows < 0) return Status::Invalid("damaged RecordBatch metadata");
        // (buffer sizes are products of this count: near 2^61 they would wrap to a few bytes; no batch here has 2^31 rows)
        if (out.Rows > INT32_MAX) return Status::Invalid("a row count of the blob is beyond 2^31 - 1");
        if (nnodes != fields.size()) return Status::Invalid("the RecordBatch has " + std::to_string(nnodes) + " columns, the schema " + std::to_string(fields.size()));
        out.BodyOffset = bodyPos;
        out.BodyLength = bodyLength;
        uint32_t b = 0;
        auto nextBuffer = [&](IpcBuffer& dst) -> bool {
            if (b >= nbuffers) return false;
            dst.Offset = fb.Read<int64_t>(buffers + 16 * size_t(b));
            dst.Length = fb.Read<int64_t>(buffers + 16 * size_t(b) + 8);
            ++b;
            return fb.Ok() && dst.Offset >= 0 && dst.Length >= 0 && dst.Offset <= bodyLength && dst.Length <= bodyLength - dst.Offset;
        };
        for (uint32_t i = 0; i < nnodes; ++i) {
            IpcColumn col;
            col.Field = fields[i];
            col.Length = fb.Read<int64_t>(nodes + 16 * size_t(i));
            col.NullCount = fb.Read<int64_t>(nodes + 16 * size_t(i) + 8);
            if (!fb.Ok() || col.Length != out.Rows || col.NullCount < 0 || col.NullCount > col.Length) return Status::Invalid("damaged RecordBatch field node");
            if (!nextBuffer(col.Validity) || !nextBuffer(col.Values)) return Status::Invalid("the RecordBatch lists too few buffers");
            col.DictId = encoding[i].first;
            col.IndexType = encoding[i].second;
            if (col.DictId >= 0) {
                bool have = false;
                for (const IpcDictionary& d : out.Dicts) have = have || d.Id == col.DictId;
                if (!have) return Status::Invalid("column '" + col.Field.Name + "' refers to a dictionary the blob does not carry");
            } else if (IsStringLike(col.Field.Type) && !nextBuffer(col.Data)) {
                return Status::Invalid("the RecordBatch lists too few buffers");
            }
            out.Columns.push_back(col);
        }
        if (b != nbuffers) return Status::Invalid("the RecordBatch lists more buffers than its columns have");
        return Status::OK();
    }
    return Status::Invalid("too many messages before the RecordBatch");
}

Status ParseLz4Frame(const uint8_t* blob, size_t size, size_t offset, size_t length, std::vector<Lz4BlockRef>& blocks, uint32_t& blockMax,
                     bool& linked) {
    blocks.clear();
    if (offset > size || length > size - offset) return Status::Invalid("LZ4 frame outside the blob");
    const uint8_t* p = blob + offset;
    size_t pos = 0;
    auto need = [&](size_t n) { return length - pos >= n; };
    if (!need(7)) return Status::Invalid("LZ4 frame too short");
    uint32_t magic;
    std::memcpy(&magic, p, 4);
    if (magic != 0x184D2204u) return Status::Invalid("not an LZ4 frame");
    const uint8_t flg = p[4], bd = p[5];
    pos = 6;
    if ((flg >> 6) != 1) return Status::Invalid("unknown LZ4 frame version");
    linked = !((flg >> 5) & 1);
    const bool blockChecksum = (flg >> 4) & 1, contentSize = (flg >> 3) & 1, dictId = flg & 1;
    const int sizeId = (bd >> 4) & 7;
    if (sizeId < 4) return Status::Invalid("bad LZ4 block size");
    blockMax = 1u << (8 + 2 * sizeId);  // 4: 64 KB, 5: 256 KB, 6: 1 MB, 7: 4 MB
    if (contentSize) pos += 8;
    if (dictId) return Status::Unsupported("LZ4 frame with a dictionary");
    pos += 1;  // header checksum
    if (pos > length) return Status::Invalid("LZ4 frame header runs past the buffer");
    for (;;) {
        if (!need(4)) return Status::Invalid("LZ4 frame without an end mark");
        uint32_t word;
        std::memcpy(&word, p + pos, 4);
        pos += 4;
        if (word == 0) break;
        const uint32_t n = word & 0x7FFFFFFFu;
        if (!need(n) || n > blockMax + 0u) return Status::Invalid("LZ4 block runs past the buffer");
        blocks.push_back(Lz4BlockRef{uint64_t(offset + pos), n, word >> 31});
        pos += n;
        if (blockChecksum) {
            if (!need(4)) return Status::Invalid("LZ4 block checksum runs past the buffer");
            pos += 4;
        }
    }
    return Status::OK();
}

}  // namespace ssa
