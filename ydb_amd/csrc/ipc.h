// Arrow IPC messages as the column shard stores and ships them: the layout half of
// NArrow::NSerialization::TNativeSerializer::DoDeserialize (ydb/core/formats/arrow/serializer/native.cpp:18-81).
//
// A blob is either a whole IPC stream (Schema message, DictionaryBatch messages for dictionary-encoded columns --
// what the dictionary transformer of ydb/core/formats/arrow/transformer/dictionary.cpp:5-7 produces --, RecordBatch
// message, end marker: DoSerializeFull, :42-58) or a
// lone RecordBatch message whose schema the reader already has (DoSerializePayload, :83-103).  Each message is
// [0xFFFFFFFF][int32 metadata size][flatbuffer Message][padding][body]; the RecordBatch metadata lists one FieldNode
// {length, null_count} per column and one {offset, length} per buffer of the body, and may name a body compression
// (LZ4_FRAME by default in the reference, native.h:54-57): then every non-empty buffer is an int64 uncompressed
// length (-1: stored as is) followed by the compressed bytes
// (contrib/libs/apache/arrow/cpp/src/arrow/ipc/reader.cc, DecompressBuffers).
//
// This file only reads metadata on the host; the bytes are decoded on the device (lz4_kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "kernels.h"
#include "plan.h"

namespace ssa {

enum IpcCodec : int32_t { IPC_RAW = -1, IPC_LZ4_FRAME = 0, IPC_ZSTD = 1 };

struct IpcBuffer {
    int64_t Offset = 0;  // from the start of the message body
    int64_t Length = 0;  // bytes in the body (compressed form when the batch is compressed)
};

struct IpcColumn {
    FieldDesc Field;         // name, type, decimal parameters; Nullable = the schema's flag
    int64_t Length = 0;
    int64_t NullCount = 0;
    IpcBuffer Validity, Values, Data;  // Values: fixed-width values, boolean bits or int32 offsets; Data: string bytes
    // A dictionary-encoded column (the schema's field carries a DictionaryEncoding): Field.Type is the type of the
    // dictionary's values, Values holds the codes of IndexType, there is no Data buffer; the values come with the
    // DictionaryBatch message of the same id (IpcBatch::Dicts).
    int64_t DictId = -1;
    int32_t IndexType = T_NONE;
};

// A DictionaryBatch message: one column of values (strings here) with a body of its own.
struct IpcDictionary {
    int64_t Id = 0;
    int64_t Rows = 0;
    int64_t NullCount = 0;
    int32_t Codec = IPC_RAW;
    size_t BodyOffset = 0;
    int64_t BodyLength = 0;
    IpcBuffer Validity, Values, Data;  // Values: int32 offsets
};

struct IpcBatch {
    int64_t Rows = 0;
    int32_t Codec = IPC_RAW;
    size_t BodyOffset = 0;   // of the RecordBatch message's body inside the blob
    int64_t BodyLength = 0;
    std::vector<IpcColumn> Columns;
    std::vector<IpcDictionary> Dicts;  // the dictionaries the columns refer to, one per id (the last one sent)
};

// Parses `blob`.  `schema` (may be empty) is used when the blob is a lone RecordBatch message; a blob that carries its
// own Schema message must agree with a non-empty `schema` in column count and types.
Status ParseIpcBlob(const uint8_t* blob, size_t size, const std::vector<FieldDesc>& schema, IpcBatch& out);

// The blocks of the frame at [offset, offset + length) of the blob.  blockMax = the frame's maximal block size (every
// block but the last decodes to exactly that many bytes when the frame was written in one call, as Arrow's codec does).
Status ParseLz4Frame(const uint8_t* blob, size_t size, size_t offset, size_t length, std::vector<Lz4BlockRef>& blocks, uint32_t& blockMax,
                     bool& linked);

}  // namespace ssa
