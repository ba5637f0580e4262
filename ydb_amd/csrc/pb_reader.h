// Minimal protobuf wire-format reader (varint / fixed32 / fixed64 / length-delimited).
// Enough to walk NKikimrSSA::TProgram messages without generated code.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>

namespace ssa {

class PbReader {
public:
    PbReader(const uint8_t* p, size_t n) : Cur(p), End(p + n) {}
    explicit PbReader(std::string_view s)
        : Cur(reinterpret_cast<const uint8_t*>(s.data())), End(Cur + s.size()) {}

    bool Ok() const { return !Bad; }
    bool Done() const { return Bad || Cur >= End; }

    // Reads the next field header; returns false at end of message or on malformed input.
    bool Next() {
        if (Done()) return false;
        uint64_t key;
        if (!Varint(key)) return false;
        FieldNo = static_cast<uint32_t>(key >> 3);
        Wire = static_cast<uint32_t>(key & 7);
        switch (Wire) {
            case 0:
                if (!Varint(Value)) return false;
                break;
            case 1:
                if (End - Cur < 8) return Fail();
                std::memcpy(&Value, Cur, 8);
                Cur += 8;
                break;
            case 5: {
                if (End - Cur < 4) return Fail();
                uint32_t v;
                std::memcpy(&v, Cur, 4);
                Value = v;
                Cur += 4;
                break;
            }
            case 2: {
                uint64_t len;
                if (!Varint(len)) return false;
                if (static_cast<uint64_t>(End - Cur) < len) return Fail();
                Bytes = std::string_view(reinterpret_cast<const char*>(Cur), len);
                Cur += len;
                break;
            }
            default:
                return Fail();  // groups are not used by ssa.proto
        }
        return true;
    }

    uint32_t FieldNo = 0;
    uint32_t Wire = 0;
    uint64_t Value = 0;        // wire types 0, 1, 5
    std::string_view Bytes;    // wire type 2

    int32_t AsInt32() const { return static_cast<int32_t>(static_cast<int64_t>(Value)); }
    int64_t AsInt64() const { return static_cast<int64_t>(Value); }
    float AsFloat() const {
        uint32_t v = static_cast<uint32_t>(Value);
        float f;
        std::memcpy(&f, &v, 4);
        return f;
    }
    double AsDouble() const {
        double d;
        std::memcpy(&d, &Value, 8);
        return d;
    }

private:
    bool Fail() {
        Bad = true;
        return false;
    }
    bool Varint(uint64_t& out) {
        out = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (Cur >= End) return Fail();
            uint8_t b = *Cur++;
            out |= static_cast<uint64_t>(b & 0x7F) << shift;
            if (!(b & 0x80)) return true;
        }
        return Fail();
    }

    const uint8_t* Cur;
    const uint8_t* End;
    bool Bad = false;
};

}  // namespace ssa
