#include "exec.h"
#include "jit.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <climits>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <set>

#include "vm_ops.h"

namespace ssa {

#define SSA_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {   \
            (void)hipGetLastError(); /* the runtime keeps the error for the next caller otherwise */   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));        \
        }   \
    } while (0)

#define SSA_RETURN_NOT_OK(expr)     \
    do {                            \
        Status s_ = (expr);         \
        if (!s_.ok()) return s_;    \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device memory
// ---------------------------------------------------------------------------------------------
namespace {
size_t BucketSize(size_t bytes) {
    size_t b = 256;
    while (b < bytes) b <<= 1;
    if (b > (size_t(1) << 26)) b = (bytes + (size_t(1) << 26) - 1) & ~((size_t(1) << 26) - 1);  // 64 MiB steps above 64 MiB
    return b;
}
}  // namespace

DevPool& DevPool::Instance() {
    static DevPool pool;
    return pool;
}

// Free blocks are kept per device: a block handed out again must live on the device of the caller (the one current
// on the calling thread -- every executor entry point sets its own).
void* DevPool::Alloc(size_t bytes, int* device) {
    const size_t b = BucketSize(bytes ? bytes : 1);
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (device) *device = dev;
    {
        std::lock_guard<std::mutex> g(Lock);
        auto it = FreeLists.find({dev, b});
        if (it != FreeLists.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            InUse += b;
            Cached -= b;
            return p;
        }
    }
    // (SSA_TEST_FAIL_ALLOC=n, tests only: the n-th hipMalloc of the process fails once, as if the device were full)
    static std::atomic<long> mallocs{0};
    auto tryMalloc = [&](void** out) {
        const long nth = ++mallocs;
        const char* fail = std::getenv("SSA_TEST_FAIL_ALLOC");
        hipError_t e = (fail && std::atol(fail) == nth) ? hipErrorOutOfMemory : hipMalloc(out, b);
        // a failed call leaves its error pending in the runtime: the next launch on this thread would report it
        if (e != hipSuccess) (void)hipGetLastError();
        return e;
    };
    void* p = nullptr;
    if (tryMalloc(&p) != hipSuccess) {
        Trim();
        if (tryMalloc(&p) != hipSuccess) return nullptr;
    }
    std::lock_guard<std::mutex> g(Lock);
    InUse += b;
    return p;
}

void DevPool::Free(void* p, size_t bytes, int device) {
    if (!p) return;
    const size_t b = BucketSize(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> g(Lock);
        InUse -= b;
        // keep at most kMaxCachedBytes of free blocks: what is beyond goes back to the device
        if (Cached + b <= kMaxCachedBytes) {
            FreeLists[{device, b}].push_back(p);
            Cached += b;
            return;
        }
    }
    (void)hipFree(p);  // (hipFree finds the owning device from the address)
}

void DevPool::Trim() {
    std::lock_guard<std::mutex> g(Lock);
    for (auto& kv : FreeLists) {
        for (void* p : kv.second) (void)hipFree(p);
        kv.second.clear();
    }
    Cached = 0;
}

Buffer::~Buffer() {
    if (Owned) DevPool::Instance().Free(Ptr, Size, Device);
}

BufferPtr Buffer::Alloc(size_t bytes) {
    auto b = std::make_shared<Buffer>();
    b->Ptr = DevPool::Instance().Alloc(bytes, &b->Device);
    b->Size = bytes;
    b->Owned = true;
    if (!b->Ptr) return nullptr;
    return b;
}

BufferPtr Buffer::Wrap(void* p, size_t bytes) {
    auto b = std::make_shared<Buffer>();
    b->Ptr = p;
    b->Size = bytes;
    b->Owned = false;
    return b;
}

int Column::StorageType() const { return Dict ? Dict->IndexType : Field.Type; }

const void* Column::ValuesPtr() const {
    if (!Values) return nullptr;
    const int w = Dict ? TypeWidth(Dict->IndexType) : IsStringLike(Field.Type) ? 4 : TypeWidth(Field.Type);
    return static_cast<const char*>(Values->Ptr) + ElemOffset * w;
}

std::vector<FieldDesc> Batch::Schema() const {
    std::vector<FieldDesc> s;
    for (auto& c : Cols) s.push_back(c.Field);
    return s;
}

int Batch::FindColumn(const std::string& name) const {
    int found = -1, count = 0;
    for (size_t i = 0; i < Cols.size(); ++i) {
        if (Cols[i].Field.Name == name) {
            found = int(i);
            ++count;
        }
    }
    return count == 1 ? found : -1;
}

namespace {
struct PhaseMark {
    const char* name;
    std::chrono::steady_clock::time_point at;
};
thread_local std::vector<PhaseMark> tPhaseMarks;
}  // namespace

bool HostPhases::On() {
    static const bool on = std::getenv("SSA_HOST_PHASES") != nullptr;
    return on;
}

void HostPhases::Mark(const char* name) {
    if (On()) tPhaseMarks.push_back(PhaseMark{name, std::chrono::steady_clock::now()});
}

void HostPhases::Report(const char* what) {
    if (!On() || tPhaseMarks.empty()) return;
    std::string line = what;
    for (size_t i = 1; i < tPhaseMarks.size(); ++i) {
        char buf[96];
        std::snprintf(buf, sizeof(buf), " %s %.1f", tPhaseMarks[i].name, std::chrono::duration<double, std::micro>(tPhaseMarks[i].at - tPhaseMarks[i - 1].at).count());
        line += buf;
    }
    std::fprintf(stderr, "%s (us)\n", line.c_str());
    tPhaseMarks.clear();
}

// ---------------------------------------------------------------------------------------------
// executor basics
// ---------------------------------------------------------------------------------------------
Executor::Executor(int device) : Device(device) {
    (void)hipSetDevice(device);
    (void)hipStreamCreateWithFlags(&StreamHandle, hipStreamNonBlocking);
    (void)hipEventCreate(&EvStart);
    (void)hipEventCreate(&EvTuneA);
    (void)hipEventCreate(&EvTuneB);
    (void)hipEventCreate(&EvStop);
    // dynamic LDS the scan kernels may ask for: everything but 1 KiB kept for their static arrays
    hipError_t cfgErr = ConfigureScanKernels(160 * 1024 - 1024);
    if (cfgErr == hipSuccess) cfgErr = ConfigureRegionKernels(128 * 1024);
    if (cfgErr != hipSuccess) std::fprintf(stderr, "ydb_ssa_hip: cannot raise the LDS limit of the scan kernels: %s\n", hipGetErrorString(cfgErr));
    StatusWord = Buffer::Alloc(256);
    if (StatusWord) (void)hipMemset(StatusWord->Ptr, 0, 256);
    void* pinned = nullptr;
    if (hipHostMalloc(&pinned, kHostWords * sizeof(int32_t), hipHostMallocDefault) == hipSuccess) {
        HostWords = static_cast<int32_t*>(pinned);
    } else {
        (void)hipGetLastError();  // (the pageable paths below still work)
    }
    hipError_t scratchErr = hipSuccess;
    NoAllocScratch = NoAllocScratchFor(device, &scratchErr);
    if (!NoAllocScratch) std::fprintf(stderr, "ydb_ssa_hip: no static scratch buffer on device %d: %s\n", device, hipGetErrorString(scratchErr));
}

Executor::~Executor() {
    if (StreamHandle) (void)hipStreamSynchronize(StreamHandle);
    if (EvStart) (void)hipEventDestroy(EvStart);
    if (EvStop) (void)hipEventDestroy(EvStop);
    if (EvTuneA) (void)hipEventDestroy(EvTuneA);
    if (EvTuneB) (void)hipEventDestroy(EvTuneB);
    if (StreamHandle) (void)hipStreamDestroy(StreamHandle);
    if (HostWords) (void)hipHostFree(HostWords);
    if (HostStaging) (void)hipHostFree(HostStaging);
    if (BlobStaging) {
        (void)hipHostFree(BlobStaging);
        for (hipEvent_t e : BlobSlotDone) (void)hipEventDestroy(e);
    }
}

bool Executor::EnsureHostStaging() {
    if (HostStaging) return true;
    void* pinned = nullptr;
    if (hipHostMalloc(&pinned, kHostStagingBytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    HostStaging = static_cast<uint8_t*>(pinned);
    return true;
}

Status Executor::Sync() {
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

void Executor::ProfileBegin() {
    if (!Profile.Enabled) return;
    ProfileCollect();
    (void)hipEventRecord(EvStart, StreamHandle);
}

// The launch's time is read later (ProfileCollect: at the next launch to be timed, or when the profile is read): waiting
// for the stop event here would keep the work that follows the kernel from being queued behind it.
void Executor::ProfileEnd(const char* kernel, int64_t algoBytes) {
    if (!Profile.Enabled) return;
    (void)hipEventRecord(EvStop, StreamHandle);
    ProfilePending = true;
    ProfilePendingBytes = algoBytes;
    Profile.LastKernel = kernel;
}

void Executor::ProfileCollect() {
    if (!ProfilePending) return;
    ProfilePending = false;
    (void)hipSetDevice(Device);
    (void)hipEventSynchronize(EvStop);
    float ms = 0;
    if (hipEventElapsedTime(&ms, EvStart, EvStop) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    Profile.ScanMs += ms;
    Profile.ScanLaunches += 1;
    Profile.ScanBytes += ProfilePendingBytes;
}

Status Executor::CheckScanStatus(bool* tableFull, bool* notUniform) {
    uint32_t st = 0;
    if (HostWords) {
        SSA_HIP(hipMemcpyAsync(HostWords, StatusWord->Ptr, sizeof(st), hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        st = uint32_t(HostWords[0]);
    } else {
        SSA_HIP(hipMemcpyAsync(&st, StatusWord->Ptr, sizeof(st), hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    return DecodeScanStatus(st, tableFull, notUniform);
}

// What a scan's status word (already on the host) says; a status other than zero is cleared on the device.
Status Executor::DecodeScanStatus(uint32_t st, bool* tableFull, bool* notUniform) {
    if (tableFull) *tableFull = false;
    if (notUniform) *notUniform = false;
    if (st == 0) return Status::OK();
    SSA_HIP(hipMemsetAsync(StatusWord->Ptr, 0, sizeof(st), StreamHandle));
    if ((st & SCAN_NOT_UNIFORM) && notUniform) {  // the scan's results are void: the caller runs the step another way
        *notUniform = true;
        return Status::OK();
    }
    if (st & SCAN_DIV_BY_ZERO) return Status::Invalid("divide by zero");
    if (st & SCAN_CAST_OVERFLOW) return Status::Invalid("Integer value not in range");
    if (st & SCAN_OVERFLOW) return Status::Invalid("overflow");
    if (st & SCAN_NAN_ORDER) {
        return Status::Unsupported("MIN / MAX of floating-point values with a NaN among them under a single one-byte GROUP BY key: the "
                                   "result depends on the order the reference's lookup-table path visits the rows in");
    }
    if (st == SCAN_TABLE_FULL && tableFull) {
        *tableFull = true;
        return Status::OK();
    }
    return Status::Invalid("scan kernel reported status " + std::to_string(st));
}

// ---------------------------------------------------------------------------------------------
// Arrow C Data Interface: import (host -> HBM)
// ---------------------------------------------------------------------------------------------
namespace {

bool ParseFormat(const char* fmt, int32_t& type, int32_t& prec, int32_t& scale) {
    prec = scale = 0;
    const std::string f(fmt ? fmt : "");
    if (f == "b") type = T_BOOL;
    else if (f == "c") type = T_I8;
    else if (f == "C") type = T_U8;
    else if (f == "s") type = T_I16;
    else if (f == "S") type = T_U16;
    else if (f == "i") type = T_I32;
    else if (f == "I") type = T_U32;
    else if (f == "l") type = T_I64;
    else if (f == "L") type = T_U64;
    else if (f == "f") type = T_F32;
    else if (f == "g") type = T_F64;
    else if (f == "u") type = T_STRING;
    else if (f == "z") type = T_BINARY;
    else if (f.rfind("tsu:", 0) == 0) type = T_TS;
    else if (f == "tDu") type = T_DUR;
    else if (f.rfind("d:", 0) == 0) {
        type = T_DEC128;
        int bits = 128;
        if (sscanf(f.c_str(), "d:%d,%d,%d", &prec, &scale, &bits) < 2) return false;
        if (bits != 128) return false;
    } else {
        return false;
    }
    return true;
}

std::string FormatOf(const FieldDesc& f) {
    switch (f.Type) {
        case T_BOOL: return "b";
        case T_I8: return "c";
        case T_U8: return "C";
        case T_I16: return "s";
        case T_U16: return "S";
        case T_I32: return "i";
        case T_U32: return "I";
        case T_I64: return "l";
        case T_U64: return "L";
        case T_F32: return "f";
        case T_F64: return "g";
        case T_STRING: return "u";
        case T_BINARY: return "z";
        case T_TS: return "tsu:";
        case T_DUR: return "tDu";
        case T_DEC128: return "d:" + std::to_string(f.Precision) + "," + std::to_string(f.Scale);
        default: return "n";
    }
}

// Copy `nbits` bits starting at bit `off` of `src` into a fresh bitmap that starts at bit 0.
std::vector<uint8_t> RealignBits(const uint8_t* src, int64_t off, int64_t nbits) {
    std::vector<uint8_t> out(size_t((nbits + 7) / 8), 0);
    if ((off & 7) == 0) {
        std::memcpy(out.data(), src + (off >> 3), out.size());
    } else {
        for (int64_t i = 0; i < nbits; ++i) {
            const int64_t b = off + i;
            if ((src[b >> 3] >> (b & 7)) & 1) out[size_t(i >> 3)] |= uint8_t(1u << (i & 7));
        }
    }
    if (nbits & 7) out.back() &= uint8_t((1u << (nbits & 7)) - 1);
    return out;
}

int64_t CountSetBits(const std::vector<uint8_t>& bits) {
    int64_t c = 0;
    for (uint8_t b : bits) c += __builtin_popcount(b);
    return c;
}

}  // namespace

Status SchemaFields(const ArrowSchema* schema, std::vector<FieldDesc>& out) {
    if (!schema || !schema->format || std::string(schema->format) != "+s") return Status::Invalid("expected a struct schema");
    for (int64_t i = 0; i < schema->n_children; ++i) {
        const ArrowSchema* s = schema->children[i];
        FieldDesc f;
        f.Name = s->name ? s->name : "";
        if (!ParseFormat(s->dictionary ? s->dictionary->format : s->format, f.Type, f.Precision, f.Scale)) {
            return Status::Unsupported(std::string("Arrow format '") + (s->format ? s->format : "") + "'");
        }
        f.Nullable = (s->flags & ARROW_FLAG_NULLABLE) != 0;
        out.push_back(f);
    }
    return Status::OK();
}

Status Executor::ImportBatch(const ArrowArray* array, const ArrowSchema* schema, BatchPtr& out) {
    if (!array || !schema || !schema->format) return Status::Invalid("null ArrowArray / ArrowSchema");
    if (std::string(schema->format) != "+s") return Status::Invalid("expected a struct array (an exported RecordBatch)");
    if (array->n_children != schema->n_children) return Status::Invalid("schema / array children mismatch");
    if (array->offset != 0) return Status::Unsupported("sliced top-level struct array");
    (void)hipSetDevice(Device);
    auto batch = std::make_shared<Batch>();
    batch->Rows = array->length;
    for (int64_t ci = 0; ci < array->n_children; ++ci) {
        const ArrowArray* a = array->children[ci];
        const ArrowSchema* s = schema->children[ci];
        Column col;
        col.Field.Name = s->name ? s->name : "";
        if (a->length != array->length) return Status::Invalid("column '" + col.Field.Name + "' has a different length");
        const ArrowArray* dictArr = a->dictionary;
        const ArrowSchema* dictSchema = s->dictionary;
        int32_t type, prec, scale;
        if (!ParseFormat(dictSchema ? dictSchema->format : s->format, type, prec, scale)) {
            return Status::Unsupported(std::string("Arrow format '") + (s->format ? s->format : "") + "' of column '" +
                                       col.Field.Name + "'");
        }
        col.Field.Type = type;
        col.Field.Precision = prec;
        col.Field.Scale = scale;
        col.Length = a->length;
        const int64_t n = a->length, off = a->offset;

        // validity
        const uint8_t* validity = (a->n_buffers > 0) ? static_cast<const uint8_t*>(a->buffers[0]) : nullptr;
        std::vector<uint8_t> validBits;
        if (validity && a->null_count != 0) {
            validBits = RealignBits(validity, off, n);
            col.NullCount = n - CountSetBits(validBits);
        } else {
            col.NullCount = 0;
        }

        if (dictArr) {
            // Dictionary-encoded strings stay encoded in HBM: the codes and, once, the dictionary.  The reference's executor
            // only ever sees the decoded column (ydb/core/formats/arrow/dictionary/conversion.h:13-15); steps over the
            // codes produce that column's results (dict.cpp).
            if (!IsStringLike(type)) return Status::Unsupported("dictionary of non-string values");
            int32_t itype, p2, s2;
            if (!ParseFormat(s->format, itype, p2, s2) || !IsInt(itype)) return Status::Unsupported("dictionary index type");
            if (dictArr->n_buffers < 3) return Status::Invalid("dictionary without offsets / data buffers");
            if (dictArr->null_count > 0) return Status::Unsupported("dictionary with null entries");
            const int iw = TypeWidth(itype);
            const char* idx = static_cast<const char*>(a->buffers[1]) + off * iw;
            const int64_t m = dictArr->length;
            for (int64_t i = 0; i < n; ++i) {
                const bool valid = validBits.empty() || ((validBits[size_t(i >> 3)] >> (i & 7)) & 1);
                if (!valid) continue;
                int64_t k = 0;
                switch (itype) {
                    case T_I8: k = reinterpret_cast<const int8_t*>(idx)[i]; break;
                    case T_U8: k = reinterpret_cast<const uint8_t*>(idx)[i]; break;
                    case T_I16: k = reinterpret_cast<const int16_t*>(idx)[i]; break;
                    case T_U16: k = reinterpret_cast<const uint16_t*>(idx)[i]; break;
                    case T_I32: k = reinterpret_cast<const int32_t*>(idx)[i]; break;
                    case T_U32: k = reinterpret_cast<const uint32_t*>(idx)[i]; break;
                    case T_I64: k = reinterpret_cast<const int64_t*>(idx)[i]; break;
                    default: k = int64_t(reinterpret_cast<const uint64_t*>(idx)[i]); break;
                }
                if (k < 0 || k >= m) return Status::Invalid("dictionary index out of range");
            }
            auto dict = std::make_shared<DictInfo>();
            dict->IndexType = itype;
            dict->Entries.Field.Type = type;
            dict->Entries.Field.Name = col.Field.Name;
            dict->Entries.Length = m;
            {
                const int32_t* doffs = static_cast<const int32_t*>(dictArr->buffers[1]);
                const char* dchars = static_cast<const char*>(dictArr->buffers[2]);
                std::vector<int32_t> rebased(size_t(m) + 1, 0);
                int32_t first = 0, last = 0;
                if (doffs) {
                    first = doffs[dictArr->offset];
                    last = doffs[dictArr->offset + m];
                    for (int64_t e = 0; e <= m; ++e) rebased[size_t(e)] = doffs[dictArr->offset + e] - first;
                }
                if (last < first) return Status::Invalid("dictionary offsets decrease");
                dict->Entries.Values = Buffer::Alloc((size_t(m) + 1) * 4 + 16);
                dict->Entries.Chars = Buffer::Alloc(size_t(last - first) + 16);
                if (!dict->Entries.Values || !dict->Entries.Chars) return Status::Hip("out of device memory");
                SSA_HIP(hipMemcpy(dict->Entries.Values->Ptr, rebased.data(), rebased.size() * 4, hipMemcpyHostToDevice));
                if (last > first) SSA_HIP(hipMemcpy(dict->Entries.Chars->Ptr, dchars + first, size_t(last - first), hipMemcpyHostToDevice));
            }
            col.Values = Buffer::Alloc(size_t(n) * iw + 16);
            if (!col.Values) return Status::Hip("out of device memory");
            if (n > 0) SSA_HIP(hipMemcpy(col.Values->Ptr, idx, size_t(n) * iw, hipMemcpyHostToDevice));
            col.Dict = dict;
        } else if (IsStringLike(type)) {
            if (a->n_buffers < 3) return Status::Invalid("string column without offsets / data buffers");
            const int32_t* offs = static_cast<const int32_t*>(a->buffers[1]);
            const char* chars = static_cast<const char*>(a->buffers[2]);
            std::vector<int32_t> rebased(size_t(n) + 1, 0);
            int32_t first = 0, last = 0;
            if (offs) {
                first = offs[off];
                last = offs[off + n];
                for (int64_t i = 0; i <= n; ++i) rebased[size_t(i)] = offs[off + i] - first;
            }
            col.Values = Buffer::Alloc((size_t(n) + 1) * 4);
            col.Chars = Buffer::Alloc(size_t(last - first) + 16);
            if (!col.Values || !col.Chars) return Status::Hip("out of device memory");
            SSA_HIP(hipMemcpy(col.Values->Ptr, rebased.data(), rebased.size() * 4, hipMemcpyHostToDevice));
            if (last > first) SSA_HIP(hipMemcpy(col.Chars->Ptr, chars + first, size_t(last - first), hipMemcpyHostToDevice));
        } else if (type == T_BOOL) {
            if (a->n_buffers < 2) return Status::Invalid("boolean column without a data buffer");
            std::vector<uint8_t> bits = RealignBits(static_cast<const uint8_t*>(a->buffers[1]), off, n);
            col.Values = Buffer::Alloc(bits.size() + 16);
            if (!col.Values) return Status::Hip("out of device memory");
            if (!bits.empty()) SSA_HIP(hipMemcpy(col.Values->Ptr, bits.data(), bits.size(), hipMemcpyHostToDevice));
        } else {
            if (a->n_buffers < 2) return Status::Invalid("column without a data buffer");
            const int w = TypeWidth(type);
            col.Values = Buffer::Alloc(size_t(n) * w + 16);
            if (!col.Values) return Status::Hip("out of device memory");
            if (n > 0) {
                SSA_HIP(hipMemcpy(col.Values->Ptr, static_cast<const char*>(a->buffers[1]) + off * w, size_t(n) * w,
                                  hipMemcpyHostToDevice));
            }
        }
        if (!validBits.empty() && col.NullCount != 0) {
            col.Validity = Buffer::Alloc(validBits.size() + 16);
            if (!col.Validity) return Status::Hip("out of device memory");
            SSA_HIP(hipMemcpy(col.Validity->Ptr, validBits.data(), validBits.size(), hipMemcpyHostToDevice));
        }
        // TDatumBatch treats a key as nullable when MayHaveNulls(): a validity buffer with a non-zero null count.
        col.Field.Nullable = (col.Validity != nullptr);
        batch->Cols.push_back(std::move(col));
    }
    out = batch;
    return Status::OK();
}

// ---------------------------------------------------------------------------------------------
// Arrow C Data Interface: export (HBM -> host)
// ---------------------------------------------------------------------------------------------
namespace {

struct ExportedArray {
    std::vector<void*> Owned;          // malloc'd blocks
    std::vector<const void*> Buffers;
    std::vector<ArrowArray*> Children;
};

struct ExportedSchema {
    std::string Format, Name;
    std::vector<ArrowSchema*> Children;
};

void ReleaseArray(ArrowArray* a) {
    if (!a || !a->release) return;
    auto* p = static_cast<ExportedArray*>(a->private_data);
    for (ArrowArray* c : p->Children) {
        if (c->release) c->release(c);
        delete c;
    }
    for (void* m : p->Owned) std::free(m);
    delete p;
    a->release = nullptr;
}

void ReleaseSchema(ArrowSchema* s) {
    if (!s || !s->release) return;
    auto* p = static_cast<ExportedSchema*>(s->private_data);
    for (ArrowSchema* c : p->Children) {
        if (c->release) c->release(c);
        delete c;
    }
    delete p;
    s->release = nullptr;
}

void InitSchema(ArrowSchema* s, const std::string& format, const std::string& name, bool nullable) {
    auto* p = new ExportedSchema();
    p->Format = format;
    p->Name = name;
    std::memset(s, 0, sizeof(*s));
    s->format = p->Format.c_str();
    s->name = p->Name.c_str();
    s->flags = nullable ? ARROW_FLAG_NULLABLE : 0;
    s->release = ReleaseSchema;
    s->private_data = p;
}

}  // namespace

Status Executor::ExportBatch(const Batch& encoded, ArrowArray* outArray, ArrowSchema* outSchema) {
    (void)hipSetDevice(Device);
    // a dictionary column leaves as the plain column it stands for (what the reference's executor would have produced)
    Batch decoded;
    bool anyDict = false;
    for (const Column& c : encoded.Cols) anyDict = anyDict || bool(c.Dict);
    if (anyDict) {
        decoded = encoded;
        for (Column& c : decoded.Cols) {
            if (!c.Dict) continue;
            Column plain;
            SSA_RETURN_NOT_OK(DecodeColumn(c, decoded.Rows, plain));
            c = plain;
        }
    }
    const Batch& batch = anyDict ? decoded : encoded;
    // a result of few rows comes over in one copy (a copy per buffer costs a round trip each: twenty for a Q1 result)
    std::vector<HostColumnView> views;
    bool packed = false;
    SSA_RETURN_NOT_OK(PackToHost(batch, views, packed));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    InitSchema(outSchema, "+s", "", false);
    auto* sp = static_cast<ExportedSchema*>(outSchema->private_data);
    auto* ap = new ExportedArray();
    std::memset(outArray, 0, sizeof(*outArray));
    outArray->length = batch.Rows;
    outArray->release = ReleaseArray;
    outArray->private_data = ap;
    ap->Buffers.push_back(nullptr);

    auto fail = [&](Status st) {
        outArray->n_children = int64_t(ap->Children.size());
        outSchema->n_children = int64_t(sp->Children.size());
        outArray->children = ap->Children.data();
        outSchema->children = sp->Children.data();
        ReleaseArray(outArray);
        ReleaseSchema(outSchema);
        return st;
    };

    for (size_t ci = 0; ci < batch.Cols.size(); ++ci) {
        const Column& col = batch.Cols[ci];
        if (col.Field.IsScalar) return fail(Status::Invalid("internal: scalar column in an exported batch"));
        auto* cs = new ArrowSchema();
        InitSchema(cs, FormatOf(col.Field), col.Field.Name, true);
        sp->Children.push_back(cs);
        auto* ca = new ArrowArray();
        auto* cp = new ExportedArray();
        std::memset(ca, 0, sizeof(*ca));
        ca->length = col.Length;
        ca->release = ReleaseArray;
        ca->private_data = cp;
        ap->Children.push_back(ca);

        const int64_t n = col.Length;
        const HostColumnView* view = packed ? &views[ci] : nullptr;
        // `bytes` of a buffer into host memory of its own: from the packed copy where there is one, else from the device
        auto download = [&](const void* dev, const uint8_t* staged, size_t bytes, void*& host) -> Status {
            host = std::malloc(bytes + 8);
            if (!host) return Status::Invalid("out of host memory");
            cp->Owned.push_back(host);
            if (view) {
                if (bytes && staged) std::memcpy(host, staged, bytes);
            } else if (bytes && dev) {
                SSA_HIP(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
            }
            return Status::OK();
        };
        // validity
        void* vhost = nullptr;
        if (col.Validity) {
            const int bitOff = view ? view->ValidBitOff : col.ValidBitOff;
            const size_t rawBytes = size_t((bitOff + n + 7) / 8);
            void* raw = nullptr;
            Status st = download(col.Validity->Ptr, view ? view->Valid : nullptr, rawBytes, raw);
            if (!st.ok()) return fail(st);
            if (bitOff != 0) {
                std::vector<uint8_t> re = RealignBits(static_cast<uint8_t*>(raw), bitOff, n);
                std::memcpy(raw, re.data(), re.size());
            }
            vhost = raw;
            int64_t set = 0;
            const uint8_t* vb = static_cast<const uint8_t*>(vhost);
            for (int64_t i = 0; i < n; ++i) set += (vb[i >> 3] >> (i & 7)) & 1;
            ca->null_count = n - set;
        } else {
            ca->null_count = 0;
        }
        cp->Buffers.push_back(vhost);
        if (IsStringLike(col.Field.Type)) {
            void* offs = nullptr;
            Status st = download(col.ValuesPtr(), view ? view->Values : nullptr, size_t(n + 1) * 4, offs);
            if (!st.ok()) return fail(st);
            int32_t* o = static_cast<int32_t*>(offs);
            if (n == 0 && !col.Values) o[0] = 0;
            const int32_t first = o[0], last = o[n];  // (the packed offsets start at 0)
            void* chars = nullptr;
            st = download(col.Chars ? static_cast<const char*>(col.Chars->Ptr) + first : nullptr, view ? view->Chars : nullptr, size_t(last - first), chars);
            if (!st.ok()) return fail(st);
            if (first != 0) {
                for (int64_t i = 0; i <= n; ++i) o[i] -= first;
            }
            cp->Buffers.push_back(offs);
            cp->Buffers.push_back(chars);
        } else if (col.Field.Type == T_BOOL) {
            const int bitOff = view ? view->ValBitOff : col.ValBitOff;
            const size_t rawBytes = size_t((bitOff + n + 7) / 8);
            void* raw = nullptr;
            Status st = download(col.Values ? col.Values->Ptr : nullptr, view ? view->Values : nullptr, rawBytes, raw);
            if (!st.ok()) return fail(st);
            if (bitOff != 0) {
                std::vector<uint8_t> re = RealignBits(static_cast<uint8_t*>(raw), bitOff, n);
                std::memcpy(raw, re.data(), re.size());
            }
            cp->Buffers.push_back(raw);
        } else {
            void* vals = nullptr;
            Status st = download(col.ValuesPtr(), view ? view->Values : nullptr, size_t(n) * TypeWidth(col.Field.Type), vals);
            if (!st.ok()) return fail(st);
            cp->Buffers.push_back(vals);
        }
        ca->n_buffers = int64_t(cp->Buffers.size());
        ca->buffers = cp->Buffers.data();
    }
    outArray->n_buffers = 1;
    outArray->buffers = ap->Buffers.data();
    outArray->n_children = int64_t(ap->Children.size());
    outArray->children = ap->Children.data();
    outSchema->n_children = int64_t(sp->Children.size());
    outSchema->children = sp->Children.data();
    return Status::OK();
}

// ---------------------------------------------------------------------------------------------
// scan argument block
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kLdsPerCU = 160 * 1024;
constexpr int kLdsStageBudget = 64 * 1024;  // staged column tiles per workgroup (keeps >= 2 workgroups per CU)
}  // namespace

// (FetchKind, payload) for an operand given as an R register (>= 0) or a column slot (<= -2).
void EncodeOperand(const ScanArgs& args, int reg, uint32_t& fk, uint32_t& payload) {
    if (reg >= 0) {
        fk = FK_REG;
        payload = uint32_t(reg);
        return;
    }
    const int c = -2 - reg;
    const DevColumn& col = args.cols[c];
    const int off = args.lds_col_off[c];
    fk = FK_SLOW;
    payload = uint32_t(c) << 24;
    if (off >= 0) {
        switch (col.type) {
            case T_I64: case T_U64: case T_F64: case T_TS: case T_DUR: fk = FK_LDS_64; break;
            case T_I32: fk = FK_LDS_I32; break;
            case T_U32: fk = FK_LDS_U32; break;
            case T_I16: fk = FK_LDS_I16; break;
            case T_U16: fk = FK_LDS_U16; break;
            case T_I8: fk = FK_LDS_I8; break;
            case T_U8: fk = FK_LDS_U8; break;
            case T_F32: fk = FK_LDS_F32; break;
            default: break;
        }
        if (fk != FK_SLOW) {
            payload |= uint32_t(off);
            if (col.validity) fk |= FK_HAS_VALIDITY;
        }
    }
}

Status Executor::FillScanArgs(const StepPlan& plan, const Batch& in, int sinkLdsBytes, int kernelId, ScanArgs& args,
                              LaunchCfg& cfg, int64_t& algoBytes) {
    std::memset(&args, 0, sizeof(args));
    args.nrows = in.Rows;
    args.ncols = int32_t(plan.SourceColumns.size());
    args.ncode = int32_t(plan.Code.size());
    args.nconsts = int32_t(plan.Consts.size());
    args.nullable = plan.Nullable;
    args.status = static_cast<uint32_t*>(StatusWord->Ptr);
    if (const char* dbg = std::getenv("SSA_DEBUG_FLAGS")) args.debug_flags = std::atoi(dbg);
    algoBytes = 0;
    int lds = 0;
    const int kTileRows = kBlock * ScanRowsPerThread;  // rows of one tile of the kernel these arguments are for
    for (int c = 0; c < args.ncols; ++c) {
        const Column& col = in.Cols[size_t(plan.SourceColumns[size_t(c)])];
        DevColumn& d = args.cols[c];
        d.values = col.ValuesPtr();
        d.validity = col.Validity ? static_cast<const uint8_t*>(col.Validity->Ptr) : nullptr;
        d.chars = col.Chars ? static_cast<const uint8_t*>(col.Chars->Ptr) : nullptr;
        d.type = col.Field.Type;
        d.chars_bytes = col.Chars ? int32_t(std::min<size_t>(col.Chars->Size, size_t(INT32_MAX))) : 0;
        d.val_bit_off = col.ValBitOff;
        d.valid_bit_off = col.ValidBitOff;
        args.lds_col_off[c] = -1;
        const int w = TypeWidth(col.Field.Type);
        const int sw = StageWidth(col.Field.Type);  // strings: their int32 offsets
        args.lds_end_off[c] = -1;
        if (sw >= 1 && (reinterpret_cast<uintptr_t>(d.values) & 15) == 0 && lds + kTileRows * sw <= kLdsStageBudget) {
            args.lds_col_off[c] = lds;
            lds += kTileRows * sw;
            if (IsStringLike(col.Field.Type)) {  // per-wave slots for the end offset of each wave's last row
                args.lds_end_off[c] = lds;
                lds += (kBlock / 64) * kEndSlotBytes;
            }
        }
        // algorithmic bytes: every referenced source column once
        if (IsStringLike(col.Field.Type)) {
            algoBytes += (in.Rows + 1) * 4;
            algoBytes += col.Chars ? int64_t(col.Chars->Size > 16 ? col.Chars->Size - 16 : 0) : 0;
        } else if (col.Field.Type == T_BOOL) {
            algoBytes += (in.Rows + 7) / 8;
        } else {
            algoBytes += in.Rows * w;
        }
        if (d.validity) algoBytes += (in.Rows + 7) / 8;
    }
    for (size_t i = 0; i < plan.Consts.size(); ++i) args.consts[i] = plan.Consts[i];
    for (size_t i = 0; i < StepTables.size() && i < size_t(kMaxLookupTables); ++i) args.tables[i] = StepTables[i];
    if (plan.Code.size() > size_t(kMaxCode)) return Status::Unsupported("step too long for one scan program");
    std::memcpy(args.pattern, plan.PatternBytes.data(), plan.PatternBytes.size());
    args.nregs = plan.NumRegsUsed;

    // Liveness of the staged columns inside the step program: the instruction that reads a column last, and the
    // first store into every register.  Columns the sink reads (aggregate arguments, keys) live to the end.
    const int kLiveToEnd = 1 << 30;
    int lastUse[kMaxCols], useCount[kMaxCols], firstStore[kMaxRegs];
    for (int c = 0; c < kMaxCols; ++c) lastUse[c] = -1, useCount[c] = 0;
    for (int r = 0; r < kMaxRegs; ++r) firstStore[r] = kLiveToEnd;
    for (size_t i = 0; i < plan.Code.size(); ++i) {
        const uint32_t ins = plan.Code[i];
        const int op = InsnOp(ins), sk = InsnSk(ins), si = InsnSi(ins);
        if (sk == SRC_COL) {
            lastUse[si] = int(i);
            ++useCount[si];
        }
        if (op == OP_STRLEN || op == OP_STRMATCH || op == OP_STRCMP || op == OP_DECPART) {
            lastUse[si] = kLiveToEnd;
            if (op == OP_STRCMP && (InsnB(ins) & 16)) lastUse[InsnA(ins)] = kLiveToEnd;
        }
        if (op == OP_ST) firstStore[si] = std::min(firstStore[si], int(i));
    }
    auto sinkUses = [&](int reg) {
        if (reg <= -2) lastUse[-2 - reg] = kLiveToEnd;
    };
    for (const AggPlan& ap : plan.Aggs) sinkUses(ap.Spec.reg);
    for (const KeyPlan& kp : plan.Keys) {
        sinkUses(kp.Spec.reg);
        if (kp.Spec.col >= 0) lastUse[kp.Spec.col] = kLiveToEnd;
    }
    const bool noTuning = std::getenv("SSA_NO_LDS_TUNING") != nullptr;

    // A register whose first store comes after the last read of an 8-byte staged column takes over that column's tile
    // (every lane reads and writes only its own rows, so overwriting in place is race-free).
    bool taken[kMaxCols] = {false};
    int regAlias[kMaxRegs];
    int freshRegs = 0;
    for (int r = 0; r < plan.NumRegsUsed; ++r) {
        regAlias[r] = -1;
        for (int c = 0; c < args.ncols && !noTuning; ++c) {
            if (taken[c] || args.lds_col_off[c] < 0 || TypeWidth(args.cols[c].type) != 8) continue;
            if (lastUse[c] >= 0 && lastUse[c] < firstStore[r]) {
                regAlias[r] = c;
                taken[c] = true;
                break;
            }
        }
        if (regAlias[r] < 0) ++freshRegs;
    }
    auto totalLds = [&](int stagedBytes) {
        int t = stagedBytes + freshRegs * kTileRows * 8 + plan.NumRegsUsed * kBlock;
        t = (t + 15) & ~15;
        return t + sinkLdsBytes;
    };
    auto blocksPerCU = [&](int total) { return std::max(1, std::min(8, kLdsPerCU / (total + 1024))); };
    // Narrow columns the program reads exactly once gain little from staging: give their tiles up when that lets one
    // more workgroup fit on a CU.
    if (!noTuning && std::getenv("SSA_UNSTAGE") != nullptr) {
        std::vector<int> cand;
        for (int c = 0; c < args.ncols; ++c) {
            if (args.lds_col_off[c] >= 0 && !taken[c] && lastUse[c] != kLiveToEnd && useCount[c] == 1 &&
                StageWidth(args.cols[c].type) <= 4 && TypeWidth(args.cols[c].type) > 0) cand.push_back(c);
        }
        std::sort(cand.begin(), cand.end(), [&](int x, int y) { return StageWidth(args.cols[x].type) < StageWidth(args.cols[y].type); });
        const int base = blocksPerCU(totalLds(lds));
        int dropped = 0;
        for (size_t n = 1; n <= cand.size(); ++n) {
            dropped += kTileRows * StageWidth(args.cols[cand[n - 1]].type);
            if (blocksPerCU(totalLds(lds - dropped)) > base) {
                for (size_t j = 0; j < n; ++j) args.lds_col_off[cand[j]] = -1, args.lds_end_off[cand[j]] = -1;
                break;
            }
        }
        // re-pack the staged tiles
        lds = 0;
        for (int c = 0; c < args.ncols; ++c) {
            if (args.lds_col_off[c] < 0) continue;
            args.lds_col_off[c] = lds;
            lds += kTileRows * StageWidth(args.cols[c].type);
            if (args.lds_end_off[c] >= 0) {
                args.lds_end_off[c] = lds;
                lds += (kBlock / 64) * kEndSlotBytes;
            }
        }
    }
    for (int r = 0; r < kMaxRegs; ++r) args.lds_reg_offs[r] = 0;
    for (int r = 0; r < plan.NumRegsUsed; ++r) {
        if (regAlias[r] >= 0) {
            args.lds_reg_offs[r] = args.lds_col_off[regAlias[r]];
        } else {
            args.lds_reg_offs[r] = lds;
            lds += kTileRows * 8;
        }
    }
    args.lds_regok_off = lds;
    lds += plan.NumRegsUsed * kBlock;
    lds = (lds + 15) & ~15;
    // string heads: only for string columns whose offsets are staged, and only while they do not cost a workgroup
    for (int c = 0; c < kMaxCols; ++c) args.lds_chars_off[c] = -1;
    for (int c = 0; c < args.ncols && !noTuning; ++c) {
        if (!IsStringLike(args.cols[c].type) || args.lds_col_off[c] < 0) continue;
        if (reinterpret_cast<uintptr_t>(args.cols[c].chars) & 15) continue;  // chunks are copied from 16-byte aligned addresses
        const int bytes = (kBlock / 64) * CharsStageBytes(ScanRowsPerThread);
        if (kLdsPerCU / (lds + sinkLdsBytes + 1024) != kLdsPerCU / (lds + bytes + sinkLdsBytes + 1024)) continue;
        args.lds_chars_off[c] = lds;
        lds += bytes;
    }
    args.lds_sink_off = lds;
    lds += sinkLdsBytes;
    if (lds > kLdsPerCU - 1024) return Status::Unsupported("step needs more LDS than one CU has");

    // pre-decode the step program against the final LDS plan
    for (size_t i = 0; i < plan.Code.size(); ++i) {
        const uint32_t ins = plan.Code[i];
        const int op = InsnOp(ins), sk = InsnSk(ins), si = InsnSi(ins), fa = InsnA(ins), fb = InsnB(ins);
        uint32_t fk = FK_NONE, payload = 0;
        if (sk == SRC_COL) EncodeOperand(args, -2 - si, fk, payload);
        else if (sk == SRC_REG) EncodeOperand(args, si, fk, payload);
        else if (sk == SRC_CONST) { fk = FK_CONST; payload = uint32_t(si); }
        uint32_t ok = OK_GENERIC, a8 = uint32_t(fa), b8 = uint32_t(fb);
        switch (op) {
            case OP_LD: ok = OK_LD; break;
            case OP_ST: ok = OK_ST; payload = uint32_t(si); break;
            case OP_PRED: ok = OK_PRED; break;
            case OP_PREDCMP: ok = OK_PREDCMP; b8 = CmpMaskOf(fb & 15) | (uint32_t(fb >> 4) << 4); break;
            case OP_CMP: ok = OK_CMP; b8 = CmpMaskOf(fb); break;
            case OP_ADD: ok = fa == CLS_F64 ? OK_ADD_F64 : (fa == CLS_I || fa == CLS_U) ? OK_ADD_I : OK_GENERIC; break;
            case OP_SUB: ok = fa == CLS_F64 ? OK_SUB_F64 : (fa == CLS_I || fa == CLS_U) ? OK_SUB_I : OK_GENERIC; break;
            case OP_RSUB: ok = fa == CLS_F64 ? OK_RSUB_F64 : OK_GENERIC; break;
            case OP_MUL: ok = fa == CLS_F64 ? OK_MUL_F64 : (fa == CLS_I || fa == CLS_U) ? OK_MUL_I : OK_GENERIC; break;
            default: break;
        }
        args.code[3 * i] = ok | (fk << 8) | (a8 << 16) | (b8 << 24);
        args.code[3 * i + 1] = payload;
        args.code[3 * i + 2] = ins;
    }

    const int64_t ntiles = (in.Rows + kTileRows - 1) / kTileRows;
    // Resident workgroups per CU as the runtime computes them (LDS and registers); more than 4 has measured slower.
    int perCU = std::min(4, ScanKernelBlocksPerCU(kernelId, uint32_t(lds)));
    if (const char* pc = std::getenv("SSA_BLOCKS_PER_CU")) perCU = std::atoi(pc);
    perCU = std::max(1, std::min(perCU, 8));
    int64_t grid = std::min<int64_t>(std::max<int64_t>(ntiles, 1), int64_t(256) * perCU);
    cfg.grid = int(grid);
    cfg.max_grid = 256 * perCU;
    cfg.lds_bytes = uint32_t(lds);
    cfg.stream = StreamHandle;
    return Status::OK();
}

// The libm functions (FN_ACOSH..FN_HYPOT) exist only in kernels built for a program: inside the generic interpreter their
// code costs every other program registers (TPC-H Q6 on the generic kernel: 4.9 -> 5.7 ms).  A program that uses one is
// built at its first scan whatever its size.
static bool NeedsBuiltKernel(const ScanArgs& args) {
    for (int i = 0; i < args.ncode; ++i) {
        const uint32_t w = args.code[i * 3 + 2];
        if ((InsnOp(w) == OP_FUNC || InsnOp(w) == OP_RFUNC) && InsnB(w) >= FN_ACOSH && InsnB(w) <= FN_HYPOT) return true;
    }
    return false;
}
static Status GenericKernelCan(const ScanArgs& args) {
    if (!NeedsBuiltKernel(args)) return Status::OK();
    return Status::Unsupported("the program uses a math-library function, which runs only in a kernel built for the program, and "
                               "none could be built (SSA_JIT=0, or see the message on stderr)");
}

// The scan kernel compiled for this launch's step program (jit.h), or nullptr: then the generic kernel runs.  The
// constant image is the {ScanArgs, Sink} pair with everything that changes from batch to batch cleared; `cfg` gets the
// grid for the residency of the compiled kernel.
hipFunction_t Executor::SpecializedKernel(const char* header, const char* entry, const std::string& defines, const ScanArgs& args,
                                          const void* sinkImage, size_t sinkBytes, int64_t rows, LaunchCfg& cfg) {
    std::vector<uint32_t> image((sizeof(ScanArgs) + sinkBytes + 3) / 4, 0u);
    ScanArgs a = args;
    a.nrows = 0;
    a.status = nullptr;
    for (int c = 0; c < kMaxCols; ++c) {
        a.cols[c].values = nullptr;
        a.cols[c].validity = nullptr;
        a.cols[c].chars = nullptr;
        a.cols[c].chars_bytes = 0;
        a.cols[c].val_bit_off = 0;
        a.cols[c].valid_bit_off = 0;
    }
    for (int i = 0; i < kMaxOffsetChecks; ++i) {
        a.checks[i].offsets = nullptr;
        a.checks[i].first = 0;
    }
    for (int i = 0; i < kMaxLookupTables; ++i) a.tables[i] = LookupTable{nullptr, nullptr, 0};
    a.slices = nullptr;
    a.block_slice = nullptr;
    // The values of the program's constants and the bytes of its patterns are data of a launch (Machine::ConstBits,
    // Machine::Pattern): a kernel is built for the program's shape -- types, nulls, pattern lengths -- and a query that
    // comes back with other literals finds it (constants and parameters are data in the reference too:
    // ydb/core/tx/program/program.cpp:253-295, :338-356).
    for (int i = 0; i < kMaxConsts; ++i) {
        if (!IsStringLike(a.consts[i].type)) a.consts[i].bits = 0;
    }
    std::memset(a.pattern, 0, sizeof(a.pattern));
    std::memcpy(image.data(), &a, sizeof(a));
    std::memcpy(reinterpret_cast<char*>(image.data()) + sizeof(ScanArgs), sinkImage, sinkBytes);
    SpecKernel k;
    std::string why;
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t policyRows = NeedsBuiltKernel(args) ? std::max(rows, kJitMinRows) : rows;
    const bool got = KernelJit::Instance().Get(Device, header, entry, image.data(), image.size() * 4, defines, cfg.lds_bytes, policyRows, k, why);
    if (std::getenv("SSA_DEBUG_PLAN")) {
        std::fprintf(stderr, "specialised %s: %s, %.1f ms, %d workgroups per CU\n", entry, got ? "ok" : "none",
                     1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), k.BlocksPerCU);
    }
    if (!got) {
        if (why.empty()) return nullptr;  // not enough rows yet
        static std::mutex lock;
        static std::set<std::string> told;
        std::lock_guard<std::mutex> g(lock);
        if (told.insert(why.substr(0, 200)).second) {
            std::fprintf(stderr, "ydb_ssa_hip: no kernel built for this program, the generic one runs (%s)\n", why.c_str());
        }
        return nullptr;
    }
    const int64_t tileRows = int64_t(kBlock) * ScanRowsPerThread;
    const int64_t ntiles = (rows + tileRows - 1) / tileRows;
    // Three resident workgroups per CU have measured best for these kernels at SF100 (a grid of 4 x 256 workgroups --
    // tile strides of a power of two -- is the slowest choice, 5 x 256 again close to 3 x 256).
    // Kernels whose tile is only a few KB (narrow rows) need more of them in flight: up to six workgroups so that a CU
    // has about 48 KB of staged columns outstanding (ClickBench Q1, 2 B/row: 4.8 TB/s at three, 5.5 TB/s at six).
    int64_t tileBytes = 0;
    for (int c = 0; c < args.ncols; ++c) {
        if (args.lds_col_off[c] >= 0) tileBytes += tileRows * StageWidth(args.cols[c].type);
    }
    int want = 3;
    if (tileBytes > 0) want = int(std::max<int64_t>(3, std::min<int64_t>(6, (48 * 1024 + tileBytes - 1) / tileBytes)));
    int perCU = std::max(1, std::min(want, k.BlocksPerCU));
    if (const char* pc = std::getenv("SSA_SPEC_BLOCKS_PER_CU")) perCU = std::max(1, std::min(8, std::atoi(pc)));  // experiments
    cfg.grid = int(std::min<int64_t>(std::max<int64_t>(ntiles, 1), int64_t(256) * perCU));
    cfg.max_grid = 256 * perCU;
    // A keyed scan of a shard-sized batch: a workgroup's set-up and the merge of its table into the step's are paid per
    // workgroup, so every workgroup takes two tiles at least (a million rows of TPC-H Q1: 63 us an apply with 768
    // workgroups of one or two tiles, 60 us with 512 of two).
    if (std::strncmp(entry, "ssa_spec_keyed", 14) == 0 && ntiles > 256 && ntiles < 2 * int64_t(cfg.max_grid)) {
        cfg.grid = int(std::max<int64_t>(256, std::min<int64_t>(cfg.grid, (ntiles + 1) / 2)));
    }
    if (const char* g = std::getenv("SSA_SPEC_GRID")) cfg.grid = int(std::min<int64_t>(std::max<int64_t>(ntiles, 1), std::atoi(g)));  // experiments
    return k.Function;
}

hipError_t Executor::LaunchSpecialized(hipFunction_t fn, const ScanArgs& args, const void* sink, size_t sinkBytes, const LaunchCfg& cfg) {
    std::vector<char> kargs(sizeof(ScanArgs) + sinkBytes);
    std::memcpy(kargs.data(), &args, sizeof(ScanArgs));
    std::memcpy(kargs.data() + sizeof(ScanArgs), sink, sinkBytes);
    size_t bytes = kargs.size();
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, kargs.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
    KernelJit::Instance().NoteLaunch();
    return hipModuleLaunchKernel(fn, unsigned(cfg.grid), 1, 1, kBlock, 1, 1, cfg.lds_bytes, cfg.stream, nullptr, extra);
}

// ---------------------------------------------------------------------------------------------
// helpers producing output columns
// ---------------------------------------------------------------------------------------------
Status Executor::PackValidity(const BufferPtr& validBytes, int64_t rows, Column& col) {
    col.Validity = Buffer::Alloc(size_t((rows + 7) / 8) + 16);
    BufferPtr counter = Buffer::Alloc(256);
    if (!col.Validity || !counter) return Status::Hip("out of device memory");
    SSA_HIP(hipMemsetAsync(counter->Ptr, 0, 8, StreamHandle));
    SSA_HIP(LaunchPackBytesToBits(static_cast<const uint8_t*>(validBytes->Ptr), rows, static_cast<uint8_t*>(col.Validity->Ptr),
                                  static_cast<unsigned long long*>(counter->Ptr), StreamHandle));
    unsigned long long zeros = 0;
    SSA_HIP(hipMemcpyAsync(&zeros, counter->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    col.NullCount = int64_t(zeros);
    col.ValidBitOff = 0;
    if (zeros == 0) col.Validity = nullptr;
    col.Field.Nullable = (col.Validity != nullptr);
    return Status::OK();
}

Status Executor::MaterializeScalar(const OutColumn& oc, int64_t rows, Column& dst) {
    // TDatumBatch::ToRecordBatch -> arrow::MakeArrayFromScalar (program.cpp:582-598)
    dst = Column();
    dst.Field.Name = oc.Name;
    dst.Field.Type = oc.Type;
    dst.Length = rows;
    const TScalar& s = oc.Scalar;
    if (IsStringLike(oc.Type)) {
        std::vector<int32_t> offs(size_t(rows) + 1);
        std::string chars;
        const size_t len = s.Valid ? s.Bytes.size() : 0;
        for (int64_t i = 0; i <= rows; ++i) offs[size_t(i)] = int32_t(size_t(i) * len);
        for (int64_t i = 0; i < rows; ++i) chars.append(s.Bytes.data(), len);
        dst.Values = Buffer::Alloc(offs.size() * 4);
        dst.Chars = Buffer::Alloc(chars.size() + 16);
        if (!dst.Values || !dst.Chars) return Status::Hip("out of device memory");
        SSA_HIP(hipMemcpyAsync(dst.Values->Ptr, offs.data(), offs.size() * 4, hipMemcpyHostToDevice, StreamHandle));
        if (!chars.empty()) SSA_HIP(hipMemcpyAsync(dst.Chars->Ptr, chars.data(), chars.size(), hipMemcpyHostToDevice, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    } else if (oc.Type == T_BOOL) {
        const size_t bytes = size_t((rows + 7) / 8);
        dst.Values = Buffer::Alloc(bytes + 16);
        if (!dst.Values) return Status::Hip("out of device memory");
        SSA_HIP(hipMemsetAsync(dst.Values->Ptr, (s.Valid && s.Bits) ? 0xFF : 0, bytes + 16, StreamHandle));
    } else {
        const int w = TypeWidth(oc.Type);
        dst.Values = Buffer::Alloc(size_t(rows) * w + 16);
        if (!dst.Values) return Status::Hip("out of device memory");
        uint64_t stored = s.Valid ? s.Bits : 0;
        if (oc.Type == T_F32) {
            float f = float(BitsToF64(stored));
            uint32_t b;
            std::memcpy(&b, &f, 4);
            stored = b;
        }
        SSA_HIP(LaunchFill(dst.Values->Ptr, w, stored, rows, StreamHandle));
    }
    if (!s.Valid) {
        const size_t bytes = size_t((rows + 7) / 8);
        dst.Validity = Buffer::Alloc(bytes + 16);
        if (!dst.Validity) return Status::Hip("out of device memory");
        SSA_HIP(hipMemsetAsync(dst.Validity->Ptr, 0, bytes + 16, StreamHandle));
        dst.NullCount = rows;
        dst.Field.Nullable = true;
    }
    return Status::OK();
}

Status Executor::CompactColumn(const Column& src, int64_t rows, const BufferPtr& bitmap, const BufferPtr& tileOffsets,
                               int64_t selected, Column& dst) {
    const uint64_t* bm = static_cast<const uint64_t*>(bitmap->Ptr);
    const int64_t* to = static_cast<const int64_t*>(tileOffsets->Ptr);
    dst = Column();
    dst.Field = src.Field;
    dst.Length = selected;
    dst.Dict = src.Dict;  // a dictionary column is compacted as its codes
    const int type = src.StorageType();
    if (IsStringLike(type)) {
        BufferPtr lens = Buffer::Alloc(size_t(selected + 1) * 4 + 16);
        dst.Values = Buffer::Alloc(size_t(selected + 1) * 4 + 16);
        const int64_t nblocks = (selected + 4095) / 4096 + 2;
        BufferPtr tmp = Buffer::Alloc(size_t(nblocks) * 8 + 16);
        if (!lens || !dst.Values || !tmp) return Status::Hip("out of device memory");
        const int32_t* offs = static_cast<const int32_t*>(src.ValuesPtr());
        SSA_HIP(LaunchCompactStringLens(offs, static_cast<int32_t*>(lens->Ptr), rows, bm, to, StreamHandle));
        SSA_HIP(LaunchExclusiveScanI32(static_cast<const int32_t*>(lens->Ptr), static_cast<int32_t*>(dst.Values->Ptr), selected,
                                       static_cast<int64_t*>(tmp->Ptr), StreamHandle));
        int32_t total = 0;
        SSA_HIP(hipMemcpyAsync(&total, static_cast<const int32_t*>(dst.Values->Ptr) + selected, 4, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        dst.Chars = Buffer::Alloc(size_t(total) + 16);
        if (!dst.Chars) return Status::Hip("out of device memory");
        SSA_HIP(LaunchCompactStringBytes(offs, static_cast<const uint8_t*>(src.Chars->Ptr), static_cast<const int32_t*>(dst.Values->Ptr),
                                         static_cast<uint8_t*>(dst.Chars->Ptr), rows, bm, to, StreamHandle));
    } else if (type == T_BOOL) {
        BufferPtr bytes = Buffer::Alloc(size_t(selected) + 16);
        dst.Values = Buffer::Alloc(size_t((selected + 7) / 8) + 16);
        if (!bytes || !dst.Values) return Status::Hip("out of device memory");
        SSA_HIP(LaunchCompactBits(static_cast<const uint8_t*>(src.Values->Ptr), src.ValBitOff, static_cast<uint8_t*>(bytes->Ptr), rows, bm,
                                  to, StreamHandle));
        SSA_HIP(LaunchPackBytesToBits(static_cast<const uint8_t*>(bytes->Ptr), selected, static_cast<uint8_t*>(dst.Values->Ptr), nullptr,
                                      StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    } else {
        const int w = TypeWidth(type);
        dst.Values = Buffer::Alloc(size_t(selected) * w + 16);
        if (!dst.Values) return Status::Hip("out of device memory");
        SSA_HIP(LaunchCompactFixed(src.ValuesPtr(), dst.Values->Ptr, w, rows, bm, to, StreamHandle));
    }
    if (src.Validity) {
        BufferPtr bytes = Buffer::Alloc(size_t(selected) + 16);
        if (!bytes) return Status::Hip("out of device memory");
        SSA_HIP(LaunchCompactBits(static_cast<const uint8_t*>(src.Validity->Ptr), src.ValidBitOff, static_cast<uint8_t*>(bytes->Ptr), rows,
                                  bm, to, StreamHandle));
        SSA_RETURN_NOT_OK(PackValidity(bytes, selected, dst));
    }
    return Status::OK();
}

// ---------------------------------------------------------------------------------------------
// steps
// ---------------------------------------------------------------------------------------------
Status Executor::RunFilterBitmap(const StepPlan& filterPlan, const Batch& in, BufferPtr& bitmap, BufferPtr& tileOffsets,
                                 int64_t& selected) {
    const int64_t rows = in.Rows;
    const int64_t nwords = (rows + 63) / 64;
    const int64_t ntiles = (rows + kTileRows - 1) / kTileRows;
    bitmap = Buffer::Alloc(size_t(nwords) * 8 + 64);
    BufferPtr tileCounts = Buffer::Alloc(size_t(ntiles) * 4 + 64);
    const int64_t scanBlocks = (ntiles + 4095) / 4096 + 2;
    tileOffsets = Buffer::Alloc(size_t(ntiles + 1 + scanBlocks) * 8 + 64);
    BufferPtr total = Buffer::Alloc(256);
    if (!bitmap || !tileCounts || !tileOffsets || !total) return Status::Hip("out of device memory");
    ScanArgs args;
    LaunchCfg cfg;
    int64_t algoBytes;
    SSA_RETURN_NOT_OK(FillScanArgs(filterPlan, in, 0, SCAN_FILTER, args, cfg, algoBytes));
    FilterSink sink{static_cast<uint64_t*>(bitmap->Ptr), static_cast<uint32_t*>(tileCounts->Ptr)};
    if (rows > 0) {
        hipFunction_t spec = nullptr;
        if (JitEnabled()) {
            const FilterSink image{};  // the sink holds addresses only
            const std::string defines = std::string("#define SSA_SPEC_U ") + std::to_string(kRowsPerThread) + "\n";
            spec = SpecializedKernel("filter_project_scan.h", "ssa_spec_filter", defines, args, &image, sizeof(image), rows, cfg);
        }
        ProfileBegin();
        if (spec) {
            SSA_HIP(LaunchSpecialized(spec, args, &sink, sizeof(sink), cfg));
        } else {
            SSA_RETURN_NOT_OK(GenericKernelCan(args));
            SSA_HIP(LaunchScanFilter(args, sink, cfg));
        }
        ProfileEnd("ScanFilterKernel", algoBytes);
    }
    SSA_HIP(LaunchTileScan(static_cast<const uint32_t*>(tileCounts->Ptr), ntiles, static_cast<int64_t*>(tileOffsets->Ptr),
                           static_cast<int64_t*>(total->Ptr), StreamHandle));
    int64_t sel = 0;
    int64_t* selAt = HostWords ? reinterpret_cast<int64_t*>(HostWords + 8) : &sel;  // (pinned: no staging copy)
    SSA_HIP(hipMemcpyAsync(selAt, total->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_RETURN_NOT_OK(CheckScanStatus());
    selected = *selAt;
    return Status::OK();
}

// Rows per thread of a kernel built for a key-less aggregation: 16 when the staged source columns add up to at most 8
// bytes per row and the program keeps no temporaries, 8 up to 16 bytes per row, else the library's 4.  (The validity
// bits of a temporary's rows share one byte per thread: 8 rows at most.)
int Executor::WideRowsPerThread(const StepPlan& plan, const Batch& in) const {
    if (std::getenv("SSA_NO_WIDE_TILES")) return kRowsPerThread;
    int bytes = 0;
    for (size_t c = 0; c < plan.SourceColumns.size(); ++c) {
        const Column& col = in.Cols[size_t(plan.SourceColumns[c])];
        if (IsStringLike(col.Field.Type) || col.Field.Type == T_DEC128 || col.Field.Type == T_BOOL || col.Validity) return kRowsPerThread;
        bytes += StageWidth(col.Field.Type);
    }
    bytes += 8 * plan.NumRegsUsed;
    if (bytes == 0) return kRowsPerThread;
    // (32 rows per thread for rows of two bytes measured no faster than 16 -- ClickBench's AdvEngineID count 0.43 against
    // 0.41 ms per 10^9 rows: with every row masked out the same kernel takes 0.32 ms, so what is left above the loads is
    // the rows' own arithmetic on 64-bit payloads, not the per-tile work; SSA_WIDE_U32=1 builds such kernels all the same)
    if (bytes <= 2 && plan.NumRegsUsed == 0 && std::getenv("SSA_WIDE_U32")) return 32;
    if (bytes <= 8 && plan.NumRegsUsed == 0) return 16;
    if (bytes <= 16) return 8;
    return kRowsPerThread;
}

Status Executor::RunNoKey(const StepPlan& plan, const Batch& in, BatchPtr& out) {
    ScanArgs args;
    LaunchCfg cfg;
    int64_t algoBytes;
    const int naggs = int(plan.Aggs.size());
    SSA_RETURN_NOT_OK(FillScanArgs(plan, in, naggs * kBlock * 8, SCAN_NOKEY, args, cfg, algoBytes));
    NoKeySink sink{};
    sink.naggs = naggs;
    for (int k = 0; k < naggs; ++k) {
        sink.aggs[k] = plan.Aggs[size_t(k)].Spec;
        if (sink.aggs[k].reg != -1) EncodeOperand(args, sink.aggs[k].reg, sink.aggs[k].fetch_kind, sink.aggs[k].fetch_payload);
    }
    // a scan of many rows runs the kernel built for this step program, when that can be had
    hipFunction_t spec = nullptr;
    if (JitEnabled()) {
        // Narrow rows: a tile of 1024 rows is a few KB and the per-tile work (staging, waits, loop) outweighs the rows'.
        // The built kernel then takes 8 or 16 rows per thread; its LDS plan is made for that tile size.
        const int wide = WideRowsPerThread(plan, in);
        if (wide != kRowsPerThread) {
            ScanArgs wargs;
            LaunchCfg wcfg;
            int64_t wbytes;
            ScanRowsPerThread = wide;
            if (FillScanArgs(plan, in, naggs * kBlock * 8, SCAN_NOKEY, wargs, wcfg, wbytes).ok()) {
                NoKeySink wsink = sink;
                for (int k = 0; k < naggs; ++k) {
                    if (wsink.aggs[k].reg != -1) EncodeOperand(wargs, wsink.aggs[k].reg, wsink.aggs[k].fetch_kind, wsink.aggs[k].fetch_payload);
                }
                const std::string defines = std::string("#define SSA_SPEC_U ") + std::to_string(wide) + "\n";
                spec = SpecializedKernel("nokey_scan.h", "ssa_spec_nokey", defines, wargs, &wsink, sizeof(wsink), in.Rows, wcfg);
                if (spec) {
                    args = wargs;
                    cfg = wcfg;
                    sink = wsink;
                }
            }
            ScanRowsPerThread = kRowsPerThread;
        }
        if (!spec) {
            const std::string defines = std::string("#define SSA_SPEC_U ") + std::to_string(kRowsPerThread) + "\n";
            spec = SpecializedKernel("nokey_scan.h", "ssa_spec_nokey", defines, args, &sink, sizeof(sink), in.Rows, cfg);
        }
    }
    HostPhases::Mark("args+kernel");
    // partials, seen bits and the result row in one block: [grid][kMaxAggs] words, [grid] words, 4 * kMaxOuts + 1 words
    if (plan.Outs.size() > size_t(kMaxOuts)) return Status::Unsupported("more output columns than a key-less aggregation writes");
    const size_t partialBytes = size_t(cfg.grid) * kMaxAggs * 8, seenBytes = (size_t(cfg.grid) * 4 + 15) & ~size_t(15);
    BufferPtr partial = Buffer::Alloc(partialBytes + seenBytes);
    BufferPtr result = Buffer::Alloc((4 * kMaxOuts + 1) * 8);
    if (!partial || !result) return Status::Hip("out of device memory");
    sink.partial = static_cast<uint64_t*>(partial->Ptr);
    sink.partial_seen = reinterpret_cast<uint32_t*>(static_cast<char*>(partial->Ptr) + partialBytes);
    ProfileBegin();
    if (spec) {
        SSA_HIP(LaunchSpecialized(spec, args, &sink, sizeof(sink), cfg));
    } else {
        SSA_RETURN_NOT_OK(GenericKernelCan(args));
        SSA_HIP(LaunchScanNoKey(args, sink, cfg));
    }
    ProfileEnd("ScanNoKeyKernel", algoBytes);

    HostPhases::Mark("scan-launch");
    // One output row; columns follow plan.Outs (aggregates in GroupBy order unless a projection reorders them).  The
    // finalize launch writes the values where the output columns hold them; the host learns the scan's status and which
    // results are null from two words.
    NoKeyOutSpec outs[kMaxOuts];
    int nouts = 0;
    for (const OutColumn& oc : plan.Outs) {
        const AggPlan& ap = plan.Aggs[size_t(oc.Index)];
        NoKeyOutSpec os{};
        os.agg = oc.Index;
        os.mode = NOKEY_OUT_PLAIN;
        if (ap.DecMid >= 0) {
            os.mode = NOKEY_OUT_DECIMAL;
            os.mid = ap.DecMid;
            os.hi = ap.DecHi;
        } else if (ap.SomeInputColumn >= 0) {  // the state is the smallest row index with a non-null argument
            const Column& src = in.Cols[size_t(ap.SomeInputColumn)];
            os.mode = NOKEY_OUT_ROW;
            os.src = src.ValuesPtr();
            os.width = TypeWidth(src.Field.Type);
            if (os.width < 1 || os.width > 8) return Status::Unsupported("some() over values wider than 8 bytes");
        } else if (oc.Type == T_F32) {
            os.mode = NOKEY_OUT_F32;
        }
        outs[nouts++] = os;
    }
    uint64_t* dres = static_cast<uint64_t*>(result->Ptr);
    uint32_t words[2] = {0, 0};
    SSA_HIP(LaunchNoKeyFinalize(sink, cfg.grid, outs, nouts, dres, static_cast<const uint32_t*>(StatusWord->Ptr),
                                reinterpret_cast<uint32_t*>(HostWords), StreamHandle));
    HostPhases::Mark("finalize-launch");
    if (HostWords) {
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        words[0] = uint32_t(HostWords[0]);
        words[1] = uint32_t(HostWords[1]);
    } else {
        SSA_HIP(hipMemcpyAsync(words, dres + 4 * kMaxOuts, 8, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    HostPhases::Mark("sync");
    SSA_RETURN_NOT_OK(DecodeScanStatus(words[0], nullptr, nullptr));

    out = std::make_shared<Batch>();
    out->Rows = 1;
    int o = 0;
    for (const OutColumn& oc : plan.Outs) {
        const AggPlan& ap = plan.Aggs[size_t(oc.Index)];
        Column col;
        col.Field.Name = oc.Name;
        col.Field.Type = oc.Type;
        if (ap.DecMid >= 0) {
            col.Field.Precision = ap.Precision;
            col.Field.Scale = ap.Scale;
        }
        col.Length = 1;
        col.Values = result;
        if (oc.Type == T_BOOL) col.ValBitOff = o * 256;
        else col.ElemOffset = int64_t(o) * 32 / TypeWidth(oc.Type);
        if (!((words[1] >> o) & 1u)) {
            col.Validity = result;
            col.ValidBitOff = (4 * o + 2) * 64;
            col.NullCount = 1;
            col.Field.Nullable = true;
        }
        out->Cols.push_back(std::move(col));
        ++o;
    }
    return Status::OK();
}

Status Executor::RunProject(const TProgramStep& step, const StepPlan& plan, const BatchPtr& inPtr, BatchPtr& out) {
    const Batch& in = *inPtr;
    const int64_t rows = in.Rows;
    BufferPtr bitmap, tileOffsets;
    int64_t selected = rows;
    bool filtered = false;
    if (plan.HasFilter) {
        StepPlan filterPlan;
        SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_FILTER_ONLY, filterPlan));
        SSA_RETURN_NOT_OK(RunFilterBitmap(filterPlan, in, bitmap, tileOffsets, selected));
        filtered = (selected != rows);  // TColumnFilter::IsTotalAllowFilter -> nothing to do
    }
    out = std::make_shared<Batch>();
    out->Rows = selected;
    out->Cols.resize(plan.Outs.size());

    // computed columns: one more scan that writes the output registers at the compacted positions
    std::vector<int> regOuts;
    for (size_t i = 0; i < plan.Outs.size(); ++i) {
        if (plan.Outs[i].Source == OutColumn::FROM_REG) regOuts.push_back(int(i));
    }
    if (regOuts.empty() && !plan.HasFilter && plan.UnreadFallible) {
        // nothing to filter or compute, yet an assign that nothing reads may fail the program: one pass for its errors
        StepPlan errorsPlan;
        BufferPtr allBits, allOffsets;
        int64_t all = rows;
        SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_FILTER_ONLY, errorsPlan));
        SSA_RETURN_NOT_OK(RunFilterBitmap(errorsPlan, in, allBits, allOffsets, all));
    }
    if (!regOuts.empty()) {
        if (regOuts.size() > size_t(kMaxOuts)) return Status::Unsupported("too many computed output columns");
        StepPlan outPlan;
        SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_OUTPUTS_ONLY, outPlan));
        ScanArgs args;
        LaunchCfg cfg;
        int64_t algoBytes;
        SSA_RETURN_NOT_OK(FillScanArgs(outPlan, in, 0, SCAN_PROJECT, args, cfg, algoBytes));
        ProjectSink sink{};
        sink.bitmap = filtered ? static_cast<const uint64_t*>(bitmap->Ptr) : nullptr;
        sink.tile_offsets = filtered ? static_cast<const int64_t*>(tileOffsets->Ptr) : nullptr;
        sink.nouts = int32_t(regOuts.size());
        std::vector<BufferPtr> validBytes(regOuts.size()), boolBytes(regOuts.size());
        for (size_t k = 0; k < regOuts.size(); ++k) {
            const OutColumn& oc = outPlan.Outs[size_t(regOuts[k])];
            Column& col = out->Cols[size_t(regOuts[k])];
            col.Field.Name = oc.Name;
            col.Field.Type = oc.Type;
            col.Length = selected;
            const int w = (oc.Type == T_BOOL) ? 1 : TypeWidth(oc.Type);
            BufferPtr vals = Buffer::Alloc(size_t(selected) * w + 16);
            if (!vals) return Status::Hip("out of device memory");
            if (oc.Type == T_BOOL) boolBytes[k] = vals; else col.Values = vals;
            sink.outs[k].values = vals->Ptr;
            sink.outs[k].type = oc.Type;
            sink.outs[k].reg = oc.Index;
            EncodeOperand(args, oc.Index, sink.outs[k].fetch_kind, sink.outs[k].fetch_payload);
            sink.outs[k].valid_bytes = nullptr;
            if (oc.Nullable) {
                validBytes[k] = Buffer::Alloc(size_t(selected) + 16);
                if (!validBytes[k]) return Status::Hip("out of device memory");
                sink.outs[k].valid_bytes = static_cast<uint8_t*>(validBytes[k]->Ptr);
            }
        }
        if (rows > 0) {
            hipFunction_t spec = nullptr;
            if (JitEnabled()) {
                ProjectSink image = sink;
                image.bitmap = nullptr;
                image.tile_offsets = nullptr;
                for (int k = 0; k < kMaxOuts; ++k) {
                    image.outs[k].values = nullptr;
                    image.outs[k].valid_bytes = nullptr;
                }
                const std::string defines = std::string("#define SSA_SPEC_U ") + std::to_string(kRowsPerThread) + "\n";
                spec = SpecializedKernel("filter_project_scan.h", "ssa_spec_project", defines, args, &image, sizeof(image), rows, cfg);
            }
            ProfileBegin();
            if (spec) {
                SSA_HIP(LaunchSpecialized(spec, args, &sink, sizeof(sink), cfg));
            } else {
                SSA_RETURN_NOT_OK(GenericKernelCan(args));
                SSA_HIP(LaunchScanProject(args, sink, cfg));
            }
            ProfileEnd("ScanProjectKernel", algoBytes);
        }
        for (size_t k = 0; k < regOuts.size(); ++k) {
            Column& col = out->Cols[size_t(regOuts[k])];
            if (boolBytes[k]) {
                col.Values = Buffer::Alloc(size_t((selected + 7) / 8) + 16);
                if (!col.Values) return Status::Hip("out of device memory");
                SSA_HIP(LaunchPackBytesToBits(static_cast<const uint8_t*>(boolBytes[k]->Ptr), selected,
                                              static_cast<uint8_t*>(col.Values->Ptr), nullptr, StreamHandle));
            }
            if (validBytes[k]) SSA_RETURN_NOT_OK(PackValidity(validBytes[k], selected, col));
        }
        SSA_RETURN_NOT_OK(CheckScanStatus());
    }

    for (size_t i = 0; i < plan.Outs.size(); ++i) {
        const OutColumn& oc = plan.Outs[i];
        Column& col = out->Cols[i];
        if (oc.Source == OutColumn::FROM_INPUT) {
            const Column& src = in.Cols[size_t(oc.Index)];
            if (!filtered) {
                col = src;  // zero-copy: buffers are shared
            } else {
                SSA_RETURN_NOT_OK(CompactColumn(src, rows, bitmap, tileOffsets, selected, col));
            }
            col.Field.Name = oc.Name;
        } else if (oc.Source == OutColumn::FROM_SCALAR) {
            SSA_RETURN_NOT_OK(MaterializeScalar(oc, selected, col));
        }
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

namespace {
uint64_t NextPow2(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
}  // namespace

// Rows of `src` at the given 64-bit row indices (~0 -> null).
Status Executor::GatherRows(const Column& src, const BufferPtr& rowIdx, int64_t n, bool keepValidity, Column& dst) {
    const uint64_t* rows = static_cast<const uint64_t*>(rowIdx->Ptr);
    dst = Column();
    dst.Field = src.Field;
    dst.Field.Nullable = false;
    dst.Length = n;
    const int type = src.Field.Type;
    BufferPtr validBytes;
    if (keepValidity) {
        validBytes = Buffer::Alloc(size_t(n) + 16);
        if (!validBytes) return Status::Hip("out of device memory");
    }
    const uint8_t* srcValid = src.Validity ? static_cast<const uint8_t*>(src.Validity->Ptr) : nullptr;
    uint8_t* dstValid = validBytes ? static_cast<uint8_t*>(validBytes->Ptr) : nullptr;
    if (IsStringLike(type)) {
        BufferPtr lens = Buffer::Alloc(size_t(n + 1) * 4 + 16);
        dst.Values = Buffer::Alloc(size_t(n + 1) * 4 + 16);
        const int64_t nblocks = (n + 4095) / 4096 + 2;
        BufferPtr tmp = Buffer::Alloc(size_t(nblocks) * 8 + 16);
        if (!lens || !dst.Values || !tmp) return Status::Hip("out of device memory");
        const int32_t* offs = static_cast<const int32_t*>(src.ValuesPtr());
        SSA_HIP(LaunchGatherStringLens(offs, rows, n, static_cast<int32_t*>(lens->Ptr), StreamHandle));
        SSA_HIP(LaunchExclusiveScanI32(static_cast<const int32_t*>(lens->Ptr), static_cast<int32_t*>(dst.Values->Ptr), n,
                                       static_cast<int64_t*>(tmp->Ptr), StreamHandle));
        int32_t total = 0;
        SSA_HIP(hipMemcpyAsync(&total, static_cast<const int32_t*>(dst.Values->Ptr) + n, 4, hipMemcpyDeviceToHost, StreamHandle));
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        dst.Chars = Buffer::Alloc(size_t(total) + 16);
        if (!dst.Chars) return Status::Hip("out of device memory");
        SSA_HIP(LaunchGatherStringBytes(offs, static_cast<const uint8_t*>(src.Chars->Ptr), rows, n,
                                        static_cast<const int32_t*>(dst.Values->Ptr), static_cast<uint8_t*>(dst.Chars->Ptr), StreamHandle));
        if (dstValid) SSA_HIP(LaunchGatherValidity(srcValid, src.ValidBitOff, rows, n, dstValid, StreamHandle));
    } else if (type == T_BOOL) {
        return Status::Unsupported("gather of a boolean column");
    } else {
        const int w = TypeWidth(type);
        dst.Values = Buffer::Alloc(size_t(n) * w + 16);
        if (!dst.Values) return Status::Hip("out of device memory");
        SSA_HIP(LaunchGatherFixed(src.ValuesPtr(), srcValid, src.ValidBitOff, w, rows, n, dst.Values->Ptr, dstValid, StreamHandle));
    }
    if (validBytes) SSA_RETURN_NOT_OK(PackValidity(validBytes, n, dst));
    return Status::OK();
}

// TProgramStep::ApplyAggregates with keys (ydb/core/formats/arrow/program.cpp:760-832): the step's
// assigns and filters run fused with a two-level hash aggregation; the table is then turned into one
// output row per group, aggregates first, keys after them.
Status Executor::RunKeyed(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out) {
    // String keys whose strings all have one length of 1, 2, 4 or 8 bytes (flags, codes, fixed-width identifiers) are
    // scanned as fixed-width keys: the key column becomes the chars buffer read as `len`-byte values, the keys pack
    // exactly and the kernel is the one of integer keys.  The scan still reads the offsets and checks every one of them
    // (OffsetCheck); a batch that turns out otherwise is run again the general way.
    KeyedHint& hint0 = KeyedHints[HintOverride ? HintOverride : &step];
    if (!hint0.UniformKeysFailed && in.Rows >= kUniformKeysMinRows && !(plan.NeedsFilteredPositions && plan.HasFilter) &&
        !std::getenv("SSA_NO_UNIFORM_KEYS")) {
        bool done = false;
        SSA_RETURN_NOT_OK(RunKeyedUniform(step, plan, in, out, done));
        if (done) return Status::OK();
        hint0.UniformKeysFailed = true;
    }
    std::vector<OffsetCheck> none;
    bool notUniform = false;
    return RunKeyedScan(step, plan, in, out, none, &notUniform);
}

// The offsets of n strings of `len` bytes each (0, len, 2 len ..): views of one ramp per length, made once and never
// written again -- results of few groups get their offsets without a launch.
Status Executor::OffsetRamp(int len, int64_t n, BufferPtr& out) {
    constexpr int64_t kRampRows = 65536;
    static_assert(kSmallEmitCounts + kSmallEmitTables <= kHostWords, "the counts of the small tables fit the pinned words");
    if (n > kRampRows) {
        out = Buffer::Alloc(size_t(n + 1) * 4 + 16);
        if (!out) return Status::Hip("out of device memory");
        SSA_HIP(LaunchIotaI32(static_cast<int32_t*>(out->Ptr), n + 1, int32_t(len), StreamHandle));
        return Status::OK();
    }
    BufferPtr& ramp = OffsetRamps[len];
    if (!ramp) {
        ramp = Buffer::Alloc(size_t(kRampRows + 1) * 4 + 16);
        if (!ramp) return Status::Hip("out of device memory");
        SSA_HIP(LaunchIotaI32(static_cast<int32_t*>(ramp->Ptr), kRampRows + 1, int32_t(len), StreamHandle));
    }
    out = ramp;
    return Status::OK();
}

// The plan of `step` with string keys `renames` (key index, name) read from fixed-width columns of `shadow` instead;
// kept per step, schema and key widths (planning is a twentieth of an apply of a shard-sized batch).
Status Executor::FixedKeysPlan(const TProgramStep& step, const Batch& shadow, const std::vector<std::pair<size_t, std::string>>& renames,
                               const std::vector<uint32_t>& lens, std::shared_ptr<const StepPlan>& kept) {
    std::string signature;
    const void* address = &step;
    signature.append(reinterpret_cast<const char*>(&address), sizeof(address));
    signature.push_back('F');
    for (uint32_t len : lens) signature.push_back(char('0' + len));
    bool stable = true;
    for (const Column& c : shadow.Cols) {
        if (c.Field.IsScalar) stable = false;
        signature.append(c.Field.Name);
        signature.push_back('\0');
        const int32_t words[4] = {c.Field.Type, c.Field.Nullable ? 1 : 0, c.Field.Precision, c.Field.Scale};
        signature.append(reinterpret_cast<const char*>(words), sizeof(words));
    }
    // (the steps of a program live as long as its executor; a rewritten copy does not)
    stable = stable && StableStep;
    if (stable) {
        auto it = PlanCache.find(signature);
        if (it != PlanCache.end()) {
            kept = it->second;
            return Status::OK();
        }
    }
    TProgramStep fixedStep = step;
    for (const auto& r : renames) {
        const std::string original = step.GroupByKeys[r.first].ColumnName;
        fixedStep.GroupByKeys[r.first].ColumnName = r.second;
        for (auto& p : fixedStep.Projection) {
            if (p.ColumnName == original) p.ColumnName = r.second;
        }
    }
    auto fresh = std::make_shared<StepPlan>();
    SSA_RETURN_NOT_OK(CompileStep(fixedStep, shadow.Schema(), CM_FULL, *fresh, /*serializedKeys=*/true));
    kept = fresh;
    if (stable) {
        if (PlanCache.size() >= 64) PlanCache.clear();
        PlanCache[signature] = kept;
    }
    return Status::OK();
}

Status Executor::RunKeyedUniform(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out, bool& done) {
    done = false;
    struct Rewrite {
        size_t Key;
        int Column;
        uint32_t First, Len;
        std::string Shadow;
    };
    std::vector<Rewrite> rewrites;
    int keyBits = 0;
    for (size_t k = 0; k < plan.Keys.size(); ++k) {
        const KeyPlan& kp = plan.Keys[k];
        if (kp.InputColumn < 0) return Status::OK();
        if (!IsStringLike(kp.Spec.type)) {
            keyBits += 8 * TypeWidth(kp.Spec.type);
            continue;
        }
        rewrites.push_back(Rewrite{k, kp.InputColumn, 0, 0, std::string("\x01fixed:") + std::to_string(k)});
    }
    if (rewrites.empty() || rewrites.size() > size_t(kMaxOffsetChecks) || in.Cols.size() + rewrites.size() > 64) return Status::OK();
    // first, second and last offset of every such column: the strings can only all have one length when
    // offsets[rows] - offsets[0] == rows * (offsets[1] - offsets[0])
    std::vector<int32_t> probe(rewrites.size() * 3, 0);
    ProbeWords words{};
    for (size_t i = 0; i < rewrites.size(); ++i) {
        const Column& col = in.Cols[size_t(rewrites[i].Column)];
        if (!col.Values || !col.Chars) return Status::OK();
        const int32_t* offs = static_cast<const int32_t*>(col.ValuesPtr());
        if (reinterpret_cast<uintptr_t>(offs) & 15) return Status::OK();
        words.addr[3 * i] = offs;
        words.addr[3 * i + 1] = offs + 1;
        words.addr[3 * i + 2] = offs + in.Rows;
    }
    if (HostWords && probe.size() <= size_t(kMaxProbeWords) && probe.size() <= size_t(kHostWords)) {
        // one launch that reads them all into pinned memory (a copy per word into pageable memory is a round trip each)
        words.n = int32_t(probe.size());
        words.out = HostWords;
        SSA_HIP(LaunchProbeI32(words, StreamHandle));
        HostPhases::Mark("probe-launch");
        SSA_HIP(hipStreamSynchronize(StreamHandle));
        for (size_t i = 0; i < probe.size(); ++i) probe[i] = HostWords[i];
        HostPhases::Mark("probe-sync");
    } else {
        for (size_t i = 0; i < rewrites.size(); ++i) {
            SSA_HIP(hipMemcpyAsync(&probe[3 * i], words.addr[3 * i], 8, hipMemcpyDeviceToHost, StreamHandle));
            SSA_HIP(hipMemcpyAsync(&probe[3 * i + 2], words.addr[3 * i + 2], 4, hipMemcpyDeviceToHost, StreamHandle));
        }
        SSA_HIP(hipStreamSynchronize(StreamHandle));
    }
    for (size_t i = 0; i < rewrites.size(); ++i) {
        const int64_t first = probe[3 * i], len = int64_t(probe[3 * i + 1]) - first, last = probe[3 * i + 2];
        if (first < 0 || (len != 1 && len != 2 && len != 4 && len != 8) || last - first != len * in.Rows) return Status::OK();
        const Column& col = in.Cols[size_t(rewrites[i].Column)];
        if (size_t(last) > col.Chars->Size) return Status::OK();
        rewrites[i].First = uint32_t(first);
        rewrites[i].Len = uint32_t(len);
        keyBits += int(8 * len);
    }
    if (keyBits > 119) return Status::OK();

    // the batch with the chars buffers as extra fixed-width columns, the step with its keys renamed to them
    Batch shadow = in;
    std::vector<OffsetCheck> checks;
    std::vector<std::pair<size_t, std::string>> renames;
    std::vector<uint32_t> lens;
    for (const Rewrite& r : rewrites) {
        const Column& src = in.Cols[size_t(r.Column)];
        Column col;
        col.Field.Name = r.Shadow;
        col.Field.Type = r.Len == 1 ? T_U8 : r.Len == 2 ? T_U16 : r.Len == 4 ? T_U32 : T_U64;
        col.Length = in.Rows;
        col.Values = Buffer::Wrap(static_cast<char*>(src.Chars->Ptr) + r.First, size_t(in.Rows) * r.Len);
        shadow.Cols.push_back(col);
        renames.emplace_back(r.Key, r.Shadow);
        lens.push_back(r.Len);
        checks.push_back(OffsetCheck{static_cast<const int32_t*>(src.ValuesPtr()), r.First, r.Len});
    }
    std::shared_ptr<const StepPlan> fixedKept;
    if (!FixedKeysPlan(step, shadow, renames, lens, fixedKept).ok() || fixedKept->Kind != StepPlan::AGG_KEYED) return Status::OK();
    const StepPlan& fixedPlan = *fixedKept;
    HostPhases::Mark("fixed-plan");
    BatchPtr res;
    bool notUniform = false;
    SSA_RETURN_NOT_OK(RunKeyedScan(step, fixedPlan, shadow, res, checks, &notUniform));
    if (notUniform) return Status::OK();
    // the key columns back as strings: the emitted values are the chars, the offsets are multiples of the length
    for (Column& col : res->Cols) {
        for (const Rewrite& r : rewrites) {
            if (col.Field.Name != r.Shadow) continue;
            const Column& src = in.Cols[size_t(r.Column)];
            const int64_t n = res->Rows;
            Column str;
            str.Field = src.Field;
            str.Field.Nullable = false;
            str.Length = n;
            SSA_RETURN_NOT_OK(OffsetRamp(int(r.Len), n, str.Values));
            str.Chars = col.Values;
            if (!str.Chars) {
                str.Chars = Buffer::Alloc(16);
                if (!str.Chars) return Status::Hip("out of device memory");
            }
            col = str;
        }
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = res;
    done = true;
    return Status::OK();
}

// The slices of a launch as the kernel reads them: each slice's column addresses, its share of the grid (in proportion
// to its tiles, at least one workgroup) and its table.  `args` / `cfg` are those of the first slice; the other slices
// must fit the LDS plan made for it (same types, validity where it has validity, aligned where its columns are staged).
Status Executor::DescribeSlices(const StepPlan& plan, ManySlices& many, ScanArgs& args, int64_t tileRows, uint64_t cap, LaunchCfg& cfg,
                                BufferPtr& descs, BufferPtr& map) {
    const size_t n = many.In.size();
    std::vector<int64_t> tiles(n);
    int64_t totalTiles = 0;
    for (size_t i = 0; i < n; ++i) {
        tiles[i] = (many.In[i]->Rows + tileRows - 1) / tileRows;
        totalTiles += tiles[i];
    }
    const int64_t wanted = std::max<int64_t>(1, std::min<int64_t>(totalTiles, std::max(cfg.max_grid, 1)));
    many.HostDescs.assign(n, SliceDesc{});
    many.HostMap.clear();
    for (size_t i = 0; i < n; ++i) {
        const Batch& b = *many.In[i];
        SliceDesc& d = many.HostDescs[i];
        d.nrows = b.Rows;
        d.table_off = uint64_t(i) * cap;
        int64_t blocks = tiles[i] == 0 ? 0 : std::max<int64_t>(1, std::min<int64_t>(tiles[i], wanted * tiles[i] / std::max<int64_t>(totalTiles, 1)));
        d.first_block = int32_t(many.HostMap.size());
        d.nblocks = int32_t(blocks);
        for (int64_t k = 0; k < blocks; ++k) many.HostMap.push_back(int32_t(i));
        if (b.Cols.size() != many.In[0]->Cols.size()) return Status::Unsupported("slices in one launch: differing schemas");
        for (int c = 0; c < args.ncols; ++c) {
            const Column& col = b.Cols[size_t(plan.SourceColumns[size_t(c)])];
            DevColumn& dc = d.cols[c];
            dc.values = col.ValuesPtr();
            dc.validity = col.Validity ? static_cast<const uint8_t*>(col.Validity->Ptr) : nullptr;
            dc.chars = col.Chars ? static_cast<const uint8_t*>(col.Chars->Ptr) : nullptr;
            dc.type = col.Field.Type;
            dc.chars_bytes = col.Chars ? int32_t(std::min<size_t>(col.Chars->Size, size_t(INT32_MAX))) : 0;
            dc.val_bit_off = col.ValBitOff;
            dc.valid_bit_off = col.ValidBitOff;
            const DevColumn& first = args.cols[c];
            if (dc.type != first.type || (dc.validity != nullptr) != (first.validity != nullptr) || col.Dict) {
                return Status::Unsupported("slices in one launch: differing schemas");
            }
            if (args.lds_col_off[c] >= 0 && (reinterpret_cast<uintptr_t>(dc.values) & 15)) return Status::Unsupported("slices in one launch: an unaligned column");
            if (args.lds_chars_off[c] >= 0 && (reinterpret_cast<uintptr_t>(dc.chars) & 15)) return Status::Unsupported("slices in one launch: an unaligned column");
        }
        if (i < many.Checks.size()) {
            if (many.Checks[i].size() != size_t(args.nchecks)) return Status::Invalid("internal: offset checks of a slice");
            for (size_t k = 0; k < many.Checks[i].size(); ++k) d.checks[k] = many.Checks[i][k];
        } else if (args.nchecks != 0) {
            return Status::Invalid("internal: offset checks of a slice");
        }
    }
    if (many.HostMap.empty()) many.HostMap.push_back(0);  // (no rows at all: one workgroup that finds no tile)
    descs = Buffer::Alloc(n * sizeof(SliceDesc) + 64);
    map = Buffer::Alloc(many.HostMap.size() * 4 + 64);
    if (!descs || !map) return Status::Hip("out of device memory");
    SSA_HIP(hipMemcpyAsync(descs->Ptr, many.HostDescs.data(), n * sizeof(SliceDesc), hipMemcpyHostToDevice, StreamHandle));
    SSA_HIP(hipMemcpyAsync(map->Ptr, many.HostMap.data(), many.HostMap.size() * 4, hipMemcpyHostToDevice, StreamHandle));
    args.slices = static_cast<const SliceDesc*>(descs->Ptr);
    args.block_slice = static_cast<const int32_t*>(map->Ptr);
    cfg.grid = int(many.HostMap.size());
    return Status::OK();
}

// Bytes a scan of `in` has to read: every source column of the plan once.
static int64_t AlgorithmicBytes(const StepPlan& plan, const Batch& in) {
    int64_t bytes = 0;
    for (int ci : plan.SourceColumns) {
        const Column& col = in.Cols[size_t(ci)];
        if (IsStringLike(col.Field.Type)) {
            bytes += (in.Rows + 1) * 4;
            bytes += col.Chars ? int64_t(col.Chars->Size > 16 ? col.Chars->Size - 16 : 0) : 0;
        } else if (col.Field.Type == T_BOOL) {
            bytes += (in.Rows + 7) / 8;
        } else {
            bytes += in.Rows * TypeWidth(col.Field.Type);
        }
        if (col.Validity) bytes += (in.Rows + 7) / 8;
    }
    return bytes;
}

// `many`: the scan covers several slices of one schema in one launch (`in` is the first of them, the plan and the LDS
// layout are made for it); the result holds the groups of all slices, slice after slice.
std::atomic<int64_t> gKeyedStepsDeferred{0};
std::atomic<int64_t> gRadixCursorOverflows{0};
std::atomic<int64_t> gKeyedStepsCompacted{0};

Status Executor::RunKeyedScan(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out,
                              const std::vector<OffsetCheck>& checks, bool* notUniform, ManySlices* many) {
    *notUniform = false;
    const size_t nslices = many ? many->In.size() : 1;
    int64_t totalRows = in.Rows, maxRows = in.Rows;
    if (many) {
        totalRows = maxRows = 0;
        for (const Batch* b : many->In) {
            totalRows += b->Rows;
            maxRows = std::max(maxRows, b->Rows);
        }
        if (plan.NeedsFilteredPositions && plan.HasFilter) return Status::Unsupported("slices in one launch: a step that filters first");
        for (const AggPlan& ap : plan.Aggs) {
            if (ap.SomeInputColumn >= 0) return Status::Unsupported("slices in one launch: an aggregate that gathers input rows");
        }
    }
    if (plan.NeedsFilteredPositions && plan.HasFilter) {
        // The result depends on where the rows stand in the filtered batch: filter first, as the reference does
        // (TProgramStep::ApplyFilters, program.cpp:798-832), then aggregate the compacted columns.
        StepPlan filterPlan;
        SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_FILTER_ONLY, filterPlan));
        BufferPtr bitmap, tileOffsets;
        int64_t selected = 0;
        SSA_RETURN_NOT_OK(RunFilterBitmap(filterPlan, in, bitmap, tileOffsets, selected));
        StepPlan rest;
        SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_OUTPUTS_ONLY, rest));
        rest.HasFilter = false;
        Batch compact;
        compact.Rows = selected;
        compact.Cols.resize(in.Cols.size());
        for (size_t i = 0; i < in.Cols.size(); ++i) {
            compact.Cols[i].Field = in.Cols[i].Field;
            compact.Cols[i].Length = selected;
        }
        for (int ci : rest.SourceColumns) {
            if (selected == in.Rows) {
                compact.Cols[size_t(ci)] = in.Cols[size_t(ci)];
            } else {
                SSA_RETURN_NOT_OK(CompactColumn(in.Cols[size_t(ci)], in.Rows, bitmap, tileOffsets, selected, compact.Cols[size_t(ci)]));
                // (the plan was made for the input's schema: a column keeps its validity slot even when no null is left)
                compact.Cols[size_t(ci)].Field = in.Cols[size_t(ci)].Field;
            }
        }
        return RunKeyedScan(step, rest, compact, out, checks, notUniform);
    }
    const int naggsUser = int(plan.Aggs.size());
    const int nkeys = int(plan.Keys.size());
    int keyBits = 0;
    bool hashed = false;
    for (const KeyPlan& kp : plan.Keys) {
        if (IsStringLike(kp.Spec.type)) hashed = true;
        else keyBits += 8 * TypeWidth(kp.Spec.type) + (kp.Spec.nullable ? 1 : 0);
    }
    if (keyBits > 119) hashed = true;
    if (hashed) {
        for (const KeyPlan& kp : plan.Keys) {
            if (kp.InputColumn < 0) return Status::Unsupported("computed GROUP BY key next to string or wide keys");
        }
    }
    const bool twoWords = hashed || keyBits > 63;
    if (many && hashed) return Status::Unsupported("slices in one launch: string or wide keys");

    // Aggregates that read the same operand the same way share one accumulator (COUNT(a), COUNT(b) and COUNT(*) over
    // non-null columns are one counter).
    std::vector<AggSpec> phys;
    std::vector<int> aggSrc;
    aggSrc.resize(size_t(naggsUser));
    for (int k = 0; k < naggsUser; ++k) {
        AggSpec spec = plan.Aggs[size_t(k)].Spec;
        if (spec.kind == AGG_COUNT && !spec.check_valid) {  // counts rows, whatever its argument
            spec.reg = -1;
            spec.arg_class = CLS_U;
        }
        int found = -1;
        for (size_t j = 0; j < phys.size(); ++j) {
            if (phys[j].kind == spec.kind && phys[j].reg == spec.reg && phys[j].check_valid == spec.check_valid &&
                phys[j].arg_class == spec.arg_class && phys[j].flags == spec.flags) found = int(j);
        }
        if (found < 0) {
            found = int(phys.size());
            phys.push_back(spec);
        }
        aggSrc[size_t(k)] = found;
    }
    const int naggs = int(phys.size());
    // Accumulators the sink can keep in registers go first: row counters and SUMs without a validity check (their
    // operand layout is checked once the LDS plan is known).
    {
        std::vector<int> order;
        auto simple = [](const AggSpec& a) {
            return !a.check_valid && ((a.kind == AGG_COUNT && a.reg == -1) || a.kind == AGG_SUM_F || a.kind == AGG_SUM_I);
        };
        for (int pass = 0; pass < 2; ++pass) {
            for (int j = 0; j < naggs; ++j) {
                if (simple(phys[size_t(j)]) == (pass == 0)) order.push_back(j);
            }
        }
        std::vector<AggSpec> sorted;
        std::vector<int> newIndex;
        newIndex.resize(size_t(naggs));
        for (int j = 0; j < naggs; ++j) {
            sorted.push_back(phys[size_t(order[size_t(j)])]);
            newIndex[size_t(order[size_t(j)])] = j;
        }
        phys = sorted;
        for (int& src : aggSrc) src = newIndex[size_t(src)];
    }

    KeyedHint& hint = KeyedHints[HintOverride ? HintOverride : &step];
    const uint64_t maxCap = std::max<uint64_t>(1024, NextPow2(uint64_t(maxRows) * 2 + 2));
    uint64_t cap = std::min<uint64_t>(std::max<uint64_t>(hint.TableCap, 1024), maxCap);
    // A step's first scan knows nothing about its groups: a table of a slot per 32 rows (4 M slots at most) costs
    // microseconds to set up, and when it is too small the scan gives up early (KeyedSink.abort_check) and is repeated
    // with 16 times the slots -- two attempts for a hundred million rows of distinct keys instead of five whole scans.
    if (hint.Groups == 0 && !many) cap = std::min<uint64_t>(std::max<uint64_t>(cap, std::min<uint64_t>(NextPow2(uint64_t(maxRows) / 32 + 1), uint64_t(1) << 22)), maxCap);
    const int abortCheck = (hint.Groups == 0 || hint.ManyGroups) ? 1 : 0;

    // Level-1 accumulators are replicated only when the table path carries the bulk of the updates: steps whose
    // aggregates all run on the register path touch LDS accumulators just for their fifth and later groups.
    bool allSimple = true;
    for (const AggSpec& a : phys) {
        allSimple = allSimple && !a.check_valid && ((a.kind == AGG_COUNT && a.reg == -1) || a.kind == AGG_SUM_F || a.kind == AGG_SUM_I);
    }
    int L = hint.LdsCap, R = 1;
    if (const char* env = std::getenv("SSA_KEYED_MANY")) {  // experiments: level-1 slots of groupings with many groups
        if (hint.ManyGroups && std::atoi(env) >= 16) L = std::atoi(env);
    }
    if (L == 16) {
        if (allSimple && naggs <= kHotAggs) {
            L = 64;  // unreplicated accumulators are small: spend LDS on fewer first-probe collisions instead
        } else {
            R = 8;
            while (R > 1 && L * R * naggs * 8 > 16 * 1024) R >>= 1;
        }
    }
    if (const char* env = std::getenv("SSA_KEYED_LDS")) {
        int l = 0, r = 1;
        if (std::sscanf(env, "%d,%d", &l, &r) >= 1 && (l & (l - 1)) == 0 && r >= 1 && (r & (r - 1)) == 0) {
            L = l;
            R = l ? r : 1;
        }
    }
    if (L == 0) R = 1;
    const int sinkLds = L * 24 + 64 + naggs * L * R * 8 + L * 8 + 16;
    int sinkExtra = 0;  // LDS behind the sink's own: the row list of the kernel that takes the rows behind a filter one by one

    // Scan arguments and sink description for a kernel that takes `rowsPerThread` rows per thread (the LDS plan, and with
    // it every operand offset in the sink, belongs to one tile size).
    int64_t algoBytes = 0;
    auto describe = [&](int rowsPerThread, ScanArgs& args, LaunchCfg& cfg, KeyedSink& sink) -> Status {
        ScanRowsPerThread = rowsPerThread;
        Status st = FillScanArgs(plan, in, sinkLds + sinkExtra, hashed ? SCAN_KEYED_HASHED : SCAN_KEYED_EXACT, args, cfg, algoBytes);
        ScanRowsPerThread = kRowsPerThread;
        if (!st.ok()) return st;
        args.nchecks = int32_t(checks.size());
        for (size_t i = 0; i < checks.size(); ++i) {
            args.checks[i] = checks[i];
            algoBytes += (in.Rows + 1) * 4;  // the offsets are read (and checked), not staged
        }
        if (many) {
            algoBytes = 0;
            for (const Batch* b : many->In) algoBytes += AlgorithmicBytes(plan, *b) + int64_t(checks.size()) * (b->Rows + 1) * 4;
        }
        sink = KeyedSink{};
        sink.use_hot = 1;
        sink.naggs = naggs;
        sink.nkeys = nkeys;
        for (int k = 0; k < naggs; ++k) {
            sink.aggs[k] = phys[size_t(k)];
            if (sink.aggs[k].reg != -1) EncodeOperand(args, sink.aggs[k].reg, sink.aggs[k].fetch_kind, sink.aggs[k].fetch_payload);
        }
        for (int k = 0; k < nkeys; ++k) {
            sink.keys[k] = plan.Keys[size_t(k)].Spec;
            if (sink.keys[k].col < 0) EncodeOperand(args, sink.keys[k].reg, sink.keys[k].fetch_kind, sink.keys[k].fetch_payload);
        }
        sink.two_words = twoWords;
        sink.hashed = hashed;
        sink.rec_raw64 = (!hashed && keyBits == 64) ? 1 : 0;
        sink.lds_cap = L;
        sink.lds_rep = R;
            for (int k = 0; k < kHotAggs && k < naggs; ++k) {
            const AggSpec& a = sink.aggs[k];
            if (a.check_valid) continue;
            if (a.kind == AGG_COUNT && a.reg == -1) {
                sink.hot_kind[k] = HOT_ROWS;
                continue;
            }
            if (a.kind != AGG_SUM_F && a.kind != AGG_SUM_I) continue;
            int off = -1;
            if (a.reg >= 0) {
                off = args.lds_reg_offs[a.reg];  // R registers hold 64-bit payloads
            } else if (a.reg <= -2 && (a.fetch_kind & 0x7F) == FK_LDS_64) {
                const int t = args.cols[-2 - a.reg].type;
                if ((a.kind == AGG_SUM_F) == (t == T_F64)) off = int(a.fetch_payload & 0xFFFFFFu);
            }
            if (off < 0) continue;
            sink.hot_kind[k] = a.kind == AGG_SUM_F ? HOT_SUM_F64 : HOT_SUM_I64;
            sink.hot_off[k] = off;
        }
        // A step qualifies for the kernel that takes the rows behind the filter one by one when every operand of the sink
        // is a staged column without a validity bitmap (or a constant, or none): that kernel reads them from the staged
        // tiles at the row's index.  The tiles are listed (eight distinct ones at most); compact_n stays 0 otherwise.
        sink.compact_n = 0;
        if (plan.HasFilter && !hashed && !many) {
            bool can = true;
            auto add = [&](uint32_t fk, uint32_t payload) {
                int w = 0;
                switch (fk) {
                    case FK_NONE: case FK_CONST: return;
                    case FK_LDS_64: w = 8; break;
                    case FK_LDS_I32: case FK_LDS_U32: case FK_LDS_F32: w = 4; break;
                    case FK_LDS_I16: case FK_LDS_U16: w = 2; break;
                    case FK_LDS_I8: case FK_LDS_U8: w = 1; break;
                    default: can = false; return;  // registers, unstaged columns, anything with a validity bitmap
                }
                const uint32_t off = payload & 0xFFFFFFu;
                for (int i = 0; i < sink.compact_n; ++i) {
                    if (sink.compact_off[i] == off) return;
                }
                if (sink.compact_n == kMaxCompact) {
                    can = false;
                    return;
                }
                sink.compact_off[sink.compact_n] = off;
                sink.compact_w[sink.compact_n++] = w;
            };
            for (int k = 0; k < naggs && can; ++k) {
                const AggSpec& a = sink.aggs[k];
                if (a.kind == AGG_MIN_S || a.kind == AGG_MAX_S) can = false;
                if (a.reg != -1) add(a.fetch_kind, a.fetch_payload);
            }
            for (int k = 0; k < nkeys && can; ++k) {
                if (sink.keys[k].col >= 0) can = false; else add(sink.keys[k].fetch_kind, sink.keys[k].fetch_payload);
            }
            if (!can) sink.compact_n = 0;
        }
        sink.compact_slots = sinkExtra > 0 ? ((sinkLds + 15) & ~15) : 0;

        return Status::OK();
    };
    ScanArgs args;
    LaunchCfg cfg;
    KeyedSink sink{};
    SSA_RETURN_NOT_OK(describe(kRowsPerThread, args, cfg, sink));
    // The kernel built for this step program may take another number of rows per thread (experiments: SSA_KEYED_U; with
    // string keys 2 rows per thread measured 20 % slower than 4 on TPC-H Q1 -- the per-tile work outweighs the occupancy).
    int specRows = kRowsPerThread;
    // narrow rows with exact keys (a filter and a small integer key): 8 rows per thread, as the key-less kernels do
    if (!hashed && plan.NumRegsUsed == 0 && WideRowsPerThread(plan, in) >= 8) specRows = 8;
    if (const char* env = std::getenv("SSA_KEYED_U")) specRows = std::atoi(env);
    if (specRows != 1 && specRows != 2 && specRows != 4 && specRows != 8 && specRows != 16 && specRows != 32) specRows = kRowsPerThread;
    ScanArgs sargs = args;
    LaunchCfg scfg = cfg;
    KeyedSink ssink = sink;
    if (specRows != kRowsPerThread && JitEnabled()) {
        if (!describe(specRows, sargs, scfg, ssink).ok()) {
            specRows = kRowsPerThread;
            sargs = args;
            scfg = cfg;
            ssink = sink;
        }
    } else {
        specRows = kRowsPerThread;
    }

    HostPhases::Mark("describe");
    if (std::getenv("SSA_DEBUG_PLAN")) {
        std::fprintf(stderr, "keyed: naggs %d L %d R %d lds %u grid %d hashed %d two %d hot", naggs, L, R, cfg.lds_bytes, cfg.grid, int(hashed), int(twoWords));
        for (int k = 0; k < kHotAggs; ++k) std::fprintf(stderr, " %d@%d", sink.hot_kind[k], sink.hot_off[k]);
        std::fprintf(stderr, "\n");
    }
    BufferPtr tLo, tHi, tAcc, tSeen, tRep;
    BufferPtr sliceDescs, blockSlices;  // a launch over slices: its descriptors on the device
    // Groupings whose table is far larger than the caches: the scan leaves the rows that miss level 1 as records, which
    // are then sorted by table region and applied region by region (DeferredArgs).  SSA_KEYED_PARTITION: 0 never, 1 always.
    bool deferWanted = false;
    bool deferTimed = false;  // this scan is one of the two that decide hint.DeferChoice
    int nvals = 0;
    bool needMask = false;
    {
        const char* env = std::getenv("SSA_KEYED_PARTITION");
        const int mode = env ? std::atoi(env) : -1;
        bool can = !many && JitEnabled() && in.Rows > 0 && in.Rows < (int64_t(1) << 31) && naggs <= 31;
        for (const AggSpec& a : phys) {
            if (a.kind == AGG_MIN_S || a.kind == AGG_MAX_S) can = false;  // (their accumulators are rows compared by their strings)
            if (a.kind != AGG_COUNT) ++nvals;
            if (a.check_valid || a.kind == AGG_MIN_F || a.kind == AGG_MAX_F) needMask = true;
        }
        // (the records and their sorted copy: at most 32 GB of HBM -- half as much again while the sort's first round gives
        // its digits room to spare and its second writes to arrays of its own)
        const double recordBytes = 2.0 * double(in.Rows) * (8 + (twoWords ? 8 : 0) + 8 * nvals + (needMask ? 4 : 0) + (hashed ? 4 : 0));
        can = can && recordBytes <= 32e9;
        deferWanted = can && mode != 0 && (mode == 1 || (hint.Groups >= (int64_t(1) << 20) && in.Rows >= (int64_t(1) << 22)));
        if (deferWanted && mode != 1) {
            // (register-held groups are no use among a million groups: not worth two timed scans of their own)
            if (hint.HotChoice < 0) hint.HotChoice = 0;
            if (hint.DeferChoice == 0) {
                deferWanted = false;
            } else if (hint.DeferChoice < 0) {
                deferTimed = true;
                deferWanted = hint.MsDeferOn == 0;  // the sorted way first, then the direct one
            }
        }
        // (probing inside LDS is cheap: the table may be up to three quarters full, and everything that walks over it is
        // the faster for it)
        if (deferWanted && hint.Groups > 0) {
            // (a table in LDS costs no traffic: the pass that emits from LDS gets tables a fifth to three eighths full, where most
            // keys sit in the first slot looked at)
            const bool sparse = !hashed && !(std::getenv("SSA_KEYED_EMIT") && std::atoi(std::getenv("SSA_KEYED_EMIT")) == 0);
            const uint64_t want = sparse ? uint64_t(hint.Groups) * 8 / 3 : uint64_t(hint.Groups) + uint64_t(hint.Groups) / 3;
            cap = std::min<uint64_t>(std::max<uint64_t>(NextPow2(want), 1024), sparse ? maxCap * 2 : maxCap);
        }
    }
    BufferPtr recLo, recHi, recVal, recMask;
    // slots of a region: what one workgroup holds in 48 KB of LDS (SSA_KEYED_REGION / SSA_KEYED_REGION_KB: experiments and tests)
    int regionSlots = 4096;
    size_t regionLds = 48 * 1024;
    if (const char* env = std::getenv("SSA_KEYED_REGION_KB")) regionLds = size_t(std::max(8, std::min(128, std::atoi(env)))) * 1024;
    while (regionSlots > 256 && size_t(regionSlots) * (8 + (twoWords ? 8 : 0) + (hashed ? 8 : 0) + 8 * size_t(naggs) + 4) > regionLds) regionSlots >>= 1;
    // (the pass that emits from LDS keeps 64-bit keys as one word)
    int emitSlots = 4096;
    {
        const bool rawKey = !hashed && keyBits == 64;
        bool anyCheck = false;
        for (const AggSpec& a : phys) anyCheck = anyCheck || a.check_valid != 0;
        const size_t slotBytes = hashed ? size_t(8 + 8 + 8 + 8 * naggs + 4) : size_t(8 + (twoWords && !rawKey ? 8 : 0) + 8 * naggs + (anyCheck ? 4 : 0) + 2);
        // (exact keys, tables of 80 KB: the pass is bound by its instructions, not by the workgroups a CU holds, and the work
        // a region costs whatever its records -- set-up, counting and listing its groups -- is paid half as often)
        const size_t emitLds = std::getenv("SSA_KEYED_REGION_KB") ? regionLds : (hashed ? regionLds : size_t(80 * 1024));
        while (emitSlots > 256 && size_t(emitSlots + 1) * slotBytes > emitLds) emitSlots >>= 1;
    }
    if (const char* env = std::getenv("SSA_KEYED_REGION")) {
        const int r = std::atoi(env);
        if (r >= 64 && r <= regionSlots && (r & (r - 1)) == 0) regionSlots = r;
        if (r >= 64 && r <= emitSlots && (r & (r - 1)) == 0) emitSlots = r;
    }
    const int specCompactTiles = ssink.compact_n;  // (the generic kernel never takes that way)
    sink.compact_n = 0;
    // The way the kernel built for the step runs this scan: as decided, else the way not yet timed, in the order 1, 0, 2.
    int hotMode = 1;
    bool timedWanted = false;
    if (JitEnabled() && totalRows > 0) {
        const bool canCompact = specCompactTiles > 0 && !deferWanted;
        if (const char* env = std::getenv("SSA_KEYED_HOT")) hint.HotChoice = std::max(0, std::min(2, std::atoi(env)));
        if (hint.HotChoice == 2 && !canCompact) hint.HotChoice = 0;
        hotMode = hint.HotChoice;
        if (hotMode < 0) {  // the way with the fewest rows timed so far, in the order 1, 0, 2
            hotMode = 1;
            for (int way : {0, 2}) {
                if ((way != 2 || canCompact) && hint.TunedRows[way] < hint.TunedRows[hotMode]) hotMode = way;
            }
        }
        timedWanted = hint.HotChoice < 0 && totalRows >= (int64_t(1) << 16) && (!many || maxRows >= (int64_t(1) << 16));
        if (hint.HotChoice < 0 && !timedWanted) hotMode = 1;
        if (hotMode == 2 && !canCompact) hotMode = 0;
        // Row by row behind the filter, the per-tile work is what is left to shrink: such a kernel takes 16 rows per thread
        // when the rows are narrow enough.
        const int wideRows = WideRowsPerThread(plan, in);
        if (hotMode == 2) {
            // (and it is given room for the list of the rows that passed: two bytes per row of a tile)
            const int rowsPer = (specRows < wideRows && wideRows >= 16 && !std::getenv("SSA_KEYED_U")) ? wideRows : specRows;
            ScanArgs wargs;
            LaunchCfg wcfg;
            KeyedSink wsink{};
            sinkExtra = std::getenv("SSA_NO_COMPACT_SLOTS") ? 0 : 16 + kBlock * rowsPer * 2;
            const bool fits = describe(rowsPer, wargs, wcfg, wsink).ok() && wsink.compact_n > 0;
            sinkExtra = 0;
            if (fits) {
                sargs = wargs;
                scfg = wcfg;
                ssink = wsink;
                specRows = rowsPer;
            } else if (rowsPer != specRows) {
                if (describe(rowsPer, wargs, wcfg, wsink).ok() && wsink.compact_n > 0) {  // (the wide tile without the list)
                    sargs = wargs;
                    scfg = wcfg;
                    ssink = wsink;
                    specRows = rowsPer;
                }
            }
        }
    }
    uint64_t totalCap = 0;
    // A scan that leaves records: its groups are built region by region in LDS and emitted from there (RegionEmitKernel),
    // unless SSA_KEYED_EMIT=0 asks for the table in HBM brought up to date region by region (RegionApplyKernel).
    bool emitWanted = !(std::getenv("SSA_KEYED_EMIT") && std::atoi(std::getenv("SSA_KEYED_EMIT")) == 0);
    bool emitNow = false, emitted = false;
    int ldsSlots = 0, smallSlots = 0;
    uint64_t regionCount = 0;
    // Groups per tile of table slots, scanned: where each tile's groups go in the output, and how many there are (the
    // count arrives with the next synchronisation of the stream).
    int64_t ntilesT = 0;
    BufferPtr tileCounts, tileOffsets, counter;
    int64_t ngroups = 0;
    int64_t* ngroupsAt = HostWords ? reinterpret_cast<int64_t*>(HostWords + 8) : &ngroups;  // (pinned: no staging copy)
    bool counted = false;
    auto countGroups = [&]() -> Status {
        ntilesT = int64_t(totalCap / kTileRows);
        tileCounts = Buffer::Alloc(size_t(ntilesT) * 4 + 64);
        tileOffsets = Buffer::Alloc(size_t(ntilesT + 1 + (ntilesT + 4095) / 4096 + 2) * 8 + 64);
        counter = Buffer::Alloc(256);
        if (!tileCounts || !tileOffsets || !counter) return Status::Hip("out of device memory");
        SSA_HIP(LaunchKeyedTileCount(sink.t_key_lo, totalCap, static_cast<uint32_t*>(tileCounts->Ptr), StreamHandle));
        SSA_HIP(LaunchTileScan(static_cast<const uint32_t*>(tileCounts->Ptr), ntilesT, static_cast<int64_t*>(tileOffsets->Ptr),
                               static_cast<int64_t*>(counter->Ptr), StreamHandle));
        SSA_HIP(hipMemcpyAsync(ngroupsAt, counter->Ptr, 8, hipMemcpyDeviceToHost, StreamHandle));
        if (many) {  // where each slice's groups begin: the offsets of the first tile of each table
            many->GroupStart.assign(nslices + 1, 0);
            const size_t tilesPerSlice = size_t(ntilesT) / nslices;
            SSA_HIP(hipMemcpy2DAsync(many->GroupStart.data(), 8, tileOffsets->Ptr, tilesPerSlice * 8, 8, nslices, hipMemcpyDeviceToHost, StreamHandle));
        }
        return Status::OK();
    };
    // The output columns, with room for `rowsCap` groups, and what the kernels that fill them need to know (`fa`; the table
    // it reads, where there is one, is set by the caller).
    KeyedFinalizeArgs fa{};
    std::vector<Column> aggCols, keyCols;
    std::vector<BufferPtr> aggValid, aggRowIdx, keyValid;
    BufferPtr repRows;
    auto prepareOutputs = [&](int64_t rowsCap) -> Status {
        fa = KeyedFinalizeArgs{};
        fa.naggs = naggsUser;
        fa.nkeys = nkeys;
        fa.two_words = twoWords;
        fa.hashed = hashed;
        aggCols.assign(size_t(naggsUser), Column{});
        keyCols.assign(size_t(nkeys), Column{});
        aggValid.assign(size_t(naggsUser), nullptr);
        aggRowIdx.assign(size_t(naggsUser), nullptr);
        keyValid.assign(size_t(nkeys), nullptr);
        if (hashed) {
            repRows = Buffer::Alloc(size_t(rowsCap) * 8 + 16);
            if (!repRows) return Status::Hip("out of device memory");
            fa.group_rep_row = static_cast<uint64_t*>(repRows->Ptr);
        }
        for (int k = 0; k < naggsUser; ++k) {
            const AggPlan& ap = plan.Aggs[size_t(k)];
            fa.aggs[k] = ap.Spec;
            fa.agg_src[k] = aggSrc[size_t(k)];
            fa.agg_nan_src[k] = ap.NanTag >= 0 ? aggSrc[size_t(ap.NanTag)] : -1;
            fa.agg_zero_src[k] = ap.ZeroTag >= 0 ? aggSrc[size_t(ap.ZeroTag)] : -1;
            Column& col = aggCols[size_t(k)];
            col.Field.Type = ap.OutType;
            col.Length = rowsCap;
            fa.agg_out_type[k] = ap.OutType == T_DEC128 ? int32_t(T_U64) : ap.OutType;  // decimal: the low-part sum
            if (ap.SomeInputColumn >= 0) {
                aggRowIdx[size_t(k)] = Buffer::Alloc(size_t(rowsCap) * 8 + 16);
                if (!aggRowIdx[size_t(k)]) return Status::Hip("out of device memory");
                fa.agg_row_index[k] = 1;
                fa.agg_vals[k] = aggRowIdx[size_t(k)]->Ptr;
                continue;
            }
            col.Values = Buffer::Alloc(size_t(rowsCap) * 8 + 16);
            if (!col.Values) return Status::Hip("out of device memory");
            fa.agg_vals[k] = col.Values->Ptr;
            if (ap.Spec.kind != AGG_COUNT && ap.Spec.check_valid) {
                aggValid[size_t(k)] = Buffer::Alloc(size_t(rowsCap) + 16);
                if (!aggValid[size_t(k)]) return Status::Hip("out of device memory");
                fa.agg_valid_bytes[k] = static_cast<uint8_t*>(aggValid[size_t(k)]->Ptr);
            }
        }
        for (int k = 0; k < nkeys; ++k) {
            fa.keys[k] = sink.keys[k];
            if (hashed) continue;
            const KeyPlan& kp = plan.Keys[size_t(k)];
            Column& col = keyCols[size_t(k)];
            col.Field.Type = kp.Spec.type;
            col.Length = rowsCap;
            col.Values = Buffer::Alloc(size_t(rowsCap) * TypeWidth(kp.Spec.type) + 16);
            if (!col.Values) return Status::Hip("out of device memory");
            fa.key_vals[k] = col.Values->Ptr;
            if (kp.Spec.nullable) {
                keyValid[size_t(k)] = Buffer::Alloc(size_t(rowsCap) + 16);
                if (!keyValid[size_t(k)]) return Status::Hip("out of device memory");
                fa.key_valid_bytes[k] = static_cast<uint8_t*>(keyValid[size_t(k)]->Ptr);
            }
        }
        return Status::OK();
    };
    // What the launch that counts and emits small tables does not do: gathers of input rows (string or wide keys, some()),
    // decimal sums, null results (their validity is packed and counted by launches of its own).
    bool smallEmitCan = HostWords != nullptr && !hashed && !std::getenv("SSA_NO_SMALL_EMIT");
    for (const AggPlan& ap : plan.Aggs) {
        if (ap.DecMid >= 0 || ap.SomeInputColumn >= 0 || (ap.Spec.kind != AGG_COUNT && ap.Spec.check_valid)) smallEmitCan = false;
    }
    for (const KeyPlan& kp : plan.Keys) smallEmitCan = smallEmitCan && !kp.Spec.nullable;
    bool smallEmitted = false, smallNow = false;
    for (;;) {
        bool timedRun = timedWanted;
        bool deferNow = false;
        bool specUsed = false;
        std::chrono::steady_clock::time_point deferT0;
        // a scan of many rows runs the kernel built for this step program, when that can be had
        hipFunction_t spec = nullptr;
        LaunchCfg specCfg = scfg;
        if (totalRows > 0) {
            if (JitEnabled()) {
                ssink.use_hot = hotMode == 1 ? 1 : 0;
                if (hotMode != 2) ssink.compact_n = 0;
                KeyedSink image = ssink;
                image.t_key_lo = image.t_key_hi = image.t_acc = nullptr;
                image.t_seen = nullptr;
                image.t_rep_row = nullptr;
                image.table_mask = 0;
                image.region_mask = 0;
                image.abort_check = 0;
                image.acc_stride = 0;
                image.rec_lo = image.rec_hi = image.rec_val = nullptr;
                image.rec_present = nullptr;
                image.rec_mask = nullptr;
                std::string defines = std::string("#define SSA_SPEC_U ") + std::to_string(specRows) + "\n#define SSA_SPEC_HASHED " +
                                      (hashed ? "1" : "0") + "\n#define SSA_SPEC_SLICES " + (many ? "1" : "0") + "\n";
                if (std::getenv("SSA_PHASE_TIMING")) defines += "#define SSA_PHASE_TIMING 1\n";  // diagnostics
                ScanRowsPerThread = specRows;
                // (more than 2^16 regions would take a third round of the sort: such tables are updated row by row)
                if (deferWanted && cap / std::min<uint64_t>(cap, uint64_t(emitWanted ? emitSlots : regionSlots)) <= 65536) {
                    image.defer = 1;
                    spec = SpecializedKernel("keyed_scan.h", hotMode == 1 ? "ssa_spec_keyed" : "ssa_spec_keyed_plain", defines + "#define SSA_SPEC_DEFER 1\n",
                                             sargs, &image, sizeof(image), totalRows, specCfg);
                    deferNow = spec != nullptr;  // (still being built: this scan updates the table row by row)
                }
                if (!deferNow) {
                    image.defer = 0;
                    specCfg = scfg;
                    if (hotMode == 2) defines += "#define SSA_SPEC_COMPACT 1\n";
                    spec = SpecializedKernel("keyed_scan.h", hotMode == 1 ? "ssa_spec_keyed" : (hotMode == 2 ? "ssa_spec_keyed_compact" : "ssa_spec_keyed_plain"),
                                             defines, sargs, &image, sizeof(image), totalRows, specCfg);
                }
                ScanRowsPerThread = kRowsPerThread;
            }
        }
        HostPhases::Mark("kernel-lookup");
        // The table in HBM.  A scan that leaves records, to be aggregated region by region and emitted from LDS (emitNow),
        // touches it only when its workgroups end and merge their own tables into it: a small table then, cut into the
        // same regions as the records (`smallSlots` slots each).
        emitNow = deferNow && emitWanted;
        uint64_t tableCap = cap;
        if (emitNow) {
            ldsSlots = int(std::min<uint64_t>(cap, uint64_t(emitSlots)));
            regionCount = cap / uint64_t(ldsSlots);
            const uint64_t entries = uint64_t(std::max(specCfg.grid, 1)) * uint64_t(std::max(L, 1));  // what the workgroups' own tables hold
            smallSlots = int(std::min<uint64_t>(uint64_t(ldsSlots), std::max<uint64_t>(16, NextPow2((4 * entries + regionCount - 1) / regionCount))));
            tableCap = regionCount * uint64_t(smallSlots);
        }
        totalCap = tableCap * nslices;  // every slice has a table of its own, one after the other
        // Small tables that the launch behind the scan counts, emits and sets back to free (SmallTablesEmitKernel) stay
        // with the step: its next scan finds them as a fresh table would be, without a launch to make them so.
        const bool timedThisRun = timedWanted && (totalRows <= 0 || spec != nullptr);
        smallNow = smallEmitCan && !timedThisRun && !deferNow && !deferTimed && tableCap <= uint64_t(kSmallTableSlots) && nslices <= size_t(kSmallEmitTables);
        bool tableReady = false;
        if (smallNow && hint.Small.Lo && hint.Small.TotalCap == totalCap && hint.Small.Naggs == naggs && hint.Small.TwoWords == twoWords) {
            tLo = hint.Small.Lo;
            tHi = hint.Small.Hi;
            tAcc = hint.Small.Acc;
            tSeen = hint.Small.Seen;
            tRep = hint.Small.Rep;
            tableReady = hint.Small.Clean;
        } else {
            tLo = Buffer::Alloc(totalCap * 8);
            tHi = Buffer::Alloc(twoWords ? totalCap * 8 : 64);
            tAcc = Buffer::Alloc(size_t(naggs) * totalCap * 8);
            tSeen = Buffer::Alloc(totalCap * 4);
            tRep = Buffer::Alloc(hashed ? totalCap * 8 : 64);
            if (!tLo || !tHi || !tAcc || !tSeen || !tRep) return Status::Hip("out of device memory");
            if (smallNow) hint.Small = KeyedHint::SmallTable{tLo, tHi, tAcc, tSeen, tRep, totalCap, naggs, twoWords, false};
        }
        if (smallNow) hint.Small.Clean = false;  // (until the launch that empties it again has been seen to end well)
        for (KeyedSink* k : {&sink, &ssink}) {
            k->table_mask = tableCap - 1;
            k->region_mask = tableCap - 1;
            k->abort_check = abortCheck;
            k->acc_stride = totalCap;
            k->t_key_lo = static_cast<uint64_t*>(tLo->Ptr);
            k->t_key_hi = static_cast<uint64_t*>(tHi->Ptr);
            k->t_acc = static_cast<uint64_t*>(tAcc->Ptr);
            k->t_seen = static_cast<uint32_t*>(tSeen->Ptr);
            k->t_rep_row = static_cast<uint64_t*>(tRep->Ptr);
        }
        if (!tableReady) {
            KeyedSink all = sink;
            all.table_mask = totalCap - 1;  // (the capacity, not a mask, when there are several tables)
            SSA_HIP(LaunchKeyedTableInit(all, twoWords != 0, hashed, StreamHandle));
        }
        HostPhases::Mark("table");
        if (totalRows > 0) {
            ssink.defer = deferNow ? 1 : 0;
            ssink.region_mask = emitNow ? uint64_t(smallSlots) - 1 : (deferNow ? std::min<uint64_t>(cap, uint64_t(regionSlots)) - 1 : cap - 1);
            if (deferNow) {
                const size_t n = size_t(in.Rows);
                if (!recLo) {
                    recLo = Buffer::Alloc(n * 8);
                    recHi = Buffer::Alloc(twoWords && !ssink.rec_raw64 ? n * 8 : (ssink.rec_raw64 ? (n / 64 + 2) * 8 : 64));  // (or the bitmap of the rows with records)
                    recVal = Buffer::Alloc(size_t(nvals) * n * 8 + 64);
                    recMask = Buffer::Alloc(needMask ? n * 4 : 64);
                    if (!recLo || !recHi || !recVal || !recMask) return Status::Hip("out of device memory");
                }
                ssink.rec_lo = static_cast<uint64_t*>(recLo->Ptr);
                ssink.rec_hi = ssink.rec_raw64 ? nullptr : static_cast<uint64_t*>(recHi->Ptr);
                ssink.rec_present = ssink.rec_raw64 ? static_cast<uint64_t*>(recHi->Ptr) : nullptr;
                ssink.rec_val = static_cast<uint64_t*>(recVal->Ptr);
                ssink.rec_mask = needMask ? static_cast<uint32_t*>(recMask->Ptr) : nullptr;
            }
            timedRun = timedRun && spec != nullptr;
            specUsed = spec != nullptr;
            if (many) {
                ScanArgs& la = spec ? sargs : args;
                LaunchCfg& lc = spec ? specCfg : cfg;
                SSA_RETURN_NOT_OK(DescribeSlices(plan, *many, la, int64_t(kBlock) * (spec ? specRows : kRowsPerThread), cap, lc, sliceDescs, blockSlices));
                HostPhases::Mark("slices");
            }
            ProfileBegin();
            if (deferTimed) {
                SSA_HIP(hipStreamSynchronize(StreamHandle));
                deferT0 = std::chrono::steady_clock::now();
            }
            if (spec) {
                if (timedRun) SSA_HIP(hipEventRecord(EvTuneA, StreamHandle));
                SSA_HIP(LaunchSpecialized(spec, sargs, &ssink, sizeof(ssink), specCfg));
                if (ssink.compact_n > 0) gKeyedStepsCompacted.fetch_add(1, std::memory_order_relaxed);
                if (timedRun) SSA_HIP(hipEventRecord(EvTuneB, StreamHandle));
            } else {
                SSA_RETURN_NOT_OK(GenericKernelCan(args));
                SSA_HIP(LaunchScanKeyed(args, sink, cfg));
            }
            ProfileEnd("ScanKeyedKernel", algoBytes);
        }
        HostPhases::Mark("scan-launch");
        if (std::getenv("SSA_PHASE_TIMING")) {  // diagnostics build only (-DSSA_PHASE_TIMING): cycles per phase, summed over waves
            unsigned long long t[8];
            SSA_HIP(hipStreamSynchronize(StreamHandle));
            SSA_HIP(hipMemcpy(t, static_cast<char*>(StatusWord->Ptr) + 64, 64, hipMemcpyDeviceToHost));
            SSA_HIP(hipMemset(static_cast<char*>(StatusWord->Ptr) + 64, 0, 64));
            std::fprintf(stderr, "phases: stage %llu vm %llu keys %llu (exact packing %llu) upsert %llu hot %llu table %llu loop %llu\n", t[0], t[1],
                         t[2] + t[6], t[6], t[3], t[4], t[5], t[7]);
        }
        // Small tables are counted and emitted right behind the scan, every table to rows of its own of output columns as
        // long as the tables: the scan's status and the group counts then come back with the one synchronisation that
        // ends the step (an apply of a shard-sized batch is made of little else than launches and waits).  When the status
        // is not zero the emitted rows are dropped.
        bool tableFull = false;
        counted = false;
        smallEmitted = false;
        if (smallNow) {
            SSA_RETURN_NOT_OK(prepareOutputs(int64_t(totalCap)));
            SmallEmitArgs se{};
            fa.capacity = totalCap;
            fa.t_key_lo = sink.t_key_lo;
            fa.t_key_hi = sink.t_key_hi;
            fa.t_acc = sink.t_acc;
            fa.t_seen = sink.t_seen;
            fa.t_rep_row = sink.t_rep_row;
            se.f = fa;
            se.slice_cap = uint32_t(tableCap);
            se.reset = 1;
            se.nacc = naggs;
            for (int k = 0; k < naggs; ++k) se.ident[k] = KeyedTableIdentity(phys[size_t(k)].kind);
            se.status = static_cast<const uint32_t*>(StatusWord->Ptr);
            se.host = reinterpret_cast<uint32_t*>(HostWords);
            SSA_HIP(LaunchSmallTablesEmit(se, int(nslices), StreamHandle));
            HostPhases::Mark("emit-launch");
            SSA_HIP(hipStreamSynchronize(StreamHandle));
            HostPhases::Mark("sync");
            SSA_RETURN_NOT_OK(DecodeScanStatus(uint32_t(HostWords[0]), &tableFull, notUniform));
            smallEmitted = !tableFull && !*notUniform;
            hint.Small.Clean = HostWords[0] == 0;
        } else {
            // A table of up to 64 K slots is counted right behind the scan: status and group count come back together.
            if (!timedRun && !deferNow && !deferTimed && totalCap <= (uint64_t(1) << 16)) {
                SSA_RETURN_NOT_OK(countGroups());
                counted = true;
            }
            SSA_RETURN_NOT_OK(CheckScanStatus(&tableFull, notUniform));
        }
        if (*notUniform) return Status::OK();
        if (timedRun && !tableFull) {  // (the stream has been synchronised by the status check)
            float ms = 0;
            if (hipEventElapsedTime(&ms, EvTuneA, EvTuneB) == hipSuccess && ms > 0) {
                hint.TunedMs[hotMode] += double(ms);
                hint.TunedRows[hotMode] += totalRows;
                const bool third = specCompactTiles > 0 && !deferWanted;  // the third way exists for this step
                constexpr int64_t kEnough = int64_t(1) << 22;
                if (hint.TunedRows[1] >= kEnough && hint.TunedRows[0] >= kEnough && (!third || hint.TunedRows[2] >= kEnough)) {
                    double perRow[3] = {0, 0, 0};
                    for (int way = 0; way < 3; ++way) perRow[way] = hint.TunedRows[way] > 0 ? hint.TunedMs[way] / double(hint.TunedRows[way]) : 0;
                    hint.HotChoice = perRow[1] <= perRow[0] ? 1 : 0;
                    if (third && perRow[2] < perRow[hint.HotChoice]) hint.HotChoice = 2;
                    if (std::getenv("SSA_DEBUG_PLAN")) {
                        std::fprintf(stderr, "keyed: ns per row plain %.4f, register-held groups %.4f, rows one by one %.4f -> %d\n", 1e6 * perRow[0],
                                     1e6 * perRow[1], 1e6 * perRow[2], hint.HotChoice);
                    }
                }
            }
        }
        if (!tableFull && deferNow && emitNow) {
            SortedRecords sorted;
            RegionGroups regionGroups;
            int64_t found = 0;
            int64_t* foundAt = HostWords ? reinterpret_cast<int64_t*>(HostWords + 8) : &found;  // (pinned: no staging copy)
            // The sort's first round without its counting pass first (a digit's place in the sorted arrays has room to
            // spare); keys that crowd into few regions run one out of room: the scan's records are still there then, and are
            // sorted again with counts -- for this step, from then on.
            uint32_t overflowed = 0;
            uint32_t* overflowAt = HostWords ? reinterpret_cast<uint32_t*>(HostWords + 12) : &overflowed;
            Status st = Status::OK();
            int64_t emittedAhead = 0;  // room of the output columns the groups were emitted into before their number was known
            for (int attempt = 0; attempt < 2; ++attempt) {
                const bool byCursors = attempt == 0 && !hint.NoRadixCursors && std::getenv("SSA_NO_RADIX_CURSORS") == nullptr;
                if (attempt == 0 && !byCursors) continue;
                sorted = SortedRecords{};
                regionGroups = RegionGroups{};
                *overflowAt = 0;
                st = SortDeferredRecords(ssink, in.Rows, nvals, needMask, 63 - __builtin_clzll(uint64_t(smallSlots)), regionCount, sorted, byCursors);
                if (st.ok()) st = AggregateDeferredRecords(ssink, sargs, sorted, in.Rows, nvals, ldsSlots, smallSlots, regionCount, regionGroups, foundAt);
                // The groups go to the output columns right behind the region pass, into columns with room for somewhat
                // more groups than the step's last scan found: their number arrives with the synchronisation below, and
                // only a scan that found more than the room is emitted again, into longer columns.
                emittedAhead = 0;
                if (st.ok() && hint.Groups > 0 && !std::getenv("SSA_NO_EMIT_AHEAD")) {
                    emittedAhead = hint.Groups + hint.Groups / 32 + 4096;
                    st = prepareOutputs(emittedAhead);
                    fa.capacity = uint64_t(emittedAhead);
                    if (st.ok()) st = EmitRegionGroups(regionGroups, fa);
                }
                if (st.ok()) {
                    uint32_t word = 0;
                    uint32_t* wordAt = HostWords ? reinterpret_cast<uint32_t*>(HostWords) : &word;
                    hipError_t e = sorted.overflow != nullptr ? hipMemcpyAsync(overflowAt, sorted.overflow, 4, hipMemcpyDeviceToHost, StreamHandle) : hipSuccess;
                    if (e == hipSuccess) e = hipMemcpyAsync(wordAt, StatusWord->Ptr, 4, hipMemcpyDeviceToHost, StreamHandle);
                    if (e == hipSuccess) e = hipStreamSynchronize(StreamHandle);  // (the count has arrived)
                    // (what the region pass said about a part of the records says nothing)
                    if (e == hipSuccess && *overflowAt != 0) e = hipMemsetAsync(StatusWord->Ptr, 0, 4, StreamHandle);
                    if (e != hipSuccess) st = Status::Hip(hipGetErrorString(e));
                    else if (*overflowAt == 0) st = DecodeScanStatus(*wordAt, &tableFull, nullptr);
                }
                if (!st.ok() || *overflowAt == 0) break;
                hint.NoRadixCursors = true;
                gRadixCursorOverflows.fetch_add(1, std::memory_order_relaxed);
                if (std::getenv("SSA_DEBUG_PLAN")) {
                    std::vector<uint32_t> words(sorted.ncursors);
                    (void)hipMemcpy(words.data(), sorted.cursors, words.size() * 4, hipMemcpyDeviceToHost);
                    uint32_t fullest = 0;
                    uint64_t sum = 0;
                    for (uint32_t w : words) fullest = std::max(fullest, w), sum += w;
                    std::fprintf(stderr, "regions: a place of the sort ran out of room (%llu records, the fullest digit %u of them, room for %u); sorting again with counts\n",
                                 (unsigned long long)sum, fullest, sorted.cursor_cap);
                }
            }
            if (st.ok() && !tableFull) {
                found = *foundAt;
                if (emittedAhead == 0 || found > emittedAhead) {
                    st = prepareOutputs(found);
                    fa.capacity = uint64_t(found);
                    if (st.ok() && found > 0) st = EmitRegionGroups(regionGroups, fa);
                }
            }
            // (the buffers of the sort and of the region pass are freed on the way out of this block: the stream has to be
            // past their last use)
            const hipError_t syncErr = hipStreamSynchronize(StreamHandle);
            if (!st.ok()) return st;
            SSA_HIP(syncErr);
            gKeyedStepsDeferred.fetch_add(1, std::memory_order_relaxed);
            if (!tableFull) {
                emitted = true;
                ngroups = found;
            }
        } else if (!tableFull && deferNow) {
            SSA_RETURN_NOT_OK(ApplyDeferredRecords(ssink, sargs, in.Rows, nvals, needMask, regionSlots, cap));
            gKeyedStepsDeferred.fetch_add(1, std::memory_order_relaxed);
            SSA_RETURN_NOT_OK(CheckScanStatus(&tableFull, nullptr));
        }
        if (!tableFull && deferTimed && specUsed && deferNow == deferWanted) {
            // (both status checks above end with the stream synchronised: scan, sort and region pass are done)
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - deferT0).count();
            (deferNow ? hint.MsDeferOn : hint.MsDeferOff) = ms;
            if (hint.MsDeferOn > 0 && hint.MsDeferOff > 0) hint.DeferChoice = hint.MsDeferOn <= hint.MsDeferOff ? 1 : 0;
            if (std::getenv("SSA_DEBUG_PLAN")) std::fprintf(stderr, "keyed: by region %.3f ms, row by row %.3f ms -> %d\n", hint.MsDeferOn, hint.MsDeferOff, hint.DeferChoice);
        }
        if (!tableFull) break;
        if (cap >= maxCap && emitNow) {  // (regions that ran full although the table is as large as it gets: the table in HBM, then)
            emitWanted = false;
            cap = maxCap;
            continue;
        }
        if (cap >= maxCap) return Status::Invalid("internal: group-by table overflow");
        cap = std::min<uint64_t>(cap * 16, maxCap);
    }

    cap = totalCap;  // (from here on the tables of all slices are one table: its slots in order are the slices in order)
    int64_t maxGroups = 0;
    if (smallEmitted) {
        // table i's groups are rows [i * slots of a table, + its count) of the output
        emitted = true;
        const uint32_t* counts = reinterpret_cast<const uint32_t*>(HostWords) + kSmallEmitCounts;
        for (size_t i = 0; i < nslices; ++i) maxGroups = std::max<int64_t>(maxGroups, counts[i]);
        ngroups = many ? int64_t(totalCap) : int64_t(counts[0]);
        if (many) {
            many->GroupStart.assign(nslices + 1, 0);
            many->GroupCount.assign(nslices, 0);
            for (size_t i = 0; i < nslices; ++i) {
                many->GroupStart[i] = int64_t(i) * int64_t(totalCap / nslices);
                many->GroupCount[i] = counts[i];
            }
            many->GroupStart[nslices] = int64_t(totalCap);
        }
    } else {
        if (!emitted) {
            if (!counted) {
                SSA_RETURN_NOT_OK(countGroups());
                SSA_HIP(hipStreamSynchronize(StreamHandle));
            }
            ngroups = *ngroupsAt;
        }
        maxGroups = ngroups;
        if (many) {
            many->GroupStart[nslices] = ngroups;
            many->GroupCount.clear();
            maxGroups = 0;
            for (size_t i = 0; i < nslices; ++i) maxGroups = std::max(maxGroups, many->GroupStart[i + 1] - many->GroupStart[i]);
        }
    }

    // next call on this step: size both table levels for what was seen
    hint.TableCap = std::min<uint64_t>(std::max<uint64_t>(NextPow2(uint64_t(maxGroups) * 2), 1024), maxCap);  // load 0.25 .. 0.5
    // Groupings of many groups keep a level-1 table too: it fills with the first 256 groups each workgroup meets, and
    // those include the heavy hitters of a skewed key (an empty search phrase, the busiest users), whose rows would
    // otherwise queue on single addresses in HBM (ClickBench SearchPhrase: 1046 ms without level 1, 3.6 ms with it).
    hint.LdsCap = maxGroups <= 8 ? 16 : 256;
    hint.ManyGroups = maxGroups > 128;
    hint.Groups = maxGroups;

    if (!emitted) {
        SSA_RETURN_NOT_OK(prepareOutputs(ngroups));
        fa.capacity = cap;
        fa.t_key_lo = sink.t_key_lo;
        fa.t_key_hi = sink.t_key_hi;
        fa.t_acc = sink.t_acc;
        fa.t_seen = sink.t_seen;
        fa.tile_offsets = static_cast<const int64_t*>(tileOffsets->Ptr);
        fa.t_rep_row = sink.t_rep_row;
        if (ngroups > 0) SSA_HIP(LaunchKeyedEmit(fa, StreamHandle));
    }
    for (Column& c : aggCols) c.Length = ngroups;
    for (Column& c : keyCols) c.Length = ngroups;
    for (int k = 0; k < naggsUser; ++k) {
        const AggPlan& ap = plan.Aggs[size_t(k)];
        if (ap.DecMid >= 0) {
            BufferPtr wide = Buffer::Alloc(size_t(ngroups) * 16 + 16);
            if (!wide) return Status::Hip("out of device memory");
            SSA_HIP(LaunchCombineDecimal(static_cast<const uint64_t*>(aggCols[size_t(k)].Values->Ptr),
                                         static_cast<const uint64_t*>(aggCols[size_t(ap.DecMid)].Values->Ptr),
                                         static_cast<const uint64_t*>(aggCols[size_t(ap.DecHi)].Values->Ptr), ngroups,
                                         static_cast<uint64_t*>(wide->Ptr), StreamHandle));
            aggCols[size_t(k)].Values = wide;
            aggCols[size_t(k)].Field.Precision = ap.Precision;
            aggCols[size_t(k)].Field.Scale = ap.Scale;
        }
        if (ap.SomeInputColumn >= 0) {
            // (on the single-byte-key path argument nulls are not looked at: the first row's value, never null)
            SSA_RETURN_NOT_OK(GatherRows(in.Cols[size_t(ap.SomeInputColumn)], aggRowIdx[size_t(k)], ngroups, ap.ArgNullable, aggCols[size_t(k)]));
        } else if (aggValid[size_t(k)]) {
            SSA_RETURN_NOT_OK(PackValidity(aggValid[size_t(k)], ngroups, aggCols[size_t(k)]));
        }
    }
    for (int k = 0; k < nkeys; ++k) {
        const KeyPlan& kp = plan.Keys[size_t(k)];
        if (hashed) {
            SSA_RETURN_NOT_OK(GatherRows(in.Cols[size_t(kp.InputColumn)], repRows, ngroups, kp.Spec.nullable != 0, keyCols[size_t(k)]));
        } else if (keyValid[size_t(k)]) {
            SSA_RETURN_NOT_OK(PackValidity(keyValid[size_t(k)], ngroups, keyCols[size_t(k)]));
        }
    }

    HostPhases::Mark("after-loop");
    out = std::make_shared<Batch>();
    out->Rows = ngroups;
    for (const OutColumn& oc : plan.Outs) {
        Column col = oc.Source == OutColumn::FROM_AGG ? aggCols[size_t(oc.Index)] : keyCols[size_t(oc.Index)];
        const bool nullable = col.Field.Nullable;
        if (oc.Source == OutColumn::FROM_KEY && plan.Keys[size_t(oc.Index)].InputColumn >= 0) {
            col.Field = in.Cols[size_t(plan.Keys[size_t(oc.Index)].InputColumn)].Field;  // keeps decimal / timestamp parameters
        }
        col.Field.Name = oc.Name;
        col.Field.Type = oc.Type;
        col.Field.Nullable = nullable;
        if (oc.Source == OutColumn::FROM_AGG && oc.Type == T_DEC128) {
            col.Field.Precision = oc.Precision;
            col.Field.Scale = oc.Scale;
        }
        out->Cols.push_back(std::move(col));
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

// A key-less step with a string MIN / MAX: the keyed sink with no key columns -- every row in one group.  Without rows
// there is no group, while the key-less aggregation still yields its one row (nulls, and zero counts).
Status Executor::RunNoKeyAsOneGroup(const TProgramStep& step, const StepPlan& plan, const Batch& in, BatchPtr& out) {
    std::vector<OffsetCheck> none;
    bool notUniform = false;
    SSA_RETURN_NOT_OK(RunKeyedScan(step, plan, in, out, none, &notUniform));
    if (out->Rows == 1) return Status::OK();
    if (out->Rows != 0) return Status::Invalid("internal: key-less aggregation produced more than one group");
    auto one = std::make_shared<Batch>();
    one->Rows = 1;
    for (const OutColumn& oc : plan.Outs) {
        Column col;
        col.Field.Name = oc.Name;
        col.Field.Type = oc.Type;
        col.Field.Precision = oc.Precision;
        col.Field.Scale = oc.Scale;
        col.Length = 1;
        const bool isCount = plan.Aggs[size_t(oc.Index)].Spec.kind == AGG_COUNT;
        col.Values = Buffer::Alloc(64);
        if (!col.Values) return Status::Hip("out of device memory");
        SSA_HIP(hipMemsetAsync(col.Values->Ptr, 0, 64, StreamHandle));
        if (IsStringLike(oc.Type)) {
            col.Chars = Buffer::Alloc(32);
            if (!col.Chars) return Status::Hip("out of device memory");
        }
        if (!isCount) {
            col.Validity = Buffer::Alloc(32);
            if (!col.Validity) return Status::Hip("out of device memory");
            SSA_HIP(hipMemsetAsync(col.Validity->Ptr, 0, 32, StreamHandle));
            col.NullCount = 1;
            col.Field.Nullable = true;
        }
        one->Cols.push_back(std::move(col));
    }
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    out = one;
    return Status::OK();
}

Status Executor::ApplyStep(const TProgramStep& step, const BatchPtr& in, BatchPtr& out) {
    for (const Column& c : in->Cols) {
        if (c.Dict) return ApplyStepDict(step, in, out);
    }
    return ApplyStepPlain(step, in, out, /*stable=*/true);
}

// The plan of `step` for the schema of `in`; kept (PlanCache) when `stable` says the step outlives the executor's calls.
Status Executor::PlanFor(const TProgramStep& step, const Batch& in, bool stable, std::shared_ptr<const StepPlan>& kept) {
    kept = nullptr;
    std::string signature;
    if (stable) {
        // everything of the input the planner looks at: names, types, which columns carry nulls, decimal parameters
        // (a scalar column's value takes part in constant folding: such batches are planned every time)
        signature.reserve(32 + 24 * in.Cols.size());
        const void* address = &step;
        signature.append(reinterpret_cast<const char*>(&address), sizeof(address));
        signature.push_back(ForceSerializedKeys ? 'S' : 'N');
        for (const Column& c : in.Cols) {
            if (c.Field.IsScalar) {
                stable = false;
                break;
            }
            signature.append(c.Field.Name);
            signature.push_back('\0');
            const int32_t words[4] = {c.Field.Type, c.Field.Nullable ? 1 : 0, c.Field.Precision, c.Field.Scale};
            signature.append(reinterpret_cast<const char*>(words), sizeof(words));
        }
        if (stable) {
            auto it = PlanCache.find(signature);
            if (it != PlanCache.end()) kept = it->second;
        }
    }
    if (!kept) {
        auto fresh = std::make_shared<StepPlan>();
        SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_FULL, *fresh, ForceSerializedKeys));
        kept = fresh;
        if (stable) {
            if (PlanCache.size() >= 64) PlanCache.clear();  // (a program sees a handful of schemas)
            PlanCache[signature] = kept;
        }
    }
    return Status::OK();
}

namespace {

// Refusals that say a step is too wide for one scan, not that it cannot run.
bool IsWidthLimit(const Status& st) {
    return st.Code == ST_UNSUPPORTED && (st.Msg.find("too many") != std::string::npos || st.Msg.find("more output columns than") != std::string::npos);
}

// `step` as `parts` steps: part p keeps the assigns its share of the aggregates (key-less aggregations) or of the
// projected columns needs, and every filter.  An assign nothing reads stays with the first part (the reference evaluates
// it, and its errors are the program's).  False: a step of this kind is not split (keyed aggregations: the parts'
// groups would have to be matched up again).
bool SplitStep(const TProgramStep& step, size_t parts, std::vector<std::shared_ptr<TProgramStep>>& out) {
    out.clear();
    if (!step.GroupByKeys.empty()) return false;
    const bool aggregates = !step.GroupBy.empty();
    if (aggregates && !step.Projection.empty()) return false;
    const size_t n = aggregates ? step.GroupBy.size() : step.Projection.size();
    if (n < 2 || parts > n) return false;
    std::vector<std::set<std::string>> needs(parts);
    std::vector<char> read(step.Assignes.size(), 0);
    for (size_t p = 0; p < parts; ++p) {
        auto part = std::make_shared<TProgramStep>();
        part->Filters = step.Filters;
        part->FilterOriginalColumnIds = step.FilterOriginalColumnIds;
        std::set<std::string>& need = needs[p];
        for (const TColumnInfo& f : step.Filters) need.insert(f.ColumnName);
        for (size_t i = p * n / parts; i < (p + 1) * n / parts; ++i) {
            if (aggregates) {
                part->GroupBy.push_back(step.GroupBy[i]);
                for (const TColumnInfo& a : step.GroupBy[i].Arguments) need.insert(a.ColumnName);
            } else {
                part->Projection.push_back(step.Projection[i]);
                need.insert(step.Projection[i].ColumnName);
            }
        }
        for (size_t i = step.Assignes.size(); i-- > 0;) {
            if (!need.count(step.Assignes[i].Column.ColumnName)) continue;
            read[i] = 1;
            for (const TColumnInfo& a : step.Assignes[i].Arguments) need.insert(a.ColumnName);
        }
        out.push_back(part);
    }
    // (the assigns nothing reads and what they read in turn: with the first part)
    for (size_t i = step.Assignes.size(); i-- > 0;) {
        if (read[i]) continue;
        needs[0].insert(step.Assignes[i].Column.ColumnName);
        for (const TColumnInfo& a : step.Assignes[i].Arguments) needs[0].insert(a.ColumnName);
    }
    for (size_t p = 0; p < parts; ++p) {
        for (const TAssign& a : step.Assignes) {
            if (needs[p].count(a.Column.ColumnName)) out[p]->Assignes.push_back(a);
        }
    }
    return true;
}

}  // namespace

Status Executor::ApplySplit(const std::vector<std::shared_ptr<TProgramStep>>& parts, const BatchPtr& in, BatchPtr& out) {
    auto whole = std::make_shared<Batch>();
    for (size_t p = 0; p < parts.size(); ++p) {
        BatchPtr piece;
        SSA_RETURN_NOT_OK(ApplyStepWhole(*parts[p], in, piece, /*stable=*/true));
        if (p == 0) whole->Rows = piece->Rows;
        if (piece->Rows != whole->Rows) return Status::Invalid("internal: the parts of a split step disagree about the rows");
        for (const Column& c : piece->Cols) whole->Cols.push_back(c);  // (a piece may be the input itself: its columns are shared, not taken)
    }
    out = whole;
    return Status::OK();
}

Status Executor::ApplyStepPlain(const TProgramStep& step, const BatchPtr& in, BatchPtr& out, bool stable) {
    if (stable) {
        auto it = SplitSteps.find(&step);
        if (it != SplitSteps.end()) return ApplySplit(it->second, in, out);
    }
    const Status st = ApplyStepWhole(step, in, out, stable);
    if (!stable || !IsWidthLimit(st)) return st;
    for (size_t parts = 2; parts <= 64; parts *= 2) {
        std::vector<std::shared_ptr<TProgramStep>> pieces;
        if (!SplitStep(step, parts, pieces)) break;
        const Status sp = ApplySplit(pieces, in, out);
        if (sp.ok()) {
            if (std::getenv("SSA_DEBUG_PLAN")) std::fprintf(stderr, "step: too wide for one scan (%s), run as %zu scans\n", st.Msg.c_str(), parts);
            SplitSteps[&step] = std::move(pieces);
            return sp;
        }
        // (plans are kept by step address: the steps of an attempt that was given up are never freed, so that no later
        // step takes their addresses and their plans)
        RetiredSteps.insert(RetiredSteps.end(), pieces.begin(), pieces.end());
        if (!IsWidthLimit(sp)) return sp;
    }
    return st;
}

Status Executor::ApplyStepWhole(const TProgramStep& step, const BatchPtr& in, BatchPtr& out, bool stable) {
    std::shared_ptr<const StepPlan> kept;
    SSA_RETURN_NOT_OK(PlanFor(step, *in, stable, kept));
    StableStep = stable;
    const StepPlan& plan = *kept;
    switch (plan.Kind) {
        case StepPlan::PASS_THROUGH:
        case StepPlan::PROJECT:
            return RunProject(step, plan, in, out);
        case StepPlan::AGG_NOKEY:
            if (plan.HasStringMinMax) return RunNoKeyAsOneGroup(step, plan, *in, out);
            return RunNoKey(plan, *in, out);
        case StepPlan::AGG_KEYED:
            return RunKeyed(step, plan, *in, out);
        default:
            return Status::Invalid("internal: unexpected plan kind");
    }
}

Status Executor::ApplyProgram(const TProgram& program, const BatchPtr& in, BatchPtr& out) {
    HostPhases::Mark("begin");
    (void)hipSetDevice(Device);
    BatchPtr cur = in;
    for (auto& step : program.Steps) {
        BatchPtr next;
        SSA_RETURN_NOT_OK(ApplyStep(*step, cur, next));
        cur = next;
    }
    out = cur;
    HostPhases::Mark("end");
    HostPhases::Report("apply:");
    return Status::OK();
}

// No device memory is allocated or freed here: the generic scan kernel (a kernel built for the program would have to
// be loaded), the caller's bitmap, the static scratch buffer, the status word made with the executor.
Status Executor::FilterFromStepNoAlloc(const TProgramStep& step, const Batch& in, uint64_t* outBits, int64_t& selected) {
    (void)hipSetDevice(Device);
    selected = 0;
    if (in.Rows > kNoAllocMaxRows) {
        return Status::Invalid("ssa_filter_from_program_noalloc takes at most " + std::to_string(kNoAllocMaxRows) + " rows, the batch has " +
                               std::to_string(in.Rows));
    }
    if (!NoAllocScratch || !StatusWord) return Status::Hip("the static scratch buffer of this device is not available");
    if (!outBits && in.Rows > 0) return Status::Invalid("null output bitmap");
    for (const Column& c : in.Cols) {
        if (c.Dict) return Status::Unsupported("a dictionary-encoded column (decoding it would allocate)");
    }
    StepPlan plan;
    SSA_RETURN_NOT_OK(CompileStep(step, in.Schema(), CM_FILTER_ONLY, plan));
    if (in.Rows == 0) return Status::OK();
    ScanArgs args;
    LaunchCfg cfg;
    int64_t algoBytes;
    SSA_RETURN_NOT_OK(FillScanArgs(plan, in, 0, SCAN_FILTER, args, cfg, algoBytes));
    SSA_RETURN_NOT_OK(GenericKernelCan(args));
    unsigned char* base = static_cast<unsigned char*>(NoAllocScratch);
    uint32_t* tileCounts = reinterpret_cast<uint32_t*>(base);
    unsigned long long* total = reinterpret_cast<unsigned long long*>(base + kNoAllocTotalOffset);
    const int64_t ntiles = (in.Rows + kTileRows - 1) / kTileRows;
    FilterSink sink{outBits, tileCounts};
    ProfileBegin();
    SSA_HIP(LaunchScanFilter(args, sink, cfg));
    ProfileEnd("ScanFilterKernel", algoBytes);
    SSA_HIP(LaunchSumTileCounts(tileCounts, ntiles, total, StreamHandle));
    unsigned long long sel = 0;
    SSA_HIP(hipMemcpyAsync(&sel, total, 8, hipMemcpyDeviceToHost, StreamHandle));
    SSA_RETURN_NOT_OK(CheckScanStatus());
    selected = int64_t(sel);
    return Status::OK();
}

Status Executor::BuildFilter(const TProgramStep& step, const BatchPtr& inEncoded, std::vector<uint8_t>& bits, int64_t& selected) {
    (void)hipSetDevice(Device);
    BatchPtr in;
    SSA_RETURN_NOT_OK(DecodedBatch(inEncoded, in));
    StepPlan plan;
    SSA_RETURN_NOT_OK(CompileStep(step, in->Schema(), CM_FILTER_ONLY, plan));
    BufferPtr bitmap, tileOffsets;
    SSA_RETURN_NOT_OK(RunFilterBitmap(plan, *in, bitmap, tileOffsets, selected));
    bits.assign(size_t((in->Rows + 7) / 8), 0);
    if (!bits.empty()) SSA_HIP(hipMemcpy(bits.data(), bitmap->Ptr, bits.size(), hipMemcpyDeviceToHost));
    return Status::OK();
}

}  // namespace ssa
