// Dictionary-encoded columns: the small kernels around the codes.
//
// The reference decodes a dictionary column before any program step sees it
// (ydb/core/formats/arrow/dictionary/conversion.h:13-15, DictionaryToArray); the results here are those of the decoded
// column, computed through the codes: a function of the column is evaluated once per dictionary entry and read through
// OP_LOOKUP, a GROUP BY groups by code.  These kernels turn a per-entry result column into the lookup table's
// payload form, and codes into row indices of the dictionary for the gathers that decode keys or whole columns.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "scan_vm.h"

namespace ssa {

namespace {

__device__ __forceinline__ int64_t LoadCode(const void* idx, int type, int64_t i) {
    switch (type) {
        case T_I8: return static_cast<const int8_t*>(idx)[i];
        case T_U8: return static_cast<const uint8_t*>(idx)[i];
        case T_I16: return static_cast<const int16_t*>(idx)[i];
        case T_U16: return static_cast<const uint16_t*>(idx)[i];
        case T_I32: return static_cast<const int32_t*>(idx)[i];
        case T_U32: return static_cast<const uint32_t*>(idx)[i];
        default: return static_cast<const int64_t*>(idx)[i];
    }
}

__global__ __launch_bounds__(kBlock) void WidenCodesKernel(const void* idx, int type, int64_t n, uint64_t entries, const uint8_t* validity,
                                                         int64_t validBitOff, uint64_t* rows) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        const bool ok = validity == nullptr || LoadBit(validity, i + validBitOff) != 0;
        const uint64_t code = uint64_t(LoadCode(idx, type, i));
        rows[i] = (ok && code < entries) ? code : ~0ull;
    }
}

__global__ __launch_bounds__(kBlock) void ColumnToPayloadKernel(const void* values, int type, int64_t valBitOff, const uint8_t* validity,
                                                              int64_t validBitOff, int64_t n, uint64_t* payload, uint8_t* valid) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kBlock) {
        uint64_t v;
        if (type == T_BOOL) {
            v = LoadBit(static_cast<const uint8_t*>(values), i + valBitOff);
        } else {
            v = ConvertRaw(LoadRaw(values, TypeWidth(type), i), type);
        }
        payload[i] = v;
        valid[i] = (validity == nullptr || LoadBit(validity, i + validBitOff) != 0) ? 1 : 0;
    }
}

int GridOf(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return int(g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g));
}

}  // namespace

hipError_t LaunchWidenCodes(const void* idx, int type, int64_t n, uint64_t entries, const uint8_t* validity, int64_t validBitOff, uint64_t* rows,
                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(WidenCodesKernel, dim3(GridOf(n)), dim3(kBlock), 0, st, idx, type, n, entries, validity, validBitOff, rows);
    return hipGetLastError();
}

hipError_t LaunchColumnToPayload(const void* values, int type, int64_t valBitOff, const uint8_t* validity, int64_t validBitOff, int64_t n,
                                 uint64_t* payload, uint8_t* valid, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ColumnToPayloadKernel, dim3(GridOf(n)), dim3(kBlock), 0, st, values, type, valBitOff, validity, validBitOff, n, payload, valid);
    return hipGetLastError();
}

}  // namespace ssa
