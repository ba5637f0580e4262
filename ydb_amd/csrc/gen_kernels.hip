// Synthetic TPC-H lineitem columns generated directly in HBM.
//
// Every value is a pure function of (seed, column id, row number) through one splitmix64 step, so any
// row range can be produced independently on any GPU and re-derived on the host for parity checks.
// Value distributions follow dbgen's rules for the columns TPC-H Q1 / Q6 read
// (ydb/library/workload/tpch/tpch_schema.sql:19-37 types: Date32 -> int32 days, Double money columns,
// Utf8 flags).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ssa {

namespace {

__device__ __forceinline__ uint64_t SplitMix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Upper 32 bits of the mixed word: every range reduction below is a 32-bit modulo.
__device__ __forceinline__ uint32_t Draw(uint64_t seed, uint64_t col, uint64_t row) {
    return uint32_t(SplitMix64(seed ^ (col << 56) ^ row) >> 32);
}

__global__ __launch_bounds__(kBlock) void GenLineitemKernel(const LineitemColumns c, int64_t row0, int64_t nrows, uint64_t seed) {
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i <= nrows; i += int64_t(gridDim.x) * kBlock) {
        if (i == nrows) {  // closing offsets
            if (c.rf_offsets) c.rf_offsets[i] = int32_t(i);
            if (c.ls_offsets) c.ls_offsets[i] = int32_t(i);
            continue;
        }
        const uint64_t row = uint64_t(row0 + i);
        const int64_t qty = 1 + int64_t(Draw(seed, 1, row) % 50);
        const int64_t cents = 90000 + int64_t(Draw(seed, 4, row) % 20001) + 100 * int64_t(Draw(seed, 5, row) % 1000);
        const int32_t ship = 8036 + int32_t(Draw(seed, 6, row) % 2526);
        const int32_t receipt = ship + 1 + int32_t(Draw(seed, 7, row) % 30);
        const int32_t cutoff = 9298;  // 1995-06-17
        int8_t rf, ls;
        if (receipt <= cutoff) rf = (Draw(seed, 8, row) & 1) ? 0 : 2; else rf = 1;
        ls = ship <= cutoff ? 0 : 1;
        if (c.l_quantity) c.l_quantity[i] = double(qty);
        if (c.l_discount) c.l_discount[i] = double(Draw(seed, 2, row) % 11) / 100.0;
        if (c.l_tax) c.l_tax[i] = double(Draw(seed, 3, row) % 9) / 100.0;
        if (c.l_extendedprice) c.l_extendedprice[i] = double(qty * cents) / 100.0;
        if (c.l_shipdate) c.l_shipdate[i] = ship;
        if (c.rf_offsets) c.rf_offsets[i] = int32_t(i);
        if (c.ls_offsets) c.ls_offsets[i] = int32_t(i);
        if (c.rf_chars) c.rf_chars[i] = rf == 0 ? 'A' : rf == 1 ? 'N' : 'R';
        if (c.ls_chars) c.ls_chars[i] = ls == 0 ? 'F' : 'O';
        if (c.rf_code) c.rf_code[i] = rf;
        if (c.ls_code) c.ls_code[i] = ls;
    }
}

}  // namespace

hipError_t LaunchGenLineitem(const LineitemColumns& c, int64_t row0, int64_t nrows, uint64_t seed, hipStream_t st) {
    int64_t grid = (nrows + kBlock) / kBlock;
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(GenLineitemKernel, dim3(uint32_t(grid)), dim3(kBlock), 0, st, c, row0, nrows, seed);
    return hipGetLastError();
}

}  // namespace ssa
