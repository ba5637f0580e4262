// Key-less aggregation: the scan VM feeding per-lane accumulators, reduced to one partial state per workgroup.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "kernels.h"
#include "scan_vm.h"

namespace ssa {

namespace {

constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ int LaneId() { return threadIdx.x & 63; }
__device__ __forceinline__ int WaveId() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint64_t ShflXor64(uint64_t v, int mask) {
    return uint64_t(__shfl_xor(static_cast<unsigned long long>(v), mask, 64));
}

__device__ __forceinline__ uint64_t AggIdentity(int kind) {
    switch (kind) {
        case AGG_MIN_I: return uint64_t(INT64_MAX);
        case AGG_MAX_I: return uint64_t(INT64_MIN);
        case AGG_MIN_U: return ~0ull;
        case AGG_MIN_F: return 0x7FF0000000000000ull;  // +inf
        case AGG_MAX_F: return 0xFFF0000000000000ull;  // -inf
        default: return 0;
    }
}

// Combine two partial states of one aggregate.  Floating min/max follow fmin/fmax: a NaN operand is
// ignored (contrib/libs/apache/arrow/cpp/src/arrow/compute/kernels/aggregate_basic_internal.h:207-228).
__device__ __forceinline__ uint64_t AggCombine(int kind, uint64_t a, uint64_t b) {
    switch (kind) {
        case AGG_COUNT: case AGG_SUM_I: return a + b;
        case AGG_SUM_F: return F64ToBits(BitsToF64(a) + BitsToF64(b));
        case AGG_MIN_I: return int64_t(b) < int64_t(a) ? b : a;
        case AGG_MAX_I: return int64_t(b) > int64_t(a) ? b : a;
        case AGG_MIN_U: return b < a ? b : a;
        case AGG_MAX_U: return b > a ? b : a;
        case AGG_MIN_F: return F64ToBits(fmin(BitsToF64(a), BitsToF64(b)));
        case AGG_MAX_F: return F64ToBits(fmax(BitsToF64(a), BitsToF64(b)));
        default: return a;
    }
}

// ---- sink: key-less aggregation ---------------------------------------------------------------------
// Per-lane accumulators live in the sink area of LDS ([aggregate][lane], 8 bytes each) so that the
// aggregate loop is an ordinary runtime loop: the U rows of a lane are folded in registers first and
// merged into the lane's slot once per tile.
// `a` and `rs` are the kernel arguments of this launch; `s` describes the aggregates and is the constant image in a
// kernel built for one step program.
template <int U>
__device__ __forceinline__ void ScanNoKeyBody() {
    const ScanArgs& a = KernArgs<NoKeySink>().a;
    const NoKeySink& rs = KernArgs<NoKeySink>().s;
#ifdef SSA_SPEC
    const NoKeySink& s = reinterpret_cast<const KArgs<NoKeySink>*>(kSpecImage)->s;
#else
    const NoKeySink& s = rs;
#endif
    Machine<U> m(a);
    uint64_t* acc = reinterpret_cast<uint64_t*>(gLds + m.S().lds_sink_off);  // [naggs][kBlock]
    uint32_t seen = 0;
    SSA_UNROLL_STATIC
    for (int k = 0; k < s.naggs; ++k) acc[k * kBlock + threadIdx.x] = AggIdentity(s.aggs[k].kind);

    const int64_t ntiles = (m.nrows + kBlock * U - 1) / (kBlock * U);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        m.BeginTile(tile * (kBlock * U), (tile + gridDim.x) * (kBlock * U));
        m.Run();
        if (m.P == 0) continue;
        SSA_UNROLL_STATIC
        for (int k = 0; k < s.naggs; ++k) {
            const AggSpec spec = s.aggs[k];
            uint32_t live = m.P;
            Val<U> x;
            if (spec.reg != -1) {
                m.FetchOperand(spec.fetch_kind, spec.fetch_payload, x);
                if (spec.check_valid) live &= x.ok;
            }
            if (live == 0) continue;
            seen |= 1u << k;
            uint64_t v;
            if (spec.kind == AGG_COUNT) {
                v = __popc(live);
            } else {
                v = AggIdentity(spec.kind);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if ((live >> u) & 1u) v = AggCombine(spec.kind, v, x.v[u]);
                }
            }
            uint64_t* slot = acc + k * kBlock + threadIdx.x;
            *slot = AggCombine(spec.kind, *slot, v);
        }
    }
    m.ReportErrors();

    // workgroup reduction: wave butterflies, then the 4 wave results through LDS
    __shared__ uint64_t lds[kWaves][kMaxAggs];
    __shared__ uint32_t ldsSeen[kWaves];
    SSA_UNROLL_STATIC
    for (int k = 0; k < s.naggs; ++k) {
        const int kind = s.aggs[k].kind;
        uint64_t v = acc[k * kBlock + threadIdx.x];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = AggCombine(kind, v, ShflXor64(v, off));
        if (LaneId() == 0) lds[WaveId()][k] = v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) seen |= __shfl_xor(seen, off, 64);
    if (LaneId() == 0) ldsSeen[WaveId()] = seen;
    __syncthreads();
    if (threadIdx.x < kMaxAggs && int(threadIdx.x) < s.naggs) {
        const int k = threadIdx.x;
        const int kind = s.aggs[k].kind;
        uint64_t v = lds[0][k];
        for (int w = 1; w < kWaves; ++w) v = AggCombine(kind, v, lds[w][k]);
        rs.partial[size_t(blockIdx.x) * kMaxAggs + k] = v;
    }
    if (threadIdx.x == 0) {
        uint32_t sn = 0;
        for (int w = 0; w < kWaves; ++w) sn |= ldsSeen[w];
        rs.partial_seen[blockIdx.x] = sn;
    }
}

}  // namespace

#ifdef SSA_SPEC
}  // namespace ssa
extern "C" __global__ __launch_bounds__(ssa::kBlock) void ssa_spec_nokey(const ssa::KArgs<ssa::NoKeySink>) {
    ssa::ScanNoKeyBody<SSA_SPEC_U>();
}
namespace ssa {
#else
namespace {
template <int U>
__global__ __launch_bounds__(kBlock) void ScanNoKeyKernel(const KArgs<NoKeySink>) {
    ScanNoKeyBody<U>();
}
}  // namespace
#endif

}  // namespace ssa
