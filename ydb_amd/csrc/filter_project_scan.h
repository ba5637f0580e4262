// Filter and projection sinks of the scan VM: the filter bitmap with per-tile counts, and computed columns written at
// the compacted positions of the selected rows.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "kernels.h"
#include "nokey_scan.h"

namespace ssa {

namespace {

// ---- sink: filter bitmap ----------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void ScanFilterBody() {
    const ScanArgs& a = KernArgs<FilterSink>().a;
    const FilterSink& s = KernArgs<FilterSink>().s;  // addresses only: nothing of it is constant
    Machine<U> m(a);
    __shared__ uint32_t ldsCount[kWaves];
    const int64_t ntiles = (m.nrows + kBlock * U - 1) / (kBlock * U);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tileBase = tile * (kBlock * U);
        m.BeginTile(tileBase, (tile + gridDim.x) * (kBlock * U));
        m.Run();
        uint32_t count = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t word = __ballot((m.P >> u) & 1u);
            const int64_t firstRow = tileBase + int64_t(WaveId()) * (64 * U) + u * 64;
            if (LaneId() == 0 && firstRow < a.nrows) s.bitmap[firstRow >> 6] = word;
            count += __popcll(word);
        }
        if (LaneId() == 0) ldsCount[WaveId()] = count;
        __syncthreads();
        if (threadIdx.x == 0) s.tile_counts[tile] = ldsCount[0] + ldsCount[1] + ldsCount[2] + ldsCount[3];
        __syncthreads();
    }
    m.ReportErrors();
}

// Position of the first row of this wave's 64-row word among the selected rows of its tile.
// wordInTile = index (0..15) of the word inside the tile of kTileRows rows.
__device__ __forceinline__ uint32_t SelectedBeforeWord(const uint64_t* bitmap, int64_t tileFirstWord, int wordInTile,
                                                       int64_t nwords) {
    const int lane = LaneId();
    uint32_t c = 0;
    if (lane < wordInTile && tileFirstWord + lane < nwords) c = __popcll(bitmap[tileFirstWord + lane]);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);  // words live in lanes 0..15
    return __shfl(c, 0, 64);
}

// ---- sink: projection (computed columns written at compacted positions) -------------------------------
// `rs` carries the addresses of this launch; `s` describes the outputs and is the constant image in a kernel built for
// one step program.
template <int U>
__device__ __forceinline__ void ScanProjectBody() {
    const ScanArgs& a = KernArgs<ProjectSink>().a;
    const ProjectSink& rs = KernArgs<ProjectSink>().s;
#ifdef SSA_SPEC
    const ProjectSink& s = reinterpret_cast<const KArgs<ProjectSink>*>(kSpecImage)->s;
#else
    const ProjectSink& s = rs;
#endif
    Machine<U> m(a);
    const int64_t ntiles = (m.nrows + kBlock * U - 1) / (kBlock * U);
    const int64_t nwords = (a.nrows + 63) >> 6;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t tileBase = tile * (kBlock * U);
        int64_t pos[U];
        uint32_t sel = 0;
        if (rs.bitmap != nullptr) {
            const int64_t tileOff = rs.tile_offsets[tile];
            if (rs.tile_offsets[tile + 1] == tileOff) continue;  // nothing selected in this tile
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int wordInTile = WaveId() * U + u;
                const int64_t wordIdx = (tileBase >> 6) + wordInTile;
                const uint64_t word = wordIdx < nwords ? rs.bitmap[wordIdx] : 0;
                const uint32_t before = SelectedBeforeWord(rs.bitmap, tileBase >> 6, wordInTile, nwords);
                const int lane = LaneId();
                sel |= uint32_t((word >> lane) & 1ull) << u;
                pos[u] = tileOff + before + __popcll(word & ((1ull << lane) - 1ull));
            }
        }
        m.BeginTile(tileBase, (tile + gridDim.x) * (kBlock * U));
        if (rs.bitmap == nullptr) {
            sel = m.act;
#pragma unroll
            for (int u = 0; u < U; ++u) pos[u] = m.Row(u);
        }
        m.Run();
        SSA_UNROLL_STATIC
        for (int k = 0; k < s.nouts; ++k) {
            ProjectOut o = s.outs[k];
            o.values = rs.outs[k].values;
            o.valid_bytes = rs.outs[k].valid_bytes;
            Val<U> x;
            m.FetchOperand(o.fetch_kind, o.fetch_payload, x);
            const int w = (o.type == T_BOOL) ? 1 : TypeWidth(o.type);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if ((sel >> u) & 1u) {
                    StoreTyped(o.values, w, pos[u], PayloadToStored(x.v[u], o.type));
                    if (o.valid_bytes != nullptr) o.valid_bytes[pos[u]] = uint8_t((x.ok >> u) & 1u);
                }
            }
        }
    }
    m.ReportErrors();
}


}  // namespace

#ifdef SSA_SPEC
}  // namespace ssa
extern "C" __global__ __launch_bounds__(ssa::kBlock) void ssa_spec_filter(const ssa::KArgs<ssa::FilterSink>) {
    ssa::ScanFilterBody<SSA_SPEC_U>();
}
extern "C" __global__ __launch_bounds__(ssa::kBlock) void ssa_spec_project(const ssa::KArgs<ssa::ProjectSink>) {
    ssa::ScanProjectBody<SSA_SPEC_U>();
}
namespace ssa {
#else
namespace {
template <int U>
__global__ __launch_bounds__(kBlock) void ScanFilterKernel(const KArgs<FilterSink>) {
    ScanFilterBody<U>();
}
template <int U>
__global__ __launch_bounds__(kBlock) void ScanProjectKernel(const KArgs<ProjectSink>) {
    ScanProjectBody<U>();
}
}  // namespace
#endif

}  // namespace ssa
