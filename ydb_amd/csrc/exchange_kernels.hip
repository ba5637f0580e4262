// Moving result batches between GPUs: the copy kernel behind pack (a batch into one flat buffer for a collective) and
// concat (the batches of all ranks into the input of the merge program).
//
// The reference merges the partial aggregates of its shards on the compute node that reads them
// (ydb/core/tx/columnshard/columnshard__scan.cpp:300-340 ships every batch's result there); here the partial tables
// stay in HBM, travel over xGMI as flat buffers and are concatenated by one launch of this kernel.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ssa {

namespace {

__global__ __launch_bounds__(kBlock) void CopySegmentsKernel(const CopySegment* segs) {
    const CopySegment s = segs[blockIdx.y];
    const uint64_t tid = uint64_t(blockIdx.x) * kBlock + threadIdx.x, nthreads = uint64_t(gridDim.x) * kBlock;
    switch (s.kind) {
        case SEG_BYTES: {
            if (((reinterpret_cast<uintptr_t>(s.src) | reinterpret_cast<uintptr_t>(s.dst)) & 15) == 0) {
                const uint64_t n16 = s.count / 16;
                const uint4* src = reinterpret_cast<const uint4*>(s.src);
                uint4* dst = reinterpret_cast<uint4*>(s.dst);
                for (uint64_t i = tid; i < n16; i += nthreads) dst[i] = src[i];
                for (uint64_t i = n16 * 16 + tid; i < s.count; i += nthreads) s.dst[i] = s.src[i];
            } else {
                for (uint64_t i = tid; i < s.count; i += nthreads) s.dst[i] = s.src[i];
            }
            break;
        }
        case SEG_I32_ADD: {  // string offsets moved behind the bytes of the parts before
            const int32_t* src = reinterpret_cast<const int32_t*>(s.src);
            int32_t* dst = reinterpret_cast<int32_t*>(s.dst);
            for (uint64_t i = tid; i < s.count; i += nthreads) dst[i] = src[i] + int32_t(s.arg);
            break;
        }
        case SEG_BITS_TO_BYTES: {  // validity bits (from bit `arg` of src) as one byte per row
            for (uint64_t i = tid; i < s.count; i += nthreads) {
                const uint64_t b = i + uint64_t(s.arg);
                s.dst[i] = (s.src[b >> 3] >> (b & 7)) & 1;
            }
            break;
        }
        case SEG_FILL: {
            for (uint64_t i = tid; i < s.count; i += nthreads) s.dst[i] = uint8_t(s.arg);
            break;
        }
        default: break;
    }
}

}  // namespace

hipError_t LaunchCopySegments(const CopySegment* deviceSegments, int nsegments, uint64_t maxCount, hipStream_t st) {
    if (nsegments <= 0) return hipSuccess;
    uint64_t blocks = (maxCount / 16 + kBlock - 1) / kBlock;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    for (int first = 0; first < nsegments; first += 65535) {
        const int n = nsegments - first < 65535 ? nsegments - first : 65535;
        hipLaunchKernelGGL(CopySegmentsKernel, dim3(unsigned(blocks), unsigned(n)), dim3(kBlock), 0, st, deviceSegments + first);
    }
    return hipGetLastError();
}

}  // namespace ssa
