// The one entry point that allocates no device memory (ssa_filter_from_program_noalloc): its scratch buffer, the
// per-device address of that buffer, and the small kernel that sums the per-tile counts.
//
// The buffer is a file-scope device array: every GPU the library runs on holds its own copy at its own address until
// the process ends.  Host code never uses the array's name as an address -- NoAllocScratchFor() resolves the device
// address once per device with hipGetSymbolAddress and everything else goes through that pointer.
#include <hip/hip_runtime.h>

#include <mutex>

#include "kernels.h"

namespace ssa {

namespace {

constexpr int kMaxDevices = 64;

// per-tile counts of the filter scan, then the total
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

std::mutex gLock;
void* gScratchPtr[kMaxDevices] = {nullptr};

__global__ __launch_bounds__(kBlock) void SumTileCountsKernel(const uint32_t* counts, int64_t ntiles, unsigned long long* total) {
    __shared__ unsigned long long waveSum[kBlock / 64];
    unsigned long long sum = 0;
    for (int64_t i = threadIdx.x; i < ntiles; i += kBlock) sum += counts[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((threadIdx.x & 63) == 0) waveSum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) *total = waveSum[0] + waveSum[1] + waveSum[2] + waveSum[3];
}

}  // namespace

// Device address of the scratch buffer on `device`, which must be the current device of the calling thread; nullptr
// with *error set when the address cannot be had.
void* NoAllocScratchFor(int device, hipError_t* error) {
    *error = hipSuccess;
    if (device < 0 || device >= kMaxDevices) {
        *error = hipErrorInvalidDevice;
        return nullptr;
    }
    std::lock_guard<std::mutex> g(gLock);
    if (gScratchPtr[device] == nullptr) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr) {
            (void)hipGetLastError();
            *error = e != hipSuccess ? e : hipErrorInvalidSymbol;
            return nullptr;
        }
        gScratchPtr[device] = ptr;
    }
    return gScratchPtr[device];
}

hipError_t LaunchSumTileCounts(const uint32_t* counts, int64_t ntiles, unsigned long long* total, hipStream_t st) {
    hipLaunchKernelGGL(SumTileCountsKernel, dim3(1), dim3(kBlock), 0, st, counts, ntiles, total);
    return hipGetLastError();
}

}  // namespace ssa
