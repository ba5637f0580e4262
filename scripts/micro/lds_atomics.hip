// Micro-benchmark: what LDS operations with random addresses cost a CU (cycles per wave-instruction), by kind of
// operation and by the number of waves that share the CU.  hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kSlots = 2048;  // 16 KB of 64-bit words
constexpr int kIters = 4096;

__device__ __forceinline__ uint32_t Next(uint32_t x) { return x * 1664525u + 1013904223u; }

template <int KIND>
__global__ void Bench(uint64_t* out, uint32_t* cycles) {
    __shared__ uint64_t t[kSlots];
    for (int i = threadIdx.x; i < kSlots; i += blockDim.x) t[i] = KIND == 1 ? ~0ull : 0ull;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
    uint64_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        x = Next(x);
        const uint32_t s = (x >> 10) & (kSlots - 1);
        if (KIND == 0) acc += t[s];                                                                   // ds_read_b64
        if (KIND == 1) acc += atomicCAS((unsigned long long*)&t[s], ~0ull, (unsigned long long)x);    // ds_cmpst_rtn_b64
        if (KIND == 2) atomicAdd((unsigned long long*)&t[s], 1ull);                                   // ds_add_u64
        if (KIND == 3) acc += atomicAdd((unsigned long long*)&t[s], 1ull);                            // ds_add_rtn_u64
        if (KIND == 4) atomicAdd((unsigned int*)&t[s], 1u);                                           // ds_add_u32
        if (KIND == 5) acc += atomicAdd((unsigned int*)&t[s], 1u);                                    // ds_add_rtn_u32
        if (KIND == 6) acc += atomicCAS((unsigned int*)&t[s], 0xFFFFFFFFu, x);                        // ds_cmpst_rtn_b32
        if (KIND == 7) t[s] = x;                                                                      // ds_write_b64
        if (KIND == 8) acc += ((uint32_t*)t)[s];                                                      // ds_read_b32
        if (KIND == 9) atomicAdd((unsigned long long*)&t[s & ~63u], 1ull);                            // ds_add_u64, all lanes of a wave one address... nearly: 32 addresses per table
        if (KIND == 10) atomicAdd((unsigned long long*)&t[(threadIdx.x & 1) ? (s & ~63u) : s], 1ull); // half the lanes on few addresses
        if (KIND == 11) atomicAdd((unsigned long long*)&t[it & (kSlots - 1)], 1ull);                  // every lane the same address
    }
    __syncthreads();
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = uint32_t(t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + t[threadIdx.x];
}

int main() {
    const char* names[] = {"ds_read_b64", "ds_cmpst_rtn_b64", "ds_add_u64", "ds_add_rtn_u64", "ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b32", "ds_write_b64", "ds_read_b32", "ds_add_u64 32 addr", "ds_add_u64 half hot", "ds_add_u64 one addr"};
    uint64_t* out;
    uint32_t* cyc;
    hipMalloc(&out, 256 * 8 * 1024 * 8);
    hipMalloc(&cyc, 256 * 8 * 4);
    for (int threads : {64, 256, 512, 1024}) {
        for (int kind = 0; kind < 12; ++kind) {
            const int grid = 256;  // one workgroup per CU
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                switch (kind) {
                    case 0: hipLaunchKernelGGL(Bench<0>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 1: hipLaunchKernelGGL(Bench<1>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 2: hipLaunchKernelGGL(Bench<2>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 3: hipLaunchKernelGGL(Bench<3>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 4: hipLaunchKernelGGL(Bench<4>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 5: hipLaunchKernelGGL(Bench<5>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 6: hipLaunchKernelGGL(Bench<6>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 7: hipLaunchKernelGGL(Bench<7>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 8: hipLaunchKernelGGL(Bench<8>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 9: hipLaunchKernelGGL(Bench<9>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 10: hipLaunchKernelGGL(Bench<10>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                    case 11: hipLaunchKernelGGL(Bench<11>, dim3(grid), dim3(threads), 0, 0, out, cyc); break;
                }
            }
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint32_t> h(grid);
            hipMemcpy(h.data(), cyc, grid * 4, hipMemcpyDeviceToHost);
            double avg = 0;
            for (uint32_t c : h) avg += c;
            avg /= grid;
            const int waves = threads / 64;
            // readcyclecounter ticks at 100 MHz on this part: report both raw ticks and per-instruction figures
            std::printf("%-18s waves/CU %2d: %9.0f ticks, kernel %.1f us -> %.2f ns per wave-instruction per CU (%.1f ns per wave: latency bound when alone)\n", names[kind], waves,
                        avg, ms * 1e3, ms * 1e6 / (double(kIters) * waves), ms * 1e6 / kIters);
        }
    }
    return 0;
}
