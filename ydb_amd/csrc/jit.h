// Run-time specialisation of the scan kernels.
//
// The library ships generic scan kernels that interpret any step program.  For a program that is going to scan many
// rows the same hand-written kernel source (scan_vm.h, keyed_scan.h, ...) is compiled once more at run time with the
// step's tables fixed as constants (see the SSA_SPEC notes in scan_vm.h): loops unroll, the interpreter's dispatch
// folds away and what is left is the straight-line code of that one program.  Compilation goes through hiprtc, loaded
// with dlopen so that the library itself does not depend on it.  Compiled kernels are kept in the memory of the
// process, keyed by the full bytes of the constant image and the kernel variant (no two programs can share a kernel by
// a hash collision); the least recently used ones are unloaded beyond 256 kernels.  Nothing is written to or read
// from disk unless SSA_JIT_CACHE_DIR names a directory: then code objects are kept there across processes (the file
// carries the whole key and is compared with it).  The compiler never runs under the cache's lock: a scan that is
// large by itself (kJitMinRows rows) builds on its own thread and waits, scans that only add up to the threshold hand
// the build to a worker thread and keep running the generic kernel until it is ready (SSA_JIT_ASYNC=0: always wait).
// Whenever anything here fails the caller keeps using the generic kernel.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace ssa {

struct SpecKernel {
    hipFunction_t Function = nullptr;
    int BlocksPerCU = 0;  // resident workgroups per CU at the LDS size given to Get()
};

class KernelJit {
public:
    static KernelJit& Instance();

    // Kernel `entry` of the translation unit `mainHeader` (one of the embedded kernel headers) built for the constant
    // image `image` (imageBytes, a multiple of 4) and the given defines (e.g. "#define SSA_SPEC_U 4\n").  `rows` is the
    // size of the scan about to run: a kernel is built once the scans asking for it add up to kJitMinRows rows
    // (SSA_JIT=1: at the first one).  Returns false when there is no such kernel (yet); `why` is empty when that is
    // only the policy, else it says what failed.
    bool Get(int device, const char* mainHeader, const char* entry, const void* image, size_t imageBytes,
             const std::string& defines, uint32_t ldsBytes, int64_t rows, SpecKernel& out, std::string& why);

    // Process-wide counters: kernels built, host milliseconds spent building them, launches that used one.
    void Stats(int64_t& built, double& buildMs, int64_t& launches) const;
    void NoteLaunch();
    // Ends background builds (a build in progress is waited for; later requests run the generic kernels).  To be called
    // before the process exits while a build may be running.
    void Shutdown();

private:
    KernelJit() = default;
};

// Building a kernel takes about half a second of host time, as much as the generic kernels need for tens of billions
// of rows: it pays for step programs that keep coming back (every portion of a table, every run of a dashboard query).
constexpr int64_t kJitMinRows = int64_t(1) << 25;
// SSA_JIT, read at every call: "0" never specialise, "1" at the first scan, otherwise by the row count above.
bool JitEnabled();

}  // namespace ssa
