#include "jit.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

namespace ssa {

// The kernel headers, embedded by the build (see Makefile: embedded_sources.cpp).
extern const int kEmbeddedCount;
extern const char* const kEmbeddedNames[];
extern const char* const kEmbeddedSources[];

namespace {

struct Rtc {
    void* Lib = nullptr;
    decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
    decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
    decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
    decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
    decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
    decltype(&hiprtcGetCode) GetCode = nullptr;
    decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
    bool Tried = false;

    bool Load(std::string& why) {
        if (Tried) {
            if (!Lib) why = "hiprtc is not available";
            return Lib != nullptr;
        }
        Tried = true;
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            Lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (Lib) break;
        }
        if (!Lib) {
            why = std::string("cannot load hiprtc: ") + dlerror();
            return false;
        }
#define SSA_RTC_SYM(field, sym)                                               \
    field = reinterpret_cast<decltype(field)>(dlsym(Lib, #sym));              \
    if (!field) {                                                             \
        why = "hiprtc lacks " #sym;                                           \
        Lib = nullptr;                                                        \
        return false;                                                         \
    }
        SSA_RTC_SYM(CreateProgram, hiprtcCreateProgram)
        SSA_RTC_SYM(CompileProgram, hiprtcCompileProgram)
        SSA_RTC_SYM(GetProgramLogSize, hiprtcGetProgramLogSize)
        SSA_RTC_SYM(GetProgramLog, hiprtcGetProgramLog)
        SSA_RTC_SYM(GetCodeSize, hiprtcGetCodeSize)
        SSA_RTC_SYM(GetCode, hiprtcGetCode)
        SSA_RTC_SYM(DestroyProgram, hiprtcDestroyProgram)
#undef SSA_RTC_SYM
        return true;
    }
};

uint64_t Fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

struct State {
    std::mutex Lock;
    Rtc R;
    std::map<std::pair<int, uint64_t>, hipFunction_t> Functions;  // (device, key) -> kernel; modules are never unloaded
    std::map<uint64_t, std::vector<char>> Code;                   // key -> code object (shared by the devices)
    std::map<uint64_t, std::string> Failed;                       // key -> reason (not retried)
    std::map<uint64_t, int64_t> SeenRows;                         // key -> rows of the scans that asked so far
    std::atomic<int64_t> Compiled{0}, Launches{0}, BuildMicros{0};
    std::map<int, std::string> Arch;  // device -> target name
};

State& St() {
    static State s;
    return s;
}

}  // namespace

KernelJit& KernelJit::Instance() {
    static KernelJit j;
    return j;
}

void KernelJit::Stats(int64_t& built, double& buildMs, int64_t& launches) const {
    built = St().Compiled.load();
    buildMs = double(St().BuildMicros.load()) / 1e3;
    launches = St().Launches.load();
}

void KernelJit::NoteLaunch() { ++St().Launches; }

bool JitEnabled() {
    const char* mode = std::getenv("SSA_JIT");
    return !(mode && mode[0] == '0');
}

bool KernelJit::Get(int device, const char* mainHeader, const char* entry, const void* image, size_t imageBytes,
                    const std::string& defines, uint32_t ldsBytes, int64_t rows, SpecKernel& out, std::string& why) {
    State& st = St();
    std::lock_guard<std::mutex> g(st.Lock);
    std::string& arch = st.Arch[device];
    if (arch.empty()) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
            why = "no device properties";
            return false;
        }
        arch = prop.gcnArchName;
    }
    uint64_t key = Fnv(mainHeader, std::strlen(mainHeader));
    key = Fnv(entry, std::strlen(entry), key);
    key = Fnv(defines.data(), defines.size(), key);
    key = Fnv(image, imageBytes, key);
    key = Fnv(arch.data(), arch.size(), key);

    hipFunction_t fn = nullptr;
    auto hit = st.Functions.find({device, key});
    if (hit != st.Functions.end()) {
        fn = hit->second;
    } else {
        auto bad = st.Failed.find(key);
        if (bad != st.Failed.end()) {
            why = bad->second;
            return false;
        }
        const char* mode = std::getenv("SSA_JIT");
        int64_t& seen = st.SeenRows[key];
        seen += rows;
        if (!(mode && mode[0] == '1') && seen < kJitMinRows) {
            why.clear();
            return false;
        }
        auto have = st.Code.find(key);
        if (have == st.Code.end()) {
            if (!st.R.Load(why)) {
                st.Failed[key] = why;
                return false;
            }
            // translation unit: the constant image, the variant's defines, then the kernel header
            std::ostringstream src;
            src << "#define SSA_SPEC 1\n" << defines;
            src << "alignas(16) static constexpr unsigned int kSpecImage[] = {";
            const uint32_t* words = static_cast<const uint32_t*>(image);
            for (size_t i = 0; i < imageBytes / 4; ++i) {
                if (i % 16 == 0) src << "\n";
                src << words[i] << "u,";
            }
            src << "\n};\n#include \"" << mainHeader << "\"\n";
            const std::string text = src.str();
            if (const char* dump = std::getenv("SSA_JIT_DUMP")) {  // diagnostics: the translation unit, for offline inspection
                if (FILE* f = std::fopen(dump, "w")) {
                    std::fwrite(text.data(), 1, text.size(), f);
                    std::fclose(f);
                }
            }
            hiprtcProgram prog = nullptr;
            if (st.R.CreateProgram(&prog, text.c_str(), "ssa_spec.hip", kEmbeddedCount, const_cast<const char**>(kEmbeddedSources),
                                   const_cast<const char**>(kEmbeddedNames)) != HIPRTC_SUCCESS) {
                why = "hiprtcCreateProgram failed";
                st.Failed[key] = why;
                return false;
            }
            const auto t0 = std::chrono::steady_clock::now();
            const std::string archOpt = "--offload-arch=" + arch;
            // The step program's loops are unrolled by pragma with the whole interpreter inlined in their bodies: the
            // default size limit of pragma-forced unrolling (16 K instructions before folding) is too small for that
            // and, once exceeded, silently leaves the generic interpreter in the built kernel.
            // No contraction of a multiply and a following add into one fused operation: in the straight-line code of a
            // built kernel two instructions of the step program sit next to each other, and the reference rounds after
            // each of them (a * b + c differed in the last bit on a quarter of the rows).  Where a fused operation is
            // wanted (the register-held groups) the source asks for it by name.
            const char* opts[] = {archOpt.c_str(), "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-inline-asm", "-ffp-contract=off",
                                  "-mllvm", "-pragma-unroll-threshold=1048576"};
            const hiprtcResult rc = st.R.CompileProgram(prog, int(sizeof(opts) / sizeof(opts[0])), opts);
            if (rc != HIPRTC_SUCCESS) {
                size_t n = 0;
                (void)st.R.GetProgramLogSize(prog, &n);
                std::string log(n, '\0');
                if (n) (void)st.R.GetProgramLog(prog, &log[0]);
                (void)st.R.DestroyProgram(&prog);
                why = "run-time compilation failed: " + log.substr(0, 2000);
                st.Failed[key] = why;
                return false;
            }
            size_t n = 0;
            (void)st.R.GetCodeSize(prog, &n);
            std::vector<char> code(n);
            if (n) (void)st.R.GetCode(prog, code.data());
            (void)st.R.DestroyProgram(&prog);
            ++st.Compiled;
            st.BuildMicros += int64_t(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            have = st.Code.emplace(key, std::move(code)).first;
        }
        (void)hipSetDevice(device);
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, have->second.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, entry);
        if (e != hipSuccess) {
            why = std::string("cannot load the compiled kernel: ") + hipGetErrorString(e);
            st.Failed[key] = why;
            return false;
        }
        st.Functions[{device, key}] = fn;
    }
    out.Function = fn;
    int n = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, ldsBytes) != hipSuccess || n <= 0) n = 1;
    out.BlocksPerCU = n;
    return true;
}

}  // namespace ssa
