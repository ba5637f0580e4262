#include "jit.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <unordered_map>
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

// Options of every run-time compilation (the target is added per device).
const char* const kCompileOptions[] = {"-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-inline-asm", "-ffp-contract=off",
                                       "-mllvm", "-pragma-unroll-threshold=1048576"};

uint64_t Fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

using Clock = std::chrono::steady_clock;

// One (kernel variant, constant image, target) and what is known about it.  The map is keyed by the entry's full
// bytes -- header, entry point, defines, target and image -- so that two programs can never share a kernel by accident.
struct Entry {
    enum EState { NONE, BUILDING, READY, FAILED } State = NONE;
    int64_t SeenRows = 0;                      // rows of the scans that asked so far
    std::string Why;                           // FAILED: the reason (not retried)
    std::shared_ptr<std::vector<char>> Code;   // READY: the code object, shared by the devices
    struct Loaded {
        hipModule_t Module = nullptr;
        hipFunction_t Function = nullptr;
    };
    std::map<int, Loaded> PerDevice;
    Clock::time_point LastUse;
};

struct Job {
    std::string Key, MainHeader, Defines, Arch;
    std::vector<uint32_t> Image;
};

struct State {
    std::mutex Lock;              // the maps below; never held while the compiler runs
    std::mutex RtcLock;           // loading hiprtc
    Rtc R;
    std::unordered_map<std::string, Entry> Entries;
    std::map<int, std::string> Arch;  // device -> target name
    std::atomic<int64_t> Compiled{0}, Launches{0}, BuildMicros{0}, Evicted{0}, FromDisk{0};
    // background builds
    std::mutex QueueLock;
    std::condition_variable QueueCv;
    std::deque<Job> Queue;
    std::thread Worker;
    bool WorkerStarted = false, Stop = false;

    // No more background builds: the queue is dropped, a build in progress is waited for.
    void StopWorker() {
        {
            std::lock_guard<std::mutex> g(QueueLock);
            Stop = true;
            Queue.clear();
        }
        QueueCv.notify_all();
        std::lock_guard<std::mutex> j(JoinLock);
        if (Worker.joinable()) Worker.join();
    }
    std::mutex JoinLock;

    ~State() { StopWorker(); }
};

State& St() {
    static State s;
    return s;
}

// What a built kernel depends on besides its own key: the kernel sources embedded in this library (their argument
// structs are the launch ABI), the compiler's options and the runtime's version.  Part of every cache key, so that a code
// object a differently built library left in SSA_JIT_CACHE_DIR is never loaded.  (SSA_JIT_KEY_SALT: tests.)
const std::string& BuildFingerprint() {
    static const std::string fp = [] {
        uint64_t a = 1469598103934665603ull, b = 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < kEmbeddedCount; ++i) {
            a = Fnv(kEmbeddedNames[i], std::strlen(kEmbeddedNames[i]) + 1, a);
            a = Fnv(kEmbeddedSources[i], std::strlen(kEmbeddedSources[i]) + 1, a);
            b = Fnv(kEmbeddedSources[i], std::strlen(kEmbeddedSources[i]) + 1, b);
        }
        std::string s = "sources:";
        char hex[40];
        std::snprintf(hex, sizeof(hex), "%016llx%016llx", static_cast<unsigned long long>(a), static_cast<unsigned long long>(b));
        s += hex;
        s += " options:";
        for (const char* o : kCompileOptions) {
            s += o;
            s += ' ';
        }
        int runtime = 0;
        if (hipRuntimeGetVersion(&runtime) != hipSuccess) (void)hipGetLastError();
        s += "runtime:" + std::to_string(runtime);
        if (const char* salt = std::getenv("SSA_JIT_KEY_SALT")) s += std::string(" salt:") + salt;
        return s;
    }();
    return fp;
}

constexpr size_t kMaxBuiltKernels = 256;        // READY entries kept (each holds a code object and a module per device)
constexpr size_t kMaxEntries = 8192;            // entries of any state
constexpr auto kEvictIdle = std::chrono::seconds(30);  // a kernel used this recently may still have launches in flight

std::string CacheFile(const std::string& key) {
    const char* dir = std::getenv("SSA_JIT_CACHE_DIR");
    if (!dir || !dir[0]) return std::string();
    char name[64];
    std::snprintf(name, sizeof(name), "/ssa_%016llx%016llx.co", static_cast<unsigned long long>(Fnv(key.data(), key.size())),
                  static_cast<unsigned long long>(Fnv(key.data(), key.size(), 0x9E3779B97F4A7C15ull)));
    return std::string(dir) + name;
}

// Code objects on disk (only when SSA_JIT_CACHE_DIR names a directory): [key length][key][code length][code].  The key
// stored in the file is compared in full, the file name is only a hash of it.
bool LoadFromDisk(const std::string& key, std::vector<char>& code) {
    const std::string path = CacheFile(key);
    if (path.empty()) return false;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    uint64_t klen = 0, clen = 0;
    if (std::fread(&klen, 8, 1, f) == 1 && klen == key.size()) {
        std::string stored(klen, '\0');
        if (std::fread(&stored[0], 1, klen, f) == klen && stored == key && std::fread(&clen, 8, 1, f) == 1 && clen > 0 && clen < (1ull << 30)) {
            code.resize(clen);
            ok = std::fread(code.data(), 1, clen, f) == clen;
        }
    }
    std::fclose(f);
    return ok;
}

void SaveToDisk(const std::string& key, const std::vector<char>& code) {
    const std::string path = CacheFile(key);
    if (path.empty()) return;
    const std::string tmp = path + ".tmp" + std::to_string(reinterpret_cast<uintptr_t>(&code));
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    const uint64_t klen = key.size(), clen = code.size();
    const bool ok = std::fwrite(&klen, 8, 1, f) == 1 && std::fwrite(key.data(), 1, klen, f) == klen && std::fwrite(&clen, 8, 1, f) == 1 &&
                    std::fwrite(code.data(), 1, clen, f) == clen;
    std::fclose(f);
    if (ok) std::rename(tmp.c_str(), path.c_str()); else std::remove(tmp.c_str());
}

// Runs the compiler.  Called without State::Lock.
bool Compile(const Job& job, std::vector<char>& code, std::string& why) {
    State& st = St();
    if (LoadFromDisk(job.Key, code)) {
        ++st.FromDisk;
        return true;
    }
    {
        std::lock_guard<std::mutex> g(st.RtcLock);
        if (!st.R.Load(why)) return false;
    }
    // translation unit: the constant image, the variant's defines, then the kernel header
    std::ostringstream src;
    src << "#define SSA_SPEC 1\n" << job.Defines;
    src << "alignas(16) static constexpr unsigned int kSpecImage[] = {";
    for (size_t i = 0; i < job.Image.size(); ++i) {
        if (i % 16 == 0) src << "\n";
        src << job.Image[i] << "u,";
    }
    src << "\n};\n#include \"" << job.MainHeader << "\"\n";
    const std::string text = src.str();
    if (const char* dump = std::getenv("SSA_JIT_DUMP")) {  // diagnostics: the translation unit, for offline inspection
        static std::atomic<int> dumps{0};  // (a file per build: <path>.0, <path>.1, ... with the entry point in a comment)
        const std::string path = std::string(dump) + "." + std::to_string(dumps.fetch_add(1));
        if (FILE* f = std::fopen(path.c_str(), "w")) {
            std::fprintf(f, "// %s\n", job.MainHeader.c_str());
            std::fwrite(text.data(), 1, text.size(), f);
            std::fclose(f);
        }
    }
    // one compilation at a time (a queue of builds, not of scans: nobody scans under this lock)
    static std::mutex compileLock;
    std::lock_guard<std::mutex> cg(compileLock);
    hiprtcProgram prog = nullptr;
    if (st.R.CreateProgram(&prog, text.c_str(), "ssa_spec.hip", kEmbeddedCount, const_cast<const char**>(kEmbeddedSources),
                           const_cast<const char**>(kEmbeddedNames)) != HIPRTC_SUCCESS) {
        why = "hiprtcCreateProgram failed";
        return false;
    }
    const auto t0 = Clock::now();
    const std::string archOpt = "--offload-arch=" + job.Arch;
    // The step program's loops are unrolled by pragma with the whole interpreter inlined in their bodies: the
    // default size limit of pragma-forced unrolling (16 K instructions before folding) is too small for that
    // and, once exceeded, silently leaves the generic interpreter in the built kernel.
    // No contraction of a multiply and a following add into one fused operation: in the straight-line code of a
    // built kernel two instructions of the step program sit next to each other, and the reference rounds after
    // each of them (a * b + c differed in the last bit on a quarter of the rows).  Where a fused operation is
    // wanted (the register-held groups) the source asks for it by name.
    std::vector<const char*> opts = {archOpt.c_str()};
    for (const char* o : kCompileOptions) opts.push_back(o);
    const hiprtcResult rc = st.R.CompileProgram(prog, int(opts.size()), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)st.R.GetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) (void)st.R.GetProgramLog(prog, &log[0]);
        (void)st.R.DestroyProgram(&prog);
        why = "run-time compilation failed: " + log.substr(0, 2000);
        return false;
    }
    size_t n = 0;
    (void)st.R.GetCodeSize(prog, &n);
    code.resize(n);
    if (n) (void)st.R.GetCode(prog, code.data());
    (void)st.R.DestroyProgram(&prog);
    ++st.Compiled;
    st.BuildMicros += int64_t(std::chrono::duration<double, std::micro>(Clock::now() - t0).count());
    SaveToDisk(job.Key, code);
    return true;
}

// Compiles and publishes the outcome in the entry (which is in state BUILDING and therefore stays in the map).
void BuildAndPublish(const Job& job) {
    std::vector<char> code;
    std::string why;
    const bool ok = Compile(job, code, why);
    State& st = St();
    std::lock_guard<std::mutex> g(st.Lock);
    auto it = st.Entries.find(job.Key);
    if (it == st.Entries.end()) return;
    Entry& e = it->second;
    if (ok) {
        e.Code = std::make_shared<std::vector<char>>(std::move(code));
        e.State = Entry::READY;
    } else {
        e.Why = why;
        e.State = Entry::FAILED;
    }
    e.LastUse = Clock::now();
}

// The compiler's libraries are loaded when they are first needed, so their exit handlers run before the destructor
// of State: a build still running on the worker thread at that time would work with objects already destroyed.  An
// exit handler registered after those libraries were loaded runs before theirs, and waits for the worker.
void StopWorkerAtExit() { St().StopWorker(); }

void WorkerLoop() {
    State& st = St();
    bool registered = false;
    for (;;) {
        Job job;
        {
            std::unique_lock<std::mutex> g(st.QueueLock);
            st.QueueCv.wait(g, [&] { return st.Stop || !st.Queue.empty(); });
            if (st.Stop) return;
            job = std::move(st.Queue.front());
            st.Queue.pop_front();
        }
        BuildAndPublish(job);
        if (!registered) {  // (after the first build: everything the compiler loads is loaded now)
            std::atexit(StopWorkerAtExit);
            registered = true;
        }
    }
}

// Called with State::Lock held: drops the least recently used kernels beyond the bounds.  A kernel is only unloaded
// after kEvictIdle without a request for it, so that no launch of it can still be queued on a stream.
void EvictIfNeeded(State& st) {
    size_t ready = 0;
    for (auto& kv : st.Entries) ready += kv.second.State == Entry::READY;
    if (ready <= kMaxBuiltKernels && st.Entries.size() <= kMaxEntries) return;
    const auto now = Clock::now();
    std::vector<std::pair<Clock::time_point, const std::string*>> order;
    for (auto& kv : st.Entries) {
        if (kv.second.State == Entry::BUILDING) continue;
        if (now - kv.second.LastUse < kEvictIdle) continue;
        order.emplace_back(kv.second.LastUse, &kv.first);
    }
    std::sort(order.begin(), order.end(), [](auto& a, auto& b) { return a.first < b.first; });
    for (auto& o : order) {
        if (ready <= kMaxBuiltKernels * 3 / 4 && st.Entries.size() <= kMaxEntries * 3 / 4) break;
        auto it = st.Entries.find(*o.second);
        if (it->second.State == Entry::READY) {
            for (auto& d : it->second.PerDevice) {
                if (d.second.Module) {
                    (void)hipSetDevice(d.first);
                    (void)hipModuleUnload(d.second.Module);
                }
            }
            --ready;
        }
        st.Entries.erase(it);
        ++st.Evicted;
    }
}

}  // namespace

KernelJit& KernelJit::Instance() {
    static KernelJit j;
    return j;
}

void KernelJit::Stats(int64_t& built, double& buildMs, int64_t& launches) const {
    built = St().Compiled.load() + St().FromDisk.load();
    buildMs = double(St().BuildMicros.load()) / 1e3;
    launches = St().Launches.load();
}

void KernelJit::NoteLaunch() { ++St().Launches; }

void KernelJit::Shutdown() { St().StopWorker(); }

bool JitEnabled() {
    const char* mode = std::getenv("SSA_JIT");
    return !(mode && mode[0] == '0');
}

bool KernelJit::Get(int device, const char* mainHeader, const char* entry, const void* image, size_t imageBytes,
                    const std::string& defines, uint32_t ldsBytes, int64_t rows, SpecKernel& out, std::string& why) {
    State& st = St();
    why.clear();
    Job job;
    bool buildHere = false;
    hipFunction_t fn = nullptr;
    {
        std::unique_lock<std::mutex> g(st.Lock);
        std::string& arch = st.Arch[device];
        if (arch.empty()) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
                (void)hipGetLastError();
                why = "no device properties";
                return false;
            }
            arch = prop.gcnArchName;
        }
        std::string key;
        key.reserve(imageBytes + defines.size() + 256);
        key.append(mainHeader).push_back('\0');
        key.append(entry).push_back('\0');
        key.append(defines).push_back('\0');
        key.append(arch).push_back('\0');
        key.append(BuildFingerprint()).push_back('\0');
        key.append(static_cast<const char*>(image), imageBytes);
        Entry& e = st.Entries[key];
        e.LastUse = Clock::now();
        if (e.State == Entry::FAILED) {
            why = e.Why;
            return false;
        }
        if (e.State == Entry::BUILDING) return false;  // another thread, or the worker, is at it: the generic kernel runs meanwhile
        if (e.State == Entry::NONE) {
            const char* mode = std::getenv("SSA_JIT");
            const bool now = mode && mode[0] == '1';
            e.SeenRows += rows;
            if (!now && e.SeenRows < kJitMinRows) return false;
            e.State = Entry::BUILDING;
            job.Key = key;
            job.MainHeader = mainHeader;
            job.Defines = defines;
            job.Arch = arch;
            job.Image.assign(static_cast<const uint32_t*>(image), static_cast<const uint32_t*>(image) + imageBytes / 4);
            // A scan that is large by itself waits for its kernel (the build costs what the generic kernel needs for
            // some ten billion rows, and whoever scans that much at once will be back); scans that only add up to
            // the threshold -- the batches of a shard -- never wait: a worker thread builds while they keep running
            // the generic kernel.
            const char* async = std::getenv("SSA_JIT_ASYNC");
            buildHere = now || rows >= kJitMinRows || (async && async[0] == '0');
            EvictIfNeeded(st);
            if (!buildHere) {
                std::lock_guard<std::mutex> q(st.QueueLock);
                if (st.Stop) {  // shut down: no new builds
                    e.State = Entry::NONE;
                    return false;
                }
                if (!st.WorkerStarted) {
                    st.Worker = std::thread(WorkerLoop);
                    st.WorkerStarted = true;
                    std::atexit(StopWorkerAtExit);
                }
                st.Queue.push_back(std::move(job));
                st.QueueCv.notify_one();
                return false;
            }
        }
        if (!buildHere) {  // READY
            Entry::Loaded& l = e.PerDevice[device];
            if (!l.Function) {
                (void)hipSetDevice(device);
                hipError_t err = hipModuleLoadData(&l.Module, e.Code->data());
                if (err == hipSuccess) err = hipModuleGetFunction(&l.Function, l.Module, entry);
                if (err != hipSuccess) {
                    (void)hipGetLastError();
                    e.Why = std::string("cannot load the compiled kernel: ") + hipGetErrorString(err);
                    e.State = Entry::FAILED;
                    why = e.Why;
                    return false;
                }
            }
            fn = l.Function;
        }
    }
    if (buildHere) {
        BuildAndPublish(job);  // the lock is not held: other programs' scans go on
        return Get(device, mainHeader, entry, image, imageBytes, defines, ldsBytes, 0, out, why);
    }
    out.Function = fn;
    int n = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, ldsBytes) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 1;
    }
    out.BlocksPerCU = n;
    return true;
}

}  // namespace ssa
