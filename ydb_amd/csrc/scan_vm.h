// Device-side scan VM: evaluates the per-step expression program (vm.h) for a tile of rows.
//
// Work decomposition: a workgroup of kBlock = 256 lanes (4 wavefronts of 64) owns a tile of
// kBlock * U consecutive rows; wave w owns the 64 * U consecutive rows tile + w * 64 * U ..., lane t of it evaluates
// rows tile + w * 64 * U + u * 64 + t (u = 0..U-1): a wave stages, reads and writes only its own rows, so every
// column access of a wavefront is 64 consecutive elements -- fully coalesced at any element width.
// At the start of a tile every fixed-width source column is staged into LDS by direct global->LDS
// loads (no VGPRs, all columns in flight together, 16 bytes per lane); the interpreter then runs the
// step's instruction list reading operands from LDS.  The VM's R registers live in LDS as well, so
// the per-lane register state is just the accumulator -- occupancy is set by the LDS plan, and several
// workgroups per CU overlap one group's staging with another's evaluation.
// Instruction words, column descriptors and constants live in the kernel-argument segment and the
// program counter is wave-uniform, so decode is scalar work and every `switch` is a scalar branch.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "kernels.h"
#include "vm_ops.h"

// Two ways to build these kernels.  Ahead of time (the library): one generic kernel per sink interprets whatever step
// program arrives in the kernel arguments.  At run time, for one step program (SSA_SPEC defined by the generated
// prelude, which also carries kSpecImage: the bytes of the launch's {ScanArgs, Sink} pair with the pointers cleared):
// every table read below goes to that constant image instead, loops over instructions, columns, keys and aggregates
// have constant trip counts and unroll, and the interpreter's switches fold away -- the compiler turns the same
// source into the straight-line code of that one program.  Addresses, row counts and the like still come from the
// kernel arguments.
#ifdef SSA_SPEC
#define SSA_UNROLL_STATIC _Pragma("unroll")
#else
#define SSA_UNROLL_STATIC
#endif

namespace ssa {

template <int U>
struct Val {
    uint64_t v[U];
    uint32_t ok;  // bit u: row u is non-null
};

template <int U>
__device__ __forceinline__ constexpr uint32_t AllMask() {
    return U >= 32 ? 0xFFFFFFFFu : (1u << (U & 31)) - 1u;
}

__device__ __forceinline__ uint64_t LoadRaw(const void* base, int width, int64_t row) {
    switch (width) {
        case 1: return static_cast<const uint8_t*>(base)[row];
        case 2: return static_cast<const uint16_t*>(base)[row];
        case 4: return static_cast<const uint32_t*>(base)[row];
        default: return static_cast<const uint64_t*>(base)[row];
    }
}

__device__ __forceinline__ uint32_t LoadBit(const uint8_t* bits, int64_t bit) {
    return (bits[bit >> 3] >> (bit & 7)) & 1u;
}

__device__ __forceinline__ uint64_t ConvertRaw(uint64_t raw, int type) {
    switch (type) {
        case T_I8: return uint64_t(int64_t(int8_t(raw)));
        case T_I16: return uint64_t(int64_t(int16_t(raw)));
        case T_I32: return uint64_t(int64_t(int32_t(raw)));
        case T_F32: return F64ToBits(double(BitsToF32(uint32_t(raw))));
        default: return raw;
    }
}

// Payload -> stored representation of `type` (low TypeWidth bytes are meaningful).
__device__ __forceinline__ uint64_t PayloadToStored(uint64_t v, int type) {
    if (type == T_F32) {
        float f = float(BitsToF64(v));
        uint32_t b;
        __builtin_memcpy(&b, &f, 4);
        return b;
    }
    return v;
}

__device__ __forceinline__ void StoreTyped(void* base, int width, int64_t idx, uint64_t v) {
    switch (width) {
        case 1: static_cast<uint8_t*>(base)[idx] = uint8_t(v); break;
        case 2: static_cast<uint16_t*>(base)[idx] = uint16_t(v); break;
        case 4: static_cast<uint32_t*>(base)[idx] = uint32_t(v); break;
        default: static_cast<uint64_t*>(base)[idx] = v; break;
    }
}

typedef const __attribute__((address_space(1))) void* GlobalPtr;
typedef __attribute__((address_space(3))) void* LdsPtr;

// Dynamic LDS of a scan workgroup: [staged column tiles][R register file][sink area].
extern __shared__ __attribute__((aligned(16))) unsigned char gLds[];

__device__ __forceinline__ uint64_t LdsLoadRaw(const unsigned char* base, int width, int idx) {
    switch (width) {
        case 1: return reinterpret_cast<const uint8_t*>(base)[idx];
        case 2: return reinterpret_cast<const uint16_t*>(base)[idx];
        case 4: return reinterpret_cast<const uint32_t*>(base)[idx];
        default: return reinterpret_cast<const uint64_t*>(base)[idx];
    }
}

// Rarely executed, code-heavy helpers are kept out of line so the interpreter loop stays small enough
// for the instruction cache (a 64-bit integer division alone expands to a few hundred instructions).
__device__ __noinline__ ArithResult ArithDivOutlined(bool mod, int cls, int dialect, int rtype, uint64_t x, uint64_t y) {
    return ArithDiv(mod, cls, dialect, rtype, x, y);
}
__device__ __noinline__ bool StrMatchOutlined(int kind, bool icase, const uint8_t* s, int32_t n, const uint8_t* p, int32_t m) {
    return StrMatch(kind, icase, s, n, p, m);
}
__device__ __noinline__ int StrCompareOutlined(const uint8_t* a, int32_t n, const uint8_t* b, int32_t m) {
    return StrCompare(a, n, b, m);
}
__device__ __noinline__ ArithResult ApplyFuncOutlined(int fn, int t, uint64_t x, uint64_t y) { return ApplyFunc(fn, t, x, y); }
__device__ __noinline__ uint64_t CastValOutlined(uint64_t v, int from, int to) { return CastVal(v, from, to); }
__device__ __noinline__ bool FitsTypeOutlined(uint64_t v, int from, int to) { return FitsType(v, from, to); }

// lt / eq / unordered bit masks of U element pairs for one value class.
template <int U>
__device__ __forceinline__ void CompareMasks(int cls, const uint64_t* x, const uint64_t* y, uint32_t& lt, uint32_t& eq,
                                             uint32_t& un) {
    lt = eq = un = 0;
    switch (cls) {
        case CLS_I:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lt |= (int64_t(x[u]) < int64_t(y[u]) ? 1u : 0u) << u;
                eq |= (x[u] == y[u] ? 1u : 0u) << u;
            }
            break;
        case CLS_U:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lt |= (x[u] < y[u] ? 1u : 0u) << u;
                eq |= (x[u] == y[u] ? 1u : 0u) << u;
            }
            break;
        case CLS_IU:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool neg = int64_t(x[u]) < 0;
                lt |= ((neg || x[u] < y[u]) ? 1u : 0u) << u;
                eq |= ((!neg && x[u] == y[u]) ? 1u : 0u) << u;
            }
            break;
        case CLS_UI:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool neg = int64_t(y[u]) < 0;
                lt |= ((!neg && x[u] < y[u]) ? 1u : 0u) << u;
                eq |= ((!neg && x[u] == y[u]) ? 1u : 0u) << u;
            }
            break;
        default:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double p = BitsToF64(x[u]), q = BitsToF64(y[u]);
                lt |= (p < q ? 1u : 0u) << u;
                eq |= (p == q ? 1u : 0u) << u;
                un |= ((p != p || q != q) ? 1u : 0u) << u;
            }
            break;
    }
}

__device__ __forceinline__ uint32_t CompareResult(int cmp, uint32_t lt, uint32_t eq, uint32_t un) {
    switch (cmp) {
        case CMP_EQ: return eq;
        case CMP_NE: return ~eq;
        case CMP_LT: return lt;
        case CMP_LE: return lt | eq;
        case CMP_GT: return ~(un | lt | eq);
        default: return ~(un | lt);
    }
}

__device__ __forceinline__ uint32_t ReadLane32(uint32_t v, int lane) {
    return uint32_t(__builtin_amdgcn_readlane(int(v), lane));
}
__device__ __forceinline__ uint64_t ReadLane64(uint64_t v, int lane) {
    const uint32_t lo = ReadLane32(uint32_t(v), lane), hi = ReadLane32(uint32_t(v >> 32), lane);
    return (uint64_t(hi) << 32) | lo;
}

// A value every lane of the wave holds alike, moved to scalar registers (values loaded through a pointer the compiler
// cannot prove uniform would otherwise occupy vector registers for the whole kernel).
__device__ __forceinline__ uint64_t Uniform64(uint64_t v) {
    const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
    const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
    return (uint64_t(hi) << 32) | lo;
}

template <int U>
struct Machine {
    static constexpr int kTile = kBlock * U;
    const ScanArgs& a;   // kernel-argument segment (read once at kernel start)
    int64_t row0;        // row of u = 0 for this lane
    uint32_t act;        // bit u: Row(u) < nrows
    bool tileFull;       // every row of the current tile is below nrows
    uint32_t P;          // bit u: row passes every Filter so far
    uint32_t err;        // ScanStatus bits raised by this lane
    Val<U> A;

    // The step program, spread over the lanes of each wave: lane i holds instruction i, the
    // description of column slot i, constant i and staging piece i.  Decode is a v_readlane with a
    // wave-uniform index -- no memory access, no latency -- instead of dependent scalar loads.
    uint32_t tCode0, tCode1, tCode2;         // pre-decoded instruction [lane]: dispatch word, operand payload, generic form
    uint32_t tColOff, tColType, tColW;       // column slot (lane & 15): LDS offset (or -1), TypeId, width
    uint32_t tColValBit, tColValidBit, tColCharsBytes;
    uint64_t tColVal, tColValid, tColChars;
    uint32_t tRegOff;                        // R register (lane): LDS offset of its value tile
    uint64_t tConstBits;                     // constant (lane)
    uint32_t tConstValid;
    uint32_t tColStageW;                     // column slot: bytes per row of the staged buffer
    uint32_t tColCharsLds;                   // column slot: LDS offset of this wave's string-head area, or -1
    uint32_t tColEndOff;                     // column slot: LDS offset of this wave's end-offset slot (staged string columns), or -1
    // string heads staged for the current tile, per column slot (lane c): first staged byte (offset inside the
    // column's chars) and number of staged bytes; 0 bytes = not staged
    uint32_t tCharsBegin, tCharsBytes, tCharsSkew;
    int waveBase;                            // first tile row of this wave: wave * 64 * U
    bool charsPending;                       // a string-head copy of this tile has been issued and not yet awaited
    // String heads one tile ahead: lane c holds, for string column c, the offsets of the first and the last row of the
    // wave's slice of tile `boundsTile`, read from memory while the previous tile was evaluated.  With them the head
    // copy of a tile is issued together with its columns -- one memory round trip per tile instead of two.
    uint32_t boundsFirst, boundsLast;
    int64_t boundsTile;
    bool anyCharsStage;                      // some string column has a head area
    uint32_t offsBad;                        // non-zero: an offset of a column scanned as fixed-width values is off (OffsetCheck)
    int ncols;
    int64_t nrows;
    int ncode;
    const DevColumn* cols;       // the batch's columns: of the launch, or of this workgroup's slice
    const OffsetCheck* checks;
#ifdef SSA_SPEC
    // the offset checks' addresses in scalar registers (read once: a load through `checks` per tile would sit in front
    // of the tile's offset loads)
    uint64_t ckOffsets[kMaxOffsetChecks];
    uint32_t ckFirst[kMaxOffsetChecks];
#endif

    // `slice`: this workgroup scans one slice of a launch over several (SliceDesc), else nullptr
    __device__ __forceinline__ explicit Machine(const ScanArgs& args, const SliceDesc* slice = nullptr) : a(args), err(0) {
        const int lane = threadIdx.x & 63;
        nrows = slice != nullptr ? int64_t(Uniform64(uint64_t(slice->nrows))) : a.nrows;
        cols = slice != nullptr ? slice->cols : a.cols;
        checks = slice != nullptr ? slice->checks : a.checks;
#ifdef SSA_SPEC
#pragma unroll
        for (int i = 0; i < kMaxOffsetChecks; ++i) {
            const bool used = i < reinterpret_cast<const ScanArgs*>(kSpecImage)->nchecks;
            ckOffsets[i] = used ? Uniform64(reinterpret_cast<uint64_t>(checks[i].offsets)) : 0ull;
            ckFirst[i] = used ? uint32_t(__builtin_amdgcn_readfirstlane(int(checks[i].first))) : 0u;
        }
#endif
        ncode = a.ncode;
        tCode0 = lane < ncode ? a.code[3 * lane] : 0u;
        tCode1 = lane < ncode ? a.code[3 * lane + 1] : 0u;
        tCode2 = lane < ncode ? a.code[3 * lane + 2] : 0u;
        const int c = lane & (kMaxCols - 1);
        tColOff = uint32_t(a.lds_col_off[c]);
        tColType = uint32_t(a.cols[c].type);
        tColW = uint32_t(TypeWidth(a.cols[c].type));
        tColValBit = uint32_t(cols[c].val_bit_off);
        tColValidBit = uint32_t(cols[c].valid_bit_off);
        tColCharsBytes = uint32_t(cols[c].chars_bytes);
        tColVal = reinterpret_cast<uint64_t>(cols[c].values);
        tColValid = reinterpret_cast<uint64_t>(cols[c].validity);
        tColChars = reinterpret_cast<uint64_t>(cols[c].chars);
        tRegOff = uint32_t(a.lds_reg_offs[lane & (kMaxRegs - 1)]);
        const int k = lane < kMaxConsts ? lane : 0;
        tConstBits = a.consts[k].bits;
        tConstValid = uint32_t(a.consts[k].valid);
        tColStageW = uint32_t(StageWidth(a.cols[c].type));
        tColCharsLds = a.lds_chars_off[c] < 0 ? 0xFFFFFFFFu
                                              : uint32_t(a.lds_chars_off[c]) + (threadIdx.x >> 6) * uint32_t(CharsStageBytes(U));
        tColEndOff = a.lds_end_off[c] < 0 ? 0xFFFFFFFFu : uint32_t(a.lds_end_off[c]) + (threadIdx.x >> 6) * uint32_t(kEndSlotBytes);
        boundsFirst = boundsLast = 0;
        boundsTile = -1;
        tCharsBegin = tCharsBytes = tCharsSkew = 0;
        anyCharsStage = false;
        SSA_UNROLL_STATIC
        for (int col = 0; col < S().ncols; ++col) anyCharsStage = anyCharsStage || S().lds_chars_off[col] >= 0;
        charsPending = false;
        offsBad = 0;
        // The tables above come from memory.  Using them here makes the compiler wait for those loads now: left pending,
        // its wait lands before their first use inside the tile loop -- between the copies of a tile, which it would
        // serialise (the explicit waits of BeginTile are invisible to it).
        asm volatile("" ::"v"(tColOff), "v"(tColType), "v"(tColValBit), "v"(tColValidBit), "v"(tColCharsBytes), "v"(tColVal),
                     "v"(tColValid), "v"(tColChars), "v"(tRegOff), "v"(tConstBits), "v"(tConstValid), "v"(tColCharsLds), "v"(tColEndOff),
                     "v"(tCode0), "v"(tCode1), "v"(tCode2));
        waveBase = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)) * (64 * U);
        ncols = a.ncols;
    }

#ifdef SSA_SPEC
    static constexpr bool kStatic = true;
    __device__ __forceinline__ const ScanArgs& S() const { return *reinterpret_cast<const ScanArgs*>(kSpecImage); }
#else
    static constexpr bool kStatic = false;
    __device__ __forceinline__ const ScanArgs& S() const { return a; }
#endif
    // Table reads: a v_readlane of the lane tables, or the constant image.
    __device__ __forceinline__ uint32_t Code0(int pc) const { return kStatic ? S().code[3 * pc] : ReadLane32(tCode0, pc); }
    __device__ __forceinline__ uint32_t Code1(int pc) const { return kStatic ? S().code[3 * pc + 1] : ReadLane32(tCode1, pc); }
    __device__ __forceinline__ uint32_t Code2(int pc) const { return kStatic ? S().code[3 * pc + 2] : ReadLane32(tCode2, pc); }
    __device__ __forceinline__ uint32_t ColOff(int c) const { return kStatic ? uint32_t(S().lds_col_off[c]) : ReadLane32(tColOff, c); }
    __device__ __forceinline__ uint32_t ColType(int c) const { return kStatic ? uint32_t(S().cols[c].type) : ReadLane32(tColType, c); }
    __device__ __forceinline__ uint32_t ColW(int c) const { return kStatic ? uint32_t(TypeWidth(S().cols[c].type)) : ReadLane32(tColW, c); }
    __device__ __forceinline__ uint32_t ColStageW(int c) const {
        return kStatic ? uint32_t(StageWidth(S().cols[c].type)) : ReadLane32(tColStageW, c);
    }
    __device__ __forceinline__ uint32_t ColCharsLds(int c) const {
        if (!kStatic) return ReadLane32(tColCharsLds, c);
        return S().lds_chars_off[c] < 0 ? 0xFFFFFFFFu : uint32_t(S().lds_chars_off[c]) + uint32_t(waveBase / (64 * U)) * uint32_t(CharsStageBytes(U));
    }
    __device__ __forceinline__ uint32_t ColEndOff(int c) const {
        if (!kStatic) return ReadLane32(tColEndOff, c);
        return S().lds_end_off[c] < 0 ? 0xFFFFFFFFu : uint32_t(S().lds_end_off[c]) + uint32_t(waveBase / (64 * U)) * uint32_t(kEndSlotBytes);
    }
    __device__ __forceinline__ uint32_t RegOff(int r) const { return kStatic ? uint32_t(S().lds_reg_offs[r]) : ReadLane32(tRegOff, r); }
    // The value of a constant is data of the launch, not of the program a kernel is built for: a built kernel reads it from
    // the kernel arguments (a scalar load the compiler hoists out of the tile loop), so that the same query with another
    // literal runs the same kernel.  What shapes the code stays with the program: a constant's type, whether it is null,
    // and, for a string, where its bytes are and how many (the bytes themselves are read through Pattern()).
    __device__ __forceinline__ uint64_t ConstBits(int k) const {
        if (!kStatic) return ReadLane64(tConstBits, k);
        return IsStringLike(S().consts[k].type) ? S().consts[k].bits : a.consts[k].bits;
    }
    __device__ __forceinline__ const uint8_t* Pattern() const { return a.pattern; }
    // Eight bytes of the pattern buffer from byte `at` on, read as words of the kernel-argument segment (scalar loads; in a
    // built kernel `at` is a constant).  `at` + 8 stays inside the buffer's 256 bytes plus the 8 of the field behind it.
    __device__ __forceinline__ uint64_t PatternWord(uint32_t at) const {
        typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
        ConstWords w = (ConstWords)(uintptr_t)(a.pattern);
        const uint32_t i = at >> 2, sh = 8u * (at & 3u);
        const uint64_t lo = uint64_t(w[i]) | (uint64_t(w[i + 1]) << 32);
        const uint64_t hi = uint64_t(w[i + 2]);
        return sh == 0 ? lo : (lo >> sh) | (hi << (64u - sh));
    }
    __device__ __forceinline__ uint32_t ConstValid(int k) const { return kStatic ? uint32_t(S().consts[k].valid) : ReadLane32(tConstValid, k); }
    __device__ __forceinline__ int NumCode() const { return kStatic ? S().ncode : ncode; }
    __device__ __forceinline__ int NumCols() const { return kStatic ? S().ncols : ncols; }

    __device__ __forceinline__ int64_t Row(int u) const { return row0 + int64_t(u) * 64; }
    // index of the lane's u-th row inside the tile (staged columns, R registers)
    __device__ __forceinline__ int Idx(int u) const { return waveBase + u * 64 + int(threadIdx.x & 63); }

    // Stage the wave's rows of every staged source column into LDS with direct global->LDS loads
    // (global_load_lds_dwordx4: 16 bytes per lane, no VGPRs involved).  Waves do not wait for each other: each one
    // loads, and later reads, only its own 64 * U rows, so a vmcnt(0) of its own is all the ordering needed.  In the
    // last, partial tile the lanes whose 16-byte chunk starts past the end of the column issue nothing (those rows are
    // inactive); a chunk that straddles the end reads at most 15 bytes of the allocation's tail padding.
    // `nextBase`: the tile this workgroup evaluates after this one (any value >= nrows when there is none).
    __device__ __forceinline__ void BeginTile(int64_t tileBase, int64_t nextBase) {
        const int lane = threadIdx.x & 63;
        row0 = tileBase + waveBase + lane;
        const bool full = (tileBase + kTile <= nrows);
        tileFull = full;
        act = AllMask<U>();
        if (!full) {  // (the same for the whole workgroup: only the table's last tile compares its rows with the end)
            act = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) act |= (Row(u) < nrows ? 1u : 0u) << u;
        }
        const int64_t slice0 = tileBase + waveBase;  // first row of the wave's slice
        // the wave's own LDS traffic of the previous tile (register stores included) has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // slice bounds of the next tile's strings, for its BeginTile
        uint32_t nextFirst = 0, nextLast = 0;
        const int64_t nextSlice0 = nextBase + waveBase;
        const bool nextWanted = anyCharsStage && nextSlice0 < nrows;
        if (nextWanted && lane < kMaxCols && lane < NumCols() && tColCharsLds != 0xFFFFFFFFu) {
            const int32_t* offs = reinterpret_cast<const int32_t*>(tColVal);
            const int64_t lastRow = (nextSlice0 + 64 * U <= nrows ? nextSlice0 + 64 * U : nrows) - 1;
            nextFirst = uint32_t(offs[nextSlice0]);
            nextLast = uint32_t(offs[lastRow]);
        }
        const bool haveBounds = anyCharsStage && boundsTile == tileBase;
        if (!(S().debug_flags & 1)) {
            SSA_UNROLL_STATIC
            for (int c = 0; c < NumCols(); ++c) {
                const int off = int(ColOff(c));
                if (off < 0) continue;
                const int w = int(ColStageW(c));
                const unsigned char* src = reinterpret_cast<const unsigned char*>(ReadLane64(tColVal, c)) + slice0 * w;
                const int sliceBytes = 64 * U * w;
                for (int chunk = 0; chunk < sliceBytes; chunk += 1024) {
                    const int b = chunk + lane * 16;
                    if (b < sliceBytes && (full || (slice0 * w + b) < nrows * int64_t(w))) {
                        __builtin_amdgcn_global_load_lds((GlobalPtr)(src + b), (LdsPtr)(gLds + off + waveBase * w + chunk), 16, 0, 0);
                    }
                }
                // a staged string column: the end offset of the slice's last row goes to the wave's slot (the tile of
                // offsets ends where the next wave's begins, and that wave's copy is not waited for here)
                const uint32_t endOff = ColEndOff(c);
                if (endOff != 0xFFFFFFFFu && slice0 < nrows && lane == 0) {
                    const int64_t endRow = slice0 + 64 * U < nrows ? slice0 + 64 * U : nrows;
                    __builtin_amdgcn_global_load_lds((GlobalPtr)(reinterpret_cast<const int32_t*>(src - slice0 * w) + endRow),
                                                     (LdsPtr)(gLds + endOff), 4, 0, 0);
                }
            }
        }
        if (haveBounds) StageStringHeads(slice0, true);
        // Offsets of string columns scanned as fixed-width values: lane t takes offsets U * t .. U * t + U - 1 of the wave's
        // slice straight into registers (16 bytes per load).  A kernel built for the program issues these loads together
        // with the copies above; the generic kernels, which have no registers to spare (a spilled lane table is not
        // restored in the lanes that are inactive at the reload), load and compare one column at a time afterwards.
        const int64_t ckRow = slice0 + int64_t(U) * lane;
        // (the address comes out of a register or through a pointer of unknown provenance: without the address space
        // spelled out these would be FLAT loads, which count against the LDS counter as well and stall the tile's LDS work)
        typedef const __attribute__((address_space(1))) int32_t* GlobalI32;
        typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) U32x4* GlobalU4;
        auto loadOffsets = [&](int i, uint32_t* v) {
#ifdef SSA_SPEC
            GlobalI32 offs = (GlobalI32)(ckOffsets[i]) + ckRow;
#else
            GlobalI32 offs = (GlobalI32)(reinterpret_cast<uint64_t>(checks[i].offsets)) + ckRow;
#endif
            if (full) {
                if constexpr (U % 4 == 0) {
#pragma unroll
                    for (int j = 0; j < U; j += 4) {
                        const U32x4 q = *(GlobalU4)(offs + j);
                        v[j] = q.x, v[j + 1] = q.y, v[j + 2] = q.z, v[j + 3] = q.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < U; ++j) v[j] = uint32_t(offs[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) v[j] = ckRow + j < nrows ? uint32_t(offs[j]) : 0u;
            }
        };
        auto compareOffsets = [&](int i, const uint32_t* v) {
#ifdef SSA_SPEC
            const uint32_t first = ckFirst[i], len = S().checks[i].len;
#else
            const uint32_t first = checks[i].first, len = S().checks[i].len;
#endif
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t want = first + len * uint32_t(ckRow + j);
                offsBad |= (full || ckRow + j < nrows) ? (v[j] ^ want) : 0u;
            }
        };
#ifdef SSA_SPEC
        uint32_t ck[kMaxOffsetChecks][U];
#pragma unroll
        for (int i = 0; i < S().nchecks; ++i) loadOffsets(i, ck[i]);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SSA_SPEC
#pragma unroll
        for (int i = 0; i < S().nchecks; ++i) compareOffsets(i, ck[i]);
#else
        for (int i = 0; i < a.nchecks; ++i) {
            uint32_t v[U];
            loadOffsets(i, v);
            compareOffsets(i, v);
        }
#endif
        asm volatile("" ::"v"(nextFirst), "v"(nextLast));  // (the compiler's own wait for these two loads belongs here too)
        if (haveBounds) charsPending = false;  // the head copies were part of that wait
        if (anyCharsStage && !haveBounds) StageStringHeads(slice0, false);
        boundsFirst = nextFirst;
        boundsLast = nextLast;
        boundsTile = nextWanted ? nextBase : -1;
    }

    // Second staging phase, for string columns: with the offsets of the wave's rows now in LDS, copy the bytes from
    // the first row's string to 8 bytes past the start of the last row's string into the wave's head area.  Nothing
    // waits for this copy here: WaitStringHeads() is called right before the first use, after other work.
    // `fromBounds`: the slice's first and last offsets were read ahead (boundsFirst / boundsLast); otherwise they come
    // from the staged offsets, which must have arrived.
    __device__ __forceinline__ void StageStringHeads(int64_t slice0, bool fromBounds) {
        const int lane = threadIdx.x & 63;
        charsPending = false;
        const int64_t last = (slice0 + 64 * U <= nrows ? slice0 + 64 * U : nrows) - 1;  // last row of the slice
        SSA_UNROLL_STATIC
        for (int c = 0; c < NumCols(); ++c) {
            const uint32_t area = ColCharsLds(c);
            if (area == 0xFFFFFFFFu) continue;
            uint32_t begin = 0, bytes = 0, skew = 0;
            if (last >= slice0) {
                uint32_t b0, bl;
                if (fromBounds) {
                    b0 = ReadLane32(boundsFirst, c);
                    bl = ReadLane32(boundsLast, c);
                } else {
                    const int32_t* l = reinterpret_cast<const int32_t*>(gLds + int(ColOff(c)));
                    b0 = uint32_t(__builtin_amdgcn_readfirstlane(l[waveBase]));
                    bl = uint32_t(__builtin_amdgcn_readfirstlane(l[waveBase + int(last - slice0)]));
                }
                const uint8_t* chars = reinterpret_cast<const uint8_t*>(ReadLane64(tColChars, c));
                const uint32_t limit = ReadLane32(tColCharsBytes, c);
                uint32_t endb = bl + 8;
                if (endb > limit) endb = limit;
                skew = uint32_t(reinterpret_cast<uintptr_t>(chars + b0) & 15u);
                // the copy moves whole 16-byte chunks: it must fit the area and stay inside the chars buffer
                if (endb > b0 && endb - b0 + skew <= uint32_t(CharsStageBytes(U) - 16) &&
                    uint64_t(b0 - skew) + ((uint64_t(endb - b0 + skew) + 15u) & ~15ull) <= uint64_t(limit)) {
                    begin = b0;
                    bytes = endb - b0;
                    const uint8_t* src = chars + b0 - skew;
                    const uint32_t total = bytes + skew;
                    for (uint32_t chunk = 0; chunk < total; chunk += 1024) {
                        const uint32_t o = chunk + uint32_t(lane) * 16;
                        if (o < total) __builtin_amdgcn_global_load_lds((GlobalPtr)(src + o), (LdsPtr)(gLds + area + chunk), 16, 0, 0);
                    }
                    charsPending = true;
                }
            }
            if (lane == c) {
                tCharsBegin = begin;
                tCharsBytes = bytes;
                tCharsSkew = skew;
            }
        }
    }

    __device__ __forceinline__ void WaitStringHeads() {
        if (charsPending) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            charsPending = false;
        }
    }

    __device__ __forceinline__ uint32_t ValidityOf(int c) const {
        const uint8_t* validity = reinterpret_cast<const uint8_t*>(ReadLane64(tColValid, c));
        if (validity == nullptr) return AllMask<U>();
        const int64_t bitOff = int64_t(ReadLane32(tColValidBit, c));
        uint32_t ok = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((act >> u) & 1u) ok |= LoadBit(validity, Row(u) + bitOff) << u;
        }
        return ok;
    }

    __device__ __forceinline__ void FetchCol(int c, Val<U>& out) {
        const int type = int(ColType(c));
        const int w = int(ColW(c));
        if (w > 0 && w <= 8) {
            const int off = int(ColOff(c));
            if (off >= 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) out.v[u] = LdsLoadRaw(gLds + off, w, Idx(u));
            } else {
                const void* values = reinterpret_cast<const void*>(ReadLane64(tColVal, c));
#pragma unroll
                for (int u = 0; u < U; ++u) out.v[u] = ((act >> u) & 1u) ? LoadRaw(values, w, Row(u)) : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) out.v[u] = ConvertRaw(out.v[u], type);
        } else if (type == T_BOOL) {
            const uint8_t* bits = reinterpret_cast<const uint8_t*>(ReadLane64(tColVal, c));
            const int64_t bitOff = int64_t(ReadLane32(tColValBit, c));
#pragma unroll
            for (int u = 0; u < U; ++u) out.v[u] = ((act >> u) & 1u) ? LoadBit(bits, Row(u) + bitOff) : 0;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) out.v[u] = 0;  // strings / wide values carry only validity here
        }
        out.ok = ValidityOf(c);
    }

    // R registers live in LDS: a lane owns elements Idx(u) of every register, so no
    // barrier is needed between a store and a later fetch.
    __device__ __forceinline__ void FetchReg(int r, Val<U>& out) {
        const uint64_t* vals = reinterpret_cast<const uint64_t*>(gLds + RegOff(r));
#pragma unroll
        for (int u = 0; u < U; ++u) out.v[u] = vals[Idx(u)];
        out.ok = (gLds + S().lds_regok_off)[r * kBlock + threadIdx.x];
    }

    __device__ __forceinline__ void StoreReg(int r, const Val<U>& in) {
        uint64_t* vals = reinterpret_cast<uint64_t*>(gLds + RegOff(r));
#pragma unroll
        for (int u = 0; u < U; ++u) vals[Idx(u)] = in.v[u];
        (gLds + S().lds_regok_off)[r * kBlock + threadIdx.x] = uint8_t(in.ok);
    }

    __device__ __forceinline__ void FetchConst(int k, Val<U>& out) {
        const uint64_t bits = ConstBits(k);
#pragma unroll
        for (int u = 0; u < U; ++u) out.v[u] = bits;
        out.ok = ConstValid(k) ? AllMask<U>() : 0u;
    }

    __device__ __forceinline__ void Fetch(int sk, int si, Val<U>& out) {
        if (sk == SRC_COL) {
            FetchCol(si, out);
        } else if (sk == SRC_CONST) {
            FetchConst(si, out);
        } else {
            FetchReg(si, out);
        }
    }

    // Bytes of the string in row u of the lane (nothing for a row past the end of the batch).  Offsets come from the
    // staged tile when the column's offsets are in LDS.  A wave stages the offsets of its own rows only (and in the
    // last tile only those below `nrows`): the end offset of the last row of its slice, and of the batch's last row,
    // is in the wave's slot.
    __device__ __forceinline__ void StringAt(int c, int u, const uint8_t*& p, int32_t& n) const {
        const int32_t* offs = reinterpret_cast<const int32_t*>(ReadLane64(tColVal, c));
        const uint8_t* chars = reinterpret_cast<const uint8_t*>(ReadLane64(tColChars, c));
        const int off = int(ColOff(c));
        const int64_t row = Row(u);
        const bool active = (act >> u) & 1u;
        int32_t b = 0, e = 0;
        const uint32_t endOff = ColEndOff(c);
        if (off >= 0 && endOff != 0xFFFFFFFFu) {
            const int32_t* l = reinterpret_cast<const int32_t*>(gLds + off);
            const int idx = Idx(u);
            b = l[idx];
            const int32_t next = l[idx + 1 < waveBase + 64 * U ? idx + 1 : idx];
            const int32_t last = *reinterpret_cast<const int32_t*>(gLds + endOff);
            e = (idx + 1 < waveBase + 64 * U && row + 1 < nrows) ? next : last;
            if (!active) b = e = 0;
        } else if (active) {
            b = offs[row];
            e = offs[row + 1];
        }
        p = chars + b;
        n = e - b;
    }

    // Bytes of column c's chars buffer from address `p` on (0 when `p` lies outside the buffer).
    __device__ __forceinline__ int64_t CharsAvailable(int c, const uint8_t* p) const {
        const uint8_t* chars = reinterpret_cast<const uint8_t*>(ReadLane64(tColChars, c));
        const int64_t off = p - chars;
        const int64_t limit = int64_t(ReadLane32(tColCharsBytes, c));
        return (off < 0 || off > limit) ? 0 : limit - off;
    }

    // Substring search over the wave's whole slice at once.  The strings of the slice are one contiguous run of bytes:
    // the wave streams it in 1 KB pieces -- every lane loads 16 consecutive bytes (coalesced) while the previous piece
    // is searched -- and tests the pattern at each of its 16 start positions on 64-bit windows assembled from its own
    // and the next lane's bytes.  A hit (rare for a selective pattern) is mapped to its row by a binary search of the
    // staged offsets and, when the match lies inside that row, recorded in the wave's match words in LDS.  Patterns of
    // 1..16 bytes; `icase` folds ASCII letters as the per-row matcher does.  Needs the column's offsets staged, a full
    // tile and a 16-byte aligned chars buffer; returns false (nothing written) otherwise.
    __device__ __forceinline__ bool WaveSubstringSearch(int c, const uint8_t* pp, int32_t pn, bool icase, uint64_t* out) {
        const int off = int(ColOff(c));
        const uint32_t endOff = ColEndOff(c);
        const uint64_t charsAddr = ReadLane64(tColChars, c);
        if (off < 0 || endOff == 0xFFFFFFFFu || !tileFull || pn < 1 || pn > 16 || (charsAddr & 15u) != 0 || U > 16) return false;
        const uint8_t* chars = reinterpret_cast<const uint8_t*>(charsAddr);
        const uint32_t limit = ReadLane32(tColCharsBytes, c) & ~15u;  // whole 16-byte pieces of the buffer
        const uint32_t* l = reinterpret_cast<const uint32_t*>(gLds + off) + waveBase;
        const uint32_t b0 = uint32_t(__builtin_amdgcn_readfirstlane(int(l[0])));
        const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int32_t*>(gLds + endOff)));
        const uint32_t lane = threadIdx.x & 63;
        uint64_t* words = reinterpret_cast<uint64_t*>(gLds + endOff + 16);
        if (lane < uint32_t(U)) words[lane] = 0;
        if (e < b0 || e > limit || limit < 16u) {  // offsets that do not describe this buffer (or its last, partial piece): the per-row path
            return false;
        }
        const uint32_t safe = e >= 16u ? (e - 16u) & ~15u : 0u;  // a whole piece inside the buffer
        // the pattern as two 64-bit words (zero behind its bytes)
        auto lower8 = [](uint64_t x) {  // ASCII upper-case bytes get bit 5
            const uint64_t t = x & 0x7F7F7F7F7F7F7F7Full;
            const uint64_t up = ((t + 0x3F3F3F3F3F3F3F3Full) & ~(t + 0x2525252525252525ull)) & ~x & 0x8080808080808080ull;
            return x | (up >> 2);
        };
        const uint32_t pat0 = uint32_t(pp - Pattern());
        uint64_t pw[2] = {PatternWord(pat0), pn > 8 ? PatternWord(pat0 + 8u) : 0ull};
        if (pn < 8) pw[0] &= (1ull << (8 * pn)) - 1ull;
        if (pn > 8 && pn < 16) pw[1] &= (1ull << (8 * (pn - 8))) - 1ull;
        if (icase) pw[0] = lower8(pw[0]), pw[1] = lower8(pw[1]);
        const uint32_t p32 = uint32_t(pw[0]);
        const uint32_t m32 = pn >= 4 ? ~0u : ((1u << (8 * pn)) - 1u);
        auto lower4 = [](uint32_t x) {  // ASCII upper-case bytes get bit 5
            const uint32_t t = x & 0x7F7F7F7Fu;
            const uint32_t up = ((t + 0x3F3F3F3Fu) & ~(t + 0x25252525u)) & ~x & 0x80808080u;
            return x | (up >> 2);
        };
        auto piece = [&](uint32_t at) {  // the lane's 16 bytes of the piece that starts at `at`
            uint4 v = make_uint4(0, 0, 0, 0);
            const uint32_t my = at + lane * 16u;
            // (through a pointer said to be global: a pointer made from an integer is of no known address space, its loads
            // are FLAT loads, which count as LDS operations too -- the compiler then waits for every load in flight at the
            // top of the loop, and the pieces meant to be on their way during the search are not)
            // (and without a branch around the load: behind a load that may or may not have been issued the compiler waits
            // for all of them just the same.  A lane past the slice reads the slice's last whole piece instead and searches
            // it like any other -- whatever it finds lies past the slice's end and is dropped with the hits there)
            {
                typedef uint32_t Words4 __attribute__((ext_vector_type(4)));
                const uint32_t from = my < e ? my : safe;
                const Words4 t = *reinterpret_cast<__attribute__((address_space(1))) const Words4*>(charsAddr + from);
                v = make_uint4(t.x, t.y, t.z, t.w);
            }
            if (icase) v = make_uint4(lower4(v.x), lower4(v.y), lower4(v.z), lower4(v.w));
            return v;
        };
        const uint32_t first = b0 & ~15u;
        // one piece searched: `cur` the lane's 16 bytes of the piece at `base`, `nxt` those of the piece behind it
        auto search = [&](const uint4& cur, const uint4& nxt, uint32_t base) {
            uint32_t d[8];
            d[0] = cur.x, d[1] = cur.y, d[2] = cur.z, d[3] = cur.w;
            // the 16 bytes behind mine: the next lane's, for the last lane the first lane's of the next piece
            // (a shift of the whole wave by one lane, a data-parallel move that leaves the last lane what it is given: the
            // next piece's first words, read as scalars -- no trip through the LDS crossbar)
            const uint32_t n0x = uint32_t(__builtin_amdgcn_readfirstlane(int(nxt.x))), n0y = uint32_t(__builtin_amdgcn_readfirstlane(int(nxt.y)));
            const uint32_t n0z = uint32_t(__builtin_amdgcn_readfirstlane(int(nxt.z))), n0w = uint32_t(__builtin_amdgcn_readfirstlane(int(nxt.w)));
            d[4] = uint32_t(__builtin_amdgcn_update_dpp(int(n0x), int(cur.x), 0x130, 0xf, 0xf, false));  // wave_shl:1
            d[5] = uint32_t(__builtin_amdgcn_update_dpp(int(n0y), int(cur.y), 0x130, 0xf, 0xf, false));
            d[6] = uint32_t(__builtin_amdgcn_update_dpp(int(n0z), int(cur.z), 0x130, 0xf, 0xf, false));
            d[7] = uint32_t(__builtin_amdgcn_update_dpp(int(n0w), int(cur.w), 0x130, 0xf, 0xf, false));
            // the pattern's first (up to) four bytes at each of the 16 start positions; the rest of a longer pattern is
            // compared for the few hits only
            uint32_t hits = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int q = j >> 2;
                const uint32_t window = __builtin_amdgcn_alignbyte(d[q + 1], d[q], uint32_t(j & 3));
                hits |= (((window ^ p32) & m32) == 0 ? 1u : 0u) << j;
            }
            const uint32_t my = base + lane * 16u;
            while (hits != 0) {
                const uint32_t pos = my + uint32_t(__ffs(int(hits)) - 1);
                hits &= hits - 1u;
                if (pos < b0 || pos + uint32_t(pn) > e) continue;
                bool rest = true;
                for (int32_t i = 4; i < pn && rest; ++i) {
                    const uint8_t ch = *reinterpret_cast<__attribute__((address_space(1))) const uint8_t*>(charsAddr + pos + uint32_t(i));
                    rest = (icase ? AsciiLower(ch) : ch) == uint8_t(pw[i >> 3] >> (8 * (i & 7)));
                }
                if (!rest) continue;
                int lo = 0, hi = 64 * U - 1;  // the last row that starts at or before pos (empty rows before it share its start)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (l[mid] <= pos) lo = mid; else hi = mid - 1;
                }
                const uint32_t rowEnd = lo + 1 < 64 * U ? l[lo + 1] : e;
                if (pos + uint32_t(pn) <= rowEnd) atomicOr(reinterpret_cast<unsigned long long*>(words + (lo >> 6)), 1ull << (lo & 63));
            }
        };
        // Three pieces go round in three sets of registers, so that two loads are on their way while a piece is searched
        // (one loop over `cur = nxt; nxt = ahead` moves registers, and a move waits for its load: nothing was in flight).
        uint4 pa = piece(first);
        uint4 pb = piece(first + 1024u);
        for (uint32_t base = first; base < e; base += 3072u) {
            uint4 pc = piece(base + 2048u);
            search(pa, pb, base);
            if (base + 1024u >= e) break;
            pa = piece(base + 3072u);
            search(pb, pc, base + 1024u);
            if (base + 2048u >= e) break;
            pb = piece(base + 4096u);
            search(pc, pa, base + 2048u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) out[u] = (words[u] >> lane) & 1ull;
        return true;
    }

    // Up to the first 8 bytes of a string (little-endian in the result, bytes past the string's end are zero).  From
    // the wave's head area when the bytes are there (three aligned LDS words and a byte shift, no branches on the
    // length); else one unaligned 8-byte load when that stays inside the column's chars buffer, byte loads next to its
    // end.
    __device__ __forceinline__ uint64_t StringHead(int c, const uint8_t* p, int32_t n) const {
        if (n <= 0) return 0;
        const uint8_t* chars = reinterpret_cast<const uint8_t*>(ReadLane64(tColChars, c));
        const int64_t off = p - chars;
        uint64_t w = 0;
        const uint32_t staged = ReadLane32(tCharsBytes, c);
        const int64_t rel = off - int64_t(ReadLane32(tCharsBegin, c));
        const int32_t m = n < 8 ? n : 8;
        if (staged != 0 && rel >= 0 && rel + m <= int64_t(staged)) {
            // (WaitStringHeads() has been called; the area has 16 spare bytes behind the staged ones)
            const uint32_t addr = ColCharsLds(c) + ReadLane32(tCharsSkew, c) + uint32_t(rel);
            const uint32_t* q = reinterpret_cast<const uint32_t*>(gLds + (addr & ~3u));
            const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
            const uint32_t sh = addr & 3u;
            w = uint64_t(__builtin_amdgcn_alignbyte(q1, q0, sh)) | (uint64_t(__builtin_amdgcn_alignbyte(q2, q1, sh)) << 32);
        } else {
            const int64_t limit = int64_t(ReadLane32(tColCharsBytes, c));
            if (off < 0 || off + m > limit) return 0;  // offsets that do not describe this buffer: never dereferenced
            if (off + 8 <= limit) {
                __builtin_memcpy(&w, p, 8);
            } else {
                for (int32_t j = 0; j < m; ++j) w |= uint64_t(p[j]) << (8 * j);
            }
        }
        return n >= 8 ? w : (w & ((1ull << (8 * n)) - 1ull));
    }

    // Group-by keys, the common case first: every string of the wave's slice has the same length L of 1..8 bytes (flags,
    // codes, fixed-width identifiers).  Then row i of the slice starts at begin + L * i and its bytes sit at that same
    // stride in the head area: for L = 1 a single byte read.  L is guessed without touching memory from the extent of the
    // staged heads (first row's start to last row's start + 8), so the reads that verify it -- the slice's end offset
    // and one 16-byte read of staged offsets per lane -- are issued together with the reads of the bytes: one LDS round
    // trip.  Returns L with the rows' bytes in w (little-endian, zero above L bytes), or 0 (w undefined) when the slice
    // is not like that (partial tile, heads not staged, lengths differ).  The decision is the same for the whole wave.
    __device__ __forceinline__ int UniformStringKeyWords(int c, uint64_t* w) const {
        constexpr uint32_t kRows = 64 * U;
        const int off = int(ColOff(c));
        const uint32_t endOff = ColEndOff(c);
        const uint32_t area = ColCharsLds(c);
        if (off < 0 || endOff == 0xFFFFFFFFu || area == 0xFFFFFFFFu || !tileFull) return 0;
        const uint32_t begin = ReadLane32(tCharsBegin, c), staged = ReadLane32(tCharsBytes, c);
        if (staged < 8u + (kRows - 1u)) return 0;
        const uint32_t L = (staged - 8u) / (kRows - 1u);
        if (L * (kRows - 1u) + 8u != staged || L > 8u) return 0;
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t base = area + ReadLane32(tCharsSkew, c);
        const uint32_t last = *reinterpret_cast<const uint32_t*>(gLds + endOff);
        // (any lane may check any offset: lane t takes offsets U * t .. U * t + U - 1 of the slice; the staged tiles start
        // at multiples of 16 bytes)
        const uint32_t* l = reinterpret_cast<const uint32_t*>(gLds + off) + waveBase + U * int(lane);
        uint32_t o[U];
        if constexpr (U == 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(l, 16));
            o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) o[u] = l[u];
        }
        if (L == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = gLds[base + uint32_t(u) * 64u + lane];
        } else if (L <= 4) {
            const uint32_t mask = L == 4 ? 0xFFFFFFFFu : ((1u << (8 * L)) - 1u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t addr = base + L * (uint32_t(u) * 64u + lane);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(gLds + (addr & ~3u));
                const uint32_t q0 = q[0], q1 = q[1];
                w[u] = __builtin_amdgcn_alignbyte(q1, q0, addr & 3u) & mask;
            }
        } else {
            const uint64_t mask = L == 8 ? ~0ull : ((1ull << (8 * L)) - 1ull);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t addr = base + L * (uint32_t(u) * 64u + lane);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(gLds + (addr & ~3u));
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
                const uint32_t sh = addr & 3u;
                w[u] = (uint64_t(__builtin_amdgcn_alignbyte(q1, q0, sh)) | (uint64_t(__builtin_amdgcn_alignbyte(q2, q1, sh)) << 32)) & mask;
            }
        }
        uint32_t bad = last ^ (begin + L * kRows);
#pragma unroll
        for (int u = 0; u < U; ++u) bad |= o[u] ^ (begin + L * (uint32_t(U) * lane + uint32_t(u)));
        return __all(bad == 0) ? int(L) : 0;
    }

    // Group-by keys: length and first 8 bytes (little-endian, zero past the end) of the strings of the lane's rows in
    // column c; rows outside `rows` (bit u) give length 0.  When the column's offsets and heads are staged everything
    // is 32-bit offset arithmetic on LDS data without a divergent branch -- two LDS reads for the offsets, three for
    // the bytes; strings whose bytes are not in the head area (a rare wave) are read from memory afterwards.
    __device__ __forceinline__ void StringKeyWords(int c, uint32_t rows, int32_t* n, uint64_t* w) const {
        const int off = int(ColOff(c));
        const uint32_t endOff = ColEndOff(c);
        const uint32_t area = ColCharsLds(c);
        rows &= act;
        if (off >= 0 && endOff != 0xFFFFFFFFu && area != 0xFFFFFFFFu) {
            const int32_t* l = reinterpret_cast<const int32_t*>(gLds + off);
            const int32_t last = *reinterpret_cast<const int32_t*>(gLds + endOff);
            const uint32_t begin = ReadLane32(tCharsBegin, c), staged = ReadLane32(tCharsBytes, c);
            const uint32_t base = area + ReadLane32(tCharsSkew, c);
            int32_t b[U];
            uint32_t missing = 0;  // bit u: the bytes of row u are not in the head area
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = Idx(u);
                const bool inSlice = idx + 1 < waveBase + 64 * U;
                b[u] = l[idx];
                const int32_t next = l[inSlice ? idx + 1 : idx];
                const int32_t e = (inSlice && Row(u) + 1 < nrows) ? next : last;
                n[u] = ((rows >> u) & 1u) ? e - b[u] : 0;
                if (n[u] < 0) n[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t m = uint32_t(n[u] < 8 ? n[u] : 8);
                const uint32_t rel = uint32_t(b[u]) - begin;  // wraps when b is below the staged range
                const bool here = rel < staged && rel + m <= staged;
                missing |= ((n[u] > 0 && !here) ? 1u : 0u) << u;
                const uint32_t addr = base + (here ? rel : 0u);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(gLds + (addr & ~3u));
                const uint32_t q0 = q[0], q1 = q[1], q2 = q[2];
                const uint32_t sh = addr & 3u;
                const uint64_t word = uint64_t(__builtin_amdgcn_alignbyte(q1, q0, sh)) | (uint64_t(__builtin_amdgcn_alignbyte(q2, q1, sh)) << 32);
                w[u] = n[u] >= 8 ? word : (word & ((1ull << (8 * n[u])) - 1ull));
            }
            if (__any(missing != 0)) {
                const uint8_t* chars = reinterpret_cast<const uint8_t*>(ReadLane64(tColChars, c));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if ((missing >> u) & 1u) w[u] = StringHead(c, chars + b[u], n[u]);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint8_t* p = nullptr;
            n[u] = 0;
            if ((rows >> u) & 1u) StringAt(c, u, p, n[u]);
            w[u] = StringHead(c, p, n[u]);
        }
    }

    // Operand fetch by pre-decoded kind: the host resolved column type, width and LDS offset, so a
    // staged column costs U LDS reads and no further dispatch.
    __device__ __forceinline__ void FetchFast(uint32_t fk, uint32_t w1, Val<U>& B) {
        const unsigned char* base = gLds + (w1 & 0xFFFFFFu);
        switch (fk & 0x7Fu) {
            case FK_NONE:
                return;
            case FK_LDS_64:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = reinterpret_cast<const uint64_t*>(base)[Idx(u)];
                break;
            case FK_LDS_I32:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = uint64_t(int64_t(reinterpret_cast<const int32_t*>(base)[Idx(u)]));
                break;
            case FK_LDS_U32:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = reinterpret_cast<const uint32_t*>(base)[Idx(u)];
                break;
            case FK_LDS_I16:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = uint64_t(int64_t(reinterpret_cast<const int16_t*>(base)[Idx(u)]));
                break;
            case FK_LDS_U16:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = reinterpret_cast<const uint16_t*>(base)[Idx(u)];
                break;
            case FK_LDS_I8:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = uint64_t(int64_t(reinterpret_cast<const int8_t*>(base)[Idx(u)]));
                break;
            case FK_LDS_U8:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = reinterpret_cast<const uint8_t*>(base)[Idx(u)];
                break;
            case FK_LDS_F32:
#pragma unroll
                for (int u = 0; u < U; ++u) B.v[u] = F64ToBits(double(reinterpret_cast<const float*>(base)[Idx(u)]));
                break;
            case FK_REG:
                FetchReg(int(w1 & 0xFFFFFFu), B);
                return;
            case FK_CONST:
                FetchConst(int(w1 & 0xFFFFFFu), B);
                return;
            default:  // FK_SLOW: unstaged / bit-packed / string column
                FetchCol(int(w1 >> 24), B);
                return;
        }
        B.ok = (fk & 0x80u) ? ValidityOf(int(w1 >> 24)) : AllMask<U>();
    }

    // Aggregate argument / key / output operand described by a pre-decoded (kind, payload) pair.
    __device__ __forceinline__ void FetchOperand(uint32_t fk, uint32_t w1, Val<U>& out) { FetchFast(fk, w1, out); }

    // cmpMask: bit 0 = lt, 1 = eq, 2 = gt, 3 = unordered contribute to the result.
    __device__ __forceinline__ uint32_t CompareByMask(int cls, uint32_t cmpMask, const uint64_t* x, const uint64_t* y) {
        uint32_t lt, eq, un;
        CompareMasks<U>(cls, x, y, lt, eq, un);
        if (cls == CLS_I || cls == CLS_U || cls == CLS_IU || cls == CLS_UI) {
            // integers are never unordered, and less / equal / greater part the rows: every combination is one of the two
            // masks or its complement (what the other is computed from is then dead code in a kernel built for the step)
            switch (cmpMask & 7u) {
                case 0: return 0;
                case 1: return lt & AllMask<U>();
                case 2: return eq & AllMask<U>();
                case 3: return (lt | eq) & AllMask<U>();
                case 4: return ~(lt | eq) & AllMask<U>();
                case 5: return ~eq & AllMask<U>();
                case 6: return ~lt & AllMask<U>();
                default: return AllMask<U>();
            }
        }
#ifdef SSA_SPEC
        if (cls == CLS_F64 || cls == CLS_F32) {
            // floating point, in a kernel built for the step (the mask is a constant there): one compare per row says
            // whether the pair's relation -- less, equal, greater or unordered -- is among those asked for
            uint32_t r = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double p = BitsToF64(x[u]), q = BitsToF64(y[u]);
                bool t;
                switch (cmpMask & 15u) {
                    case 0: t = false; break;
                    case 1: t = p < q; break;
                    case 2: t = p == q; break;
                    case 3: t = p <= q; break;
                    case 4: t = p > q; break;
                    case 5: t = p < q || p > q; break;
                    case 6: t = p >= q; break;
                    case 7: t = p == p && q == q; break;
                    case 8: t = p != p || q != q; break;
                    case 9: t = !(p >= q); break;
                    case 10: t = !(p < q || p > q); break;
                    case 11: t = !(p > q); break;
                    case 12: t = !(p <= q); break;
                    case 13: t = p != q; break;
                    case 14: t = !(p < q); break;
                    default: t = true; break;
                }
                r |= (t ? 1u : 0u) << u;
            }
            return r;
        }
#endif
        const uint32_t gt = ~(lt | eq | un);
        return ((lt & (0u - (cmpMask & 1u))) | (eq & (0u - ((cmpMask >> 1) & 1u))) | (gt & (0u - ((cmpMask >> 2) & 1u))) |
                (un & (0u - ((cmpMask >> 3) & 1u)))) & AllMask<U>();
    }

    // Runs the step program for the current tile.  Leaves outputs in R[] and the row predicate in P.
    __device__ __forceinline__ void Run() {
        P = act;
        A.ok = AllMask<U>();
#pragma unroll
        for (int u = 0; u < U; ++u) A.v[u] = 0;

        const int ncodeEff = (S().debug_flags & 2) ? 0 : NumCode();
        SSA_UNROLL_STATIC
        for (int pc = 0; pc < ncodeEff; ++pc) {
            const uint32_t w0 = Code0(pc);
            const uint32_t w1 = Code1(pc);
            const uint32_t ok = w0 & 0xFFu, fk = (w0 >> 8) & 0xFFu;
            const int fa = int((w0 >> 16) & 0xFFu), fb = int(w0 >> 24);
            Val<U> B;
            FetchFast(fk, w1, B);
            switch (ok) {
                case OK_LD:
                    A = B;
                    break;
                case OK_ST:
                    StoreReg(int(w1 & 0xFFFFFFu), A);
                    break;
                case OK_PREDCMP: {  // B = column, constant fa, class fb >> 4, mask fb & 15
                    uint64_t c[U];
                    const uint64_t cbits = ConstBits(fa);
#pragma unroll
                    for (int u = 0; u < U; ++u) c[u] = cbits;
#ifdef SSA_SPEC
                    // A staged column of at most 32 bits against a constant that fits them (the same for the whole launch):
                    // compared in 32 bits, half the instructions per row of the 64-bit payloads' compare.
                    {
                        const int cls = fb >> 4;
                        const bool narrowSigned = cls == CLS_I && (fk == FK_LDS_I32 || fk == FK_LDS_I16 || fk == FK_LDS_I8);
                        const bool narrowUnsigned = cls == CLS_U && (fk == FK_LDS_U32 || fk == FK_LDS_U16 || fk == FK_LDS_U8);
                        const bool fits = narrowSigned ? int64_t(cbits) == int64_t(int32_t(uint32_t(cbits))) : (cbits >> 32) == 0;
                        if ((narrowSigned || narrowUnsigned) && fits) {
                            const uint32_t c32 = uint32_t(cbits);
                            uint32_t lt = 0, eq = 0;
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const uint32_t x32 = uint32_t(B.v[u]);
                                lt |= ((narrowSigned ? int32_t(x32) < int32_t(c32) : x32 < c32) ? 1u : 0u) << u;
                                eq |= (x32 == c32 ? 1u : 0u) << u;
                            }
                            uint32_t r;
                            switch (uint32_t(fb) & 7u) {
                                case 0: r = 0; break;
                                case 1: r = lt; break;
                                case 2: r = eq; break;
                                case 3: r = lt | eq; break;
                                case 4: r = ~(lt | eq); break;
                                case 5: r = ~eq; break;
                                case 6: r = ~lt; break;
                                default: r = ~0u; break;
                            }
                            P &= r & AllMask<U>();
                            break;
                        }
                    }
#endif
                    P &= CompareByMask(fb >> 4, uint32_t(fb & 15), B.v, c);
                    break;
                }
                case OK_CMP: {
                    const uint32_t r = CompareByMask(fa, uint32_t(fb), A.v, B.v);
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = (r >> u) & 1u;
                    A.ok &= B.ok;
                    break;
                }
                case OK_ADD_F64:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = F64ToBits(BitsToF64(A.v[u]) + BitsToF64(B.v[u]));
                    A.ok &= B.ok;
                    break;
                case OK_SUB_F64:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = F64ToBits(BitsToF64(A.v[u]) - BitsToF64(B.v[u]));
                    A.ok &= B.ok;
                    break;
                case OK_RSUB_F64:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = F64ToBits(BitsToF64(B.v[u]) - BitsToF64(A.v[u]));
                    A.ok &= B.ok;
                    break;
                case OK_MUL_F64:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = F64ToBits(BitsToF64(A.v[u]) * BitsToF64(B.v[u]));
                    A.ok &= B.ok;
                    break;
                case OK_ADD_I:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = A.v[u] + B.v[u];
                    A.ok &= B.ok;
                    break;
                case OK_SUB_I:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = A.v[u] - B.v[u];
                    A.ok &= B.ok;
                    break;
                case OK_MUL_I:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = A.v[u] * B.v[u];
                    A.ok &= B.ok;
                    break;
                case OK_PRED: {
                    uint32_t t = 0;
#pragma unroll
                    for (int u = 0; u < U; ++u) t |= (A.v[u] != 0 ? 1u : 0u) << u;
                    P &= t;
                    break;
                }
                default: {
                    const uint32_t w2 = Code2(pc);
                    GenericOp(InsnOp(w2), InsnSi(w2), InsnA(w2), InsnB(w2), B);
                    break;
                }
            }
        }
    }

    // Every other instruction, in its generic (op, class, type) form.  B was fetched by the caller.
    __device__ __forceinline__ void GenericOp(int op, int si, int fa, int fb, Val<U>& B) {
    switch (op) {
        case OP_ADD: case OP_SUB: case OP_RSUB: case OP_MUL: {
            if (fa == CLS_F64) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double x = BitsToF64(A.v[u]), y = BitsToF64(B.v[u]);
                    const double r = op == OP_ADD ? x + y : op == OP_SUB ? x - y : op == OP_RSUB ? y - x : x * y;
                    A.v[u] = F64ToBits(r);
                }
            } else if (fa == CLS_F32) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float x = float(BitsToF64(A.v[u])), y = float(BitsToF64(B.v[u]));
                    const float r = op == OP_ADD ? x + y : op == OP_SUB ? x - y : op == OP_RSUB ? y - x : x * y;
                    A.v[u] = F64ToBits(double(r));
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t x = A.v[u], y = B.v[u];
                    A.v[u] = op == OP_ADD ? x + y : op == OP_SUB ? x - y : op == OP_RSUB ? y - x : x * y;
                }
            }
            A.ok &= B.ok;
            break;
        }
        case OP_DIV: case OP_RDIV: case OP_MOD: case OP_RMOD: {
            const bool rev = (op == OP_RDIV || op == OP_RMOD);
            const bool mod = (op == OP_MOD || op == OP_RMOD);
            const int dialect = fb & 1, rtype = fb >> 1;
            uint32_t ok = A.ok & B.ok;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t x = rev ? B.v[u] : A.v[u];
                const uint64_t y = rev ? A.v[u] : B.v[u];
                const bool live = (ok >> u) & 1u;
                if (dialect == DIALECT_ARROW && !live) {
                    A.v[u] = 0;  // the not-null kernel variant leaves 0 under a null slot and never divides there
                } else {
                    ArithResult r = ArithDivOutlined(mod, fa, dialect, rtype, x, y);
                    A.v[u] = r.v;
                    if (r.null) ok &= ~(1u << u);
                    if (r.error && ((act >> u) & 1u)) err |= SCAN_DIV_BY_ZERO;
                }
            }
            A.ok = ok;
            break;
        }
        case OP_FUNC: case OP_RFUNC: {
            const bool binary = IsBinaryFunc(fb);
            if (binary) A.ok &= B.ok;
            // the checked kernels visit valid slots only and leave 0 under the others
            const uint32_t visit = IsNotNullFunc(fb) ? (act & A.ok) : act;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t x = op == OP_RFUNC ? B.v[u] : A.v[u];
                const uint64_t y = op == OP_RFUNC ? A.v[u] : binary ? B.v[u] : 0;
                uint64_t v = 0;
                if (!IsNotNullFunc(fb) || ((visit >> u) & 1u)) {
                    const ArithResult r = ApplyFuncOutlined(fb, fa, x, y);
                    v = r.v;
                    if ((visit >> u) & 1u) err |= (r.error ? SCAN_DIV_BY_ZERO : 0u) | (r.overflow ? SCAN_OVERFLOW : 0u);
                }
                A.v[u] = v;
            }
            break;
        }
        case OP_SELECT: {
            Val<U> C;
            FetchReg(fa, C);
            uint32_t ok = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool p = A.v[u] != 0;
                A.v[u] = p ? B.v[u] : C.v[u];
                ok |= (((p ? B.ok : C.ok) >> u) & 1u) << u;
            }
            A.ok = ok;
            break;
        }
        case OP_LOOKUP: {
            const uint64_t* payload = a.tables[fa].payload;
            const uint8_t* valid = a.tables[fa].valid;
            const uint64_t n = a.tables[fa].n;
            uint32_t ok = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint64_t e = (((A.ok >> u) & 1u) && A.v[u] < n) ? A.v[u] : n;  // (a negative code is a huge unsigned one)
                A.v[u] = payload[e];
                ok |= uint32_t(valid[e] != 0) << u;
            }
            A.ok = ok;
            break;
        }
        case OP_CMP: {
            uint32_t lt, eq, un;
            CompareMasks<U>(fa, A.v, B.v, lt, eq, un);
            const uint32_t r = CompareResult(fb, lt, eq, un);
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = (r >> u) & 1u;
            A.ok &= B.ok;
            break;
        }
        case OP_AND: case OP_OR: case OP_XOR: {
            uint32_t av = 0, bv = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                av |= (A.v[u] != 0 ? 1u : 0u) << u;
                bv |= (B.v[u] != 0 ? 1u : 0u) << u;
            }
            uint32_t rv, rok = A.ok & B.ok;
            if (op == OP_AND) {
                rv = av & bv;
                // Kleene: a definite false on either side decides the result
                if (fb == DIALECT_YQL) rok |= (A.ok & ~av) | (B.ok & ~bv);
            } else if (op == OP_OR) {
                rv = av | bv;
                if (fb == DIALECT_YQL) rok |= (A.ok & av) | (B.ok & bv);
            } else {
                rv = av ^ bv;
            }
            // the value byte is the plain and/or/xor of the value bytes, also under null slots
            // (ydb/library/yql/minikql/comp_nodes/mkql_block_logical.cpp:190-192)
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = (rv >> u) & 1u;
            A.ok = rok;
            break;
        }
        case OP_NOT:
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = A.v[u] ? 0 : 1;
            break;
        case OP_ISNULL:
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = ((A.ok >> u) & 1u) ? 0 : 1;
            A.ok = AllMask<U>();
            break;
        case OP_ISVALID:
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = (A.ok >> u) & 1u;
            A.ok = AllMask<U>();
            break;
        case OP_WRAP:
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = WrapTo(A.v[u], fa);
            break;
        case OP_CAST:
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = CastValOutlined(A.v[u], fa, fb);
            break;
        case OP_CHECK:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (((act & A.ok) >> u) & 1u) {
                    if (!FitsTypeOutlined(A.v[u], fa, fb)) err |= SCAN_CAST_OVERFLOW;
                }
            }
            break;
        case OP_COALESCE: {
#pragma unroll
            for (int u = 0; u < U; ++u) A.v[u] = ((A.ok >> u) & 1u) ? A.v[u] : B.v[u];
            A.ok |= B.ok;
            break;
        }
        case OP_ROWIDX:
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint64_t r = uint64_t(Row(u));
                if (fa & ROWIDX_LOOKUP_ORDER) {
                    const uint64_t stream = int64_t(r) < (nrows & ~int64_t(3)) ? (r & 3ull) : 4ull;
                    r |= stream << kRowStreamShift;
                }
                if (fa & ROWIDX_NAN_TAG) {
                    const double d = BitsToF64(A.v[u]);
                    r = (r << 1) | (d != d ? 1ull : 0ull);
                }
                if (fa & ROWIDX_ZERO_TAG) r = BitsToF64(A.v[u]) == 0.0 ? ((r << 1) | (A.v[u] >> 63)) : ~0ull;
                A.v[u] = r;
            }
            break;
        case OP_DECPART: {
            // 128-bit sums are carried as three 64-bit sums of these parts (the 32-bit halves of the low word cannot
            // overflow a 64-bit accumulator below 2^32 rows)
            A.ok = ValidityOf(si);
            const uint64_t* words = reinterpret_cast<const uint64_t*>(ReadLane64(tColVal, si));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint64_t v = 0;
                if (((act & A.ok) >> u) & 1u) {
                    const uint64_t w = words[2 * Row(u) + (fa == 2 ? 1 : 0)];
                    v = fa == 0 ? (w & 0xFFFFFFFFull) : fa == 1 ? (w >> 32) : w;
                }
                A.v[u] = v;
            }
            break;
        }
        case OP_STRLEN: {
            A.ok = ValidityOf(si);
            const int32_t* offs = reinterpret_cast<const int32_t*>(ReadLane64(tColVal, si));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint64_t len = 0;
                if (((act & A.ok) >> u) & 1u) len = uint64_t(int64_t(offs[Row(u) + 1] - offs[Row(u)]));
                A.v[u] = len;  // 0 under null slots (not-null kernel variant)
            }
            break;
        }
        case OP_STRMATCH: {
            const uint64_t pat = ConstBits(fa);
            const uint8_t* pp = Pattern() + uint32_t(pat >> 32);
            const int32_t pn = int32_t(uint32_t(pat));
            A.ok = ValidityOf(si) & (ConstValid(fa) ? AllMask<U>() : 0u);
#ifdef SSA_SPEC
            // (only in kernels built for a program: inlined into the generic interpreter the search costs every program
            // some twenty registers -- Q6 on the generic kernel 3.5 -> 2.9 TB/s)
            if ((fb & 15) == STR_CONTAINS && WaveSubstringSearch(si, pp, pn, (fb & 16) != 0, A.v)) break;
#endif
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint64_t m = 0;
                if ((act >> u) & 1u) {
                    const uint8_t* s;
                    int32_t n;
                    StringAt(si, u, s, n);
                    m = StrMatchOutlined(fb & 15, (fb & 16) != 0, s, n, pp, pn) ? 1 : 0;
                }
                A.v[u] = m;
            }
            break;
        }
        case OP_STRCMP: {
            uint32_t ok = ValidityOf(si);
            const bool rhsCol = (fb & 16) != 0;
            const uint8_t* pp = nullptr;
            int32_t pn = 0;
            if (!rhsCol) {
                const uint64_t pat = ConstBits(fa);
                pp = Pattern() + uint32_t(pat >> 32);
                pn = int32_t(uint32_t(pat));
                ok &= ConstValid(fa) ? AllMask<U>() : 0u;
            } else {
                ok &= ValidityOf(fa);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint64_t m = 0;
                if ((act >> u) & 1u) {
                    const uint8_t* s;
                    int32_t n;
                    StringAt(si, u, s, n);
                    if (rhsCol) StringAt(fa, u, pp, pn);
                    m = CmpFromOrder(fb & 15, StrCompareOutlined(s, n, pp, pn)) ? 1 : 0;
                }
                A.v[u] = m;
            }
            A.ok = ok;
            break;
        }
        case OP_PRED: {
            uint32_t t = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) t |= (A.v[u] != 0 ? 1u : 0u) << u;
            P &= t;
            break;
        }
        default:
            break;
        }
    }

    __device__ __forceinline__ void ReportErrors() {
        if (offsBad != 0) err |= SCAN_NOT_UNIFORM;
        if (err != 0 && a.status != nullptr) atomicOr(a.status, err);
    }
};

// Kernel arguments are a {ScanArgs, Sink} pair passed by value.  The kernels index into them with a
// wave-uniform program counter, which the compiler can only turn into scalar loads if the struct is
// read in place from the kernel-argument segment (a by-value aggregate that is indexed dynamically
// would otherwise be copied to per-lane scratch memory).
template <class T>
__device__ __forceinline__ const T& KernArgsAs() {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const T*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return *static_cast<const T*>(nullptr);  // host pass: never executed
#endif
}

template <class Sink>
__device__ __forceinline__ const KArgs<Sink>& KernArgs() {
    return KernArgsAs<KArgs<Sink>>();
}

}  // namespace ssa
