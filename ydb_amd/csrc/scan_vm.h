// Device-side scan VM: evaluates the per-step expression program (vm.h) for a tile of rows.
//
// Work decomposition: a workgroup of kBlock = 256 lanes (4 wavefronts of 64) owns a tile of
// kBlock * U consecutive rows; lane t evaluates rows tile + u * kBlock + t (u = 0..U-1), so every
// column access of a wavefront is 64 consecutive elements -- fully coalesced at any element width.
// At the start of a tile every fixed-width source column is staged into LDS by direct global->LDS
// loads (no VGPRs, all columns in flight together, 16 bytes per lane); the interpreter then runs the
// step's instruction list reading operands from LDS.  The VM's R registers live in LDS as well, so
// the per-lane register state is just the accumulator -- occupancy is set by the LDS plan, and several
// workgroups per CU overlap one group's staging with another's evaluation.
// Instruction words, column descriptors and constants live in the kernel-argument segment and the
// program counter is wave-uniform, so decode is scalar work and every `switch` is a scalar branch.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "vm_ops.h"

namespace ssa {

template <int U>
struct Val {
    uint64_t v[U];
    uint32_t ok;  // bit u: row u is non-null
};

template <int U>
__device__ __forceinline__ constexpr uint32_t AllMask() {
    return (1u << U) - 1u;
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

template <int U>
struct Machine {
    static constexpr int kTile = kBlock * U;
    const ScanArgs& a;   // lives in the kernel-argument segment: uniform scalar loads
    int64_t row0;        // row of u = 0 for this lane
    uint32_t act;        // bit u: row0 + u * kBlock < nrows
    uint32_t P;          // bit u: row passes every Filter so far
    uint32_t err;        // ScanStatus bits raised by this lane
    bool staged;         // the current tile's columns sit in LDS (full tiles only)
    Val<U> A;

    __device__ __forceinline__ explicit Machine(const ScanArgs& args) : a(args), err(0), staged(false) {}

    __device__ __forceinline__ int64_t Row(int u) const { return row0 + int64_t(u) * kBlock; }

    // Stage the tile's fixed-width source columns into LDS with direct global->LDS loads
    // (global_load_lds_dwordx4: one wave-instruction moves 1 KiB, no VGPRs are involved), then
    // make them visible to the workgroup.  Callers must have passed a barrier since the last
    // read of the previous tile.  Partial tiles are read from global memory on demand instead.
    __device__ __forceinline__ void BeginTile(int64_t tileBase) {
        row0 = tileBase + threadIdx.x;
        act = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) act |= (Row(u) < a.nrows ? 1u : 0u) << u;
        staged = (tileBase + kTile <= a.nrows);
        __syncthreads();  // every wave is done reading the previous tile
        if (staged) {
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            const int lane = threadIdx.x & 63;
            int piece = 0;
            for (int c = 0; c < a.ncols; ++c) {
                const int off = a.lds_col_off[c];
                if (off < 0) continue;
                const int w = TypeWidth(a.cols[c].type);
                const unsigned char* src = static_cast<const unsigned char*>(a.cols[c].values) + tileBase * w;
                const int npieces = (kTile * w) >> 10;
                for (int p = 0; p < npieces; ++p, ++piece) {
                    if ((piece & (kBlock / 64 - 1)) == wave) {
                        __builtin_amdgcn_global_load_lds((GlobalPtr)(src + p * 1024 + lane * 16),
                                                         (LdsPtr)(gLds + off + p * 1024), 16, 0, 0);
                    }
                }
            }
        }
        __syncthreads();  // waits for the LDS-DMA (vmcnt(0)) before the barrier
    }

    __device__ __forceinline__ void FetchCol(int c, Val<U>& out) {
        const DevColumn& col = a.cols[c];
        const int type = col.type;
        const int w = TypeWidth(type);
        if (w > 0 && w <= 8) {
            const int off = a.lds_col_off[c];
            if (staged && off >= 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) out.v[u] = LdsLoadRaw(gLds + off, w, u * kBlock + int(threadIdx.x));
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) out.v[u] = ((act >> u) & 1u) ? LoadRaw(col.values, w, Row(u)) : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) out.v[u] = ConvertRaw(out.v[u], type);
        } else if (type == T_BOOL) {
            const uint8_t* bits = static_cast<const uint8_t*>(col.values);
#pragma unroll
            for (int u = 0; u < U; ++u) out.v[u] = ((act >> u) & 1u) ? LoadBit(bits, Row(u) + col.val_bit_off) : 0;
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) out.v[u] = 0;  // strings / wide values carry only validity here
        }
        out.ok = ColValidity(col);
    }

    // R registers live in LDS: lane t owns elements [u * kBlock + t] of every register, so no
    // barrier is needed between a store and a later fetch.
    __device__ __forceinline__ void FetchReg(int r, Val<U>& out) {
        const uint64_t* vals = reinterpret_cast<const uint64_t*>(gLds + a.lds_reg_off) + size_t(r) * kTile;
#pragma unroll
        for (int u = 0; u < U; ++u) out.v[u] = vals[u * kBlock + threadIdx.x];
        out.ok = (gLds + a.lds_regok_off)[r * kBlock + threadIdx.x];
    }

    __device__ __forceinline__ void StoreReg(int r, const Val<U>& in) {
        uint64_t* vals = reinterpret_cast<uint64_t*>(gLds + a.lds_reg_off) + size_t(r) * kTile;
#pragma unroll
        for (int u = 0; u < U; ++u) vals[u * kBlock + threadIdx.x] = in.v[u];
        (gLds + a.lds_regok_off)[r * kBlock + threadIdx.x] = uint8_t(in.ok);
    }

    __device__ __forceinline__ void FetchConst(int k, Val<U>& out) {
        const uint64_t bits = a.consts[k].bits;
#pragma unroll
        for (int u = 0; u < U; ++u) out.v[u] = bits;
        out.ok = a.consts[k].valid ? AllMask<U>() : 0u;
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

    // Aggregate argument / key / output operand: register (>= 0) or column slot (<= -2).
    __device__ __forceinline__ void FetchOperand(int reg, Val<U>& out) {
        if (reg >= 0) {
            FetchReg(reg, out);
        } else {
            FetchCol(-2 - reg, out);
        }
    }

    __device__ __forceinline__ void StringAt(const DevColumn& col, int64_t row, const uint8_t*& p, int32_t& n) const {
        const int32_t* offs = static_cast<const int32_t*>(col.values);
        const int32_t b = offs[row], e = offs[row + 1];
        p = col.chars + b;
        n = e - b;
    }

    __device__ __forceinline__ uint32_t ColValidity(const DevColumn& col) const {
        if (col.validity == nullptr) return AllMask<U>();
        uint32_t ok = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((act >> u) & 1u) ok |= LoadBit(col.validity, Row(u) + col.valid_bit_off) << u;
        }
        return ok;
    }

    // Runs the step program for the current tile.  Leaves outputs in R[] and the row predicate in P.
    __device__ __forceinline__ void Run() {
        P = act;
        A.ok = AllMask<U>();
#pragma unroll
        for (int u = 0; u < U; ++u) A.v[u] = 0;

        for (int pc = 0; pc < a.ncode; ++pc) {
            const uint32_t ins = a.code[pc];
            const int op = InsnOp(ins);
            if (op == OP_END) break;
            const int sk = InsnSk(ins), si = InsnSi(ins), fa = InsnA(ins), fb = InsnB(ins);
            switch (op) {
                case OP_LD:
                    Fetch(sk, si, A);
                    break;
                case OP_ST:
                    StoreReg(si, A);
                    break;
                case OP_ADD: case OP_SUB: case OP_RSUB: case OP_MUL: {
                    Val<U> B;
                    Fetch(sk, si, B);
                    const int bop = (op == OP_RSUB) ? OP_SUB : op;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint64_t x = (op == OP_RSUB) ? B.v[u] : A.v[u];
                        const uint64_t y = (op == OP_RSUB) ? A.v[u] : B.v[u];
                        A.v[u] = ArithBasic(bop, fa, x, y);
                    }
                    A.ok &= B.ok;
                    break;
                }
                case OP_DIV: case OP_RDIV: case OP_MOD: case OP_RMOD: {
                    Val<U> B;
                    Fetch(sk, si, B);
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
                            ArithResult r = ArithDiv(mod, fa, dialect, rtype, x, y);
                            A.v[u] = r.v;
                            if (r.null) ok &= ~(1u << u);
                            if (r.error && ((act >> u) & 1u)) err |= SCAN_DIV_BY_ZERO;
                        }
                    }
                    A.ok = ok;
                    break;
                }
                case OP_CMP: {
                    Val<U> B;
                    Fetch(sk, si, B);
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = CompareVals(fb, fa, A.v[u], B.v[u]) ? 1 : 0;
                    A.ok &= B.ok;
                    break;
                }
                case OP_AND: case OP_OR: case OP_XOR: {
                    Val<U> B;
                    Fetch(sk, si, B);
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
                    for (int u = 0; u < U; ++u) A.v[u] = CastVal(A.v[u], fa, fb);
                    break;
                case OP_CHECK:
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (((act & A.ok) >> u) & 1u) {
                            if (!FitsType(A.v[u], fa, fb)) err |= SCAN_CAST_OVERFLOW;
                        }
                    }
                    break;
                case OP_COALESCE: {
                    Val<U> B;
                    Fetch(sk, si, B);
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = ((A.ok >> u) & 1u) ? A.v[u] : B.v[u];
                    A.ok |= B.ok;
                    break;
                }
                case OP_ROWIDX:
#pragma unroll
                    for (int u = 0; u < U; ++u) A.v[u] = uint64_t(Row(u));
                    break;
                case OP_STRLEN: {
                    const DevColumn& col = a.cols[si];
                    A.ok = ColValidity(col);
                    const int32_t* offs = static_cast<const int32_t*>(col.values);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        uint64_t len = 0;
                        if (((act & A.ok) >> u) & 1u) len = uint64_t(int64_t(offs[Row(u) + 1] - offs[Row(u)]));
                        A.v[u] = len;  // 0 under null slots (not-null kernel variant)
                    }
                    break;
                }
                case OP_STRMATCH: {
                    const DevColumn& col = a.cols[si];
                    const uint64_t pat = a.consts[fa].bits;
                    const uint8_t* pp = a.pattern + uint32_t(pat >> 32);
                    const int32_t pn = int32_t(uint32_t(pat));
                    A.ok = ColValidity(col) & (a.consts[fa].valid ? AllMask<U>() : 0u);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        uint64_t m = 0;
                        if ((act >> u) & 1u) {
                            const uint8_t* s;
                            int32_t n;
                            StringAt(col, Row(u), s, n);
                            m = StrMatch(fb & 15, (fb & 16) != 0, s, n, pp, pn) ? 1 : 0;
                        }
                        A.v[u] = m;
                    }
                    break;
                }
                case OP_STRCMP: {
                    const DevColumn& col = a.cols[si];
                    uint32_t ok = ColValidity(col);
                    const bool rhsCol = (fb & 16) != 0;
                    const uint8_t* pp = nullptr;
                    int32_t pn = 0;
                    if (!rhsCol) {
                        const uint64_t pat = a.consts[fa].bits;
                        pp = a.pattern + uint32_t(pat >> 32);
                        pn = int32_t(uint32_t(pat));
                        ok &= a.consts[fa].valid ? AllMask<U>() : 0u;
                    } else {
                        ok &= ColValidity(a.cols[fa]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        uint64_t m = 0;
                        if ((act >> u) & 1u) {
                            const uint8_t* s;
                            int32_t n;
                            StringAt(col, Row(u), s, n);
                            if (rhsCol) StringAt(a.cols[fa], Row(u), pp, pn);
                            m = CmpFromOrder(fb & 15, StrCompare(s, n, pp, pn)) ? 1 : 0;
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
    }

    __device__ __forceinline__ void ReportErrors() {
        if (err != 0 && a.status != nullptr) atomicOr(a.status, err);
    }
};

// Kernel arguments are a {ScanArgs, Sink} pair passed by value.  The kernels index into them with a
// wave-uniform program counter, which the compiler can only turn into scalar loads if the struct is
// read in place from the kernel-argument segment (a by-value aggregate that is indexed dynamically
// would otherwise be copied to per-lane scratch memory).
template <class Sink>
struct KArgs {
    ScanArgs a;
    Sink s;
};

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
