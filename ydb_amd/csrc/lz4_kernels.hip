// LZ4 frames of compressed Arrow IPC bodies decoded on the device.
//
// The format is the published LZ4 frame / block format (lz4_Frame_format.md, lz4_Block_format.md; the reference
// vendors the library as contrib/libs/lz4 and reaches it through arrow::util::Codec, serializer/native.h:54-57).  Arrow
// writes every buffer as one frame of 64 KB blocks that may refer back into the previous block ("linked" blocks), so
// a frame decodes front to back.  One wavefront takes one frame: the compressed block is staged into LDS with wide
// loads, the token stream is parsed by the whole wave in lockstep (wave-uniform scalars), and the bytes of every
// literal run and every match are copied by the 64 lanes side by side.  The last 64 KB of output live in an LDS ring,
// which is where matches read from (LDS accesses of one wave complete in issue order, so a match sees the bytes
// earlier sequences wrote); every byte goes to the ring and to the output buffer in HBM, 64 consecutive bytes per
// store instruction.  Parallelism across frames comes from the grid: a blob has up to three buffers per column, a
// portion many blobs.  Bound: latency of the serial token chain, not HBM.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace ssa {

namespace {

constexpr uint32_t kRing = 65536;

__global__ __launch_bounds__(64) void Lz4DecodeKernel(const uint8_t* blob, const Lz4FrameTask* tasks, const Lz4BlockRef* blocks,
                                                     uint32_t inBytes, uint32_t* status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lz4Lds[];
    uint8_t* ring = lz4Lds;
    uint8_t* in = lz4Lds + kRing;
    const uint32_t lane = threadIdx.x;
    const Lz4FrameTask t = tasks[blockIdx.x];
    uint8_t* dst = t.dst;
    uint64_t op = 0;
    bool bad = false;
    for (uint32_t b = 0; b < t.nblocks && !bad; ++b) {
        const Lz4BlockRef blk = blocks[t.first_block + b];
        const uint8_t* src = blob + blk.Src;
        const uint32_t n = blk.SrcBytes;
        if (blk.Stored) {
            if (op + n > t.dst_bytes) {
                bad = true;
                break;
            }
            for (uint32_t i = lane; i < n; i += 64) {
                const uint8_t x = src[i];
                ring[(op + i) & (kRing - 1)] = x;
                dst[op + i] = x;
            }
            op += n;
            continue;
        }
        if (n + 32 > inBytes) {
            bad = true;
            break;
        }
        // stage the compressed block: whole 16-byte pieces from the aligned address below it (the blob's allocation is
        // 256-byte aligned and padded, so the pieces around the block are readable)
        const uint32_t skew = uint32_t(reinterpret_cast<uintptr_t>(src) & 15u);
        const uint4* a = reinterpret_cast<const uint4*>(src - skew);
        const uint32_t pieces = (skew + n + 15u) >> 4;
        for (uint32_t i = lane; i < pieces; i += 64) reinterpret_cast<uint4*>(in)[i] = a[i];
        const uint8_t* q = in + skew;
        uint32_t ip = 0;
        const uint64_t blockStart = op;
        while (ip < n) {
            // The common sequence -- at most five literals, a match of at most 18 bytes -- lies inside the eight bytes
            // that start at its token: one LDS round trip brings token, literals and offset, a second the match's source.
            // (The staged block is followed by 32 readable bytes.)  Everything else, and the end of a block, takes the
            // general code below.
            {
                uint64_t w = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) w |= uint64_t(q[ip + k]) << (8 * k);
                const uint32_t tk = uint32_t(w) & 0xFFu, fl = tk >> 4, fm = (tk & 15u) + 4u;
                if (fl <= 5u && (tk & 15u) != 15u && ip + 3u + fl <= n) {
                    const uint32_t off = uint32_t(w >> (8u * (1u + fl))) & 0xFFFFu;
                    const uint64_t mp = op + fl;  // where the match lands
                    const uint64_t reach = t.linked ? mp : mp - blockStart;
                    if (off == 0 || off > reach || mp + fm > t.dst_bytes) {
                        bad = true;
                        break;
                    }
                    if (lane < fl) {
                        const uint8_t x = uint8_t(w >> (8u * (1u + lane)));
                        ring[(op + lane) & (kRing - 1)] = x;
                        dst[op + lane] = x;
                    }
                    if (lane < fm) {
                        uint32_t r = lane;
                        if (off < fm) r = lane - off * uint32_t(__fdividef(float(lane) + 0.5f, float(off)));  // lane % off (lane < 19)
                        const uint8_t x = ring[(mp - off + r) & (kRing - 1)];
                        ring[(mp + lane) & (kRing - 1)] = x;
                        dst[mp + lane] = x;
                    }
                    ip += 3u + fl;
                    op = mp + fm;
                    continue;
                }
            }
            const uint32_t token = q[ip++];
            uint32_t lit = token >> 4;
            if (lit == 15) {
                uint32_t x;
                do {
                    if (ip >= n) {
                        bad = true;
                        break;
                    }
                    x = q[ip++];
                    lit += x;
                } while (x == 255);
            }
            if (bad || lit > n - ip || op + lit > t.dst_bytes) {
                bad = true;
                break;
            }
            for (uint32_t i = lane; i < lit; i += 64) {
                const uint8_t x = q[ip + i];
                ring[(op + i) & (kRing - 1)] = x;
                dst[op + i] = x;
            }
            ip += lit;
            op += lit;
            if (ip >= n) break;  // the last sequence of a block is literals only
            if (n - ip < 2) {
                bad = true;
                break;
            }
            const uint32_t off = uint32_t(q[ip]) | (uint32_t(q[ip + 1]) << 8);
            ip += 2;
            uint32_t ml = token & 15u;
            if (ml == 15) {
                uint32_t x;
                do {
                    if (ip >= n) {
                        bad = true;
                        break;
                    }
                    x = q[ip++];
                    ml += x;
                } while (x == 255);
            }
            ml += 4;
            const uint64_t reach = t.linked ? op : op - blockStart;  // how far back a match may point
            if (bad || off == 0 || off > reach || op + ml > t.dst_bytes) {
                bad = true;
                break;
            }
            if (off >= 64) {
                // 64 bytes at a time: the source of a piece was written by earlier pieces at the latest
                for (uint32_t base = 0; base < ml; base += 64) {
                    const uint32_t i = base + lane;
                    if (i < ml) {
                        const uint8_t x = ring[(op + i - off) & (kRing - 1)];
                        ring[(op + i) & (kRing - 1)] = x;
                        dst[op + i] = x;
                    }
                }
            } else {
                // the match overlaps its own output: a pattern of `off` bytes repeated
                for (uint32_t base = 0; base < ml; base += 64) {
                    const uint32_t i = base + lane;
                    if (i < ml) {
                        const uint8_t x = ring[(op - off + (i % off)) & (kRing - 1)];
                        ring[(op + i) & (kRing - 1)] = x;
                        dst[op + i] = x;
                    }
                }
            }
            op += ml;
        }
    }
    if ((bad || op != t.dst_bytes) && lane == 0) atomicOr(status, uint32_t(SCAN_BAD_BLOB));
}

}  // namespace

hipError_t LaunchLz4Decode(const uint8_t* blob, const Lz4FrameTask* tasks, int64_t ntasks, const Lz4BlockRef* blocks, uint32_t maxBlockBytes,
                           uint32_t* status, hipStream_t st) {
    if (ntasks <= 0) return hipSuccess;
    const uint32_t inBytes = (maxBlockBytes + 32u + 255u) & ~255u;
    const uint32_t lds = kRing + inBytes;
    static int configured = 0;  // bytes the attribute was last raised to (per process; the value only grows)
    if (int(lds) > configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&Lz4DecodeKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) return e;
        configured = 160 * 1024 - 1024;
    }
    if (lds > 160u * 1024u - 1024u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(Lz4DecodeKernel, dim3(uint32_t(ntasks)), dim3(64), lds, st, blob, tasks, blocks, inBytes, status);
    return hipGetLastError();
}

}  // namespace ssa
