// LZ4 frames of compressed Arrow IPC bodies decoded on the device.
//
// The format is the published LZ4 frame / block format (lz4_Frame_format.md, lz4_Block_format.md; the reference
// vendors the library as contrib/libs/lz4 and reaches it through arrow::util::Codec, serializer/native.h:54-57).  Arrow
// writes every buffer as one frame of 64 KB blocks that may refer back into the previous block ("linked" blocks), so
// the copies of a frame go front to back; the parse of a block needs nothing but the block.  Two ways of decoding follow.
// Lz4DecodeKernel: one wavefront takes one frame and does both -- the token stream is parsed by the whole wave in
// lockstep out of registers (wave-uniform scalars), and the bytes of every literal run and every match are copied by
// the 64 lanes side by side.  The last 64 KB of output live in an LDS ring, which is where matches read from (LDS
// accesses of one wave complete in issue order, so a match sees the bytes earlier sequences wrote); every byte goes to
// the ring and to the output buffer in HBM, 64 consecutive bytes per store instruction.
// Lz4ParseKernel + Lz4BatchCopyKernel (the default, given room for the records): every block is parsed into one record
// per sequence, a block per lane; then a workgroup per frame does the copies, a batch of up to 1024 records at a time, with
// nothing serial inside a batch (the bytes' sources are resolved by pointer jumping) and the frame's last 64 KB in LDS.
// Parallelism across frames comes from the grid: a blob has up to three buffers per column, a portion many blobs.
// Bound: per frame, the chain of batches (a dozen barriers each); across frames, instruction issue -- not HBM.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace ssa {

namespace {

constexpr uint32_t kRing = 65536;

// One wave, one instruction at a time: the decode is bound by the number of instructions a sequence takes (there is no
// second wave on the SIMD to fill the gaps), so the common sequence is kept short and free of branches.
__global__ __launch_bounds__(64) void Lz4DecodeKernel(const uint8_t* blob, const Lz4FrameTask* tasks, const Lz4BlockRef* blocks, uint32_t* status) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing];
    const uint32_t lane = threadIdx.x;
    const float laneF = float(lane);
    const Lz4FrameTask t = tasks[blockIdx.x];
    // (a pointer read from memory is of no known address space: said to be global, the stores below do not count as LDS
    // operations, which the waits for the ring would wait for too)
    __attribute__((address_space(1))) uint8_t* dst = (__attribute__((address_space(1))) uint8_t*)t.dst;
    const uint32_t total = uint32_t(t.dst_bytes);  // (the host hands over frames of less than 4 GB)
    uint32_t op = 0;
    bool bad = false;
    for (uint32_t b = 0; b < t.nblocks && !bad; ++b) {
        const Lz4BlockRef blk = blocks[t.first_block + b];
        const uint8_t* src = blob + blk.Src;
        const uint32_t n = blk.SrcBytes;
        if (n == 0) continue;
        if (blk.Stored) {
            if (n > total - op) {
                bad = true;
                break;
            }
            for (uint32_t base = 0; base < n; base += 1024) {  // (sixteen loads in flight per lane: the copy is bound by their latency)
                uint8_t x[16];
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    const uint32_t i = base + k * 64 + lane;
                    x[k] = i < n ? src[i] : uint8_t(0);
                }
#pragma unroll
                for (uint32_t k = 0; k < 16; ++k) {
                    const uint32_t i = base + k * 64 + lane;
                    if (i < n) {
                        ring[(op + i) & (kRing - 1)] = x[k];
                        dst[op + i] = x[k];
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // (see rebase below)
            op += n;
            continue;
        }
        // The compressed bytes are parsed out of registers.  Register A holds 256 bytes of the block, one 32-bit word
        // per lane, counted from the aligned word below the block's first byte; B holds the 256 bytes behind them.  A
        // word comes out of A by lane number (v_readlane: the position is the same for the whole wave), so token, lengths
        // and offset of a sequence are scalar arithmetic with no trip to memory.  When the parse nears A's end, A moves
        // up to the word the parse stands at (lanes shifted through A and B) and B is loaded anew: those bytes are needed
        // 240 bytes of input later.  Words past the block's last are not read.
        const uint32_t skew = uint32_t(reinterpret_cast<uintptr_t>(src) & 3u);
        const uint32_t* words = reinterpret_cast<const uint32_t*>(src - skew);
        const uint32_t lastWord = (skew + n - 1u) >> 2;
        auto load = [&](uint32_t first) -> uint32_t {
            const uint32_t i = first + lane;
            return words[i < lastWord ? i : lastWord];
        };
        uint32_t wbase = 0;  // the word in A's lane 0
        uint32_t A = load(0), B = load(64);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // (as in rebase below)
        auto rebase = [&](uint32_t d) {
            const uint32_t s = d - wbase;
            if (s < 64u) {
                const uint32_t j = (lane + s) & 63u;
                const uint32_t a = uint32_t(__shfl(int(A), int(j))), c = uint32_t(__shfl(int(B), int(j)));
                A = lane + s < 64u ? a : c;
            } else {  // (a long run of literals was stepped over)
                A = load(d);
                // waited for here, so that the common path, which reads A right away, need not wait for whatever loads
                // and stores are in flight: vmcnt 0, the other counters left alone
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            B = load(d + 64u);
            wbase = d;
        };
        // twelve bytes of the block from byte `at` on
        auto fetch = [&](uint32_t at, uint64_t& w0, uint32_t& w1) {
            const uint32_t y = skew + at;
            uint32_t r = (y >> 2) - wbase;
            if (r > 60u) {
                rebase(y >> 2);
                r = 0;
            }
            const uint32_t x0 = uint32_t(__builtin_amdgcn_readlane(int(A), int(r)));
            const uint32_t x1 = uint32_t(__builtin_amdgcn_readlane(int(A), int(r + 1u)));
            const uint32_t x2 = uint32_t(__builtin_amdgcn_readlane(int(A), int(r + 2u)));
            const uint32_t x3 = uint32_t(__builtin_amdgcn_readlane(int(A), int(r + 3u)));
            const uint32_t sh = 8u * (y & 3u);
            const uint32_t b0 = uint32_t((uint64_t(x0) | (uint64_t(x1) << 32)) >> sh);
            const uint32_t b1 = uint32_t((uint64_t(x1) | (uint64_t(x2) << 32)) >> sh);
            w1 = uint32_t((uint64_t(x2) | (uint64_t(x3) << 32)) >> sh);
            w0 = uint64_t(b0) | (uint64_t(b1) << 32);
        };
        uint32_t ip = 0;
        // the bytes of a length that follow a token's 15: adds them to `len`, stops behind the first that is not 255
        auto readLength = [&](uint32_t& len) {
            for (;;) {
                if (ip >= n) {
                    bad = true;
                    return;
                }
                uint64_t a;
                uint32_t unused;
                fetch(ip, a, unused);
                for (int k = 0; k < 8; ++k) {
                    if (ip >= n) {
                        bad = true;
                        return;
                    }
                    const uint32_t x = uint32_t(a) & 0xFFu;
                    a >>= 8;
                    ++ip;
                    len += x;
                    if (x != 255u) return;
                }
            }
        };
        const uint32_t blockStart = op;
        while (ip < n) {
            uint64_t w0;
            uint32_t w1;
            fetch(ip, w0, w1);
            const uint32_t token = uint32_t(w0) & 0xFFu;
            // The common sequence -- at most nine literals, a match of at most 18 bytes -- lies inside the twelve bytes
            // that start at its token.  Its bytes, literals and match alike, are written by one lane each in one step:
            // a literal comes out of the scalar registers, a match byte out of the ring -- or out of the scalar registers
            // too, where the match reaches back into the literals of its own sequence.
            // Sequences without literals (a token below 15: three bytes, columns of few distinct values are made of
            // them): up to four come out of one fetch.
            if (__builtin_expect(token < 15u, 0) && ip + 3u <= n) {
                uint32_t avail = 12u;
                do {
                    const uint32_t fm = (uint32_t(w0) & 15u) + 4u;
                    const uint32_t off = uint32_t(w0 >> 8) & 0xFFFFu;
                    const uint32_t reach = t.linked ? op : op - blockStart;
                    if (off == 0 || off > reach || fm > total - op) {
                        bad = true;
                        break;
                    }
                    uint32_t r = lane;
                    if (off < fm) r = lane - __umul24(uint32_t((laneF + 0.5f) * __builtin_amdgcn_rcpf(float(off))), off);  // lane mod off (lane < 18)
                    if (lane < fm) {
                        const uint8_t x = ring[(op - off + r) & (kRing - 1)];
                        ring[(op + lane) & (kRing - 1)] = x;
                        dst[op + lane] = x;
                    }
                    ip += 3u;
                    op += fm;
                    w0 = (w0 >> 24) | (uint64_t(w1) << 40);
                    w1 >>= 24;
                    avail -= 3u;
                } while (avail >= 3u && (uint32_t(w0) & 0xFFu) < 15u && ip + 3u <= n);
                if (bad) break;
                continue;
            }
            {
                const uint32_t fl = token >> 4, fm = (token & 15u) + 4u;
                if (fl <= 9u && (token & 15u) != 15u && ip + 3u + fl <= n) {
                    const uint64_t mid = (w0 >> 32) | (uint64_t(w1) << 32);  // bytes 4..11
                    const uint32_t p = 8u * (1u + fl);
                    const uint32_t off = uint32_t(p <= 48u ? w0 >> p : mid >> (p - 32u)) & 0xFFFFu;
                    const uint32_t mp = op + fl;  // where the match lands
                    const uint32_t reach = t.linked ? mp : mp - blockStart;
                    if (off == 0 || off > reach || fl + fm > total - op) {
                        bad = true;
                        break;
                    }
                    // match byte m comes from `off` bytes back; a match that overlaps its own output repeats a pattern of
                    // `off` bytes: m mod off (m < 18: the rounded quotient is exact)
                    const uint32_t m = lane - fl;
                    uint32_t r = m;
                    if (off < fm) r = m - __umul24(uint32_t((laneF - float(fl) + 0.5f) * __builtin_amdgcn_rcpf(float(off))), off);
                    const uint32_t back = off - r;                 // bytes before the match's start
                    const bool literal = lane < fl || back <= fl;  // (for a lane of the match: its source is a literal)
                    const uint32_t li = lane < fl ? lane : fl - back;
                    const uint32_t k = 8u * (li + 1u);
                    const uint8_t fromWords = uint8_t((k <= 56u ? w0 : mid) >> (k <= 56u ? k : k - 32u));
                    const uint8_t fromRing = ring[(mp - back) & (kRing - 1)];
                    if (lane < fl + fm) {
                        const uint8_t x = literal ? fromWords : fromRing;
                        ring[(op + lane) & (kRing - 1)] = x;
                        dst[op + lane] = x;
                    }
                    ip += 3u + fl;
                    op = mp + fm;
                    continue;
                }
            }
            ++ip;
            uint32_t lit = token >> 4;
            if (lit == 15u) readLength(lit);
            if (bad || lit > n - ip || lit > total - op) {
                bad = true;
                break;
            }
            for (uint32_t base = 0; base < lit; base += 64) {
                const uint32_t m = lit - base < 64u ? lit - base : 64u;
                const uint32_t y0 = skew + ip + base;
                uint8_t x;
                if (((y0 + m - 1u) >> 2) - wbase < 128u) {  // the piece lies in A and B
                    const uint32_t yb = y0 + lane;
                    const uint32_t wi = (yb >> 2) - wbase;
                    const uint32_t a = uint32_t(__shfl(int(A), int(wi & 63u))), c = uint32_t(__shfl(int(B), int(wi & 63u)));
                    x = uint8_t((wi < 64u ? a : c) >> (8u * (yb & 3u)));
                } else {
                    x = lane < m ? src[ip + base + lane] : uint8_t(0);
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // (here, not at the top of the loop: see rebase)
                }
                if (lane < m) {
                    ring[(op + base + lane) & (kRing - 1)] = x;
                    dst[op + base + lane] = x;
                }
            }
            ip += lit;
            op += lit;
            if (ip >= n) break;  // the last sequence of a block is literals only
            if (n - ip < 2) {
                bad = true;
                break;
            }
            fetch(ip, w0, w1);
            const uint32_t off = uint32_t(w0) & 0xFFFFu;
            ip += 2;
            uint32_t ml = token & 15u;
            if (ml == 15u) readLength(ml);
            ml += 4;
            const uint32_t reach = t.linked ? op : op - blockStart;  // how far back a match may point
            if (bad || off == 0 || off > reach || ml > total - op) {
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
                // the match overlaps its own output: a pattern of `off` bytes repeated.  Its first bytes come from the
                // pattern itself; later ones from the match's own output a whole number of periods back, written by an
                // earlier piece (a match longer than the ring has overwritten the pattern's own place by then)
                const uint32_t periods = ((64u + off - 1u) / off) * off;
                for (uint32_t base = 0; base < ml; base += 64) {
                    const uint32_t i = base + lane;
                    if (i < ml) {
                        const uint8_t x = i < periods ? ring[(op - off + (i % off)) & (kRing - 1)] : ring[(op + i - periods) & (kRing - 1)];
                        ring[(op + i) & (kRing - 1)] = x;
                        dst[op + i] = x;
                    }
                }
            }
            op += ml;
        }
    }
    if ((bad || op != total) && lane == 0) atomicOr(status, uint32_t(SCAN_BAD_BLOB));
}


// ---- the same decode split in two: the parse of all blocks side by side, then the copies frame by frame -------------
// The parse of a block needs nothing but the block (tokens, lengths, offsets); only the copies chain through the frame
// (a match may read what the previous block wrote).  Lz4ParseKernel turns every block into one record per sequence;
// Lz4BatchCopyKernel is left with the copies.

constexpr uint32_t kSeqExternal = 0x40000000u;  // Lz4Seq::lit: the literals are not in the record (lo = where they start in the block)
constexpr uint32_t kSeqCount = 0x00FFFFFFu;

// A block per lane: the token arithmetic is a chain of dependent integer operations with nothing to share between lanes,
// so sixty-four blocks advance with every instruction (one block per wave left the parse bound by the one scalar unit of
// a compute unit).  A lane reads its block sixteen bytes at a time from wherever it stands -- its cache lines are its own
// for a dozen sequences -- and writes its records one by one.
struct __attribute__((packed, aligned(4))) Lz4Window {
    uint32_t w[4];
};

__global__ __launch_bounds__(64) void Lz4ParseKernel(const uint8_t* blob, const Lz4BlockRef* blocks, uint32_t nblocks, const uint64_t* recStart, Lz4Seq* recs,
                                                    Lz4BlockSummary* summaries, uint32_t* status) {
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (g >= nblocks) return;
    const Lz4BlockRef blk = blocks[g];
    const uint8_t* src = blob + blk.Src;
    const uint32_t n = blk.SrcBytes;
    if (blk.Stored || n == 0) {
        summaries[g] = Lz4BlockSummary{0, 0, 0, 0};
        return;
    }
    uint4* out = reinterpret_cast<uint4*>(recs + recStart[g]);
    const uint32_t cap = n / 3u + 2u;  // (a sequence with a match takes three bytes at least, the last one at least one)
    uint32_t k = 0;
    uint32_t produced = 0;     // bytes the block's sequences so far decode to
    uint32_t reachesBack = 0;  // how far before the block's first byte its matches read (0: the block stands alone)
    bool bad = false;
    // A lane reads its block through a window of 256 bytes in LDS (64 words, word i of every lane side by side).  The
    // lanes stand at sixty-four different places in sixty-four different cache lines: read straight from memory, some lane
    // misses its line at every sequence and the wave waits for memory every time.  Here the wave goes to memory for all
    // of its lanes at once -- when the first of them nears the end of its window every lane drops what it has consumed
    // and loads as much behind it -- and reads LDS in between.  (The blob has padding behind it: bytes past a block are
    // loaded, never used.)
    __shared__ uint32_t win[64 * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t skew = uint32_t(reinterpret_cast<uintptr_t>(src) & 3u);
    const Lz4Window* const groups = reinterpret_cast<const Lz4Window*>(src - skew);  // the block as groups of four aligned words
    uint32_t wd = 0;  // the window holds the words [wd, wd + 64) of the block, counted from the aligned word below its first byte; a multiple of 4
    auto refill = [&](uint32_t newWd, uint32_t loadFrom) {
        const uint32_t count = (newWd + 64u - loadFrom) >> 2;  // groups to load: 0 .. 16
#pragma unroll
        for (uint32_t half = 0; half < 2; ++half) {
            if (half == 1 && !__any(count > 8u)) break;
            Lz4Window t[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t q = half * 8u + j;
                if (q < count) t[j] = groups[(loadFrom >> 2) + q];
            }
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t q = half * 8u + j;
                if (q < count) {
                    const uint32_t w = loadFrom + 4u * q;
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) win[((w + i) & 63u) * 64u + lane] = t[j].w[i];
                }
            }
        }
        wd = newWd;
    };
    refill(0, 0);
    // twelve bytes of the block from byte `at` on
    auto fetch = [&](uint32_t at, uint64_t& w0, uint32_t& w1) {
        const uint32_t y = skew + at;
        const uint32_t d = y >> 2;
        // (every lane that is still parsing takes part whenever one of them has to)
        if (__any(d + 4u > wd + 64u)) {
            const uint32_t newWd = d & ~3u;
            refill(newWd, newWd > wd + 64u ? newWd : wd + 64u);
        }
        uint32_t x[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) x[i] = win[((d + i) & 63u) * 64u + lane];
        const uint32_t sh = 8u * (y & 3u);
        const uint32_t b0 = uint32_t((uint64_t(x[0]) | (uint64_t(x[1]) << 32)) >> sh);
        const uint32_t b1 = uint32_t((uint64_t(x[1]) | (uint64_t(x[2]) << 32)) >> sh);
        w1 = uint32_t((uint64_t(x[2]) | (uint64_t(x[3]) << 32)) >> sh);
        w0 = uint64_t(b0) | (uint64_t(b1) << 32);
    };
    auto emit = [&](uint32_t lit, uint32_t match, uint32_t lo, uint32_t hi) {
        if (k >= cap) {
            bad = true;
            return;
        }
        out[k] = uint4{lit, match, lo, hi};
        ++k;
    };
    // every match: its offset is checked here, against the bytes the block itself has produced, so that the copy kernel
    // checks a block once (its size, and how far it reaches back) instead of every sequence
    auto account = [&](uint32_t lit, uint32_t ml, uint32_t off) {
        if (lit > 0x7FFFFFFFu - produced) {
            bad = true;
            return;
        }
        produced += lit;
        if (ml != 0) {
            if (off == 0 || ml > 0x7FFFFFFFu - produced) {
                bad = true;
                return;
            }
            if (off > produced && off - produced > reachesBack) reachesBack = off - produced;
            produced += ml;
        }
    };
    uint32_t ip = 0;
    // the bytes of a length that follow a token's 15: adds them to `len`, stops behind the first that is not 255
    auto readLength = [&](uint32_t& len) {
        for (;;) {
            if (ip >= n) {
                bad = true;
                return;
            }
            const uint32_t x = src[ip];
            ++ip;
            if (x > 0x7FFFFFFFu - len) {
                bad = true;
                return;
            }
            len += x;
            if (x != 255u) return;
        }
    };
    while (ip < n && !bad) {
        uint64_t w0;
        uint32_t w1;
        fetch(ip, w0, w1);
        const uint32_t token = uint32_t(w0) & 0xFFu;
        const uint32_t fl = token >> 4;
        if (fl <= 8u && (token & 15u) != 15u && ip + 3u + fl <= n) {  // the record holds the literals themselves
            const uint64_t mid = (w0 >> 32) | (uint64_t(w1) << 32);  // bytes 4..11
            const uint32_t p = 8u * (1u + fl);
            const uint32_t off = uint32_t(p <= 48u ? w0 >> p : mid >> (p - 32u)) & 0xFFFFu;
            const uint32_t fm = (token & 15u) + 4u;
            const uint64_t lits = (w0 >> 8) | (uint64_t(w1) << 56);  // bytes 1..8
            account(fl, fm, off);
            emit(fl, fm | (off << 16), uint32_t(lits), uint32_t(lits >> 32));
            ip += 3u + fl;
            continue;
        }
        ++ip;
        uint32_t lit = fl;
        if (lit == 15u) readLength(lit);
        if (bad || lit > n - ip || lit > kSeqCount) {
            bad = true;
            break;
        }
        const uint32_t litAt = ip;
        ip += lit;
        if (ip >= n) {  // the last sequence of a block is literals only
            account(lit, 0, 0);
            emit(lit | kSeqExternal, 0, litAt, 0);
            break;
        }
        if (n - ip < 2) {
            bad = true;
            break;
        }
        const uint32_t off = uint32_t(src[ip]) | (uint32_t(src[ip + 1]) << 8);
        ip += 2;
        uint32_t ml = token & 15u;
        if (ml == 15u) readLength(ml);
        if (bad || ml > 0x7FFFFFF0u) {
            bad = true;
            break;
        }
        ml += 4;
        account(lit, ml, off);
        // (a record's match length has sixteen bits: a longer match is as many records, each reading `off` bytes back)
        uint32_t piece = ml < 65535u ? ml : 65535u;
        emit(lit | kSeqExternal, piece | (off << 16), litAt, 0);
        for (ml -= piece; ml > 0 && !bad; ml -= piece) {
            piece = ml < 65535u ? ml : 65535u;
            emit(0, piece | (off << 16), 0, 0);
        }
    }
    summaries[g] = bad ? Lz4BlockSummary{0, 0, 0, 0} : Lz4BlockSummary{k, produced, reachesBack, 0};
    if (bad) atomicOr(status, uint32_t(SCAN_BAD_BLOB));
}

// ---- the copies of a frame, 1024 sequences at a time --------------------------------------------------------------
// Lz4BatchCopyKernel: a workgroup per frame takes the records of a block in batches of up to 1024 sequences / 8 KB of
// output.  Inside a batch nothing is serial: a prefix sum of the records' lengths places every sequence, every output
// byte finds its sequence (the records' first bytes are marked, a running maximum carries the owner along) and its
// source -- a literal, a byte in front of the batch (in the ring, final), or an earlier byte of the batch.  Chains of
// the last kind are shortened by pointer jumping (a few rounds: as many as the logarithm of the longest chain of
// matches inside the batch), until every byte points at one whose value is known; those are written to the ring first,
// the others fetch theirs from there.  The ring in LDS holds the frame's last 64 KB and a batch more, so that no match
// reaches past it; whole words go out to the ring and to HBM.  A sequence longer than a batch is copied by all threads
// together, a match among them in pieces that read what the pieces before wrote.
#ifdef SSA_LZ4_STATS
__device__ unsigned long long gLz4Stats[16];
#define LZ4_T(i) do { if (tid == 0) { const uint32_t now_ = uint32_t(wall_clock64()); tacc[i] += now_ - tlast; tlast = now_; } } while (0)
#define LZ4_C(i, v) do { if (tid == 0) tacc[i] += (v); } while (0)
#else
#define LZ4_T(i) do {} while (0)
#define LZ4_C(i, v) do {} while (0)
#endif
constexpr uint32_t kBatchThreads = 1024;
constexpr uint32_t kBatchWaves = kBatchThreads / 64;
constexpr uint32_t kBatchCap = 8 * kBatchThreads;                      // output bytes of a batch, counted from the aligned word below its first
constexpr uint32_t kBatchRing = 65536 + kBatchCap + 256;  // bytes of the ring
constexpr uint32_t kBatchRecAt = kBatchRing;
constexpr uint32_t kBatchFromAt = kBatchRecAt + kBatchThreads * 16;
constexpr uint32_t kBatchWordsAt = kBatchFromAt + (kBatchCap + 8) * 2;
constexpr uint32_t kBatchLds = kBatchWordsAt + 6 * kBatchWaves * 4;
static_assert(kBatchRing % 4 == 0 && kBatchFromAt % 16 == 0 && kBatchWordsAt % 16 == 0, "alignment of the batch copy's LDS");

// Inclusive scans across a wave by data-parallel primitives (row shifts, then the rows' last lanes broadcast): no LDS round trips.
template <bool kMax>
__device__ __forceinline__ uint32_t WaveScan(uint32_t v) {
    auto op = [](uint32_t a, uint32_t b) { return kMax ? (a > b ? a : b) : a + b; };
    v = op(uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false)), v);  // row_shr:1
    v = op(uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false)), v);  // row_shr:2
    v = op(uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false)), v);  // row_shr:4
    v = op(uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false)), v);  // row_shr:8
    v = op(uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false)), v);  // row_bcast:15 into rows 1 and 3
    v = op(uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false)), v);  // row_bcast:31 into rows 2 and 3
    return v;
}

__global__ __launch_bounds__(kBatchThreads) void Lz4BatchCopyKernel(const uint8_t* blob, const Lz4FrameTask* tasks, const Lz4BlockRef* blocks,
                                                                    const uint64_t* recStart, const Lz4Seq* recs, const Lz4BlockSummary* summaries,
                                                                    uint32_t* status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ring = lds;
    uint32_t* ring32 = reinterpret_cast<uint32_t*>(lds);
    uint4* brec = reinterpret_cast<uint4*>(lds + kBatchRecAt);
    uint16_t* from16 = reinterpret_cast<uint16_t*>(lds + kBatchFromAt);
    uint32_t* wsh = reinterpret_cast<uint32_t*>(lds + kBatchWordsAt);  // per wave: lengths, counts, ends, owners (two chunks), flags of the jumping rounds
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const Lz4FrameTask t = tasks[blockIdx.x];
    __attribute__((address_space(1))) uint8_t* dst = (__attribute__((address_space(1))) uint8_t*)t.dst;
    const uint32_t total = uint32_t(t.dst_bytes);
    uint32_t op = 0;   // bytes of the frame written
    uint32_t rop = 0;  // op's place in the ring (op mod kBatchRing)
    bool bad = false;
#ifdef SSA_LZ4_STATS
    uint32_t tacc[11] = {0};
    uint32_t tlast = uint32_t(wall_clock64());
#endif
    // the place in the ring of the byte `rel` bytes from op (|rel| below the ring's size)
    auto ringAt = [&](int32_t rel) -> uint32_t {
        int32_t i = int32_t(rop) + rel;
        i += i < 0 ? int32_t(kBatchRing) : 0;
        i -= i >= int32_t(kBatchRing) ? int32_t(kBatchRing) : 0;
        return uint32_t(i);
    };
    for (uint32_t b = 0; b < t.nblocks && !bad; ++b) {
        const uint64_t g = t.first_block + b;
        const Lz4BlockRef blk = blocks[g];
        const uint8_t* bsrc = blob + blk.Src;
        const uint32_t n = blk.SrcBytes;
        if (n == 0) continue;
        if (blk.Stored) {
            if (n > total - op) {
                bad = true;
                break;
            }
            for (uint32_t i0 = 0; i0 < n; i0 += 16u * kBatchThreads) {  // (a piece is far shorter than the ring: no two of its bytes share a place)
                uint8_t x[16];
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) {
                    const uint32_t i = i0 + q * kBatchThreads + tid;
                    x[q] = i < n ? bsrc[i] : uint8_t(0);
                }
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) {
                    const uint32_t i = i0 + q * kBatchThreads + tid;
                    if (i < n) {
                        ring[(rop + i) % kBatchRing] = x[q];
                        dst[op + i] = x[q];
                    }
                }
                __syncthreads();
            }
            op += n;
            rop = (rop + n) % kBatchRing;
            continue;
        }
        const Lz4BlockSummary sum = summaries[g];
        // (the parse has checked every sequence against the block itself: what is left is the block against the frame)
        if (sum.Sequences == 0 || sum.Bytes > total - op || sum.ReachesBack > (t.linked ? op : 0u)) {
            bad = true;
            break;
        }
        const uint32_t nrec = sum.Sequences;
        const uint4* rec = reinterpret_cast<const uint4*>(recs + recStart[g]);
        uint32_t k = 0;
        while (k < nrec) {
            const uint32_t sh = op & 3u;            // the batch is laid out from the aligned word below op: position p is byte op - sh + p
            const uint32_t cap = kBatchCap - sh;
            const uint32_t navail = nrec - k < kBatchThreads ? nrec - k : kBatchThreads;
            const uint4 x = tid < navail ? rec[k + tid] : uint4{0, 0, 0, 0};
            const uint32_t lit = x.x & kSeqCount, ml = x.y & 0xFFFFu, off = x.y >> 16;
            const uint32_t len = lit + ml < kBatchCap + 1u ? lit + ml : kBatchCap + 1u;
            LZ4_T(0);
            // the marks of the records' first bytes start from zero (the array is free: its last readers are behind a barrier)
            reinterpret_cast<uint4*>(from16)[tid] = uint4{0, 0, 0, 0};
            if (tid == 0) {
                reinterpret_cast<uint4*>(from16)[kBatchThreads] = uint4{0, 0, 0, 0};
                wsh[5 * kBatchWaves] = 0;  // (the flags of the jumping rounds)
                wsh[5 * kBatchWaves + 1] = 0;
                wsh[5 * kBatchWaves + 2] = 0;
            }
            uint32_t end = WaveScan<false>(len);
            if (lane == 63) wsh[w] = end;
            __syncthreads();
            for (uint32_t q = 0; q < w; ++q) end += wsh[q];
            const uint32_t start = end - len;
            const bool ok = tid < navail && end <= cap;
            {
                const uint64_t oks = __ballot(ok);
                const uint32_t cnt = uint32_t(__popcll(oks));
                const uint32_t lastEnd = uint32_t(__builtin_amdgcn_readlane(int(end), int(cnt ? cnt - 1u : 0u)));
                if (lane == 0) {
                    wsh[kBatchWaves + w] = cnt;
                    wsh[2 * kBatchWaves + w] = cnt ? lastEnd : 0u;
                }
            }
            if (ok) {
                brec[tid] = uint4{(start + sh) | (lit << 14) | ((x.x & kSeqExternal) ? 0x80000000u : 0u), off, x.z, x.w};
                if (len != 0) from16[sh + start] = uint16_t(tid + 1u);
            }
            __syncthreads();
            uint32_t m = 0, bytes = 0;  // the batch: its records and what they decode to
#pragma unroll
            for (uint32_t q = 0; q < kBatchWaves; ++q) {
                m += wsh[kBatchWaves + q];
                bytes = max(bytes, wsh[2 * kBatchWaves + q]);
            }
            LZ4_T(1);
            LZ4_C(8, 1);
            LZ4_C(9, m);
            if (m == 0) {
                // A sequence longer than a batch: all threads copy its literals, then its match.
                const uint4 s = rec[k];
                const uint32_t slit = s.x & kSeqCount, sml = s.y & 0xFFFFu, soff = s.y >> 16;
                if (slit > total - op || sml > total - op - slit) {
                    bad = true;
                    break;
                }
                const uint64_t inl = uint64_t(s.z) | (uint64_t(s.w) << 32);
                for (uint32_t i0 = 0; i0 < slit; i0 += 32768u) {
                    const uint32_t i1 = slit - i0 < 32768u ? slit : i0 + 32768u;
                    for (uint32_t i = i0 + tid; i < i1; i += kBatchThreads) {
                        const uint8_t v = (s.x & kSeqExternal) ? bsrc[s.z + i] : uint8_t(inl >> (8u * (i & 7u)));
                        ring[(rop + i) % kBatchRing] = v;
                        dst[op + i] = v;
                    }
                    __syncthreads();
                }
                op += slit;
                rop = (rop + slit) % kBatchRing;
                if (sml != 0) {
                    // Bytes further on than a piece read the match's own output a whole number of periods back (what lies
                    // `soff` back of them may have left the ring by then): written by the pieces before.
                    const uint32_t back = soff >= kBatchCap ? soff : soff * ((kBatchCap + soff - 1u) / soff);
                    for (uint32_t i0 = 0; i0 < sml; i0 += kBatchCap) {
                        uint8_t v[8];
#pragma unroll
                        for (uint32_t q = 0; q < 8; ++q) {
                            const uint32_t i = i0 + q * kBatchThreads + tid;
                            v[q] = 0;
                            if (i < sml) v[q] = ring[ringAt(i < kBatchCap ? int32_t(i % soff) - int32_t(soff) : int32_t(i) - int32_t(back))];
                        }
#pragma unroll
                        for (uint32_t q = 0; q < 8; ++q) {
                            const uint32_t i = i0 + q * kBatchThreads + tid;
                            if (i < sml) {
                                ring[ringAt(int32_t(i))] = v[q];
                                dst[op + i] = v[q];
                            }
                        }
                        __syncthreads();
                    }
                    rop = ringAt(int32_t(sml));
                    op += sml;
                }
                k += 1;
                LZ4_T(7);
                continue;
            }
            const uint32_t nch = (sh + bytes + 4u * kBatchThreads - 1u) / (4u * kBatchThreads);  // chunks of a word per thread
            // the sequence every byte belongs to: the last mark at or before it
            uint32_t own[2][4];
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                if (c < nch) {
                    const uint2 mk = reinterpret_cast<const uint2*>(from16)[tid + kBatchThreads * c];
                    own[c][0] = mk.x & 0xFFFFu;
                    own[c][1] = max(own[c][0], mk.x >> 16);
                    own[c][2] = max(own[c][1], mk.y & 0xFFFFu);
                    own[c][3] = max(own[c][2], mk.y >> 16);
                    const uint32_t s = WaveScan<true>(own[c][3]);
                    const uint32_t before = uint32_t(__builtin_amdgcn_update_dpp(0, int(s), 0x138, 0xf, 0xf, false));  // wave_shr:1
                    if (lane == 63) wsh[3 * kBatchWaves + c * kBatchWaves + w] = s;
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) own[c][i] = max(own[c][i], before);
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                if (c < nch) {
                    uint32_t carry = 0;
                    for (uint32_t q = 0; q < c * kBatchWaves + w; ++q) carry = max(carry, wsh[3 * kBatchWaves + q]);
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) own[c][i] = max(own[c][i], carry);
                }
            }
            LZ4_T(2);
            // every byte's source; the values of literals and of bytes that come from in front of the batch
            uint32_t from[2][4];
            uint32_t val[2] = {0, 0}, inside[2] = {0, 0}, open[2] = {0, 0};
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                if (c < nch) {
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        const uint32_t p = 4u * (tid + kBatchThreads * c) + i;
                        from[c][i] = p;
                        if (p >= sh && p < sh + bytes) {
                            inside[c] |= 1u << i;
                            const uint4 r = brec[own[c][i] - 1u];
                            const uint32_t rs = r.x & 0x3FFFu, rlit = (r.x >> 14) & 0x3FFFu;
                            const uint32_t at = p - rs;
                            uint32_t v = 0;
                            if (at < rlit) {
                                v = (r.x >> 31) ? uint32_t(bsrc[r.z + at]) : uint32_t((uint64_t(r.z) | (uint64_t(r.w) << 32)) >> (8u * (at & 7u))) & 0xFFu;
                            } else {
                                uint32_t mi = at - rlit;
                                if (mi >= r.y) {  // the match repeats a pattern of r.y bytes: mi mod r.y (both below 8192: the rounded quotient is off by one at most)
                                    const uint32_t qf = uint32_t((float(mi) + 0.5f) * __builtin_amdgcn_rcpf(float(r.y)));
                                    mi -= qf * r.y;
                                    if (int32_t(mi) < 0) mi += r.y;
                                    else if (mi >= r.y) mi -= r.y;
                                }
                                const int32_t s = int32_t(rs + rlit + mi) - int32_t(r.y);
                                if (s < int32_t(sh)) {
                                    v = ring[ringAt(s - int32_t(sh))];
                                } else {
                                    from[c][i] = uint32_t(s);
                                    open[c] |= 1u << i;
                                }
                            }
                            val[c] |= v << (8u * i);
                        }
                    }
                    reinterpret_cast<uint2*>(from16)[tid + kBatchThreads * c] = uint2{from[c][0] | (from[c][1] << 16), from[c][2] | (from[c][3] << 16)};
                }
            }
            __syncthreads();
            LZ4_T(3);
            // pointer jumping: a byte whose source is itself waiting takes over that byte's source
            uint32_t live[2] = {open[0], open[1]};
            for (uint32_t round = 0;; ++round) {
                int changed = 0;
                uint32_t* flag = wsh + 5 * kBatchWaves + (round % 3u);  // three flags in turn: the one after this round's is cleared meanwhile
                if (tid == 0) wsh[5 * kBatchWaves + ((round + 1u) % 3u)] = 0;
#pragma unroll
                for (uint32_t c = 0; c < 2; ++c) {
                    if (c < nch && live[c] != 0) {
#pragma unroll
                        for (uint32_t i = 0; i < 4; ++i) {
                            if (live[c] & (1u << i)) {
                                const uint32_t f = from[c][i];
                                const uint32_t gq = from16[f];
                                if (gq != f) {
                                    from[c][i] = gq;
                                    changed = 1;
                                } else {
                                    live[c] &= ~(1u << i);
                                }
                            }
                        }
                        reinterpret_cast<uint2*>(from16)[tid + kBatchThreads * c] = uint2{from[c][0] | (from[c][1] << 16), from[c][2] | (from[c][3] << 16)};
                    }
                }
                LZ4_C(10, 1);
                if (__ballot(changed != 0) != 0 && lane == 0) *flag = 1;
                __syncthreads();
                if (*flag == 0) break;
            }
            LZ4_T(4);
            // the known bytes into the ring, then the others fetch theirs
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                if (c < nch) {
                    const uint32_t known = inside[c] & ~open[c];
                    if (known != 0 && (open[c] != 0 || inside[c] != 0xFu)) {  // (a word that is complete goes out whole below)
#pragma unroll
                        for (uint32_t i = 0; i < 4; ++i) {
                            if (known & (1u << i)) ring[ringAt(int32_t(4u * (tid + kBatchThreads * c) + i) - int32_t(sh))] = uint8_t(val[c] >> (8u * i));
                        }
                    } else if (known != 0) {
                        ring32[ringAt(int32_t(4u * (tid + kBatchThreads * c)) - int32_t(sh)) >> 2] = val[c];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                if (c < nch && inside[c] != 0) {
                    const uint32_t p0 = 4u * (tid + kBatchThreads * c);
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        if (open[c] & (1u << i)) val[c] |= uint32_t(ring[ringAt(int32_t(from[c][i]) - int32_t(sh))]) << (8u * i);
                    }
                    if (inside[c] == 0xFu) {
                        if (open[c] != 0) ring32[ringAt(int32_t(p0) - int32_t(sh)) >> 2] = val[c];
                        *reinterpret_cast<__attribute__((address_space(1))) uint32_t*>(dst + (op - sh + p0)) = val[c];
                    } else {
#pragma unroll
                        for (uint32_t i = 0; i < 4; ++i) {
                            if (inside[c] & (1u << i)) {
                                const uint8_t v = uint8_t(val[c] >> (8u * i));
                                if (open[c] & (1u << i)) ring[ringAt(int32_t(p0 + i) - int32_t(sh))] = v;
                                dst[op - sh + p0 + i] = v;
                            }
                        }
                    }
                }
            }
            rop = ringAt(int32_t(bytes));
            op += bytes;
            k += m;
            LZ4_T(5);
        }
    }
    if ((bad || op != total) && tid == 0) atomicOr(status, uint32_t(SCAN_BAD_BLOB));
#ifdef SSA_LZ4_STATS
    if (tid == 0) {
        for (int i = 0; i < 11; ++i) atomicAdd(&gLz4Stats[i], (unsigned long long)tacc[i]);
    }
#endif
}

}  // namespace

hipError_t LaunchLz4Decode(const uint8_t* blob, const Lz4FrameTask* tasks, int64_t ntasks, const Lz4BlockRef* blocks, uint32_t* status, hipStream_t st) {
    if (ntasks <= 0) return hipSuccess;
    hipLaunchKernelGGL(Lz4DecodeKernel, dim3(uint32_t(ntasks)), dim3(64), 0, st, blob, tasks, blocks, status);
    return hipGetLastError();
}

hipError_t LaunchLz4DecodeSplit(const uint8_t* blob, const Lz4FrameTask* tasks, int64_t ntasks, const Lz4BlockRef* blocks, int64_t nblocks,
                                const uint64_t* recStart, Lz4Seq* recs, Lz4BlockSummary* summaries, uint32_t* status, hipStream_t st) {
    if (ntasks <= 0 || nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(Lz4ParseKernel, dim3(uint32_t((nblocks + 63) / 64)), dim3(64), 0, st, blob, blocks, uint32_t(nblocks), recStart, recs, summaries, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    static std::atomic<uint64_t> attributeSetOn{0};  // one bit per device
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!(attributeSetOn.load() >> (dev & 63) & 1u)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&Lz4BatchCopyKernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(kBatchLds));
        if (e != hipSuccess) return e;
        attributeSetOn.fetch_or(uint64_t(1) << (dev & 63));
    }
    hipLaunchKernelGGL(Lz4BatchCopyKernel, dim3(uint32_t(ntasks)), dim3(kBatchThreads), kBatchLds, st, blob, tasks, blocks, recStart, recs, summaries, status);
#ifdef SSA_LZ4_STATS
    {
        (void)hipStreamSynchronize(st);
        unsigned long long h[16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gLz4Stats), sizeof(h));
        std::fprintf(stderr, "lz4 batch copy, %lld frames: record load %.1f, place %.1f, owners %.1f, sources %.1f, jumping %.1f, output %.1f, long %.1f (us per frame); batches %llu, records per batch %.1f, rounds per batch %.2f\n",
                     (long long)ntasks, h[0] / 100.0 / ntasks, h[1] / 100.0 / ntasks, h[2] / 100.0 / ntasks, h[3] / 100.0 / ntasks, h[4] / 100.0 / ntasks, h[5] / 100.0 / ntasks,
                     h[7] / 100.0 / ntasks, h[8], double(h[9]) / double(h[8] ? h[8] : 1), double(h[10]) / double(h[8] ? h[8] : 1));
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(gLz4Stats), z, sizeof(z));
    }
#endif
    return hipGetLastError();
}

}  // namespace ssa
