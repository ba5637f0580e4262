/* CPU restatement of LZ4 frame decoding.  TEST INFRASTRUCTURE -- the GPU product never links or calls this.
 *
 * The reference decompresses IPC buffers through arrow::util::Codec (LZ4_FRAME,
 * ydb/core/formats/arrow/serializer/native.h:54-57; contrib/libs/apache/arrow/cpp/src/arrow/util/compression_lz4.cc:60-110)
 * into the library it vendors as contrib/libs/lz4 (LZ4F_decompress, lz4frame.c; LZ4_decompress_safe, lz4.c).  This file
 * follows the published format those implement: frame = magic 0x184D2204, FLG, BD, optional content size / dictionary
 * id, header checksum, then blocks [u32 size, high bit: stored][data][optional checksum] up to a zero size, optional
 * content checksum; block = sequences of token (literal length << 4 | match length - 4), extended lengths as runs of 255,
 * literals, 16-bit little-endian match offset.  With linked blocks a match may reach into the output of earlier blocks.
 * Checksums are skipped, not verified.  Pinned against the reference's own library built from its sources
 * (oracle/_ref/liblz4_ref.so, in the build container) and against buffers written by Arrow (tests/golden/ipc).
 *
 * Returns the number of bytes produced, or a negative error. */
#include <stdint.h>
#include <string.h>

int64_t orc_lz4_frame_decode(const uint8_t* src, int64_t n, uint8_t* dst, int64_t cap) {
    int64_t ip = 0, op = 0;
    if (n < 7) return -1;
    uint32_t magic;
    memcpy(&magic, src, 4);
    if (magic != 0x184D2204u) return -2;
    const uint8_t flg = src[4], bd = src[5];
    if ((flg >> 6) != 1) return -3;
    const int independent = (flg >> 5) & 1, block_checksum = (flg >> 4) & 1, content_size = (flg >> 3) & 1,
              content_checksum = (flg >> 2) & 1, dict_id = flg & 1;
    const int size_id = (bd >> 4) & 7;
    if (size_id < 4) return -4;
    const int64_t block_max = (int64_t)1 << (8 + 2 * size_id);
    ip = 6;
    if (content_size) ip += 8;
    if (dict_id) ip += 4;
    ip += 1; /* header checksum */
    for (;;) {
        if (n - ip < 4) return -5;
        uint32_t word;
        memcpy(&word, src + ip, 4);
        ip += 4;
        if (word == 0) break;
        const int64_t bn = word & 0x7FFFFFFFu;
        if (bn > n - ip || bn > block_max) return -6;
        const uint8_t* b = src + ip;
        if (word >> 31) {
            if (bn > cap - op) return -7;
            memcpy(dst + op, b, (size_t)bn);
            op += bn;
        } else {
            const int64_t block_start = op;
            int64_t p = 0;
            while (p < bn) {
                const uint32_t token = b[p++];
                int64_t lit = token >> 4;
                if (lit == 15) {
                    uint32_t x;
                    do {
                        if (p >= bn) return -8;
                        x = b[p++];
                        lit += x;
                    } while (x == 255);
                }
                if (lit > bn - p || lit > cap - op) return -9;
                memcpy(dst + op, b + p, (size_t)lit);
                p += lit;
                op += lit;
                if (p >= bn) break; /* the last sequence has no match */
                if (bn - p < 2) return -10;
                const int64_t off = b[p] | ((int64_t)b[p + 1] << 8);
                p += 2;
                int64_t ml = token & 15;
                if (ml == 15) {
                    uint32_t x;
                    do {
                        if (p >= bn) return -11;
                        x = b[p++];
                        ml += x;
                    } while (x == 255);
                }
                ml += 4;
                const int64_t reach = independent ? op - block_start : op;
                if (off == 0 || off > reach || ml > cap - op) return -12;
                for (int64_t i = 0; i < ml; ++i) dst[op + i] = dst[op + i - off]; /* byte by byte: the match may overlap itself */
                op += ml;
            }
        }
        ip += bn;
        if (block_checksum) ip += 4;
    }
    if (content_checksum) ip += 4;
    return op;
}
