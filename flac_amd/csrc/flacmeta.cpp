// flacmeta.cpp — CPU side of libflacgpu: the metadata walk that precedes the
// frame engine.  Follows the reference's (*Metadata).Read and its Marshal*
// helpers (flacmeta.go:519-551, :612-641, :644-717, :749-850) far enough to
// hand the frame engine what it needs: the STREAMINFO values, the block table,
// the seek points and the byte offset of frame 0.
#include "../../include/flacgpu.h"

#include <cstring>

namespace {

inline uint32_t be24(const uint8_t *p) { return (uint32_t(p[0]) << 16) | (uint32_t(p[1]) << 8) | p[2]; }
inline uint64_t be64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v = (v << 8) | p[i];
    return v;
}

// HeaderType, flacmeta.go:142-162: wire codes 0..6 map to enum values 1..7.
inline uint32_t header_type(uint32_t wire) { return wire <= 6 ? wire + 1 : uint32_t(FLACGPU_BLOCK_INVALID); }

}  // namespace

extern "C" {

int flacgpu_parse_block_header(const uint8_t *b, size_t n, flacgpu_block *out)
{
    if (!b || !out) return FLACGPU_ERR_ARG;
    if (n < 4) return FLACGPU_ERR_TRUNCATED;
    std::memset(out, 0, sizeof *out);
    out->last = b[0] >> 7;                       // flacmeta.go:541
    out->length = be24(b + 1);                   // :542
    out->type = header_type(b[0] & 0x7F);        // :544-545
    if (out->type == FLACGPU_BLOCK_INVALID) return FLACGPU_ERR_BLOCK_TYPE;  // :546
    return FLACGPU_OK;
}

int flacgpu_parse_streaminfo(const uint8_t *b, size_t n, flacgpu_streaminfo *out)
{
    if (!b || !out) return FLACGPU_ERR_ARG;
    if (n < 34) return FLACGPU_ERR_TRUNCATED;    // the reference panics here (:687..)
    flacgpu_streaminfo s;
    std::memset(&s, 0, sizeof s);
    s.min_blocksize = uint16_t((b[0] << 8) | b[1]);
    if (s.min_blocksize > 0 && s.min_blocksize < 16) return FLACGPU_ERR_STREAMINFO;   // :688
    s.max_blocksize = uint16_t((b[2] << 8) | b[3]);
    if (s.max_blocksize < 16) return FLACGPU_ERR_STREAMINFO;                          // :695
    // The reference takes uint32(bits >> 24) of MaxBlockSize|MinFrameSize|MaxFrameSize
    // (:699), which keeps the low byte of MaxBlockSize in bits 24-31.  The 24-bit
    // field is returned here; the two agree whenever MaxBlockSize % 256 == 0.
    s.min_framesize = be24(b + 4);
    s.max_framesize = be24(b + 7);                                                    // :700
    const uint64_t w = be64(b + 10);                                                  // :702
    s.sample_rate = uint32_t(w >> 44);                                                // :707
    if (s.sample_rate == 0 || s.sample_rate >= 655350) return FLACGPU_ERR_STREAMINFO; // :708
    s.channels = uint8_t(((w >> 41) & 0x7) + 1);                                      // :711
    s.bits_per_sample = uint8_t(((w >> 36) & 0x1F) + 1);                              // :712
    s.total_samples = w & ((uint64_t(1) << 36) - 1);                                  // :713
    std::memcpy(s.md5, b + 18, 16);                                                   // :714
    *out = s;
    return FLACGPU_OK;
}

int flacgpu_parse_seektable(const uint8_t *b, size_t n, flacgpu_seekpoint *out, size_t cap, size_t *count)
{
    if (!b || !count) return FLACGPU_ERR_ARG;
    if (n % 18 != 0) return FLACGPU_ERR_SEEKTABLE;             // flacmeta.go:823
    const size_t pts = n / 18;
    *count = pts;
    if (!out) return FLACGPU_OK;
    if (pts > cap) return FLACGPU_ERR_CAPACITY;
    for (size_t i = 0; i < pts; ++i) {                          // :635-639
        const uint8_t *p = b + 18 * i;
        out[i].sample_number = be64(p);
        out[i].offset = be64(p + 8);
        out[i].frame_samples = uint16_t((p[16] << 8) | p[17]);
    }
    return FLACGPU_OK;
}

int flacgpu_parse_metadata(const uint8_t *buf, size_t len, flacgpu_metadata *out)
{
    if (!buf || !out) return FLACGPU_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    if (len < 4) return FLACGPU_ERR_TRUNCATED;                  // flacmeta.go:754-757
    if (std::memcmp(buf, "fLaC", 4) != 0) return FLACGPU_ERR_SIGNATURE;   // :759
    size_t pos = 4;
    uint32_t seen = 0;   // bit per once-only block kind
    for (;;) {
        if (len - pos < 4) return FLACGPU_ERR_TRUNCATED;        // :765-768
        flacgpu_block blk;
        int rc = flacgpu_parse_block_header(buf + pos, 4, &blk);
        if (rc) return rc;                                      // :770-773
        pos += 4;
        if (len - pos < blk.length) return FLACGPU_ERR_TRUNCATED;   // :776-779
        blk.payload_offset = pos;
        if (blk.type != FLACGPU_BLOCK_PICTURE) {                // only PICTURE may repeat (:799-801)
            if (seen & (1u << blk.type)) return FLACGPU_ERR_DUPLICATE;   // :783,793,804,810,820,831
            seen |= 1u << blk.type;
        }
        if (blk.type == FLACGPU_BLOCK_STREAMINFO) {
            rc = flacgpu_parse_streaminfo(buf + pos, blk.length, &out->streaminfo);   // :786
            if (rc) return rc;
            out->has_streaminfo = 1;
        } else if (blk.type == FLACGPU_BLOCK_SEEKTABLE) {
            if (blk.length % 18 != 0) return FLACGPU_ERR_SEEKTABLE;               // :823
        }
        if (out->n_blocks >= FLACGPU_MAX_BLOCKS) return FLACGPU_ERR_CAPACITY;
        out->blocks[out->n_blocks++] = blk;
        pos += blk.length;
        if (blk.last) break;                                    // :845
    }
    out->first_frame_offset = pos;                              // :849
    return FLACGPU_OK;
}

const char *flacgpu_status_string(int status)
{
    switch (status) {
    case FLACGPU_OK: return "ok";
    case FLACGPU_ERR_ARG: return "invalid argument";
    case FLACGPU_ERR_SIGNATURE: return "not a valid FLAC signature";
    case FLACGPU_ERR_BLOCK_TYPE: return "invalid or unknown block type";
    case FLACGPU_ERR_DUPLICATE: return "two blocks of a kind that may appear once";
    case FLACGPU_ERR_TRUNCATED: return "input ends inside a header or block";
    case FLACGPU_ERR_STREAMINFO: return "STREAMINFO value out of bounds";
    case FLACGPU_ERR_SEEKTABLE: return "SEEKTABLE block length is not a multiple of 18";
    case FLACGPU_ERR_CAPACITY: return "output capacity too small";
    case FLACGPU_ERR_DEVICE: return "HIP runtime error";
    case FLACGPU_ERR_NOMEM: return "out of memory";
    case FLACGPU_ERR_UNSUPPORTED: return "unsupported stream";
    default: return "unknown status";
    }
}

}  // extern "C"
