// flacmeta.cpp — CPU side of libflacgpu: the metadata walk that precedes the
// frame engine.  Follows the reference's (*Metadata).Read and its Marshal*
// helpers (flacmeta.go:519-551, :612-641, :644-717, :749-850) far enough to
// hand the frame engine what it needs: the STREAMINFO values, the block table,
// the seek points and the byte offset of frame 0.
#include "../../include/flacgpu.h"

#include <cstring>

namespace {

inline uint32_t be24(const uint8_t *p) { return (uint32_t(p[0]) << 16) | (uint32_t(p[1]) << 8) | p[2]; }
inline uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline uint32_t le32(const uint8_t *p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
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

int flacgpu_parse_vorbis_comment(const uint8_t *b, size_t n, flacgpu_vorbis_comment *out, uint32_t *offsets, uint32_t *lengths,
                                 size_t cap, size_t *count)
{
    if (!b || !out || !count) return FLACGPU_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    size_t o = 0;
    if (n < 8) return FLACGPU_ERR_TRUNCATED;
    const uint32_t vl = le32(b);                                   // flacmeta.go:736
    o = 4;
    if (vl > n - o || n - o - vl < 4) return FLACGPU_ERR_TRUNCATED;
    out->vendor_offset = (uint32_t)o; out->vendor_length = vl;     // :737
    o += vl;
    out->total_comments = le32(b + o);                             // :739
    o += 4;
    *count = out->total_comments;
    int rc = FLACGPU_OK;
    for (uint32_t k = 0; k < out->total_comments; ++k) {           // :741-744
        if (n - o < 4) return FLACGPU_ERR_TRUNCATED;
        const uint32_t cl = le32(b + o);
        o += 4;
        if (cl > n - o) return FLACGPU_ERR_TRUNCATED;
        if (offsets && lengths) {
            if (k < cap) { offsets[k] = (uint32_t)o; lengths[k] = cl; } else rc = FLACGPU_ERR_CAPACITY;
        }
        o += cl;
    }
    return rc;
}

int flacgpu_parse_picture(const uint8_t *b, size_t n, flacgpu_picture *out)
{
    if (!b || !out) return FLACGPU_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    size_t o = 0;
    auto u32 = [&](uint32_t &v) { if (n - o < 4) return false; v = be32(b + o); o += 4; return true; };
    uint32_t len = 0;
    if (!u32(out->picture_type)) return FLACGPU_ERR_TRUNCATED;     // flacmeta.go:586
    if (!u32(len) || len > n - o) return FLACGPU_ERR_TRUNCATED;    // :588-589
    out->mime_offset = (uint32_t)o; out->mime_length = len; o += len;
    if (!u32(len) || len > n - o) return FLACGPU_ERR_TRUNCATED;    // :591-594
    out->description_offset = (uint32_t)o; out->description_length = len; o += len;
    if (!u32(out->width) || !u32(out->height) || !u32(out->color_depth) || !u32(out->num_colors) || !u32(out->length))
        return FLACGPU_ERR_TRUNCATED;                              // :595-599
    out->blob_offset = (uint32_t)o;
    out->blob_length = (uint32_t)(out->length < n - o ? out->length : n - o);   // buf.Next gives what is left (:601)
    return FLACGPU_OK;
}

const char *flacgpu_picture_type_string(uint32_t t)
{
    static const char *names[21] = {                               // flacmeta.go:117-139
        "Other", "File Icon", "Other File Icon", "Cover (front)", "Cover (back)", "Leaflet Page", "Media",
        "Lead Artist/Lead Performer/Soloist", "Artist/Performer", "Conductor", "Band/Orchestra", "Composer",
        "Lyricist/Text Writer", "Recording Location", "During Recording", "During Performance",
        "Movie/Video Screen Capture", "A Bright Coloured Fish", "Illustration", "Band/Artist Logotype",
        "Publisher/Studio Logotype"};
    return t <= 20 ? names[t] : "UNKNOWN";                         // :165-171
}

int flacgpu_parse_application(const uint8_t *b, size_t n, flacgpu_application *out)
{
    if (!b || !out) return FLACGPU_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    if (n < 4) return FLACGPU_ERR_TRUNCATED;                       // Uint32 of a short slice panics (:360)
    out->id = be32(b);
    const size_t skip = n < 32 ? n : 32;                           // buf.Next(ApplicationIdLen): 32 BYTES (:360)
    if ((n - skip) % 8 != 0) return FLACGPU_ERR_APPLICATION;       // :361
    out->data_offset = (uint32_t)skip; out->data_length = (uint32_t)(n - skip);   // :364
    return FLACGPU_OK;
}

int flacgpu_parse_cuesheet(const uint8_t *b, size_t n, flacgpu_cuesheet *out, flacgpu_cuesheet_track *tracks, size_t track_cap,
                           flacgpu_cuesheet_index *indexes, size_t index_cap, size_t *n_indexes)
{
    if (!b || !out) return FLACGPU_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    if (n_indexes) *n_indexes = 0;
    // 128 bytes catalog number, 8 lead-in, 259 reserved (first bit: compact disc), 1 track count (:403-408)
    if (n < 128 + 8 + 259 + 1) return FLACGPU_ERR_TRUNCATED;
    std::memcpy(out->media_catalog_number, b, 128);
    out->leadin_samples = be64(b + 128);
    out->is_compact_disc = b[136] >> 7;
    out->total_tracks = b[395];
    if (out->total_tracks == 0) return FLACGPU_ERR_CUESHEET;       // :409
    size_t o = 396, ni = 0;
    int rc = FLACGPU_OK;
    for (uint32_t i = 0; i < out->total_tracks; ++i) {
        if (n - o < 36) return FLACGPU_ERR_TRUNCATED;              // a track record is 36 bytes (:414)
        flacgpu_cuesheet_track t;
        std::memset(&t, 0, sizeof t);
        t.offset = be64(b + o);                                    // :470
        t.number = b[o + 8];
        if (t.number == 0) return FLACGPU_ERR_CUESHEET;            // :472-474
        std::memcpy(t.isrc, b + o + 9, 12);                        // :475
        t.type = (b[o + 21] >> 7) & 1;                             // :479
        t.pre_emphasis = (b[o + 21] >> 6) & 1;                     // :480
        t.index_points = b[o + 35];                                // :481
        t.first_index = (uint32_t)ni;
        o += 36;
        for (uint32_t j = 0; j < t.index_points; ++j) {
            if (n - o < 12) return FLACGPU_ERR_TRUNCATED;          // an index record is 12 bytes (:419)
            flacgpu_cuesheet_index x;
            x.sample_offset = be64(b + o);                         // :510
            if (x.sample_offset % 588 != 0) return FLACGPU_ERR_CUESHEET;   // :511-513
            x.index_point = b[o + 8];                              // :514
            if (indexes) { if (ni < index_cap) indexes[ni] = x; else rc = FLACGPU_ERR_CAPACITY; }
            ++ni;
            o += 12;
        }
        if (tracks) { if (i < track_cap) tracks[i] = t; else rc = FLACGPU_ERR_CAPACITY; }
    }
    if (n_indexes) *n_indexes = ni;
    return rc;
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
        } else if (blk.type == FLACGPU_BLOCK_APPLICATION) {
            flacgpu_application a;
            rc = flacgpu_parse_application(buf + pos, blk.length, &a);             // :813-816
            if (rc) return rc;
        } else if (blk.type == FLACGPU_BLOCK_CUESHEET) {
            flacgpu_cuesheet c;
            rc = flacgpu_parse_cuesheet(buf + pos, blk.length, &c, nullptr, 0, nullptr, 0, nullptr);   // :835-838
            if (rc) return rc;
        } else if (blk.type == FLACGPU_BLOCK_VORBIS_COMMENT) {
            flacgpu_vorbis_comment v;
            size_t cnt = 0;
            rc = flacgpu_parse_vorbis_comment(buf + pos, blk.length, &v, nullptr, nullptr, 0, &cnt);   // :796 (a short block panics there)
            if (rc) return rc;
        } else if (blk.type == FLACGPU_BLOCK_PICTURE) {
            flacgpu_picture p;
            rc = flacgpu_parse_picture(buf + pos, blk.length, &p);                 // :800
            if (rc) return rc;
        }
        if (out->n_blocks >= FLACGPU_MAX_BLOCKS) return FLACGPU_ERR_CAPACITY;
        out->blocks[out->n_blocks++] = blk;
        pos += blk.length;
        if (blk.last) break;                                    // :845
    }
    out->first_frame_offset = pos;                              // :849
    return FLACGPU_OK;
}

// Measures the ID3v2 tag(s) in front of the "fLaC" marker.  Not part of Read: the reference
// refuses such files (flacmeta.go:754-761) and flacgpu_parse_metadata stays as strict.
int flacgpu_skip_id3v2(const uint8_t *buf, size_t len, size_t *offset)
{
    if (!buf || !offset) return FLACGPU_ERR_ARG;
    size_t pos = 0;
    *offset = 0;
    for (;;) {
        if (len - pos < 10) break;   // no room for another tag header: whatever follows is not a tag
        const uint8_t *t = buf + pos;
        if (t[0] != 'I' || t[1] != 'D' || t[2] != '3') break;
        if (t[3] == 0xFF || t[4] == 0xFF) break;
        if ((t[6] | t[7] | t[8] | t[9]) & 0x80) break;       // not a sync-safe size: not a tag
        const size_t body = ((size_t)t[6] << 21) | ((size_t)t[7] << 14) | ((size_t)t[8] << 7) | (size_t)t[9];
        const size_t total = 10 + body + ((t[5] & 0x10) ? 10 : 0);
        if (len - pos < total) return FLACGPU_ERR_TRUNCATED;
        pos += total;
        *offset = pos;
    }
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
    case FLACGPU_ERR_CUESHEET: return "invalid CUESHEET block";
    case FLACGPU_ERR_APPLICATION: return "malformed APPLICATION block";
    default: return "unknown status";
    }
}

}  // extern "C"
