/*
 * flac_oracle.h — CPU oracle for the FLAC frame path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar, dependency-free C restatement of (a) the STREAMINFO /
 * block-header arithmetic of the reference Go package (flacmeta.go:519-551,
 * :644-717) and (b) the FLAC frame format (RFC 9639), which the reference does
 * not implement at all.  It exists so the tests can check the HIP library
 * bit-for-bit, and so bench.py can time a CPU baseline.  The product library
 * never links, loads or calls anything in this directory.
 *
 * Parity pinning: the decoder is pinned by the two fixture files of the
 * reference (STREAMINFO MD5 signatures asserted at flacmeta_test.go:37 and
 * :121, seek-table offsets at :81-85 and :225-232, frame-size bounds at
 * :31-32 and :115-116).  Features the fixtures do not exercise (LPC, fixed
 * orders 1-4, 5-bit Rice, escapes, wasted bits, >2 channels, 24-bit) are
 * "parity unpinned" by the reference; for those the oracle is cross-checked
 * against an independently written Python decoder (tests/pydecoder.py) and by
 * encode->decode round trips.
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors StreaminfoBlock (flacmeta.go:263-273); md5 kept as 16 raw bytes. */
typedef struct {
    uint16_t min_blocksize;
    uint16_t max_blocksize;
    uint32_t min_framesize;
    uint32_t max_framesize;
    uint32_t sample_rate;
    uint8_t  channels;
    uint8_t  bits_per_sample;
    uint64_t total_samples;
    uint8_t  md5[16];
} orc_streaminfo;

/* One resolved audio frame. */
typedef struct {
    uint64_t offset;        /* bytes from frame 0 of the stream            */
    uint32_t length;        /* bytes, header through CRC-16                */
    uint32_t blocksize;     /* inter-channel samples                       */
    uint64_t first_sample;  /* inter-channel sample index of sample 0      */
    uint64_t number;        /* coded number (frame or sample index)        */
    uint8_t  variable;      /* blocking strategy bit                       */
    uint8_t  ch_assign;     /* 0..10                                       */
    uint8_t  bps;           /* bits per sample of this frame               */
    uint8_t  header_len;    /* bytes incl. CRC-8                           */
    uint32_t sample_rate;
    int32_t  status;        /* ORC_FRAME_*                                 */
} orc_frame;

enum {
    ORC_FRAME_OK = 0,
    ORC_FRAME_BAD_CRC16 = 1,   /* no span starting here has a valid CRC-16 */
    ORC_FRAME_BAD_SUBFRAME = 2,/* reserved code / invalid field            */
    ORC_FRAME_TRUNCATED = 3,   /* ran past the end of the frame            */
    ORC_FRAME_LENGTH = 4,      /* decoded length != resolved length        */
    ORC_FRAME_RANGE = 5,       /* output would fall outside the buffer     */
    ORC_FRAME_UNSUPPORTED = 6
};

/* Status codes of the stream-level functions. */
enum {
    ORC_OK = 0,
    ORC_ERR_ARG = 1,
    ORC_ERR_STREAMINFO = 2,
    ORC_ERR_CAPACITY = 3
};

typedef struct {
    uint64_t sync_candidates;   /* 0xFFF8/0xFFF9 byte pairs                  */
    uint64_t header_candidates; /* ... that pass header validity + CRC-8     */
    uint64_t frames;            /* resolved frames (any status)              */
    uint64_t frames_ok;         /* status == ORC_FRAME_OK after decode/scan  */
    uint64_t samples;           /* inter-channel samples in OK frames        */
    uint64_t first_bad_frame;   /* index of the first frame whose status is not OK after the decode; ~0 = none */
    int32_t  first_error;       /* its ORC_FRAME_* status, 0 when there is none */
    uint32_t pad;
} orc_stats;

/* flacmeta.go:644-717 (with MinFrameSize masked to 24 bits, see :699). */
int orc_parse_streaminfo(const uint8_t *b, size_t n, orc_streaminfo *out);

uint8_t  orc_crc8(const uint8_t *p, size_t n);
uint16_t orc_crc16(const uint8_t *p, size_t n);
void     orc_md5(const uint8_t *p, size_t n, uint8_t out[16]);

/* Frame discovery over the bytes that follow Metadata.Read (flacmeta.go:845-849). */
int orc_scan_frames(const uint8_t *data, size_t len, const orc_streaminfo *si,
                    orc_frame *frames, size_t cap, size_t *nframes, orc_stats *stats);

/* Decode one resolved frame into planar int32 (out[ch * blocksize + n]),
 * after decorrelation and wasted-bits restore.  Returns an ORC_FRAME_* code. */
int orc_decode_frame(const uint8_t *frame, size_t len, const orc_streaminfo *si,
                     int32_t *out, size_t out_cap, orc_frame *info);

/* Whole stream: scan, decode, interleave and pack to signed little-endian
 * ceil(bps/8)-byte samples (the byte string the STREAMINFO MD5 hashes).
 * pcm must hold total inter-channel samples * channels * ceil(bps/8) bytes;
 * frames are written at first_sample, so dropped frames leave a gap. */
int orc_decode_stream(const uint8_t *data, size_t len, const orc_streaminfo *si,
                      uint8_t *pcm, size_t pcm_cap, size_t *pcm_len, orc_stats *stats);

/* The same for a frame range of a stream: the sample with inter-channel index sample_base lands at pcm[0]
 * (frames that start before it get ORC_FRAME_RANGE), and the frame table as the decode left it (statuses
 * include what the subframe decoder found) is copied to frames_out when that is not NULL (frames_cap
 * entries; *n_frames = frames found). */
int orc_decode_stream_ex(const uint8_t *data, size_t len, const orc_streaminfo *si,
                         uint8_t *pcm, size_t pcm_cap, uint64_t sample_base,
                         orc_frame *frames_out, size_t frames_cap, size_t *n_frames,
                         size_t *pcm_len, orc_stats *stats);

/* Multi-threaded batch wrapper used for CPU-baseline timing: n independent
 * streams, statically partitioned over nthreads POSIX threads. */
int orc_decode_batch(const uint8_t *data, const uint64_t *offs, const uint64_t *lens,
                     const orc_streaminfo *infos, uint32_t n,
                     uint8_t *pcm, const uint64_t *pcm_offs, const uint64_t *pcm_caps,
                     uint64_t *pcm_lens, orc_stats *stats, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
