/*
 * flacgpu.h — C ABI of libflacgpu, the MI355X (gfx950) FLAC frame engine.
 *
 * The library takes over where the reference Go package stops: after
 * (*Metadata).Read (flacmeta.go:749-850) has consumed the metadata blocks, the
 * reader sits on byte 0 of audio frame 0 (flacmeta.go:845-849).  Everything from
 * that byte to the end of the file is this library's input, together with the
 * values Read produced: the StreaminfoBlock (flacmeta.go:263-273) and,
 * optionally, the SeekpointBlocks (flacmeta.go:256-260).
 *
 * The reference has no FFI of its own; these are the entry points a cgo
 * binding of `package flac` would declare.
 * Plain pointers and sizes only; every function returns an int status
 * (0 = FLACGPU_OK) and never aborts or throws across the boundary — malformed
 * input, which makes the reference panic (it indexes short slices without a
 * length check, e.g. flacmeta.go:687,693,702), yields a status here.
 *
 * Threading: the metadata functions are pure.  A handle owns one GPU's
 * workspace; calls on one handle must be serialised by the caller, distinct
 * handles are independent (one handle per GPU, one host thread or process per
 * handle).
 */
#ifndef FLACGPU_H
#define FLACGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLACGPU_API_VERSION 2
/* The version the library was built with; a binding compares it with the header it was generated from. */
int flacgpu_api_version(void);

/* ----------------------------------------------------------------- status */
enum {
    FLACGPU_OK = 0,
    FLACGPU_ERR_ARG = 1,         /* null pointer, bad size, unsorted streams            */
    FLACGPU_ERR_SIGNATURE = 2,   /* first four bytes are not "fLaC"  (flacmeta.go:759)  */
    FLACGPU_ERR_BLOCK_TYPE = 3,  /* invalid or unknown block type    (flacmeta.go:546)  */
    FLACGPU_ERR_DUPLICATE = 4,   /* two blocks of a once-only kind   (flacmeta.go:783..)*/
    FLACGPU_ERR_TRUNCATED = 5,   /* input ends inside a header or block (:766, :777)    */
    FLACGPU_ERR_STREAMINFO = 6,  /* STREAMINFO bounds violated       (flacmeta.go:688..)*/
    FLACGPU_ERR_SEEKTABLE = 7,   /* length not a multiple of 18      (flacmeta.go:823)  */
    FLACGPU_ERR_CAPACITY = 8,    /* caller's output array too small                     */
    FLACGPU_ERR_DEVICE = 9,      /* HIP runtime error; see flacgpu_last_error           */
    FLACGPU_ERR_NOMEM = 10,
    FLACGPU_ERR_UNSUPPORTED = 11,
    FLACGPU_ERR_CUESHEET = 12,   /* no tracks, track number 0, index offset not a multiple of 588 (flacmeta.go:409, :473, :511) */
    FLACGPU_ERR_APPLICATION = 13 /* data length not a multiple of 8 (flacmeta.go:361) */
};

/* Per-frame status in flacgpu_frame.status. */
enum {
    FLACGPU_FRAME_OK = 0,
    FLACGPU_FRAME_BAD_CRC16 = 1,    /* no span starting at this header has a zero CRC-16 */
    FLACGPU_FRAME_BAD_SUBFRAME = 2, /* reserved subframe type / invalid field            */
    FLACGPU_FRAME_TRUNCATED = 3,    /* bitstream runs past the end of the frame          */
    FLACGPU_FRAME_LENGTH = 4,       /* decoded length differs from the resolved length   */
    FLACGPU_FRAME_RANGE = 5,        /* samples fall outside total_samples / out_capacity */
    FLACGPU_FRAME_UNSUPPORTED = 6,
    FLACGPU_FRAME_SKIPPED = 7       /* header candidate inside another frame's span      */
};

/* -------------------------------------------------- metadata (CPU, pure) */

/* MetadataBlockType as the reference numbers it: FlacSignature takes iota 0
 * (flacmeta.go:28-37), so STREAMINFO = 1 ... PICTURE = 7, INVALID = 127; the
 * wire code is this value minus one (HeaderType, flacmeta.go:142-162). */
enum {
    FLACGPU_BLOCK_STREAMINFO = 1,
    FLACGPU_BLOCK_PADDING = 2,
    FLACGPU_BLOCK_APPLICATION = 3,
    FLACGPU_BLOCK_SEEKTABLE = 4,
    FLACGPU_BLOCK_VORBIS_COMMENT = 5,
    FLACGPU_BLOCK_CUESHEET = 6,
    FLACGPU_BLOCK_PICTURE = 7,
    FLACGPU_BLOCK_INVALID = 127
};

/* StreaminfoBlock, flacmeta.go:263-273 (MD5Signature as 16 raw bytes). */
typedef struct flacgpu_streaminfo {
    uint16_t min_blocksize;
    uint16_t max_blocksize;
    uint32_t min_framesize;
    uint32_t max_framesize;
    uint32_t sample_rate;
    uint8_t  channels;
    uint8_t  bits_per_sample;
    uint64_t total_samples;
    uint8_t  md5[16];
} flacgpu_streaminfo;

/* SeekpointBlock, flacmeta.go:256-260. */
typedef struct flacgpu_seekpoint {
    uint64_t sample_number;   /* 0xFFFFFFFFFFFFFFFF = placeholder */
    uint64_t offset;          /* bytes from frame 0               */
    uint16_t frame_samples;
} flacgpu_seekpoint;

/* MetadataBlockHeader, flacmeta.go:235-240, plus where its payload lies. */
typedef struct flacgpu_block {
    uint32_t type;            /* FLACGPU_BLOCK_*                  */
    uint32_t length;          /* payload bytes                    */
    uint32_t last;            /* 0 / 1                            */
    uint32_t reserved;
    uint64_t payload_offset;  /* from byte 0 of the file          */
} flacgpu_block;

#define FLACGPU_MAX_BLOCKS 128

/* What Metadata.Read establishes that the frame engine needs. */
typedef struct flacgpu_metadata {
    flacgpu_streaminfo streaminfo;
    uint32_t has_streaminfo;
    uint32_t n_blocks;
    flacgpu_block blocks[FLACGPU_MAX_BLOCKS];
    uint64_t first_frame_offset;   /* reader position on return of Read, flacmeta.go:845-849 */
} flacgpu_metadata;

/* MarshalMetadataBlockHeader, flacmeta.go:519-551.  b: 4 bytes. */
int flacgpu_parse_block_header(const uint8_t *b, size_t n, flacgpu_block *out);

/* MarshalStreaminfoBlock, flacmeta.go:644-717.  b: the 34-byte payload. */
int flacgpu_parse_streaminfo(const uint8_t *b, size_t n, flacgpu_streaminfo *out);

/* MarshalSeekpointBlock, flacmeta.go:612-641 (with the length rule of :823). */
int flacgpu_parse_seektable(const uint8_t *b, size_t n, flacgpu_seekpoint *out, size_t cap, size_t *count);

/* (*Metadata).Read, flacmeta.go:749-850: marker, block loop, duplicate rules. */
int flacgpu_parse_metadata(const uint8_t *buf, size_t len, flacgpu_metadata *out);

/* Leading ID3v2 tags.  The reference refuses any file whose first four bytes are not
 * "fLaC" (flacmeta.go:754-761), and so does flacgpu_parse_metadata; files in the wild often
 * carry an ID3v2 tag in front of the marker.  This helper only measures it: *offset = bytes
 * of ID3v2 tag(s) at the start of buf (0 when there is none), so that a caller who wants to
 * accept such files passes buf + *offset on.  A tag is "ID3", two version bytes (neither
 * 0xFF), a flags byte, a 28-bit sync-safe size (four bytes below 0x80), the tag body, and
 * a 10-byte footer when flag 0x10 is set.  FLACGPU_ERR_TRUNCATED when a tag's header is
 * complete but the buffer ends inside the tag. */
int flacgpu_skip_id3v2(const uint8_t *buf, size_t len, size_t *offset);

/* The payload parsers below restate the reference's remaining Marshal* functions.  They
 * allocate nothing: strings and blobs are reported as (offset, length) into the payload the
 * caller passed.  A payload too short for its own fields, which makes the reference panic
 * (it slices without a length check), yields FLACGPU_ERR_TRUNCATED. */

/* VorbisCommentBlock, flacmeta.go:276-280; MarshalVorbisCommentBlock, flacmeta.go:720-746
 * (lengths are little-endian, :736-741).  offsets / lengths: one entry per comment. */
typedef struct flacgpu_vorbis_comment {
    uint32_t vendor_offset, vendor_length;
    uint32_t total_comments;
} flacgpu_vorbis_comment;
int flacgpu_parse_vorbis_comment(const uint8_t *b, size_t n, flacgpu_vorbis_comment *out,
                                 uint32_t *offsets, uint32_t *lengths, size_t cap, size_t *count);

/* PictureBlock, flacmeta.go:243-253; MarshalPictureBlock, flacmeta.go:554-604. */
typedef struct flacgpu_picture {
    uint32_t picture_type;          /* numeric; flacgpu_picture_type_string gives the reference's name */
    uint32_t mime_offset, mime_length;
    uint32_t description_offset, description_length;
    uint32_t width, height, color_depth, num_colors;
    uint32_t length;                /* Length field */
    uint32_t blob_offset, blob_length;   /* blob_length = min(length, bytes left), as buf.Next gives */
} flacgpu_picture;
int flacgpu_parse_picture(const uint8_t *b, size_t n, flacgpu_picture *out);
/* PictureType / PictureTypes, flacmeta.go:117-139, :165-171: "UNKNOWN" above 20. */
const char *flacgpu_picture_type_string(uint32_t picture_type);

/* ApplicationBlock, flacmeta.go:197-200; MarshalApplicationBlock, flacmeta.go:349-366.  The
 * reference takes ApplicationIdLen (a bit count, 32) as a byte count (:360): the id is the
 * first 4 bytes, the data starts at byte 32.  Restated as it behaves. */
typedef struct flacgpu_application {
    uint32_t id;
    uint32_t data_offset, data_length;
} flacgpu_application;
int flacgpu_parse_application(const uint8_t *b, size_t n, flacgpu_application *out);

/* CuesheetBlock / CuesheetTrack / TrackIndex, flacmeta.go:203-232; MarshalCuesheetBlock,
 * MarshalCuesheetTrack, MarshalCuesheetTrackIndex, flacmeta.go:369-516. */
typedef struct flacgpu_cuesheet_index {
    uint64_t sample_offset;
    uint8_t  index_point;
} flacgpu_cuesheet_index;
typedef struct flacgpu_cuesheet_track {
    uint64_t offset;
    uint8_t  number;
    char     isrc[12];
    uint8_t  type;              /* 0 audio, 1 non-audio */
    uint8_t  pre_emphasis;
    uint8_t  index_points;
    uint32_t first_index;       /* position of this track's first index in the index array */
} flacgpu_cuesheet_track;
typedef struct flacgpu_cuesheet {
    char     media_catalog_number[128];   /* NUL padding kept, as the reference keeps it (:403) */
    uint64_t leadin_samples;
    uint8_t  is_compact_disc;
    uint8_t  total_tracks;
} flacgpu_cuesheet;
int flacgpu_parse_cuesheet(const uint8_t *b, size_t n, flacgpu_cuesheet *out,
                           flacgpu_cuesheet_track *tracks, size_t track_cap,
                           flacgpu_cuesheet_index *indexes, size_t index_cap, size_t *n_indexes);

const char *flacgpu_status_string(int status);

/* ------------------------------------------------------ frame engine (GPU) */

typedef struct flacgpu_handle flacgpu_handle;

typedef struct flacgpu_options {
    uint32_t struct_size;     /* sizeof(flacgpu_options)                          */
    int32_t  device;          /* HIP device ordinal                               */
    uint32_t flags;           /* FLACGPU_FLAG_*                                   */
    uint32_t output_format;   /* FLACGPU_OUT_*: how decoded samples are laid out  */
} flacgpu_options;

#define FLACGPU_FLAG_TIMING     1u   /* record HIP events around every kernel stage */
/* After the decode, hash every stream's PCM on the device (one lane per stream) and compare it with
 * StreaminfoBlock.MD5Signature (flacmeta.go:664, :714); the outcome is flacgpu_stream_result.md5_status.
 * Checked are streams with FLACGPU_OUT_PACKED output (the byte string the signature is defined over),
 * a known TotalSamples, a signature that is not all zeros, sample_base 0, and every frame decoded. */
#define FLACGPU_FLAG_VERIFY_MD5 2u

/* Output layouts of the decoder.  PACKED is the default and the only one the MD5 check applies to. */
#define FLACGPU_OUT_PACKED 0u   /* interleaved, ceil(bps / 8) bytes per sample, little-endian, sign-extended */
#define FLACGPU_OUT_S32    1u   /* interleaved, 4 bytes per sample (int32 little-endian) whatever the stream's bps */
#define FLACGPU_OUT_PLANAR 2u   /* one plane per channel: channel c of a stream starts at out_offset +
                                   c * (out_capacity / channels rounded down to a multiple of 4), samples of
                                   ceil(bps / 8) bytes in time order */

#define FLACGPU_MD5_NOT_CHECKED 0u
#define FLACGPU_MD5_MATCH       1u
#define FLACGPU_MD5_MISMATCH    2u

/* One stream of a batch.  All streams of a batch live in ONE device buffer
 * (d_in), in ascending, non-overlapping in_offset order; their PCM goes to ONE
 * device buffer (d_out). */
typedef struct flacgpu_stream {
    uint64_t in_offset;       /* bytes from d_in to byte 0 of frame 0             */
    uint64_t in_length;       /* bytes of audio frames, below 2 GiB per entry (a longer stream is passed as
                                 several entries cut at SEEKTABLE offsets: frames carry their own numbers) */
    uint64_t out_offset;      /* bytes from d_out to the sample with index sample_base */
    uint64_t out_capacity;    /* bytes this stream may write                      */
    uint64_t sample_base;     /* inter-channel index of the sample that lands at out_offset: 0 for a whole stream;
                                 for a frame range cut at a seek point (SeekpointBlock.SampleNumber,
                                 flacmeta.go:256-260) that point's sample number, so that a range decoded on its
                                 own GPU needs an output region for its own samples only.  Frames that start
                                 before it get FLACGPU_FRAME_RANGE. */
    flacgpu_streaminfo info;  /* from flacgpu_parse_metadata / Metadata.Read      */
} flacgpu_stream;

typedef struct flacgpu_stream_result {
    uint64_t header_candidates; /* headers passing validity + CRC-8               */
    uint64_t frames;            /* resolved frames, any status                    */
    uint64_t frames_ok;
    uint64_t samples;           /* inter-channel samples in OK frames             */
    uint64_t pcm_bytes;         /* end of the highest OK frame, bytes from out_offset (PLANAR: within a plane) */
    uint64_t first_bad_frame;   /* index, among the stream's resolved frames in stream order, of the first one whose
                                   status is not FLACGPU_FRAME_OK; UINT64_MAX when there is none.  The reference
                                   reports which thing failed (flacmeta.go:756-788); this is the frame engine's
                                   counterpart, and flacgpu_get_frames has the whole table. */
    int32_t  first_error;       /* that frame's FLACGPU_FRAME_* status, 0 when there is none */
    uint32_t md5_status;        /* FLACGPU_MD5_* (FLACGPU_FLAG_VERIFY_MD5)        */
} flacgpu_stream_result;

/* One resolved frame, as flacgpu_scan_frames reports it. */
typedef struct flacgpu_frame {
    uint64_t offset;          /* bytes from frame 0 of its stream                 */
    uint64_t first_sample;    /* inter-channel sample index                       */
    uint64_t number;          /* coded frame / sample number                      */
    uint32_t length;          /* bytes, sync code through CRC-16                  */
    uint32_t blocksize;
    uint32_t stream;          /* index into the batch                             */
    int32_t  status;          /* FLACGPU_FRAME_*                                  */
    uint8_t  variable;        /* blocking-strategy bit                            */
    uint8_t  ch_assign;       /* 0..10                                            */
    uint8_t  bps;
    uint8_t  header_len;      /* bytes including CRC-8                            */
    uint32_t reserved;
} flacgpu_frame;

/* Device time of the last decode/scan call, milliseconds (FLACGPU_FLAG_TIMING). */
typedef struct flacgpu_timing {
    float scan_ms;       /* sync search + header validation + CRC-8 (candidate mask),
                            running CRC-16 cut at every sync code                       */
    float compact_ms;    /* prefix sum + ordered candidate list                         */
    float crc16_ms;      /* CRC-16 of every candidate span from the scan's records      */
    float resolve_ms;    /* span resolution + range checks                              */
    float skim_ms;       /* subframe start positions                                    */
    float decode_ms;     /* Rice + prediction + decorrelation + PCM packing             */
    float results_ms;    /* per-stream reduction                                        */
    float total_ms;      /* first kernel to last kernel                                 */
    uint64_t candidates;
    uint64_t in_bytes;
    uint64_t out_bytes;
    uint32_t decode_classes;  /* decode kernel classes the batch used: bit (taps == 32) + 2 * arithmetic
                                 (0 = 24-bit multiplies, 1 = 32-bit, 2 = 64-bit sums, 3 = 16-bit pairs) */
    uint32_t crc16_reread_spans;  /* candidate spans whose CRC-16 needed a second read of their bytes (the last
                                     span of every stream, and spans through 32 KiB of input with more than 63
                                     sync codes); every other span's CRC-16 comes out of the scan's single pass */
    uint32_t launches;        /* kernels the call launched (streams are grouped by channel count and output
                                 sample size: one skim and one set of decode launches per distinct shape) */
    uint32_t shape_groups;    /* distinct (channels, output bytes per sample) shapes in the batch */
} flacgpu_timing;

/* Stream ordering (every entry point with a hip_stream argument).  hip_stream is a
 * hipStream_t; all of the call's kernels and copies run on it, and the call returns when
 * they are complete.  NULL selects the handle's own stream, a *blocking* stream: its work
 * is ordered after everything already issued to the legacy default stream (stream 0).
 * Whatever produces d_in (or still writes d_out) must therefore be on hip_stream, on
 * stream 0 when hip_stream is NULL, or synchronised by the caller before the call; work on
 * any other non-blocking stream is not ordered with the library's.
 *
 * Readable extent of d_in: the scan, the skim and the decoder load d_in in whole 16-byte
 * vectors, so d_in must be 16-byte aligned (checked) and readable up to the next multiple
 * of 16 bytes after in_bytes; the bytes past in_bytes are never interpreted. */
int  flacgpu_create(const flacgpu_options *opts, flacgpu_handle **out);
void flacgpu_destroy(flacgpu_handle *h);
const char *flacgpu_last_error(const flacgpu_handle *h);

/* Frame discovery only: fills `frames` (host array of `cap` entries) with the
 * resolved frames of all streams in stream order, and *n_frames.  d_in is a
 * device pointer.  hip_stream: a hipStream_t, or NULL for the handle's own. */
int flacgpu_scan_frames(flacgpu_handle *h, const void *d_in, uint64_t in_bytes,
                        const flacgpu_stream *streams, uint32_t n_streams,
                        flacgpu_frame *frames, uint64_t cap, uint64_t *n_frames,
                        flacgpu_stream_result *results, void *hip_stream);

/* The frame table of the handle's last flacgpu_scan_frames or decode call, in stream order: after a decode
 * the statuses include what the subframe decoder found (FLACGPU_FRAME_BAD_SUBFRAME, _TRUNCATED, _LENGTH,
 * _RANGE), so a caller who sees frames_ok < frames learns which frames failed and why.  frames: host array
 * of cap entries (NULL with cap 0 just counts); FLACGPU_ERR_CAPACITY when cap is too small, with *n_frames
 * set to the number needed.  After flacgpu_decode_batch (host buffers, decoded in chunks of streams) the table
 * is that of the last chunk; flacgpu_frame.stream then counts from that chunk's first stream. */
int flacgpu_get_frames(flacgpu_handle *h, flacgpu_frame *frames, uint64_t cap, uint64_t *n_frames);

/* Decode a device-resident batch: frame scan, Rice/LPC reconstruction,
 * decorrelation, and packing to interleaved signed little-endian samples of
 * ceil(bps/8) bytes (the byte string the STREAMINFO MD5 is defined over; other
 * layouts: flacgpu_options.output_format), frame i of a stream written at
 * (first_sample - sample_base) * channels * bytes.  Frames whose status is not
 * OK leave their part of d_out unspecified.  Returns when the work is complete. */
int flacgpu_decode_batch_device(flacgpu_handle *h, const void *d_in, uint64_t in_bytes,
                                void *d_out, uint64_t out_bytes,
                                const flacgpu_stream *streams, uint32_t n_streams,
                                flacgpu_stream_result *results, void *hip_stream);

/* The same decode for callers that cannot allocate device memory.  It has no counterpart in
 * the reference (which has no device at all): it exists for tight serving loops and harnesses
 * that forbid hipMalloc.  It is NOT capturable into a HIP graph: like the handle entry point it
 * reads two counts back to the host in the middle (frame header candidates, per-stream ranges)
 * and synchronises hip_stream for them and again for the results.  No handle: the workspace is
 * a static device array of about 3.5 MiB that every GPU the library runs on holds until the
 * process ends, so the call allocates and frees nothing (its first call on a device looks
 * the array's address up, builds 2.6 KB of CRC tables in it and synchronises).  Output is
 * FLACGPU_OUT_PACKED, no timing, no MD5 check.  Limits, checked, with
 * FLACGPU_ERR_CAPACITY beyond them: in_bytes <= FLACGPU_NOALLOC_MAX_IN_BYTES, n_streams <=
 * FLACGPU_NOALLOC_MAX_STREAMS, and at most FLACGPU_NOALLOC_MAX_CANDIDATES frame header
 * candidates in the batch (frames plus false sync codes that pass CRC-8).  `device`: HIP
 * device ordinal.  hip_stream: a hipStream_t; NULL is the legacy default stream.  Results
 * equal those of flacgpu_decode_batch_device.  Every call on a GPU uses that GPU's one
 * workspace: calls that overlap on one GPU race and must be serialised by the caller. */
#define FLACGPU_NOALLOC_MAX_IN_BYTES   (4u << 20)
#define FLACGPU_NOALLOC_MAX_STREAMS    256u
#define FLACGPU_NOALLOC_MAX_CANDIDATES 32768u
int flacgpu_decode_batch_device_noalloc(int device, const void *d_in, uint64_t in_bytes,
                                        void *d_out, uint64_t out_bytes,
                                        const flacgpu_stream *streams, uint32_t n_streams,
                                        flacgpu_stream_result *results, void *hip_stream);
/* Text of the last error of the entry point above on `device`. */
const char *flacgpu_noalloc_last_error(int device);

/* Host-buffer convenience: bufs[i] points at frame 0 of stream i (lens[i]
 * bytes); PCM lands in out_pcm[i] (out_caps[i] bytes).  Copies in and out over
 * PCIe around flacgpu_decode_batch_device. */
int flacgpu_decode_batch(flacgpu_handle *h, const uint8_t *const *bufs, const uint64_t *lens,
                         const flacgpu_streaminfo *infos, uint32_t n_streams,
                         uint8_t *const *out_pcm, const uint64_t *out_caps,
                         flacgpu_stream_result *results);

int flacgpu_get_timing(const flacgpu_handle *h, flacgpu_timing *out);

/* ------------------------------------------------------------ encoder (GPU) */

/* The reference has no encoder; these entry points are what a cgo binding of
 * `package flac` would declare for the write side.  What they must agree with
 * in the reference is the layout of what they write in front of the frames:
 * the "fLaC" marker (flacmeta.go:754-761), one block header (flacmeta.go:519-551,
 * last = 1, type STREAMINFO, length 34) and the STREAMINFO block exactly as
 * MarshalStreaminfoBlock reads it (flacmeta.go:644-717), so that
 * (*Metadata).Read on the output leaves the reader on frame 0. */

typedef struct flacgpu_encode_options {
    uint32_t struct_size;          /* sizeof(flacgpu_encode_options)                               */
    uint32_t blocksize;            /* inter-channel samples per frame, 16..16384                   */
    uint32_t max_lpc_order;        /* 0 (fixed predictors only) .. 32                              */
    uint32_t qlp_precision;        /* bits per LPC coefficient 5..15; 0 = by sample size and block */
    uint32_t min_partition_order;  /* Rice partition order range, 0..8; min == max forces it       */
    uint32_t max_partition_order;
    int32_t  stereo_mode;          /* FLACGPU_STEREO_* or the pair of every frame: 0 left/right, 1 left/side,
                                      2 side/right, 3 mid/side                                     */
    uint32_t force_lpc_order;      /* 0 = estimate the order per subframe; else every predicted subframe is LPC of
                                      exactly this order (constant, and verbatim where smaller, still apply) */
    uint32_t flags;                /* FLACGPU_ENC_*                                                */
    uint32_t seektable_interval;   /* a seek point every this many frames (SEEKTABLE block after STREAMINFO,
                                      SeekpointBlock flacmeta.go:256-260); 0 = no seek table       */
} flacgpu_encode_options;

#define FLACGPU_ENC_MD5 1u   /* hash the PCM on the device for STREAMINFO (one lane per stream) */

/* Per-frame choice of the stereo pair.  EXACT plans all four of left, right, mid and side and takes the
 * pair whose subframes are smallest.  ESTIMATE picks the pair first, from the prediction error of short
 * linear predictors (orders 1..4) of the four signals, and plans its two channels only: about half the
 * analysis work for output a few tenths of a percent larger. */
#define FLACGPU_STEREO_EXACT    (-1)
#define FLACGPU_STEREO_ESTIMATE (-2)

/* One stream to encode.  All PCM of a batch lives in ONE device buffer (d_pcm) as
 * interleaved signed little-endian samples of ceil(bps/8) bytes: the byte string
 * the STREAMINFO MD5 is defined over and the decoder writes.  in_offset must be a
 * multiple of 4.  The complete .flac stream goes to d_out + out_offset. */
typedef struct flacgpu_pcm_stream {
    uint64_t in_offset;
    uint64_t total_samples;    /* inter-channel samples (StreaminfoBlock.TotalSamples, flacmeta.go:662) */
    uint64_t out_offset;       /* multiple of 4 */
    uint64_t out_capacity;     /* see flacgpu_encode_bound */
    uint32_t sample_rate;
    uint8_t  channels;         /* 1..8 */
    uint8_t  bits_per_sample;  /* 4..24 */
    uint8_t  reserved[2];
    uint8_t  md5[16];          /* stored in STREAMINFO unless FLACGPU_ENC_MD5 is set; zeros = not computed */
} flacgpu_pcm_stream;

typedef struct flacgpu_encode_result {
    uint64_t out_length;       /* bytes of the .flac stream (metadata + frames), also when it did not fit */
    uint64_t frames;
    uint32_t min_framesize;
    uint32_t max_framesize;
    int32_t  status;           /* FLACGPU_OK, or FLACGPU_ERR_CAPACITY: nothing usable was written  */
    uint32_t reserved;
    uint8_t  md5[16];          /* as stored in STREAMINFO */
} flacgpu_encode_result;

/* Device time of the last encode call, milliseconds (FLACGPU_FLAG_TIMING). */
typedef struct flacgpu_encode_timing {
    float stats_ms;      /* stereo pair estimate; per candidate channel: fixed-predictor errors, autocorrelation */
    float design_ms;     /* Levinson-Durbin, order choice, coefficient quantisation   */
    float choose_ms;     /* residuals, Rice partition search, cheapest subframe       */
    float layout_ms;     /* stereo decisions, frame sizes and offsets, headers        */
    float pack_ms;       /* residual coding and bit packing                           */
    float crc_ms;        /* CRC-16 of every frame                                     */
    float md5_ms;        /* runs beside the other stages on a second stream           */
    float total_ms;      /* first kernel to last kernel of the main stream            */
    uint64_t frames;
    uint64_t in_bytes;   /* PCM bytes encoded                                         */
    uint64_t out_bytes;  /* .flac bytes produced                                      */
    float pick_ms;       /* the part of stats_ms spent on the stereo pair estimate    */
    float reserved;
} flacgpu_encode_timing;

/* Largest possible size of one encoded stream without a seek table: a capacity
 * that always fits.  With seektable_interval set, add flacgpu_seektable_bytes. */
uint64_t flacgpu_encode_bound(uint64_t total_samples, uint32_t channels, uint32_t bits_per_sample, uint32_t blocksize);

/* Bytes of the SEEKTABLE block (header + 18 per point) the encoder writes. */
uint64_t flacgpu_seektable_bytes(uint64_t total_samples, uint32_t blocksize, uint32_t seektable_interval);

/* Encodes a device-resident batch into complete .flac streams (marker,
 * STREAMINFO, fixed-blocksize frames).  d_pcm / d_out: device pointers, 16-byte
 * aligned, each readable up to the next multiple of 16 bytes.  Returns when the
 * work is complete; a stream that does not fit its capacity gets
 * FLACGPU_ERR_CAPACITY in its result and does not fail the call. */
int flacgpu_encode_batch_device(flacgpu_handle *h, const void *d_pcm, uint64_t pcm_bytes,
                                void *d_out, uint64_t out_bytes,
                                const flacgpu_pcm_stream *streams, uint32_t n_streams,
                                const flacgpu_encode_options *opts,
                                flacgpu_encode_result *results, void *hip_stream);

/* Host-buffer convenience: pcm[i] -> out[i] (out_caps[i] bytes), copies over PCIe
 * around flacgpu_encode_batch_device.  streams[i].in_offset / out_offset are ignored. */
int flacgpu_encode_batch(flacgpu_handle *h, const uint8_t *const *pcm, const flacgpu_pcm_stream *streams,
                         uint32_t n_streams, const flacgpu_encode_options *opts,
                         uint8_t *const *out, const uint64_t *out_caps, flacgpu_encode_result *results);

int flacgpu_get_encode_timing(const flacgpu_handle *h, flacgpu_encode_timing *out);

/* MD5 of n byte ranges of a device buffer, one GPU lane per range: the check of
 * decoded PCM against StreaminfoBlock.MD5Signature (flacmeta.go:664, :714)
 * without copying the PCM to the host.  offsets / lengths: host arrays, bytes
 * from d_buf; digests: host array of 16 * n bytes.  MD5 is a serial chain, so
 * this pays off for many ranges (one per stream), not for one long range. */
int flacgpu_md5_batch_device(flacgpu_handle *h, const void *d_buf, uint64_t buf_bytes,
                             const uint64_t *offsets, const uint64_t *lengths, uint32_t n,
                             uint8_t *digests, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
