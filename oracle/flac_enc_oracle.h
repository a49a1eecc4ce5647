/*
 * flac_enc_oracle.h — CPU restatement of libflacgpu's batched FLAC encoder.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker of the HIP encoder; the product
 * never links, loads or calls it.
 *
 * The reference (flacmeta.go) has no encoder, so parity here is unpinned by the
 * reference's own fixtures.  The encoder is pinned by round trips: its output,
 * decoded by the decoder oracle (pinned by the reference's fixtures), must
 * equal the source PCM, and the STREAMINFO block it writes must read back
 * through the restatement of MarshalStreaminfoBlock (flacmeta.go:644-717).
 */
#ifndef FLAC_ENC_ORACLE_H
#define FLAC_ENC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same fields and meaning as flacgpu_encode_options (without struct_size / reserved). */
typedef struct {
    uint32_t blocksize;
    uint32_t max_lpc_order;
    uint32_t qlp_precision;
    uint32_t min_partition_order;
    uint32_t max_partition_order;
    int32_t  stereo_mode;
    uint32_t force_lpc_order;
    uint32_t flags;
    uint32_t seektable_interval;
} orc_enc_opts;

/* Tukey(0.5) analysis window of a block of bs samples, as single-precision values. */
void orc_enc_window(uint32_t bs, float *w);

/* Upper bound of the encoded size of one stream (marker + STREAMINFO + frames). */
uint64_t orc_encode_bound(uint64_t total_samples, int channels, int bps, uint32_t blocksize);

/* Encodes one stream.  pcm: interleaved signed little-endian samples of
 * ceil(bps/8) bytes.  md5: 16 bytes to store in STREAMINFO, or NULL to hash
 * the PCM here.  Writes a complete .flac stream ("fLaC", one STREAMINFO
 * block, frames).  Returns an ORC_* status of flac_oracle.h. */
int orc_encode_stream(const uint8_t *pcm, uint64_t total_samples, int channels, int bps, uint32_t sample_rate,
                      const orc_enc_opts *o, const uint8_t *md5, uint8_t *out, size_t cap, size_t *out_len,
                      uint64_t *n_frames, uint32_t *min_framesize, uint32_t *max_framesize);

#ifdef __cplusplus
}
#endif
#endif
