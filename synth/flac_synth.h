/*
 * flac_synth.h — synthetic-corpus generator: deterministic PCM signals and a
 * scalar CPU FLAC encoder that turns them into complete .flac byte strings.
 *
 * The reference package (mtw00/flac) has no encoder and no FLAC tool exists in
 * the build image, so every stream beyond the two reference fixtures has to be
 * manufactured here.  This code only produces INPUTS for tests and benchmarks;
 * it is not part of the decode path and the HIP library does not link it.
 */
#ifndef FLAC_SYNTH_H
#define FLAC_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t blocksize;            /* samples per frame (16..65535)                       */
    uint32_t max_lpc_order;        /* 0 = fixed predictors only, <= 32                    */
    uint32_t qlp_precision;        /* 0 = pick from bps/blocksize, else 5..15             */
    uint32_t min_partition_order;  /* Rice partition order search range                   */
    uint32_t max_partition_order;
    int32_t  force_lpc_order;      /* >0: every subframe is LPC of exactly this order     */
    int32_t  force_partition_order;/* >=0: exactly this Rice partition order              */
    int32_t  stereo_mode;          /* -1 best of four, 0 L/R, 1 L/S, 2 S/R, 3 M/S         */
    int32_t  force_verbatim;       /* 1: every subframe VERBATIM                          */
    int32_t  force_fixed_order;    /* >=0: every subframe FIXED of this order             */
    int32_t  rice_method;          /* -1 auto (5-bit only when needed), 0, 1              */
    int32_t  escape_mode;          /* 0 never, 1 when cheaper, 2 every partition          */
    int32_t  variable_blocksize;   /* 1: variable blocking strategy, sizes cycle below    */
    uint32_t var_sizes[4];         /* block sizes used in turn when variable_blocksize    */
    uint32_t seek_every_frames;    /* >0: emit a SEEKTABLE with a point every N frames    */
    int32_t  no_constant;          /* 1: never emit CONSTANT subframes                    */
} synth_enc_opts;

void   synth_default_opts(synth_enc_opts *o, uint32_t blocksize, uint32_t max_lpc_order);
size_t synth_encode_bound(uint64_t nsamples, int channels, int bps, uint32_t blocksize);

/* pcm: interleaved int32, nsamples inter-channel samples.  Writes a complete
 * file (marker, STREAMINFO, optional SEEKTABLE, frames).  Returns 0 on success. */
int synth_encode(const int32_t *pcm, uint64_t nsamples, int channels, int bps, uint32_t sample_rate,
                 const synth_enc_opts *opts, uint8_t *out, size_t cap, size_t *out_len,
                 size_t *first_frame_offset);

/* Deterministic test signal, seeded by (seed, file_index): per channel a sum of
 * 3-5 sinusoids (50..8000 Hz, -20..-12 dBFS) plus AR(2)-coloured noise at
 * about -40 dBFS, channels correlated; every 8th block of `blocksize` samples is
 * a low-order polynomial so that fixed predictors win there. */
void synth_pcm(uint32_t seed, uint32_t file_index, uint64_t nsamples, int channels, int bps,
               uint32_t sample_rate, uint32_t blocksize, int32_t *out);

/* nfiles files (indices first_index..), each synthesised and encoded; file i is
 * written at out + i * slot and its length stored in out_lens[i] and the offset
 * of frame 0 in first_frame[i].  Work is spread over nthreads POSIX threads.
 * md5s (optional) receives 16 bytes per file. */
int synth_make_batch(uint32_t seed, uint32_t first_index, uint32_t nfiles, uint64_t nsamples,
                     int channels, int bps, uint32_t sample_rate, const synth_enc_opts *opts,
                     uint8_t *out, size_t slot, uint64_t *out_lens, uint64_t *first_frame,
                     uint8_t *md5s, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
