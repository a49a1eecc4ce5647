/* Sanitizer harness (CPU only) of the oracle, oracle/flac_oracle.c and oracle/flac_enc_oracle.c.
 *
 * usage: san_oracle file.flac ...   For every file: a clean decode (must give the STREAMINFO MD5), then
 * decodes of damaged copies (truncations, bit flips, a run of 0xFF, a short output buffer), every input in a
 * heap block of exactly its size; then the decoded PCM through the encoder oracle and back.  Nothing here may
 * trip AddressSanitizer or UndefinedBehaviorSanitizer, whatever the bytes are. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/flac_oracle.h"
#include "../../oracle/flac_enc_oracle.h"

static uint32_t rng_state = 0x1234567u;
static uint32_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

static int decode_copy(const uint8_t *src, size_t n, const orc_streaminfo *si, size_t pcm_cap, orc_stats *st, uint8_t **pcm_out, size_t *pcm_len)
{
    uint8_t *b = malloc(n ? n : 1);
    uint8_t *pcm = malloc(pcm_cap ? pcm_cap : 1);
    memcpy(b, src, n);
    memset(pcm, 0, pcm_cap);
    size_t len = 0;
    const int rc = orc_decode_stream(b, n, si, pcm, pcm_cap, &len, st);
    free(b);
    if (pcm_out) { *pcm_out = pcm; *pcm_len = len; } else free(pcm);
    return rc;
}

int main(int argc, char **argv)
{
    unsigned long runs = 0;
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 2; }
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *d = malloc((size_t)size);
        if (fread(d, 1, (size_t)size, f) != (size_t)size) return 2;
        fclose(f);
        /* metadata walk: marker, blocks until the last one (the harness's own few lines, not under test) */
        size_t pos = 4;
        orc_streaminfo si;
        memset(&si, 0, sizeof si);
        for (;;) {
            const int last = d[pos] >> 7, type = d[pos] & 0x7F;
            const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
            if (type == 0 && orc_parse_streaminfo(d + pos + 4, len, &si) != ORC_OK) { fprintf(stderr, "%s: STREAMINFO\n", argv[a]); return 2; }
            pos += 4 + len;
            if (last) break;
        }
        for (size_t n = 0; n <= 34; ++n) { orc_streaminfo t; uint8_t *b = malloc(n ? n : 1); memcpy(b, d + 8, n); (void)orc_parse_streaminfo(b, n, &t); free(b); }
        const uint8_t *audio = d + pos;
        const size_t alen = (size_t)size - pos;
        const size_t pcm_cap = (size_t)si.total_samples * si.channels * ((si.bits_per_sample + 7) / 8);
        orc_stats st;
        uint8_t *pcm = NULL;
        size_t pcm_len = 0;
        if (decode_copy(audio, alen, &si, pcm_cap, &st, &pcm, &pcm_len) != ORC_OK || st.frames != st.frames_ok) { fprintf(stderr, "%s: clean decode failed\n", argv[a]); return 1; }
        uint8_t md5[16];
        orc_md5(pcm, pcm_len, md5);
        if (memcmp(md5, si.md5, 16) != 0) { fprintf(stderr, "%s: MD5 mismatch\n", argv[a]); return 1; }
        ++runs;
        /* damaged copies */
        uint8_t *w = malloc(alen ? alen : 1);
        const int variants = alen > 200000 ? 12 : 60;
        for (int v = 0; v < variants; ++v) {
            memcpy(w, audio, alen);
            size_t n = alen;
            switch (v % 6) {
            case 0: n = alen ? rng() % alen : 0; break;                                              /* cut anywhere */
            case 1: for (int k = 0; k < 3 && alen; ++k) w[rng() % alen] ^= (uint8_t)(1u << (rng() & 7)); break;
            case 2: if (alen > 64) memset(w + rng() % (alen - 64), 0xFF, 64); break;                 /* run of sync-like bytes */
            case 3: if (alen > 64) memset(w + rng() % (alen - 64), 0x00, 64); break;                 /* endless unary run */
            case 4: if (alen > 16) { const size_t at = rng() % (alen - 16); for (int k = 0; k < 16; ++k) w[at + k] = (uint8_t)rng(); } break;
            case 5: n = alen > 3 ? alen - 1 - rng() % 3 : 0; break;                                  /* cut inside the last frame */
            }
            (void)decode_copy(w, n, &si, v % 5 == 4 ? pcm_cap / 2 : pcm_cap, &st, NULL, NULL);
            orc_frame frames[64];
            size_t nf = 0;
            uint8_t *b = malloc(n ? n : 1);
            memcpy(b, w, n);
            (void)orc_scan_frames(b, n, &si, frames, 64, &nf, &st);
            free(b);
            ++runs;
        }
        free(w);
        /* the PCM through the encoder oracle and back */
        if (si.bits_per_sample <= 24 && si.total_samples > 0) {
            orc_enc_opts o;
            memset(&o, 0, sizeof o);
            o.blocksize = si.max_blocksize > 16384 ? 4096 : si.max_blocksize; o.max_lpc_order = 12; o.max_partition_order = 6; o.stereo_mode = -1;
            const size_t cap = (size_t)orc_encode_bound(si.total_samples, si.channels, si.bits_per_sample, o.blocksize);
            uint8_t *out = malloc(cap);
            size_t out_len = 0;
            uint64_t nfr = 0;
            uint32_t minf = 0, maxf = 0;
            if (orc_encode_stream(pcm, si.total_samples, si.channels, si.bits_per_sample, si.sample_rate, &o, si.md5, out, cap, &out_len,
                                  &nfr, &minf, &maxf) == ORC_OK) {
                orc_streaminfo si2;
                if (orc_parse_streaminfo(out + 8, 34, &si2) != ORC_OK) { fprintf(stderr, "%s: encoder STREAMINFO\n", argv[a]); return 1; }
                uint8_t *pcm2 = NULL;
                size_t len2 = 0;
                if (decode_copy(out + 42, out_len - 42, &si2, pcm_cap, &st, &pcm2, &len2) != ORC_OK || len2 != pcm_len || memcmp(pcm, pcm2, pcm_len) != 0) {
                    fprintf(stderr, "%s: encoder round trip differs\n", argv[a]);
                    return 1;
                }
                free(pcm2);
                ++runs;
            }
            free(out);
        }
        free(pcm);
        free(d);
    }
    printf("san_oracle: %lu runs, no sanitizer report\n", runs);
    return 0;
}
