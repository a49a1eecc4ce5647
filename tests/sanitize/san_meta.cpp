// Sanitizer harness (CPU only) of the metadata restatement, flac_amd/csrc/flacmeta.cpp: the reference's
// parsers index short slices without a length check and panic (flacmeta.go:360, :404, :473, :687-702);
// the restatement must return a status instead and may never read outside the bytes it was given.  Every
// input is copied into a heap block of exactly its size, so AddressSanitizer sees any over-read.
//
// usage: san_meta file.flac ...    For every file: every truncation of its metadata section, single-byte
// damage at every metadata byte (three values each), and every block payload handed to every block parser
// at every truncated length.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/flacgpu.h"

static unsigned long g_calls = 0, g_ok = 0;

static void run_all(const uint8_t *src, size_t n)
{
    uint8_t *b = (uint8_t *)malloc(n ? n : 1);   // exact size: one byte past the end is poisoned
    memcpy(b, src, n);
    flacgpu_metadata md;
    g_ok += flacgpu_parse_metadata(b, n, &md) == FLACGPU_OK; ++g_calls;
    size_t off = 0;
    (void)flacgpu_skip_id3v2(b, n, &off); ++g_calls;
    free(b);
}

static void run_payload(const uint8_t *src, size_t n)
{
    uint8_t *b = (uint8_t *)malloc(n ? n : 1);
    memcpy(b, src, n);
    flacgpu_block blk; flacgpu_streaminfo si; flacgpu_vorbis_comment vc; flacgpu_picture pic; flacgpu_application app;
    flacgpu_cuesheet cs;
    static flacgpu_seekpoint pts[64];
    static uint32_t offs[64], lens[64];
    static flacgpu_cuesheet_track tr[128];
    static flacgpu_cuesheet_index ix[256];
    size_t cnt = 0;
    (void)flacgpu_parse_block_header(b, n, &blk);
    (void)flacgpu_parse_streaminfo(b, n, &si);
    (void)flacgpu_parse_seektable(b, n, pts, 64, &cnt);
    (void)flacgpu_parse_vorbis_comment(b, n, &vc, offs, lens, 64, &cnt);
    (void)flacgpu_parse_picture(b, n, &pic);
    (void)flacgpu_parse_application(b, n, &app);
    (void)flacgpu_parse_cuesheet(b, n, &cs, tr, 128, ix, 256, &cnt);
    (void)flacgpu_parse_cuesheet(b, n, &cs, nullptr, 0, nullptr, 0, &cnt);
    g_calls += 8;
    free(b);
}

int main(int argc, char **argv)
{
    for (int a = 1; a < argc; ++a) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 2; }
        std::vector<uint8_t> d;
        uint8_t buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + got);
        fclose(f);
        flacgpu_metadata md;
        if (flacgpu_parse_metadata(d.data(), d.size(), &md) != FLACGPU_OK) { fprintf(stderr, "%s: does not parse\n", argv[a]); return 2; }
        const size_t meta = (size_t)md.first_frame_offset;
        for (size_t n = 0; n <= meta; ++n) run_all(d.data(), n);                       // every truncation
        std::vector<uint8_t> m(d.begin(), d.begin() + meta);
        const size_t step = meta > 4096 ? 7 : 1;                                         // padding-heavy files: sample the bytes
        for (size_t i = 0; i < meta; i += (i < 1400 ? 1 : step))
            for (uint8_t x : {(uint8_t)0xFF, (uint8_t)0x80, (uint8_t)0x01}) {
                const uint8_t keep = m[i];
                m[i] ^= x;
                run_all(m.data(), meta);
                m[i] = keep;
            }
        for (uint32_t k = 0; k < md.n_blocks; ++k) {                                     // payloads, truncated, to every parser
            const flacgpu_block &blk = md.blocks[k];
            const size_t len = blk.length > 1200 ? 1200 : blk.length;
            for (size_t n = 0; n <= len; ++n) run_payload(d.data() + blk.payload_offset, n);
        }
        // an ID3v2 tag in front, truncated everywhere
        std::vector<uint8_t> t = {'I', 'D', '3', 4, 0, 0x10, 0, 0, 1, 20};
        t.resize(10 + 148 + 10, 0x55);
        t.insert(t.end(), d.begin(), d.begin() + (meta < 300 ? meta : 300));
        for (size_t n = 0; n <= t.size(); ++n) run_all(t.data(), n);
    }
    printf("san_meta: %lu calls, %lu complete parses, no sanitizer report\n", g_calls, g_ok);
    return 0;
}
