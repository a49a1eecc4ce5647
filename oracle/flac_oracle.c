/*
 * flac_oracle.c — CPU oracle for the FLAC frame path.  TEST INFRASTRUCTURE ONLY
 * (see flac_oracle.h).  Nothing in the product library links or calls this file.
 *
 * Rules restated here (the HIP library implements the same rules; the tests
 * compare the two bit for bit):
 *
 *  STREAMINFO   flacmeta.go:644-717.  34 bytes big-endian: u16 min/max block
 *               size, u24 min/max frame size, u20 rate, u3 channels-1,
 *               u5 bps-1, u36 total samples, 16 bytes MD5.  Errors as the
 *               reference: 0 < MinBlockSize < 16 (:688), MaxBlockSize < 16
 *               (:695), SampleRate == 0 or >= 655350 (:708).  The reference
 *               computes MinFrameSize as uint32(bits>>24) (:699), which leaks
 *               the low byte of MaxBlockSize into bits 24-31; the value meant by
 *               the format is the 24-bit field, which is what is returned here
 *               (identical for every block size whose low byte is zero, i.e. for
 *               both reference fixtures).
 *
 *  Frames       RFC 9639 section 9.  A frame candidate is a byte pair 0xFF,
 *               0xF8|0xF9 whose header has no reserved code, a well-formed coded
 *               number, a CRC-8 that matches, the channel count and sample size of
 *               STREAMINFO and a block size <= MaxBlockSize.  Candidates are
 *               resolved in order: candidate i becomes a frame ending at the first
 *               of the next ORC_LOOKAHEAD candidates (or the end of the data) for
 *               which the CRC-16 of the whole span is zero; if none is, it is
 *               recorded with status BAD_CRC16, ends at the next candidate and is
 *               not decoded.
 */
#include "flac_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_LOOKAHEAD 4

/* ------------------------------------------------------------------ CRC --- */

static uint8_t  crc8_tab[256];
static uint16_t crc16_tab[256];
static int      tabs_ready = 0;

static void init_tabs(void)
{
    if (tabs_ready) return;
    for (int i = 0; i < 256; i++) {
        uint8_t c = (uint8_t)i;
        for (int k = 0; k < 8; k++) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
        crc8_tab[i] = c;
        uint16_t d = (uint16_t)(i << 8);
        for (int k = 0; k < 8; k++) d = (uint16_t)((d & 0x8000) ? ((d << 1) ^ 0x8005) : (d << 1));
        crc16_tab[i] = d;
    }
    tabs_ready = 1;
}

uint8_t orc_crc8(const uint8_t *p, size_t n)
{
    init_tabs();
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) c = crc8_tab[c ^ p[i]];
    return c;
}

uint16_t orc_crc16(const uint8_t *p, size_t n)
{
    init_tabs();
    uint16_t c = 0;
    for (size_t i = 0; i < n; i++) c = (uint16_t)((c << 8) ^ crc16_tab[(c >> 8) ^ p[i]]);
    return c;
}

/* ------------------------------------------------------------------ MD5 --- */
/* RFC 1321. */

static const uint32_t md5_k[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
    0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
    0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
    0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
    0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
    0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
    0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t md5_s[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22,
                                  5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

static void md5_block(uint32_t st[4], const uint8_t *p)
{
    uint32_t m[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) |
               ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
    for (int i = 0; i < 64; i++) {
        uint32_t f;
        int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        uint32_t t = a + f + md5_k[i] + m[g];
        a = d; d = c; c = b;
        b = b + ((t << md5_s[i]) | (t >> (32 - md5_s[i])));
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

void orc_md5(const uint8_t *p, size_t n, uint8_t out[16])
{
    uint32_t st[4] = {0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476};
    size_t i = 0;
    for (; i + 64 <= n; i += 64) md5_block(st, p + i);
    uint8_t tail[128];
    size_t r = n - i;
    memset(tail, 0, sizeof tail);
    memcpy(tail, p + i, r);
    tail[r] = 0x80;
    size_t tl = (r < 56) ? 64 : 128;
    uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; k++) tail[tl - 8 + k] = (uint8_t)(bits >> (8 * k));
    md5_block(st, tail);
    if (tl == 128) md5_block(st, tail + 64);
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 4; j++) out[4 * k + j] = (uint8_t)(st[k] >> (8 * j));
}

/* ----------------------------------------------------------- STREAMINFO --- */

int orc_parse_streaminfo(const uint8_t *b, size_t n, orc_streaminfo *out)
{
    if (!b || !out || n < 34) return ORC_ERR_ARG;
    memset(out, 0, sizeof *out);
    out->min_blocksize = (uint16_t)((b[0] << 8) | b[1]);                 /* flacmeta.go:687 */
    if (out->min_blocksize > 0 && out->min_blocksize < 16) return ORC_ERR_STREAMINFO; /* :688 */
    out->max_blocksize = (uint16_t)((b[2] << 8) | b[3]);                 /* :694 */
    if (out->max_blocksize < 16) return ORC_ERR_STREAMINFO;              /* :695 */
    out->min_framesize = ((uint32_t)b[4] << 16) | ((uint32_t)b[5] << 8) | b[6];  /* :699, 24-bit field */
    out->max_framesize = ((uint32_t)b[7] << 16) | ((uint32_t)b[8] << 8) | b[9];  /* :700 */
    uint64_t bits = 0;
    for (int i = 0; i < 8; i++) bits = (bits << 8) | b[10 + i];          /* :702 */
    out->sample_rate = (uint32_t)(bits >> 44);                           /* :707 */
    if (out->sample_rate == 0 || out->sample_rate >= 655350) return ORC_ERR_STREAMINFO; /* :708 */
    out->channels = (uint8_t)(((bits >> 41) & 7) + 1);                   /* :711 */
    out->bits_per_sample = (uint8_t)(((bits >> 36) & 31) + 1);           /* :712 */
    out->total_samples = bits & 0xFFFFFFFFFULL;                          /* :713 */
    memcpy(out->md5, b + 18, 16);                                        /* :714 */
    return ORC_OK;
}

/* --------------------------------------------------------- frame header --- */

/* Returns header length (incl. CRC-8) or 0 when p does not start a plausible frame. */
static int parse_header(const uint8_t *p, size_t avail, const orc_streaminfo *si, orc_frame *f)
{
    if (avail < 6) return 0;
    if (p[0] != 0xFF || (p[1] & 0xFE) != 0xF8) return 0;
    int variable = p[1] & 1;
    int bs_code = p[2] >> 4, sr_code = p[2] & 15;
    int ch = p[3] >> 4, ss_code = (p[3] >> 1) & 7;
    if (p[3] & 1) return 0;
    if (bs_code == 0 || sr_code == 15 || ch > 10 || ss_code == 3) return 0;

    /* coded number, UTF-8 style */
    size_t pos = 4;
    uint8_t x = p[pos++];
    int extra;
    uint64_t num;
    if (x < 0x80) { extra = 0; num = x; }
    else if (x < 0xC0) return 0;
    else if (x < 0xE0) { extra = 1; num = x & 0x1F; }
    else if (x < 0xF0) { extra = 2; num = x & 0x0F; }
    else if (x < 0xF8) { extra = 3; num = x & 0x07; }
    else if (x < 0xFC) { extra = 4; num = x & 0x03; }
    else if (x < 0xFE) { extra = 5; num = x & 0x01; }
    else if (x == 0xFE) { extra = 6; num = 0; }
    else return 0;
    if (extra == 6 && !variable) return 0;
    if (pos + (size_t)extra > avail) return 0;
    for (int i = 0; i < extra; i++) {
        uint8_t c = p[pos++];
        if ((c & 0xC0) != 0x80) return 0;
        num = (num << 6) | (c & 0x3F);
    }

    uint32_t bs;
    if (bs_code == 1) bs = 192;
    else if (bs_code <= 5) bs = 576u << (bs_code - 2);
    else if (bs_code == 6) { if (pos + 1 > avail) return 0; bs = (uint32_t)p[pos] + 1; pos += 1; }
    else if (bs_code == 7) { if (pos + 2 > avail) return 0; bs = (((uint32_t)p[pos] << 8) | p[pos + 1]) + 1; pos += 2; }
    else bs = 256u << (bs_code - 8);

    static const uint32_t rates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    uint32_t sr;
    if (sr_code < 12) sr = sr_code ? rates[sr_code] : si->sample_rate;
    else if (sr_code == 12) { if (pos + 1 > avail) return 0; sr = (uint32_t)p[pos] * 1000; pos += 1; }
    else { if (pos + 2 > avail) return 0; sr = ((uint32_t)p[pos] << 8) | p[pos + 1]; if (sr_code == 14) sr *= 10; pos += 2; }

    if (pos + 1 > avail) return 0;
    if (orc_crc8(p, pos) != p[pos]) return 0;
    pos += 1;

    static const uint8_t sizes[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    int bps = ss_code ? sizes[ss_code] : si->bits_per_sample;
    int nch = (ch < 8) ? ch + 1 : 2;
    if (nch != si->channels) return 0;
    if (bps != si->bits_per_sample) return 0;
    if (bs > si->max_blocksize) return 0;

    f->blocksize = bs;
    f->number = num;
    f->variable = (uint8_t)variable;
    f->ch_assign = (uint8_t)ch;
    f->bps = (uint8_t)bps;
    f->header_len = (uint8_t)pos;
    f->sample_rate = sr;
    f->first_sample = variable ? num : num * (uint64_t)si->max_blocksize;
    return (int)pos;
}

/* ----------------------------------------------------------- frame scan --- */

int orc_scan_frames(const uint8_t *data, size_t len, const orc_streaminfo *si,
                    orc_frame *frames, size_t cap, size_t *nframes, orc_stats *stats)
{
    if (!data || !si || !nframes) return ORC_ERR_ARG;
    init_tabs();
    orc_stats st;
    memset(&st, 0, sizeof st);
    st.first_bad_frame = ~(uint64_t)0;

    /* pass 1: candidates */
    size_t ccap = 1024, nc = 0;
    orc_frame *cand = (orc_frame *)malloc(ccap * sizeof *cand);
    for (size_t p = 0; p + 1 < len; p++) {
        if (data[p] != 0xFF || (data[p + 1] & 0xFE) != 0xF8) continue;
        st.sync_candidates++;
        orc_frame f;
        memset(&f, 0, sizeof f);
        if (!parse_header(data + p, len - p, si, &f)) continue;
        f.offset = p;
        if (nc == ccap) { ccap *= 2; cand = (orc_frame *)realloc(cand, ccap * sizeof *cand); }
        cand[nc++] = f;
    }
    st.header_candidates = nc;

    /* pass 2: resolve by CRC-16 */
    size_t nf = 0, cur = 0;
    int rc = ORC_OK;
    while (cur < nc) {
        size_t found = 0;
        int ok = 0;
        for (size_t j = cur + 1; j <= cur + ORC_LOOKAHEAD && j <= nc; j++) {
            size_t end = (j < nc) ? (size_t)cand[j].offset : len;
            if (orc_crc16(data + cand[cur].offset, end - (size_t)cand[cur].offset) == 0) { found = j; ok = 1; break; }
        }
        orc_frame f = cand[cur];
        size_t next;
        if (ok) { next = found; f.status = ORC_FRAME_OK; }
        else { next = cur + 1; f.status = ORC_FRAME_BAD_CRC16; }
        size_t end = (next < nc) ? (size_t)cand[next].offset : len;
        f.length = (uint32_t)(end - (size_t)f.offset);
        if (frames) {
            if (nf < cap) frames[nf] = f; else rc = ORC_ERR_CAPACITY;
        }
        nf++;
        if (ok) { st.frames_ok++; st.samples += f.blocksize; }
        else if (st.first_bad_frame == ~(uint64_t)0) { st.first_bad_frame = nf - 1; st.first_error = f.status; }
        cur = next;
    }
    st.frames = nf;
    free(cand);
    *nframes = nf;
    if (stats) *stats = st;
    return rc;
}

/* ------------------------------------------------------------ bit reader --- */

typedef struct {
    const uint8_t *buf;
    size_t len, bytepos;
    uint64_t acc;
    int nacc;
    int err;
} br_t;

static inline void br_refill(br_t *b)
{
    while (b->nacc <= 56 && b->bytepos < b->len) {
        b->acc |= (uint64_t)b->buf[b->bytepos++] << (56 - b->nacc);
        b->nacc += 8;
    }
}

static inline uint32_t br_read(br_t *b, int n)
{
    if (n == 0) return 0;
    br_refill(b);
    if (b->nacc < n) { b->err = 1; return 0; }
    uint32_t v = (uint32_t)(b->acc >> (64 - n));
    b->acc <<= n;
    b->nacc -= n;
    return v;
}

static inline int32_t br_read_signed(br_t *b, int n)
{
    if (n == 0) return 0;
    uint32_t v = br_read(b, n);
    if (n < 32 && (v >> (n - 1))) v |= ~0u << n;
    return (int32_t)v;
}

static inline uint32_t br_unary(br_t *b)
{
    uint32_t q = 0;
    for (;;) {
        br_refill(b);
        if (b->nacc == 0) { b->err = 1; return q; }
        if (b->acc == 0) { q += (uint32_t)b->nacc; b->nacc = 0; continue; }
        int z = __builtin_clzll(b->acc);
        q += (uint32_t)z;
        b->acc = (b->acc << z) << 1;   /* two steps: z + 1 may be 64 */
        b->nacc -= z + 1;
        return q;
    }
}

static inline size_t br_bitpos(const br_t *b) { return b->bytepos * 8 - (size_t)b->nacc; }

/* ---------------------------------------------------------- frame decode --- */

static int ilog2_u32(uint32_t v) { return 31 - __builtin_clz(v); }

static int decode_residual(br_t *b, int32_t *s, uint32_t bs, int order)
{
    uint32_t method = br_read(b, 2);
    if (method > 1) return ORC_FRAME_BAD_SUBFRAME;
    int pbits = method ? 5 : 4;
    uint32_t esc = method ? 31 : 15;
    uint32_t po = br_read(b, 4);
    if (b->err) return ORC_FRAME_TRUNCATED;
    if (bs & ((1u << po) - 1)) return ORC_FRAME_BAD_SUBFRAME;
    uint32_t psamp = bs >> po;
    if (psamp < (uint32_t)order) return ORC_FRAME_BAD_SUBFRAME;
    uint32_t n = (uint32_t)order;
    for (uint32_t p = 0; p < (1u << po); p++) {
        uint32_t cnt = psamp - (p == 0 ? (uint32_t)order : 0);
        uint32_t k = br_read(b, pbits);
        if (k == esc) {
            int w = (int)br_read(b, 5);
            for (uint32_t i = 0; i < cnt; i++) s[n++] = br_read_signed(b, w);
        } else {
            for (uint32_t i = 0; i < cnt; i++) {
                uint32_t q = br_unary(b);
                uint32_t u = (q << k) | br_read(b, (int)k);
                s[n++] = (int32_t)((u >> 1) ^ (0u - (u & 1)));
            }
        }
        if (b->err) return ORC_FRAME_TRUNCATED;
    }
    return ORC_FRAME_OK;
}

static int decode_subframe(br_t *b, int32_t *s, uint32_t bs, int bps)
{
    uint32_t hdr = br_read(b, 8);
    if (b->err) return ORC_FRAME_TRUNCATED;
    if (hdr & 0x80) return ORC_FRAME_BAD_SUBFRAME;
    uint32_t type = (hdr >> 1) & 0x3F;
    int wasted = 0;
    if (hdr & 1) {
        wasted = (int)br_unary(b) + 1;
        if (b->err) return ORC_FRAME_TRUNCATED;
    }
    bps -= wasted;
    if (bps < 1) return ORC_FRAME_BAD_SUBFRAME;
    if (bps > 32) return ORC_FRAME_UNSUPPORTED;

    if (type == 0) {
        int32_t v = br_read_signed(b, bps);
        for (uint32_t i = 0; i < bs; i++) s[i] = v;
    } else if (type == 1) {
        for (uint32_t i = 0; i < bs; i++) s[i] = br_read_signed(b, bps);
    } else if (type >= 8 && type <= 12) {
        int order = (int)type - 8;
        if ((uint32_t)order > bs) return ORC_FRAME_BAD_SUBFRAME;
        for (int i = 0; i < order; i++) s[i] = br_read_signed(b, bps);
        int rc = decode_residual(b, s, bs, order);
        if (rc) return rc;
        /* 32-bit wrapping arithmetic: the predictors are linear, so the low 32 bits are exact */
        uint32_t *u = (uint32_t *)s;
        switch (order) {
        case 1: for (uint32_t i = 1; i < bs; i++) u[i] += u[i - 1]; break;
        case 2: for (uint32_t i = 2; i < bs; i++) u[i] += 2 * u[i - 1] - u[i - 2]; break;
        case 3: for (uint32_t i = 3; i < bs; i++) u[i] += 3 * u[i - 1] - 3 * u[i - 2] + u[i - 3]; break;
        case 4: for (uint32_t i = 4; i < bs; i++) u[i] += 4 * u[i - 1] - 6 * u[i - 2] + 4 * u[i - 3] - u[i - 4]; break;
        default: break;
        }
    } else if (type >= 32) {
        int order = (int)type - 31;
        if ((uint32_t)order > bs) return ORC_FRAME_BAD_SUBFRAME;
        for (int i = 0; i < order; i++) s[i] = br_read_signed(b, bps);
        uint32_t pc = br_read(b, 4);
        if (pc == 15) return ORC_FRAME_BAD_SUBFRAME;
        int prec = (int)pc + 1;
        int shift = br_read_signed(b, 5);
        if (shift < 0) return ORC_FRAME_BAD_SUBFRAME;
        int32_t coef[32];
        for (int i = 0; i < order; i++) coef[i] = br_read_signed(b, prec);
        if (b->err) return ORC_FRAME_TRUNCATED;
        int rc = decode_residual(b, s, bs, order);
        if (rc) return rc;
        /* 32-bit accumulation is exact when bps + precision + floor(log2(order)) <= 32;
         * otherwise accumulate in 64 bits.  The HIP library uses the same rule. */
        if (bps + prec + ilog2_u32((uint32_t)order) <= 32) {
            for (uint32_t i = (uint32_t)order; i < bs; i++) {
                uint32_t sum = 0;
                for (int j = 0; j < order; j++) sum += (uint32_t)coef[j] * (uint32_t)s[i - 1 - j];
                s[i] = (int32_t)((uint32_t)s[i] + (uint32_t)((int32_t)sum >> shift));
            }
        } else {
            for (uint32_t i = (uint32_t)order; i < bs; i++) {
                int64_t sum = 0;
                for (int j = 0; j < order; j++) sum += (int64_t)coef[j] * (int64_t)s[i - 1 - j];
                s[i] = (int32_t)((uint32_t)s[i] + (uint32_t)(sum >> shift));
            }
        }
    } else {
        return ORC_FRAME_BAD_SUBFRAME;
    }
    if (b->err) return ORC_FRAME_TRUNCATED;
    /* RFC 9639: a decoded sample must fit the subframe's bit depth; a stream where it does not is invalid */
    if (bps < 32) {
        const int32_t lo = -(int32_t)(1u << (bps - 1)), hi = (int32_t)((1u << (bps - 1)) - 1);
        for (uint32_t i = 0; i < bs; i++) if (s[i] < lo || s[i] > hi) return ORC_FRAME_BAD_SUBFRAME;
    }
    if (wasted) {
        uint32_t *u = (uint32_t *)s;
        for (uint32_t i = 0; i < bs; i++) u[i] <<= wasted;
    }
    return ORC_FRAME_OK;
}

int orc_decode_frame(const uint8_t *frame, size_t len, const orc_streaminfo *si,
                     int32_t *out, size_t out_cap, orc_frame *info)
{
    orc_frame f;
    memset(&f, 0, sizeof f);
    int hl = parse_header(frame, len, si, &f);
    if (!hl) return ORC_FRAME_BAD_SUBFRAME;
    int nch = si->channels;
    uint32_t bs = f.blocksize;
    if ((size_t)nch * bs > out_cap) return ORC_FRAME_RANGE;
    if (info) { uint64_t off = info->offset; *info = f; info->offset = off; info->length = (uint32_t)len; }

    br_t b;
    memset(&b, 0, sizeof b);
    b.buf = frame; b.len = len; b.bytepos = (size_t)hl;
    for (int c = 0; c < nch; c++) {
        int side = (f.ch_assign == 8 && c == 1) || (f.ch_assign == 9 && c == 0) || (f.ch_assign == 10 && c == 1);
        int rc = decode_subframe(&b, out + (size_t)c * bs, bs, f.bps + side);
        if (rc) return rc;
    }
    size_t endbyte = (br_bitpos(&b) + 7) / 8 + 2;
    if (endbyte != len) return ORC_FRAME_LENGTH;

    uint32_t *c0 = (uint32_t *)out, *c1 = (uint32_t *)out + bs;
    if (f.ch_assign == 8) {               /* left/side: R = L - S */
        for (uint32_t i = 0; i < bs; i++) c1[i] = c0[i] - c1[i];
    } else if (f.ch_assign == 9) {        /* side/right: L = S + R */
        for (uint32_t i = 0; i < bs; i++) c0[i] = c0[i] + c1[i];
    } else if (f.ch_assign == 10) {       /* mid/side */
        for (uint32_t i = 0; i < bs; i++) {
            int32_t m = (int32_t)c0[i], sd = (int32_t)c1[i];
            uint32_t mm = ((uint32_t)m << 1) | ((uint32_t)sd & 1);
            c0[i] = (uint32_t)((int32_t)(mm + (uint32_t)sd) >> 1);
            c1[i] = (uint32_t)((int32_t)(mm - (uint32_t)sd) >> 1);
        }
    }
    return ORC_FRAME_OK;
}

/* --------------------------------------------------------- stream decode --- */

int orc_decode_stream(const uint8_t *data, size_t len, const orc_streaminfo *si,
                      uint8_t *pcm, size_t pcm_cap, size_t *pcm_len, orc_stats *stats)
{
    return orc_decode_stream_ex(data, len, si, pcm, pcm_cap, 0, NULL, 0, NULL, pcm_len, stats);
}

int orc_decode_stream_ex(const uint8_t *data, size_t len, const orc_streaminfo *si,
                         uint8_t *pcm, size_t pcm_cap, uint64_t sample_base,
                         orc_frame *frames_out, size_t frames_cap, size_t *n_frames,
                         size_t *pcm_len, orc_stats *stats)
{
    if (!data || !si || !pcm) return ORC_ERR_ARG;
    size_t nf = 0;
    orc_stats st;
    /* a frame is at least 10 bytes (6 header, 1 subframe header, 1 payload, 2 CRC) */
    size_t cap = len / 10 + 16;
    orc_frame *frames = (orc_frame *)malloc(cap * sizeof *frames);
    int rc = orc_scan_frames(data, len, si, frames, cap, &nf, &st);
    if (rc) { free(frames); return rc; }

    int nch = si->channels;
    int bytes = (si->bits_per_sample + 7) / 8;
    int32_t *tmp = (int32_t *)malloc((size_t)nch * si->max_blocksize * sizeof(int32_t));
    size_t maxend = 0;
    st.frames_ok = 0;
    st.samples = 0;
    for (size_t i = 0; i < nf; i++) {
        orc_frame *f = &frames[i];
        if (f->status != ORC_FRAME_OK) continue;
        uint64_t endsample = f->first_sample + f->blocksize;
        if ((si->total_samples && endsample > si->total_samples) || f->first_sample < sample_base ||
            (endsample - sample_base) * (uint64_t)nch * bytes > pcm_cap) { f->status = ORC_FRAME_RANGE; continue; }
        int s = orc_decode_frame(data + f->offset, f->length, si, tmp, (size_t)nch * si->max_blocksize, NULL);
        f->status = s;
        if (s != ORC_FRAME_OK) continue;
        uint8_t *o = pcm + (f->first_sample - sample_base) * (uint64_t)nch * bytes;
        for (uint32_t n = 0; n < f->blocksize; n++)
            for (int c = 0; c < nch; c++) {
                uint32_t v = (uint32_t)tmp[(size_t)c * f->blocksize + n];
                for (int k = 0; k < bytes; k++) *o++ = (uint8_t)(v >> (8 * k));
            }
        size_t e = (size_t)((endsample - sample_base) * (uint64_t)nch * bytes);
        if (e > maxend) maxend = e;
        st.frames_ok++;
        st.samples += f->blocksize;
    }
    st.first_bad_frame = ~(uint64_t)0;
    st.first_error = 0;
    st.pad = 0;
    for (size_t i = 0; i < nf; i++)
        if (frames[i].status != ORC_FRAME_OK) { st.first_bad_frame = i; st.first_error = frames[i].status; break; }
    if (frames_out)
        for (size_t i = 0; i < nf && i < frames_cap; i++) frames_out[i] = frames[i];
    if (n_frames) *n_frames = nf;
    free(tmp);
    free(frames);
    if (pcm_len) *pcm_len = maxend;
    if (stats) *stats = st;
    return ORC_OK;
}

/* ---------------------------------------------------------- batch (CPU) --- */

typedef struct {
    const uint8_t *data; const uint64_t *offs, *lens; const orc_streaminfo *infos;
    uint8_t *pcm; const uint64_t *pcm_offs, *pcm_caps; uint64_t *pcm_lens; orc_stats *stats;
    uint32_t n; int tid, nthreads; int rc;
} batch_arg;

static void *batch_worker(void *p)
{
    batch_arg *a = (batch_arg *)p;
    for (uint32_t i = (uint32_t)a->tid; i < a->n; i += (uint32_t)a->nthreads) {
        size_t pl = 0;
        orc_stats st;
        memset(&st, 0, sizeof st);
        int rc = orc_decode_stream(a->data + a->offs[i], (size_t)a->lens[i], &a->infos[i],
                                   a->pcm + a->pcm_offs[i], (size_t)a->pcm_caps[i], &pl, &st);
        if (rc) a->rc = rc;
        if (a->pcm_lens) a->pcm_lens[i] = pl;
        if (a->stats) a->stats[i] = st;
    }
    return NULL;
}

int orc_decode_batch(const uint8_t *data, const uint64_t *offs, const uint64_t *lens,
                     const orc_streaminfo *infos, uint32_t n,
                     uint8_t *pcm, const uint64_t *pcm_offs, const uint64_t *pcm_caps,
                     uint64_t *pcm_lens, orc_stats *stats, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    init_tabs();
    pthread_t th[256];
    batch_arg args[256];
    for (int t = 0; t < nthreads; t++) {
        batch_arg a = {data, offs, lens, infos, pcm, pcm_offs, pcm_caps, pcm_lens, stats, n, t, nthreads, 0};
        args[t] = a;
        pthread_create(&th[t], NULL, batch_worker, &args[t]);
    }
    int rc = ORC_OK;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); if (args[t].rc) rc = args[t].rc; }
    return rc;
}
