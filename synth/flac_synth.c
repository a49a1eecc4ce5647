/*
 * flac_synth.c — deterministic PCM synthesis and a scalar CPU FLAC encoder
 * (see flac_synth.h).  Produces inputs for tests and benchmarks only.
 *
 * The encoder writes streams that follow RFC 9639: fixed predictors of order
 * 0-4, LPC up to order 32 (Tukey-windowed autocorrelation, Levinson-Durbin,
 * coefficient quantisation), partitioned Rice coding with 4- or 5-bit
 * parameters and optional escape partitions, wasted-bits detection, CONSTANT
 * and VERBATIM subframes, the three stereo decorrelation modes, fixed or
 * variable blocking, and a STREAMINFO block carrying the MD5 of the packed PCM.
 * Residuals are computed with the same accumulator-width rule the decoders use
 * (32-bit when bps + precision + floor(log2(order)) <= 32, else 64-bit), so an
 * encode -> decode round trip is exact by construction.
 */
#include "flac_synth.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ checksums --- */

static uint8_t  t8[256];
static uint16_t t16[256];
static int      t_ready = 0;

static void tables(void)
{
    if (t_ready) return;
    for (int i = 0; i < 256; i++) {
        unsigned a = (unsigned)i, b = (unsigned)i << 8;
        for (int k = 0; k < 8; k++) {
            a = (a & 0x80) ? ((a << 1) ^ 0x107) : (a << 1);
            b = (b & 0x8000) ? ((b << 1) ^ 0x18005) : (b << 1);
        }
        t8[i] = (uint8_t)a;
        t16[i] = (uint16_t)b;
    }
    t_ready = 1;
}

static uint8_t sum8(const uint8_t *p, size_t n)
{
    uint8_t c = 0;
    while (n--) c = t8[c ^ *p++];
    return c;
}

static uint16_t sum16(const uint8_t *p, size_t n)
{
    uint16_t c = 0;
    while (n--) c = (uint16_t)((c << 8) ^ t16[(c >> 8) ^ *p++]);
    return c;
}

/* Streaming MD5 (RFC 1321), constants generated from sines as the RFC defines them. */
typedef struct { uint32_t h[4]; uint8_t buf[64]; size_t fill; uint64_t total; } md5_ctx;
static uint32_t md5_T[64];
static int md5_T_ready = 0;

static void md5_init(md5_ctx *c)
{
    if (!md5_T_ready) {
        for (int i = 0; i < 64; i++) md5_T[i] = (uint32_t)(fabs(sin((double)(i + 1))) * 4294967296.0);
        md5_T_ready = 1;
    }
    c->h[0] = 0x67452301; c->h[1] = 0xefcdab89; c->h[2] = 0x98badcfe; c->h[3] = 0x10325476;
    c->fill = 0; c->total = 0;
}

#define ROL(x, s) (((x) << (s)) | ((x) >> (32 - (s))))
static void md5_compress(md5_ctx *c, const uint8_t *p)
{
    static const int rot[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
    uint32_t w[16];
    memcpy(w, p, 64); /* little-endian host */
    uint32_t a = c->h[0], b = c->h[1], cc = c->h[2], d = c->h[3];
    for (int i = 0; i < 64; i++) {
        int r = i >> 4;
        uint32_t f; int g;
        switch (r) {
        case 0: f = d ^ (b & (cc ^ d)); g = i; break;
        case 1: f = cc ^ (d & (b ^ cc)); g = (5 * i + 1) & 15; break;
        case 2: f = b ^ cc ^ d; g = (3 * i + 5) & 15; break;
        default: f = cc ^ (b | ~d); g = (7 * i) & 15; break;
        }
        uint32_t t = a + f + md5_T[i] + w[g];
        a = d; d = cc; cc = b; b += ROL(t, rot[r][i & 3]);
    }
    c->h[0] += a; c->h[1] += b; c->h[2] += cc; c->h[3] += d;
}

static void md5_update(md5_ctx *c, const uint8_t *p, size_t n)
{
    c->total += n;
    if (c->fill) {
        size_t take = 64 - c->fill; if (take > n) take = n;
        memcpy(c->buf + c->fill, p, take); c->fill += take; p += take; n -= take;
        if (c->fill == 64) { md5_compress(c, c->buf); c->fill = 0; }
    }
    while (n >= 64) { md5_compress(c, p); p += 64; n -= 64; }
    if (n) { memcpy(c->buf, p, n); c->fill = n; }
}

static void md5_final(md5_ctx *c, uint8_t out[16])
{
    uint64_t bits = c->total * 8;
    uint8_t pad[72] = {0x80};
    size_t padlen = (c->fill < 56) ? 56 - c->fill : 120 - c->fill;
    md5_update(c, pad, padlen);
    uint8_t lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
    md5_update(c, lenb, 8);
    memcpy(out, c->h, 16);
}

/* ----------------------------------------------------------- bit writer --- */

typedef struct { uint8_t *p; size_t cap, pos; uint64_t acc; int n; int ovf; } bw_t;

static inline void bw_put(bw_t *w, uint32_t v, int bits)
{
    if (bits <= 0) return;
    if (bits < 32) v &= (1u << bits) - 1;
    w->acc = (w->acc << bits) | v;
    w->n += bits;
    while (w->n >= 8) {
        uint8_t byte = (uint8_t)(w->acc >> (w->n - 8));
        if (w->pos < w->cap) w->p[w->pos++] = byte; else w->ovf = 1;
        w->n -= 8;
    }
}

static inline void bw_unary(bw_t *w, uint32_t q)
{
    while (q >= 32) { bw_put(w, 0, 32); q -= 32; }
    bw_put(w, 1, (int)q + 1);
}

static inline void bw_align(bw_t *w) { if (w->n) bw_put(w, 0, 8 - w->n); }

/* ------------------------------------------------------------- options --- */

void synth_default_opts(synth_enc_opts *o, uint32_t blocksize, uint32_t max_lpc_order)
{
    memset(o, 0, sizeof *o);
    o->blocksize = blocksize;
    o->max_lpc_order = max_lpc_order;
    o->min_partition_order = 0;
    o->max_partition_order = 6;
    o->force_lpc_order = 0;
    o->force_partition_order = -1;
    o->stereo_mode = -1;
    o->force_fixed_order = -1;
    o->rice_method = -1;
    o->escape_mode = 0;
}

size_t synth_encode_bound(uint64_t nsamples, int channels, int bps, uint32_t blocksize)
{
    uint64_t frames = (nsamples + blocksize - 1) / blocksize + 1;
    uint64_t raw = nsamples * (uint64_t)channels * (uint64_t)((bps + 8) / 8 + 1);
    return (size_t)(raw + frames * (64 + 16 * (uint64_t)channels) + 8192 + frames * 18);
}

/* ------------------------------------------------------- subframe model --- */

typedef struct {
    int type;            /* 0 constant, 1 verbatim, 2 fixed, 3 lpc */
    int order, wasted, bps;   /* bps after removing wasted bits */
    int prec, shift;
    int32_t coef[32];
    int method, po;
    uint64_t bits;
} sf_head;

typedef struct {
    sf_head h;
    int32_t *x;        /* samples with wasted bits removed, bs entries */
    int32_t *res;      /* residual, bs entries (first `order` unused)  */
    uint8_t *k;        /* Rice parameter per partition                 */
    uint8_t *escw;     /* 0xFF = Rice coded, else escape width         */
} sf_plan;

static int ilog2u(uint32_t v) { return 31 - __builtin_clz(v); }

static inline int sbits(int32_t v) /* bits to hold v as two's complement */
{
    uint32_t u = (uint32_t)(v < 0 ? ~v : v);
    return u ? 33 - __builtin_clz(u) : 1;
}

/* Rice parameter search over partition orders.  Returns estimated bits and fills po/k/escw/method. */
static uint64_t rice_search(const synth_enc_opts *o, const int32_t *res, uint32_t bs, int order, sf_plan *pl)
{
    uint32_t min_po = o->min_partition_order, max_po = o->max_partition_order;
    if (o->force_partition_order >= 0) min_po = max_po = (uint32_t)o->force_partition_order;
    if (max_po > 15) max_po = 15;
    /* the block must divide evenly and the first partition must hold the warm-up */
    while (max_po > 0 && ((bs & ((1u << max_po) - 1)) || (bs >> max_po) < (uint32_t)order + (o->force_partition_order >= 0 ? 0 : 1))) max_po--;
    if (min_po > max_po) min_po = max_po;

    int kmax = (o->rice_method == 0) ? 14 : 30;
    uint64_t best = ~0ULL;
    uint32_t nparts_max = 1u << max_po;
    uint64_t *sums = (uint64_t *)malloc(nparts_max * sizeof(uint64_t));
    int32_t *mx = (int32_t *)malloc(nparts_max * sizeof(int32_t));
    uint8_t *kk = (uint8_t *)malloc(nparts_max);
    uint8_t *ee = (uint8_t *)malloc(nparts_max);

    for (uint32_t po = max_po + 1; po-- > min_po;) {
        uint32_t np = 1u << po, ps = bs >> po;
        if (po == max_po) {
            for (uint32_t p = 0; p < np; p++) {
                uint32_t a = p * ps + (p == 0 ? (uint32_t)order : 0), b = (p + 1) * ps;
                uint64_t s = 0; int mb = 0;
                for (uint32_t i = a; i < b; i++) {
                    int32_t r = res[i];
                    s += ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
                    int sb = sbits(r); if (sb > mb) mb = sb;
                }
                sums[p] = s; mx[p] = mb;
            }
        } else {
            for (uint32_t p = 0; p < np; p++) {
                sums[p] = sums[2 * p] + sums[2 * p + 1];
                mx[p] = mx[2 * p] > mx[2 * p + 1] ? mx[2 * p] : mx[2 * p + 1];
            }
        }
        uint64_t total = 0; int need5 = 0;
        for (uint32_t p = 0; p < np; p++) {
            uint32_t cnt = ps - (p == 0 ? (uint32_t)order : 0);
            uint64_t bb = ~0ULL; int bk = 0;
            for (int k = 0; k <= kmax; k++) {
                uint64_t b = (uint64_t)cnt * (uint64_t)(k + 1) + (sums[p] >> k);
                if (b < bb) { bb = b; bk = k; }
            }
            kk[p] = (uint8_t)bk; ee[p] = 0xFF;
            int ew = cnt ? mx[p] : 0;
            uint64_t eb = 5 + (uint64_t)cnt * (uint64_t)ew;
            if (ew <= 31 && (o->escape_mode == 2 || (o->escape_mode == 1 && eb < bb))) { ee[p] = (uint8_t)ew; bb = eb; }
            if (bk > 14) need5 = 1;
            total += bb;
        }
        int method = (o->rice_method == 1 || need5) ? 1 : 0;
        total += (uint64_t)np * (method ? 5 : 4);
        if (total < best) {
            best = total; pl->h.po = (int)po; pl->h.method = method;
            memcpy(pl->k, kk, np); memcpy(pl->escw, ee, np);
        }
    }
    free(sums); free(mx); free(kk); free(ee);
    return best + 6;
}

static void fixed_residual(const int32_t *x, uint32_t bs, int order, int32_t *res)
{
    const uint32_t *u = (const uint32_t *)x;
    uint32_t *r = (uint32_t *)res;
    switch (order) {
    case 0: memcpy(res, x, bs * sizeof(int32_t)); break;
    case 1: for (uint32_t i = 1; i < bs; i++) r[i] = u[i] - u[i - 1]; break;
    case 2: for (uint32_t i = 2; i < bs; i++) r[i] = u[i] - 2 * u[i - 1] + u[i - 2]; break;
    case 3: for (uint32_t i = 3; i < bs; i++) r[i] = u[i] - 3 * u[i - 1] + 3 * u[i - 2] - u[i - 3]; break;
    default: for (uint32_t i = 4; i < bs; i++) r[i] = u[i] - 4 * u[i - 1] + 6 * u[i - 2] - 4 * u[i - 3] + u[i - 4]; break;
    }
}

/* Returns 0 when some residual does not fit the Rice value range. */
static int lpc_residual(const int32_t *x, uint32_t bs, int order, const int32_t *coef, int shift, int wide, int32_t *res)
{
    for (uint32_t i = (uint32_t)order; i < bs; i++) {
        int64_t pred;
        if (!wide) {
            uint32_t s = 0;
            for (int j = 0; j < order; j++) s += (uint32_t)coef[j] * (uint32_t)x[i - 1 - j];
            pred = (int32_t)s >> shift;
        } else {
            int64_t s = 0;
            for (int j = 0; j < order; j++) s += (int64_t)coef[j] * (int64_t)x[i - 1 - j];
            pred = s >> shift;
        }
        int64_t r = (int64_t)x[i] - pred;
        if (r <= -2147483647LL - 1 || r > 2147483647LL) return 0;
        res[i] = (int32_t)r;
    }
    return 1;
}

static int default_precision(int bps, uint32_t bs)
{
    if (bps < 16) { int p = 2 + bps / 2; return p < 5 ? 5 : p; }
    if (bps == 16) {
        return bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
    }
    return bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
}

/* Plans the cheapest encoding of one channel signal. */
static void plan_subframe(const synth_enc_opts *o, const int32_t *sig, uint32_t bs, int bps, int stream_bps,
                          sf_plan *pl, double *win, int32_t *scratch)
{
    sf_head *h = &pl->h;
    memset(h, 0, sizeof *h);
    /* wasted bits */
    uint32_t ored = 0;
    for (uint32_t i = 0; i < bs; i++) ored |= (uint32_t)sig[i];
    int wasted = ored ? __builtin_ctz(ored) : 0;
    if (wasted >= bps) wasted = bps - 1;
    h->wasted = wasted; h->bps = bps - wasted;
    for (uint32_t i = 0; i < bs; i++) pl->x[i] = sig[i] >> wasted;
    const int32_t *x = pl->x;
    int sb = h->bps;

    int constant = 1;
    for (uint32_t i = 1; i < bs; i++) if (x[i] != x[0]) { constant = 0; break; }
    if (constant && !o->no_constant && !o->force_verbatim && o->force_fixed_order < 0 && o->force_lpc_order <= 0) {
        h->type = 0; h->bits = 8 + (uint64_t)sb + (wasted ? (uint64_t)wasted : 0);
        return;
    }
    uint64_t verb_bits = 8 + (uint64_t)bs * (uint64_t)sb;
    if (o->force_verbatim) { h->type = 1; h->bits = verb_bits; return; }

    /* fixed predictors: total absolute error of orders 0..4 */
    uint64_t best_bits = ~0ULL;
    int tried = 0;
    if (o->force_lpc_order <= 0 || bs <= (uint32_t)o->force_lpc_order) {
        int fo;
        if (o->force_fixed_order >= 0) fo = o->force_fixed_order;
        else {
            uint64_t e[5] = {0, 0, 0, 0, 0};
            int32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
            for (uint32_t i = 0; i < bs; i++) {
                int32_t d0 = x[i], d1 = d0 - l0, d2 = d1 - l1, d3 = d2 - l2, d4 = d3 - l3;
                if (i >= 4) {
                    e[0] += (uint64_t)llabs((long long)d0); e[1] += (uint64_t)llabs((long long)d1);
                    e[2] += (uint64_t)llabs((long long)d2); e[3] += (uint64_t)llabs((long long)d3);
                    e[4] += (uint64_t)llabs((long long)d4);
                }
                l0 = d0; l1 = d1; l2 = d2; l3 = d3;
            }
            fo = 0;
            for (int k = 1; k <= 4; k++) if (e[k] < e[fo]) fo = k;
        }
        if ((uint32_t)fo > bs) fo = (int)bs;
        if (fo > 4) fo = 4;
        fixed_residual(x, bs, fo, pl->res);
        h->order = fo;
        uint64_t rb = rice_search(o, pl->res, bs, fo, pl);
        best_bits = 8 + (uint64_t)fo * (uint64_t)sb + rb;
        h->type = 2; h->order = fo; h->bits = best_bits;
        tried = 1;
    }

    /* LPC */
    uint32_t maxo = o->max_lpc_order;
    if (o->force_lpc_order > 0) maxo = (uint32_t)o->force_lpc_order;
    if (maxo > 32) maxo = 32;
    if (maxo > 0 && bs > maxo && o->force_fixed_order < 0) {
        double ac[33], lp[33][32], err[33];
        /* windowed signal in doubles, autocorrelation lags 0..maxo */
        double *xd = (double *)malloc(bs * sizeof(double));
        for (uint32_t i = 0; i < bs; i++) xd[i] = (double)x[i] * win[i];
        for (uint32_t lag = 0; lag <= maxo; lag++) {
            double s = 0;
            for (uint32_t i = lag; i < bs; i++) s += xd[i] * xd[i - lag];
            ac[lag] = s;
        }
        free(xd);
        int chosen = 0;
        if (ac[0] > 0) {
            /* Levinson-Durbin */
            double e = ac[0], tmp[32];
            uint32_t usable = 0;
            for (uint32_t i = 0; i < maxo; i++) {
                double r = -ac[i + 1];
                for (uint32_t j = 0; j < i; j++) r -= tmp[j] * ac[i - j];
                if (e <= 0) break;
                r /= e;
                tmp[i] = r;
                for (uint32_t j = 0; j < i / 2; j++) {
                    double t = tmp[j];
                    tmp[j] += r * tmp[i - 1 - j];
                    tmp[i - 1 - j] += r * t;
                }
                if (i & 1) tmp[i / 2] += tmp[i / 2] * r;
                e *= (1.0 - r * r);
                for (uint32_t j = 0; j <= i; j++) lp[i + 1][j] = -tmp[j];
                err[i + 1] = e;
                usable = i + 1;
            }
            if (usable) {
                int prec = o->qlp_precision ? (int)o->qlp_precision : default_precision(stream_bps, bs);
                if (prec > 15) prec = 15;
                if (prec < 5) prec = 5;
                uint32_t ord;
                if (o->force_lpc_order > 0) ord = usable < (uint32_t)o->force_lpc_order ? 0 : (uint32_t)o->force_lpc_order;
                else {
                    double scale = 0.5 / (M_LN2 * M_LN2 * (double)bs), bestb = 1e300;
                    ord = 1;
                    for (uint32_t k = 1; k <= usable; k++) {
                        double v = err[k] * scale, bpr = v > 0 ? 0.5 * log2(v) : 0.0;
                        if (bpr < 0) bpr = 0;
                        double tot = bpr * (double)(bs - k) + (double)k * (double)(sb + prec);
                        if (tot < bestb) { bestb = tot; ord = k; }
                    }
                }
                if (ord) {
                    /* subframes of up to 17 bits keep a precision that decodes with 32-bit sums */
                    if (sb <= 17 && !o->qlp_precision) {
                        int cap = 32 - sb - ilog2u(ord);
                        if (prec > cap) prec = cap;
                    }
                    /* quantise */
                    double cmax = 0;
                    for (uint32_t j = 0; j < ord; j++) { double a = fabs(lp[ord][j]); if (a > cmax) cmax = a; }
                    if (cmax > 0) {
                        int ex; (void)frexp(cmax, &ex);
                        int shift = prec - ex - 1;
                        if (shift > 15) shift = 15;
                        int32_t q[32];
                        int qmax = (1 << (prec - 1)) - 1, qmin = -qmax - 1;
                        double carry = 0;
                        if (shift >= 0) {
                            for (uint32_t j = 0; j < ord; j++) {
                                carry += lp[ord][j] * (double)(1 << shift);
                                long v = lround(carry);
                                if (v > qmax) v = qmax;
                                if (v < qmin) v = qmin;
                                carry -= (double)v; q[j] = (int32_t)v;
                            }
                        } else {
                            for (uint32_t j = 0; j < ord; j++) {
                                carry += lp[ord][j] / (double)(1 << -shift);
                                long v = lround(carry);
                                if (v > qmax) v = qmax;
                                if (v < qmin) v = qmin;
                                carry -= (double)v; q[j] = (int32_t)v;
                            }
                            shift = 0;
                        }
                        int wide = (sb + prec + ilog2u(ord)) > 32;
                        if (lpc_residual(x, bs, (int)ord, q, shift, wide, scratch)) {
                            sf_plan alt = *pl;
                            uint8_t *ak = (uint8_t *)malloc(2u << 15);
                            alt.k = ak; alt.escw = ak + (1u << 15);
                            uint64_t rb = rice_search(o, scratch, bs, (int)ord, &alt);
                            uint64_t lb = 8 + (uint64_t)ord * (uint64_t)sb + 9 + (uint64_t)ord * (uint64_t)prec + rb;
                            if (lb < best_bits || o->force_lpc_order > 0 || !tried) {
                                h->type = 3; h->order = (int)ord; h->prec = prec; h->shift = shift;
                                memcpy(h->coef, q, ord * sizeof(int32_t));
                                h->po = alt.h.po; h->method = alt.h.method;
                                memcpy(pl->k, alt.k, 1u << alt.h.po);
                                memcpy(pl->escw, alt.escw, 1u << alt.h.po);
                                memcpy(pl->res, scratch, bs * sizeof(int32_t));
                                best_bits = lb; h->bits = lb; chosen = 1;
                            }
                            free(ak);
                        }
                    }
                }
            }
        }
        if (!chosen && !tried) {
            /* forced LPC impossible (silent block): fall back to fixed order 0 */
            fixed_residual(x, bs, 0, pl->res);
            uint64_t rb = rice_search(o, pl->res, bs, 0, pl);
            h->type = 2; h->order = 0; h->bits = best_bits = 8 + rb;
        }
    }
    if (verb_bits < best_bits && o->force_lpc_order <= 0 && o->force_fixed_order < 0) { h->type = 1; h->bits = verb_bits; }
}

static void write_subframe(bw_t *w, const sf_plan *pl, uint32_t bs)
{
    const sf_head *h = &pl->h;
    int typecode = h->type == 0 ? 0 : h->type == 1 ? 1 : h->type == 2 ? 8 + h->order : 31 + h->order;
    bw_put(w, 0, 1);
    bw_put(w, (uint32_t)typecode, 6);
    bw_put(w, h->wasted ? 1 : 0, 1);
    if (h->wasted) bw_unary(w, (uint32_t)h->wasted - 1);
    int sb = h->bps;
    if (h->type == 0) { bw_put(w, (uint32_t)pl->x[0], sb); return; }
    if (h->type == 1) { for (uint32_t i = 0; i < bs; i++) bw_put(w, (uint32_t)pl->x[i], sb); return; }
    for (int i = 0; i < h->order; i++) bw_put(w, (uint32_t)pl->x[i], sb);
    if (h->type == 3) {
        bw_put(w, (uint32_t)h->prec - 1, 4);
        bw_put(w, (uint32_t)h->shift, 5);
        for (int i = 0; i < h->order; i++) bw_put(w, (uint32_t)h->coef[i], h->prec);
    }
    bw_put(w, (uint32_t)h->method, 2);
    bw_put(w, (uint32_t)h->po, 4);
    uint32_t np = 1u << h->po, ps = bs >> h->po;
    int pb = h->method ? 5 : 4;
    for (uint32_t p = 0; p < np; p++) {
        uint32_t a = p * ps + (p == 0 ? (uint32_t)h->order : 0), b = (p + 1) * ps;
        if (pl->escw[p] != 0xFF) {
            bw_put(w, h->method ? 31 : 15, pb);
            bw_put(w, pl->escw[p], 5);
            for (uint32_t i = a; i < b; i++) bw_put(w, (uint32_t)pl->res[i], pl->escw[p]);
        } else {
            int k = pl->k[p];
            bw_put(w, (uint32_t)k, pb);
            for (uint32_t i = a; i < b; i++) {
                int32_t r = pl->res[i];
                uint32_t u = ((uint32_t)r << 1) ^ (uint32_t)(r >> 31);
                bw_unary(w, u >> k);
                bw_put(w, u, k);
            }
        }
    }
}

static int utf8_put(uint8_t *p, uint64_t v)
{
    if (v < 0x80) { p[0] = (uint8_t)v; return 1; }
    int n = v < 0x800 ? 2 : v < 0x10000 ? 3 : v < 0x200000 ? 4 : v < 0x4000000 ? 5 : v < 0x80000000ULL ? 6 : 7;
    static const uint8_t lead[8] = {0, 0, 0xC0, 0xE0, 0xF0, 0xF8, 0xFC, 0xFE};
    for (int i = n - 1; i > 0; i--) { p[i] = (uint8_t)(0x80 | (v & 0x3F)); v >>= 6; }
    p[0] = (uint8_t)(lead[n] | v);
    return n;
}

static void sf_plan_alloc(sf_plan *p, uint32_t bs)
{
    p->x = (int32_t *)malloc(bs * sizeof(int32_t));
    p->res = (int32_t *)malloc(bs * sizeof(int32_t));
    p->k = (uint8_t *)malloc(2u << 15);
    p->escw = p->k + (1u << 15);
}
static void sf_plan_free(sf_plan *p) { free(p->x); free(p->res); free(p->k); }

int synth_encode(const int32_t *pcm, uint64_t nsamples, int channels, int bps, uint32_t sample_rate,
                 const synth_enc_opts *opts, uint8_t *out, size_t cap, size_t *out_len,
                 size_t *first_frame_offset)
{
    if (!pcm || !opts || !out || channels < 1 || channels > 8 || bps < 4 || bps > 32) return 1;
    if (opts->blocksize < 16 || opts->blocksize > 65535) return 1;
    tables();
    const synth_enc_opts *o = opts;
    uint32_t maxbs = o->blocksize, minbs = o->blocksize;
    if (o->variable_blocksize)
        for (int i = 0; i < 4; i++) if (o->var_sizes[i]) {
            if (o->var_sizes[i] > maxbs) maxbs = o->var_sizes[i];
            if (o->var_sizes[i] < minbs) minbs = o->var_sizes[i];
        }

    /* metadata */
    uint64_t est_frames = nsamples / minbs + 2;
    uint32_t nseek = o->seek_every_frames ? (uint32_t)((est_frames + o->seek_every_frames - 1) / o->seek_every_frames) : 0;
    size_t pos = 0;
    if (cap < 42 + 4 + (size_t)nseek * 18) return 2;
    memcpy(out, "fLaC", 4); pos = 4;
    out[pos] = nseek ? 0x00 : 0x80; out[pos + 1] = 0; out[pos + 2] = 0; out[pos + 3] = 34; pos += 4;
    size_t si_pos = pos; pos += 34;
    size_t seek_pos = 0;
    if (nseek) {
        uint32_t l = nseek * 18;
        out[pos] = 0x83; out[pos + 1] = (uint8_t)(l >> 16); out[pos + 2] = (uint8_t)(l >> 8); out[pos + 3] = (uint8_t)l; pos += 4;
        seek_pos = pos;
        for (uint32_t i = 0; i < nseek; i++) { memset(out + pos, 0xFF, 8); memset(out + pos + 8, 0, 10); pos += 18; }
    }
    size_t audio0 = pos;
    if (first_frame_offset) *first_frame_offset = audio0;

    /* working storage */
    int nsig = channels == 2 ? 4 : channels;
    int32_t **sig = (int32_t **)malloc((size_t)nsig * sizeof(int32_t *));
    sf_plan *plans = (sf_plan *)malloc((size_t)nsig * sizeof(sf_plan));
    for (int i = 0; i < nsig; i++) { sig[i] = (int32_t *)malloc(maxbs * sizeof(int32_t)); sf_plan_alloc(&plans[i], maxbs); }
    int32_t *scratch = (int32_t *)malloc(maxbs * sizeof(int32_t));
    double *win = (double *)malloc(maxbs * sizeof(double));
    uint32_t win_bs = 0;

    md5_ctx md; md5_init(&md);
    int bytes = (bps + 7) / 8;
    uint8_t *packbuf = (uint8_t *)malloc((size_t)maxbs * (size_t)channels * (size_t)bytes);

    uint32_t min_fs = 0xFFFFFF, max_fs = 0;
    uint64_t done = 0, frame_no = 0;
    uint32_t seek_written = 0;
    int rc = 0;
    static const uint32_t std_rates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};

    while (done < nsamples) {
        uint32_t bs = o->variable_blocksize && o->var_sizes[frame_no & 3] ? o->var_sizes[frame_no & 3] : o->blocksize;
        if ((uint64_t)bs > nsamples - done) bs = (uint32_t)(nsamples - done);
        const int32_t *src = pcm + done * (uint64_t)channels;

        /* MD5 over the packed little-endian samples */
        {
            uint8_t *q = packbuf;
            for (uint32_t i = 0; i < bs * (uint32_t)channels; i++) {
                uint32_t v = (uint32_t)src[i];
                for (int k = 0; k < bytes; k++) *q++ = (uint8_t)(v >> (8 * k));
            }
            md5_update(&md, packbuf, (size_t)(q - packbuf));
        }

        if (bs != win_bs) {
            /* Tukey window, half of it tapered */
            uint32_t np = bs / 4;
            for (uint32_t i = 0; i < bs; i++) win[i] = 1.0;
            for (uint32_t i = 0; i < np; i++) {
                double v = 0.5 - 0.5 * cos(M_PI * (double)(i + 1) / (double)(np + 1));
                win[i] = v; win[bs - 1 - i] = v;
            }
            win_bs = bs;
        }

        int ch_assign;
        const sf_plan *use[8];
        if (channels == 2) {
            for (uint32_t i = 0; i < bs; i++) {
                int32_t l = src[2 * i], r = src[2 * i + 1];
                sig[0][i] = l; sig[1][i] = r;
                sig[2][i] = (int32_t)(((int64_t)l + (int64_t)r) >> 1);
                sig[3][i] = (int32_t)((int64_t)l - (int64_t)r);
            }
            int mode = o->stereo_mode;
            if (bps == 32 && mode != 0) mode = 0; /* side would need 33 bits */
            int need[4] = {mode < 0 || mode == 0 || mode == 1, mode < 0 || mode == 0 || mode == 2,
                           mode < 0 || mode == 3, mode < 0 || mode == 1 || mode == 2 || mode == 3};
            for (int s = 0; s < 4; s++)
                if (need[s]) plan_subframe(o, sig[s], bs, bps + (s == 3), bps, &plans[s], win, scratch);
            if (mode < 0) {
                uint64_t b0 = plans[0].h.bits + plans[1].h.bits, b1 = plans[0].h.bits + plans[3].h.bits;
                uint64_t b2 = plans[3].h.bits + plans[1].h.bits, b3 = plans[2].h.bits + plans[3].h.bits;
                mode = 0; uint64_t bb = b0;
                if (b1 < bb) { bb = b1; mode = 1; }
                if (b2 < bb) { bb = b2; mode = 2; }
                if (b3 < bb) { bb = b3; mode = 3; }
            }
            switch (mode) {
            case 1: ch_assign = 8; use[0] = &plans[0]; use[1] = &plans[3]; break;
            case 2: ch_assign = 9; use[0] = &plans[3]; use[1] = &plans[1]; break;
            case 3: ch_assign = 10; use[0] = &plans[2]; use[1] = &plans[3]; break;
            default: ch_assign = 1; use[0] = &plans[0]; use[1] = &plans[1]; break;
            }
        } else {
            for (int c = 0; c < channels; c++) {
                for (uint32_t i = 0; i < bs; i++) sig[c][i] = src[(size_t)i * (size_t)channels + (size_t)c];
                plan_subframe(o, sig[c], bs, bps, bps, &plans[c], win, scratch);
                use[c] = &plans[c];
            }
            ch_assign = channels - 1;
        }

        /* frame header */
        size_t fstart = pos;
        if (cap - pos < 32) { rc = 2; break; }
        uint8_t *hp = out + pos;
        int hn = 0;
        hp[hn++] = 0xFF;
        hp[hn++] = (uint8_t)(0xF8 | (o->variable_blocksize ? 1 : 0));
        int bs_code, bs_extra = 0;
        if (bs == 192) bs_code = 1;
        else if (bs == 576 || bs == 1152 || bs == 2304 || bs == 4608) bs_code = 2 + ilog2u(bs / 576);
        else if (bs >= 256 && bs <= 32768 && (bs & (bs - 1)) == 0) bs_code = 8 + ilog2u(bs / 256);
        else if (bs <= 256) { bs_code = 6; bs_extra = 1; }
        else { bs_code = 7; bs_extra = 2; }
        int sr_code = 0;
        for (int i = 1; i < 12; i++) if (std_rates[i] == sample_rate) sr_code = i;
        hp[hn++] = (uint8_t)((bs_code << 4) | sr_code);
        int ss_code = bps == 8 ? 1 : bps == 12 ? 2 : bps == 16 ? 4 : bps == 20 ? 5 : bps == 24 ? 6 : bps == 32 ? 7 : 0;
        hp[hn++] = (uint8_t)((ch_assign << 4) | (ss_code << 1));
        hn += utf8_put(hp + hn, o->variable_blocksize ? done : frame_no);
        if (bs_extra == 1) hp[hn++] = (uint8_t)(bs - 1);
        if (bs_extra == 2) { hp[hn++] = (uint8_t)((bs - 1) >> 8); hp[hn++] = (uint8_t)(bs - 1); }
        hp[hn] = sum8(hp, (size_t)hn); hn++;
        pos += (size_t)hn;

        bw_t w = {out, cap, pos, 0, 0, 0};
        for (int c = 0; c < channels; c++) write_subframe(&w, use[c], bs);
        bw_align(&w);
        if (w.ovf || w.pos + 2 > cap) { rc = 2; break; }
        pos = w.pos;
        uint16_t c16 = sum16(out + fstart, pos - fstart);
        out[pos++] = (uint8_t)(c16 >> 8); out[pos++] = (uint8_t)c16;

        uint32_t fs = (uint32_t)(pos - fstart);
        if (fs < min_fs) min_fs = fs;
        if (fs > max_fs) max_fs = fs;
        if (nseek && frame_no % o->seek_every_frames == 0 && seek_written < nseek) {
            uint8_t *sp = out + seek_pos + (size_t)seek_written * 18;
            uint64_t off = fstart - audio0;
            for (int k = 0; k < 8; k++) { sp[k] = (uint8_t)(done >> (56 - 8 * k)); sp[8 + k] = (uint8_t)(off >> (56 - 8 * k)); }
            sp[16] = (uint8_t)(bs >> 8); sp[17] = (uint8_t)bs;
            seek_written++;
        }
        done += bs; frame_no++;
    }

    /* STREAMINFO */
    uint8_t *s = out + si_pos;
    uint32_t smin = o->variable_blocksize ? minbs : o->blocksize, smax = maxbs;
    if (min_fs == 0xFFFFFF) min_fs = 0;
    s[0] = (uint8_t)(smin >> 8); s[1] = (uint8_t)smin; s[2] = (uint8_t)(smax >> 8); s[3] = (uint8_t)smax;
    s[4] = (uint8_t)(min_fs >> 16); s[5] = (uint8_t)(min_fs >> 8); s[6] = (uint8_t)min_fs;
    s[7] = (uint8_t)(max_fs >> 16); s[8] = (uint8_t)(max_fs >> 8); s[9] = (uint8_t)max_fs;
    uint64_t v = ((uint64_t)sample_rate << 44) | ((uint64_t)(channels - 1) << 41) | ((uint64_t)(bps - 1) << 36) | (nsamples & 0xFFFFFFFFFULL);
    for (int k = 0; k < 8; k++) s[10 + k] = (uint8_t)(v >> (56 - 8 * k));
    md5_final(&md, s + 18);

    for (int i = 0; i < nsig; i++) { free(sig[i]); sf_plan_free(&plans[i]); }
    free(sig); free(plans); free(scratch); free(win); free(packbuf);
    if (out_len) *out_len = pos;
    return rc;
}

/* --------------------------------------------------------- PCM synthesis --- */

typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r)
{
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static inline double rng_uni(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline double rng_gauss(rng_t *r)
{
    /* sum of four uniforms, variance 1 */
    double s = rng_uni(r) + rng_uni(r) + rng_uni(r) + rng_uni(r) - 2.0;
    return s * 1.7320508075688772;
}

typedef struct {
    int ntones;
    double re[5], im[5], cr[5], ci[5], amp[5];
    double y1, y2, sigma;
} source_t;

static void source_init(source_t *s, rng_t *r, double fs, double full)
{
    s->ntones = 3 + (int)(rng_next(r) % 3);
    double total = 0;
    for (int i = 0; i < s->ntones; i++) {
        double f = 50.0 + rng_uni(r) * 7950.0;
        double db = -20.0 + rng_uni(r) * 8.0;
        double ph = rng_uni(r) * 2.0 * M_PI, wv = 2.0 * M_PI * f / fs;
        s->amp[i] = pow(10.0, db / 20.0);
        total += s->amp[i];
        s->re[i] = cos(ph); s->im[i] = sin(ph); s->cr[i] = cos(wv); s->ci[i] = sin(wv);
    }
    double norm = total > 0.7 ? 0.7 / total : 1.0;
    for (int i = 0; i < s->ntones; i++) s->amp[i] *= norm * full;
    s->y1 = s->y2 = 0;
    s->sigma = 0.01 * full / 2.31;
}

static inline double source_step(source_t *s, rng_t *r)
{
    double v = 0;
    for (int i = 0; i < s->ntones; i++) {
        double re = s->re[i] * s->cr[i] - s->im[i] * s->ci[i];
        double im = s->re[i] * s->ci[i] + s->im[i] * s->cr[i];
        s->re[i] = re; s->im[i] = im;
        v += s->amp[i] * im;
    }
    double y = 1.3 * s->y1 - 0.5 * s->y2 + s->sigma * rng_gauss(r);
    s->y2 = s->y1; s->y1 = y;
    return v + y;
}

void synth_pcm(uint32_t seed, uint32_t file_index, uint64_t nsamples, int channels, int bps,
               uint32_t sample_rate, uint32_t blocksize, int32_t *out)
{
    rng_t r = {((uint64_t)seed + file_index) * 0x9E3779B97F4A7C15ULL + 0x1234567ULL};
    if (!r.s) r.s = 1;
    for (int i = 0; i < 8; i++) (void)rng_next(&r);
    double full = (double)(1u << (bps - 1));
    source_t src[9];
    for (int c = 0; c <= channels; c++) source_init(&src[c], &r, (double)sample_rate, full);
    double wc = channels > 1 ? 0.894427191 : 0.0, wi = channels > 1 ? 0.447213595 : 1.0;
    int32_t lim = (int32_t)(full - 1);
    int32_t base[8], slope[8];
    for (uint64_t n = 0; n < nsamples; n++) {
        uint64_t blk = n / blocksize;
        uint32_t k = (uint32_t)(n % blocksize);
        if ((blk & 7) == 7) {
            /* piecewise-linear block: exact for a second-order fixed predictor */
            if (k == 0)
                for (int c = 0; c < channels; c++) {
                    base[c] = (int32_t)((rng_uni(&r) - 0.5) * full * 0.5);
                    slope[c] = (int32_t)(rng_next(&r) % 7) - 3;
                }
            for (int c = 0; c < channels; c++) {
                uint32_t half = blocksize / 2;
                int64_t v = base[c] + (int64_t)slope[c] * (k < half ? (int64_t)k : (int64_t)(2 * half - k));
                if (v > lim) v = lim;
                if (v < -lim) v = -lim;
                out[n * (uint64_t)channels + (uint64_t)c] = (int32_t)v;
            }
            continue;
        }
        double common = channels > 1 ? source_step(&src[channels], &r) : 0.0;
        for (int c = 0; c < channels; c++) {
            double v = wc * common + wi * source_step(&src[c], &r);
            long q = lrint(v);
            if (q > lim) q = lim;
            if (q < -lim) q = -lim;
            out[n * (uint64_t)channels + (uint64_t)c] = (int32_t)q;
        }
    }
}

/* ---------------------------------------------------------------- batch --- */

typedef struct {
    uint32_t seed, first, nfiles; uint64_t nsamples; int channels, bps; uint32_t rate;
    const synth_enc_opts *opts; uint8_t *out; size_t slot; uint64_t *lens, *ff; uint8_t *md5s;
    int tid, nthreads, rc;
} mk_arg;

static void *mk_worker(void *p)
{
    mk_arg *a = (mk_arg *)p;
    int32_t *pcm = (int32_t *)malloc(a->nsamples * (uint64_t)a->channels * sizeof(int32_t));
    for (uint32_t i = (uint32_t)a->tid; i < a->nfiles; i += (uint32_t)a->nthreads) {
        synth_pcm(a->seed, a->first + i, a->nsamples, a->channels, a->bps, a->rate, a->opts->blocksize, pcm);
        size_t len = 0, ff = 0;
        int rc = synth_encode(pcm, a->nsamples, a->channels, a->bps, a->rate, a->opts,
                              a->out + (size_t)i * a->slot, a->slot, &len, &ff);
        if (rc) a->rc = rc;
        a->lens[i] = len; a->ff[i] = ff;
        if (a->md5s) memcpy(a->md5s + 16 * (size_t)i, a->out + (size_t)i * a->slot + 8 + 18, 16);
    }
    free(pcm);
    return NULL;
}

int synth_make_batch(uint32_t seed, uint32_t first_index, uint32_t nfiles, uint64_t nsamples,
                     int channels, int bps, uint32_t sample_rate, const synth_enc_opts *opts,
                     uint8_t *out, size_t slot, uint64_t *out_lens, uint64_t *first_frame,
                     uint8_t *md5s, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    tables();
    { md5_ctx c; md5_init(&c); }
    pthread_t th[256];
    mk_arg args[256];
    for (int t = 0; t < nthreads; t++) {
        mk_arg a = {seed, first_index, nfiles, nsamples, channels, bps, sample_rate, opts, out, slot,
                    out_lens, first_frame, md5s, t, nthreads, 0};
        args[t] = a;
        pthread_create(&th[t], NULL, mk_worker, &args[t]);
    }
    int rc = 0;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); if (args[t].rc) rc = args[t].rc; }
    return rc;
}
