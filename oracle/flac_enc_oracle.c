/*
 * flac_enc_oracle.c — CPU restatement of the batched FLAC encoder of libflacgpu.
 * TEST INFRASTRUCTURE ONLY (see flac_enc_oracle.h): the product never links,
 * loads or calls it.
 *
 * The reference Go package (flacmeta.go) has no encoder; what it fixes for this
 * path is the STREAMINFO layout the encoder must write (flacmeta.go:644-717,
 * field widths at :81-96) and the marker / block header in front of it
 * (flacmeta.go:519-551, :754-761).  The frame format is RFC 9639.  Parity of the
 * encoder is therefore "unpinned by the reference"; it is pinned instead by
 * (1) decoding its output with the decoder oracle (flac_oracle.c, itself pinned
 * by the reference's fixtures) back to the source PCM, and (2) the STREAMINFO
 * fields read back through the restatement of MarshalStreaminfoBlock.
 *
 * The code below states the encoder's decisions in the exact arithmetic order
 * the HIP kernels use, so that the two produce identical bytes:
 *   - a block of bs samples is cut into runs of 16 consecutive samples, one per
 *     "lane" t = 0 .. ceil(bs/16)-1; 64 consecutive lanes are a "wave";
 *   - floating-point sums are formed per lane in ascending sample order with
 *     fused multiply-adds, then over the lanes of a wave by a butterfly
 *     (partners at distance 1, 2, 4, 8, 16, 32), then over waves in ascending
 *     order; no other floating-point operation is contracted;
 *   - Rice partitions are whole runs, so a partition order needs bs to be a
 *     multiple of 16 << order.
 */
#include "flac_enc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "flac_oracle.h"

#define RUN 16           /* samples per lane */
#define MAX_BS 16384
#define MAX_LANES (MAX_BS / RUN)

/* ------------------------------------------------------------ bit writer --- */

typedef struct { uint8_t *p; size_t cap, pos; uint64_t acc; int n; int ovf; } bw_t;

static void bw_put(bw_t *w, uint32_t v, int bits)
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

static void bw_zeros(bw_t *w, uint64_t n)
{
    while (n >= 32) { bw_put(w, 0, 32); n -= 32; }
    bw_put(w, 0, (int)n);
}

/* ---------------------------------------------------------------- helpers --- */

static int ilog2_32(uint32_t v) { return 31 - __builtin_clz(v); }
static int ilog2_64(uint64_t v) { return 63 - __builtin_clzll(v); }

/* log2(v) for v > 0 (normal numbers): exponent plus a cubic fit of log2 on [1, 2),
 * plain multiplications and additions only (max error 1.5e-4). */
static double est_log2(double v)
{
    uint64_t b;
    memcpy(&b, &v, 8);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    b = (b & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &b, 8);
    double x = m - 1.0;
    double p = 0x1.48f04fbb3fa68p-4 * x;
    p = p + -0x1.e4bf84b19febep-3;
    p = p * x;
    p = p + 0x1.c09645d8fab54p-2;
    double y = x * (1.0 - x);
    y = y * p;
    return ((double)e + x) + y;
}

/* round half away from zero */
static long round_half_away(double v)
{
    return v >= 0 ? (long)(v + 0.5) : -(long)(0.5 - v);
}

static int default_precision(int bps, uint32_t bs)
{
    if (bps < 16) { int p = 2 + bps / 2; return p < 5 ? 5 : p; }
    if (bps == 16)
        return bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
    return bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
}

void orc_enc_window(uint32_t bs, float *w)
{
    uint32_t np = bs / 4;
    for (uint32_t i = 0; i < bs; i++) w[i] = 1.0f;
    for (uint32_t i = 0; i < np; i++) {
        double v = 0.5 - 0.5 * cos(3.14159265358979323846 * (double)(i + 1) / (double)(np + 1));
        w[i] = (float)v;
        w[bs - 1 - i] = (float)v;
    }
}

/* ------------------------------------------------------------------ plans --- */

typedef struct {
    int type;              /* 0 constant, 1 verbatim, 2 fixed, 3 lpc */
    int order, wasted, sb; /* sb: sample bits after removing wasted bits */
    int prec, shift;
    int32_t coef[32];
    int po, method;
    uint8_t k[256];
    uint64_t bits;
} plan_t;

typedef struct { int po, method; uint8_t k[256]; uint64_t bits; } rice_t;   /* bits: method + order + parameters + codes */

/* Rice parameter of one partition from the sum of its folded residuals: the cheapest of a window of
 * four parameters around log2(sum / count), by the estimate count * (k + 1) + (sum >> k). */
static int rice_param(uint64_t sum, uint32_t cnt, uint64_t *est)
{
    if (cnt == 0) { *est = 0; return 0; }
    if (sum == 0) { *est = cnt; return 0; }
    int k0 = ilog2_64(sum) - ilog2_32(cnt);
    int lo = k0 - 2, hi = k0 + 1;
    if (lo < 0) lo = 0;
    if (lo > 30) lo = 30;
    if (hi < 0) hi = 0;
    if (hi > 30) hi = 30;
    uint64_t best = ~0ull; int bk = lo;
    for (int k = lo; k <= hi; k++) {
        uint64_t b = (uint64_t)cnt * (uint64_t)(k + 1) + (sum >> k);
        if (b < best) { best = b; bk = k; }
    }
    *est = best;
    return bk;
}

static void rice_search(const orc_enc_opts *o, const uint32_t *u, uint32_t bs, int order, rice_t *out)
{
    uint32_t lanes = (bs + RUN - 1) / RUN;
    int L = (int)o->max_partition_order;
    if (L > 8) L = 8;
    while (L > 0 && ((bs % ((uint32_t)RUN << L)) != 0 || (bs >> L) < (uint32_t)order + 1)) L--;
    int lo = (int)o->min_partition_order;
    if (lo > L) lo = L;

    static _Thread_local uint64_t psum[MAX_LANES];
    uint64_t heap[512];
    uint8_t kk[512];
    for (uint32_t t = 0; t < lanes; t++) {
        uint64_t s = 0;
        for (uint32_t j = 0; j < RUN; j++) {
            uint32_t i = t * RUN + j;
            if (i >= (uint32_t)order && i < bs) s += u[i];
        }
        psum[t] = s;
    }
    uint32_t P = 1u << L, tpp = L ? (bs >> L) / RUN : lanes;
    for (uint32_t p = 0; p < P; p++) {
        uint64_t s = 0;
        for (uint32_t q = 0; q < tpp; q++) s += psum[p * tpp + q];
        heap[P + p] = s;
    }
    for (uint32_t n = P - 1; n >= 1; n--) heap[n] = heap[2 * n] + heap[2 * n + 1];

    uint64_t best = ~0ull;
    for (int l = L; l >= lo; l--) {
        uint64_t total = 0; int need5 = 0;
        for (uint32_t p = 0; p < (1u << l); p++) {
            uint32_t n = (1u << l) + p;
            uint32_t cnt = (bs >> l) - (p == 0 ? (uint32_t)order : 0);
            uint64_t est;
            kk[n] = (uint8_t)rice_param(heap[n], cnt, &est);
            if (kk[n] > 14) need5 = 1;
            total += est;
        }
        total += (uint64_t)(need5 ? 5 : 4) << l;
        if (total < best) { best = total; out->po = l; out->method = need5; }
    }
    uint32_t np = 1u << out->po;
    memcpy(out->k, kk + np, np);
    /* exact size */
    uint64_t bits = 6 + (uint64_t)np * (out->method ? 5 : 4);
    uint32_t psize = bs >> out->po;
    for (uint32_t i = (uint32_t)order; i < bs; i++) {
        int k = out->k[out->po ? i / psize : 0];
        bits += (u[i] >> k) + (uint32_t)k + 1;
    }
    out->bits = bits;
}

static uint32_t fold(int32_t r) { return ((uint32_t)r << 1) ^ (uint32_t)(r >> 31); }

/* Plans one channel signal x (bs samples of cbps bits). */
static void plan_subframe(const orc_enc_opts *o, const int32_t *sig, uint32_t bs, int cbps, int stream_bps,
                          const float *win, plan_t *pl)
{
    static _Thread_local int32_t x[MAX_BS];
    static _Thread_local uint32_t uf[MAX_BS], ul[MAX_BS];
    static _Thread_local float xw[MAX_BS];
    memset(pl, 0, sizeof *pl);
    uint32_t ored = 0;
    for (uint32_t i = 0; i < bs; i++) ored |= (uint32_t)sig[i];
    int wasted = ored ? __builtin_ctz(ored) : 0;
    if (wasted >= cbps) wasted = cbps - 1;
    int sb = cbps - wasted;
    pl->wasted = wasted; pl->sb = sb;
    int constant = 1;
    for (uint32_t i = 0; i < bs; i++) { x[i] = sig[i] >> wasted; if (sig[i] != sig[0]) constant = 0; }
    uint64_t hdr = 8 + (uint64_t)wasted;
    if (constant) { pl->type = 0; pl->bits = hdr + (uint64_t)sb; return; }
    uint64_t verb_bits = hdr + (uint64_t)bs * (uint64_t)sb;

    /* fixed predictors */
    uint64_t e[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = 4; i < bs; i++) {
        int64_t d0 = x[i], d1 = (int64_t)x[i] - x[i - 1], d2 = (int64_t)x[i] - 2 * (int64_t)x[i - 1] + x[i - 2];
        int64_t d3 = (int64_t)x[i] - 3 * (int64_t)x[i - 1] + 3 * (int64_t)x[i - 2] - x[i - 3];
        int64_t d4 = (int64_t)x[i] - 4 * (int64_t)x[i - 1] + 6 * (int64_t)x[i - 2] - 4 * (int64_t)x[i - 3] + x[i - 4];
        e[0] += (uint64_t)llabs(d0); e[1] += (uint64_t)llabs(d1); e[2] += (uint64_t)llabs(d2);
        e[3] += (uint64_t)llabs(d3); e[4] += (uint64_t)llabs(d4);
    }
    int fo = 0;
    for (int k = 1; k <= 4; k++) if (e[k] < e[fo]) fo = k;
    if ((uint32_t)fo > bs) fo = (int)bs;
    for (uint32_t i = (uint32_t)fo; i < bs; i++) {
        const uint32_t *v = (const uint32_t *)x;
        uint32_t r;
        switch (fo) {
        case 0: r = v[i]; break;
        case 1: r = v[i] - v[i - 1]; break;
        case 2: r = v[i] - 2 * v[i - 1] + v[i - 2]; break;
        case 3: r = v[i] - 3 * v[i - 1] + 3 * v[i - 2] - v[i - 3]; break;
        default: r = v[i] - 4 * v[i - 1] + 6 * v[i - 2] - 4 * v[i - 3] + v[i - 4]; break;
        }
        uf[i] = fold((int32_t)r);
    }
    rice_t rf;
    rice_search(o, uf, bs, fo, &rf);
    uint64_t fixed_bits = hdr + (uint64_t)fo * (uint64_t)sb + rf.bits;

    /* LPC */
    uint32_t maxo = o->force_lpc_order ? o->force_lpc_order : o->max_lpc_order;
    if (maxo > 32) maxo = 32;
    int have_lpc = 0, ord = 0, prec = 0, shift = 0;
    int32_t q[32];
    rice_t rl;
    uint64_t lpc_bits = 0;
    if (maxo > 0 && bs > maxo) {
        uint32_t lanes = (bs + RUN - 1) / RUN, waves = (lanes + 63) / 64;
        for (uint32_t i = 0; i < bs; i++) xw[i] = (float)x[i] * win[i];
        double ac[33];
        static _Thread_local double part[MAX_LANES + 64];
        for (uint32_t lag = 0; lag <= maxo; lag++) {
            for (uint32_t t = 0; t < waves * 64; t++) {
                double a = 0.0;
                for (uint32_t j = 0; j < RUN; j++) {
                    uint32_t i = t * RUN + j;
                    if (i < bs && i >= lag) a = fma((double)xw[i], (double)xw[i - lag], a);
                }
                part[t] = a;
            }
            double total = 0.0;
            for (uint32_t w = 0; w < waves; w++) {
                double *v = part + w * 64, nv[64];
                for (int m = 1; m < 64; m <<= 1) {
                    for (int t = 0; t < 64; t++) nv[t] = v[t] + v[t ^ m];
                    memcpy(v, nv, sizeof nv);
                }
                total = w ? total + v[0] : v[0];
            }
            ac[lag] = total;
        }
        if (ac[0] > 0) {
            int prec0 = o->qlp_precision ? (int)o->qlp_precision : default_precision(stream_bps, bs);
            if (prec0 > 15) prec0 = 15;
            if (prec0 < 5) prec0 = 5;
            double err = ac[0], tmp[32], best[32], bestb = 0;
            const double scale = 0x1.0a6a4c7c29f9ap+0 / (double)bs;   /* 0.5 / ln(2)^2 / bs */
            for (uint32_t i = 0; i < maxo; i++) {
                double r = -ac[i + 1];
                for (uint32_t j = 0; j < i; j++) r -= tmp[j] * ac[i - j];
                if (err <= 0) break;
                r /= err;
                tmp[i] = r;
                for (uint32_t j = 0; j < i / 2; j++) {
                    double t = tmp[j];
                    tmp[j] += r * tmp[i - 1 - j];
                    tmp[i - 1 - j] += r * t;
                }
                if (i & 1) tmp[i / 2] += tmp[i / 2] * r;
                err *= (1.0 - r * r);
                uint32_t k = i + 1;
                int take;
                if (o->force_lpc_order) take = (k == o->force_lpc_order);
                else {
                    double v = err * scale, bpr = v > 0 ? 0.5 * est_log2(v) : 0.0;
                    if (bpr < 0) bpr = 0;
                    double tot = bpr * (double)(bs - k) + (double)k * (double)(sb + prec0);
                    take = (ord == 0 || tot < bestb);
                    if (take) bestb = tot;
                }
                if (take) { ord = (int)k; for (uint32_t j = 0; j < k; j++) best[j] = -tmp[j]; }
            }
            if (ord) {
                prec = prec0;
                if (sb <= 17 && !o->qlp_precision) {
                    int cap = 32 - sb - ilog2_32((uint32_t)ord);
                    if (prec > cap) prec = cap;
                }
                double cmax = 0;
                for (int j = 0; j < ord; j++) { double a = fabs(best[j]); if (a > cmax) cmax = a; }
                if (cmax > 0) {
                    uint64_t cb;
                    memcpy(&cb, &cmax, 8);
                    int ex = (int)((cb >> 52) & 0x7FF) - 1022;          /* cmax = f * 2^ex, 0.5 <= f < 1 */
                    shift = prec - ex - 1;
                    if (shift > 15) shift = 15;
                    int qmax = (1 << (prec - 1)) - 1, qmin = -qmax - 1;
                    double mul = shift >= 0 ? (double)(1 << shift) : 1.0 / (double)(1 << -shift);
                    if (shift < 0) shift = 0;
                    double carry = 0;
                    for (int j = 0; j < ord; j++) {
                        carry += best[j] * mul;
                        long v = round_half_away(carry);
                        if (v > qmax) v = qmax;
                        if (v < qmin) v = qmin;
                        carry -= (double)v;
                        q[j] = (int32_t)v;
                    }
                    int wide = (sb + prec + ilog2_32((uint32_t)ord)) > 32;
                    have_lpc = 1;
                    for (uint32_t i = (uint32_t)ord; i < bs && have_lpc; i++) {
                        int64_t pred;
                        if (!wide) {
                            uint32_t s = 0;
                            for (int j = 0; j < ord; j++) s += (uint32_t)q[j] * (uint32_t)x[i - 1 - j];
                            pred = (int32_t)s >> shift;
                        } else {
                            int64_t s = 0;
                            for (int j = 0; j < ord; j++) s += (int64_t)q[j] * (int64_t)x[i - 1 - j];
                            pred = s >> shift;
                        }
                        int64_t r = (int64_t)x[i] - pred;
                        if (r <= -2147483647LL - 1 || r > 2147483647LL) have_lpc = 0;
                        else ul[i] = fold((int32_t)r);
                    }
                    if (have_lpc) {
                        rice_search(o, ul, bs, ord, &rl);
                        lpc_bits = hdr + (uint64_t)ord * (uint64_t)sb + 9 + (uint64_t)ord * (uint64_t)prec + rl.bits;
                    }
                }
            }
        }
    }

    pl->type = 2; pl->order = fo; pl->po = rf.po; pl->method = rf.method; pl->bits = fixed_bits;
    memcpy(pl->k, rf.k, sizeof rf.k);
    if (have_lpc && (lpc_bits < fixed_bits || o->force_lpc_order)) {
        pl->type = 3; pl->order = ord; pl->prec = prec; pl->shift = shift;
        memcpy(pl->coef, q, sizeof q);
        pl->po = rl.po; pl->method = rl.method; pl->bits = lpc_bits;
        memcpy(pl->k, rl.k, sizeof rl.k);
    }
    if (verb_bits < pl->bits) { pl->type = 1; pl->bits = verb_bits; }   /* also under a forced order */
}

/* Stereo pair of a frame when stereo_mode is -2, from short linear predictors, before any channel is
 * planned.  The signals are w*l, w*r and w*l - w*r in single precision (w: the analysis window; the
 * samples as they are, no wasted-bit shift).  Their autocorrelation at lags 0..4 is summed in single
 * precision: per lane over its run of 16 in ascending order with fused multiply-adds, over the 16 lanes
 * of a row by a butterfly (partners at distance 1, 2, 4, 8), over the rows in ascending order.  Mid's
 * autocorrelation is taken from those three: side + 4 mid = 2 (left + right).  Levinson-Durbin in double precision gives the prediction error of orders 1..4 and, by the rule
 * that picks an LPC order, the estimated size of the subframe; the pair with the smallest sum of sizes
 * wins, earlier pairs (left/right, left/side, side/right, mid/side) on ties. */
static int estimate_stereo_mode(const orc_enc_opts *o, int32_t *const sig[4], uint32_t bs, int bps, const float *win)
{
    static _Thread_local float x[3][MAX_BS];       /* left, right, side */
    for (uint32_t i = 0; i < bs; i++) {
        float wl = (float)sig[0][i] * win[i], wr = (float)sig[1][i] * win[i];
        x[0][i] = wl; x[1][i] = wr; x[2][i] = wl - wr;
    }
    const uint32_t lanes = (bs + RUN - 1) / RUN, rows = (lanes + 15) / 16;
    int prec0 = o->qlp_precision ? (int)o->qlp_precision : default_precision(bps, bs);
    if (prec0 > 15) prec0 = 15;
    if (prec0 < 5) prec0 = 5;
    double est[4], acs[3][5];
    for (int c = 0; c < 3; c++) {
        for (uint32_t lag = 0; lag < 5; lag++) {
            float total = 0.0f;
            for (uint32_t row = 0; row < rows; row++) {
                float v[16], nv[16];
                for (uint32_t q = 0; q < 16; q++) {
                    float a = 0.0f;
                    for (uint32_t j = 0; j < RUN; j++) {
                        uint32_t i = (row * 16 + q) * RUN + j;
                        if (i < bs && i >= lag) a = fmaf(x[c][i], x[c][i - lag], a);
                    }
                    v[q] = a;
                }
                for (int m = 1; m < 16; m <<= 1) {
                    for (int q = 0; q < 16; q++) nv[q] = v[q] + v[q ^ m];
                    memcpy(v, nv, sizeof nv);
                }
                total = row ? total + v[0] : v[0];
            }
            acs[c][lag] = (double)total;
        }
    }
    for (int c = 0; c < 4; c++) {                  /* left, right, mid, side */
        double ac[5];
        for (int lag = 0; lag < 5; lag++)
            ac[lag] = c == 0 ? acs[0][lag] : c == 1 ? acs[1][lag] : c == 3 ? acs[2][lag]
                      : (2.0 * (acs[0][lag] + acs[1][lag]) - acs[2][lag]) * 0.25;
        const int cbps = bps + (c == 3);
        double e = 0.0;
        if (ac[0] > 0) {
            const double scale = 0x1.0a6a4c7c29f9ap+0 / (double)bs;   /* 0.5 / ln(2)^2 / bs */
            double err = ac[0], tmp[4] = {0, 0, 0, 0};
            int have = 0;
            for (uint32_t i = 0; i < 4 && i + 1 < bs; i++) {
                double r = -ac[i + 1];
                for (uint32_t j = 0; j < i; j++) r -= tmp[j] * ac[i - j];
                if (err <= 0) break;
                r /= err;
                tmp[i] = r;
                for (uint32_t j = 0; j < i / 2; j++) {
                    double t = tmp[j];
                    tmp[j] += r * tmp[i - 1 - j];
                    tmp[i - 1 - j] += r * t;
                }
                if (i & 1) tmp[i / 2] += tmp[i / 2] * r;
                err *= (1.0 - r * r);
                double v = err * scale, bpr = v > 0 ? 0.5 * est_log2(v) : 0.0;
                if (bpr < 0) bpr = 0;
                double tot = bpr * (double)(bs - (i + 1)) + (double)(i + 1) * (double)(cbps + prec0);
                if (!have || tot < e) { e = tot; have = 1; }
            }
        }
        est[c] = e;
    }
    const double b[4] = {est[0] + est[1], est[0] + est[3], est[3] + est[1], est[2] + est[3]};
    int mode = 0;
    for (int k = 1; k < 4; k++) if (b[k] < b[mode]) mode = k;
    return mode;
}

/* Writes the subframe a plan describes; x: the channel signal before removing wasted bits. */
static void write_subframe(bw_t *w, const plan_t *pl, const int32_t *sig, uint32_t bs)
{
    int typecode = pl->type == 0 ? 0 : pl->type == 1 ? 1 : pl->type == 2 ? 8 + pl->order : 31 + pl->order;
    bw_put(w, (uint32_t)(typecode << 1) | (pl->wasted ? 1u : 0u), 8);
    if (pl->wasted) { bw_zeros(w, (uint64_t)pl->wasted - 1); bw_put(w, 1, 1); }
    int sb = pl->sb, wst = pl->wasted;
    if (pl->type == 0) { bw_put(w, (uint32_t)(sig[0] >> wst), sb); return; }
    if (pl->type == 1) { for (uint32_t i = 0; i < bs; i++) bw_put(w, (uint32_t)(sig[i] >> wst), sb); return; }
    for (int i = 0; i < pl->order; i++) bw_put(w, (uint32_t)(sig[i] >> wst), sb);
    if (pl->type == 3) {
        bw_put(w, (uint32_t)pl->prec - 1, 4);
        bw_put(w, (uint32_t)pl->shift, 5);
        for (int i = 0; i < pl->order; i++) bw_put(w, (uint32_t)pl->coef[i], pl->prec);
    }
    bw_put(w, (uint32_t)pl->method, 2);
    bw_put(w, (uint32_t)pl->po, 4);
    int wide = pl->type == 3 && (sb + pl->prec + ilog2_32((uint32_t)pl->order)) > 32;
    uint32_t psize = bs >> pl->po;
    for (uint32_t i = (uint32_t)pl->order; i < bs; i++) {
        uint32_t p = pl->po ? i / psize : 0;
        if (i == (uint32_t)pl->order || (pl->po && i % psize == 0)) bw_put(w, pl->k[p], pl->method ? 5 : 4);
        int64_t pred = 0;
        if (pl->type == 3) {
            if (!wide) {
                uint32_t s = 0;
                for (int j = 0; j < pl->order; j++) s += (uint32_t)pl->coef[j] * (uint32_t)(sig[i - 1 - j] >> wst);
                pred = (int32_t)s >> pl->shift;
            } else {
                int64_t s = 0;
                for (int j = 0; j < pl->order; j++) s += (int64_t)pl->coef[j] * (int64_t)(sig[i - 1 - j] >> wst);
                pred = s >> pl->shift;
            }
        } else {
            int64_t a = i >= 1 ? sig[i - 1] >> wst : 0, b = i >= 2 ? sig[i - 2] >> wst : 0;
            int64_t c = i >= 3 ? sig[i - 3] >> wst : 0, d = i >= 4 ? sig[i - 4] >> wst : 0;
            switch (pl->order) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = 2 * a - b; break;
            case 3: pred = 3 * a - 3 * b + c; break;
            default: pred = 4 * a - 6 * b + 4 * c - d; break;
            }
        }
        int32_t r = (int32_t)((int64_t)(sig[i] >> wst) - pred);
        uint32_t u = fold(r);
        int k = pl->k[p];
        bw_zeros(w, u >> k);
        bw_put(w, (1u << k) | (k ? (u & ((1u << k) - 1)) : 0u), k + 1);
    }
    /* an order equal to the block size leaves no residual but the first parameter is still written */
    if ((uint32_t)pl->order >= bs) bw_put(w, pl->k[0], pl->method ? 5 : 4);
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

static int32_t load_sample(const uint8_t *p, int bytes)
{
    uint32_t v = 0;
    for (int k = 0; k < bytes; k++) v |= (uint32_t)p[k] << (8 * k);
    int sh = 32 - 8 * bytes;
    return (int32_t)(v << sh) >> sh;
}

uint64_t orc_encode_bound(uint64_t total_samples, int channels, int bps, uint32_t blocksize)
{
    uint64_t frames = (total_samples + blocksize - 1) / blocksize;
    uint64_t per_frame = 16 + 2 + ((uint64_t)blocksize * (uint64_t)channels * (uint64_t)(bps + 1) + 7) / 8 + 8ull * (uint64_t)channels;
    return 42 + frames * per_frame;
}

int orc_encode_stream(const uint8_t *pcm, uint64_t total_samples, int channels, int bps, uint32_t sample_rate,
                      const orc_enc_opts *o, const uint8_t *md5, uint8_t *out, size_t cap, size_t *out_len,
                      uint64_t *n_frames, uint32_t *min_fs_out, uint32_t *max_fs_out)
{
    if (!pcm || !o || !out || channels < 1 || channels > 8 || bps < 4 || bps > 24) return ORC_ERR_ARG;
    if (o->blocksize < 16 || o->blocksize > MAX_BS) return ORC_ERR_ARG;
    if (cap < 42) return ORC_ERR_CAPACITY;
    const uint32_t B = o->blocksize;
    const int bytes = (bps + 7) / 8;
    const int ncand = channels == 2 ? 4 : channels;
    static const uint32_t std_rates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};

    int32_t *sig[8];
    for (int c = 0; c < ncand; c++) sig[c] = (int32_t *)malloc(B * sizeof(int32_t));
    plan_t *plans = (plan_t *)malloc((size_t)ncand * sizeof(plan_t));
    float *win = (float *)malloc(B * sizeof(float));
    uint32_t win_bs = 0;

    const uint64_t total_frames = (total_samples + B - 1) / B;
    const uint64_t npoints = o->seektable_interval ? (total_frames + o->seektable_interval - 1) / o->seektable_interval : 0;
    const size_t meta_len = 42 + (o->seektable_interval ? 4 + 18 * (size_t)npoints : 0);
    if (cap < meta_len) return ORC_ERR_CAPACITY;
    size_t pos = meta_len;
    uint32_t min_fs = 0xFFFFFF, max_fs = 0;
    uint64_t frame_no = 0;
    int rc = ORC_OK;
    for (uint64_t done = 0; done < total_samples; done += B, frame_no++) {
        uint32_t bs = (uint64_t)B > total_samples - done ? (uint32_t)(total_samples - done) : B;
        const uint8_t *src = pcm + done * (uint64_t)channels * (uint64_t)bytes;
        if (bs != win_bs) { orc_enc_window(bs, win); win_bs = bs; }
        int ch_assign;
        const plan_t *use[8];
        const int32_t *use_sig[8];
        if (channels == 2) {
            for (uint32_t i = 0; i < bs; i++) {
                int32_t l = load_sample(src + (size_t)(2 * i) * bytes, bytes), r = load_sample(src + (size_t)(2 * i + 1) * bytes, bytes);
                sig[0][i] = l; sig[1][i] = r; sig[2][i] = (l + r) >> 1; sig[3][i] = l - r;
            }
            static const int pick[4][2] = {{0, 1}, {0, 3}, {3, 1}, {2, 3}};
            int mode = o->stereo_mode;
            if (mode == -2) mode = estimate_stereo_mode(o, sig, bs, bps, win);
            /* a pair known beforehand: its two channels are planned; -1: all four, the cheapest pair by exact size */
            for (int s = 0; s < 4; s++)
                if (mode < 0 || s == pick[mode][0] || s == pick[mode][1]) plan_subframe(o, sig[s], bs, bps + (s == 3), bps, win, &plans[s]);
            if (mode < 0) {
                uint64_t b0 = plans[0].bits + plans[1].bits, b1 = plans[0].bits + plans[3].bits;
                uint64_t b2 = plans[3].bits + plans[1].bits, b3 = plans[2].bits + plans[3].bits;
                mode = 0; uint64_t bb = b0;
                if (b1 < bb) { bb = b1; mode = 1; }
                if (b2 < bb) { bb = b2; mode = 2; }
                if (b3 < bb) { bb = b3; mode = 3; }
            }
            static const int assign[4] = {1, 8, 9, 10};
            ch_assign = assign[mode];
            for (int c = 0; c < 2; c++) { use[c] = &plans[pick[mode][c]]; use_sig[c] = sig[pick[mode][c]]; }
        } else {
            for (int c = 0; c < channels; c++) {
                for (uint32_t i = 0; i < bs; i++) sig[c][i] = load_sample(src + ((size_t)i * (size_t)channels + (size_t)c) * bytes, bytes);
                plan_subframe(o, sig[c], bs, bps, bps, win, &plans[c]);
                use[c] = &plans[c]; use_sig[c] = sig[c];
            }
            ch_assign = channels - 1;
        }

        size_t fstart = pos;
        if (cap - pos < 32) { rc = ORC_ERR_CAPACITY; break; }
        if (o->seektable_interval && frame_no % o->seektable_interval == 0) {
            /* SeekpointBlock, flacmeta.go:256-260: sample number, offset from frame 0, frame samples */
            uint8_t *sp = out + 46 + (size_t)(frame_no / o->seektable_interval) * 18;
            uint64_t off = fstart - meta_len;
            for (int k = 0; k < 8; k++) { sp[k] = (uint8_t)(done >> (56 - 8 * k)); sp[8 + k] = (uint8_t)(off >> (56 - 8 * k)); }
            sp[16] = (uint8_t)(bs >> 8); sp[17] = (uint8_t)bs;
        }
        uint8_t *hp = out + pos;
        int hn = 0;
        hp[hn++] = 0xFF;
        hp[hn++] = 0xF8;
        int bs_code, bs_extra = 0;
        if (bs == 192) bs_code = 1;
        else if (bs == 576 || bs == 1152 || bs == 2304 || bs == 4608) bs_code = 2 + ilog2_32(bs / 576);
        else if (bs >= 256 && bs <= 32768 && (bs & (bs - 1)) == 0) bs_code = 8 + ilog2_32(bs / 256);
        else if (bs <= 256) { bs_code = 6; bs_extra = 1; }
        else { bs_code = 7; bs_extra = 2; }
        int sr_code = 0;
        for (int i = 1; i < 12; i++) if (std_rates[i] == sample_rate) sr_code = i;
        hp[hn++] = (uint8_t)((bs_code << 4) | sr_code);
        int ss_code = bps == 8 ? 1 : bps == 12 ? 2 : bps == 16 ? 4 : bps == 20 ? 5 : bps == 24 ? 6 : 0;
        hp[hn++] = (uint8_t)((ch_assign << 4) | (ss_code << 1));
        hn += utf8_put(hp + hn, frame_no);
        if (bs_extra == 1) hp[hn++] = (uint8_t)(bs - 1);
        if (bs_extra == 2) { hp[hn++] = (uint8_t)((bs - 1) >> 8); hp[hn++] = (uint8_t)(bs - 1); }
        hp[hn] = orc_crc8(hp, (size_t)hn); hn++;
        pos += (size_t)hn;

        bw_t w = {out, cap, pos, 0, 0, 0};
        uint64_t expect = 0;
        for (int c = 0; c < channels; c++) { write_subframe(&w, use[c], use_sig[c], bs); expect += use[c]->bits; }
        if (w.n) bw_put(&w, 0, 8 - w.n);
        if (w.ovf || w.pos + 2 > cap) { rc = ORC_ERR_CAPACITY; break; }
        if (w.pos - pos != (expect + 7) / 8) { rc = ORC_ERR_ARG; break; }   /* planned size != written size */
        pos = w.pos;
        uint16_t c16 = orc_crc16(out + fstart, pos - fstart);
        out[pos++] = (uint8_t)(c16 >> 8); out[pos++] = (uint8_t)c16;
        uint32_t fs = (uint32_t)(pos - fstart);
        if (fs < min_fs) min_fs = fs;
        if (fs > max_fs) max_fs = fs;
    }

    /* marker, block header, STREAMINFO (flacmeta.go:754-761, :519-551, :644-717) */
    if (min_fs == 0xFFFFFF) min_fs = 0;
    memcpy(out, "fLaC", 4);
    out[4] = o->seektable_interval ? 0x00 : 0x80; out[5] = 0; out[6] = 0; out[7] = 34;
    if (o->seektable_interval) {
        uint32_t len = (uint32_t)(meta_len - 46);
        out[42] = 0x83; out[43] = (uint8_t)(len >> 16); out[44] = (uint8_t)(len >> 8); out[45] = (uint8_t)len;
    }
    uint8_t *s = out + 8;
    s[0] = (uint8_t)(B >> 8); s[1] = (uint8_t)B; s[2] = (uint8_t)(B >> 8); s[3] = (uint8_t)B;
    s[4] = (uint8_t)(min_fs >> 16); s[5] = (uint8_t)(min_fs >> 8); s[6] = (uint8_t)min_fs;
    s[7] = (uint8_t)(max_fs >> 16); s[8] = (uint8_t)(max_fs >> 8); s[9] = (uint8_t)max_fs;
    uint64_t v = ((uint64_t)sample_rate << 44) | ((uint64_t)(channels - 1) << 41) | ((uint64_t)(bps - 1) << 36) | (total_samples & 0xFFFFFFFFFULL);
    for (int k = 0; k < 8; k++) s[10 + k] = (uint8_t)(v >> (56 - 8 * k));
    if (md5) memcpy(s + 18, md5, 16);
    else orc_md5(pcm, (size_t)(total_samples * (uint64_t)channels * (uint64_t)bytes), s + 18);

    for (int c = 0; c < ncand; c++) free(sig[c]);
    free(plans); free(win);
    if (out_len) *out_len = pos;
    if (n_frames) *n_frames = frame_no;
    if (min_fs_out) *min_fs_out = min_fs;
    if (max_fs_out) *max_fs_out = max_fs;
    return rc;
}
