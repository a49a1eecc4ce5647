// frame_scan.hpp — frame discovery kernels (gfx950).
//
//   k_scan_mask     every byte of the batch is tested for the sync pair
//                   0xFF 0xF8|0xF9; a hit is kept when the header that starts
//                   there is valid for its stream (no reserved code, coded
//                   number well-formed, CRC-8 matches, channel count / sample
//                   size / block size agree with STREAMINFO).  Output: one mask
//                   bit per input byte (uint16 per 16-byte granule) and a
//                   candidate count per 64 KiB tile.
//   k_tile_prefix   exclusive prefix sum of the tile counts.
//   k_compact       ordered candidate list (offset, header fields) from the mask.
//   k_stream_ranges candidate range of every stream.
//   k_crc16_spans   CRC-16 of the bytes between consecutive candidates; one
//                   wave per span, lanes load consecutive 16-byte vectors and
//                   the partial remainders are joined by table multiplications
//                   in GF(2)[x] / 0x18005.
//   k_resolve_*     candidate i is a frame when the CRC-16 of a span that starts
//                   at i and ends at one of the next kLookahead candidates (or the
//                   end of the stream) is zero; output range checks.
//
// All of it is HBM-bound byte/integer work: loads are 16 bytes per lane,
// coalesced; per-candidate work (header parse, CRC-8) is rare and goes straight
// to global memory.
#pragma once

#include "device_types.hpp"

namespace flacgpu {

// ------------------------------------------------------------ header parse --

struct FrameHdr {
    uint32_t bs, hdrlen, ch_assign, bps, variable;
    uint64_t num;
};

__device__ inline uint32_t crc8_update(uint32_t c, uint32_t byte)
{
    c ^= byte;
#pragma unroll
    for (int k = 0; k < 8; ++k) c = (c & 0x80) ? ((c << 1) ^ 0x07) & 0xFF : (c << 1) & 0xFF;
    return c;
}

// p points at the candidate sync byte; avail = bytes left in its stream.
__device__ inline bool parse_frame_header(const uint8_t *p, uint64_t avail, const StreamDev &s, FrameHdr &h)
{
    if (avail < 6) return false;
    uint8_t b[16];
    const uint32_t nb = avail < 16 ? (uint32_t)avail : 16u;
    for (uint32_t i = 0; i < 16; ++i) b[i] = i < nb ? p[i] : 0;
    if (b[0] != 0xFF || (b[1] & 0xFE) != 0xF8) return false;
    const uint32_t variable = b[1] & 1;
    const uint32_t bs_code = b[2] >> 4, sr_code = b[2] & 15;
    const uint32_t ca = b[3] >> 4, ss_code = (b[3] >> 1) & 7;
    if ((b[3] & 1) || bs_code == 0 || sr_code == 15 || ca > 10 || ss_code == 3) return false;

    const uint32_t x = b[4];
    uint32_t extra;
    uint64_t num;
    if (x < 0x80) { extra = 0; num = x; }
    else if (x < 0xC0) return false;
    else if (x < 0xE0) { extra = 1; num = x & 0x1F; }
    else if (x < 0xF0) { extra = 2; num = x & 0x0F; }
    else if (x < 0xF8) { extra = 3; num = x & 0x07; }
    else if (x < 0xFC) { extra = 4; num = x & 0x03; }
    else if (x < 0xFE) { extra = 5; num = x & 0x01; }
    else if (x == 0xFE) { extra = 6; num = 0; }
    else return false;
    if (extra == 6 && !variable) return false;
    uint32_t pos = 5;
    if (pos + extra > nb) return false;
    for (uint32_t i = 0; i < extra; ++i) {
        const uint32_t c = b[pos++];
        if ((c & 0xC0) != 0x80) return false;
        num = (num << 6) | (c & 0x3F);
    }
    uint32_t bs;
    if (bs_code == 1) bs = 192;
    else if (bs_code <= 5) bs = 576u << (bs_code - 2);
    else if (bs_code == 6) { if (pos + 1 > nb) return false; bs = (uint32_t)b[pos] + 1; pos += 1; }
    else if (bs_code == 7) { if (pos + 2 > nb) return false; bs = (((uint32_t)b[pos] << 8) | b[pos + 1]) + 1; pos += 2; }
    else bs = 256u << (bs_code - 8);
    if (sr_code == 12) pos += 1;
    else if (sr_code >= 13) pos += 2;
    if (pos + 1 > nb) return false;
    uint32_t c = 0;
    for (uint32_t i = 0; i < pos; ++i) c = crc8_update(c, b[i]);
    if (c != b[pos]) return false;
    pos += 1;

    const uint32_t sizes = 0x20181410u;  // codes 4..7 -> 16, 20, 24, 32
    uint32_t bps;
    if (ss_code == 0) bps = s.bps;
    else if (ss_code == 1) bps = 8;
    else if (ss_code == 2) bps = 12;
    else bps = (sizes >> (8 * (ss_code - 4))) & 0xFF;
    const uint32_t nch = ca < 8 ? ca + 1 : 2;
    if (nch != s.channels || bps != s.bps || bs > s.max_bs) return false;
    h.bs = bs; h.hdrlen = pos; h.ch_assign = ca; h.bps = bps; h.variable = variable; h.num = num;
    return true;
}

// Stream that contains global byte offset g, or -1.
__device__ inline int find_stream(const StreamDev *streams, uint32_t n, uint64_t g)
{
    uint32_t lo = 0, hi = n;  // first stream with in_off > g
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (streams[mid].in_off <= g) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return -1;
    const StreamDev &s = streams[lo - 1];
    return (g - s.in_off < s.in_len) ? (int)(lo - 1) : -1;
}

// byte-wise "is zero" detector: 0x80 in every byte of v that is zero (exact).
__device__ inline uint32_t zero_bytes(uint32_t v)
{
    const uint32_t y = ((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v;
    return ~y & 0x80808080u;
}
// 0x80-per-byte mask -> 4-bit mask
__device__ inline uint32_t movemask4(uint32_t m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xF; }

// ------------------------------------------------------------------ scan ---

__global__ __launch_bounds__(256) void k_scan_mask(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                   const StreamDev *__restrict__ streams, uint32_t n_streams,
                                                   uint16_t *__restrict__ mask, uint32_t *__restrict__ tile_count)
{
    __shared__ uint32_t s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const uint64_t tile_base = (uint64_t)blockIdx.x * kScanTileBytes;
    uint32_t mine = 0;
    constexpr int iters = kScanTileBytes / kGranule / 256;
#pragma unroll 2
    for (int it = 0; it < iters; ++it) {
        const uint64_t g0 = tile_base + ((uint64_t)it * 256 + threadIdx.x) * kGranule;
        if (g0 >= in_bytes) break;
        uint32_t w[5];
        if (g0 + kGranule + 1 <= in_bytes) {
            const uint4 v = *reinterpret_cast<const uint4 *>(in + g0);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            w[4] = in[g0 + kGranule];
        } else {
            // last granule of the batch: byte loads, zero padded
            for (int k = 0; k < 5; ++k) {
                uint32_t x = 0;
                for (int j = 0; j < 4; ++j) {
                    const uint64_t a = g0 + (uint64_t)(4 * k + j);
                    if (a < in_bytes) x |= (uint32_t)in[a] << (8 * j);
                }
                w[k] = x;
            }
        }
        // FF[i]: byte i == 0xFF.  F8[i]: (byte i & 0xFE) == 0xF8.  A candidate starts at i when FF[i] & F8[i+1].
        uint32_t hits = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ff = zero_bytes(~w[k]);
            const uint32_t f8a = zero_bytes((w[k] & 0xFEFEFEFEu) ^ 0xF8F8F8F8u);
            const uint32_t f8b = zero_bytes((w[k + 1] & 0xFEFEFEFEu) ^ 0xF8F8F8F8u);
            const uint32_t next = (f8a >> 8) | (f8b << 24);
            hits |= movemask4(ff & next) << (4 * k);
        }
        uint32_t keep = 0;
        while (hits) {
            const uint32_t i = __builtin_ctz(hits);
            hits &= hits - 1;
            const uint64_t g = g0 + i;
            const int si = find_stream(streams, n_streams, g);
            if (si >= 0) {
                const StreamDev s = streams[si];
                FrameHdr h;
                if (parse_frame_header(in + g, s.in_off + s.in_len - g, s, h)) keep |= 1u << i;
            }
        }
        mask[g0 / kGranule] = (uint16_t)keep;
        mine += __builtin_popcount(keep);
    }
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_count;
}

// Exclusive prefix sum of n counts by one workgroup of 1024 threads; total -> *total.
__global__ __launch_bounds__(1024) void k_tile_prefix(const uint32_t *__restrict__ count, uint32_t *__restrict__ base,
                                                      uint32_t n, uint32_t *__restrict__ total)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t a = threadIdx.x * per, b = min(a + per, n);
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; ++i) sum += count[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (uint32_t i = a; i < b; ++i) { base[i] = run; run += count[i]; }
    if (threadIdx.x == 1023) *total = s_part[1023];
}

// One workgroup per tile: 4096 mask words (uint16) = 1024 uint64; thread t owns words 4t..4t+3.
__global__ __launch_bounds__(256) void k_compact(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                 const StreamDev *__restrict__ streams, uint32_t n_streams,
                                                 const uint64_t *__restrict__ mask64, uint64_t n_mask64,
                                                 const uint32_t *__restrict__ tile_count,
                                                 const uint32_t *__restrict__ tile_base, uint32_t cap, CandArrays c)
{
    if (tile_count[blockIdx.x] == 0) return;
    __shared__ uint32_t s_scan[256];
    const uint64_t w0 = (uint64_t)blockIdx.x * (kScanTileBytes / 64) + (uint64_t)threadIdx.x * 4;
    uint64_t m[4];
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = (w0 + k < n_mask64) ? mask64[w0 + k] : 0;
        cnt += __builtin_popcountll(m[k]);
    }
    s_scan[threadIdx.x] = cnt;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t r = tile_base[blockIdx.x] + s_scan[threadIdx.x] - cnt;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t bits = m[k];
        while (bits) {
            const uint32_t i = __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint64_t g = (w0 + k) * 64 + i;
            if (r < cap) {
                const int si = find_stream(streams, n_streams, g);
                const StreamDev s = streams[si];
                FrameHdr h;
                parse_frame_header(in + g, s.in_off + s.in_len - g, s, h);
                c.goff[r] = g;
                c.num[r] = h.num;
                c.info[r] = pack_info(h.bs, h.hdrlen, h.ch_assign, h.bps, h.variable);
                c.stream[r] = (uint32_t)si;
            }
            ++r;
        }
    }
}

// cand_begin / cand_end of every stream by binary search over the candidate offsets.
__global__ void k_stream_ranges(StreamDev *streams, uint32_t n_streams, const uint64_t *__restrict__ goff, uint32_t n_cand)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const uint64_t a = streams[s].in_off, b = a + streams[s].in_len;
    uint32_t lo = 0, hi = n_cand;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (goff[mid] < a) lo = mid + 1; else hi = mid; }
    const uint32_t begin = lo;
    hi = n_cand;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (goff[mid] < b) lo = mid + 1; else hi = mid; }
    streams[s].cand_begin = begin;
    streams[s].cand_end = lo;
    streams[s].pending = 0;
}

// ---------------------------------------------------------------- CRC-16 ---

// a * b in GF(2)[x] modulo x^16 + x^15 + x^2 + 1
__device__ inline uint32_t gf_mulmod(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        r <<= 1;
        if (r & 0x10000) r ^= 0x18005;
        if ((b >> i) & 1) r ^= a;
    }
    return r & 0xFFFF;
}

// x^(8n) mod P, from a table of x^(8 * 2^k)
__device__ inline uint32_t gf_xpow8(uint32_t n, const uint16_t *pow2)
{
    uint32_t r = 1;
    for (int k = 0; n; ++k, n >>= 1)
        if (n & 1) r = gf_mulmod(r, pow2[k]);
    return r;
}

// CRC-16 of `len` bytes followed by `after` zero-weighted bytes: crc(msg) * x^(8*after)
__device__ inline uint32_t crc16_shift(uint32_t crc, uint32_t after, const uint16_t *pow2)
{
    return after ? gf_mulmod(crc, gf_xpow8(after, pow2)) : crc;
}

// pow2[k] = x^(8 * 2^k) mod P, for gf_xpow8
__device__ inline void crc16_pow_table(uint16_t *pow2)
{
    if (threadIdx.x == 0) {
        uint32_t p = 0x100;  // x^8
        for (int k = 0; k < 32; ++k) { pow2[k] = (uint16_t)p; p = gf_mulmod(p, p); }
    }
}

// Lookup tables of k_crc16_spans, built once per handle (uint16 entries):
//   piece[m][v]  (16 x 256)  v * x^(16 + 8m): byte v with m more bytes behind it in its 16-byte piece
//   skip[h][v]   (2 x 256)   multiplication of a 16-bit remainder by x^(8 * 1024), split by byte
//   lanew[l]     (64)        x^(8 * 16 * (63 - l))
//   remw[r]      (16)        x^(8 * r)
constexpr int kCrcPiece = 0, kCrcSkip = 16 * 256, kCrcLane = kCrcSkip + 512, kCrcRem = kCrcLane + 64,
              kCrcTableEntries = kCrcRem + 16;

__global__ void k_crc16_build_tables(uint16_t *t)
{
    __shared__ uint16_t s_pow2[32];
    crc16_pow_table(s_pow2);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)kCrcTableEntries; i += blockDim.x) {
        uint32_t v;
        if (i < (uint32_t)kCrcSkip) {
            const uint32_t m = i >> 8, byte = i & 255;
            v = gf_mulmod(byte, gf_xpow8(2 + m, s_pow2));             // byte * x^(16 + 8m)
        } else if (i < (uint32_t)kCrcLane) {
            const uint32_t hsel = (i - kCrcSkip) >> 8, byte = i & 255;
            v = gf_mulmod(hsel ? byte << 8 : byte, gf_xpow8(1024, s_pow2));
        } else if (i < (uint32_t)kCrcRem) {
            v = gf_xpow8(16 * (63 - (i - kCrcLane)), s_pow2);
        } else {
            v = gf_xpow8(i - kCrcRem, s_pow2);
        }
        t[i] = (uint16_t)v;
    }
}

// CRC-16 of the 16 bytes of a vector taken as a message of its own (init 0): sixteen independent lookups.
__device__ inline uint32_t crc16_piece(const uint4 v, const uint16_t *piece)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t byte = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
        c ^= piece[(15 - k) * 256 + byte];
    }
    return c;
}

// CRC-16 of every candidate span [goff[r], next), next = the following candidate of the
// same stream or the end of the stream.  One wave per span at a time (waves stride over
// the spans); lanes load consecutive 16-byte vectors (1 KiB per wave per step), reduce
// each to its remainder with table lookups, and carry it over the 1 KiB stride by a
// table multiplication; the 64 partial remainders are weighted by lane and XOR-ed.
__global__ __launch_bounds__(256) void k_crc16_spans(const uint8_t *__restrict__ in, const StreamDev *__restrict__ streams,
                                                     CandArrays c, uint32_t n_cand, const uint16_t *__restrict__ tables)
{
    __shared__ __align__(16) uint16_t s_t[kCrcTableEntries];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kCrcTableEntries / 2; i += blockDim.x)
        reinterpret_cast<uint32_t *>(s_t)[i] = reinterpret_cast<const uint32_t *>(tables)[i];
    __syncthreads();
    const uint16_t *piece = s_t + kCrcPiece, *skip = s_t + kCrcSkip;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t lanew = s_t[kCrcLane + lane];
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_cand; r += waves) {
        const uint64_t a = c.goff[r];
        const uint32_t si = c.stream[r];
        uint64_t b = streams[si].in_off + streams[si].in_len;
        if (r + 1 < n_cand && c.stream[r + 1] == si) b = c.goff[r + 1];
        const uint64_t A = a & ~(uint64_t)15;
        const uint32_t lead = (uint32_t)(a - A);
        const uint32_t nfull = (uint32_t)((b - A) >> 4), rem = (uint32_t)((b - A) & 15);
        const uint32_t steps = (nfull + 63) >> 6, pad = steps * 64 - nfull;
        const uint4 *src = reinterpret_cast<const uint4 *>(in + A);
        uint32_t acc = 0;
        for (uint32_t t = 0; t < steps; ++t) {
            const int32_t j = (int32_t)(t * 64 + lane) - (int32_t)pad;
            uint32_t pc = 0;
            if (j >= 0) {
                uint4 v = src[j];
                if (j == 0 && lead) {   // bytes before the span count as zeros (no effect with init 0)
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int32_t nz = (int32_t)lead - 4 * i;
                        if (nz >= 4) w[i] = 0;
                        else if (nz > 0) w[i] &= 0xFFFFFFFFu << (8 * nz);
                    }
                    v = uint4{w[0], w[1], w[2], w[3]};
                }
                pc = crc16_piece(v, piece);
            }
            acc = skip[acc & 0xFF] ^ skip[256 + (acc >> 8)] ^ pc;
        }
        acc = gf_mulmod(acc, lanew);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc ^= __shfl_xor(acc, d);
        if (lane == 0) {
            const uint32_t crc = gf_mulmod(acc, s_t[kCrcRem + rem]);
            // the bytes after the last whole vector (at most 15), byte at a time:
            // crc = (crc << 8) ^ T[(crc >> 8) ^ byte] with T[v] = v * x^16 = piece[0][v]
            uint32_t tcrc = 0;
            const uint64_t t0 = A + ((uint64_t)nfull << 4);
            for (uint64_t g = (t0 > a ? t0 : a); g < b; ++g)
                tcrc = ((tcrc << 8) & 0xFFFF) ^ piece[(tcrc >> 8) ^ in[g]];
            c.segcrc[r] = (uint16_t)(crc ^ tcrc);
            c.seglen[r] = (uint32_t)(b - a);
        }
    }
}

// --------------------------------------------------------------- resolve ---

// Output-range check shared by both resolve kernels.
__device__ inline int range_status(const StreamDev &s, uint32_t info, uint64_t num)
{
    const uint64_t first = info_var(info) ? num : num * (uint64_t)s.max_bs;
    const uint64_t end = first + info_bs(info);
    if (s.total_samples && end > s.total_samples) return kFrameRange;
    if (end * s.channels * s.bytes_ps > s.out_cap) return kFrameRange;
    return kFrameOK;
}

// Fast path, one thread per candidate: a span with zero CRC-16 is a frame.
__global__ void k_resolve_fast(StreamDev *streams, CandArrays c, uint32_t n_cand)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_cand) return;
    const uint32_t si = c.stream[r];
    c.len[r] = c.seglen[r];
    if (c.segcrc[r] == 0) {
        c.status[r] = range_status(streams[si], c.info[r], c.num[r]);
    } else {
        c.status[r] = kFrameBadCrc16;
        streams[si].pending = 1;
    }
}

// Slow path, one thread per stream that has a span with a non-zero CRC-16: walk
// the candidates in order, trying up to kLookahead end points for each.
__global__ void k_resolve_slow(StreamDev *streams, uint32_t n_streams, CandArrays c)
{
    __shared__ uint16_t s_pow2[32];
    crc16_pow_table(s_pow2);
    __syncthreads();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams || !streams[s].pending) return;
    const StreamDev st = streams[s];
    uint32_t cur = st.cand_begin;
    while (cur < st.cand_end) {
        uint32_t crc = 0, total = 0, found = 0;
        for (uint32_t j = cur; j < cur + kLookahead && j < st.cand_end; ++j) {
            // crc([cur, j+1)) = crc([cur, j)) * x^(8 * seglen[j]) + segcrc[j]
            crc = crc16_shift(crc, c.seglen[j], s_pow2) ^ c.segcrc[j];
            total += c.seglen[j];
            if (crc == 0) { found = j + 1; break; }
        }
        if (found) {
            c.len[cur] = total;
            c.status[cur] = range_status(st, c.info[cur], c.num[cur]);
            for (uint32_t j = cur + 1; j < found; ++j) { c.status[j] = kFrameSkipped; c.len[j] = 0; }
            cur = found;
        } else {
            c.len[cur] = c.seglen[cur];
            c.status[cur] = kFrameBadCrc16;
            ++cur;
        }
    }
}

// ---------------------------------------------------------------- results --

// One wave per stream: frames, OK frames, samples and the end of the last OK frame.
__global__ __launch_bounds__(256) void k_results(const StreamDev *__restrict__ streams, uint32_t n_streams,
                                                 CandArrays c, ResultDev *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t s = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (s >= n_streams) return;
    const StreamDev st = streams[s];
    uint64_t frames = 0, ok = 0, samples = 0, end = 0;
    for (uint32_t r = st.cand_begin + lane; r < st.cand_end; r += 64) {
        const int status = c.status[r];
        if (status == kFrameSkipped) continue;
        ++frames;
        if (status == kFrameOK) {
            const uint32_t info = c.info[r];
            const uint64_t first = info_var(info) ? c.num[r] : c.num[r] * (uint64_t)st.max_bs;
            ++ok;
            samples += info_bs(info);
            const uint64_t e = (first + info_bs(info)) * st.channels * st.bytes_ps;
            end = e > end ? e : end;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        frames += __shfl_xor(frames, d);
        ok += __shfl_xor(ok, d);
        samples += __shfl_xor(samples, d);
        const uint64_t o = __shfl_xor(end, d);
        end = o > end ? o : end;
    }
    if (lane == 0) {
        ResultDev res;
        res.header_candidates = st.cand_end - st.cand_begin;
        res.frames = frames; res.frames_ok = ok; res.samples = samples; res.pcm_bytes = end;
        out[s] = res;
    }
}

}  // namespace flacgpu
