// frame_scan.hpp — frame discovery kernels (gfx950).
//
//   k_scan_tiles    every byte of the batch is tested for the sync pair
//                   0xFF 0xF8|0xF9; a hit is kept when the header that starts
//                   there is valid for its stream (no reserved code, coded
//                   number well-formed, CRC-8 matches, channel count / sample
//                   size / block size agree with STREAMINFO).  Output: a sparse
//                   mask with one bit per input byte (uint16 per 16-byte granule,
//                   written only where not zero), a bitmap of the granules whose
//                   mask entry is set, a candidate count per 64 KiB tile, and the
//                   cut records of the frame CRC-16.
//   k_tile_prefix_* exclusive prefix sum of the tile counts (two levels).
//   k_compact       ordered candidate offsets from the bitmap and the mask.
//   k_cand_fill     header fields of every candidate.
//   k_stream_ranges candidate range of every stream.
//   k_span_crc      CRC-16 of the bytes between consecutive candidates from the
//                   records k_scan_tiles leaves where it cuts its running CRC-16
//                   (every sync code, and the end of every range a wave steps over).
//   k_crc16_spans   the same by reading the bytes: one wave per span, lanes load
//                   consecutive 16-byte vectors into wide accumulators.  For the few
//                   spans the records cannot give, and for the encoder.
//   k_resolve_*     candidate i is a frame when the CRC-16 of a span that starts
//                   at i and ends at one of the next kLookahead candidates (or the
//                   end of the stream) is zero; candidates inside an accepted frame
//                   are struck out; output range checks.
//
// All of it is HBM-bound byte/integer work: loads are 16 bytes per lane,
// coalesced; per-candidate work (header parse, CRC-8) is rare and goes straight
// to global memory.
#pragma once

#include "device_types.hpp"
#include "variants.hpp"

namespace flacgpu {

// ------------------------------------------------------------ header parse --

struct FrameHdr {
    uint32_t bs, hdrlen, ch_assign, bps, variable;
    uint64_t num;
};

__device__ inline uint32_t crc8_update(uint32_t c, uint32_t byte)
{
    // (c ^ byte) * x^8 modulo x^8 + x^2 + x + 1: x^8 = x^2 + x + 1, so the product is t * (x^2 + x + 1) (ten bits), and its two
    // bits above x^7 fold down the same way (degree 3 at most: no third step).  Seven operations instead of a loop over bits.
    const uint32_t t = c ^ byte;
    const uint32_t h = t ^ (t << 1) ^ (t << 2);
    const uint32_t hi = h >> 8;
    return (h & 0xFFu) ^ hi ^ (hi << 1) ^ (hi << 2);
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

// ---------------------------------------------------------------- CRC-16 arithmetic ---

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

// Lookup tables, built once per handle (uint16 entries):
//   pow2[k]      (32)        x^(8 * 2^k), for gf_xpow8
//   inv[r]       (16)        x^(-8 * r): x has order 32767 modulo 0x18005 = (x + 1)(x^15 + x + 1)
// and for k_span_crc, which joins the scan's cut records:
//   invpad[n]    (1025)      x^(-8 * n), the zero bytes between a cut and the end of its 1 KiB step
//   powlo[n]     (128)       x^(8 * n)
//   powhi[n]     (513)       x^(8 * 128 * n): x^(8 * d) = powhi[d / 128] * powlo[d % 128] for d up to 64 KiB (a range is at most a scan tile)
constexpr int kCrcPow2 = 0, kCrcInv = kCrcPow2 + 32, kCrcCoreEntries = kCrcInv + 16, kCrcInvPad = kCrcCoreEntries,
              kCrcPowLo = kCrcInvPad + 1025, kCrcPowHi = kCrcPowLo + 128, kCrcTableEntries = (kCrcPowHi + 513 + 1) & ~1;

__global__ void k_crc16_build_tables(uint16_t *t)
{
    __shared__ uint16_t s_pow2[32];
    crc16_pow_table(s_pow2);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)kCrcTableEntries; i += blockDim.x) {
        uint32_t v;
        if (i < (uint32_t)kCrcInv) v = s_pow2[i - kCrcPow2];
        else if (i < (uint32_t)kCrcInvPad) v = gf_xpow8((32767u - (i - kCrcInv)) % 32767u, s_pow2);
        else if (i < (uint32_t)kCrcPowLo) v = gf_xpow8((32767u - (i - kCrcInvPad)) % 32767u, s_pow2);
        else if (i < (uint32_t)kCrcPowHi) v = gf_xpow8(i - kCrcPowLo, s_pow2);
        else v = gf_xpow8(128 * (i - kCrcPowHi), s_pow2);
        t[i] = (uint16_t)v;
    }
}

// CRC-16 of 128 message bits (init 0) without tables: byte-table lookups from 64 lanes collide in the LDS banks
// and made the kernel LDS-bound.  The polynomial factors: x^16 + x^15 + x^2 + 1 = (x + 1)(x^15 + x + 1).  Modulo x + 1 a polynomial is the
// parity of its bits.  Modulo Q = x^15 + x + 1, x^15 = x + 1, so a high part h (bits 15 and up) folds down as
// h + h*x; x^32 = x^4 + x^2, x^64 = x^8 + x^4, x^96 = x^12 + x^10 + x^8 + x^6 and x^16 = x^2 + x.  The
// remainder r modulo Q and the parity p of the message give the CRC as r + c*Q with c = p + parity(r)
// (Q has three terms, so adding it flips the parity).
__device__ inline uint32_t crc16_fold15(uint32_t w)   // congruent to w modulo Q; below 2^18 for any w, below 2^15 for w < 2^19
{
    const uint32_t h = w >> 15;
    return (w & 0x7FFFu) ^ h ^ (h << 1);
}
// Remainder modulo Q of a four-word message (most significant word first); below 2^17.
__device__ inline uint32_t crc16_words_q(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    const uint32_t a = crc16_fold15(w0), b = crc16_fold15(w1), c = crc16_fold15(w2), d = crc16_fold15(w3);
    const uint32_t t = (a << 12) ^ (a << 10) ^ (a << 8) ^ (a << 6) ^ (b << 8) ^ (b << 4) ^ (c << 4) ^ (c << 2) ^ d;   // < 2^30
    return crc16_fold15(t);          // < 2^17
}
// Times x^16 = x^2 + x modulo Q (the CRC is the remainder of message * x^16), reduced below 2^15; q < 2^17.
__device__ inline uint32_t crc16_times_x16(uint32_t q)
{
    return crc16_fold15((q << 2) ^ (q << 1));
}
// The CRC from the remainder modulo Q and the message's parity (its low bit).
__device__ inline uint32_t crc16_join(uint32_t q, uint32_t parity)
{
    const uint32_t flip = (parity + __popc(q)) & 1u;
    return q ^ (0x8003u & (0u - flip));
}
// Wide accumulators.  A lane of a wave that steps over a message 1 KiB at a time sees pieces that are 8192 bit
// positions apart, and modulo P = x^16 + x^15 + x^2 + 1 (x has order 32767)
//     x^32768 = x,     x^8192 = x^8 + x^4 + x,     x^128 = x^8 + x^2 + x
// (relations with an even number of terms hold modulo x + 1 as well as modulo x^15 + x + 1, so modulo P).  Pieces
// four steps apart therefore differ by a factor x: a lane keeps four accumulators, one per step index modulo 4,
// and adds a piece to its accumulator as  A = A * x + piece  on the unreduced 128-bit value (a one-bit shift over
// five words and four XORs) instead of reducing every piece to 16 bits.  A fifth word takes the bits shifted out:
// 32 steps per accumulator, then crc16_acc_fold brings it back to 128 bits (x^128 above).  The four accumulators
// are joined by Horner's rule with x^8192 and reduced once per message (crc16_acc_merge).
struct CrcAcc { uint32_t w[5]; };   // w[0] least significant
__device__ inline void crc16_acc_clear(CrcAcc &A) { A.w[0] = A.w[1] = A.w[2] = A.w[3] = A.w[4] = 0; }
__device__ inline void crc16_acc_step(CrcAcc &A, const uint4 v)   // v: 16 message bytes in memory order
{
    A.w[4] = __builtin_amdgcn_alignbit(A.w[4], A.w[3], 31);
    A.w[3] = __builtin_amdgcn_alignbit(A.w[3], A.w[2], 31) ^ __builtin_bswap32(v.x);
    A.w[2] = __builtin_amdgcn_alignbit(A.w[2], A.w[1], 31) ^ __builtin_bswap32(v.y);
    A.w[1] = __builtin_amdgcn_alignbit(A.w[1], A.w[0], 31) ^ __builtin_bswap32(v.z);
    A.w[0] = (A.w[0] << 1) ^ __builtin_bswap32(v.w);
}
// t * x^128 folded onto (lo, hi) = the words 128 and 96 bit positions below t
__device__ inline void crc16_fold128(uint32_t t, uint32_t &lo, uint32_t &hi)
{
    lo ^= (t << 8) ^ (t << 2) ^ (t << 1);
    hi ^= (t >> 24) ^ (t >> 30) ^ (t >> 31);
}
__device__ inline void crc16_acc_fold(CrcAcc &A)
{
    crc16_fold128(A.w[4], A.w[0], A.w[1]);
    A.w[4] = 0;
}
// A lane's pieces as one remainder: n0 holds the newest step (weight 1), n1 the step before it (x^8192), n2, n3;
// Horner's rule with x^8192 on the unreduced words, the words from bit 128 up folded down, then the piece
// arithmetic above.  q: remainder modulo Q of the lane's message, without the CRC's factor x^16 (below 2^17);
// parity: of the lane's message (low bit).  WORDS = 6 holds
// any accumulators (160 bits and three multiplications by x^8 + ...); 5 is enough when no accumulator has more
// than three steps behind it (131 + 24 bits).
template <int WORDS>
__device__ inline void crc16_acc_merge(const CrcAcc &n0, const CrcAcc &n1, const CrcAcc &n2, const CrcAcc &n3, uint32_t &q,
                                       uint32_t &parity)
{
    static_assert(WORDS == 5 || WORDS == 6, "");
    uint32_t h[6] = {n3.w[0], n3.w[1], n3.w[2], n3.w[3], n3.w[4], 0u};
    auto horner = [&](const CrcAcc &n) {   // h = h * (x^8 + x^4 + x) + n
#pragma unroll
        for (int i = WORDS - 1; i >= 1; --i)
            h[i] = __builtin_amdgcn_alignbit(h[i], h[i - 1], 24) ^ __builtin_amdgcn_alignbit(h[i], h[i - 1], 28) ^
                   __builtin_amdgcn_alignbit(h[i], h[i - 1], 31) ^ (i < 5 ? n.w[i] : 0u);
        h[0] = (h[0] << 8) ^ (h[0] << 4) ^ (h[0] << 1) ^ n.w[0];
    };
    horner(n2); horner(n1); horner(n0);
    if (WORDS == 6) crc16_fold128(h[5], h[1], h[2]);
    crc16_fold128(h[4], h[0], h[1]);
    parity = __popc(h[0]) + __popc(h[1]) + __popc(h[2]) + __popc(h[3]);
    q = crc16_words_q(h[3], h[2], h[1], h[0]);
}
// The remainders of the 64 lanes of a wave as one CRC-16, the same in every lane: lane l's message stands
// 128 * (63 - l) bit positions above lane 63's.  Modulo Q, x^128 = x^8 + x^2 + x, x^256 = x^4 + x, x^512 = x^8 + x^2,
// x^1024 = x^4 + x^2 + x, x^2048 = x^8 + x^4 + x^2: four doubling steps inside rows of 16 lanes (data-parallel
// moves, no LDS), the four row results joined on the scalar side, where the CRC's factor x^16 goes in as well;
// modulo x + 1 every weight is 1.
template <int CTRL>
__device__ inline uint32_t crc16_dpp(uint32_t v)   // lanes without a source read zero
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ inline uint32_t crc16_wave_join(uint32_t q, uint32_t parity)   // q < 2^17: crc16_acc_merge's
{
    uint32_t t;
    t = crc16_dpp<0x111>(q); q ^= (t << 8) ^ (t << 2) ^ (t << 1);                    // row_shr:1, < 2^25
    t = crc16_dpp<0x112>(q); q = crc16_fold15(q ^ (t << 4) ^ (t << 1));              // row_shr:2, < 2^29 before the fold
    t = crc16_dpp<0x114>(q); q ^= (t << 8) ^ (t << 2);                               // row_shr:4, < 2^23
    t = crc16_dpp<0x118>(q); q = crc16_fold15(q ^ (t << 4) ^ (t << 2) ^ (t << 1));   // row_shr:8, < 2^27 before the fold
    uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)q, 15);
#pragma unroll
    for (int row = 1; row < 4; ++row)
        r = crc16_fold15((r << 8) ^ (r << 4) ^ (r << 2)) ^ (uint32_t)__builtin_amdgcn_readlane((int)q, 16 * row + 15);
    const uint32_t odd = (uint32_t)__popcll(__ballot((parity & 1u) != 0));
    return crc16_join(crc16_times_x16(r), odd);
}

// Bytes [from, to) of a 16-byte piece, the rest zero.
__device__ inline uint4 crc16_keep_bytes(uint4 v, uint32_t from, uint32_t to)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int32_t lo = (int32_t)from - 4 * i, hi = (int32_t)to - 4 * i;   // kept byte range inside word i
        uint32_t m = 0xFFFFFFFFu;
        if (lo >= 4 || hi <= 0) m = 0;
        else {
            if (lo > 0) m &= 0xFFFFFFFFu << (8 * lo);
            if (hi < 4) m &= 0xFFFFFFFFu >> (8 * (4 - hi));
        }
        w[i] &= m;
    }
    return uint4{w[0], w[1], w[2], w[3]};
}

// The first n bytes (0 .. 16) of a piece given as message words (most significant first), the rest zero.
__device__ inline void crc16_keep_head(uint32_t n, uint32_t &w0, uint32_t &w1, uint32_t &w2, uint32_t &w3)
{
    uint32_t *w[4] = {&w0, &w1, &w2, &w3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int32_t nb = (int32_t)n - 4 * i;   // bytes of word i that are kept
        const uint32_t m = nb <= 0 ? 0u : nb >= 4 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (8 * (4 - nb));
        *w[i] &= m;
    }
}

// Cut records of the scan.  k_scan_tiles has every byte in registers once; it keeps the wide accumulators of the
// CRC-16 running while it searches, and cuts the running message at every sync code it sees (validated or not) and
// at the end of the range a wave steps over (32 KiB).  A cut leaves a record: where it is, and the CRC-16 of the bytes since
// the cut before it, zero-padded to the end of the cut's 1 KiB step.  The CRC-16 of a span between two candidates
// (both are sync codes, so both are cuts) is then a few multiplications by powers of x over the records in between
// (k_span_crc): the compressed bytes are not read a second time for it.
//   segrec[range * kSegSlots + 0]      = number of cuts in the range (kSegOverflow: more than fit) | CRC-16 up to the range's end << 16
//   segrec[range * kSegSlots + 1 + i]  = offset of cut i inside the range | CRC-16 up to the cut << 16
// A range ends with a cut of its own, and a cut costs about 300 vector instructions, so longer ranges are cheaper: two
// waves per 64 KiB tile, 32 KiB each, instead of four with 16 KiB (one cut of 3.25 per 16 KiB saved in 16-bit stereo
// with 4096-sample blocks).  One wave per tile would leave a CU with too few waves (the tile's LDS allows 13 blocks).
constexpr uint32_t kScanWaves = variant::kScanWaves;                            // waves per scan tile
constexpr uint32_t kScanThreads = 64 * kScanWaves;
constexpr uint32_t kSegRangeBytes = kScanTileBytes / kScanWaves;                // one wave's share of a scan tile
constexpr uint32_t kSegRangeLog2 = kSegRangeBytes == 65536 ? 16 : kSegRangeBytes == 32768 ? 15 : 14;
constexpr uint32_t kSegSlots = 32 * (kSegRangeBytes / 16384);                   // the same room per byte of input whatever the range
constexpr uint32_t kSegOverflow = 0xFFFFu;
static_assert((1u << kSegRangeLog2) == kSegRangeBytes && kSegRangeBytes * kScanWaves == kScanTileBytes, "a scan tile is its waves' ranges");
constexpr uint32_t kRangeChunks = kSegRangeBytes / 1024;                        // 1 KiB wave steps per range
// crc16_acc_merge<5> holds accumulators with up to eight steps behind them (128 + 7 bits, three multiplications by x^8 + ...: 159)
static_assert(kRangeChunks / 4 <= 8 || kScanWaves == 1, "more steps per accumulator need the six-word merge");

// ------------------------------------------------------------------ scan ---

constexpr uint32_t kChunkBytes = 1024;                            // one wave step: 64 lanes x 16 bytes
constexpr uint32_t kTileChunks = kScanTileBytes / kChunkBytes;    // 64
constexpr uint32_t kTileGranules = kScanTileBytes / kGranule;     // 4096
constexpr uint32_t kHitCap = 2048;                                // sync hits a workgroup queues for validation
constexpr uint32_t kPrefixGroup = 1024;                           // tiles per first-level prefix group

// Lane mask of the lanes whose low / high 16 bits of x equal c.  The sub-dword operand select does the
// field extraction inside the compare, and the result lands in a scalar register pair.
__device__ inline uint64_t eq_lo16(uint32_t x, uint32_t c)
{
    uint64_t m;
    asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(c));
    return m;
}
__device__ inline uint64_t eq_hi16(uint32_t x, uint32_t c)
{
    uint64_t m;
    asm("v_cmp_eq_u32_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:DWORD" : "=s"(m) : "v"(x), "v"(c));
    return m;
}

// Stream that contains global byte offset g among streams [lo, hi), or -1.
__device__ inline int find_stream_in(const StreamDev *streams, uint32_t lo, uint32_t hi, uint64_t g)
{
    const uint32_t first = lo;
    while (lo < hi) {   // first stream with in_off > g
        const uint32_t mid = (lo + hi) >> 1;
        if (streams[mid].in_off <= g) lo = mid + 1; else hi = mid;
    }
    if (lo == first) return -1;
    const StreamDev &s = streams[lo - 1];
    return (g - s.in_off < s.in_len) ? (int)(lo - 1) : -1;
}

// First stream that can own a byte of each scan tile (the last stream that starts at or before the tile's first byte),
// one thread per tile.  The scan's workgroups would otherwise each walk the stream table by bisection before
// they touch their tile: two dozen dependent loads in front of 40 microseconds of work.
__global__ __launch_bounds__(256) void k_tile_streams(const StreamDev *__restrict__ streams, uint32_t n_streams, uint32_t n_tiles,
                                                      uint32_t *__restrict__ tile_first)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint64_t tile_base = (uint64_t)t * kScanTileBytes;
    uint32_t lo = 0, hi = n_streams;   // first stream with in_off > tile_base
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (streams[mid].in_off <= tile_base) lo = mid + 1; else hi = mid; }
    tile_first[t] = lo ? lo - 1 : 0;
}

// One workgroup per 64 KiB tile.
//   1. detect: each wave steps over its range in 1 KiB chunks, four chunks loaded ahead.  A sync code
//      at byte j is the 16-bit field (byte j, byte j+1) with (field & 0xFEFF) == 0xF8FF; sixteen
//      sub-dword compares per chunk give the hit lanes of every byte position as scalar masks, so the
//      common chunk without any sync code costs no per-lane work beyond the compares.
//   2. queue: hit positions go to a list in LDS (a tile with more hits than the list holds is redone
//      byte by byte; that does not happen with audio).
//   3. validate: the queued positions are checked densely, one per thread (header fields, coded number,
//      CRC-8, agreement with STREAMINFO); survivors set their bit in the tile's byte mask in LDS.
//   4. write-out: mask words that are not zero go to `mask16` (sparse: the rest of the array is not
//      written and never read), `bitmap64` gets one bit per 16-byte granule ("mask16 entry is valid and
//      not zero"), `tile_count` the number of candidates.
__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                    const StreamDev *__restrict__ streams, uint32_t n_streams,
                                                    uint16_t *__restrict__ mask16, uint64_t *__restrict__ bitmap64,
                                                    uint32_t *__restrict__ tile_count, uint32_t *__restrict__ segrec,
                                                    const uint32_t *__restrict__ tile_first)
{
    __shared__ __align__(16) uint32_t s_mask[kScanTileBytes / 32];
    __shared__ uint16_t s_hits[kHitCap];
    __shared__ uint32_t s_nhits, s_count;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    variant::ClockProbe probe;
    probe.begin();
#pragma unroll
    for (int k = 0; k < (int)(kScanTileBytes / 32 / kScanThreads); ++k) s_mask[k * kScanThreads + tid] = 0;   // consecutive lanes, consecutive banks
    if (tid == 0) { s_nhits = 0; s_count = 0; }
    __syncthreads();
    const uint64_t tile_base = (uint64_t)blockIdx.x * kScanTileBytes;
    // streams that can own a byte of this tile: [s_lo, s_hi)
    uint32_t s_lo, s_hi;
    {
        s_lo = tile_first[blockIdx.x];     // k_tile_streams
        // first stream at or after s_lo with in_off >= end of tile: usually the next one or the one after, so gallop, then bisect
        const uint64_t tile_end = tile_base + kScanTileBytes;
        uint32_t lo = s_lo, hi = s_lo, step = 1;
        while (hi < n_streams && streams[hi].in_off < tile_end) { lo = hi + 1; hi += step; step <<= 1; }
        if (hi > n_streams) hi = n_streams;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (streams[mid].in_off < tile_end) lo = mid + 1; else hi = mid; }
        s_hi = lo;
    }
    // What bytes 2 and 3 of a frame header can be here, as bit tables indexed by the code (parse_frame_header's tests on
    // them): any stream's when the tile has bytes of several streams, else that one stream's channel count, sample
    // size and largest block.  A sync code whose next bytes fail them is no candidate, and is dropped before it costs a cut.
    uint32_t ok_ca = 0x07FFu, ok_ss = 0xF7u, ok_bs = 0xFFFEu;
    if (s_hi - s_lo == 1) {
        const uint32_t channels = streams[s_lo].channels, bps = streams[s_lo].bps, max_bs = streams[s_lo].max_bs;
        ok_ca = channels == 2 ? 0x0702u : (1u << (channels - 1)) & 0xFFu;
        ok_ss = 1u | (bps == 8 ? 2u : bps == 12 ? 4u : bps == 16 ? 16u : bps == 20 ? 32u : bps == 24 ? 64u : bps == 32 ? 128u : 0u);
        ok_bs = 0x00C0u;   // codes 6 and 7: the size follows the header's number
        for (uint32_t c = 1; c < 16; ++c) {
            const uint32_t bs = c == 1 ? 192u : c <= 5 ? 576u << (c - 2) : 256u << (c - 8);
            if (c != 6 && c != 7 && bs <= max_bs) ok_bs |= 1u << c;
        }
    }
    auto validate = [&](uint32_t off) {
        const uint64_t g = tile_base + off;
        const int si = find_stream_in(streams, s_lo, s_hi, g);
        if (si < 0) return;
        const StreamDev s = streams[si];
        FrameHdr h;
        if (parse_frame_header(in + g, s.in_off + s.in_len - g, s, h)) atomicOr(&s_mask[off >> 5], 1u << (off & 31));
    };

    // CRC-16 of the bytes since the last cut: accumulator k takes the chunks whose index is k modulo 4
    CrcAcc acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) crc16_acc_clear(acc[k]);
    uint32_t *const rec = segrec + ((uint64_t)blockIdx.x * kScanWaves + wave) * kSegSlots;
    uint32_t ncuts = 0;
    // CRC-16 of everything accumulated, as if the message ended with chunk k (index modulo 4): the same in every lane
    auto wave_flush = [&](int k) {
        uint32_t q, parity;
        crc16_acc_merge<(kRangeChunks / 4 <= 8 ? 5 : 6)>(acc[k], acc[(k + 3) & 3], acc[(k + 2) & 3], acc[(k + 1) & 3], q, parity);
        return crc16_wave_join(q, parity);
    };

    const uint32_t field = 0xF8FFu;
    const uint8_t *const tile_in = in + tile_base;
    // offset, from the tile's first byte, of the last 16-byte vector that holds a byte of the input (the grid has no
    // tile without one)
    const uint32_t last_vec = (uint32_t)min((uint64_t)kScanTileBytes - kGranule, (in_bytes - tile_base - 1) & ~(uint64_t)(kGranule - 1));
    for (uint32_t round = 0; round < kRangeChunks / 4; ++round) {
        const uint32_t chunk0 = wave * kRangeChunks + round * 4;
        uint4 v[4];
        // A lane past the end of the input loads the input's last vector again instead (one 32-bit minimum per load,
        // and the address is the tile's scalar base plus a 32-bit lane offset): what such lanes see is never used,
        // positions at or after in_bytes belong to no stream and a cut there ends no span.
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t off = min((chunk0 + k) * kChunkBytes + lane * kGranule, last_vec);
            v[k] = *reinterpret_cast<const uint4 *>(tile_in + off);
        }
        const uint64_t g_after = tile_base + (uint64_t)(chunk0 + 4) * kChunkBytes;
        // the four bytes after the round's last chunk (its last lane's sync test needs one, the header test three more)
        uint32_t word_after = 0;
        if (g_after + 4 <= in_bytes) word_after = *reinterpret_cast<const uint32_t *>(in + g_after);
        else for (uint32_t i = 0; g_after + i < in_bytes; ++i) word_after |= (uint32_t)in[g_after + i] << (8 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // first byte of the granule that follows this lane's
            uint32_t nb = crc16_dpp<0x130>(v[k].x);   // wave_shl:1, lane l reads lane l + 1
            const uint32_t wrap = k < 3 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v[k < 3 ? k + 1 : 3].x) : word_after;
            if (lane == 63) nb = wrap;
            const uint32_t w[5] = {v[k].x, v[k].y, v[k].z, v[k].w, nb};
            uint64_t m[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t a = w[q] & 0xFEFFFEFFu;                                            // bytes 4q .. 4q+3
                const uint32_t u = __builtin_amdgcn_alignbit(w[q + 1], w[q], 8) & 0xFEFFFEFFu;    // bytes 4q+1 .. 4q+4
                m[4 * q + 0] = eq_lo16(a, field);
                m[4 * q + 1] = eq_lo16(u, field);
                m[4 * q + 2] = eq_hi16(a, field);
                m[4 * q + 3] = eq_hi16(u, field);
            }
            uint64_t any = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) any |= m[j];
            // The chunk joins its accumulator whole; the rare chunk with a sync code takes its bytes out again below and
            // adds them piece by piece (one path through the common case: a branch with the step on one side and the cuts
            // on the other made the compiler copy the other accumulators' registers on the common side).
            crc16_acc_step(acc[k], v[k]);
            if (any) {
                // Most sync codes are chance (one per 32 KiB of audio): keep those whose bytes 2 .. 4 can begin a header.
                any = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (m[j]) {   // the same for the whole wave
                        const uint32_t b2 = (w[(j + 2) >> 2] >> (8 * ((j + 2) & 3))) & 0xFFu, b3 = (w[(j + 3) >> 2] >> (8 * ((j + 3) & 3))) & 0xFFu,
                                       b4 = (w[(j + 4) >> 2] >> (8 * ((j + 4) & 3))) & 0xFFu;
                        const uint32_t codes = (ok_bs >> (b2 >> 4)) & (ok_ca >> (b3 >> 4)) & (ok_ss >> ((b3 >> 1) & 7u)) & ~b3 & 1u;
                        const bool plausible = codes != 0 && (b2 & 15u) != 15u && b4 - 0x80u >= 0x40u && b4 != 0xFFu;
                        m[j] &= __ballot(plausible);
                    }
                    any |= m[j];
                }
            }
            if (any) {
                uint32_t hits = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) hits |= (uint32_t)((m[j] >> lane) & 1) << j;
                const uint32_t my_hits = hits;
                if (hits) {
                    uint32_t slot = atomicAdd(&s_nhits, (uint32_t)__builtin_popcount(hits));
                    const uint32_t off0 = (chunk0 + k) * kChunkBytes + lane * kGranule;
                    while (hits) {
                        const uint32_t off = off0 + __builtin_ctz(hits);
                        hits &= hits - 1;
                        if (slot < kHitCap) s_hits[slot] = (uint16_t)off;   // else: the whole tile is redone below
                        ++slot;
                    }
                }
                // cut the running CRC-16 at every hit, in stream order: the bytes of this chunk before the hit end a
                // message, the rest begins the next one
                uint32_t c0 = __builtin_bswap32(v[k].x), c1 = __builtin_bswap32(v[k].y), c2 = __builtin_bswap32(v[k].z),
                         c3 = __builtin_bswap32(v[k].w);   // the chunk's bytes not yet added, as message words
                acc[k].w[3] ^= c0; acc[k].w[2] ^= c1; acc[k].w[1] ^= c2; acc[k].w[0] ^= c3;   // (the step above added them all)
                for (uint64_t hl = __ballot(my_hits != 0); hl; hl &= hl - 1) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(hl);
                    for (uint32_t hb = (uint32_t)__builtin_amdgcn_readlane((int)my_hits, (int)l); hb; hb &= hb - 1) {
                        const uint32_t o = (uint32_t)__builtin_ctz(hb);
                        uint32_t p0 = c0, p1 = c1, p2 = c2, p3 = c3;   // what of it lies before the hit
                        if (lane == l) crc16_keep_head(o, p0, p1, p2, p3);
                        if (lane > l) p0 = p1 = p2 = p3 = 0;
                        c0 ^= p0; c1 ^= p1; c2 ^= p2; c3 ^= p3;
                        acc[k].w[3] ^= p0; acc[k].w[2] ^= p1; acc[k].w[1] ^= p2; acc[k].w[0] ^= p3;
                        const uint32_t crc = wave_flush(k);
                        if (ncuts < kSegSlots - 1 && lane == 0)
                            rec[1 + ncuts] = ((round * 4 + k) * kChunkBytes + l * kGranule + o) | (crc << 16);
                        ++ncuts;
#pragma unroll
                        for (int q = 0; q < 4; ++q) crc16_acc_clear(acc[q]);
                    }
                }
                acc[k].w[3] ^= c0; acc[k].w[2] ^= c1; acc[k].w[1] ^= c2; acc[k].w[0] ^= c3;
            }
        }
    }
    {
        const uint32_t crc = wave_flush(3);
        if (lane == 0) rec[0] = (ncuts < kSegSlots ? ncuts : kSegOverflow) | (crc << 16);
    }
    __syncthreads();
    const uint32_t nh = s_nhits;
    if (nh <= kHitCap) {
        for (uint32_t i = tid; i < nh; i += kScanThreads) validate(s_hits[i]);
    } else {
        // more sync codes than the list holds (not audio: a crafted or degenerate input): every thread
        // tests the byte positions it owns one by one
        constexpr uint32_t kOwn = kScanTileBytes / kScanThreads;
        for (uint32_t i = 0; i < kOwn; ++i) {
            const uint32_t off = tid * kOwn + i;
            const uint64_t g = tile_base + off;
            if (g + 1 < in_bytes && in[g] == 0xFF && (in[g + 1] & 0xFE) == 0xF8) validate(off);
        }
    }
    __syncthreads();

    // A virtual thread v of 256 owns granules 16v .. 16v+15 = mask words 8v .. 8v+7; a thread plays 256 / kScanThreads of them.
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t rep = 0; rep < 256 / kScanThreads; ++rep) {
        const uint32_t vt = rep * kScanThreads + tid;
        uint32_t nz = 0;
        const uint64_t granule0 = (uint64_t)blockIdx.x * kTileGranules + vt * 16;
        // (two 16-byte reads: word by word, lanes eight words apart would share four of the 32 banks)
        const uint4 m_lo = reinterpret_cast<const uint4 *>(s_mask)[vt * 2], m_hi = reinterpret_cast<const uint4 *>(s_mask)[vt * 2 + 1];
        const uint32_t m_own[8] = {m_lo.x, m_lo.y, m_lo.z, m_lo.w, m_hi.x, m_hi.y, m_hi.z, m_hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = m_own[k];
            if (x) {
                cnt += __builtin_popcount(x);
                if (x & 0xFFFFu) { mask16[granule0 + 2 * k] = (uint16_t)x; nz |= 1u << (2 * k); }
                if (x >> 16) { mask16[granule0 + 2 * k + 1] = (uint16_t)(x >> 16); nz |= 2u << (2 * k); }
            }
        }
        // four virtual threads' 16 flags make one 64-bit bitmap word (1 KiB of input); they are four neighbouring lanes
        const uint32_t base = lane & ~3u;
        const uint32_t n0 = (uint32_t)__shfl((int)nz, (int)base), n1 = (uint32_t)__shfl((int)nz, (int)base + 1),
                       n2 = (uint32_t)__shfl((int)nz, (int)base + 2), n3 = (uint32_t)__shfl((int)nz, (int)base + 3);
        if ((lane & 3) == 0)
            bitmap64[(uint64_t)blockIdx.x * kTileChunks + (vt >> 2)] = (uint64_t)(n0 | (n1 << 16)) | ((uint64_t)(n2 | (n3 << 16)) << 32);
    }
    if (cnt) atomicAdd(&s_count, cnt);
    __syncthreads();
    if (tid == 0) tile_count[blockIdx.x] = s_count;
    probe.end("scan", 4096);
}

// Exclusive prefix sum of the tile counts in two levels: within groups of kPrefixGroup tiles (one
// workgroup each), then over the group totals (one workgroup).  base of tile t = tile_base[t] + group_base[t / kPrefixGroup].
__device__ inline uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *s_wave, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < 16; ++w) { const uint32_t x = s_wave[w]; if (w < wave) before += x; all += x; }
    __syncthreads();
    total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(1024) void k_tile_prefix_local(const uint32_t *__restrict__ count, uint32_t *__restrict__ base,
                                                            uint32_t n, uint32_t *__restrict__ group_sum)
{
    __shared__ uint32_t s_wave[16];
    const uint32_t i = blockIdx.x * kPrefixGroup + threadIdx.x;
    const uint32_t v = i < n ? count[i] : 0;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
    if (i < n) base[i] = ex;
    if (threadIdx.x == 0) group_sum[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_tile_prefix_groups(const uint32_t *__restrict__ group_sum, uint32_t *__restrict__ group_base,
                                                             uint32_t n_groups, uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t s_wave[16];
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += 1024) {
        const uint32_t i = g0 + threadIdx.x;
        const uint32_t v = i < n_groups ? group_sum[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, s_wave, total);
        if (i < n_groups) group_base[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// Ordered candidate offsets from the bitmap and the sparse mask: one wave per tile, lane l owns
// bitmap word l (1 KiB of input).
__global__ __launch_bounds__(256) void k_compact(const uint16_t *__restrict__ mask16, const uint64_t *__restrict__ bitmap64,
                                                 const uint32_t *__restrict__ tile_count, const uint32_t *__restrict__ tile_base,
                                                 const uint32_t *__restrict__ group_base, uint32_t n_tiles, uint32_t cap,
                                                 uint64_t *__restrict__ goff)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles || tile_count[tile] == 0) return;
    const uint64_t word = (uint64_t)tile * kTileChunks + lane;
    const uint64_t bits = bitmap64[word];
    uint32_t cnt = 0;
    for (uint64_t b = bits; b; b &= b - 1) cnt += __builtin_popcount(mask16[word * 64 + __builtin_ctzll(b)]);
    uint32_t inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= (uint32_t)d) inc += o;
    }
    uint32_t r = tile_base[tile] + group_base[tile / kPrefixGroup] + inc - cnt;
    for (uint64_t b = bits; b; b &= b - 1) {
        const uint64_t granule = word * 64 + __builtin_ctzll(b);
        for (uint32_t m = mask16[granule]; m; m &= m - 1) {
            if (r < cap) goff[r] = granule * kGranule + __builtin_ctz(m);
            ++r;
        }
    }
}

// Header fields of every candidate, one thread per candidate.
__global__ __launch_bounds__(256) void k_cand_fill(const uint8_t *__restrict__ in, const StreamDev *__restrict__ streams,
                                                   uint32_t n_streams, CandArrays c, uint32_t n_cand)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_cand) return;
    const uint64_t g = c.goff[r];
    const int si = find_stream(streams, n_streams, g);
    const StreamDev s = streams[si < 0 ? 0 : si];
    FrameHdr h;
    h.bs = 0; h.hdrlen = 0; h.ch_assign = 0; h.bps = 1; h.variable = 0; h.num = 0;
    if (si >= 0) parse_frame_header(in + g, s.in_off + s.in_len - g, s, h);
    c.num[r] = h.num;
    c.info[r] = pack_info(h.bs, h.hdrlen, h.ch_assign, h.bps, h.variable);
    c.stream[r] = (uint32_t)(si < 0 ? 0 : si);
    // the candidate's span: up to the next candidate, or to the end of its stream
    uint64_t end = s.in_off + s.in_len;
    if (r + 1 < n_cand) { const uint64_t nx = c.goff[r + 1]; if (nx < end) end = nx; }
    c.seglen[r] = si < 0 ? 0u : (uint32_t)(end - g);
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

// ---------------------------------------------------------- CRC-16 spans ---

// CRC-16 of every candidate span [goff[r], goff[r] + seglen[r]).  One wave per span at a time (waves
// stride over the spans).  The span is cut into 16-byte pieces on 16-byte boundaries; the bytes of the
// first piece before the span and of the last piece after it are taken as zeros (leading zeros do not
// change a CRC that starts from zero; trailing zeros multiply it by a known power of x, divided out at
// the end).  Lanes load consecutive pieces (1 KiB per wave per step) and add each to one of four wide
// accumulators (crc16_acc_step); at the end of the span a lane reduces its accumulators to one remainder,
// and the 64 remainders are weighted by lane and XOR-ed.  The next step's piece (or the first piece of the
// wave's next span) is loaded before the current one is processed.
// With `list`, the spans are those of the candidates list[0 .. *list_count) only.
__global__ __launch_bounds__(256) void k_crc16_spans(const uint8_t *__restrict__ in, CandArrays c, uint32_t n_cand,
                                                     const uint16_t *__restrict__ tables, const uint32_t *__restrict__ list = nullptr,
                                                     const uint32_t *__restrict__ list_count = nullptr)
{
    __shared__ __align__(16) uint16_t s_t[kCrcCoreEntries];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kCrcCoreEntries / 2; i += blockDim.x)
        reinterpret_cast<uint32_t *>(s_t)[i] = reinterpret_cast<const uint32_t *>(tables)[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);

    struct Span { uint64_t A; uint32_t lead, npieces, steps, pad, trail; };
    // a span belongs to the whole wave: its geometry goes through the first lane into scalar registers, so the
    // step loop and its first-step / last-step cases are scalar branches, not lane masks
    auto uniform32 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto geometry = [&](uint64_t a, uint32_t len) {
        Span g;
        a = ((uint64_t)uniform32((uint32_t)(a >> 32)) << 32) | uniform32((uint32_t)a);
        len = uniform32(len);
        g.A = a & ~(uint64_t)15;
        g.lead = (uint32_t)(a - g.A);
        const uint32_t total = g.lead + len;
        g.npieces = (total + 15) >> 4;
        g.trail = (16 - (total & 15)) & 15;          // zero bytes after the span in its last piece
        g.steps = (g.npieces + 63) >> 6;
        g.pad = g.steps * 64 - g.npieces;            // empty lanes in front of the first step
        return g;
    };
    auto load_piece = [&](const Span &g, uint32_t t) {
        const int32_t j = (int32_t)(t * 64 + lane) - (int32_t)g.pad;
        return (j >= 0 && g.npieces) ? *reinterpret_cast<const uint4 *>(in + g.A + (uint64_t)j * 16) : uint4{0u, 0u, 0u, 0u};
    };
    if (list) n_cand = *list_count;
    auto span_of = [&](uint32_t i) {
        const uint32_t r = list ? list[i] : i;
        return geometry(c.goff[r], c.seglen[r]);
    };
    uint32_t i = uniform32(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (i >= n_cand) return;
    Span g = span_of(i);
    uint4 cur = load_piece(g, 0);
    while (i < n_cand) {
        const uint32_t r = uniform32(list ? list[i] : i);
        const uint32_t r_next = i + waves;
        Span g_next = g;
        if (r_next < n_cand) g_next = span_of(r_next);
        CrcAcc a0, a1, a2, a3;   // by the step's distance from the span's last step, modulo 4
        crc16_acc_clear(a0); crc16_acc_clear(a1); crc16_acc_clear(a2); crc16_acc_clear(a3);
        const uint32_t steps = uniform32(g.steps);
        const bool more = uniform32(r_next < n_cand ? 1u : 0u) != 0;
        for (uint32_t t = 0; t < steps; ++t) {
            uint4 nxt = uint4{0u, 0u, 0u, 0u};
            if (t + 1 < steps) nxt = load_piece(g, t + 1);
            else if (more) nxt = load_piece(g_next, 0);
            const int32_t j = (int32_t)(t * 64 + lane) - (int32_t)g.pad;
            if (t == 0 && g.lead && j == 0) cur = crc16_keep_bytes(cur, g.lead, 16);
            if (t + 1 == steps && g.trail && j == (int32_t)g.npieces - 1) cur = crc16_keep_bytes(cur, 0, 16 - g.trail);
            switch ((steps - 1 - t) & 3u) {
            case 0: crc16_acc_step(a0, cur); break;
            case 1: crc16_acc_step(a1, cur); break;
            case 2: crc16_acc_step(a2, cur); break;
            default: crc16_acc_step(a3, cur); break;
            }
            if ((t & 127u) == 127u) { crc16_acc_fold(a0); crc16_acc_fold(a1); crc16_acc_fold(a2); crc16_acc_fold(a3); }
            cur = nxt;
        }
        uint32_t q, parity;
        crc16_acc_merge<6>(a0, a1, a2, a3, q, parity);
        const uint32_t acc = crc16_wave_join(q, parity);
        if (lane == 0) c.segcrc[r] = (uint16_t)(g.trail ? gf_mulmod(acc, s_t[kCrcInv + g.trail]) : acc);
        i = r_next;
        g = g_next;
    }
}

// CRC-16 of every candidate span from the scan's cut records, one thread per candidate.  A span that runs from one
// candidate to the next starts and ends at a cut; the records of the cuts after its start, up to its end, each hold
// the CRC-16 of the bytes since the cut before (zero-padded to the end of a 1 KiB step, which a multiplication by a
// negative power of x takes out again), and  crc(A B) = crc(A) * x^(8 |B|) + crc(B).  Spans the records cannot
// give (the last span of a stream ends at the stream's end, not at a cut; a range with more cuts than its slots
// hold) are listed in fb_list for k_crc16_spans, which reads their bytes.
__global__ __launch_bounds__(256) void k_span_crc(CandArrays c, uint32_t n_cand, const uint32_t *__restrict__ segrec,
                                                  const uint16_t *__restrict__ tables, uint32_t *__restrict__ fb_list,
                                                  uint32_t *__restrict__ fb_count)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_cand) return;
    const uint64_t a = c.goff[r];
    const uint32_t len = c.seglen[r];
    if (len == 0) { c.segcrc[r] = 0; return; }
    const uint64_t e = a + len;
    bool ok = r + 1 < n_cand && c.goff[r + 1] == e;
    uint32_t crc = 0;
    if (ok) {
        uint64_t range = a >> kSegRangeLog2;
        const uint32_t *rec = segrec + range * kSegSlots;
        uint32_t hdr = rec[0], n = hdr & 0xFFFFu, i = 1;
        if (n == kSegOverflow) ok = false;
        else {
            const uint32_t off_a = (uint32_t)a & (kSegRangeBytes - 1);
            while (i <= n && (rec[i] & 0xFFFFu) != off_a) ++i;   // the cut at the span's first byte
            ok = i <= n;
        }
        uint64_t pos = a;
        while (ok && pos < e) {
            uint64_t b;
            uint32_t v, pad;
            if (i < n) {             // the next cut of this range
                const uint32_t x = rec[++i], off = x & 0xFFFFu;
                b = (range << kSegRangeLog2) + off;
                v = x >> 16;
                pad = 1024u - (off & 1023u);    // zero bytes between the cut and the end of its step
            } else {                 // the rest of the range; go on in the next one
                b = (range + 1) << kSegRangeLog2;
                v = hdr >> 16;
                pad = 0;
                if (b < e) {
                    ++range;
                    rec += kSegSlots;
                    hdr = rec[0]; n = hdr & 0xFFFFu; i = 0;
                    if (n == kSegOverflow) ok = false;
                }
            }
            if (b > e) { ok = false; break; }
            if (pad) v = gf_mulmod(v, tables[kCrcInvPad + pad]);
            if (crc) {
                const uint32_t d = (uint32_t)(b - pos);   // at most a range
                crc = gf_mulmod(gf_mulmod(crc, tables[kCrcPowHi + (d >> 7)]), tables[kCrcPowLo + (d & 127u)]);
            }
            crc ^= v;
            pos = b;
        }
    }
    if (ok) c.segcrc[r] = (uint16_t)crc;
    else fb_list[atomicAdd(fb_count, 1u)] = r;
}

// --------------------------------------------------------------- resolve ---

// Output-range check shared by both resolve kernels.
__device__ inline int range_status(const StreamDev &s, uint32_t info, uint64_t num)
{
    const uint64_t first = info_var(info) ? num : num * (uint64_t)s.max_bs;
    const uint64_t end = first + info_bs(info);
    if (s.total_samples && end > s.total_samples) return kFrameRange;
    if (first < s.sample_base) return kFrameRange;
    const uint64_t rel = end - s.sample_base;   // samples from the start of the stream's output region
    if (s.plane_stride ? rel * s.bytes_ps > s.plane_stride : rel * s.channels * s.bytes_ps > s.out_cap) return kFrameRange;
    return kFrameOK;
}

// One thread per candidate: the frame that starts here, if any.  The CRC-16 of [r, j+1) follows from the
// span remainders: crc([r, j+1)) = crc([r, j)) * x^(8 * seglen[j]) + segcrc[j].  next[r] = the candidate
// after the frame (r + 1 when no end point closes the CRC).
__global__ __launch_bounds__(256) void k_resolve_chain(StreamDev *streams, CandArrays c, uint32_t n_cand,
                                                       const uint16_t *__restrict__ pow2)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_cand) return;
    const uint32_t si = c.stream[r];
    uint32_t crc = c.segcrc[r], total = c.seglen[r], found = crc == 0 ? r + 1 : 0;
    if (!found) {
        const uint32_t cand_end = streams[si].cand_end;
        for (uint32_t j = r + 1; j < r + kLookahead && j < cand_end; ++j) {
            const uint32_t sl = c.seglen[j];
            crc = crc16_shift(crc, sl, pow2) ^ c.segcrc[j];
            total += sl;
            if (crc == 0) { found = j + 1; break; }
        }
    }
    if (found) {
        c.len[r] = total;
        // the subframe kernels count a frame's bits in 32 bits from a 64-byte-aligned base
        c.status[r] = total >= (1u << 29) - 64 ? kFrameUnsupported : range_status(streams[si], c.info[r], c.num[r]);
        c.next[r] = found;
        if (found != r + 1) streams[si].pending = 1;   // candidates inside this frame have to be struck out
    } else {
        c.len[r] = c.seglen[r];
        c.status[r] = kFrameBadCrc16;
        c.next[r] = r + 1;
    }
}

// One wave per stream in which some frame spans several candidates: follow next[] from the stream's
// first candidate; candidates the walk does not visit lie inside an accepted frame.
__global__ __launch_bounds__(256) void k_resolve_walk(const StreamDev *__restrict__ streams, uint32_t n_streams, CandArrays c)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_streams || !streams[s].pending) return;
    const uint32_t begin = streams[s].cand_begin, end = streams[s].cand_end;
    uint32_t cur = begin;   // wave-uniform
    for (uint32_t base = begin; base < end; base += 64) {
        const uint32_t r = base + lane;
        const uint32_t nx = r < end ? c.next[r] : 0xFFFFFFFFu;
        uint64_t reach = 0;
        while (cur < end && cur - base < 64) {
            const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cur - base));
            reach |= 1ull << i;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)i);
        }
        if (r < end && !((reach >> lane) & 1)) { c.status[r] = kFrameSkipped; c.len[r] = 0; }
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
    uint32_t bad = 0xFFFFFFFFu;   // first candidate of this lane that is a frame with an error
    for (uint32_t r = st.cand_begin + lane; r < st.cand_end; r += 64) {
        const int status = c.status[r];
        if (status == kFrameSkipped) continue;
        ++frames;
        if (status == kFrameOK) {
            const uint32_t info = c.info[r];
            const uint64_t first = info_var(info) ? c.num[r] : c.num[r] * (uint64_t)st.max_bs;
            ++ok;
            samples += info_bs(info);
            const uint64_t e = (first - st.sample_base + info_bs(info)) * (st.plane_stride ? 1u : st.channels) * st.bytes_ps;
            end = e > end ? e : end;
        } else if (bad == 0xFFFFFFFFu) {
            bad = r;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        frames += __shfl_xor(frames, d);
        ok += __shfl_xor(ok, d);
        samples += __shfl_xor(samples, d);
        const uint64_t o = __shfl_xor(end, d);
        end = o > end ? o : end;
        bad = min(bad, (uint32_t)__shfl_xor((int)bad, d));
    }
    // the first bad frame's index counts the frames in front of it, not the candidates
    uint64_t before = 0;
    int32_t bad_status = 0;
    if (bad != 0xFFFFFFFFu) {
        for (uint32_t r = st.cand_begin + lane; r < bad; r += 64) before += c.status[r] != kFrameSkipped ? 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
        bad_status = c.status[bad];
    }
    if (lane == 0) {
        ResultDev res;
        res.header_candidates = st.cand_end - st.cand_begin;
        res.frames = frames; res.frames_ok = ok; res.samples = samples; res.pcm_bytes = end;
        res.first_bad_frame = bad != 0xFFFFFFFFu ? before : ~0ull;
        res.first_error = bad_status;
        res.pad = 0;
        out[s] = res;
    }
}

}  // namespace flacgpu
