// subframe.hpp — subframe kernels (gfx950): bit reader, Rice codes, the skim
// pass that finds where each subframe starts, and the decode pass that
// reconstructs, decorrelates and packs PCM.
//
// Work decomposition.  Inside a frame the bit position of everything depends on
// what came before, so a frame is walked serially; frames are independent.
//   k_skim     one lane per frame.  Walks subframes 0..C-2 without reconstructing
//              anything (headers, partition parameters, and for each Rice code
//              only its length) and records the bit position where every
//              subframe starts.
//   k_decode   one lane per subframe (frame f, channel c); the lanes of a wave
//              advance sample by sample in lockstep, so prediction history
//              lives in registers, stereo decorrelation is a lane exchange, and
//              the wave stages 32 samples of all its frames in LDS and writes
//              them out as contiguous runs per frame.
// Arithmetic is 32-bit except LPC sums when bps + precision + floor(log2(order))
// exceeds 32, which accumulate in 64 bits.
#pragma once

#include "device_types.hpp"

namespace flacgpu {

// -------------------------------------------------------------- bit reader --

// MSB-first reader over one frame.  Positions are bits from the 4-byte-aligned
// word at or before the frame's first byte.  Bytes past the end of the frame
// read as zero.
struct BitReader {
    const uint32_t *w;
    uint32_t full_words;   // words entirely inside [base, frame end)
    uint32_t tail_bytes;   // valid bytes in word `full_words`
    uint32_t idx, bit, hi, lo;

    __device__ inline uint32_t load(uint32_t i) const
    {
        if (i < full_words) return __builtin_bswap32(w[i]);
        if (i == full_words && tail_bytes) {
            const uint8_t *p = reinterpret_cast<const uint8_t *>(w + i);
            uint32_t v = 0;
            for (uint32_t k = 0; k < tail_bytes; ++k) v |= (uint32_t)p[k] << (24 - 8 * k);
            return v;
        }
        return 0;
    }
    __device__ inline void init(const uint8_t *in, uint64_t goff, uint32_t frame_len)
    {
        const uint32_t a = (uint32_t)(goff & 3);
        w = reinterpret_cast<const uint32_t *>(in + (goff - a));
        full_words = (a + frame_len) >> 2;
        tail_bytes = (a + frame_len) & 3;
        idx = bit = hi = lo = 0;
    }
    __device__ inline void seek(uint32_t bitpos)
    {
        idx = bitpos >> 5; bit = bitpos & 31;
        hi = load(idx); lo = load(idx + 1);
    }
    __device__ inline uint32_t pos() const { return idx * 32 + bit; }
    __device__ inline uint32_t peek() const
    {
        return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (32 - bit));
    }
    __device__ inline void skip(uint32_t n)   // n <= 32
    {
        bit += n;
        if (bit >= 32) { bit -= 32; ++idx; hi = lo; lo = load(idx + 1); }
    }
    __device__ inline void skip_long(uint32_t n) { seek(pos() + n); }
    __device__ inline uint32_t read(uint32_t n)   // n <= 32
    {
        const uint32_t v = n ? peek() >> (32 - n) : 0;
        skip(n);
        return v;
    }
    __device__ inline int32_t read_signed(uint32_t n)
    {
        const int32_t v = n ? (int32_t)peek() >> (32 - n) : 0;
        skip(n);
        return v;
    }
};

// Unary-coded count: zeros up to and including the terminating one are consumed.
// limit: bit position past which the frame has ended (guards the all-zero tail).
__device__ inline uint32_t read_unary(BitReader &br, uint32_t limit)
{
    uint32_t q = 0, w = br.peek();
    while (w == 0) {
        if (br.pos() > limit) return q;
        q += 32; br.skip(32); w = br.peek();
    }
    const uint32_t z = __clz(w);
    br.skip(z + 1);
    return q + z;
}

__device__ inline int32_t unzigzag(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1))); }

// One Rice code with parameter k (0..30).
__device__ inline int32_t read_rice(BitReader &br, uint32_t k, uint32_t limit)
{
    const uint32_t w = br.peek();
    uint32_t q, low;
    if (w != 0) {
        q = __clz(w);
        const uint32_t n = q + 1 + k;
        if (n <= 32) {
            low = k ? (w << (q + 1)) >> (32 - k) : 0;
            br.skip(n);
        } else {
            br.skip(q + 1);
            low = br.read(k);
        }
    } else {
        q = read_unary(br, limit);
        low = br.read(k);
    }
    return unzigzag((q << k) | low);
}

// Length-only version for the skim pass.
__device__ inline void skip_rice(BitReader &br, uint32_t k, uint32_t limit)
{
    const uint32_t w = br.peek();
    if (w != 0) {
        const uint32_t n = __clz(w) + 1 + k;
        if (n <= 32) br.skip(n);
        else { br.skip(n - k); br.skip(k); }
    } else {
        (void)read_unary(br, limit);
        br.skip(k);
    }
}

// ---------------------------------------------------------- subframe header --

struct SubHdr {
    uint32_t kind;     // 0 constant, 1 verbatim, 2 predicted (fixed or LPC)
    uint32_t order;
    uint32_t lpc;      // 1 when coefficients are in the stream
    uint32_t wasted;
    uint32_t sbps;     // sample width after side-channel bit and wasted bits
};

// Returns a kFrame* status.
__device__ inline int read_subframe_header(BitReader &br, uint32_t bps, uint32_t bs, uint32_t limit, SubHdr &h)
{
    const uint32_t hdr = br.read(8);
    if (hdr & 0x80) return kFrameBadSubframe;
    const uint32_t type = (hdr >> 1) & 0x3F;
    h.wasted = 0;
    if (hdr & 1) h.wasted = read_unary(br, limit) + 1;
    if (br.pos() > limit) return kFrameTruncated;
    if (h.wasted >= bps) return kFrameBadSubframe;
    h.sbps = bps - h.wasted;
    if (h.sbps > 32) return kFrameUnsupported;
    h.order = 0; h.lpc = 0;
    if (type == 0) h.kind = 0;
    else if (type == 1) h.kind = 1;
    else if (type >= 8 && type <= 12) { h.kind = 2; h.order = type - 8; }
    else if (type >= 32) { h.kind = 2; h.order = type - 31; h.lpc = 1; }
    else return kFrameBadSubframe;
    if (h.order > bs) return kFrameBadSubframe;
    return kFrameOK;
}

__device__ inline bool is_side(uint32_t ch_assign, uint32_t c)
{
    return (ch_assign == 8 && c == 1) || (ch_assign == 9 && c == 0) || (ch_assign == 10 && c == 1);
}

// -------------------------------------------------------------------- skim --

// Walks one whole subframe without reconstructing it.  Returns a kFrame* status.
__device__ inline int skim_subframe(BitReader &br, uint32_t bps, uint32_t bs, uint32_t limit)
{
    SubHdr h;
    int rc = read_subframe_header(br, bps, bs, limit, h);
    if (rc) return rc;
    if (h.kind == 0) { br.skip(h.sbps); return br.pos() > limit ? kFrameTruncated : kFrameOK; }
    if (h.kind == 1) {
        const uint64_t n = (uint64_t)bs * h.sbps;
        if (br.pos() + n > limit) return kFrameTruncated;
        br.skip_long((uint32_t)n);
        return kFrameOK;
    }
    if (br.pos() + (uint64_t)h.order * h.sbps > limit) return kFrameTruncated;
    br.skip_long(h.order * h.sbps);
    if (h.lpc) {
        const uint32_t pc = br.read(4);
        if (pc == 15) return kFrameBadSubframe;
        const int32_t shift = br.read_signed(5);
        if (shift < 0) return kFrameBadSubframe;
        br.skip_long(h.order * (pc + 1));
    }
    const uint32_t method = br.read(2);
    if (method > 1) return kFrameBadSubframe;
    const uint32_t pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const uint32_t po = br.read(4);
    if (br.pos() > limit) return kFrameTruncated;
    if (bs & ((1u << po) - 1)) return kFrameBadSubframe;
    const uint32_t psamp = bs >> po;
    if (psamp < h.order) return kFrameBadSubframe;
    uint32_t parts_left = 1u << po, part_rem = 0, k = 0;
    uint32_t first = 1;
    for (;;) {
        if (part_rem == 0) {
            if (br.pos() > limit) return kFrameTruncated;
            if (parts_left == 0) break;
            --parts_left;
            uint32_t cnt = psamp - (first ? h.order : 0);
            first = 0;
            k = br.read(pbits);
            if (k == esc) {
                const uint32_t wd = br.read(5);
                const uint64_t n = (uint64_t)cnt * wd;
                if (br.pos() + n > limit) return kFrameTruncated;
                br.skip_long((uint32_t)n);
                cnt = 0;
            }
            part_rem = cnt;
        } else {
            skip_rice(br, k, limit);
            --part_rem;
        }
    }
    return kFrameOK;
}

// One lane per candidate frame r in [r_begin, r_end); C channels.  Records
// subpos[r*8 + c] for every channel and turns frames that fail to parse into
// their error status.
__global__ __launch_bounds__(256) void k_skim(const uint8_t *__restrict__ in, CandArrays c, uint32_t r_begin,
                                              uint32_t r_end, uint32_t C)
{
    const uint32_t r = r_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= r_end) return;
    if (c.status[r] != kFrameOK) return;
    const uint32_t info = c.info[r];
    const uint32_t bs = info_bs(info), bps = info_bps(info), ca = info_ca(info);
    const uint64_t goff = c.goff[r];
    const uint32_t len = c.len[r];
    BitReader br;
    br.init(in, goff, len);
    const uint32_t a = (uint32_t)(goff & 3);
    const uint32_t limit = (a + len) * 8;   // first bit after the frame
    br.seek((a + info_hl(info)) * 8);
    int rc = kFrameOK;
    for (uint32_t ch = 0; ch < C; ++ch) {
        c.subpos[(uint64_t)r * 8 + ch] = br.pos();
        if (ch + 1 == C) break;   // the last subframe is checked by its decode lane
        rc = skim_subframe(br, bps + (is_side(ca, ch) ? 1 : 0), bs, limit);
        if (rc) break;
    }
    if (rc) c.status[r] = rc;
}

// ------------------------------------------------------------------ decode --

constexpr int kTile = 32;   // samples staged in LDS between write-outs

struct DecodeParams {
    const uint8_t *in;
    uint8_t *out;
    const StreamDev *streams;
    CandArrays c;
    uint32_t r_begin, r_end;
    uint32_t C;            // channels of every frame in [r_begin, r_end)
    uint32_t B;            // bytes per sample in the output
};

// The sample loop of one wave, specialised on the largest predictor order among
// its lanes (MAXORD) and on whether any lane needs 64-bit sums (WIDE).
template <int MAXORD, bool WIDE>
__device__ inline void decode_loop(BitReader &br, const SubHdr &h, bool active, bool dwords, uint32_t bs,
                                   uint32_t bs_max, uint32_t limit, uint32_t ca, uint32_t ch, uint32_t C, uint32_t B,
                                   uint32_t frame_in_wave, uint32_t frames_in_wave, uint8_t *tile, uint32_t row_stride,
                                   uint8_t *dst_row /* this frame's output */, const uint64_t *s_dst,
                                   const uint32_t *s_bs, int &status)
{
    int32_t coef[MAXORD], hist[MAXORD];
#pragma unroll
    for (int j = 0; j < MAXORD; ++j) { coef[j] = 0; hist[j] = 0; }
    uint32_t shift = 0;
    bool wide = false;
    uint32_t kind = active ? h.kind : 0;
    const uint32_t order = h.order;
    int32_t constval = 0;
    if (active && kind == 0) constval = br.read_signed(h.sbps);
    // residual partition state
    uint32_t pbits = 4, esc = 15, psamp = 0, parts_left = 0, part_rem = 0, k = 0, escw = 0;
    bool escaped = false;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t row_bytes = kTile * C * B;

    bool first_part = true;
    // Reads what follows the warm-up samples: LPC parameters (or the fixed
    // polynomial as coefficients) and the residual coding header.
    auto start_residual = [&]() {
        if (h.lpc) {
            const uint32_t pc = br.read(4);
            const int32_t sh = br.read_signed(5);
            if (pc == 15 || sh < 0) status = kFrameBadSubframe;
            shift = (uint32_t)sh & 31;
            const uint32_t prec = pc + 1;
#pragma unroll
            for (int j = 0; j < MAXORD; ++j)
                if ((uint32_t)j < order) coef[j] = br.read_signed(prec);
            wide = (h.sbps + prec + (31 - __clz(order))) > 32;
        } else {
            if (order == 1) { coef[0] = 1; }
            if (order == 2) { coef[0] = 2; coef[1] = -1; }
            if (order == 3) { coef[0] = 3; coef[1] = -3; coef[2] = 1; }
            if (order == 4) { coef[0] = 4; coef[1] = -6; coef[2] = 4; coef[3] = -1; }
        }
        const uint32_t method = br.read(2);
        const uint32_t po = br.read(4);
        if (method > 1) status = kFrameBadSubframe;
        pbits = method ? 5 : 4; esc = method ? 31 : 15;
        psamp = bs >> po;
        if ((bs & ((1u << po) - 1)) || psamp < order) status = kFrameBadSubframe;
        parts_left = 1u << po;
        part_rem = 0;
        if (br.pos() > limit && status == kFrameOK) status = kFrameTruncated;
    };
    // Opens partitions until one holds a sample (or none is left).
    auto next_partition = [&]() {
        while (part_rem == 0 && parts_left && status == kFrameOK) {
            part_rem = psamp - (first_part ? order : 0);
            first_part = false;
            --parts_left;
            k = br.read(pbits);
            escaped = (k == esc);
            if (escaped) escw = br.read(5);
            if (br.pos() > limit) status = kFrameTruncated;
        }
    };

    for (uint32_t n0 = 0; n0 < bs_max; n0 += kTile) {
        const uint32_t n1 = min(n0 + (uint32_t)kTile, bs_max);
        for (uint32_t n = n0; n < n1; ++n) {
            const bool live = active && n < bs && status == kFrameOK;
            int32_t v = constval;
            bool predicted = false;
            if (live && kind != 0) {
                if (kind == 1 || n < order) {
                    v = br.read_signed(h.sbps);
                } else {
                    if (n == order) start_residual();
                    if (part_rem == 0) next_partition();
                    if (status == kFrameOK && part_rem == 0) status = kFrameTruncated;
                    if (status == kFrameOK) {
                        --part_rem;
                        v = escaped ? br.read_signed(escw) : read_rice(br, k, limit);
                        predicted = true;
                    }
                }
            }
            // prediction: every lane evaluates the full MAXORD-tap sum; unused taps are zero
            int32_t pred;
            if (WIDE) {
                int64_t sum = 0;
#pragma unroll
                for (int j = 0; j < MAXORD; ++j) sum += (int64_t)coef[j] * (int64_t)hist[j];
                pred = wide ? (int32_t)(sum >> shift) : ((int32_t)(uint32_t)sum) >> shift;
            } else {
                uint32_t sum = 0;
#pragma unroll
                for (int j = 0; j < MAXORD; ++j) sum += (uint32_t)coef[j] * (uint32_t)hist[j];
                pred = (int32_t)sum >> shift;
            }
            if (predicted) v = (int32_t)((uint32_t)v + (uint32_t)pred);
#pragma unroll
            for (int j = MAXORD - 1; j > 0; --j) hist[j] = hist[j - 1];
            hist[0] = v;

            uint32_t o = (uint32_t)v << h.wasted;
            if (C == 2) {
                const uint32_t p = (uint32_t)__shfl_xor((int)o, 1);
                if (ca == 8) { if (ch == 1) o = p - o; }
                else if (ca == 9) { if (ch == 0) o = o + p; }
                else if (ca == 10) {
                    const uint32_t m = ch == 0 ? o : p, s = ch == 0 ? p : o;
                    const uint32_t mm = (m << 1) | (s & 1);
                    o = ch == 0 ? (uint32_t)((int32_t)(mm + s) >> 1) : (uint32_t)((int32_t)(mm - s) >> 1);
                }
            }
            if (live) {
                uint8_t *t = tile + frame_in_wave * row_stride + ((n - n0) * C + ch) * B;
                if (B == 2) *reinterpret_cast<uint16_t *>(t) = (uint16_t)o;
                else if (B == 4) *reinterpret_cast<uint32_t *>(t) = o;
                else if (B == 3) { t[0] = (uint8_t)o; t[1] = (uint8_t)(o >> 8); t[2] = (uint8_t)(o >> 16); }
                else t[0] = (uint8_t)o;
            }
        }
        // write the staged samples of every frame of this wave: rows of up to kTile*C*B bytes
        __builtin_amdgcn_wave_barrier();
        if (dwords) {
            const uint32_t row_words = row_bytes >> 2;
            for (uint32_t i = lane; i < frames_in_wave * row_words; i += 64) {
                const uint32_t f = i / row_words, col = i - f * row_words;
                const uint32_t fbs = s_bs[f];
                const uint32_t valid = fbs > n0 ? min((uint32_t)kTile, fbs - n0) * C * B : 0;
                if (col * 4 < valid) {
                    const uint32_t val = *reinterpret_cast<const uint32_t *>(tile + f * row_stride + col * 4);
                    *reinterpret_cast<uint32_t *>(s_dst[f] + (uint64_t)n0 * C * B + col * 4) = val;
                }
            }
        } else {
            for (uint32_t i = lane; i < frames_in_wave * row_bytes; i += 64) {
                const uint32_t f = i / row_bytes, col = i - f * row_bytes;
                const uint32_t fbs = s_bs[f];
                const uint32_t valid = fbs > n0 ? min((uint32_t)kTile, fbs - n0) * C * B : 0;
                if (col < valid) *reinterpret_cast<uint8_t *>(s_dst[f] + (uint64_t)n0 * C * B + col) = tile[f * row_stride + col];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // a predicted subframe made only of warm-up samples still carries its parameters
    if (active && status == kFrameOK && kind == 2 && order >= bs) { start_residual(); next_partition(); }
    (void)dst_row;
}

__device__ inline uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}

// One lane per (frame, channel); a wave holds floor(64 / C) frames.
__global__ __launch_bounds__(256) void k_decode(DecodeParams p)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t C = p.C, B = p.B;
    const uint32_t lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    const uint32_t F = 64 / C;                      // frames per wave
    const uint32_t row_stride = kTile * C * B + 4;  // +4: rows start on different LDS banks
    const uint32_t wave_bytes = ((F * row_stride + 15) & ~15u) + F * 16;
    uint8_t *tile = smem + wave_in_block * wave_bytes;
    uint64_t *s_dst = reinterpret_cast<uint64_t *>(tile + ((F * row_stride + 15) & ~15u));
    uint32_t *s_bs = reinterpret_cast<uint32_t *>(s_dst + F);

    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const uint32_t fl = lane / C, ch = lane - fl * C;
    const uint32_t r = p.r_begin + wave * F + fl;
    bool active = fl < F && r < p.r_end;
    int status = kFrameOK;
    if (active) status = p.c.status[r];
    active = active && status == kFrameOK;

    uint32_t bs = 0, ca = 0, limit = 0;
    SubHdr h;
    h.kind = 0; h.order = 0; h.lpc = 0; h.wasted = 0; h.sbps = 0;
    BitReader br;
    br.w = nullptr; br.full_words = 0; br.tail_bytes = 0; br.idx = br.bit = br.hi = br.lo = 0;
    uint64_t dst = 0;
    if (active) {
        const uint32_t info = p.c.info[r];
        const uint64_t goff = p.c.goff[r];
        const uint32_t len = p.c.len[r];
        const StreamDev &s = p.streams[p.c.stream[r]];
        bs = info_bs(info); ca = info_ca(info);
        const uint32_t a = (uint32_t)(goff & 3);
        limit = (a + len) * 8;
        br.init(p.in, goff, len);
        br.seek(p.c.subpos[(uint64_t)r * 8 + ch]);
        const uint64_t first = info_var(info) ? p.c.num[r] : p.c.num[r] * (uint64_t)s.max_bs;
        dst = reinterpret_cast<uint64_t>(p.out) + s.out_off + first * C * B;
        const int rc = read_subframe_header(br, info_bps(info) + (is_side(ca, ch) ? 1 : 0), bs, limit, h);
        if (rc) { status = rc; h.kind = 0; h.order = 0; }
    }
    if (fl < F && ch == 0) { s_dst[fl] = dst; s_bs[fl] = active ? bs : 0; }
    // predictor precision is stored after the warm-up samples: look ahead for the accumulator width
    bool need_wide = false;
    if (active && status == kFrameOK && h.kind == 2 && h.lpc) {
        BitReader t = br;
        t.skip_long(h.order * h.sbps);
        const uint32_t prec = t.read(4) + 1;
        need_wide = (h.sbps + prec + (31 - __clz(h.order))) > 32;
    }
    const uint32_t maxord = wave_max_u32(active && status == kFrameOK ? h.order : 0);
    const uint32_t bs_max = wave_max_u32(active ? bs : 0);
    const bool any_wide = __any(need_wide);
    // whole-dword write-out needs every frame's destination 4-byte aligned
    const bool dwords = ((C * B) & 3) == 0 && !__any(active && (dst & 3));
    __builtin_amdgcn_wave_barrier();

#define FLACGPU_DECODE(MO, W)                                                                                       \
    decode_loop<MO, W>(br, h, active, dwords, bs, bs_max, limit, ca, ch, C, B, fl, F, tile, row_stride, nullptr, \
                       s_dst, s_bs, status)
    if (!any_wide) {
        if (maxord <= 4) FLACGPU_DECODE(4, false);
        else if (maxord <= 8) FLACGPU_DECODE(8, false);
        else if (maxord <= 12) FLACGPU_DECODE(12, false);
        else FLACGPU_DECODE(32, false);
    } else {
        if (maxord <= 12) FLACGPU_DECODE(12, true);
        else FLACGPU_DECODE(32, true);
    }
#undef FLACGPU_DECODE

    if (active) {
        if (status == kFrameOK) {
            // a subframe ends where the next begins; the last one ends at the CRC-16, byte aligned
            const uint32_t end = br.pos();
            if (ch + 1 == C) {
                const uint32_t want = limit - 16;
                if (end > limit) status = kFrameTruncated;
                else if (((end + 7) & ~7u) != want) status = kFrameLength;
            } else if (end != p.c.subpos[(uint64_t)r * 8 + ch + 1]) {
                status = end > limit ? kFrameTruncated : kFrameLength;
            }
        }
        if (status != kFrameOK) atomicMax(&p.c.status[r], status);
    }
}

}  // namespace flacgpu
