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
//              lives in registers (renamed, not moved: the sample loop is
//              unrolled by the predictor length), stereo decorrelation is a lane
//              exchange, and the wave stages a tile of samples of all its frames
//              in LDS and writes it out as contiguous runs per frame.
//              The kernel is instantiated per predictor length / accumulator
//              width so that short predictors do not pay the registers of long
//              ones; every instantiation is launched over the same grid and a
//              wave runs in exactly one of them.
// Arithmetic is 32-bit except LPC sums when bps + precision + floor(log2(order))
// exceeds 32, which accumulate in 64 bits.
#pragma once

#include <type_traits>

#include "device_types.hpp"
#include "variants.hpp"

namespace flacgpu {

// -------------------------------------------------------------- bit reader --

// Every lane walks its own bitstream, so a load per code would touch 64 different
// cache lines per wave instruction and use 4 bytes of each.  Instead each lane
// owns a small ring in LDS that it refills 32 bytes (one sector) at a time
// (ring_round).  A refill is issued one round (8 decode steps) before it is
// written to the ring, so its latency overlaps decoding.
constexpr uint32_t kRingLog2 = variant::kRingLog2;
constexpr uint32_t kRingWords = 1u << kRingLog2;   // ring capacity per lane, in 32-bit words
constexpr uint32_t kRingSlots = kRingWords + 1;    // the last slot repeats slot 0, so that words w and w + 1 are always one two-word read
constexpr uint32_t kUnitWords = variant::kUnitWords;   // refill granule: 32 bytes
constexpr uint32_t kRoundSteps = variant::kRoundSteps;   // decode steps (of at most 32 bits) between refill rounds
constexpr uint32_t kRingBytesPerWave = kRingSlots * 64 * 4;
constexpr uint32_t kNoFill = 0xFFFFFFFFu;
// A lane fetches its next unit when it has this many unread words or fewer.  The unit lands one round later
// on the ring's oldest words, which the reader must have passed by then: at most kRingWords - kUnitWords.
constexpr uint32_t kFetchBelow = kRingWords >= 32 ? kRingWords - kUnitWords - 7 : kRingWords - kUnitWords;
constexpr uint32_t kSkimPairK = variant::kSkimPairK;
// Compile-time loops: the body gets its index as a std::integral_constant (the unrolled loops index register arrays).
template <int N, int I = 0, class F>
__device__ inline void static_for_each(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_each<N, I + 1>(f); }
}


// a * b + c on the low 24 bits of a and b (sign-extended), low 32 bits of the result: one instruction
// where a 32-bit multiply and an add are two (both multiplies issue at the full rate on gfx950).  Exact
// whenever a and b fit 24 bits.  Written as asm because the compiler only selects it when it can see
// both sign-extensions in the same block.
__device__ inline int32_t mad_i24(int32_t a, int32_t b, int32_t c)
{
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Two 16 x 16-bit products and an addend in one instruction: lo(a) * lo(b) + hi(a) * hi(b) + c, halves taken as
// signed, sum wrapping in 32 bits.  Issues at the rate of the 24-bit multiply-add, so a predictor whose samples
// and coefficients fit 16 bits costs half the multiplies.
__device__ inline int32_t dot2_i16(uint32_t a, uint32_t b, int32_t c)
{
    int32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ inline int32_t dot2_i16(uint32_t a, uint32_t b)
{
    int32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (lo, hi) -> lo's low half in bits 0-15, hi's low half in bits 16-31
__device__ inline uint32_t pack_lo_hi(int32_t lo, int32_t hi)
{
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}

// MSB-first reader over one lane's bitstream.  Its whole state is a bit position P, counted from the
// 64-byte-aligned address at or before the frame's first byte.  Stream word w of the lane sits at
// ring[(w % kRingWords) * 64] (ring already points at the lane's column), byte order corrected; the ring
// holds words [fill - kRingWords, fill).  A word the ring does not hold is read from global memory
// directly (after a seek, or when a lane outruns the refill schedule).  Reads never go past the end of the
// input buffer (a lane that runs off the end of the batch sees zeros or a repeat of the last 16 bytes, and
// stops at its frame's bit limit).
struct BitReader {
    const uint32_t *gbase;   // global words from the aligned base
    const uint4 *spare;      // 32 readable bytes of device memory, the same for every lane
    uint32_t nwords;         // words readable from gbase
    uint32_t *ring;
    uint32_t fill;
    uint32_t P;              // bit position

    // Direct read of stream word w.  Only cold paths use it.  The load and its wait are one asm statement
    // on purpose: a compiler-visible load in a rarely taken branch makes hipcc wait for ALL outstanding
    // memory operations wherever the paths join, i.e. once per sample in the hot loop, which would
    // serialise the refill prefetches.
    __device__ inline uint32_t gload(uint32_t w) const
    {
        if (w >= nwords) return 0u;
        uint32_t v;
        const uint32_t *p = gbase + w;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return __builtin_bswap32(v);
    }
    __device__ inline uint32_t word(uint32_t w) const
    {
        if (w < fill) return ring[(w & (kRingWords - 1)) * 64];
        return gload(w);
    }
    // spare: 32 readable bytes of device memory, loaded (and ignored) by lanes that have nothing to fetch
    // (all from the same address: one cache line per wave instruction, not 64)
    __device__ inline void init(const uint8_t *in, uint64_t in_vecs, uint64_t goff, uint32_t *lane_ring, const uint4 *spare)
    {
        const uint64_t b = goff & ~(uint64_t)63;
        gbase = reinterpret_cast<const uint32_t *>(in + b);
        nwords = (uint32_t)min((in_vecs - (b >> 4)) * 4, (uint64_t)0xFFFFFFFFu);
        this->spare = spare;
        ring = lane_ring;
        fill = P = 0;
    }
    __device__ inline void null(uint32_t *lane_ring, const uint4 *spare)
    {
        gbase = nullptr; this->spare = spare; nwords = 0; ring = lane_ring;
        fill = P = 0;
    }
    __device__ inline void seek(uint32_t bitpos)
    {
        P = bitpos;
        fill = (P >> 5) & ~(kUnitWords - 1);   // ring restarts at this unit; until it is refilled, words come from global memory
    }
    __device__ inline uint32_t pos() const { return P; }
    // The 32 bits at P for the speculative loops, which assume both words are in the ring (and test
    // `starved` afterwards) and keep the position negated: Q = -P (mod 2^32).  Bit P sits in
    // word (P - 1) / 32 at offset ((P - 1) % 32) + 1, counted 1 .. 32, so the window is that word and the
    // next funnel-shifted right by (32 - offset) = Q % 32: one instruction, and no special case when P is a
    // multiple of 32.  The word's ring slot is 31 - (Q / 32) % 32.
    __device__ inline uint32_t peek_ring_neg(uint32_t Q) const
    {
        typedef __attribute__((address_space(3))) const uint32_t lds_word;
        const uint32_t top = (uint32_t)(uintptr_t)(lds_word *)(ring + (kRingWords - 1) * 64);   // LDS byte address of slot 31
        const uint32_t idx = __builtin_amdgcn_ubfe(Q, 5, kRingLog2);
        lds_word *q = (lds_word *)(uintptr_t)mad_i24((int32_t)idx, -256, (int32_t)top);
        return __builtin_amdgcn_alignbit(q[0], q[64], Q);
    }
    // The same in two halves, for loops that fetch the window of the next code before they finish the current one:
    // the two ring words (no wait), and the funnel shift that needs them.
    __device__ inline uint2 fetch_ring_neg(uint32_t Q) const
    {
        typedef __attribute__((address_space(3))) const uint32_t lds_word;
        const uint32_t top = (uint32_t)(uintptr_t)(lds_word *)(ring + (kRingWords - 1) * 64);
        const uint32_t idx = __builtin_amdgcn_ubfe(Q, 5, kRingLog2);
        lds_word *q = (lds_word *)(uintptr_t)mad_i24((int32_t)idx, -256, (int32_t)top);
        return uint2{q[0], q[64]};
    }
    static __device__ inline uint32_t window_neg(uint2 words, uint32_t Q) { return __builtin_amdgcn_alignbit(words.x, words.y, Q); }
    // The 64 bits at P, for a ring with two mirror slots (see ring_round<true>): high and low word.
    __device__ inline uint2 peek64_ring_neg(uint32_t Q) const
    {
        typedef __attribute__((address_space(3))) const uint32_t lds_word;
        const uint32_t top = (uint32_t)(uintptr_t)(lds_word *)(ring + (kRingWords - 1) * 64);
        const uint32_t idx = __builtin_amdgcn_ubfe(Q, 5, kRingLog2);
        lds_word *q = (lds_word *)(uintptr_t)mad_i24((int32_t)idx, -256, (int32_t)top);
        const uint32_t a = q[0], b = q[64], c = q[128];
        return uint2{__builtin_amdgcn_alignbit(a, b, Q), __builtin_amdgcn_alignbit(b, c, Q)};
    }
    __device__ inline bool starved() const { return (P >> 5) + 1 >= fill; }
    __device__ inline uint32_t peek() const
    {
        const uint32_t w = P >> 5;
        const uint64_t win = (((uint64_t)word(w)) << 32) | word(w + 1);
        return (uint32_t)((win << (P & 31)) >> 32);
    }
    __device__ inline void skip(uint32_t n) { P += n; }
    __device__ inline void skip_long(uint32_t n) { seek(P + n); }
    __device__ inline uint32_t read(uint32_t n)   // n <= 32
    {
        const uint32_t v = n ? peek() >> (32 - n) : 0;
        P += n;
        return v;
    }
    __device__ inline int32_t read_signed(uint32_t n)
    {
        const int32_t v = n ? (int32_t)peek() >> (32 - n) : 0;
        P += n;
        return v;
    }
};

// What a lane holds between two refill rounds: the 32 bytes it loaded for its ring and the ring
// position they belong to.
struct RingFetch {
    uint4 v[kUnitWords / 4];
    uint32_t at;   // `fill` when the load was issued, or kNoFill
    __device__ inline void clear()
    {
        at = kNoFill;
#pragma unroll
        for (uint32_t i = 0; i < kUnitWords / 4; ++i) v[i] = uint4{0u, 0u, 0u, 0u};
    }
};

// One refill round; all 64 lanes call it together, every kRoundSteps decode steps.
//   1. commit: the 32 bytes loaded in the previous round are written to the ring (unless
//      the reader has been repositioned since);
//   2. issue: a lane with 17 or fewer unread words loads its next 32 bytes, to be
//      committed in the next round.  Only whole 32-byte units inside the input are fetched:
//      the last few bytes of the input buffer, when they do not fill one, are read word by
//      word (BitReader::gload).
// A lane that starts a round with at least 10 unread words has the two words of its window in
// the ring until the next round, whatever it consumes in between (at most 8 words), so the
// decode loops never wait for memory.  After repositioning, three rounds in a row restore the margin.
// Each lane loads for itself: 64 lanes touch 64 different lines either way, and the two 16-byte
// loads of a lane fall into one 32-byte sector.
template <bool MIRROR2 = false>   // MIRROR2: the ring has a second extra slot that repeats slot 1 (three-word windows)
__device__ inline void ring_round(BitReader &br, RingFetch &rf, bool active)
{
    // ---- commit
    if (rf.at == br.fill) {   // a valid `fill` is never kNoFill
        const uint32_t slot = br.fill & (kRingWords - 1);   // a multiple of 8
        uint32_t *dst = br.ring + slot * 64;
        const uint32_t w0 = __builtin_bswap32(rf.v[0].x);
        dst[0 * 64] = w0;
        dst[1 * 64] = __builtin_bswap32(rf.v[0].y);
        dst[2 * 64] = __builtin_bswap32(rf.v[0].z);
        dst[3 * 64] = __builtin_bswap32(rf.v[0].w);
#pragma unroll
        for (uint32_t i = 1; i < kUnitWords / 4; ++i) {
            dst[(4 * i + 0) * 64] = __builtin_bswap32(rf.v[i].x);
            dst[(4 * i + 1) * 64] = __builtin_bswap32(rf.v[i].y);
            dst[(4 * i + 2) * 64] = __builtin_bswap32(rf.v[i].z);
            dst[(4 * i + 3) * 64] = __builtin_bswap32(rf.v[i].w);
        }
        if (slot == 0) {
            br.ring[kRingWords * 64] = w0;
            if (MIRROR2) br.ring[(kRingWords + 1) * 64] = __builtin_bswap32(rf.v[0].y);
        }
        br.fill += kUnitWords;
    }
    // ---- issue
    const uint32_t rdw = br.P >> 5;
    if (active && (int32_t)(br.fill - rdw) < -(int32_t)kUnitWords + 1) br.fill = rdw & ~(kUnitWords - 1);
    const bool want = active && (int32_t)(br.fill - rdw) <= (int32_t)kFetchBelow && br.fill + kUnitWords <= br.nwords;
    rf.at = want ? br.fill : kNoFill;
    // the loads are unconditional (a lane with nothing to fetch reads the spare bytes): a branch
    // around them would make the compiler copy the result, and so wait for it, right here
    const uint4 *src = want ? reinterpret_cast<const uint4 *>(br.gbase + br.fill) : br.spare;
#pragma unroll
    for (uint32_t i = 0; i < kUnitWords / 4; ++i) rf.v[i] = src[want ? i : (i & 1)];   // (the spare bytes are 32)
}

// Unary-coded count: zeros up to and including the terminating one are consumed.
// limit: bit position past which the frame has ended (bounds an all-zero run).
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

// Leading zeros of w; 0xFFFFFFFF when w is zero (the hardware's answer, without the clamp to 32 that
// __clz adds).  The speculative loops keep the largest count they saw and test it once per group.
__device__ inline uint32_t lead_zeros_raw(uint32_t w)
{
    uint32_t q;
    asm("v_ffbh_u32 %0, %1" : "=v"(q) : "v"(w));
    return q;
}

// `width` bits of w from bit (offset mod 32) up: the instruction takes the offset's low five bits itself.
__device__ inline uint32_t bfe_mod32(uint32_t w, uint32_t offset, uint32_t width)
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "v"(offset), "v"(width));
    return r;
}

__device__ inline int32_t unzigzag(uint32_t u) { return (int32_t)((u >> 1) ^ (0u - (u & 1))); }

// One Rice code with parameter k (0..30), general form.
__device__ inline int32_t read_rice(BitReader &br, uint32_t k, uint32_t limit)
{
    const uint32_t w = br.peek();
    uint32_t q, low;
    const uint32_t n = __clz(w) + 1 + k;   // clz(0) = 32, so n > 32 when the window is all zeros
    if (n <= 32) {
        q = n - 1 - k;
        low = __builtin_amdgcn_ubfe(w, 32 - n, k);
        br.skip(n);
    } else {
        q = read_unary(br, limit);
        low = br.read(k);
    }
    return unzigzag((q << k) | low);
}

// Length-only version for the skim pass.
__device__ inline void skip_rice(BitReader &br, uint32_t k, uint32_t limit)
{
    const uint32_t n = __clz(br.peek()) + 1 + k;
    if (n <= 32) br.skip(n);
    else {
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

__device__ inline uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}

// -------------------------------------------------------------------- skim --

// One lane per candidate frame of `map` (the frames of all streams with C channels).  Records
// subpos[r*8 + c] for every channel and turns frames that fail to parse into
// their error status.  Subframes 0 .. C-2 are walked one after the other; the
// last one is checked by its decode lane.
//
// The lanes of a wave move through a subframe's residual in step, by sample
// index, in groups of 16 samples: partition boundaries are multiples of the
// partition size in every lane, so the lanes reach them together, at group
// boundaries.  A group in which every lane is inside a Rice partition runs
// branch-free and speculatively, two codes per 32-bit window when the Rice
// parameters are small: a lane whose codes did not fit (or whose ring ran dry)
// is rolled back and walks the group code by code.
__global__ __launch_bounds__(256) void k_skim(const uint8_t *__restrict__ in, uint64_t in_vecs, CandArrays c,
                                              FrameMap map, uint32_t C, const uint4 *spare)
{
    constexpr uint32_t kSlots = kRingSlots + 1;   // a second mirror slot: the wide-window loop reads three consecutive words
    __shared__ __align__(16) uint32_t s_ring[4 * kSlots * 64];
    constexpr uint32_t G = 16;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *wave_ring = s_ring + (threadIdx.x >> 6) * (kSlots * 64);
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    variant::ClockProbe probe;
    probe.begin();
    bool active = item < map.count;
    const uint32_t r = active ? map.candidate(item) : 0u;
    if (active) active = c.status[r] == kFrameOK;
    BitReader br;
    br.null(wave_ring + lane, spare);
    uint32_t bs = 0, bps = 0, ca = 0, limit = 0;
    if (active) {
        const uint32_t info = c.info[r];
        bs = info_bs(info); bps = info_bps(info); ca = info_ca(info);
        const uint64_t goff = c.goff[r];
        br.init(in, in_vecs, goff, wave_ring + lane, spare);
        const uint32_t a = (uint32_t)(goff & 63);
        limit = (a + c.len[r]) * 8;   // first bit after the frame
        br.seek((a + info_hl(info)) * 8);
    }
    bool alive = active;
    int rc = kFrameOK;
    RingFetch rf;
    rf.clear();

    for (uint32_t ch = 0; ch + 1 < C; ++ch) {
        uint32_t order = 0, psamp = 0, parts_left = 0, part_rem = 0, k = 0, pbits = 4, esc = 15;
        bool first_part = true, escaped = false, rice = false;
        auto kill = [&](int code) { rc = code; alive = false; rice = false; };
        // ---- subframe header, warm-up samples, predictor parameters, residual header
        if (alive) {
            c.subpos[(uint64_t)r * 8 + ch] = br.pos();
            SubHdr h;
            const int hrc = read_subframe_header(br, bps + (is_side(ca, ch) ? 1 : 0), bs, limit, h);
            if (hrc) kill(hrc);
            else if (h.kind == 0) br.skip(h.sbps);
            else if (h.kind == 1) {
                const uint64_t n = (uint64_t)bs * h.sbps;
                if (br.pos() + n > limit) kill(kFrameTruncated); else br.skip_long((uint32_t)n);
            } else {
                order = h.order;
                if (br.pos() + (uint64_t)order * h.sbps > limit) kill(kFrameTruncated);
                else {
                    if (order) br.skip_long(order * h.sbps);
                    if (h.lpc) {
                        const uint32_t pc = br.read(4);
                        const int32_t shift = br.read_signed(5);
                        if (pc == 15 || shift < 0) kill(kFrameBadSubframe); else br.skip_long(order * (pc + 1));
                    }
                    if (alive) {
                        const uint32_t method = br.read(2);
                        const uint32_t po = br.read(4);
                        pbits = method ? 5 : 4; esc = method ? 31 : 15;
                        psamp = bs >> po;
                        if (method > 1 || (bs & ((1u << po) - 1)) || psamp < order) kill(kFrameBadSubframe);
                        else { parts_left = 1u << po; rice = true; }
                    }
                }
            }
            if (alive && br.pos() > limit) kill(kFrameTruncated);
        }
        // Opens partitions until one holds a sample.  Escape-coded partitions are skipped over here, in one
        // move; their samples then pass as groups that read nothing.
        auto open_partition = [&]() {
            while (part_rem == 0 && parts_left && rice) {
                part_rem = psamp - (first_part ? order : 0);
                first_part = false;
                --parts_left;
                k = br.read(pbits);
                escaped = (k == esc);
                if (escaped) {
                    const uint32_t wd = br.read(5);
                    const uint64_t n = (uint64_t)part_rem * wd;
                    if (br.pos() + n > limit) kill(kFrameTruncated);
                    else if (n) br.skip_long((uint32_t)n);
                }
                if (rice && br.pos() > limit) kill(kFrameTruncated);
            }
            if (part_rem == 0) rice = false;   // no partition left
        };

        ring_round<true>(br, rf, rice);
        ring_round<true>(br, rf, rice);
        const uint32_t bs_max = wave_max_u32(rice ? bs : 0);
        for (uint32_t n0 = 0; n0 < bs_max; n0 += G) {
            ring_round<true>(br, rf, rice);
            if (rice && n0 < bs && br.pos() > limit) kill(kFrameTruncated);
            if (rice && n0 < bs && part_rem == 0 && n0 >= order) open_partition();
            const bool on = rice && n0 < bs;
            const bool aligned = !on || (n0 >= order && n0 + G <= bs && part_rem >= G);
            bool redo = true;
            bool run_generic = !__all(aligned);
            if (!run_generic) {
                const bool reads = on && !escaped;
                const uint32_t s_P = br.P;
                uint32_t longest = 0;
                bool starved = false;
                // Two codes per window only pay when a pair seldom exceeds 32 bits: one lane that has to be redone
                // costs the whole wave a walk through the group.  With Rice parameter k the unary parts of a pair
                // may total 30 - 2k.  The limit on k is tuned on synthetic audio (7: 26 % slower than 9; 9, 11, 14 equal).
                const bool pairs = __all(!reads || k <= kSkimPairK);
                uint32_t Q = 0u - br.P;   // the read position, negated (see peek_ring_neg)
                if (pairs) {
                    // two codes per window: the second is read from the window shifted by the first
#pragma unroll
                    for (int P = 0; P < (int)G / 2; ++P) {
                        const uint32_t w = br.peek_ring_neg(Q);
                        const uint32_t n1 = __clz(w) + 1 + k;
                        const uint32_t w2 = w << (n1 & 31);
                        const uint32_t tot = n1 + __clz(w2) + 1 + k;
                        longest = max(longest, tot);
                        Q -= tot;
                    }
                } else if constexpr (variant::kSkimRawCounts) {
                    // Two codes per 64-bit window on the hardware's unclamped leading-zero count (all ones for a zero
                    // window) and negated lengths, as the decoder's unrolled groups do: four dependent instructions
                    // fewer per pair than with clamped counts and positive lengths.  A lane whose streams are few
                    // (one long multichannel stream) is bound by exactly this chain.  A code fits its window when its unary
                    // count is at most 31 - k; the largest count of the group is tested once.
                    const uint32_t neg_k1 = 0u - (k + 1);
                    uint32_t qor = 0;   // largest unary count
#pragma unroll
                    for (int P = 0; P < (int)G / 2; ++P) {
                        if (2 * P == (int)kRoundSteps) {
                            br.P = 0u - Q;
                            starved = starved || br.starved();
                            ring_round<true>(br, rf, reads);
                        }
                        const uint2 w = br.peek64_ring_neg(Q);
                        const uint32_t q1 = lead_zeros_raw(w.x);
                        const uint32_t neg_n1 = neg_k1 - q1;
                        const uint32_t w2 = __builtin_amdgcn_alignbit(w.x, w.y, neg_n1);   // the window n1 bits on: the shift takes -n1 mod 32 itself
                        const uint32_t q2 = lead_zeros_raw(w2);
                        qor = max(qor, max(q1, q2));
                        Q += neg_n1 + (neg_k1 - q2);
                    }
                    longest = qor > 31u - k ? 33u : 0u;
                } else {
                    if constexpr (variant::kSkimWide) {
                        // two codes per 64-bit window: either code may be up to 32 bits long.  The OR of the lengths
                        // exceeds 32 when one of them does (and, harmlessly, when one is exactly 32 beside another).
#pragma unroll
                        for (int P = 0; P < (int)G / 2; ++P) {
                            if (2 * P == (int)kRoundSteps) {
                                br.P = 0u - Q;
                                starved = starved || br.starved();
                                ring_round<true>(br, rf, reads);
                            }
                            const uint2 w = br.peek64_ring_neg(Q);
                            const uint32_t n1 = __clz(w.x) + 1 + k;
                            const uint32_t w2 = __builtin_amdgcn_alignbit(w.x, w.y, 32u - n1);   // the window n1 bits on (n1 <= 32, else the lane is redone)
                            const uint32_t n2 = __clz(w2) + 1 + k;
                            longest |= n1 | n2;
                            Q -= n1 + n2;
                        }
                    } else {
#pragma unroll
                        for (int P = 0; P < (int)G; ++P) {
                            if (P == (int)kRoundSteps) {
                                br.P = 0u - Q;
                                starved = starved || br.starved();
                                ring_round<true>(br, rf, reads);
                            }
                            const uint32_t nb = __clz(br.peek_ring_neg(Q)) + 1 + k;
                            longest = max(longest, nb);
                            Q -= nb;
                        }
                    }
                }
                br.P = 0u - Q;
                starved = starved || br.starved();
                redo = reads && (starved || longest > 32);
                if (on && !redo) part_rem -= G;
                if (redo || !reads) br.P = s_P;
                if (redo && !pairs) {   // a refill round acted on the speculative read position: forget the ring
                    br.fill = (s_P >> 5) & ~(kUnitWords - 1);
                }
                run_generic = __any(redo);
            }
            if (run_generic) {
                for (uint32_t n = n0; n < n0 + G; ++n) {
                    if (n - n0 == kRoundSteps) ring_round<true>(br, rf, rice);
                    if (redo && rice && n >= order && n < bs) {
                        if (part_rem == 0) open_partition();
                        if (rice) {
                            if (!escaped) skip_rice(br, k, limit);
                            --part_rem;
                        }
                    }
                }
            }
        }
    }
    if (alive && C) c.subpos[(uint64_t)r * 8 + (C - 1)] = br.pos();
    if (active && rc) c.status[r] = rc;
    probe.end("skim", 256);
}

// ------------------------------------------------------------------ decode --

struct DecodeParams {
    const uint8_t *in;
    uint64_t in_vecs;      // 16-byte vectors readable from `in`
    uint8_t *out;
    const StreamDev *streams;
    const uint4 *spare;    // 32 readable bytes of device memory (see BitReader::init)
    CandArrays c;
    FrameMap map;          // the frames of the launch: all streams of one shape
    uint32_t C;            // channels of every frame of the launch
    uint32_t B;            // bytes per sample in the output
    uint32_t *class_flags; // [8]: set when some wave of the run belongs to that (taps, arithmetic) class
    uint32_t check_flag;   // leave at once when this instantiation's flag is not set
};

// Predictor classes: a wave runs in the smallest class that holds the largest
// order among its lanes.  U = history ring size and unroll factor, NT = taps.
__host__ __device__ constexpr int class_unroll(int nt) { return nt <= 16 ? 16 : 32; }

// 2, 3 and 4 dwords at a dword-aligned address
struct __attribute__((packed, aligned(4))) Dwords2 { uint32_t a, b; };
struct __attribute__((packed, aligned(4))) Dwords3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(4))) Dwords4 { uint32_t a, b, c, d; };

constexpr uint32_t kTile = 16;        // samples staged in LDS between write-outs
constexpr uint32_t kPairStride = 68;  // 16-bit stereo, 16-bit pair class: words per pair of sample columns (one word per lane holds two
                                      // consecutive samples): 64 lanes + 4, so that the write-out's 8-byte reads (one frame's two
                                      // channels; 32 lanes: 4 frames x 8 column pairs) cover all 64 banks once
constexpr uint32_t kRawStride = 66;   // words per sample column of the tile: 64 lanes + 2, so that the write-out's
                                      // 8-byte reads of 16-bit stereo (16 lanes: 4 frames x 4 columns) cover all banks once

// LDS bytes one wave needs: the refill ring, the sample tile (one 32-bit word per lane and sample: samples are
// staged as coded, whatever the output format).  Three blocks of four waves fit a CU's 160 KiB; a per-frame
// table of destinations and block sizes beside the tile made that two for mono streams (64 frames per wave),
// so what the write-out needs to know about a frame comes from the frame's first lane by a lane read instead.
__host__ __device__ inline uint32_t decode_wave_lds(uint32_t C)
{
    (void)C;
    return kRingBytesPerWave + kTile * kPairStride * 4;   // the larger of the two tile layouts
}

// One lane per (frame, channel); a wave holds floor(64 / C) frames.
// NT: predictor taps evaluated per sample (12 or 32).
// ARITH: 0 = 24-bit multiplies (every subframe of the wave is at most 24 bits wide),
//        1 = 32-bit multiplies, 2 = 64-bit sums,
//        3 = every subframe at most 17 bits wide and no 64-bit sums: the unrolled body multiplies pairs of
//            taps with v_dot2_i32_i16 on samples packed two to a register.  That is exact while the samples
//            fit 16 bits, which a 17-bit side channel almost always does; a lane that meets one that does
//            not is rolled back like a lane with an over-long Rice code, and the generic body (24-bit
//            multiplies) decodes until its history fits again.
// LAYOUT: 0 = any channel count and sample size whose rows are shorter than a line (C * B < 8: the general
//         write-out in four-slot items); 1 = C == 2 and B == 2 (STEREO16), the write-out specialised for it;
//         2 = C == 1 and B == 2 in the 16-bit pair class (MONO16); 3 = C * B >= 8 (WIDE: the general write-out in
//         sixteen-slot items, tile rows 65 words apart); 4 = planar output: every channel of a stream has a plane of
//         its own (StreamDev::plane_stride apart), a lane's samples go to its channel's plane, four per store.
// Lanes stage their samples as coded; stereo decorrelation and packing to the output's sample size happen
// in the write-out, once per output word.
//
// Samples are processed in groups of U.  A group in which every lane is in the
// middle of a Rice partition (or constant / finished) runs the unrolled fast
// body: no partition, warm-up or end-of-block tests, history renamed instead of
// moved.  Any other group (warm-up samples, coefficient parsing, partition
// boundary inside the group, escape-coded or verbatim data, the tail of a block)
// runs the generic rolled body with a shift-register history.
template <int NT, int ARITH, int LAYOUT>
__global__ __launch_bounds__(256, NT == 12 ? 3 : variant::kWaves32) void k_decode(DecodeParams p)   // LDS admits three blocks per CU; the 12-tap classes keep their registers within that
{
    extern __shared__ __align__(16) uint8_t smem[];
    constexpr int U = class_unroll(NT);
    // 16-bit stereo in the 16-bit pair class stages its samples as 16-bit halves: twice as many samples fit the
    // tile, and a write-out then covers whole 128-byte lines of every frame (64-byte halves of lines, written
    // seconds of wave time apart, cost the memory system a quarter of the kernel's time).  The unrolled body of
    // that class only keeps samples that fit 16 bits; the few that do not (generic body, 17-bit side channels)
    // are marked in `widemask`, one bit per tile column, and send the write-out down its slow path.
    constexpr bool STEREO16 = LAYOUT == 1, MONO16 = LAYOUT == 2, WIDE = LAYOUT == 3, PLANAR = LAYOUT == 4;
    // words per sample column of the raw tile.  WIDE: a lane of the write-out reads channel j of column g (+ 8) of
    // frame f, word 65 * column + C * f + j; with C = 8 the 32 lanes of a half wave (f = 0..3, g = 0..7) then
    // fall into 32 different banks (with 66 words they shared 8)
    constexpr uint32_t RS = WIDE ? 65u : kRawStride;
    static_assert(!MONO16 || ARITH == 3, "the mono layout stages packed 16-bit pairs, which only the 16-bit pair class makes");
    constexpr bool TILE16 = (STEREO16 && ARITH == 3) || MONO16;
    constexpr uint32_t kT = TILE16 ? 2 * kTile : kTile;
    // The instantiation launched first runs every wave and flags the classes its waves turn out to belong
    // to; the launches that follow are over in microseconds when no wave is theirs.
    constexpr bool DOT = ARITH == 3;
    if (p.check_flag && p.class_flags[(NT == 12 ? 0 : 1) + 2 * ARITH] == 0) return;
    const uint32_t C = STEREO16 ? 2u : MONO16 ? 1u : p.C, B = (STEREO16 || MONO16) ? 2u : p.B;
    const uint32_t CB = C * B;
    const uint32_t lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    const uint32_t F = 64 / C;
    const uint32_t wave_bytes = decode_wave_lds(C);
    uint32_t *wave_ring = reinterpret_cast<uint32_t *>(smem + wave_in_block * wave_bytes);
    uint8_t *tile = smem + wave_in_block * wave_bytes + kRingBytesPerWave;
    uint32_t *tile_raw = reinterpret_cast<uint32_t *>(tile);
    // TILE16: [column pair][lane] words; a word = (sample 2j + 1 in the low half, sample 2j in the high half) of its
    // lane, which is how the 16-bit pair class packs its history anyway
    uint16_t *tile_half = reinterpret_cast<uint16_t *>(tile);
    typedef short pair16 __attribute__((ext_vector_type(2)));

    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const uint32_t fl = lane / C, ch = lane - fl * C;
    const uint32_t item = wave * F + fl;
    bool active = fl < F && item < p.map.count;
    const uint32_t r = active ? p.map.candidate(item) : 0u;
    int status = kFrameOK;
    if (active) status = p.c.status[r];
    active = active && status == kFrameOK;

    uint32_t bs = 0, ca = 0, limit = 0;
    SubHdr h;
    h.kind = 0; h.order = 0; h.lpc = 0; h.wasted = 0; h.sbps = 0;
    BitReader br;
    br.null(wave_ring + lane, p.spare);
    uint64_t dst = 0;
    if (active) {
        const uint32_t info = p.c.info[r];
        const uint64_t goff = p.c.goff[r];
        const StreamDev &s = p.streams[p.c.stream[r]];
        bs = info_bs(info); ca = info_ca(info);
        limit = ((uint32_t)(goff & 63) + p.c.len[r]) * 8;
        br.init(p.in, p.in_vecs, goff, wave_ring + lane, p.spare);
        br.seek(p.c.subpos[(uint64_t)r * 8 + ch]);
        const uint64_t first = info_var(info) ? p.c.num[r] : p.c.num[r] * (uint64_t)s.max_bs;
        // from p.out (the range check has seen to first >= sample_base); planar: this lane's own plane
        dst = PLANAR ? s.out_off + ch * s.plane_stride + (first - s.sample_base) * B : s.out_off + (first - s.sample_base) * CB;
        const int rc = read_subframe_header(br, info_bps(info) + (is_side(ca, ch) ? 1 : 0), bs, limit, h);
        if (rc) { status = rc; h.kind = 0; h.order = 0; }
    }
    // predictor precision is stored after the warm-up samples: look ahead for the accumulator width
    bool need_wide = false;
    if (active && status == kFrameOK && h.kind == 2 && h.lpc) {
        BitReader t = br;
        t.skip_long(h.order * h.sbps);
        const uint32_t prec = t.read(4) + 1;
        need_wide = (h.sbps + prec + (31 - __clz(h.order))) > 32;
    }
    const uint32_t maxord = wave_max_u32(active && status == kFrameOK ? h.order : 0);
    const bool any_wide = __any(need_wide);
    const bool any_big = __any(active && status == kFrameOK && h.sbps > 24);
    const bool any_over17 = __any(active && status == kFrameOK && h.sbps > 17);
    // a wave is decoded by exactly one instantiation: 12 or 32 taps x {16-bit pairs, 24-bit, 32-bit, 64-bit} arithmetic
    const int cls = maxord <= 12 ? 12 : 32;
    const int arith = any_wide ? 2 : any_big ? 1 : any_over17 ? 0 : 3;
    if (cls != NT || arith != ARITH) {
        if (lane == 0) p.class_flags[(cls == 12 ? 0 : 1) + 2 * arith] = 1;
        return;
    }
    if (!p.check_flag && lane == 0) p.class_flags[(NT == 12 ? 0 : 1) + 2 * ARITH] = 1;   // for the timing report only
    variant::ClockProbe probe;
    probe.begin();

    const uint32_t bs_max = wave_max_u32(active ? bs : 0);
    // a frame's destination, block size (bits 0-19; 0 when the frame is not decoded) and channel assignment
    // (bits 20-23), as held by the lane of its first channel
    const uint32_t my_meta = (active ? bs : 0) | (ca << 20);
    auto frame_dst = [&](uint32_t f) { return (uint64_t)__shfl((unsigned long long)dst, (int)(f * C)); };
    auto frame_meta = [&](uint32_t f) { return (uint32_t)__shfl((int)my_meta, (int)(f * C)); };
    // whole-dword write-out needs every frame's destination 4-byte aligned (a tile row of kTile samples is a
    // whole number of dwords for every sample size and channel count)
    const bool dwords = !__any(active && ((reinterpret_cast<uint64_t>(p.out) + dst) & 3));
    __builtin_amdgcn_wave_barrier();
    // 16-bit stereo write-out: this lane packs sample columns 4*(lane & 3) .. +3 of frames 16*it + (lane >> 2),
    // it = 0, 1.  Their channel assignments (4 bits each), where they go (offset of the lane's first
    // column from p.out) and how long they are:
    // With the 16-bit tile a row is 32 samples = 128 bytes: this lane packs sample columns 4*(lane & 7) .. +3 of
    // frames 8*it + (lane >> 3), it = 0 .. 3.
    constexpr int NIT = TILE16 ? 4 : 2;
    uint32_t modes = 0;
    uint64_t fdst[NIT] = {};
    uint32_t fbs[NIT] = {};
    // The decorrelation of a pair (a, b), whatever the frame's channel assignment, as
    //   left = a + lmul * ((b + half) >> half),  right = (a & rmask) + rmul * (b >> half)
    // with per-frame constants: half = 1 for mid/side; lmul = 1 when b is a side channel that the left sample
    // needs (side/right, mid/side); rmask = all ones and rmul = -1 when the right sample is a minus b's part
    // (left/side, mid/side), else 0 and 1.
    int32_t dc_half[NIT] = {}, dc_lmul[NIT] = {}, dc_rmul[NIT] = {};
    uint32_t dc_rmask[NIT] = {};
    uint32_t dc_half2[NIT] = {}, dc_lmul2[NIT] = {}, dc_rmul2[NIT] = {};   // the same in both 16-bit halves of a word
    bool row16[NIT] = {};   // the frame's rows of four words start on 16-byte boundaries
    // Other layouts: a frame's 16 samples x C channels are 16C sample slots in output order.  WIDE: a lane's
    // work item is sixteen slots of its own frame, two runs of eight consecutive slots 8C slots apart = 16 B
    // output bytes: a frame has C items, the wave F * C <= 64 of them, lane = f * C + g.  (Four stores of B
    // dwords per lane, four slots each, cost the 24-bit 8-channel shape a sixth of its time: the wave's store
    // instructions queue; sixteen consecutive slots per lane put the LDS reads of a half wave into a quarter
    // of the banks.)
    // Divisions by C and 4C are multiplications by rounded-up 16-bit reciprocals (exact while dividend x
    // divisor < 2^16; dividends here are below 256, divisors at most 32).
    const uint32_t inv_c = 65536u / C + 1, inv_4c = 65536u / (4 * C) + 1;
    // What never changes from one write-out to the next is worked out once: the LDS addresses of the item's
    // slots (the block's LDS is below 64 KiB), where its bytes go (offset from p.out of the frame's first
    // tile), and in one word the frame's block size (bits 0-19; 0 for a lane beyond the wave's frames), the
    // sample column of its last slot (20-23) and the channel assignment (24-27).
    typedef __attribute__((address_space(3))) const uint32_t lds_word;
    // Rows shorter than a 128-byte line (C * B < 8: mono, 8-bit and 24-bit stereo ...) keep the finer mapping
    // instead: an item is four slots (B dwords, one store), 4C items per frame, item = 64 * it + lane, it = 0 .. 3,
    // so that one store instruction covers whole rows of consecutive frames rather than a sliver of every row.
    constexpr bool wide_items = WIDE;
    uint32_t q_src[16] = {};      // wide: the item's sixteen slots; else [it * 4 + j]
    uint64_t q_dst[4] = {};       // wide: [0] only
    uint32_t q_meta[4] = {};
    if (MONO16) {
        // a row is 32 samples = 64 bytes: this lane copies sample columns 8*(lane & 3) .. +7 of frames 16*it + (lane >> 2)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint32_t f = it * 16 + (lane >> 2);
            const uint32_t meta = frame_meta(f);
            fdst[it] = frame_dst(f) + (lane & 3) * 16;
            fbs[it] = meta & 0xFFFFFu;
            row16[it] = dwords && ((reinterpret_cast<uint64_t>(p.out) + fdst[it]) & 15) == 0;
        }
    } else if (STEREO16) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint32_t f = TILE16 ? it * 8 + (lane >> 3) : it * 16 + (lane >> 2);
            const uint32_t meta = frame_meta(f);
            modes |= ((meta >> 20) & 15u) << (4 * it);
            fdst[it] = frame_dst(f) + (TILE16 ? (lane & 7) : (lane & 3)) * 16;
            fbs[it] = meta & 0xFFFFFu;
            const uint32_t m = (meta >> 20) & 15u;
            dc_half[it] = m == 10 ? 1 : 0;
            dc_lmul[it] = (m == 9 || m == 10) ? 1 : 0;
            dc_rmul[it] = (m == 8 || m == 10) ? -1 : 1;
            dc_rmask[it] = (m == 8 || m == 10) ? 0xFFFFFFFFu : 0u;
            dc_half2[it] = (uint32_t)dc_half[it] * 0x10001u;
            dc_lmul2[it] = (uint32_t)dc_lmul[it] * 0x10001u;
            dc_rmul2[it] = ((uint32_t)dc_rmul[it] & 0xFFFFu) * 0x10001u;
            row16[it] = dwords && ((reinterpret_cast<uint64_t>(p.out) + fdst[it]) & 15) == 0;
        }
    } else if (PLANAR) {
        // an item is four consecutive samples of one lane (= one channel of one frame): item = 64 * it + lane covers
        // source lane 16 * it + (lane >> 2), samples 4 * (lane & 3) .. + 3, so four neighbours write one lane's 16 samples
        const uint32_t tile_lds = (uint32_t)(uintptr_t)(lds_word *)tile_raw;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t src = it * 16 + (lane >> 2), quad = lane & 3, f = (src * inv_c) >> 16;
            const bool ok = f < F;
#pragma unroll
            for (int j = 0; j < 4; ++j) q_src[it * 4 + j] = tile_lds + ((4 * quad + j) * RS + src) * 4;
            const uint64_t dst_l = (uint64_t)__shfl((unsigned long long)dst, (int)src);
            const uint32_t meta = frame_meta(ok ? f : 0u);
            q_dst[it] = ok ? dst_l + quad * 4 * B : 0;
            // block size | last sample column of the item (20-23) | channel assignment (24-27) | second channel of a pair (28)
            q_meta[it] = ok ? (meta & 0xFFFFFu) | ((4 * quad + 3) << 20) | (((meta >> 20) & 15u) << 24) | ((src & 1u) << 28) : 0u;
        }
    } else {
        const uint32_t tile_lds = (uint32_t)(uintptr_t)(lds_word *)tile_raw;
        if (wide_items) {
            const uint32_t f = fl, g = ch;          // this lane's item belongs to the frame it decodes a channel of
            const bool ok = f < F;
            uint32_t col = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // two runs of eight slots, 8 * C slots apart (with eight channels: sample columns g and g + 8)
                const uint32_t slot = 8 * (g + (j >= 8 ? C : 0u)) + (uint32_t)(j & 7);
                col = (slot * inv_c) >> 16;
                q_src[j] = tile_lds + (ok ? (col * RS + f * C + (slot - col * C)) * 4 : 0u);
            }
            const uint32_t fc = ok ? f : 0u;
            const uint64_t dst_f = frame_dst(fc);
            const uint32_t meta = frame_meta(fc);
            q_dst[0] = ok ? dst_f + g * 8 * B : 0;
            q_meta[0] = ok ? (meta & 0xFFFFFu) | (col << 20) | (((meta >> 20) & 15u) << 24) : 0u;
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t item = it * 64 + lane, f = (item * inv_4c) >> 16, g = item - f * 4 * C;
                const bool ok = f < F;
                uint32_t col = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t slot = 4 * g + j;
                    col = (slot * inv_c) >> 16;
                    q_src[it * 4 + j] = tile_lds + (ok ? (col * RS + f * C + (slot - col * C)) * 4 : 0u);
                }
                const uint32_t fc = ok ? f : 0u;
                const uint64_t dst_f = frame_dst(fc);
                const uint32_t meta = frame_meta(fc);
                q_dst[it] = ok ? dst_f + g * 4 * B : 0;
                q_meta[it] = ok ? (meta & 0xFFFFFu) | (col << 20) | (((meta >> 20) & 15u) << 24) : 0u;
            }
        }
    }

    // ---- per-lane decoder state --------------------------------------------------------
    int32_t coef[NT];
    int32_t ring[U];    // ring[i] = most recent sample n with n % U == i
    // 16-bit pair class: pk[i] = (sample n, sample n - 1) for the most recent n with n % U == i, as 16-bit halves;
    // cp[i] = (coef[2i], coef[2i + 1]).  The packs are kept from one unrolled group to the next and rebuilt from
    // `ring` after a generic group (packs_ok is the same in every lane).  hist16: this lane's last NT samples fit 16 bits.
    constexpr int NP = NT / 2;
    uint32_t pk[DOT ? U : 1], cp[DOT ? NP : 1];
    bool packs_ok = false, hist16 = true;
    if (DOT) {
#pragma unroll
        for (int j = 0; j < NP; ++j) cp[j] = 0;
#pragma unroll
        for (int j = 0; j < U; ++j) pk[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) coef[j] = 0;
#pragma unroll
    for (int j = 0; j < U; ++j) ring[j] = 0;
    uint32_t shift = 0;
    bool started = false;
    const uint32_t kind = active ? h.kind : 0;
    const uint32_t order = h.order;
    int32_t constval = 0;
    if (active && kind == 0) constval = br.read_signed(h.sbps);
    uint32_t pbits = 4, esc = 15, psamp = 0, parts_left = 0, part_rem = 0, k = 0, escw = 0;
    bool escaped = false, first_part = true;
    // A constant subframe goes through the unrolled groups as a predicted one whose residual is always zero: order-1
    // predictor with coefficient 1 and shift 0 over a history of the constant, Rice parameter 0, and a ring of
    // all-one words, so that every code the lane reads there is the one-bit code of zero.  That spares the
    // unrolled body a select per sample.  A constant the packed history cannot hold (17-bit side channels) keeps
    // its lane, and with it the wave, in the generic groups.
    const bool const_fast = kind != 0 || !DOT || (((uint32_t)constval + 0x8000u) >> 16) == 0;
    if (active && kind == 0) {
        coef[0] = 1;
        if (DOT) cp[0] = pack_lo_hi(1, 0);
#pragma unroll
        for (int j = 0; j < U; ++j) ring[j] = constval;
        for (uint32_t sl = 0; sl < kRingSlots; ++sl) br.ring[sl * 64] = 0xFFFFFFFFu;
    }
    const bool reads_stream = active && kind != 0;

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
            for (int j = 0; j < NT; ++j)
                if ((uint32_t)j < order) coef[j] = br.read_signed(prec);
        } else {
            if (order == 1) { coef[0] = 1; }
            if (order == 2) { coef[0] = 2; coef[1] = -1; }
            if (order == 3) { coef[0] = 3; coef[1] = -3; coef[2] = 1; }
            if (order == 4) { coef[0] = 4; coef[1] = -6; coef[2] = 4; coef[3] = -1; }
        }
        if (DOT) {
#pragma unroll
            for (int j = 0; j < NP; ++j) cp[j] = pack_lo_hi(coef[2 * j], coef[2 * j + 1]);   // coefficients have at most 15 bits
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
    // Staging of one sample into the LDS tile; `col` = sample index inside the tile.  Samples that do not
    // fit the subframe's width mark the frame invalid: v + 2^(sbits-1) has no bit above sbits-1 exactly
    // when v is in range, and the OR of those sums is tested once at the end.
    // The unrolled groups collect their sums in `govf`, which joins `ovf` only when the lane keeps the group's
    // results.  In the 16-bit pair class a 17-bit lane tests its samples against 16 bits there instead (half_fast):
    // not an error when it fails, but the group's products were wrong and the lane is redone.
    const uint32_t sbits = h.sbps ? h.sbps : 32u;
    const uint32_t half = 1u << (sbits - 1);
    // 17-bit lanes of the pair class: a subframe 17 bits wide after its wasted bits (the packed history holds 16), or
    // one whose samples are 17 bits wide with their wasted bits (a side channel: the packed tile holds 16)
    const bool pair17 = DOT && (h.sbps == 17 || h.sbps + h.wasted == 17);
    const uint32_t fast_bits = pair17 ? sbits - 1 : sbits;   // what a sample of an unrolled group has to fit
    const uint32_t half_fast = pair17 ? ((1u << fast_bits) >> 1) : half;
    uint32_t ovf = 0, govf = 0;
    int32_t gmax = 0, gmin = 0;   // largest and smallest sample of an unrolled group (two three-operand instructions per pair of samples)
    // TILE16: columns of the current tile whose staged value (sample << wasted bits) does not fit 16 bits.  A
    // constant subframe with such a value marks every column for good.
    const uint32_t widemask_rest = (TILE16 && active && h.kind == 0 && ((((uint32_t)constval << h.wasted) + 0x8000u) >> 16)) ? 0xFFFFFFFFu : 0u;
    uint32_t widemask = widemask_rest;
    auto emit = [&](int32_t v, bool live, uint32_t col, auto always) {
        constexpr bool ALWAYS = decltype(always)::value;   // fast groups store from every lane: a dead lane's column is never written out
        if (ALWAYS) { gmax = max(gmax, v); gmin = min(gmin, v); } else ovf |= (uint32_t)v + half;
        const uint32_t x = (uint32_t)v << h.wasted;
        if (TILE16) {
            // generic groups store sample by sample; the unrolled groups store their packed pairs themselves
            if (!ALWAYS && live) {
                if ((x + 0x8000u) >> 16) widemask |= 1u << col;
                tile_half[((col >> 1) * kPairStride + lane) * 2 + (~col & 1u)] = (uint16_t)x;
            }
        } else {
            if (ALWAYS || live) tile_raw[col * RS + lane] = x;
        }
    };
    // Sum of coef[j] * x[j] over the NT taps in the wave's arithmetic class, shifted.  The taps are
    // summed in three independent chains and the most recent sample (tap 0) is added last: a dependent
    // vector instruction waits about ten cycles for its operand, and tap 0 is the only one that depends
    // on the previous sample's result.
    auto predict = [&](auto tap) -> int32_t {
        if (ARITH == 2) {
            int64_t s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                const int64_t m = (int64_t)coef[j] * (int64_t)tap(j);
                if (j % 3 == 1) s0 += m; else if (j % 3 == 2) s1 += m; else s2 += m;
            }
            const int64_t sum = (s0 + s1) + s2 + (int64_t)coef[0] * (int64_t)tap(0);
            // A lane whose own subframe does not need 64 bits (sbps + precision + floor(log2(order)) <= 32, so
            // its sum fits 32 bits) gets the same low word from the 64-bit shift as from a 32-bit one.
            return (int32_t)(sum >> shift);
        } else if (ARITH == 1) {
            uint32_t s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                const uint32_t m = (uint32_t)coef[j] * (uint32_t)tap(j);
                if (j % 3 == 1) s0 += m; else if (j % 3 == 2) s1 += m; else s2 += m;
            }
            const uint32_t sum = (s0 + s1) + s2 + (uint32_t)coef[0] * (uint32_t)tap(0);
            return (int32_t)sum >> shift;
        } else {
            // 16-bit pair class (this is its generic body): the coefficients live packed in cp[] only, which spares
            // the unrolled body NT registers; here they are taken apart again, tap by tap
            auto cf = [&](int j) -> int32_t {
                if constexpr (DOT) return (j & 1) ? (int32_t)cp[j >> 1] >> 16 : __builtin_amdgcn_sbfe((int32_t)cp[j >> 1], 0, 16);
                else return coef[j];
            };
            int32_t s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                if (j % 3 == 1) s0 = mad_i24(cf(j), tap(j), s0);
                else if (j % 3 == 2) s1 = mad_i24(cf(j), tap(j), s1);
                else s2 = mad_i24(cf(j), tap(j), s2);
            }
            const int32_t sum = mad_i24(cf(0), tap(0), (s0 + s1) + s2);
            return sum >> shift;
        }
    };
    // Undoes the inter-channel decorrelation of one sample pair (channel assignment m: 8 left/side,
    // 9 side/right, 10 mid/side; mid/side: left = mid + ceil(side / 2), right = mid - floor(side / 2)).
    auto decorrelate = [](uint32_t m, uint32_t a, uint32_t b, uint32_t &L, uint32_t &R) {
        L = m == 10 ? a + (uint32_t)(((int32_t)b + 1) >> 1) : m == 9 ? a + b : a;
        R = m == 10 ? a - (uint32_t)((int32_t)b >> 1) : m == 8 ? a - b : b;
    };
    // Write-out of the staged samples [t0, t0 + kT) of every frame of this wave.
    auto flush = [&](uint32_t t0) {
        if (variant::kNoFlush && t0 != 0xFFFFFFF0u) return;
        __builtin_amdgcn_wave_barrier();
        if (MONO16) {
            // a word of the tile is two consecutive samples of one frame, the later one in the low half; the
            // output wants the earlier one there
            const uint32_t cp4 = (lane & 3) * 4, c8 = (lane & 3) * 8;
            bool whole = true;
#pragma unroll
            for (int it = 0; it < NIT; ++it) whole = whole && row16[it] && t0 + c8 + 8 <= fbs[it];
            if (__all(whole)) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    uint32_t word[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t w = tile_raw[(cp4 + i) * kPairStride + it * 16 + (lane >> 2)];
                        word[i] = __builtin_amdgcn_alignbit(w, w, 16);
                    }
                    if (!variant::kNoStore16 || (word[0] == 0x12345678u && word[1] == 0x9abcdef0u && word[2] == word[3]))   // the words never have these values
                    *reinterpret_cast<uint4 *>(p.out + fdst[it] + (uint64_t)t0 * 2) = uint4{word[0], word[1], word[2], word[3]};
                }
            } else {
#pragma unroll 1
                for (int it = 0; it < NIT; ++it) {
                    uint64_t dst_it = 0;
                    uint32_t bs_it = 0;
#pragma unroll
                    for (int k = 0; k < NIT; ++k)
                        if (k == it) { dst_it = fdst[k]; bs_it = fbs[k]; }
                    uint8_t *at = p.out + dst_it + (uint64_t)t0 * 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t w = tile_raw[(cp4 + i) * kPairStride + (uint32_t)it * 16 + (lane >> 2)];
                        const uint32_t n = t0 + c8 + 2 * i;
                        if (n < bs_it) { at[4 * i] = (uint8_t)(w >> 16); at[4 * i + 1] = (uint8_t)(w >> 24); }
                        if (n + 1 < bs_it) { at[4 * i + 2] = (uint8_t)w; at[4 * i + 3] = (uint8_t)(w >> 8); }
                    }
                }
            }
            widemask = widemask_rest;
        } else if (STEREO16) {
            // four output words (left | right << 16) per lane and row: the lanes of a row write 64 (TILE16: 128)
            // contiguous bytes of one frame
            const uint32_t c4 = (TILE16 ? (lane & 7) : (lane & 3)) * 4;
            // one row's four sample pairs (a, b) = (channel 0, channel 1)
            auto load_row = [&](int it, uint2 (&ab)[4]) {
                if (TILE16) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {   // a column pair: both channels' packed words of the frame
                        const uint2 w = *reinterpret_cast<const uint2 *>(tile_raw + (c4 / 2 + j) * kPairStride + 2 * (it * 8 + (lane >> 3)));
                        ab[2 * j] = uint2{(uint32_t)((int32_t)w.x >> 16), (uint32_t)((int32_t)w.y >> 16)};
                        ab[2 * j + 1] = uint2{(uint32_t)__builtin_amdgcn_sbfe((int32_t)w.x, 0, 16), (uint32_t)__builtin_amdgcn_sbfe((int32_t)w.y, 0, 16)};
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        ab[i] = *reinterpret_cast<const uint2 *>(tile_raw + (c4 + i) * kRawStride + 2 * (it * 16 + (lane >> 2)));
                }
            };
            // every lane's rows inside their blocks and aligned (all tiles but a block's ragged last one, in
            // waves whose frames are all being decoded), no staged value wider than its half word: no
            // branches, seven instructions per output word
            bool whole = true;
#pragma unroll
            for (int it = 0; it < NIT; ++it) whole = whole && row16[it] && t0 + c4 + 4 <= fbs[it];
            if (TILE16) whole = whole && widemask == 0;
            if (__all(whole)) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    uint32_t word[4];
                    if (TILE16) {
                        // two samples at a time in 16-bit halves (everything staged fits them, and only the low 16
                        // bits of a result are stored): four instructions per output word
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const uint2 w = *reinterpret_cast<const uint2 *>(tile_raw + (c4 / 2 + j) * kPairStride + 2 * (it * 8 + (lane >> 3)));
                            const pair16 a = __builtin_bit_cast(pair16, w.x), b = __builtin_bit_cast(pair16, w.y);
                            const pair16 hf = __builtin_bit_cast(pair16, dc_half2[it]);
                            const pair16 hb = b >> hf, hc = hb + (b & hf);
                            const pair16 L = hc * __builtin_bit_cast(pair16, dc_lmul2[it]) + a;
                            const pair16 R = hb * __builtin_bit_cast(pair16, dc_rmul2[it]) + (a & __builtin_bit_cast(pair16, dc_rmask[it]));
                            const uint32_t l = __builtin_bit_cast(uint32_t, L), r = __builtin_bit_cast(uint32_t, R);
                            word[2 * j] = __builtin_amdgcn_perm(r, l, 0x07060302u);       // the earlier sample sits in the high halves
                            word[2 * j + 1] = __builtin_amdgcn_perm(r, l, 0x05040100u);
                        }
                    } else {
                        uint2 ab[4];
                        load_row(it, ab);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int32_t a = (int32_t)ab[i].x, b = (int32_t)ab[i].y;
                            const int32_t hb = b >> dc_half[it], hc = (b + dc_half[it]) >> dc_half[it];
                            const int32_t L = mad_i24(hc, dc_lmul[it], a);
                            const int32_t R = mad_i24(hb, dc_rmul[it], (int32_t)((uint32_t)a & dc_rmask[it]));
                            word[i] = pack_lo_hi(L, R);
                        }
                    }
                    if (!variant::kNoStore16 || (word[0] == 0x12345678u && word[1] == 0x9abcdef0u && word[2] == word[3]))
                    *reinterpret_cast<uint4 *>(p.out + fdst[it] + (uint64_t)t0 * 4) = uint4{word[0], word[1], word[2], word[3]};
                }
            } else {
#pragma unroll 1
                for (int it = 0; it < NIT; ++it) {
                    const uint32_t m = (modes >> (4 * it)) & 15u;
                    uint32_t wm0 = 0, wm1 = 0;
                    if (TILE16) {   // the frame's two lanes know which of their staged values lost their top bit
                        const uint32_t f = (uint32_t)it * 8 + (lane >> 3);
                        wm0 = (uint32_t)__shfl((int)widemask, (int)(2 * f));
                        wm1 = (uint32_t)__shfl((int)widemask, (int)(2 * f + 1));
                    }
                    uint2 ab[4];
                    load_row(it, ab);
                    uint32_t word[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint32_t a = ab[i].x, b = ab[i].y;
                        // a value that does not fit 16 bits has 17: its top bit is the inverse of bit 15
                        if ((wm0 >> (c4 + i)) & 1) a ^= 0xFFFF0000u;
                        if ((wm1 >> (c4 + i)) & 1) b ^= 0xFFFF0000u;
                        uint32_t L, R;
                        decorrelate(m, a, b, L, R);
                        word[i] = (L & 0xFFFFu) | (R << 16);
                    }
                    uint64_t dst_it = 0;
                    uint32_t bs_it = 0;
#pragma unroll
                    for (int k = 0; k < NIT; ++k)
                        if (k == it) { dst_it = fdst[k]; bs_it = fbs[k]; }
                    const uint32_t n = t0 + c4;
                    uint8_t *at = p.out + dst_it + (uint64_t)t0 * 4;
                    if (dwords && n + 4 <= bs_it && (reinterpret_cast<uint64_t>(at) & 15) == 0) {
                        *reinterpret_cast<uint4 *>(at) = uint4{word[0], word[1], word[2], word[3]};
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (n + i < bs_it) {
                                uint8_t *q = at + 4 * i;
                                const uint32_t x = word[i];
                                if (dwords) *reinterpret_cast<uint32_t *>(q) = x;
                                else { q[0] = (uint8_t)x; q[1] = (uint8_t)(x >> 8); q[2] = (uint8_t)(x >> 16); q[3] = (uint8_t)(x >> 24); }
                            }
                    }
                }
            }
            widemask = widemask_rest;
        } else if (PLANAR) {
            auto pack = [&](const uint32_t (&v)[4], uint32_t &d0, uint32_t &d1, uint32_t &d2) {
                d0 = (v[0] & 0xFFFFFFu) | (v[1] << 24);
                d1 = ((v[1] >> 8) & 0xFFFFu) | (v[2] << 16);
                d2 = ((v[2] >> 16) & 0xFFu) | (v[3] << 8);
            };
            // the pair's other channel sits one word beside this one in the tile
            auto channel_value = [&](uint32_t mine, uint32_t other, uint32_t m, bool second) {
                uint32_t L, R;
                decorrelate(m, second ? other : mine, second ? mine : other, L, R);
                return second ? R : L;
            };
            bool ragged = false;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t fbs_f = q_meta[it] & 0xFFFFFu;
                if (t0 >= fbs_f) continue;
                if (!dwords || ((q_meta[it] >> 20) & 15u) >= fbs_f - t0) { ragged = true; continue; }
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *(lds_word *)(uintptr_t)q_src[it * 4 + j];
                if (C == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = channel_value(v[j], *(lds_word *)(uintptr_t)(q_src[it * 4 + j] ^ 4u), (q_meta[it] >> 24) & 15u, (q_meta[it] >> 28) != 0);
                }
                uint8_t *q = p.out + q_dst[it] + (uint64_t)t0 * B;
                if (B == 1) *reinterpret_cast<uint32_t *>(q) = (v[0] & 0xFFu) | ((v[1] & 0xFFu) << 8) | ((v[2] & 0xFFu) << 16) | (v[3] << 24);
                else if (B == 2) *reinterpret_cast<Dwords2 *>(q) = Dwords2{(v[0] & 0xFFFFu) | (v[1] << 16), (v[2] & 0xFFFFu) | (v[3] << 16)};
                else if (B == 3) { uint32_t d0, d1, d2; pack(v, d0, d1, d2); *reinterpret_cast<Dwords3 *>(q) = Dwords3{d0, d1, d2}; }
                else *reinterpret_cast<Dwords4 *>(q) = Dwords4{v[0], v[1], v[2], v[3]};
            }
            if (__any(ragged))
#pragma unroll 1
            for (uint32_t it = 0; it < 4; ++it) {
                const uint32_t src = it * 16 + (lane >> 2), quad = lane & 3, f = (src * inv_c) >> 16;
                const uint64_t dst_l = (uint64_t)__shfl((unsigned long long)dst, (int)src);   // lane reads: before the lanes part ways
                const uint32_t meta = frame_meta(f < F ? f : 0u);
                if (f >= F) continue;
                const uint32_t fbs_f = meta & 0xFFFFFu;
                if (t0 >= fbs_f) continue;
                const uint32_t ncols = min(kTile, fbs_f - t0);
                if (dwords && 4 * quad + 3 < ncols) continue;   // written above
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t col = 4 * quad + j;
                    if (col >= ncols) continue;
                    uint32_t x = tile_raw[col * RS + src];
                    if (C == 2) x = channel_value(x, tile_raw[col * RS + (src ^ 1u)], (meta >> 20) & 15u, (src & 1u) != 0);
                    uint8_t *at = p.out + dst_l + (uint64_t)(t0 + col) * B;
                    for (uint32_t y = 0; y < B; ++y) at[y] = (uint8_t)(x >> (8 * y));
                }
            }
        } else {
            // four 24-bit samples as three dwords
            auto pack = [&](const uint32_t (&v)[4], uint32_t &d0, uint32_t &d1, uint32_t &d2) {
                d0 = (v[0] & 0xFFFFFFu) | (v[1] << 24);
                d1 = ((v[1] >> 8) & 0xFFFFu) | (v[2] << 16);
                d2 = ((v[2] >> 16) & 0xFFu) | (v[3] << 8);
            };
            // An item whose slots all exist in the frame and whose destination is dword aligned (all but a
            // block's ragged last tile): everything about it but t0 was worked out before the sample loop.
            bool ragged = false;
            if (wide_items) {
                const uint32_t fbs_f = q_meta[0] & 0xFFFFFu;
                if (t0 < fbs_f) {
                    if (!dwords || ((q_meta[0] >> 20) & 15u) >= fbs_f - t0) ragged = true;
                    else {
                        uint32_t v[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) v[j] = *(lds_word *)(uintptr_t)q_src[j];
                        if (C == 2) {
                            const uint32_t m = q_meta[0] >> 24;
#pragma unroll
                            for (int j = 0; j < 16; j += 2) {
                                uint32_t L, R;
                                decorrelate(m, v[j], v[j + 1], L, R); v[j] = L; v[j + 1] = R;
                            }
                        }
                        // multi-dword global stores need dword alignment only; a wave's stores count against the
                        // same counter as its refill loads, so fewer and wider ones wait less
                        uint8_t *q = p.out + q_dst[0] + (uint64_t)t0 * CB;
                        if (!(variant::kNoStoreGeneral && t0 != 0xFFFFFFF0u)) {
                            // the item's two runs of eight slots lie 8 * C slots apart
                            uint8_t *const q2 = q + 8 * CB;
                            if (B == 4) {
                                reinterpret_cast<Dwords4 *>(q)[0] = Dwords4{v[0], v[1], v[2], v[3]};
                                reinterpret_cast<Dwords4 *>(q)[1] = Dwords4{v[4], v[5], v[6], v[7]};
                                reinterpret_cast<Dwords4 *>(q2)[0] = Dwords4{v[8], v[9], v[10], v[11]};
                                reinterpret_cast<Dwords4 *>(q2)[1] = Dwords4{v[12], v[13], v[14], v[15]};
                            } else if (B == 3) {
                                uint32_t d[12];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const uint32_t w[4] = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
                                    pack(w, d[3 * k], d[3 * k + 1], d[3 * k + 2]);
                                }
                                *reinterpret_cast<Dwords4 *>(q) = Dwords4{d[0], d[1], d[2], d[3]};
                                *reinterpret_cast<Dwords2 *>(q + 16) = Dwords2{d[4], d[5]};
                                *reinterpret_cast<Dwords4 *>(q2) = Dwords4{d[6], d[7], d[8], d[9]};
                                *reinterpret_cast<Dwords2 *>(q2 + 16) = Dwords2{d[10], d[11]};
                            } else if (B == 2) {
                                uint32_t d[8];
#pragma unroll
                                for (int k = 0; k < 8; ++k) d[k] = (v[2 * k] & 0xFFFFu) | (v[2 * k + 1] << 16);
                                *reinterpret_cast<Dwords4 *>(q) = Dwords4{d[0], d[1], d[2], d[3]};
                                *reinterpret_cast<Dwords4 *>(q2) = Dwords4{d[4], d[5], d[6], d[7]};
                            } else {
                                uint32_t d[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    d[k] = (v[4 * k] & 0xFFu) | ((v[4 * k + 1] & 0xFFu) << 8) | ((v[4 * k + 2] & 0xFFu) << 16) | (v[4 * k + 3] << 24);
                                *reinterpret_cast<Dwords2 *>(q) = Dwords2{d[0], d[1]};
                                *reinterpret_cast<Dwords2 *>(q2) = Dwords2{d[2], d[3]};
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const uint32_t fbs_f = q_meta[it] & 0xFFFFFu;
                    if (t0 >= fbs_f) continue;
                    if (!dwords || ((q_meta[it] >> 20) & 15u) >= fbs_f - t0) { ragged = true; continue; }
                    uint32_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = *(lds_word *)(uintptr_t)q_src[it * 4 + j];
                    if (C == 2) {
                        const uint32_t m = q_meta[it] >> 24;
                        uint32_t L, R;
                        decorrelate(m, v[0], v[1], L, R); v[0] = L; v[1] = R;
                        decorrelate(m, v[2], v[3], L, R); v[2] = L; v[3] = R;
                    }
                    uint8_t *q = p.out + q_dst[it] + (uint64_t)t0 * CB;
                    if (variant::kNoStoreGeneral && t0 != 0xFFFFFFF0u) continue;
                    if (B == 1) *reinterpret_cast<uint32_t *>(q) = (v[0] & 0xFFu) | ((v[1] & 0xFFu) << 8) | ((v[2] & 0xFFu) << 16) | (v[3] << 24);
                    else if (B == 2) *reinterpret_cast<Dwords2 *>(q) = Dwords2{(v[0] & 0xFFFFu) | (v[1] << 16), (v[2] & 0xFFFFu) | (v[3] << 16)};
                    else if (B == 3) { uint32_t d0, d1, d2; pack(v, d0, d1, d2); *reinterpret_cast<Dwords3 *>(q) = Dwords3{d0, d1, d2}; }
                    else *reinterpret_cast<Dwords4 *>(q) = Dwords4{v[0], v[1], v[2], v[3]};
                }
            }
            if (__any(ragged))
#pragma unroll 1
            for (uint32_t it = 0; it < 4; ++it) {
                const uint32_t item = it * 64 + lane, f = (item * inv_4c) >> 16, g = item - f * 4 * C;
                const uint32_t fc = f < F ? f : 0u;
                const uint64_t dst_f = frame_dst(fc);      // lane reads: before the lanes part ways
                const uint32_t meta = frame_meta(fc);
                if (f >= F) continue;
                const uint32_t fbs_f = meta & 0xFFFFFu;
                if (t0 >= fbs_f) continue;
                const uint32_t ncols = min(kTile, fbs_f - t0);   // sample columns of this tile that exist in the frame
                // written above?  WIDE: these four slots are half of run g >> 1 of the frame, which belongs to the item
                // of channel lane (g >> 1) % C, whose last slot is the end of its second run
                const uint32_t last_slot = wide_items ? 8 * (((g >> 1) % C) + C) + 7 : 4 * g + 3;
                if (dwords && ((last_slot * inv_c) >> 16) < ncols) continue;
                uint32_t v[4], nvalid = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t slot = 4 * g + j, col = (slot * inv_c) >> 16, chn = slot - col * C;
                    v[j] = tile_raw[col * RS + f * C + chn];
                    nvalid += col < ncols ? 1u : 0u;
                }
                if (C == 2) {   // slots are (a, b) of two consecutive samples
                    const uint32_t m = (meta >> 20) & 15u;
                    uint32_t L, R;
                    decorrelate(m, v[0], v[1], L, R); v[0] = L; v[1] = R;
                    decorrelate(m, v[2], v[3], L, R); v[2] = L; v[3] = R;
                }
                // a block that ends inside these four slots, or an odd destination: byte by byte
                uint8_t *at = p.out + dst_f + (uint64_t)t0 * CB + g * 4 * B;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)j < nvalid)
                        for (uint32_t y = 0; y < B; ++y) at[j * B + y] = (uint8_t)(v[j] >> (8 * y));
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // One extra group runs past the end of the block when some lane's subframe is all warm-up
    // samples: its predictor parameters still have to be consumed.
    RingFetch rf;
    rf.clear();
    // A refill round consumes the loads of the round before it, and a wave's loads and stores complete in
    // issue order: a round that follows a write-out waits for the write-out's stores as well.  So the round
    // between two groups runs before the write-out, not after it, and the stores have half a group to
    // complete before the next round (in the middle of the next group) has to wait for them.
    ring_round(br, rf, reads_stream && status == kFrameOK);
    ring_round(br, rf, reads_stream && status == kFrameOK);
    ring_round(br, rf, reads_stream && status == kFrameOK);
    const bool trailing = __any(active && status == kFrameOK && kind == 2 && order >= bs);
    const uint32_t n_end = ((bs_max + U - 1) / U) * U + (trailing ? U : 0);
    for (uint32_t n0 = 0; n0 < n_end; n0 += U) {
        const bool on = active && status == kFrameOK && n0 < bs;
        if (on && kind == 2 && started && part_rem == 0) next_partition();
        if (on && br.pos() > limit) status = kFrameTruncated;
        const bool rice_on = on && kind == 2 && status == kFrameOK;
        const bool unstarted = active && status == kFrameOK && kind == 2 && !started;
        const bool simple = unstarted ? false
                                      : (!on || (kind == 0 && const_fast && widemask_rest == 0) ||
                                         (rice_on && !escaped && part_rem >= (uint32_t)U && n0 + U <= bs && (!DOT || hist16)));
        bool redo = true;            // lanes the generic body below has to process
        bool run_generic = variant::kNoFastGroups || !__all(simple);
        if (!run_generic) {
            // ---- fast group: U Rice codes per lane, speculatively and without branches ------
            // Every code is assumed to fit the 32-bit window and every word to be in the ring.
            // A lane for which that fails (a code longer than 32 bits, or the reader ran past
            // `fill`) has produced garbage: its state is rolled back to the start of the group
            // and the generic body redoes the group for it.  Lanes that have no residual to
            // read (constant subframes, blocks that have ended) let their reader run along and
            // get it back afterwards.
            const uint32_t s_P = br.P;
            govf = 0;
            gmax = 0; gmin = 0;
            if (DOT && !packs_ok) {
#pragma unroll
                for (int m = U - 1; m >= U - NT + 1; m -= 1) pk[m] = pack_lo_hi(ring[m], ring[m - 1]);
            }
            // what a rolled-back lane needs back: its last NT samples.  The 16-bit pair class keeps them as the NP packs
            // that hold them (a lane in an unrolled group has a history that fits 16 bits, or it has no residual)
            int32_t s_hist[DOT ? 1 : NT];
            uint32_t s_pk[DOT ? NP : 1];
            if constexpr (DOT) {
#pragma unroll
                for (int i = 0; i < NP; ++i) s_pk[i] = pk[U - 1 - 2 * i];
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) s_hist[j] = ring[U - 1 - j];
            }
            uint32_t longest_q = 0;
            bool starved = false;
            uint32_t Q = 0u - br.P;              // the read position, negated (see peek_ring_neg)
            const uint32_t neg_k1 = 0u - (k + 1);
            // The window of a code is fetched as soon as the code before it has given its length, ahead of that
            // code's own arithmetic and of its store to the tile: LDS operations stay in program order, and a
            // fetch placed after the store would leave the LDS latency exposed once per sample.  The window of
            // the code after a refill round is fetched before the round writes to the ring: the words it needs
            // were there already unless the lane is starved, which the round's test finds out.
            // The scheduler would sink the fetch below the store again (it sees a short latency and no conflict);
            // the scheduling barrier after the fetch keeps the rest of the sample's instructions behind it.
            uint2 raw = br.fetch_ring_neg(Q);
            uint32_t Qf = Q;
            // What follows a sample's Rice code u (the zigzag-coded residual): prediction, history, staging, and the
            // refill round or write-out that falls behind sample P.  PC: P as a compile-time constant.
            uint32_t *const tile_group = tile_raw + ((n0 & (kT - 1)) >> 1) * kPairStride + lane;   // TILE16: this lane's first pair of the group
            auto finish_sample = [&](auto PC, uint32_t u) {
                constexpr int P = decltype(PC)::value;
                const int32_t res = unzigzag(u);
                int32_t pred;
                if constexpr (DOT && variant::kNoPredictor) {
                    pred = 0;
                } else if constexpr (DOT) {
                    // pair i = taps 2i and 2i + 1 = samples n - 1 - 2i and n - 2 - 2i = pk[(n - 1 - 2i) % U]; two chains,
                    // and the pair with the newest sample, the only one that waits for the previous result, last
                    // one chain, oldest pair first: only its last link waits for the previous sample
                    int32_t sa = dot2_i16(pk[(P + 2 * U - 1 - 2 * (NP - 1)) % U], cp[NP - 1]);
#pragma unroll
                    for (int i = NP - 2; i >= 1; --i) sa = dot2_i16(pk[(P + 2 * U - 1 - 2 * i) % U], cp[i], sa);
                    pred = dot2_i16(pk[(P + U - 1) % U], cp[0], sa) >> shift;
                } else {
                    pred = predict([&](int j) { return ring[(P + U - 1 - j) % U]; });
                }
                int32_t decoded = (int32_t)((uint32_t)res + (uint32_t)pred);
                asm volatile("" : "+v"(decoded));   // keeps the arithmetic out of a branch: the select below stays a v_cndmask
                const int32_t v = decoded;
                ring[P] = v;
                if constexpr (DOT) pk[P] = pack_lo_hi(v, ring[(P + U - 1) % U]);
                emit(v, on, ((n0 + (uint32_t)P) & (kT - 1)), std::true_type{});
                if constexpr (TILE16 && (P & 1)) {   // samples P - 1 and P, both known to fit 16 bits if the lane keeps this group
                    const pair16 shifted = __builtin_bit_cast(pair16, pk[P]) << pair16{(short)h.wasted, (short)h.wasted};
                    tile_group[(P >> 1) * kPairStride] = __builtin_bit_cast(uint32_t, shifted);
                }
                if ((P + 1) % kRoundSteps == 0 && P + 1 < U) {   // the refill round between samples P and P + 1 (before a write-out, see above)
                    br.P = 0u - Q;
                    starved = starved || br.starved();
                    ring_round(br, rf, rice_on);
                }
                if (U > (int)kT && P == (int)kT - 1) flush(n0);
            };
            // (Two codes per window fetch, as the skim does, were tried here too: one fetch, address and funnel shift
            // less per pair, but the pair's lengths form a longer dependent chain, and the kernel ran 5 % slower.)
            {
                static_for_each<U>([&](auto PC) {
                    constexpr int P = decltype(PC)::value;
                    // Rice code: unary part = leading zeros of the window, then k low bits.  The code's length
                    // nb = q + 1 + k is kept negated: -nb mod 32 is where the low bits start, and it advances Q.
                    const uint32_t w = BitReader::window_neg(raw, Qf);
                    const uint32_t q = lead_zeros_raw(w);
                    longest_q |= q;
                    const uint32_t neg_nb = neg_k1 - q;
                    Q += neg_nb;
                    if (P + 1 < U) {
                        raw = br.fetch_ring_neg(Q);
                        Qf = Q;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const uint32_t low = bfe_mod32(w, neg_nb, k);
                    finish_sample(std::integral_constant<int, P>{}, (q << k) | low);
                });
            }
            br.P = 0u - Q;
            starved = starved || br.starved();
            // q + 1 + k > 32 for some code?  The OR of the counts is at least their maximum (a zero window counts
            // as 2^32 - 1 zeros), so the test never misses one; it is wrong the safe way now and then.
            redo = rice_on && (variant::kForceRedo || starved || longest_q > 31 - k);
            govf = ((uint32_t)gmax + half_fast) | ((uint32_t)gmin + half_fast);   // in range exactly when both ends are
            // a sample that does not fit 16 bits (less its wasted bits: the tile holds 16) was multiplied and staged as if it did
            if (pair17) redo = redo || (rice_on && (govf >> fast_bits));
            else if (!redo && (rice_on || kind == 0)) ovf |= govf;   // a lane past the end of its block decoded noise
            if (rice_on && !redo) part_rem -= U;
            run_generic = __any(redo);
            packs_ok = !run_generic;
            if (redo || !rice_on) br.P = s_P;
            if (redo) {
                // the refill round inside the group acted on the speculative read position: forget the
                // ring (words come from global memory until the next rounds have refilled it)
                br.fill = (s_P >> 5) & ~(kUnitWords - 1);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (DOT) ring[U - 1 - j] = (j & 1) ? (int32_t)s_pk[j >> 1] >> 16 : __builtin_amdgcn_sbfe((int32_t)s_pk[j >> 1], 0, 16);
                    else ring[U - 1 - j] = s_hist[j];
                }
            }
        }
        if (run_generic) {
            // ---- generic group: one sample at a time, shift-register history ----------------
            int32_t hist[NT];   // hist[j] = sample n-1-j
#pragma unroll
            for (int j = 0; j < NT; ++j) hist[j] = ring[U - 1 - j];
            // A group of two tiles: the lanes that keep their fast-group results have the second tile's
            // samples in the staging columns, and this loop writes the first tile out again.  They stage
            // each tile again from their history registers (ring[P] = sample n0 + P of the group).
            auto restage = [&](int first) {
                if (!redo) {
#pragma unroll
                    for (int P = 0; P < (int)kT; ++P)
                        tile_raw[(uint32_t)P * RS + lane] = (uint32_t)ring[(first + P) % U] << h.wasted;
                }
            };
            if (U > (int)kT) restage(0);
            for (uint32_t n = n0; n < n0 + U; ++n) {
                if (n != n0 && ((n - n0) % kRoundSteps) == 0)
                    ring_round(br, rf, reads_stream && status == kFrameOK);
                if (U > (int)kT && n - n0 == kT) { flush(n0); restage((int)kT); }
                const bool ok = active && status == kFrameOK && redo;
                const bool live = ok && n < bs;
                if (ok && kind == 2 && !started && n >= order) {
                    start_residual();
                    started = true;
                    if (n >= bs) next_partition();
                }
                int32_t v = constval;
                bool predicted = false;
                if (live && kind != 0 && status == kFrameOK) {
                    if (kind == 1 || n < order) {
                        v = br.read_signed(h.sbps);
                    } else {
                        if (part_rem == 0) {
                            next_partition();
                            if (status == kFrameOK && part_rem == 0) status = kFrameTruncated;
                        }
                        if (status == kFrameOK) {
                            --part_rem;
                            v = escaped ? br.read_signed(escw) : read_rice(br, k, limit);
                            predicted = true;
                        }
                    }
                }
                const int32_t pred = predict([&](int j) { return hist[j]; });
                if (predicted) v = (int32_t)((uint32_t)v + (uint32_t)pred);
                if (redo) {   // lanes that kept their fast-group result keep their history
#pragma unroll
                    for (int j = NT - 1; j > 0; --j) hist[j] = hist[j - 1];
                    hist[0] = v;
                }
                emit(v, live && status == kFrameOK, n & (kT - 1), std::false_type{});
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) ring[U - 1 - j] = hist[j];
            if (DOT) {
                uint32_t t = 0;
#pragma unroll
                for (int j = 0; j < NT; ++j) t |= (uint32_t)hist[j] + 0x8000u;
                hist16 = (t >> 16) == 0;
                packs_ok = false;
            }
        }
        const uint32_t done = n0 + U;
        ring_round(br, rf, reads_stream && status == kFrameOK);
        if ((done & (kT - 1)) == 0 || done >= n_end) {
            const uint32_t t0 = (done - 1) & ~(kT - 1);   // tile that holds the group's last sample
            if (t0 < bs_max) flush(t0);
        }
    }

    if (active) {
        if (status == kFrameOK && sbits < 32 && (ovf >> sbits)) status = kFrameBadSubframe;   // a sample wider than its subframe
        if (status == kFrameOK) {
            // a subframe ends where the next begins; the last one ends at the CRC-16, byte aligned
            const uint32_t end = br.pos();
            if (ch + 1 == C) {
                if (end > limit) status = kFrameTruncated;
                else if (((end + 7) & ~7u) != limit - 16) status = kFrameLength;
            } else if (end != p.c.subpos[(uint64_t)r * 8 + ch + 1]) {
                status = end > limit ? kFrameTruncated : kFrameLength;
            }
        }
        if (status != kFrameOK) atomicMax(&p.c.status[r], status);
    }
    probe.end("decode", 1024);
}

// ORs the class flags of one run of streams into a mask (bit = class index) that the timing report carries.
__global__ void k_or_class_flags(const uint32_t *__restrict__ flags, uint32_t *__restrict__ mask)
{
    if (threadIdx.x < 8 && flags[threadIdx.x]) atomicOr(mask, 1u << threadIdx.x);
}

}  // namespace flacgpu
