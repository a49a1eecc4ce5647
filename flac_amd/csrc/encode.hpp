// encode.hpp — gfx950 kernels of the batched FLAC encoder (RFC 9639 frames behind a
// STREAMINFO block laid out as flacmeta.go:644-717 reads it).
//
// One encode call runs:
//   k_enc_stats    workgroup per (frame, candidate channel): wasted bits, constant test, fixed-predictor
//                  error sums, windowed autocorrelation
//   k_enc_design   lane per candidate: Levinson-Durbin, order choice, coefficient quantisation
//   k_enc_choose   workgroup per candidate: residuals, Rice partition search -> a "plan" with the exact
//                  subframe size in bits
//   k_enc_layout   workgroup per stream: stereo decision per frame, frame sizes, their prefix sum (= where
//                  every frame and subframe goes), frame headers, STREAMINFO
//   k_enc_pack     workgroup per (frame, channel): residuals again, code lengths, their prefix sum, bits
//                  assembled in LDS and copied to the frame's place
//   (CRC-16 of every frame by the decoder's span kernel), k_enc_write_crc
//   k_enc_md5      lane per stream, optional
//
// A block of bs samples is cut into runs of kRun = 16 consecutive samples, one per lane.  All
// floating-point sums are formed in a fixed order (per lane ascending, butterfly over the wave,
// waves ascending) so that a scalar restatement of the same order gives the same bits.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace flacgpu {

constexpr int kRun = 16;
constexpr int kEncMaxBs = 16384;
constexpr int kEncMaxPo = 8;
constexpr int kPlanHeadBytes = 16;
constexpr int kPlanCoefBytes = 64;   // 32 x int16

// Samples of a block in LDS: kFront zeros in front (history of the first lane, whole runs), then the block.
// One unused word follows every run of 16, so a lane's run starts 17 words after its neighbour's and the
// lanes of a wave, each reading its own run and the ones before it, fall into different banks.
constexpr int kFront = 48;
__host__ __device__ constexpr uint32_t enc_lds_words(int NT) { return (uint32_t)((kFront + NT * kRun) / kRun * (kRun + 1) + 1); }
__device__ inline uint32_t lds_pos(uint32_t j) { return j + (j >> 4); }            // j = kFront + sample index
// Position of sample t + j * NT (the lane's j-th sample as the block is loaded): the lane's first position plus
// a constant, since NT is a multiple of the run length.
template <int NT>
__device__ inline uint32_t lds_share_pos(uint32_t t, int j) { return lds_pos((uint32_t)kFront + t) + (uint32_t)(j * (NT + NT / kRun)); }
__device__ inline uint32_t lds_lane_base(uint32_t t) { return (uint32_t)(kFront + kFront / kRun) + t * (kRun + 1); }
// word offset from lds_lane_base of the sample r places after the lane's first one (r may be negative)
__host__ __device__ constexpr int lds_rel(int r) { return r + (r >= 0 ? r / kRun : -((-r + kRun - 1) / kRun)); }

struct EncStreamDev {
    uint64_t in_off;          // bytes from d_pcm to sample 0
    uint64_t out_off;         // bytes from d_out to the "fLaC" marker
    uint64_t out_cap;
    uint64_t total_samples;   // inter-channel
    uint64_t out_len;         // result: bytes written
    uint32_t frame_begin;     // index of frame 0 in the frame arrays
    uint32_t n_frames;
    uint32_t unit_begin;      // index of (frame 0, candidate 0) in the plan array
    uint32_t pack_begin;      // index of (frame 0, channel 0) among the pack units
    uint32_t sample_rate;
    uint32_t tail_bs;         // samples in the last frame
    uint32_t win_off;         // floats from the window table to the window of a full block
    uint32_t win_off_tail;    // ... of the last block
    uint32_t min_fs, max_fs;  // result
    int32_t  status;          // result
    uint8_t  channels, bps, bytes_ps, ncand;
    uint8_t  md5[16];
    uint32_t meta_len;        // bytes in front of frame 0: 42, plus 4 + 18 per seek point
    int32_t  pair;            // kPairNone: every candidate channel has a plan (ncand of them per frame);
                              // 0..3: stereo, this channel pair in every frame, two plans per frame;
                              // kPairEstimated: stereo, the pair of each frame is in EncOptsDev::modes
    uint32_t pad;
};
constexpr int kPairNone = -1, kPairEstimated = 4;

struct EncOptsDev {
    uint32_t blocksize, max_lpc_order, qlp_precision, min_po, max_po;
    int32_t  stereo_mode;
    uint32_t force_lpc_order;
    uint32_t plan_stride;     // bytes per plan
    uint32_t seek_every;      // frames per seek point, 0 = no SEEKTABLE block
    uint32_t total_units, total_pack, total_frames;
    uint8_t *modes;           // per frame of a kPairEstimated stream: its channel pair (0..3), written by k_enc_pick
};

struct PlanHead {             // kPlanHeadBytes
    uint32_t bits;            // exact size of the subframe
    uint8_t  type;            // 0 constant, 1 verbatim, 2 fixed, 3 lpc
    uint8_t  order, wasted, sb, prec, shift, po, method;
    uint32_t pad;
};

struct FrameRec {             // 64 bytes
    uint64_t out_off;         // byte offset of the frame in d_out
    uint32_t bytes;           // 0: not written (its stream does not fit its capacity)
    uint32_t stream;
    uint32_t bs;
    uint8_t  hdr_len, ch_assign;
    uint8_t  cand[8];         // candidate channel coded as each channel (its plan: that candidate's, or the
                              // channel's own in a stream with two plans per frame)
    uint8_t  pad[2];
    uint32_t sf_bit[8];       // bit offset of each subframe from the frame start
};

// ------------------------------------------------------------------ helpers ---

// Lane exchanges inside a row of 16 lanes as DPP moves (vector ALU, no LDS traffic): partners at distance
// 1 and 2 inside each quad, then the mirror images inside 8 and inside 16 lanes.  After the first two
// steps a quad holds one value, after the third an 8-lane group does, so the mirrors pair the same groups
// as partners at distance 4 and 8 would.
#define FLACGPU_DPP_XOR1 0xB1         /* quad_perm [1,0,3,2] */
#define FLACGPU_DPP_XOR2 0x4E         /* quad_perm [2,3,0,1] */
#define FLACGPU_DPP_HALF_MIRROR 0x141
#define FLACGPU_DPP_MIRROR 0x140
template <int CTRL>
__device__ inline uint32_t dpp_move(uint32_t v)
{
    // every lane has a source lane for these patterns, so the previous value of the destination is dead:
    // bound_ctrl with a constant spares the copy the compiler would make to preserve it
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ inline uint64_t dpp_move64(uint64_t v)
{
    return ((uint64_t)dpp_move<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_move<CTRL>((uint32_t)v);
}
template <int CTRL>
__device__ inline double dpp_move_f64(double v)
{
    return __hiloint2double((int)dpp_move<CTRL>((uint32_t)__double2hiint(v)), (int)dpp_move<CTRL>((uint32_t)__double2loint(v)));
}
// Reductions over a row of 16 lanes: every lane of the row ends with the row's value.
__device__ inline uint32_t row_or(uint32_t v)
{
    v |= dpp_move<FLACGPU_DPP_XOR1>(v); v |= dpp_move<FLACGPU_DPP_XOR2>(v);
    v |= dpp_move<FLACGPU_DPP_HALF_MIRROR>(v); v |= dpp_move<FLACGPU_DPP_MIRROR>(v);
    return v;
}
__device__ inline uint32_t row_sum32(uint32_t v)
{
    v += dpp_move<FLACGPU_DPP_XOR1>(v); v += dpp_move<FLACGPU_DPP_XOR2>(v);
    v += dpp_move<FLACGPU_DPP_HALF_MIRROR>(v); v += dpp_move<FLACGPU_DPP_MIRROR>(v);
    return v;
}
__device__ inline uint64_t row_sum64(uint64_t v)
{
    v += dpp_move64<FLACGPU_DPP_XOR1>(v); v += dpp_move64<FLACGPU_DPP_XOR2>(v);
    v += dpp_move64<FLACGPU_DPP_HALF_MIRROR>(v); v += dpp_move64<FLACGPU_DPP_MIRROR>(v);
    return v;
}
// Butterfly sum over a row: partners at distance 1, 2, 4, 8; every lane ends with the same value.
__device__ inline double row_butterfly(double v)
{
    v = v + dpp_move_f64<FLACGPU_DPP_XOR1>(v); v = v + dpp_move_f64<FLACGPU_DPP_XOR2>(v);
    v = v + dpp_move_f64<FLACGPU_DPP_HALF_MIRROR>(v); v = v + dpp_move_f64<FLACGPU_DPP_MIRROR>(v);
    return v;
}
__device__ inline uint32_t wave_or(uint32_t v)
{
    v = row_or(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ inline uint32_t wave_sum32(uint32_t v)
{
    v = row_sum32(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ inline uint32_t wave_min32(uint32_t v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, m); v = o < v ? o : v; }
    return v;
}
__device__ inline uint32_t wave_max32(uint32_t v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, m); v = o > v ? o : v; }
    return v;
}
__device__ inline uint64_t wave_sum64(uint64_t v)
{
    v = row_sum64(v);
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    uint64_t r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16)
        r += ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, l);
    return r;
}

// Exclusive prefix sum over the workgroup; s_w needs NT / 64 entries.
template <int NT>
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= (uint32_t)d) inc += n;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t x = s_w[w];
        if ((uint32_t)w < wave) base += x;
        tot += x;
    }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

__device__ inline int ilog2_32(uint32_t v) { return 31 - __clz((int)v); }
__device__ inline int ilog2_64(uint64_t v) { return 63 - __clzll((long long)v); }
__device__ inline uint32_t fold(int32_t r) { return ((uint32_t)r << 1) ^ (uint32_t)(r >> 31); }

// Signed little-endian sample of B bytes.
__device__ inline int32_t load_le(const uint8_t *p, int B)
{
    uint32_t v;
    if (B == 2) v = *reinterpret_cast<const uint16_t *>(p);
    else {
        v = p[0];
        if (B > 2) v |= ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
    const int sh = 32 - 8 * B;
    return (int32_t)(v << sh) >> sh;
}

// Sample i of candidate channel `cand` of a frame whose interleaved PCM starts at src.
// Stereo streams have four candidates: left, right, mid = (l + r) >> 1, side = l - r.
__device__ inline int32_t cand_sample(const uint8_t *src, uint32_t i, int C, int B, int cand, bool stereo4)
{
    if (stereo4) {
        int32_t l, r;
        if (B == 2) {
            const uint32_t lr = *reinterpret_cast<const uint32_t *>(src + (size_t)i * 4);
            l = (int32_t)(lr << 16) >> 16;
            r = (int32_t)lr >> 16;
        } else {
            l = load_le(src + (size_t)(2 * i) * B, B);
            r = load_le(src + (size_t)(2 * i + 1) * B, B);
        }
        return cand == 0 ? l : cand == 1 ? r : cand == 2 ? (l + r) >> 1 : l - r;
    }
    return load_le(src + ((size_t)i * C + cand) * B, B);
}

// The lane's share of a block as it is loaded: samples t, t + NT, t + 2 NT, ... (consecutive lanes read
// consecutive samples).  Every load is issued, with the index clamped into the block, so that the 16 of them
// overlap; samples beyond the block come back as zero.  16-bit stereo, the common case, has its own loop.
template <int NT>
__device__ inline void load_block_share(const uint8_t *src, uint32_t bs, int C, int B, int cand, bool stereo, int32_t (&v)[kRun])
{
    const uint32_t t = threadIdx.x;
    if (stereo && B == 2) {
        uint32_t lr[kRun];
        if (bs == (uint32_t)(NT * kRun)) {   // a whole block: one address per lane and constant offsets, no index to clamp
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src) + t;
#pragma unroll
            for (int j = 0; j < kRun; ++j) lr[j] = p[j * NT];
        } else {
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                const uint32_t i = t + (uint32_t)j * NT;
                lr[j] = *reinterpret_cast<const uint32_t *>(src + (size_t)(i < bs ? i : bs - 1) * 4);
            }
        }
        // one loop per candidate: the choice is the same for the whole workgroup
        if (cand == 0) {
#pragma unroll
            for (int j = 0; j < kRun; ++j) v[j] = (int32_t)(lr[j] << 16) >> 16;
        } else if (cand == 1) {
#pragma unroll
            for (int j = 0; j < kRun; ++j) v[j] = (int32_t)lr[j] >> 16;
        } else if (cand == 2) {
#pragma unroll
            for (int j = 0; j < kRun; ++j) v[j] = (((int32_t)(lr[j] << 16) >> 16) + ((int32_t)lr[j] >> 16)) >> 1;
        } else {
#pragma unroll
            for (int j = 0; j < kRun; ++j) v[j] = ((int32_t)(lr[j] << 16) >> 16) - ((int32_t)lr[j] >> 16);
        }
        if (t + (uint32_t)(kRun - 1) * NT >= bs) {   // only blocks shorter than the workgroup's span
#pragma unroll
            for (int j = 0; j < kRun; ++j)
                if (t + (uint32_t)j * NT >= bs) v[j] = 0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = t + (uint32_t)j * NT;
            v[j] = cand_sample(src, i < bs ? i : bs - 1, C, B, cand, stereo);
        }
#pragma unroll
        for (int j = 0; j < kRun; ++j)
            if (t + (uint32_t)j * NT >= bs) v[j] = 0;
    }
}

// Stream that owns plan unit (which = 0) or pack unit (which = 1) idx.  hints[k] is the owner of unit k << 10
// (built on the host), so the search runs over the few streams between two hints.
__device__ inline int find_enc_stream(const EncStreamDev *s, uint32_t n, const uint32_t *hints, uint32_t idx, int which)
{
    uint32_t lo = hints[idx >> 10], hi = hints[(idx >> 10) + 1] + 1;   // last stream in [lo, hi) whose begin <= idx
    if (hi > n) hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t b = which == 0 ? s[mid].unit_begin : s[mid].pack_begin;
        if (b <= idx) lo = mid; else hi = mid;
    }
    return (int)lo;
}

// Workgroups are dealt to the 8 XCDs round-robin; this makes consecutive units share an XCD (and its L2).
__device__ inline uint32_t xcd_unit(uint32_t bid, uint32_t grid)
{
    const uint32_t per = grid >> 3;
    return (bid & 7) * per + (bid >> 3);
}

// log2 of a positive normal number: exponent plus a cubic fit on [1, 2), plain multiplications and additions.
__device__ inline double est_log2(double v)
{
    uint64_t b = (uint64_t)__double_as_longlong(v);
    const int e = (int)((b >> 52) & 0x7FF) - 1023;
    b = (b & 0xFFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    const double m = __longlong_as_double((long long)b);
    const double x = m - 1.0;
    double p = 0x1.48f04fbb3fa68p-4 * x;
    p = p + -0x1.e4bf84b19febep-3;
    p = p * x;
    p = p + 0x1.c09645d8fab54p-2;
    double y = x * (1.0 - x);
    y = y * p;
    return ((double)e + x) + y;
}

__device__ inline int default_precision(int bps, uint32_t bs)
{
    if (bps < 16) { const int p = 2 + bps / 2; return p < 5 ? 5 : p; }
    if (bps == 16)
        return bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
    return bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
}

// Rice parameter of one partition from the sum of its folded residuals: the cheapest of four parameters
// around log2(sum / count) by the estimate count * (k + 1) + (sum >> k).
__device__ inline uint32_t rice_param(uint64_t sum, uint32_t cnt, uint32_t &est)
{
    if (cnt == 0) { est = 0; return 0; }
    if (sum == 0) { est = cnt; return 0; }
    if ((sum >> 31) == 0) {
        // The common case in 32-bit arithmetic, the four candidates side by side: cnt <= 65536 and k + 1 <= 31, so
        // cnt * (k + 1) + (sum >> k) stays below 2^32 and every value is the one the 64-bit loop below would find.
        const uint32_t s32 = (uint32_t)sum;
        const int k0 = ilog2_32(s32) - ilog2_32(cnt);   // at most 30
        int lo = k0 - 2, hi = k0 + 1;
        lo = lo < 0 ? 0 : lo;
        hi = hi < 0 ? 0 : hi > 30 ? 30 : hi;
        uint32_t best = 0xFFFFFFFFu, base = cnt * (uint32_t)(lo + 1);
        int bk = lo;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int k = lo + d;
            const uint32_t b = base + (s32 >> (k & 31));
            if (k <= hi && b < best) { best = b; bk = k; }
            base += cnt;
        }
        est = best;
        return (uint32_t)bk;
    }
    const int k0 = ilog2_64(sum) - ilog2_32(cnt);
    int lo = k0 - 2, hi = k0 + 1;
    lo = lo < 0 ? 0 : lo > 30 ? 30 : lo;
    hi = hi < 0 ? 0 : hi > 30 ? 30 : hi;
    uint64_t best = ~0ull;
    int bk = lo;
    for (int k = lo; k <= hi; ++k) {
        const uint64_t b = (uint64_t)cnt * (uint64_t)(k + 1) + (sum >> k);
        if (b < best) { best = b; bk = k; }
    }
    est = (uint32_t)best;
    return (uint32_t)bk;
}

// Largest usable partition order: whole runs per partition and room for the warm-up in partition 0.
__device__ inline int max_partition_order(uint32_t bs, uint32_t order, uint32_t max_po)
{
    int L = (int)(max_po > (uint32_t)kEncMaxPo ? (uint32_t)kEncMaxPo : max_po);
    while (L > 0 && ((bs % ((uint32_t)kRun << L)) != 0 || (bs >> L) < order + 1)) --L;
    return L;
}

// Residual of a fixed predictor, 32-bit wrapping arithmetic.  xs: the block in LDS (see lds_pos), lbase: the
// lane's run.
template <int ORDER>
__device__ inline void fixed_residual_of(const int32_t (&w)[4 + kRun], uint32_t (&u)[kRun])
{
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
        const uint32_t a = (uint32_t)w[4 + j], b = (uint32_t)w[3 + j], c = (uint32_t)w[2 + j], d = (uint32_t)w[1 + j],
                       e = (uint32_t)w[j];
        const uint32_t r = ORDER == 0 ? a : ORDER == 1 ? a - b : ORDER == 2 ? a - 2 * b + c : ORDER == 3 ? a - 3 * b + 3 * c - d
                                                                                              : a - 4 * b + 6 * c - 4 * d + e;
        u[j] = fold((int32_t)r);
    }
}
__device__ inline void fixed_residual(const int32_t *xs, uint32_t lbase, int order, uint32_t (&u)[kRun])
{
    int32_t w[4 + kRun];
#pragma unroll
    for (int m = 0; m < 4 + kRun; ++m) w[m] = xs[lbase + lds_rel(m - 4)];
    // the order is the same for the whole workgroup: one loop per order
    switch (order) {
    case 0: fixed_residual_of<0>(w, u); break;
    case 1: fixed_residual_of<1>(w, u); break;
    case 2: fixed_residual_of<2>(w, u); break;
    case 3: fixed_residual_of<3>(w, u); break;
    default: fixed_residual_of<4>(w, u); break;
    }
}

// |a - b| + c on unsigned values, one instruction (the compiler spells __usad as max - min + c).
__device__ inline uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ inline int32_t mad_i24(int32_t a, int32_t b, int32_t c)
{
    int32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// lo(a) * lo(b) + hi(a) * hi(b) + c on signed 16-bit halves, the sum wrapping in 32 bits: two taps of a predictor whose
// samples and coefficients fit 16 bits in one instruction.
__device__ inline int32_t dot2_i16(uint32_t a, uint32_t b, int32_t c)
{
    int32_t r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (lo, hi) -> lo's low half in bits 0-15, hi's low half in bits 16-31
__device__ inline uint32_t pack_halves(int32_t lo, int32_t hi)
{
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}

// Bits [lo, hi) of a lane's 16-sample run, as a mask: the samples i0 + j with first <= i0 + j < end.
__device__ inline uint32_t run_mask(uint32_t i0, uint32_t first, uint32_t end)
{
    const uint32_t lo = first > i0 ? (first - i0 > (uint32_t)kRun ? (uint32_t)kRun : first - i0) : 0;
    const uint32_t hi = end > i0 ? (end - i0 > (uint32_t)kRun ? (uint32_t)kRun : end - i0) : 0;
    return hi > lo ? ((1u << hi) - 1) & ~((1u << lo) - 1) : 0;
}

// LPC residual of the lane's run.  xs / lbase: the block in LDS and the lane's run; cq: the M coefficients in
// LDS (zero beyond the order).  MODE 0: 24-bit multiply-adds (samples of at most 24 bits, coefficients of at
// most 15), 1: 32-bit wrapping sums, 2: 64-bit sums, 3: as 0 for samples of at most 16 bits, two taps per instruction
// (the same sums modulo 2^32).  The taps are taken in passes of at most 16 so that a
// pass's coefficients, its window of samples and the 16 running sums fit the registers of a 1024-lane
// workgroup.  The result is nonzero when some residual of the run may not fit 32 bits (r + 2^31 - 1 has bits
// above 31 exactly when r is outside [-(2^31 - 1), 2^31 - 1]; MODE 0 has a cheaper test that also reports
// some residuals that fit); the caller then finds out which with lpc_residual_bad.
template <int M, int MODE>
__device__ inline uint32_t lpc_residual(const int32_t *xs, uint32_t lbase, const int32_t *cq, int shift, uint32_t (&u)[kRun])
{
    constexpr int TP = M <= 16 ? M : 16;        // taps per pass
    using acc_t = typename std::conditional<MODE == 2, int64_t, uint32_t>::type;
    acc_t acc[kRun];
#pragma unroll
    for (int j = 0; j < kRun; ++j) acc[j] = 0;
#pragma unroll
    for (int base = 0; base < M; base += TP) {
        // the run's position goes through an empty asm so that the compiler treats every pass's window as
        // loads of its own: it can neither start them before the previous pass's arithmetic nor merge the
        // windows of the three arithmetic variants a caller branches between (which made all of them live)
        uint32_t lb = lbase, cb = 0;
        asm volatile("" : "+v"(lb), "+v"(cb));
        int32_t c[TP], w[TP + kRun - 1];        // w[m] = sample (j - 1 - q) for m = (TP - 1 - qq) + j, q = base + qq
#pragma unroll
        for (int qq = 0; qq < TP; ++qq) c[qq] = cq[cb + base + qq];
#pragma unroll
        for (int m = 0; m < TP + kRun - 1; ++m) w[m] = xs[lb + lds_rel(m - TP - base)];
        if constexpr (MODE == 3) {
            static_assert(TP % 2 == 0, "taps in pairs");
            uint32_t cp[TP / 2], sp[TP + kRun - 1];   // (c[2p], c[2p + 1]) and (w[m], w[m - 1])
#pragma unroll
            for (int pp = 0; pp < TP / 2; ++pp) cp[pp] = pack_halves(c[2 * pp], c[2 * pp + 1]);
#pragma unroll
            for (int m = 1; m < TP + kRun - 1; ++m) sp[m] = pack_halves(w[m], w[m - 1]);
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                int32_t sum = (int32_t)acc[j];
#pragma unroll
                for (int qq = 0; qq < TP; qq += 2) sum = dot2_i16(cp[qq / 2], sp[TP - 1 - qq + j], sum);
                acc[j] = (acc_t)(uint32_t)sum;
            }
        } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            if (MODE == 2) {
#pragma unroll
                for (int qq = 0; qq < TP; ++qq) acc[j] += (int64_t)c[qq] * (int64_t)w[TP - 1 - qq + j];
            } else if (MODE == 1) {
#pragma unroll
                for (int qq = 0; qq < TP; ++qq) acc[j] += (uint32_t)c[qq] * (uint32_t)w[TP - 1 - qq + j];
            } else {
                int32_t s0 = (int32_t)acc[j], s1 = 0;
#pragma unroll
                for (int qq = 0; qq < TP; qq += 2) {
                    s0 = mad_i24(c[qq], w[TP - 1 - qq + j], s0);
                    if (qq + 1 < TP) s1 = mad_i24(c[qq + 1], w[TP - 2 - qq + j], s1);
                }
                acc[j] = (acc_t)((uint32_t)s0 + (uint32_t)s1);
            }
        }
        }
    }
    uint32_t high = 0, lb2 = lbase;
    asm volatile("" : "+v"(lb2));
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
        if (MODE == 0 || MODE == 3) {
            // samples of at most 24 bits: a prediction inside [-2^30, 2^30) leaves a residual that fits with room
            // to spare; anything else is reported (bit 31 of the sum below) and settled by lpc_residual_bad
            const int32_t pred = (int32_t)(uint32_t)acc[j] >> shift;
            high |= (uint32_t)pred + 0x40000000u;
            u[j] = fold((int32_t)((uint32_t)xs[lb2 + j] - (uint32_t)pred));
        } else {
            const int64_t pred = MODE == 2 ? (int64_t)acc[j] >> shift : (int64_t)((int32_t)(uint32_t)acc[j] >> shift);
            const int64_t r = (int64_t)xs[lb2 + j] - pred;
            high |= (uint32_t)((uint64_t)(r + 0x7FFFFFFFLL) >> 32);
            u[j] = fold((int32_t)r);
        }
    }
    if (MODE == 0 || MODE == 3) high &= 0x80000000u;
    return high;
}

// Which residuals of the run do not fit 32 bits (bit j): the slow restatement of lpc_residual's test.
template <int M>
__device__ __noinline__ uint32_t lpc_residual_bad(const int32_t *xs, uint32_t lbase, const int32_t *cq, int shift, bool wide)
{
    uint32_t bad = 0;
    for (int j = 0; j < kRun; ++j) {
        int64_t pred;
        if (wide) {
            int64_t s = 0;
            for (int q = 0; q < M; ++q) s += (int64_t)cq[q] * (int64_t)xs[lbase + lds_rel(j - 1 - q)];
            pred = s >> shift;
        } else {
            uint32_t s = 0;
            for (int q = 0; q < M; ++q) s += (uint32_t)cq[q] * (uint32_t)xs[lbase + lds_rel(j - 1 - q)];
            pred = (int32_t)s >> shift;
        }
        const int64_t r = (int64_t)xs[lbase + j] - pred;
        bad |= (uint32_t)((r != (int64_t)(int32_t)r) || (r == -2147483647LL - 1)) << j;
    }
    return bad;
}

// ----------------------------------------------------------------- analysis ---
//
// Three kernels per encode call:
//   k_enc_stats   workgroup per candidate channel: wasted bits, constant test, fixed-predictor error
//                 sums, windowed autocorrelation                          -> CandStats
//   k_enc_design  LANE per candidate channel: Levinson-Durbin, order choice, coefficient quantisation
//                 (serial arithmetic, so one lane each and registers only) -> CandStats
//   k_enc_choose  workgroup per candidate channel: residuals of the best fixed predictor and of the LPC
//                 predictor, Rice partition search for both, cheapest subframe -> plan

struct CandStats {            // followed by (M + 1) doubles: autocorrelation lags 0..M
    uint32_t constant;        // all samples equal: the plan is already written
    uint32_t wasted;
    uint32_t fo;              // best fixed order
    uint32_t lpc_ok;
    int32_t  ord, prec, shift, wide;
    int16_t  q[32];
};
constexpr int kStatsHeadBytes = 96;
__host__ __device__ inline uint32_t stats_stride(int M) { return (uint32_t)(kStatsHeadBytes + (M + 1) * 8); }

// Candidate channel a workgroup (or lane) works on.
struct UnitInfo {
    const EncStreamDev *st;
    uint32_t f, cand, bs;
    int C, B, cbps;
    bool stereo, last;
    bool valid;               // false: a padding unit behind the stream's last frame (unit ranges start even)
    const uint8_t *src;
};

// Candidate channels (0 left, 1 right, 2 mid, 3 side) of the two subframes of a stereo pair
// (0 left/right, 1 left/side, 2 side/right, 3 mid/side).
__host__ __device__ inline uint32_t pair_cand(int mode, uint32_t slot)
{
    return slot == 0 ? (mode < 2 ? 0u : mode == 2 ? 3u : 2u) : ((mode == 0 || mode == 2) ? 1u : 3u);
}

__device__ inline UnitInfo unit_info(const uint8_t *pcm, const EncStreamDev *streams, uint32_t n_streams, const uint32_t *hints,
                                     const EncOptsDev &o, uint32_t unit)
{
    UnitInfo u;
    const int si = find_enc_stream(streams, n_streams, hints, unit, 0);
    u.st = &streams[si];
    const uint32_t ncand = u.st->ncand, rel = unit - u.st->unit_begin;
    u.f = rel / ncand;
    u.cand = rel - u.f * ncand;
    u.valid = u.f < u.st->n_frames;
    if (u.valid && u.st->pair != kPairNone)
        u.cand = pair_cand(u.st->pair == kPairEstimated ? (int)o.modes[u.st->frame_begin + u.f] : u.st->pair, u.cand);
    u.last = u.f + 1 == u.st->n_frames;
    u.bs = u.last ? u.st->tail_bs : o.blocksize;
    u.C = u.st->channels; u.B = u.st->bytes_ps;
    u.stereo = u.C == 2;
    u.cbps = u.st->bps + (u.stereo && u.cand == 3 ? 1 : 0);
    u.src = pcm + u.st->in_off + (uint64_t)u.f * o.blocksize * (uint32_t)(u.C * u.B);
    return u;
}

template <int NT>
struct StatsShared {
    uint64_t e[5][16];               // fixed-predictor error sums, spread over 16 slots by lane (integer sums: any order)
    double   rsum[NT / 16][17];      // per row of 16 lanes: autocorrelation sums of one lag group
    uint32_t ored, neq;
};

template <int M, int NT>
__global__ __launch_bounds__(NT) void k_enc_stats(const uint8_t *__restrict__ pcm, const EncStreamDev *__restrict__ streams,
                                                  uint32_t n_streams, const uint32_t *__restrict__ hints, EncOptsDev o,
                                                  const float *__restrict__ windows, uint8_t *__restrict__ stats,
                                                  uint8_t *__restrict__ plans, uint32_t unit_base, uint32_t unit_end)
{
    extern __shared__ __align__(16) int32_t xs[];   // enc_lds_words(NT): see lds_pos
    __shared__ StatsShared<NT> S;
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint32_t unit = unit_base + xcd_unit(blockIdx.x, gridDim.x);
    if (unit >= unit_end) return;
    const UnitInfo u = unit_info(pcm, streams, n_streams, hints, o, unit);
    if (!u.valid) return;
    const uint32_t bs = u.bs;
    const float *win = windows + (u.last ? u.st->win_off_tail : u.st->win_off);
    CandStats *cs = reinterpret_cast<CandStats *>(stats + (uint64_t)unit * stats_stride(M));
    double *cs_ac = reinterpret_cast<double *>(stats + (uint64_t)unit * stats_stride(M) + kStatsHeadBytes);

    if (t == 0) { S.ored = 0; S.neq = 0; }
    if (t < 80) (&S.e[0][0])[t] = 0;
    // ---- load, wasted bits, constant test
    int32_t v[kRun];
    uint32_t ored = 0, neq = 0;
    const int32_t first = cand_sample(u.src, 0, u.C, u.B, (int)u.cand, u.stereo);
    load_block_share<NT>(u.src, bs, u.C, u.B, (int)u.cand, u.stereo, v);
    if (bs == (uint32_t)(NT * kRun)) {   // the block fills the workgroup's span: no sample to leave out
#pragma unroll
        for (int j = 0; j < kRun; ++j) { ored |= (uint32_t)v[j]; neq |= (uint32_t)(v[j] ^ first); }
    } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = t + (uint32_t)j * NT;
            ored |= (uint32_t)v[j];
            neq |= i < bs ? (uint32_t)(v[j] ^ first) : 0u;
        }
    }
    __syncthreads();
    ored = wave_or(ored); neq = wave_or(neq);
    if (lane == 0) { if (ored) atomicOr(&S.ored, ored); if (neq) atomicOr(&S.neq, neq); }
    __syncthreads();
    int wasted = S.ored ? __ffs((int)S.ored) - 1 : 0;
    if (wasted >= u.cbps) wasted = u.cbps - 1;
    const int sb = u.cbps - wasted;
    if (S.neq == 0) {
        if (t == 0) {
            PlanHead h = {};
            h.bits = 8 + (uint32_t)wasted + (uint32_t)sb; h.type = 0; h.wasted = (uint8_t)wasted; h.sb = (uint8_t)sb;
            *reinterpret_cast<PlanHead *>(plans + (uint64_t)unit * o.plan_stride) = h;
            cs->constant = 1; cs->wasted = (uint32_t)wasted; cs->fo = 0; cs->lpc_ok = 0;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kRun; ++j) xs[lds_share_pos<NT>(t, j)] = v[j] >> wasted;
    if (t < lds_lane_base(0)) xs[t] = 0;
    __syncthreads();

    const uint32_t i0 = t * kRun, lbase = lds_lane_base(t);
    int32_t mine[kRun];
    // ---- fixed predictors: sums of absolute errors of orders 0..4 over samples 4 .. bs-1
    {
        int32_t w[4 + kRun];
#pragma unroll
        for (int m = 0; m < 4 + kRun; ++m) w[m] = xs[lbase + lds_rel(m - 4)];
#pragma unroll
        for (int j = 0; j < kRun; ++j) mine[j] = w[4 + j];
        // 8 errors of at most 2^28 fit 32 bits; the two halves of the run are widened and added.  Only the
        // first lane (samples 0..3 do not count) and the lanes at the end of the block need the range test.
        uint64_t e64[5] = {0, 0, 0, 0, 0};
        const uint32_t counted = run_mask(i0, 4, bs);
        if (__all(counted == 0xFFFFu)) {
            // Differences of order k are differences of the differences of order k-1.  Everything is kept
            // with 2^31 added (top bit flipped), which turns signed order into unsigned order, so that
            // |d| = |D(i) - D(i-1)| is one unsigned sum-of-absolute-differences instruction that also adds
            // to the running sum.
            constexpr uint32_t Z = 0x80000000u;
            uint32_t p0 = (uint32_t)w[3] ^ Z, p1, p2, p3;   // D0..D3 of the sample before the run
            {
                const uint32_t x0 = (uint32_t)w[0] ^ Z, x1 = (uint32_t)w[1] ^ Z, x2 = (uint32_t)w[2] ^ Z;
                const uint32_t a1 = (x1 - x0) ^ Z, a2 = (x2 - x1) ^ Z, a3 = (p0 - x2) ^ Z;   // first differences
                const uint32_t b2 = (a2 - a1) ^ Z, b3 = (a3 - a2) ^ Z;                       // second
                p1 = a3; p2 = b3; p3 = (b3 - b2) ^ Z;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint32_t e32[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int jj = 0; jj < kRun / 2; ++jj) {
                    const int j = half * (kRun / 2) + jj;
                    const uint32_t c0 = (uint32_t)w[4 + j] ^ Z;
                    e32[0] = sad_u32(c0, Z, e32[0]);
                    e32[1] = sad_u32(c0, p0, e32[1]);
                    const uint32_t c1 = (c0 - p0) ^ Z;
                    e32[2] = sad_u32(c1, p1, e32[2]);
                    const uint32_t c2 = (c1 - p1) ^ Z;
                    e32[3] = sad_u32(c2, p2, e32[3]);
                    const uint32_t c3 = (c2 - p2) ^ Z;
                    e32[4] = sad_u32(c3, p3, e32[4]);
                    p0 = c0; p1 = c1; p2 = c2; p3 = c3;
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) e64[k] += e32[k];
            }
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint32_t e32[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int jj = 0; jj < kRun / 2; ++jj) {
                    const int j = half * (kRun / 2) + jj;
                    const int32_t a = w[4 + j], b = w[3 + j], c = w[2 + j], d = w[1 + j], e = w[j];
                    const int32_t d1 = a - b, d2 = a - 2 * b + c, d3 = a - 3 * b + 3 * c - d, d4 = a - 4 * b + 6 * c - 4 * d + e;
                    if ((counted >> j) & 1) {
                        e32[0] += (uint32_t)(a < 0 ? -a : a); e32[1] += (uint32_t)(d1 < 0 ? -d1 : d1); e32[2] += (uint32_t)(d2 < 0 ? -d2 : d2);
                        e32[3] += (uint32_t)(d3 < 0 ? -d3 : d3); e32[4] += (uint32_t)(d4 < 0 ? -d4 : d4);
                    }
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) e64[k] += e32[k];
            }
        }
        // five LDS additions per lane (four lanes of a wave share a slot) instead of five 64-bit sums over the row in registers
#pragma unroll
        for (int k = 0; k < 5; ++k) atomicAdd((unsigned long long *)&S.e[k][t & 15], (unsigned long long)e64[k]);
    }
    // ---- windowed autocorrelation, lags 0..maxo
    uint32_t maxo = o.force_lpc_order ? o.force_lpc_order : o.max_lpc_order;
    if (maxo > (uint32_t)M) maxo = M;
    const bool try_lpc = maxo > 0 && bs > maxo;
    if (try_lpc) {
        // the samples in LDS are replaced by their windowed single-precision values
        double own[kRun];
        {
            float wv[kRun];
#pragma unroll
            for (int q = 0; q < kRun / 4; ++q) {
                const uint32_t i = i0 + 4 * q;
                const float4 f4 = *reinterpret_cast<const float4 *>(win + (i < bs ? i : 0));   // tables are padded to 4
                wv[4 * q] = f4.x; wv[4 * q + 1] = f4.y; wv[4 * q + 2] = f4.z; wv[4 * q + 3] = f4.w;
            }
            __syncthreads();   // every lane has read its integer samples
            float *xf = reinterpret_cast<float *>(xs);
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                const float x = i0 + j < bs ? (float)mine[j] * wv[j] : 0.0f;
                xf[lbase + j] = x;
                own[j] = (double)x;
            }
            __syncthreads();
        }
        const float *xf = reinterpret_cast<const float *>(xs);
        constexpr int LG = M <= 12 ? 13 : 17;
#pragma unroll
        for (int base = 0; base <= M; base += LG) {
            if (base > (int)maxo) break;
            double acc[LG];
#pragma unroll
            for (int l = 0; l < LG; ++l) acc[l] = 0.0;
            // window element m is sample i0 - (base + LG - 1) + m; it meets own[j] at lag base + LG - 1 + j - m
#pragma unroll
            for (int m = 0; m < LG - 1 + kRun; ++m) {
                const double d = (double)xf[lbase + lds_rel(m - (base + LG - 1))];   // zero padding below sample 0
#pragma unroll
                for (int j = 0; j < kRun; ++j) {
                    const int l = LG - 1 + j - m;   // lag within the group
                    if (l >= 0 && l < LG) acc[l] = __fma_rn(own[j], d, acc[l]);
                }
            }
            if constexpr (M <= 12) {
                // The sum of a lag over the 16 lanes of a row is a balanced tree, neighbours first.  The lanes do not run
                // it as a butterfly on their own registers (four exchanges and additions for each of the 13 sums, every
                // lane ending with the same value): the windowed samples are no longer needed, so the sums are laid out
                // in their place, and one lane per (lag, row, half of the row) adds eight of them in the tree's order;
                // the two halves of a row sit in neighbouring lanes and meet in one exchange.  Seven lags per pass fit.
                constexpr int PASS = 7;
                constexpr uint32_t TS = NT + NT / 8;   // sums of one lag: eight of a lane group, then a pad
                double *T = reinterpret_cast<double *>(xs);
                static_assert(PASS * TS * 2 <= enc_lds_words(NT), "a pass of sums fits the sample buffer");
#pragma unroll
                for (int l0 = 0; l0 < LG; l0 += PASS) {
                    const int np = LG - l0 < PASS ? LG - l0 : PASS;
                    __syncthreads();   // the windowed samples, or the sums of the pass before, have been read
#pragma unroll
                    for (int l = 0; l < PASS; ++l)
                        if (l < np) T[l * TS + t + (t >> 3)] = acc[l0 + l];
                    __syncthreads();
                    if (t < (uint32_t)(NT / 8) * (uint32_t)np) {
                        const uint32_t g = t % (NT / 8), l = t / (NT / 8);   // g: row * 2 + half
                        const double *q = T + l * TS + 9 * g;
                        double sum = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                        sum = sum + dpp_move_f64<FLACGPU_DPP_XOR1>(sum);
                        if (!(g & 1)) S.rsum[g >> 1][l0 + l] = sum;
                    }
                }
            } else {
#pragma unroll
                for (int l = 0; l < LG; ++l) {
                    const double rs = row_butterfly(acc[l]);
                    if ((lane & 15) == 0) S.rsum[t >> 4][l] = rs;
                }
            }
            __syncthreads();
            // the butterfly goes on over the four rows of a wave, then the waves are added in ascending order
            if (t < (uint32_t)LG && base + t <= maxo) {
                double total = 0.0;
                for (int w = 0; w < NT / 64; ++w) {
                    const double ws = (S.rsum[4 * w][t] + S.rsum[4 * w + 1][t]) + (S.rsum[4 * w + 2][t] + S.rsum[4 * w + 3][t]);
                    total = w ? total + ws : ws;
                }
                cs_ac[base + t] = total;
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
    }
    if (t < 16) {   // (the barriers above are behind every lane's additions)
        uint64_t e[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) e[k] = row_sum64(S.e[k][t]);
        int fo = 0;
        for (int k = 1; k <= 4; ++k) if (e[k] < e[fo]) fo = k;
        if (t != 0) return;
        if ((uint32_t)fo > bs) fo = (int)bs;
        cs->constant = 0; cs->wasted = (uint32_t)wasted; cs->fo = (uint32_t)fo; cs->lpc_ok = 0;
    }
}

// ---------------------------------------------------------------- pair pick ---
//
// Channel pair of every frame of a kPairEstimated stream, before any candidate is analysed, from short
// linear predictors: the windowed autocorrelation of left, right and side at lags 0..4 in single precision
// (mid's follows from them: side + 4 mid = 2 (left + right) for sums of products), the prediction error of
// orders 1..4 (Levinson-Durbin) and from it the estimated size of the subframe; the pair with the smallest
// sum wins.  The signals are w*l, w*r and w*l - w*r (the samples as they are, no wasted-bit shift).  Sums are
// formed per lane over its run of 16 in ascending order with fused multiply-adds, over the 16 lanes of a row
// by a butterfly, over the rows in ascending order.

template <int NT>
struct PickShared {
    float  rs[NT / 16][15];          // per row of 16 lanes: left, right, side x 5 lags
};

// Left and right samples of the lane's share of a stereo block (see load_block_share).
template <int NT>
__device__ inline void load_block_lr(const uint8_t *src, uint32_t bs, int B, int32_t (&l)[kRun], int32_t (&r)[kRun])
{
    const uint32_t t = threadIdx.x;
    if (B == 2) {
        uint32_t lr[kRun];
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = t + (uint32_t)j * NT;
            lr[j] = *reinterpret_cast<const uint32_t *>(src + (size_t)(i < bs ? i : bs - 1) * 4);
        }
#pragma unroll
        for (int j = 0; j < kRun; ++j) { l[j] = (int32_t)(lr[j] << 16) >> 16; r[j] = (int32_t)lr[j] >> 16; }
    } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = t + (uint32_t)j * NT, ii = i < bs ? i : bs - 1;
            l[j] = load_le(src + (size_t)(2 * ii) * B, B);
            r[j] = load_le(src + (size_t)(2 * ii + 1) * B, B);
        }
    }
}

__device__ inline float row_butterfly_f32(float v)
{
    v = v + __uint_as_float(dpp_move<FLACGPU_DPP_XOR1>(__float_as_uint(v)));
    v = v + __uint_as_float(dpp_move<FLACGPU_DPP_XOR2>(__float_as_uint(v)));
    v = v + __uint_as_float(dpp_move<FLACGPU_DPP_HALF_MIRROR>(__float_as_uint(v)));
    v = v + __uint_as_float(dpp_move<FLACGPU_DPP_MIRROR>(__float_as_uint(v)));
    return v;
}

// DIRECT: every estimated stream has 16-bit samples.  A lane then loads its run of 16 stereo samples and the four before it
// straight from the PCM (five 16-byte loads of whole (left, right) words, five of the window) instead of loading samples a
// workgroup apart and passing them through LDS to get runs: the same values in the same registers, no LDS for the samples
// and none requested at launch, so more workgroups fit a CU.
struct __attribute__((packed, aligned(4))) EncDwords4 { uint32_t a, b, c, d; };

template <int NT, bool DIRECT>
__global__ __launch_bounds__(NT) void k_enc_pick(const uint8_t *__restrict__ pcm, const EncStreamDev *__restrict__ streams,
                                                 uint32_t n_streams, const uint32_t *__restrict__ hints, EncOptsDev o,
                                                 const float *__restrict__ windows, float *__restrict__ sums,
                                                 uint32_t pair_base, uint32_t pair_end)
{
    extern __shared__ __align__(16) int32_t xs[];   // !DIRECT: two blocks of enc_lds_words(NT): w*l and w*r, see lds_pos
    float *xl = reinterpret_cast<float *>(xs), *xr = xl + enc_lds_words(NT);
    __shared__ PickShared<NT> S;
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint32_t pi = pair_base + xcd_unit(blockIdx.x, gridDim.x);
    if (pi >= pair_end) return;
    const uint32_t unit = 2 * pi;                   // unit ranges start even: (frame, first channel of the pair)
    const EncStreamDev &st = streams[find_enc_stream(streams, n_streams, hints, unit, 0)];
    if (st.pair != kPairEstimated) return;
    const uint32_t f = (unit - st.unit_begin) >> 1;
    if (f >= st.n_frames) return;
    const bool last = f + 1 == st.n_frames;
    const uint32_t bs = last ? st.tail_bs : o.blocksize;
    const int B = st.bytes_ps;
    const uint8_t *src = pcm + st.in_off + (uint64_t)f * o.blocksize * (uint32_t)(2 * B);
    const float *win = windows + (last ? st.win_off_tail : st.win_off);
    if constexpr (!DIRECT) {
        int32_t l[kRun], r[kRun];
        load_block_lr<NT>(src, bs, B, l, r);
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = t + (uint32_t)j * NT;
            const float w = win[i < bs ? i : 0];
            const uint32_t at = lds_share_pos<NT>(t, j);
            xl[at] = i < bs ? (float)l[j] * w : 0.0f;
            xr[at] = i < bs ? (float)r[j] * w : 0.0f;
        }
        if (t < lds_lane_base(0)) { xl[t] = 0.0f; xr[t] = 0.0f; }
        __syncthreads();
    }
    {
        const uint32_t lbase = lds_lane_base(t);
        float wl[4 + kRun], wr[4 + kRun];   // the windowed samples i0 - 4 .. i0 + 15 of the lane's run (zero outside the block)
        if constexpr (DIRECT) {
            const uint32_t i0 = t * kRun;
            if (i0 >= 4 && i0 + kRun <= bs) {
                const EncDwords4 *ps = reinterpret_cast<const EncDwords4 *>(src + (size_t)(i0 - 4) * 4);
                const float4 *pw = reinterpret_cast<const float4 *>(win + (i0 - 4));   // window tables start on 16-byte boundaries
#pragma unroll
                for (int q = 0; q < (4 + kRun) / 4; ++q) {
                    const EncDwords4 v = ps[q];
                    const float4 w = pw[q];
                    const uint32_t lr[4] = {v.a, v.b, v.c, v.d};
                    const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        wl[4 * q + e] = (float)((int32_t)(lr[e] << 16) >> 16) * ww[e];
                        wr[4 * q + e] = (float)((int32_t)lr[e] >> 16) * ww[e];
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4 + kRun; ++m) {
                    const int32_t i = (int32_t)i0 + m - 4;
                    const bool in = i >= 0 && (uint32_t)i < bs;
                    const uint32_t lr = *reinterpret_cast<const uint32_t *>(src + (size_t)(in ? i : 0) * 4);
                    const float w = win[in ? i : 0];
                    wl[m] = in ? (float)((int32_t)(lr << 16) >> 16) * w : 0.0f;
                    wr[m] = in ? (float)((int32_t)lr >> 16) * w : 0.0f;
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4 + kRun; ++m) { wl[m] = xl[lbase + lds_rel(m - 4)]; wr[m] = xr[lbase + lds_rel(m - 4)]; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x[4 + kRun], acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < 4 + kRun; ++m) x[m] = c == 0 ? wl[m] : c == 1 ? wr[m] : wl[m] - wr[m];
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
#pragma unroll
                for (int q = 0; q < 5; ++q) acc[q] = __fmaf_rn(x[4 + j], x[4 + j - q], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float rsum = row_butterfly_f32(acc[q]);
                if ((lane & 15) == 0) S.rs[t >> 4][c * 5 + q] = rsum;
            }
        }
    }
    __syncthreads();
    if (t < 15) {
        float total = S.rs[0][t];
        for (int row = 1; row < NT / 16; ++row) total = total + S.rs[row][t];
        sums[(uint64_t)(st.frame_begin + f) * 16 + t] = total;   // k_enc_pick_mode takes it from here
    }
}

// Channel pair of a frame from the 15 sums k_enc_pick left for it (left, right, side x lags 0..4), one LANE per frame:
// four short Levinson-Durbin recursions and a comparison, serial arithmetic that kept one wave of every k_enc_pick
// workgroup busy while the other three waited.
__global__ __launch_bounds__(64) void k_enc_pick_mode(const EncStreamDev *__restrict__ streams, uint32_t n_streams,
                                                      const uint32_t *__restrict__ hints, EncOptsDev o,
                                                      const float *__restrict__ sums, uint32_t n_pairs)
{
    const uint32_t pi = blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= n_pairs) return;
    const uint32_t unit = 2 * pi;
    const EncStreamDev &st = streams[find_enc_stream(streams, n_streams, hints, unit, 0)];
    if (st.pair != kPairEstimated) return;
    const uint32_t f = (unit - st.unit_begin) >> 1;
    if (f >= st.n_frames) return;
    const uint32_t bs = f + 1 == st.n_frames ? st.tail_bs : o.blocksize;
    float sac[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) sac[q] = sums[(uint64_t)(st.frame_begin + f) * 16 + q];
    double ests[4];
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t) {
        // estimated size of the subframe of signal t: the cheapest of orders 1..4 by the rule of k_enc_design
        const int cbps = (int)st.bps + (t == 3 ? 1 : 0);
        int prec0 = o.qlp_precision ? (int)o.qlp_precision : default_precision((int)st.bps, bs);
        prec0 = prec0 > 15 ? 15 : prec0 < 5 ? 5 : prec0;
        double ac[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const double al = (double)sac[q], ar = (double)sac[5 + q], as = (double)sac[10 + q];
            ac[q] = t == 0 ? al : t == 1 ? ar : t == 3 ? as : (2.0 * (al + ar) - as) * 0.25;
        }
        double est = 0.0;
        if (ac[0] > 0) {
            const double scale = 0x1.0a6a4c7c29f9ap+0 / (double)bs;   // 0.5 / ln(2)^2 / bs
            double err = ac[0], tmp[4] = {0.0, 0.0, 0.0, 0.0};
            bool alive = true, have = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if ((uint32_t)(i + 1) < bs && alive) {
                    double r = -ac[i + 1];
#pragma unroll
                    for (int j = 0; j < i; ++j) r -= tmp[j] * ac[i - j];
                    if (err <= 0) alive = false;
                    else {
                        r /= err;
                        tmp[i] = r;
#pragma unroll
                        for (int j = 0; j < i / 2; ++j) {
                            const double tj = tmp[j];
                            tmp[j] += r * tmp[i - 1 - j];
                            tmp[i - 1 - j] += r * tj;
                        }
                        if (i & 1) tmp[i / 2] += tmp[i / 2] * r;
                        err *= (1.0 - r * r);
                        const double vv = err * scale;
                        double bpr = vv > 0 ? 0.5 * est_log2(vv) : 0.0;
                        if (bpr < 0) bpr = 0;
                        const double tot = bpr * (double)(bs - (uint32_t)(i + 1)) + (double)(i + 1) * (double)(cbps + prec0);
                        if (!have || tot < est) { est = tot; have = true; }
                    }
                }
            }
        }
        ests[t] = est;
    }
    const double b[4] = {ests[0] + ests[1], ests[0] + ests[3], ests[3] + ests[1], ests[2] + ests[3]};
    int mode = 0;
    for (int k = 1; k < 4; ++k) if (b[k] < b[mode]) mode = k;
    o.modes[st.frame_begin + f] = (uint8_t)mode;
}

// Levinson-Durbin, order choice and coefficient quantisation: one lane per candidate channel, every array
// in registers (all loops unrolled, indices known at compile time).
template <int M>
__global__ __launch_bounds__(64) void k_enc_design(const EncStreamDev *__restrict__ streams, uint32_t n_streams,
                                                   const uint32_t *__restrict__ hints, EncOptsDev o, uint8_t *__restrict__ stats)
{
    const uint32_t unit = blockIdx.x * 64 + threadIdx.x;
    if (unit >= o.total_units) return;
    CandStats *cs = reinterpret_cast<CandStats *>(stats + (uint64_t)unit * stats_stride(M));
    const UnitInfo u = unit_info(nullptr, streams, n_streams, hints, o, unit);
    if (!u.valid || cs->constant) return;
    const uint32_t bs = u.bs;
    uint32_t maxo = o.force_lpc_order ? o.force_lpc_order : o.max_lpc_order;
    if (maxo > (uint32_t)M) maxo = M;
    if (!(maxo > 0 && bs > maxo)) return;
    const double *cs_ac = reinterpret_cast<const double *>(stats + (uint64_t)unit * stats_stride(M) + kStatsHeadBytes);
    double ac[M + 1];
#pragma unroll
    for (int l = 0; l <= M; ++l) ac[l] = (uint32_t)l <= maxo ? cs_ac[l] : 0.0;
    if (!(ac[0] > 0)) return;
    const int sb = u.cbps - (int)cs->wasted, stream_bps = u.st->bps;
    int prec0 = o.qlp_precision ? (int)o.qlp_precision : default_precision(stream_bps, bs);
    prec0 = prec0 > 15 ? 15 : prec0 < 5 ? 5 : prec0;
    double err = ac[0], bestb = 0;
    const double scale = 0x1.0a6a4c7c29f9ap+0 / (double)bs;   // 0.5 / ln(2)^2 / bs
    double tmp[M], best[M];
#pragma unroll
    for (int j = 0; j < M; ++j) { tmp[j] = 0.0; best[j] = 0.0; }
    int ord = 0;
    bool alive = true;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        if ((uint32_t)i < maxo && alive) {
            double r = -ac[i + 1];
#pragma unroll
            for (int j = 0; j < i; ++j) r -= tmp[j] * ac[i - j];
            if (err <= 0) alive = false;
            else {
                r /= err;
                tmp[i] = r;
#pragma unroll
                for (int j = 0; j < i / 2; ++j) {
                    const double tj = tmp[j];
                    tmp[j] += r * tmp[i - 1 - j];
                    tmp[i - 1 - j] += r * tj;
                }
                if (i & 1) tmp[i / 2] += tmp[i / 2] * r;
                err *= (1.0 - r * r);
                const uint32_t k = (uint32_t)i + 1;
                bool take;
                if (o.force_lpc_order) take = (k == o.force_lpc_order);
                else {
                    const double vv = err * scale;
                    double bpr = vv > 0 ? 0.5 * est_log2(vv) : 0.0;
                    if (bpr < 0) bpr = 0;
                    const double tot = bpr * (double)(bs - k) + (double)k * (double)(sb + prec0);
                    take = (ord == 0 || tot < bestb);
                    if (take) bestb = tot;
                }
                if (take) {
                    ord = (int)k;
#pragma unroll
                    for (int j = 0; j <= i; ++j) best[j] = -tmp[j];
                }
            }
        }
    }
    if (!ord) return;
    int prec = prec0;
    if (sb <= 17 && !o.qlp_precision) {
        const int cap = 32 - sb - ilog2_32((uint32_t)ord);
        if (prec > cap) prec = cap;
    }
    double cmax = 0;
#pragma unroll
    for (int j = 0; j < M; ++j) { const double a = __builtin_fabs(best[j]); if (j < ord && a > cmax) cmax = a; }
    if (!(cmax > 0)) return;
    const uint64_t cb = (uint64_t)__double_as_longlong(cmax);
    const int ex = (int)((cb >> 52) & 0x7FF) - 1022;
    int shift = prec - ex - 1;
    if (shift > 15) shift = 15;
    const int qmax = (1 << (prec - 1)) - 1, qmin = -qmax - 1;
    const double mul = shift >= 0 ? (double)(1 << shift) : 1.0 / (double)(1 << -shift);
    if (shift < 0) shift = 0;
    double carry = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        int qv = 0;
        if (j < M && j < ord) {
            carry += best[j < M ? j : 0] * mul;
            qv = carry >= 0 ? (int)(carry + 0.5) : -(int)(0.5 - carry);
            qv = qv > qmax ? qmax : qv < qmin ? qmin : qv;
            carry -= (double)qv;
        }
        cs->q[j] = (int16_t)qv;
    }
    cs->ord = ord; cs->prec = prec; cs->shift = shift;
    cs->wide = (sb + prec + ilog2_32((uint32_t)ord)) > 32;
    cs->lpc_ok = 1;
}

template <int NT>
struct ChooseShared {
    uint64_t psum[2][NT];            // per lane: sum of folded residuals (0 fixed, 1 lpc)
    uint64_t heap[2][512];           // partition sums, node n has children 2n and 2n + 1
    uint32_t lvl_total[2][kEncMaxPo + 1];
    uint32_t need5[2];
    uint32_t exact[2];
    uint32_t lpc_bad;
    int32_t  q[32];
    int32_t  po[2], method[2];
    uint8_t  kb[2][512];
};

// Partition search of the residuals of both predictors at once (slot 0 fixed, slot 1 lpc when use1): fills
// S.po / S.method / S.kb / S.exact.  Called by the whole workgroup.  uf / ul: the lane's folded residuals
// (zero outside the coded range), mask_f / mask_l: which of the lane's samples are coded.
// Sums of the partitions of every order are kept as a heap (node n has children 2n and 2n + 1); a pass
// fills three levels from the level below them, so an order-6 search takes two passes.
template <int NT>
__device__ inline void rice_search_both(ChooseShared<NT> &S, const EncOptsDev &o, uint32_t bs, uint32_t order0, uint32_t order1,
                                        bool use1, const uint32_t (&uf)[kRun], const uint32_t (&ul)[kRun], uint32_t mask_f,
                                        uint32_t mask_l)
{
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint32_t lanes = (bs + kRun - 1) / kRun;
    const uint32_t nsearch = use1 ? 2 : 1;
    const uint32_t ord[2] = {order0, order1};
    int L[2], lo[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        L[w] = max_partition_order(bs, ord[w], o.max_po);
        lo[w] = (int)o.min_po > L[w] ? L[w] : (int)o.min_po;
    }
    const int Lmax = (use1 && L[1] > L[0]) ? L[1] : L[0];
    // leaves: the finest partitions, whole runs of lanes
    for (uint32_t w = 0; w < nsearch; ++w) {
        if (L[w] == 0) {
            const uint64_t ws = wave_sum64(t < lanes ? S.psum[w][t] : 0);
            if (lane == 0 && ws) atomicAdd((unsigned long long *)&S.heap[w][1], (unsigned long long)ws);
        }
    }
    for (uint32_t idx = t; idx < (nsearch << Lmax); idx += NT) {
        const uint32_t w = idx >> Lmax, p = idx & ((1u << Lmax) - 1);
        if (L[w] > 0 && p < (1u << L[w])) {
            const uint32_t tpp = (bs >> L[w]) / kRun;
            uint64_t sum = 0;
            for (uint32_t q = 0; q < tpp; ++q) sum += S.psum[w][p * tpp + q];
            S.heap[w][(1u << L[w]) + p] = sum;
        }
    }
    __syncthreads();
    // coarser levels, three per pass: nodes [2^(l-3), 2^l) from their descendants at level l, where l starts
    // at each search's own finest level
    for (int step = 0; 3 * step < Lmax; ++step) {
        for (uint32_t idx = t; idx < (nsearch << Lmax); idx += NT) {
            const uint32_t w = idx >> Lmax, n = idx & ((1u << Lmax) - 1);
            const int l = L[w] - 3 * step;
            if (l > 0 && n < (1u << l) && n >= (1u << (l >= 3 ? l - 3 : 0))) {
                const int k = ilog2_32(n);
                if (k >= lo[w]) {
                    const uint32_t span = 1u << (l - k), first = ((n - (1u << k)) << (l - k)) + (1u << l);
                    uint64_t sum = 0;
                    for (uint32_t q = 0; q < span; ++q) sum += S.heap[w][first + q];
                    S.heap[w][n] = sum;
                }
            }
        }
        __syncthreads();
    }
    // every partition of every order: parameter and estimated size
    for (uint32_t idx = t; idx < (nsearch << (Lmax + 1)); idx += NT) {
        const uint32_t w = idx >> (Lmax + 1), n = idx & ((2u << Lmax) - 1);
        if (n >= (1u << lo[w]) && n < (2u << L[w])) {
            const int l = ilog2_32(n);
            const uint32_t p = n - (1u << l);
            const uint32_t cnt = (bs >> l) - (p == 0 ? ord[w] : 0);
            uint32_t est;
            const uint32_t k = rice_param(S.heap[w][n], cnt, est);
            S.kb[w][n] = (uint8_t)k;
            atomicAdd(&S.lvl_total[w][l], est);
            if (k > 14) atomicOr(&S.need5[w], 1u << l);
        }
    }
    __syncthreads();
    if ((t == 0 || t == 64) && (t >> 6) < nsearch) {
        const uint32_t w = t >> 6;
        uint32_t best = 0xFFFFFFFFu;
        int bl = L[w], bm = 0;
        for (int l = L[w]; l >= lo[w]; --l) {
            const int n5 = (S.need5[w] >> l) & 1;
            const uint32_t total = S.lvl_total[w][l] + ((n5 ? 5u : 4u) << l);
            if (total < best) { best = total; bl = l; bm = n5; }
        }
        S.po[w] = bl; S.method[w] = bm;
    }
    __syncthreads();
    // exact size of the chosen partitioning
    uint32_t exact0 = 0, exact1 = 0;
    if (t < lanes) {
        const int po0 = S.po[0], po1 = S.po[1];
        const uint32_t part0 = po0 ? (t * kRun) / (bs >> po0) : 0, part1 = po1 ? (t * kRun) / (bs >> po1) : 0;
        const uint32_t k0 = S.kb[0][(1u << po0) + part0], k1 = use1 ? S.kb[1][(1u << po1) + part1] : 0;
        // residuals outside the coded range are zero
#pragma unroll
        for (int j = 0; j < kRun; ++j) { exact0 += uf[j] >> k0; exact1 += ul[j] >> k1; }
        exact0 += (uint32_t)__popc(mask_f) * (k0 + 1);
        exact1 = use1 ? exact1 + (uint32_t)__popc(mask_l) * (k1 + 1) : 0;
    }
    exact0 = row_sum32(exact0);
    exact1 = row_sum32(exact1);
    if ((lane & 15) == 0) {
        if (exact0) atomicAdd(&S.exact[0], exact0);
        if (exact1) atomicAdd(&S.exact[1], exact1);
    }
}

template <int M, int NT>
__global__ __launch_bounds__(NT) void k_enc_choose(const uint8_t *__restrict__ pcm, const EncStreamDev *__restrict__ streams,
                                                   uint32_t n_streams, const uint32_t *__restrict__ hints, EncOptsDev o,
                                                   const uint8_t *__restrict__ stats, uint8_t *__restrict__ plans,
                                                   uint32_t unit_base, uint32_t unit_end)
{
    extern __shared__ __align__(16) int32_t xs[];   // enc_lds_words(NT): see lds_pos
    __shared__ ChooseShared<NT> S;
    const uint32_t t = threadIdx.x;
    const uint32_t unit = unit_base + xcd_unit(blockIdx.x, gridDim.x);
    if (unit >= unit_end) return;
    const CandStats *cs = reinterpret_cast<const CandStats *>(stats + (uint64_t)unit * stats_stride(M));
    const UnitInfo u = unit_info(pcm, streams, n_streams, hints, o, unit);
    if (!u.valid || cs->constant) return;
    const uint32_t bs = u.bs;
    uint8_t *plan = plans + (uint64_t)unit * o.plan_stride;
    const int wasted = (int)cs->wasted, sb = u.cbps - wasted;
    const uint32_t hdr = 8 + (uint32_t)wasted;
    const int fo = (int)cs->fo, lpc_ok = (int)cs->lpc_ok, ord = lpc_ok ? cs->ord : 0;
    const int prec = cs->prec, shift = cs->shift, wide = cs->wide;

    if (t == 0) {
        S.lpc_bad = 0;
        for (int w = 0; w < 2; ++w) {
            S.need5[w] = 0; S.exact[w] = 0; S.heap[w][1] = 0; S.po[w] = 0; S.method[w] = 0;
            for (int l = 0; l <= kEncMaxPo; ++l) S.lvl_total[w][l] = 0;
        }
    }
    if (t < 32) S.q[t] = lpc_ok ? (int32_t)cs->q[t] : 0;
    {
        int32_t v[kRun];
        load_block_share<NT>(u.src, bs, u.C, u.B, (int)u.cand, u.stereo, v);
#pragma unroll
        for (int j = 0; j < kRun; ++j) xs[lds_share_pos<NT>(t, j)] = v[j] >> wasted;
    }
    if (t < lds_lane_base(0)) xs[t] = 0;
    __syncthreads();

    // ---- residuals of both predictors
    const uint32_t i0 = t * kRun, lbase = lds_lane_base(t);
    const uint32_t mask_f = run_mask(i0, (uint32_t)fo, bs), mask_l = lpc_ok ? run_mask(i0, (uint32_t)ord, bs) : 0;
    uint32_t uf[kRun], ul[kRun];
    if (lpc_ok) {
        uint32_t high;
        if (wide) high = lpc_residual<M, 2>(xs, lbase, S.q, shift, ul);
        else if (sb <= 16) high = lpc_residual<M, 3>(xs, lbase, S.q, shift, ul);
        else if (sb <= 24) high = lpc_residual<M, 0>(xs, lbase, S.q, shift, ul);
        else high = lpc_residual<M, 1>(xs, lbase, S.q, shift, ul);
        // only residuals that will be coded count
        if (high && (lpc_residual_bad<M>(xs, lbase, S.q, shift, wide != 0) & mask_l)) atomicOr(&S.lpc_bad, 1u);
    } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) ul[j] = 0;
    }
    fixed_residual(xs, lbase, fo, uf);
    // residuals outside the coded range (warm-up, beyond the block) are cleared: the first and last waves only
    if (!__all(mask_f == 0xFFFFu && (!lpc_ok || mask_l == 0xFFFFu))) {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            if (!((mask_f >> j) & 1)) uf[j] = 0;
            if (!((mask_l >> j) & 1)) ul[j] = 0;
        }
    }
    {
        uint64_t sf = 0, sl = 0;
#pragma unroll
        for (int j = 0; j < kRun; ++j) { sf += uf[j]; sl += ul[j]; }
        S.psum[0][t] = sf;
        S.psum[1][t] = sl;
    }
    __syncthreads();
    const bool use_lpc = lpc_ok && !S.lpc_bad;
    rice_search_both<NT>(S, o, bs, (uint32_t)fo, (uint32_t)ord, use_lpc, uf, ul, mask_f, mask_l);
    __syncthreads();

    // ---- choice and plan
    const uint32_t pbits0 = S.method[0] ? 5 : 4, pbits1 = S.method[1] ? 5 : 4;
    const uint32_t fixed_bits = hdr + (uint32_t)fo * (uint32_t)sb + 6 + (pbits0 << S.po[0]) + S.exact[0];
    uint32_t lpc_bits = 0xFFFFFFFFu;
    if (use_lpc)
        lpc_bits = hdr + (uint32_t)ord * (uint32_t)sb + 9 + (uint32_t)ord * (uint32_t)prec + 6 + (pbits1 << S.po[1]) + S.exact[1];
    const uint32_t verb_bits = hdr + bs * (uint32_t)sb;
    int type = 2, which = 0;
    uint32_t bits = fixed_bits;
    if (use_lpc && (lpc_bits < fixed_bits || o.force_lpc_order)) { type = 3; which = 1; bits = lpc_bits; }
    if (verb_bits < bits) { type = 1; bits = verb_bits; }   // also under a forced order: the size bound rests on it
    if (t == 0) {
        PlanHead h = {};
        h.bits = bits; h.type = (uint8_t)type; h.wasted = (uint8_t)wasted; h.sb = (uint8_t)sb;
        h.order = (uint8_t)(type == 3 ? ord : type == 2 ? fo : 0);
        h.prec = (uint8_t)(type == 3 ? prec : 0); h.shift = (uint8_t)(type == 3 ? shift : 0);
        h.po = (uint8_t)S.po[which]; h.method = (uint8_t)S.method[which];
        *reinterpret_cast<PlanHead *>(plan) = h;
    }
    if (type == 3 && t < 32) reinterpret_cast<int16_t *>(plan + kPlanHeadBytes)[t] = (int16_t)S.q[t];
    if (type >= 2) {
        const uint32_t np = 1u << S.po[which];
        for (uint32_t p = t; p < np; p += NT) plan[kPlanHeadBytes + kPlanCoefBytes + p] = S.kb[which][np + p];
    }
}

// ------------------------------------------------------------------- layout ---

__device__ inline uint32_t crc8_byte(uint32_t c, uint32_t byte)
{
    // (c ^ byte) * x^8 modulo x^8 + x^2 + x + 1, by the relation x^8 = x^2 + x + 1 applied twice
    const uint32_t t = c ^ byte;
    const uint32_t h = t ^ (t << 1) ^ (t << 2);
    const uint32_t hi = h >> 8;
    return (h & 0xFFu) ^ hi ^ (hi << 1) ^ (hi << 2);
}

// Header of a fixed-blocksize frame.  Returns its length in bytes (CRC-8 included).
__device__ inline uint32_t build_frame_header(uint8_t *hp, uint32_t bs, uint32_t sample_rate, uint32_t bps, uint32_t ch_assign,
                                              uint64_t number)
{
    uint32_t n = 0;
    hp[n++] = 0xFF;
    hp[n++] = 0xF8;
    uint32_t bs_code, bs_extra = 0;
    if (bs == 192) bs_code = 1;
    else if (bs == 576 || bs == 1152 || bs == 2304 || bs == 4608) bs_code = 2 + (uint32_t)ilog2_32(bs / 576);
    else if (bs >= 256 && bs <= 32768 && (bs & (bs - 1)) == 0) bs_code = 8 + (uint32_t)ilog2_32(bs / 256);
    else if (bs <= 256) { bs_code = 6; bs_extra = 1; }
    else { bs_code = 7; bs_extra = 2; }
    const uint32_t rates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    uint32_t sr_code = 0;
    for (uint32_t i = 1; i < 12; ++i) if (rates[i] == sample_rate) sr_code = i;
    hp[n++] = (uint8_t)((bs_code << 4) | sr_code);
    const uint32_t ss = bps == 8 ? 1 : bps == 12 ? 2 : bps == 16 ? 4 : bps == 20 ? 5 : bps == 24 ? 6 : 0;
    hp[n++] = (uint8_t)((ch_assign << 4) | (ss << 1));
    if (number < 0x80) hp[n++] = (uint8_t)number;
    else {
        const uint32_t len = number < 0x800 ? 2 : number < 0x10000 ? 3 : number < 0x200000 ? 4 : number < 0x4000000 ? 5
                             : number < 0x80000000ull ? 6 : 7;
        const uint8_t lead[8] = {0, 0, 0xC0, 0xE0, 0xF0, 0xF8, 0xFC, 0xFE};
        uint64_t v = number;
        for (uint32_t i = len - 1; i > 0; --i) { hp[n + i] = (uint8_t)(0x80 | (v & 0x3F)); v >>= 6; }
        hp[n] = (uint8_t)(lead[len] | v);
        n += len;
    }
    if (bs_extra == 1) hp[n++] = (uint8_t)(bs - 1);
    if (bs_extra == 2) { hp[n++] = (uint8_t)((bs - 1) >> 8); hp[n++] = (uint8_t)(bs - 1); }
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; ++i) c = crc8_byte(c, hp[i]);
    hp[n++] = (uint8_t)c;
    return n;
}

// Zeroes the part of 32-bit word `word` of `out` that lies inside [lo, hi).
__device__ inline void zero_word_clipped(uint8_t *out, uint64_t word, uint64_t lo, uint64_t hi)
{
    const uint64_t a = word * 4;
    if (a >= lo && a + 4 <= hi) { *reinterpret_cast<uint32_t *>(out + a) = 0; return; }
    for (uint32_t b = 0; b < 4; ++b)
        if (a + b >= lo && a + b < hi) out[a + b] = 0;
}

// One workgroup per stream: frame decisions, sizes, offsets, headers, STREAMINFO.
__global__ __launch_bounds__(256) void k_enc_layout(EncStreamDev *__restrict__ streams, EncOptsDev o, const uint8_t *__restrict__ plans,
                                                    FrameRec *__restrict__ frames, uint64_t *__restrict__ crc_off,
                                                    uint32_t *__restrict__ crc_len, uint8_t *__restrict__ out, int md5_later)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_min, s_max;
    __shared__ uint64_t s_carry;
    EncStreamDev &st = streams[blockIdx.x];
    const uint32_t t = threadIdx.x, lane = t & 63;
    const uint32_t nf = st.n_frames, ncand = st.ncand, C = st.channels;
    if (t == 0) { s_min = 0xFFFFFFFFu; s_max = 0; s_carry = 0; }
    __syncthreads();
    uint32_t mn = 0xFFFFFFFFu, mx = 0;
    for (uint32_t f0 = 0; f0 < nf; f0 += 256) {
        const uint32_t f = f0 + t;
        uint32_t bytes = 0;
        FrameRec fr = {};
        if (f < nf) {
            const uint8_t *pl = plans + (uint64_t)(st.unit_begin + f * ncand) * o.plan_stride;
            uint32_t pb[8];
            for (uint32_t c = 0; c < ncand; ++c) pb[c] = reinterpret_cast<const PlanHead *>(pl + (uint64_t)c * o.plan_stride)->bits;
            fr.stream = blockIdx.x;
            fr.bs = (f + 1 == nf) ? st.tail_bs : o.blocksize;
            if (C == 2) {
                int mode = st.pair == kPairNone ? o.stereo_mode : st.pair == kPairEstimated ? (int)o.modes[st.frame_begin + f] : st.pair;
                if (mode < 0) {
                    const uint32_t b0 = pb[0] + pb[1], b1 = pb[0] + pb[3], b2 = pb[3] + pb[1], b3 = pb[2] + pb[3];
                    mode = 0;
                    uint32_t bb = b0;
                    if (b1 < bb) { bb = b1; mode = 1; }
                    if (b2 < bb) { bb = b2; mode = 2; }
                    if (b3 < bb) { bb = b3; mode = 3; }
                }
                fr.cand[0] = (uint8_t)pair_cand(mode, 0);
                fr.cand[1] = (uint8_t)pair_cand(mode, 1);
                fr.ch_assign = mode == 0 ? 1 : (uint8_t)(7 + mode);
            } else {
                for (uint32_t c = 0; c < C; ++c) fr.cand[c] = (uint8_t)c;
                fr.ch_assign = (uint8_t)(C - 1);
            }
            uint8_t hb[16];
            fr.hdr_len = (uint8_t)build_frame_header(hb, fr.bs, st.sample_rate, st.bps, fr.ch_assign, f);
            uint32_t bit = 8u * fr.hdr_len;
            for (uint32_t c = 0; c < C; ++c) { fr.sf_bit[c] = bit; bit += pb[st.pair == kPairNone ? fr.cand[c] : c]; }
            bytes = (bit + 7) / 8 + 2;
            fr.bytes = bytes;
            mn = bytes < mn ? bytes : mn;
            mx = bytes > mx ? bytes : mx;
        }
        uint32_t total;
        const uint32_t excl = block_excl_scan<256>(bytes, s_w, total);
        if (f < nf) {
            fr.out_off = st.out_off + st.meta_len + s_carry + excl;
            frames[st.frame_begin + f] = fr;
        }
        __syncthreads();
        if (t == 0) s_carry += total;
        __syncthreads();
    }
    mn = wave_min32(mn); mx = wave_max32(mx);
    if (lane == 0) { atomicMin(&s_min, mn); atomicMax(&s_max, mx); }
    __syncthreads();
    const uint64_t out_len = st.meta_len + s_carry;
    const bool fits = out_len <= st.out_cap;
    if (t == 0) {
        st.out_len = out_len;
        st.min_fs = nf ? s_min : 0;
        st.max_fs = s_max;
        st.status = fits ? 0 : 8;   // FLACGPU_ERR_CAPACITY
        if (fits) {
            // marker, block header (flacmeta.go:519-551) and STREAMINFO (flacmeta.go:644-717)
            uint8_t *p = out + st.out_off;
            p[0] = 'f'; p[1] = 'L'; p[2] = 'a'; p[3] = 'C';
            p[4] = o.seek_every ? 0x00 : 0x80; p[5] = 0; p[6] = 0; p[7] = 34;
            if (o.seek_every) {
                // SEEKTABLE block header (type 3, last), flacmeta.go:519-551; its points are written below
                const uint32_t len = st.meta_len - 46;
                p[42] = 0x83; p[43] = (uint8_t)(len >> 16); p[44] = (uint8_t)(len >> 8); p[45] = (uint8_t)len;
            }
            uint8_t *s = p + 8;
            const uint32_t Bk = o.blocksize, mnf = st.min_fs, mxf = st.max_fs;
            s[0] = (uint8_t)(Bk >> 8); s[1] = (uint8_t)Bk; s[2] = (uint8_t)(Bk >> 8); s[3] = (uint8_t)Bk;
            s[4] = (uint8_t)(mnf >> 16); s[5] = (uint8_t)(mnf >> 8); s[6] = (uint8_t)mnf;
            s[7] = (uint8_t)(mxf >> 16); s[8] = (uint8_t)(mxf >> 8); s[9] = (uint8_t)mxf;
            const uint64_t v = ((uint64_t)st.sample_rate << 44) | ((uint64_t)(C - 1) << 41) | ((uint64_t)(st.bps - 1) << 36) |
                               (st.total_samples & 0xFFFFFFFFFull);
            for (int k = 0; k < 8; ++k) s[10 + k] = (uint8_t)(v >> (56 - 8 * k));
            if (!md5_later) for (int k = 0; k < 16; ++k) s[18 + k] = st.md5[k];
        }
    }
    // second pass: clear the words subframes share with their neighbours, write the frame headers
    for (uint32_t f = t; f < nf; f += 256) {
        FrameRec &fr = frames[st.frame_begin + f];
        const uint32_t g = st.frame_begin + f;
        if (!fits) { fr.bytes = 0; crc_off[g] = 0; crc_len[g] = 0; continue; }
        const uint64_t lo = fr.out_off, hi = lo + fr.bytes;
        const uint8_t *pl = plans + (uint64_t)(st.unit_begin + f * ncand) * o.plan_stride;
        for (uint32_t c = 0; c < C; ++c) {
            const uint32_t bits = reinterpret_cast<const PlanHead *>(pl + (uint64_t)(st.pair == kPairNone ? fr.cand[c] : c) * o.plan_stride)->bits;
            const uint64_t gs = lo * 8 + fr.sf_bit[c], ge = gs + bits - 1;
            zero_word_clipped(out, gs >> 5, lo, hi);
            zero_word_clipped(out, ge >> 5, lo, hi);
        }
        uint8_t hb[16];
        const uint32_t hl = build_frame_header(hb, fr.bs, st.sample_rate, st.bps, fr.ch_assign, f);
        for (uint32_t i = 0; i < hl; ++i) out[lo + i] = hb[i];
        crc_off[g] = lo;
        crc_len[g] = fr.bytes - 2;
        if (o.seek_every && f % o.seek_every == 0) {
            // SeekpointBlock, flacmeta.go:256-260 / :612-641: sample number, byte offset from frame 0, frame samples
            uint8_t *sp = out + st.out_off + 46 + (uint64_t)(f / o.seek_every) * 18;
            const uint64_t sample = (uint64_t)f * o.blocksize, off = lo - (st.out_off + st.meta_len);
            for (int k = 0; k < 8; ++k) { sp[k] = (uint8_t)(sample >> (56 - 8 * k)); sp[8 + k] = (uint8_t)(off >> (56 - 8 * k)); }
            sp[16] = (uint8_t)(fr.bs >> 8); sp[17] = (uint8_t)fr.bs;
        }
    }
}

// --------------------------------------------------------------------- pack ---

// ORs `width` (<= 32) bits of value, most significant first, at bit position pos of an LDS bit buffer.
__device__ inline void put_bits(uint32_t *buf, uint32_t pos, uint32_t value, uint32_t width)
{
    if (width == 0) return;
    const uint32_t w = pos >> 5, sh = pos & 31;
    const uint64_t v = (uint64_t)value << (64 - width - sh);
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    if (hi) atomicOr(&buf[w], hi);
    if (lo) atomicOr(&buf[w + 1], lo);
}

template <int M, int NT>
__global__ __launch_bounds__(NT) void k_enc_pack(const uint8_t *__restrict__ pcm, const EncStreamDev *__restrict__ streams,
                                                 uint32_t n_streams, const uint32_t *__restrict__ hints, EncOptsDev o,
                                                 const uint8_t *__restrict__ plans,
                                                 const FrameRec *__restrict__ frames, uint8_t *__restrict__ out, uint32_t buf_words,
                                                 uint32_t unit_base, uint32_t unit_end)
{
    extern __shared__ __align__(16) int32_t lds[];   // xs: enc_lds_words(NT) (see lds_pos), then the bit buffer
    int32_t *xs = lds;
    uint32_t *buf = reinterpret_cast<uint32_t *>(lds + enc_lds_words(NT));
    __shared__ uint32_t s_w[NT / 64];
    __shared__ int32_t s_q[32];
    __shared__ uint8_t s_k[256];
    const uint32_t t = threadIdx.x;
    const uint32_t unit = unit_base + xcd_unit(blockIdx.x, gridDim.x);
    if (unit >= unit_end) return;
    const int si = find_enc_stream(streams, n_streams, hints, unit, 1);
    const EncStreamDev &st = streams[si];
    const uint32_t C = st.channels, rel = unit - st.pack_begin;
    const uint32_t f = rel / C, ch = rel - f * C;
    const FrameRec &fr = frames[st.frame_begin + f];
    if (fr.bytes == 0) return;
    const uint32_t cand = fr.cand[ch], bs = fr.bs, ncand = st.ncand;
    const uint8_t *plan = plans + (uint64_t)(st.unit_begin + f * ncand + (st.pair == kPairNone ? cand : ch)) * o.plan_stride;
    const PlanHead ph = *reinterpret_cast<const PlanHead *>(plan);
    const int B = st.bytes_ps;
    const bool stereo4 = C == 2;
    const uint8_t *src = pcm + st.in_off + (uint64_t)f * o.blocksize * (uint32_t)(C * B);
    const uint64_t G = fr.out_off * 8 + fr.sf_bit[ch];
    const uint32_t phase = (uint32_t)(G & 31);
    const uint64_t word0 = G >> 5;
    const uint32_t nwords = (phase + ph.bits + 31) >> 5;
    const int wasted = ph.wasted, sb = ph.sb, order = ph.order, type = ph.type;
    const uint32_t smask = sb >= 32 ? 0xFFFFFFFFu : (1u << sb) - 1;

    for (uint32_t i = t; i < nwords + 1 && i < buf_words; i += NT) buf[i] = 0;
    {
        int32_t v[kRun];
        load_block_share<NT>(src, bs, (int)C, B, (int)cand, stereo4, v);
#pragma unroll
        for (int j = 0; j < kRun; ++j) xs[lds_share_pos<NT>(t, j)] = v[j] >> wasted;
    }
    if (t < lds_lane_base(0)) xs[t] = 0;
    if (t < 32) s_q[t] = type == 3 ? (int32_t)reinterpret_cast<const int16_t *>(plan + kPlanHeadBytes)[t] : 0;
    if (type >= 2) for (uint32_t p = t; p < (1u << ph.po); p += NT) s_k[p] = plan[kPlanHeadBytes + kPlanCoefBytes + p];
    __syncthreads();

    uint32_t pos = phase;
    if (t == 0) {
        const uint32_t code = type == 0 ? 0 : type == 1 ? 1 : type == 2 ? 8 + (uint32_t)order : 31 + (uint32_t)order;
        put_bits(buf, pos, (code << 1) | (wasted ? 1u : 0u), 8);
        if (wasted) put_bits(buf, pos + 8 + (uint32_t)wasted - 1, 1, 1);
    }
    pos += 8 + (uint32_t)wasted;
    const uint32_t i0 = t * kRun, lbase = lds_lane_base(t);
    if (type == 0) {
        if (t == 0) put_bits(buf, pos, (uint32_t)xs[lds_lane_base(0)] & smask, (uint32_t)sb);
    } else if (type == 1) {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = i0 + j;
            if (i < bs) put_bits(buf, pos + i * (uint32_t)sb, (uint32_t)xs[lbase + j] & smask, (uint32_t)sb);
        }
    } else {
        if (t < (uint32_t)order) put_bits(buf, pos + t * (uint32_t)sb, (uint32_t)xs[lds_pos(kFront + t)] & smask, (uint32_t)sb);
        pos += (uint32_t)order * (uint32_t)sb;
        if (type == 3) {
            const uint32_t prec = ph.prec;
            if (t == 0) { put_bits(buf, pos, prec - 1, 4); put_bits(buf, pos + 4, ph.shift, 5); }
            if (t < (uint32_t)order) put_bits(buf, pos + 9 + t * prec, (uint32_t)s_q[t] & ((1u << prec) - 1), prec);
            pos += 9 + (uint32_t)order * prec;
        }
        const uint32_t pbits = ph.method ? 5 : 4;
        if (t == 0) { put_bits(buf, pos, ph.method, 2); put_bits(buf, pos + 2, ph.po, 4); put_bits(buf, pos + 6, s_k[0], pbits); }
        pos += 6 + pbits;
        // residual codes
        uint32_t u[kRun];
        if (type == 3) {
            const bool wide = (sb + (int)ph.prec + ilog2_32((uint32_t)order)) > 32;
            if (wide) (void)lpc_residual<M, 2>(xs, lbase, s_q, ph.shift, u);
            else if (sb <= 16) (void)lpc_residual<M, 3>(xs, lbase, s_q, ph.shift, u);
            else if (sb <= 24) (void)lpc_residual<M, 0>(xs, lbase, s_q, ph.shift, u);
            else (void)lpc_residual<M, 1>(xs, lbase, s_q, ph.shift, u);
        } else {
            fixed_residual(xs, lbase, order, u);
        }
        const uint32_t po = ph.po, psize = bs >> po;
        const uint32_t part = po ? i0 / psize : 0;
        const bool starts = po && i0 > 0 && i0 < bs && (i0 - part * psize) == 0;
        const uint32_t k = s_k[part < (1u << po) ? part : 0];
        uint32_t mybits = starts ? pbits : 0;
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = i0 + j;
            if (i >= (uint32_t)order && i < bs) mybits += (u[j] >> k) + k + 1;
        }
        uint32_t total;
        uint32_t at = pos + block_excl_scan<NT>(mybits, s_w, total);
        if (starts) { put_bits(buf, at, k, pbits); at += pbits; }
        const uint32_t kmask = (1u << k) - 1;
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
            const uint32_t i = i0 + j;
            if (i >= (uint32_t)order && i < bs) {
                const uint32_t q = u[j] >> k;
                put_bits(buf, at + q, (1u << k) | (u[j] & kmask), k + 1);
                at += q + k + 1;
            }
        }
    }
    __syncthreads();
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    for (uint32_t w = t; w < nwords; w += NT) {
        const uint32_t val = __builtin_bswap32(buf[w]);
        if (w == 0 || w + 1 == nwords) { if (val) atomicOr(&out32[word0 + w], val); }
        else out32[word0 + w] = val;
    }
}

__global__ void k_enc_write_crc(const FrameRec *__restrict__ frames, const uint16_t *__restrict__ crc, uint32_t n, uint8_t *__restrict__ out)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const FrameRec &fr = frames[g];
    if (!fr.bytes) return;
    const uint32_t c = crc[g];
    out[fr.out_off + fr.bytes - 2] = (uint8_t)(c >> 8);
    out[fr.out_off + fr.bytes - 1] = (uint8_t)c;
}

// ---------------------------------------------------------------------- MD5 ---

__device__ inline uint32_t rol32(uint32_t x, uint32_t s) { return (x << s) | (x >> (32 - s)); }

__device__ inline void md5_block(uint32_t (&h)[4], const uint32_t (&w)[16], const uint32_t *T)
{
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int r = i >> 4;
        uint32_t fn;
        int g;
        if (r == 0) { fn = d ^ (b & (c ^ d)); g = i; }
        else if (r == 1) { fn = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; }
        else if (r == 2) { fn = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { fn = c ^ (b | ~d); g = (7 * i) & 15; }
        const int rot[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
        const uint32_t tt = a + fn + T[i] + w[g];
        a = d; d = c; c = b; b += rol32(tt, (uint32_t)rot[r][i & 3]);
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

// MD5 (RFC 1321) of len bytes at p, by one lane.  T: the 64 sine-derived constants, in LDS.
__device__ inline void md5_bytes(const uint8_t *p, uint64_t len, const uint32_t *T, uint8_t *digest)
{
    uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    const bool aligned = (((uintptr_t)p) & 15) == 0;
    uint64_t done = 0;
    uint32_t w[16];
    for (; done + 64 <= len; done += 64) {
        if (aligned) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 v = *reinterpret_cast<const uint4 *>(p + done + 16 * q);
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint8_t *b = p + done + 4 * q;
                w[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
            }
        }
        md5_block(h, w, T);
    }
    // tail: remaining bytes, 0x80, zeros, 64-bit length in bits
    const uint32_t rem = (uint32_t)(len - done);
    const uint64_t bits = len * 8;
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t idx = (uint32_t)blk * 64 + (uint32_t)q * 4 + (uint32_t)b;
                uint32_t byte = 0;
                if (idx < rem) byte = p[done + idx];
                else if (idx == rem) byte = 0x80;
                word |= byte << (8 * b);
            }
            w[q] = word;
        }
        const bool lastblk = (blk == 0 && rem < 56) || blk == 1;
        if (lastblk) { w[14] = (uint32_t)bits; w[15] = (uint32_t)(bits >> 32); }
        md5_block(h, w, T);
        if (lastblk) break;
    }
    for (int q = 0; q < 4; ++q)
        for (int b = 0; b < 4; ++b) digest[4 * q + b] = (uint8_t)(h[q] >> (8 * b));
}

// MD5 of each stream's packed PCM (the STREAMINFO signature, flacmeta.go:714), one lane per stream.
__global__ __launch_bounds__(64) void k_enc_md5(const uint8_t *__restrict__ pcm, EncStreamDev *__restrict__ streams, uint32_t n,
                                                const uint32_t *__restrict__ T, uint8_t *__restrict__ out)
{
    __shared__ uint32_t s_T[64];
    s_T[threadIdx.x] = T[threadIdx.x];
    __syncthreads();
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    EncStreamDev &st = streams[s];
    md5_bytes(pcm + st.in_off, st.total_samples * st.channels * st.bytes_ps, s_T, st.md5);
    if (out && st.out_cap >= 42)
        for (int q = 0; q < 16; ++q) out[st.out_off + 26 + q] = st.md5[q];
}

// MD5 of n byte ranges [off[i], off[i] + len[i]) of base, one lane per range; digests to out[16 * i].
__global__ __launch_bounds__(64) void k_md5_ranges(const uint8_t *__restrict__ base, const uint64_t *__restrict__ off,
                                                   const uint64_t *__restrict__ len, uint32_t n, const uint32_t *__restrict__ T,
                                                   uint8_t *__restrict__ out)
{
    __shared__ uint32_t s_T[64];
    s_T[threadIdx.x] = T[threadIdx.x];
    __syncthreads();
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n) return;
    md5_bytes(base + off[s], len[s], s_T, out + 16ull * s);
}

}  // namespace flacgpu
