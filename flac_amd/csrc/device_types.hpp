// device_types.hpp — structures shared by the host driver and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flacgpu {

// Per-stream record in device memory (built on the host from flacgpu_stream).
struct StreamDev {
    uint64_t in_off;         // bytes from d_in to frame 0
    uint64_t in_len;
    uint64_t out_off;        // bytes from d_out
    uint64_t out_cap;
    uint64_t total_samples;  // 0 = unknown
    uint64_t sample_base;    // inter-channel index of the sample that lands at out_off (frame ranges of a long stream)
    uint64_t plane_stride;   // planar output: bytes from one channel's plane to the next; 0 = interleaved
    uint32_t max_bs;         // STREAMINFO MaxBlockSize
    uint32_t cand_begin;     // [cand_begin, cand_end) in the candidate arrays
    uint32_t cand_end;
    uint32_t pending;        // some frame of this stream spans more than one candidate
    uint8_t  channels;
    uint8_t  bps;
    uint8_t  bytes_ps;       // bytes per sample in the output: ceil(bps / 8), or 4 with 32-bit output
    uint8_t  pad;
    uint32_t pad2;
};
static_assert(sizeof(StreamDev) % 8 == 0, "StreamDev is copied as an array");

// Per-stream reduction written by k_results.
struct ResultDev {
    uint64_t header_candidates;
    uint64_t frames;
    uint64_t frames_ok;
    uint64_t samples;
    uint64_t pcm_bytes;
    uint64_t first_bad_frame;   // index among the stream's frames (skipped candidates not counted); ~0 = none
    int32_t  first_error;
    uint32_t pad;
};

// Which candidates a skim or decode launch works on: item i of the launch is candidate base[j] + (i - start[j]) for the
// segment j with start[j] <= i < start[j + 1].  Streams of one shape (channel count, output sample size) need
// not be neighbours in the batch: their candidate ranges are the segments.  One segment needs no table.
struct FrameMap {
    const uint32_t *start;   // [nseg + 1] item index at which each segment starts; null = one segment
    const uint32_t *base;    // [nseg] first candidate of each segment
    uint32_t nseg;
    uint32_t first;          // one segment: its first candidate
    uint32_t count;          // items in all
    __device__ inline uint32_t candidate(uint32_t i) const
    {
        if (!start) return first + i;
        uint32_t lo = 0, hi = nseg;   // the last segment with start <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (start[mid] <= i) lo = mid; else hi = mid;
        }
        return base[lo] + (i - start[lo]);
    }
};

// Candidate info word: blocksize:16 | header_len:5 | ch_assign:4 | (bps-1):5 | variable:1
__host__ __device__ inline uint32_t pack_info(uint32_t bs, uint32_t hl, uint32_t ca, uint32_t bps, uint32_t var)
{
    return (bs & 0xFFFF) | (hl << 16) | (ca << 21) | ((bps - 1) << 25) | (var << 30);
}
__host__ __device__ inline uint32_t info_bs(uint32_t i) { return i & 0xFFFF; }
__host__ __device__ inline uint32_t info_hl(uint32_t i) { return (i >> 16) & 31; }
__host__ __device__ inline uint32_t info_ca(uint32_t i) { return (i >> 21) & 15; }
__host__ __device__ inline uint32_t info_bps(uint32_t i) { return ((i >> 25) & 31) + 1; }
__host__ __device__ inline uint32_t info_var(uint32_t i) { return (i >> 30) & 1; }

// Candidate (= potential frame) arrays, structure of arrays, in stream order.
struct CandArrays {
    uint64_t *goff;     // byte offset from d_in
    uint64_t *num;      // coded frame / sample number
    uint32_t *info;     // pack_info
    uint32_t *stream;   // owning stream
    uint32_t *len;      // span length in bytes once resolved
    int32_t  *status;   // FLACGPU_FRAME_*
    uint16_t *segcrc;   // CRC-16 of [goff[r], goff[r+1]) (or to the end of the stream)
    uint32_t *seglen;   // its length
    uint32_t *subpos;   // [r * 8 + c]: bit offset of subframe c from the frame's aligned word base
    uint32_t *next;     // candidate that follows the frame starting here
};

constexpr int kFrameOK = 0, kFrameBadCrc16 = 1, kFrameBadSubframe = 2, kFrameTruncated = 3,
              kFrameLength = 4, kFrameRange = 5, kFrameUnsupported = 6, kFrameSkipped = 7;

constexpr int kLookahead = 4;          // candidates tried as the end of a frame
constexpr int kScanTileBytes = 65536;  // bytes of input per scan workgroup
constexpr int kGranule = 16;           // bytes per mask word (uint16)

}  // namespace flacgpu
