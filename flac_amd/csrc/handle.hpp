// handle.hpp — the library handle and the small host helpers both translation units
// (flacgpu.hip: decode, flacgpu_encode.hip: encode) share.
#pragma once

#include "../../include/flacgpu.h"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "device_types.hpp"

namespace flacgpu {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct HostPipe;   // host_stage.hpp: pinned staging buffers and copy streams of the host-buffer entry points

enum { EV_START, EV_SCAN, EV_COMPACT, EV_CRC, EV_RESOLVE, EV_SKIM, EV_DECODE, EV_RESULTS, EV_COUNT };
enum { EE_START, EE_PICK, EE_STATS, EE_DESIGN, EE_CHOOSE, EE_LAYOUT, EE_PACK, EE_CRC, EE_END, EE_MD5_START, EE_MD5_END, EE_COUNT };

}  // namespace flacgpu

struct flacgpu_handle {
    int device = 0;
    uint32_t flags = 0;
    hipStream_t own_stream = nullptr;
    std::string err;
    // decode
    flacgpu::DevBuf streams, results, mask, bitmap, tile_count, tile_base, group_sum, group_base, counter, crc_tables, segrec;
    flacgpu::DevBuf goff, num, info, stream, len, status, segcrc, seglen, subpos, next;
    flacgpu::DevBuf shape_segs;   // FrameMap tables of the shape groups: per group its starts, then its bases
    std::vector<uint32_t> h_shape_segs;
    uint32_t cand_cap = 0;
    uint32_t output_format = 0;   // FLACGPU_OUT_*
    uint32_t launches = 0;        // kernels launched by the current call
    uint32_t shape_groups = 0;
    std::vector<uint8_t> h_md5;   // digests of the MD5 check
    hipEvent_t ev[flacgpu::EV_COUNT] = {};
    flacgpu_timing timing = {};
    std::vector<flacgpu::StreamDev> h_streams;
    std::vector<flacgpu::ResultDev> h_results;
    flacgpu::HostPipe *pipe = nullptr;   // created by the first host-buffer call
    bool fixed = false;    // buffers are carved from a static workspace: never freed, never grown
    uint32_t n_cand = 0;   // state of the last scan, used by flacgpu_scan_frames
    uint32_t n_reread = 0; // spans of the last scan whose CRC-16 took a second read of their bytes
    // encode
    flacgpu::DevBuf e_streams, e_plans, e_frames, e_crc_off, e_crc_len, e_crc, e_windows, e_md5_T, e_stats, e_hints, e_modes;
    hipStream_t md5_stream = nullptr;
    hipEvent_t eev[flacgpu::EE_COUNT] = {};
    flacgpu_encode_timing enc_timing = {};
    bool enc_ready = false;
};

namespace flacgpu {

inline int fail(flacgpu_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, e_ == hipErrorOutOfMemory ? FLACGPU_ERR_NOMEM : FLACGPU_ERR_DEVICE, \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

inline int reserve(flacgpu_handle *h, DevBuf &b, size_t bytes)
{
    if (b.bytes >= bytes) return FLACGPU_OK;
    if (h->fixed) return fail(h, FLACGPU_ERR_CAPACITY, "the batch needs more workspace than the entry point that allocates nothing has");
    if (b.p) { HIP_TRY(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return FLACGPU_OK;
}

inline void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.bytes = 0;
}

// CRC-16 (polynomial 0x8005) of n byte spans [off[i], off[i] + len[i]) of `base`, by the span kernel of the
// decoder (defined in flacgpu.hip).  base must be readable in whole 16-byte vectors around every span.
__attribute__((visibility("hidden"))) void launch_crc16_spans(flacgpu_handle *h, const uint8_t *base, const uint64_t *off,
                                                              const uint32_t *len, uint16_t *crc, uint32_t n, hipStream_t st);

// The handle's staging pipeline, created on first use / freed (defined in flacgpu.hip).
__attribute__((visibility("hidden"))) HostPipe *host_pipe(flacgpu_handle *h);
__attribute__((visibility("hidden"))) void host_pipe_release(flacgpu_handle *h);

// Frees the encoder's part of a handle (defined in flacgpu_encode.hip).
__attribute__((visibility("hidden"))) void encode_release(flacgpu_handle *h);

}  // namespace flacgpu
