// host_stage.hpp — the PCIe side of the host-buffer entry points (flacgpu_decode_batch, flacgpu_encode_batch).
//
// The caller's buffers are ordinary pageable memory, one per stream.  Copying them one hipMemcpy at a time runs at
// a fraction of the bus rate (the runtime stages every copy through its own small pinned buffer, serially).  Here
// the batch is cut into chunks of streams; a chunk's inputs are gathered by a few host threads into a pinned
// buffer the handle owns and go to the device in ONE copy, its outputs come back in one copy into a second pinned
// buffer and are scattered to the caller's buffers by the same threads.  Two pinned buffers each way and three HIP
// streams (in, compute, out) let chunk i + 1 travel to the device while chunk i is computed and chunk i - 1 travels
// back.  Device arenas and pinned buffers are grow-only: a steady-state call allocates nothing.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <future>
#include <thread>
#include <vector>

#include "handle.hpp"

namespace flacgpu {

struct HostPipe {
    void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
    size_t cap_in = 0, cap_out = 0;
    DevBuf d_in[2], d_out[2];        // device arenas of one chunk, two of each: chunk i + 1 arrives while chunk i is computed
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_h2d[2] = {}, ev_done[2] = {}, ev_d2h[2] = {};
    bool ready = false;
};

struct HostItem {
    const void *src; uint64_t src_len;    // the caller's input
    void *dst; uint64_t dst_cap;          // the caller's output buffer
    uint64_t in_off, out_off;             // place inside the chunk's arenas (filled here)
};

constexpr uint64_t kStageAlign = 256;
constexpr uint64_t kStageChunkBytes = 96ull << 20;   // input + output bytes of a chunk, about

inline void parallel_for_bytes(size_t n, const std::function<uint64_t(size_t)> &bytes_of, const std::function<void(size_t)> &body)
{
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned T = std::max(1u, std::min(8u, hw ? hw / 2 : 4u));
    uint64_t total = 0;
    for (size_t i = 0; i < n; ++i) total += bytes_of(i);
    if (T == 1 || n < 2 || total < (4u << 20)) { for (size_t i = 0; i < n; ++i) body(i); return; }
    // contiguous ranges of items with about equal bytes
    std::vector<size_t> cut(T + 1, n);
    cut[0] = 0;
    uint64_t acc = 0;
    unsigned t = 1;
    for (size_t i = 0; i < n && t < T; ++i) {
        acc += bytes_of(i);
        if (acc >= total * t / T) cut[t++] = i + 1;
    }
    std::vector<std::thread> th;
    for (unsigned k = 1; k < T; ++k)
        if (cut[k] < cut[k + 1]) th.emplace_back([&, k] { for (size_t i = cut[k]; i < cut[k + 1]; ++i) body(i); });
    for (size_t i = cut[0]; i < cut[1]; ++i) body(i);
    for (auto &x : th) x.join();
}

inline int pipe_prepare(flacgpu_handle *h, HostPipe &p, size_t need_in, size_t need_out)
{
    if (!p.ready) {
        HIP_TRY(hipStreamCreateWithFlags(&p.s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p.s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&p.ev_h2d[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p.ev_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p.ev_d2h[i], hipEventDisableTiming));
        }
        p.ready = true;
    }
    if (need_in > p.cap_in) {
        for (int i = 0; i < 2; ++i) { if (p.pin_in[i]) HIP_TRY(hipHostFree(p.pin_in[i])); p.pin_in[i] = nullptr; }
        p.cap_in = 0;
        const size_t want = need_in + need_in / 4;
        for (int i = 0; i < 2; ++i) HIP_TRY(hipHostMalloc(&p.pin_in[i], want, hipHostMallocDefault));
        p.cap_in = want;
    }
    if (need_out > p.cap_out) {
        for (int i = 0; i < 2; ++i) { if (p.pin_out[i]) HIP_TRY(hipHostFree(p.pin_out[i])); p.pin_out[i] = nullptr; }
        p.cap_out = 0;
        const size_t want = need_out + need_out / 4;
        for (int i = 0; i < 2; ++i) HIP_TRY(hipHostMalloc(&p.pin_out[i], want, hipHostMallocDefault));
        p.cap_out = want;
    }
    for (int i = 0; i < 2; ++i) {
        int rc;
        if ((rc = reserve(h, p.d_in[i], need_in + 256))) return rc;
        if ((rc = reserve(h, p.d_out[i], need_out + 256))) return rc;
    }
    return FLACGPU_OK;
}

inline void pipe_release(HostPipe &p)
{
    for (int i = 0; i < 2; ++i) {
        if (p.pin_in[i]) (void)hipHostFree(p.pin_in[i]);
        if (p.pin_out[i]) (void)hipHostFree(p.pin_out[i]);
        release(p.d_in[i]);
        release(p.d_out[i]);
        if (p.ev_h2d[i]) (void)hipEventDestroy(p.ev_h2d[i]);
        if (p.ev_done[i]) (void)hipEventDestroy(p.ev_done[i]);
        if (p.ev_d2h[i]) (void)hipEventDestroy(p.ev_d2h[i]);
    }
    if (p.s_in) (void)hipStreamDestroy(p.s_in);
    if (p.s_out) (void)hipStreamDestroy(p.s_out);
    p = HostPipe();
}

// Runs `items` through the pipeline.  compute(first, count, d_in, in_bytes, d_out, out_bytes, stream) works on the
// chunk items[first .. first + count) whose inputs lie at items[i].in_off from d_in, writes their outputs at
// items[i].out_off from d_out, returns a status, and is complete (stream idle) when it returns.
// out_len(i): bytes of item i's output to hand back once its chunk has been computed (at most dst_cap).
// zero_out: clear a chunk's output arena before computing (decode: frames that fail leave zeros).
typedef std::function<int(size_t, size_t, uint8_t *, uint64_t, uint8_t *, uint64_t, hipStream_t)> ChunkCompute;
inline int run_staged(flacgpu_handle *h, HostPipe &p, std::vector<HostItem> &items, hipStream_t main, bool zero_out,
                      const ChunkCompute &compute, const std::function<uint64_t(size_t)> &out_len)
{
    const size_t n = items.size();
    // chunks: runs of items of about kStageChunkBytes; offsets inside the chunk's arenas
    struct Chunk { size_t first, count; uint64_t in_bytes, out_bytes; };
    std::vector<Chunk> chunks;
    uint64_t max_in = 0, max_out = 0;
    for (size_t i = 0; i < n;) {
        Chunk c = {i, 0, 0, 0};
        while (i < n && (c.count == 0 || c.in_bytes + c.out_bytes + items[i].src_len + items[i].dst_cap <= kStageChunkBytes)) {
            items[i].in_off = c.in_bytes;
            items[i].out_off = c.out_bytes;
            c.in_bytes += (items[i].src_len + kStageAlign - 1) & ~(kStageAlign - 1);
            c.out_bytes += (items[i].dst_cap + kStageAlign - 1) & ~(kStageAlign - 1);
            ++c.count; ++i;
        }
        max_in = std::max(max_in, c.in_bytes);
        max_out = std::max(max_out, c.out_bytes);
        chunks.push_back(c);
    }
    int rc = pipe_prepare(h, p, (size_t)max_in + 64, (size_t)max_out + 64);
    if (rc) return rc;
    const int device = h->device;

    auto stage_in = [&](size_t ci) -> int {   // gather + one copy to the device; slot ci & 1
        (void)hipSetDevice(device);
        const Chunk &c = chunks[ci];
        const int s = (int)(ci & 1);
        if (ci >= 2 && hipEventSynchronize(p.ev_h2d[s]) != hipSuccess) return FLACGPU_ERR_DEVICE;   // the copy that used this pinned buffer
        uint8_t *pin = (uint8_t *)p.pin_in[s];
        parallel_for_bytes(c.count, [&](size_t k) { return items[c.first + k].src_len; },
                           [&](size_t k) { const HostItem &it = items[c.first + k]; if (it.src_len) std::memcpy(pin + it.in_off, it.src, it.src_len); });
        if (ci >= 2 && hipStreamWaitEvent(p.s_in, p.ev_done[s], 0) != hipSuccess) return FLACGPU_ERR_DEVICE;   // the compute that read this arena
        if (c.in_bytes && hipMemcpyAsync(p.d_in[s].p, pin, c.in_bytes, hipMemcpyHostToDevice, p.s_in) != hipSuccess) return FLACGPU_ERR_DEVICE;
        if (hipEventRecord(p.ev_h2d[s], p.s_in) != hipSuccess) return FLACGPU_ERR_DEVICE;
        return FLACGPU_OK;
    };
    auto stage_out = [&](size_t ci) -> int {   // wait for the chunk's copy back, scatter to the caller's buffers
        (void)hipSetDevice(device);
        const Chunk &c = chunks[ci];
        const int s = (int)(ci & 1);
        if (hipEventSynchronize(p.ev_d2h[s]) != hipSuccess) return FLACGPU_ERR_DEVICE;
        const uint8_t *pin = (const uint8_t *)p.pin_out[s];
        parallel_for_bytes(c.count, [&](size_t k) { return out_len(c.first + k); },
                           [&](size_t k) {
                               const HostItem &it = items[c.first + k];
                               const uint64_t len = std::min<uint64_t>(out_len(c.first + k), it.dst_cap);
                               if (len) std::memcpy(it.dst, pin + it.out_off, len);
                           });
        return FLACGPU_OK;
    };

    std::future<int> f_in = std::async(std::launch::async, stage_in, (size_t)0), f_out;
    int result = FLACGPU_OK;
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
        const Chunk &c = chunks[ci];
        const int s = (int)(ci & 1);
        int r = f_in.get();
        if (ci + 1 < chunks.size()) f_in = std::async(std::launch::async, stage_in, ci + 1);
        if (r == FLACGPU_OK && hipStreamWaitEvent(main, p.ev_h2d[s], 0) != hipSuccess) r = FLACGPU_ERR_DEVICE;
        if (r == FLACGPU_OK && ci >= 2 && hipStreamWaitEvent(main, p.ev_d2h[s], 0) != hipSuccess) r = FLACGPU_ERR_DEVICE;   // the copy back that read this arena
        if (r == FLACGPU_OK && zero_out && c.out_bytes && hipMemsetAsync(p.d_out[s].p, 0, c.out_bytes, main) != hipSuccess) r = FLACGPU_ERR_DEVICE;
        if (r == FLACGPU_OK)
            r = compute(c.first, c.count, (uint8_t *)p.d_in[s].p, c.in_bytes, (uint8_t *)p.d_out[s].p, c.out_bytes, main);
        if (r == FLACGPU_OK && hipEventRecord(p.ev_done[s], main) != hipSuccess) r = FLACGPU_ERR_DEVICE;
        // the pinned buffer of this slot is free once the scatter of chunk ci - 2 has finished (f_out of ci - 1 ran after it)
        if (f_out.valid()) { const int ro = f_out.get(); if (r == FLACGPU_OK) r = ro; }
        if (r == FLACGPU_OK) {
            // one copy back: what the items of the chunk hold, up to the end of the last one
            uint64_t end = 0;
            for (size_t k = 0; k < c.count; ++k) {
                const uint64_t len = std::min<uint64_t>(out_len(c.first + k), items[c.first + k].dst_cap);
                if (len) end = std::max(end, items[c.first + k].out_off + len);
            }
            if (hipStreamWaitEvent(p.s_out, p.ev_done[s], 0) != hipSuccess) r = FLACGPU_ERR_DEVICE;
            if (r == FLACGPU_OK && end && hipMemcpyAsync(p.pin_out[s], p.d_out[s].p, end, hipMemcpyDeviceToHost, p.s_out) != hipSuccess) r = FLACGPU_ERR_DEVICE;
            if (r == FLACGPU_OK && hipEventRecord(p.ev_d2h[s], p.s_out) != hipSuccess) r = FLACGPU_ERR_DEVICE;
            if (r == FLACGPU_OK) f_out = std::async(std::launch::async, stage_out, ci);
        }
        if (r != FLACGPU_OK) { result = r; break; }
    }
    if (f_in.valid()) (void)f_in.get();
    if (f_out.valid()) { const int ro = f_out.get(); if (result == FLACGPU_OK) result = ro; }
    (void)hipStreamSynchronize(p.s_in);
    (void)hipStreamSynchronize(p.s_out);
    if (result == FLACGPU_ERR_DEVICE && h->err.empty()) h->err = "HIP error in the host staging pipeline";
    return result;
}

}  // namespace flacgpu
