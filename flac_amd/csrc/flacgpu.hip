// flacgpu.hip — host driver of libflacgpu: workspace management, kernel
// sequencing and the C ABI declared in include/flacgpu.h.
//
// A decode call runs, on one HIP stream:
//   scan mask -> tile prefix -> compact -> stream ranges -> CRC-16 spans ->
//   resolve -> (per run of streams with equal channels / sample bytes)
//   skim -> decode -> per-stream results.
// Two small device-to-host reads happen in between: the candidate count (to
// size the candidate arrays) and the per-stream candidate ranges (to size the
// skim / decode grids).
#include "../../include/flacgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "frame_scan.hpp"
#include "subframe.hpp"

using namespace flacgpu;

#include "handle.hpp"
#include "host_stage.hpp"

#include <dlfcn.h>

#include <cstdlib>

namespace {

CandArrays cand_arrays(flacgpu_handle *h)
{
    CandArrays c;
    c.goff = (uint64_t *)h->goff.p; c.num = (uint64_t *)h->num.p; c.info = (uint32_t *)h->info.p;
    c.stream = (uint32_t *)h->stream.p; c.len = (uint32_t *)h->len.p; c.status = (int32_t *)h->status.p;
    c.segcrc = (uint16_t *)h->segcrc.p; c.seglen = (uint32_t *)h->seglen.p; c.subpos = (uint32_t *)h->subpos.p;
    c.next = (uint32_t *)h->next.p;
    return c;
}

int reserve_candidates(flacgpu_handle *h, uint32_t n)
{
    if (n <= h->cand_cap) return FLACGPU_OK;
    if (h->fixed) return fail(h, FLACGPU_ERR_CAPACITY, "more frame header candidates than FLACGPU_NOALLOC_MAX_CANDIDATES");
    const size_t cap = (size_t)n + n / 8 + 64;
    int rc;
    if ((rc = reserve(h, h->goff, cap * 8))) return rc;
    if ((rc = reserve(h, h->num, cap * 8))) return rc;
    if ((rc = reserve(h, h->info, cap * 4))) return rc;
    if ((rc = reserve(h, h->stream, cap * 4))) return rc;
    if ((rc = reserve(h, h->len, cap * 4))) return rc;
    if ((rc = reserve(h, h->status, cap * 4))) return rc;
    if ((rc = reserve(h, h->segcrc, cap * 2))) return rc;
    if ((rc = reserve(h, h->seglen, cap * 4))) return rc;
    if ((rc = reserve(h, h->subpos, cap * 8 * 4))) return rc;
    if ((rc = reserve(h, h->next, cap * 4))) return rc;
    h->cand_cap = (uint32_t)cap;
    return FLACGPU_OK;
}

// rocTX ranges around the host side of every stage (what rocprofv3 --marker-trace shows), only when the
// environment asks for them: FLACGPU_ROCTX=1 loads the rocTX library at the first call; without it, or
// without the library, nothing happens.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = std::getenv("FLACGPU_ROCTX");
        if (!e || !*e || *e == '0') return;
        // rocprofv3 listens to the rocprofiler-sdk flavour of the library, older tools to roctracer's
        void *lib = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
            if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};

const Roctx &roctx()
{
    static const Roctx r;
    return r;
}
thread_local bool g_range_open = false;   // a stage range of this thread's current call is pushed and not yet popped

// End of stage `which` on stream s: its timing event, and the range of the stage that follows.
void mark(flacgpu_handle *h, int which, hipStream_t s)
{
    if (h->flags & FLACGPU_FLAG_TIMING) (void)hipEventRecord(h->ev[which], s);
    const Roctx &rt = roctx();
    if (rt.push) {
        static const char *const next_stage[EV_COUNT] = {"flacgpu:scan", "flacgpu:compact", "flacgpu:crc16", "flacgpu:resolve",
                                                         "flacgpu:skim", "flacgpu:decode", "flacgpu:results", nullptr};
        if (g_range_open) { (void)rt.pop(); g_range_open = false; }
        if (next_stage[which]) { (void)rt.push(next_stage[which]); g_range_open = true; }
    }
}

// Closes the stage range an error return left open (every entry point that calls mark() holds one of these).
struct RangeGuard {
    ~RangeGuard()
    {
        if (g_range_open) { (void)roctx().pop(); g_range_open = false; }
    }
};

// Validates the batch description and builds the device stream table.
int build_streams(flacgpu_handle *h, uint64_t in_bytes, uint64_t out_bytes, const flacgpu_stream *streams, uint32_t n,
                  bool need_out)
{
    h->h_streams.resize(n);
    uint64_t prev_end = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const flacgpu_stream &s = streams[i];
        if (s.in_offset < prev_end) return fail(h, FLACGPU_ERR_ARG, "streams must be in ascending, non-overlapping in_offset order");
        if (s.in_offset > in_bytes || s.in_length > in_bytes - s.in_offset) return fail(h, FLACGPU_ERR_ARG, "stream exceeds the input buffer");
        if (s.in_length >= (1ull << 31)) return fail(h, FLACGPU_ERR_UNSUPPORTED, "stream of 2 GiB or more");
        if (need_out && (s.out_offset > out_bytes || s.out_capacity > out_bytes - s.out_offset))
            return fail(h, FLACGPU_ERR_ARG, "stream exceeds the output buffer");
        if (s.info.total_samples && s.sample_base > s.info.total_samples) return fail(h, FLACGPU_ERR_ARG, "sample_base beyond TotalSamples");
        if (s.info.channels < 1 || s.info.channels > 8 || s.info.bits_per_sample < 4 || s.info.bits_per_sample > 32 ||
            s.info.max_blocksize < 16)
            return fail(h, FLACGPU_ERR_STREAMINFO, "STREAMINFO out of bounds");
        prev_end = s.in_offset + s.in_length;
        StreamDev d;
        std::memset(&d, 0, sizeof d);
        d.in_off = s.in_offset; d.in_len = s.in_length; d.out_off = s.out_offset;
        d.out_cap = need_out ? s.out_capacity : ~0ull;
        d.total_samples = s.info.total_samples; d.max_bs = s.info.max_blocksize;
        d.sample_base = s.sample_base;
        d.channels = s.info.channels; d.bps = s.info.bits_per_sample;
        d.bytes_ps = h->output_format == FLACGPU_OUT_S32 ? 4 : (uint8_t)((s.info.bits_per_sample + 7) / 8);
        // planar: equal planes, each a whole number of dwords
        if (need_out && h->output_format == FLACGPU_OUT_PLANAR) {
            d.plane_stride = (s.out_capacity / s.info.channels) & ~(uint64_t)3;
            if (d.plane_stride == 0) { d.plane_stride = 4; d.out_cap = 0; }   // no room for a plane: every frame is out of range
        }
        h->h_streams[i] = d;
    }
    return FLACGPU_OK;
}

// Frame discovery: leaves the candidate arrays, statuses and per-stream ranges on the device
// and h->h_streams[].cand_begin/end on the host.
int run_scan(flacgpu_handle *h, const uint8_t *d_in, uint64_t in_bytes, uint32_t n, hipStream_t st)
{
    int rc;
    const uint64_t n_tiles64 = (in_bytes + kScanTileBytes - 1) / kScanTileBytes;
    if (n_tiles64 == 0 || n_tiles64 > 0x7FFFFFFFull) return fail(h, FLACGPU_ERR_ARG, "input size out of range");
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    const uint32_t n_groups = (n_tiles + kPrefixGroup - 1) / kPrefixGroup;
    if ((rc = reserve(h, h->streams, (size_t)n * sizeof(StreamDev)))) return rc;
    if ((rc = reserve(h, h->results, (size_t)n * sizeof(ResultDev)))) return rc;
    if ((rc = reserve(h, h->mask, (size_t)n_tiles * kTileGranules * 2))) return rc;
    if ((rc = reserve(h, h->bitmap, (size_t)n_tiles * kTileChunks * 8))) return rc;
    if ((rc = reserve(h, h->tile_count, (size_t)n_tiles * 4))) return rc;
    if ((rc = reserve(h, h->tile_base, (size_t)n_tiles * 4))) return rc;
    if ((rc = reserve(h, h->group_sum, (size_t)n_groups * 4))) return rc;
    if ((rc = reserve(h, h->group_base, (size_t)n_groups * 4))) return rc;
    if ((rc = reserve(h, h->counter, 64))) return rc;
    if ((rc = reserve(h, h->segrec, (size_t)n_tiles * kScanWaves * kSegSlots * 4))) return rc;

    HIP_TRY(hipMemcpyAsync(h->streams.p, h->h_streams.data(), (size_t)n * sizeof(StreamDev), hipMemcpyHostToDevice, st));
    mark(h, EV_START, st);
    // tile_base is free until the prefix kernels write it: it holds every tile's first stream for the scan
    k_tile_streams<<<(n_tiles + 255) / 256, 256, 0, st>>>((const StreamDev *)h->streams.p, n, n_tiles, (uint32_t *)h->tile_base.p);
    k_scan_tiles<<<n_tiles, kScanThreads, 0, st>>>(d_in, in_bytes, (const StreamDev *)h->streams.p, n, (uint16_t *)h->mask.p,
                                          (uint64_t *)h->bitmap.p, (uint32_t *)h->tile_count.p, (uint32_t *)h->segrec.p,
                                          (const uint32_t *)h->tile_base.p);
    mark(h, EV_SCAN, st);
    k_tile_prefix_local<<<n_groups, 1024, 0, st>>>((const uint32_t *)h->tile_count.p, (uint32_t *)h->tile_base.p, n_tiles,
                                                   (uint32_t *)h->group_sum.p);
    k_tile_prefix_groups<<<1, 1024, 0, st>>>((const uint32_t *)h->group_sum.p, (uint32_t *)h->group_base.p, n_groups,
                                             (uint32_t *)h->counter.p);
    uint32_t n_cand = 0;
    HIP_TRY(hipMemcpyAsync(&n_cand, h->counter.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->n_cand = n_cand;
    if ((rc = reserve_candidates(h, n_cand ? n_cand : 1))) return rc;
    CandArrays c = cand_arrays(h);
    if (n_cand) {
        k_compact<<<(n_tiles + 3) / 4, 256, 0, st>>>((const uint16_t *)h->mask.p, (const uint64_t *)h->bitmap.p,
                                                     (const uint32_t *)h->tile_count.p, (const uint32_t *)h->tile_base.p,
                                                     (const uint32_t *)h->group_base.p, n_tiles, h->cand_cap, c.goff);
        k_cand_fill<<<(n_cand + 255) / 256, 256, 0, st>>>(d_in, (const StreamDev *)h->streams.p, n, c, n_cand);
    }
    k_stream_ranges<<<(n + 255) / 256, 256, 0, st>>>((StreamDev *)h->streams.p, n, c.goff, n_cand);
    mark(h, EV_COMPACT, st);
    if (n_cand) {
        // span CRCs from the scan's cut records; the few spans the records cannot give (h->next as their list, which the
        // resolve step overwrites afterwards) are read again
        uint32_t *fb_count = (uint32_t *)h->counter.p + 1;
        HIP_TRY(hipMemsetAsync(fb_count, 0, 4, st));
        k_span_crc<<<(n_cand + 255) / 256, 256, 0, st>>>(c, n_cand, (const uint32_t *)h->segrec.p, (const uint16_t *)h->crc_tables.p,
                                                         c.next, fb_count);
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_cand + 3) / 4, 256 * 8);
        k_crc16_spans<<<blocks, 256, 0, st>>>(d_in, c, n_cand, (const uint16_t *)h->crc_tables.p, c.next, fb_count);
    }
    mark(h, EV_CRC, st);
    if (n_cand) {
        k_resolve_chain<<<(n_cand + 255) / 256, 256, 0, st>>>((StreamDev *)h->streams.p, c, n_cand,
                                                              (const uint16_t *)h->crc_tables.p + kCrcPow2);
        k_resolve_walk<<<(n + 3) / 4, 256, 0, st>>>((const StreamDev *)h->streams.p, n, c);
    }
    mark(h, EV_RESOLVE, st);
    h->launches = n_cand ? 11 : 5;   // tile streams, scan, two prefix kernels, stream ranges; with candidates: compact, fill, two CRC-16, two resolve
    HIP_TRY(hipMemcpyAsync(h->h_streams.data(), h->streams.p, (size_t)n * sizeof(StreamDev), hipMemcpyDeviceToHost, st));
    h->n_reread = 0;
    if (n_cand) HIP_TRY(hipMemcpyAsync(&h->n_reread, (uint32_t *)h->counter.p + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return FLACGPU_OK;
}

int fetch_results(flacgpu_handle *h, uint32_t n, flacgpu_stream_result *results, hipStream_t st)
{
    k_results<<<(n + 3) / 4, 256, 0, st>>>((const StreamDev *)h->streams.p, n, cand_arrays(h), (ResultDev *)h->results.p);
    mark(h, EV_RESULTS, st);
    std::vector<ResultDev> &tmp = h->h_results;
    tmp.resize(n);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->results.p, (size_t)n * sizeof(ResultDev), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (results)
        for (uint32_t i = 0; i < n; ++i) {
            results[i].header_candidates = tmp[i].header_candidates;
            results[i].frames = tmp[i].frames;
            results[i].frames_ok = tmp[i].frames_ok;
            results[i].samples = tmp[i].samples;
            results[i].pcm_bytes = tmp[i].pcm_bytes;
            results[i].first_bad_frame = tmp[i].first_bad_frame;
            results[i].first_error = tmp[i].first_error;
            results[i].md5_status = FLACGPU_MD5_NOT_CHECKED;
        }
    ++h->launches;
    return FLACGPU_OK;
}

// The frame table of the last scan or decode of this handle (candidate arrays still on the device), in stream order.
int frame_table(flacgpu_handle *h, flacgpu_frame *frames, uint64_t cap, uint64_t *n_frames)
{
    const uint32_t nc = h->n_cand;
    std::vector<uint64_t> goff(nc), num(nc);
    std::vector<uint32_t> info(nc), stream(nc), len(nc);
    std::vector<int32_t> status(nc);
    if (nc) {
        HIP_TRY(hipMemcpy(status.data(), h->status.p, (size_t)nc * 4, hipMemcpyDeviceToHost));
        if (frames) {
            HIP_TRY(hipMemcpy(goff.data(), h->goff.p, (size_t)nc * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(num.data(), h->num.p, (size_t)nc * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(info.data(), h->info.p, (size_t)nc * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(stream.data(), h->stream.p, (size_t)nc * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(len.data(), h->len.p, (size_t)nc * 4, hipMemcpyDeviceToHost));
        }
    }
    uint64_t nf = 0;
    int ret = FLACGPU_OK;
    for (uint32_t r = 0; r < nc; ++r) {
        if (status[r] == kFrameSkipped) continue;
        if (frames) {
            if (nf < cap) {
                const StreamDev &s = h->h_streams[stream[r]];
                flacgpu_frame f;
                std::memset(&f, 0, sizeof f);
                f.offset = goff[r] - s.in_off;
                f.number = num[r];
                f.variable = (uint8_t)info_var(info[r]);
                f.first_sample = f.variable ? num[r] : num[r] * (uint64_t)s.max_bs;
                f.length = len[r];
                f.blocksize = info_bs(info[r]);
                f.stream = stream[r];
                f.status = status[r];
                f.ch_assign = (uint8_t)info_ca(info[r]);
                f.bps = (uint8_t)info_bps(info[r]);
                f.header_len = (uint8_t)info_hl(info[r]);
                frames[nf] = f;
            } else {
                ret = FLACGPU_ERR_CAPACITY;
            }
        }
        ++nf;
    }
    *n_frames = nf;
    if (ret) return fail(h, ret, "frame array too small");
    return FLACGPU_OK;
}

void collect_timing(flacgpu_handle *h, uint64_t in_bytes, uint64_t out_bytes)
{
    flacgpu_timing t;
    std::memset(&t, 0, sizeof t);
    t.candidates = h->n_cand; t.in_bytes = in_bytes; t.out_bytes = out_bytes; t.crc16_reread_spans = h->n_reread;
    t.launches = h->launches; t.shape_groups = h->shape_groups;
    if (h->flags & FLACGPU_FLAG_TIMING) {
        auto el = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, h->ev[a], h->ev[b]); return ms; };
        t.scan_ms = el(EV_START, EV_SCAN);
        t.compact_ms = el(EV_SCAN, EV_COMPACT);
        t.crc16_ms = el(EV_COMPACT, EV_CRC);
        t.resolve_ms = el(EV_CRC, EV_RESOLVE);
        t.skim_ms = el(EV_RESOLVE, EV_SKIM);
        t.decode_ms = el(EV_SKIM, EV_DECODE);
        t.results_ms = el(EV_DECODE, EV_RESULTS);
        t.total_ms = el(EV_START, EV_RESULTS);
    }
    h->timing = t;
}

}  // namespace

namespace flacgpu {

HostPipe *host_pipe(flacgpu_handle *h)
{
    if (!h->pipe) h->pipe = new HostPipe();
    return h->pipe;
}

void host_pipe_release(flacgpu_handle *h)
{
    if (h->pipe) { pipe_release(*h->pipe); delete h->pipe; h->pipe = nullptr; }
}

void launch_crc16_spans(flacgpu_handle *h, const uint8_t *base, const uint64_t *off, const uint32_t *len, uint16_t *crc,
                        uint32_t n, hipStream_t st)
{
    if (!n) return;
    CandArrays c = {};
    c.goff = const_cast<uint64_t *>(off);
    c.seglen = const_cast<uint32_t *>(len);
    c.segcrc = crc;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 3) / 4, 256 * 8);
    k_crc16_spans<<<blocks, 256, 0, st>>>(base, c, n, (const uint16_t *)h->crc_tables.p);
}

}  // namespace flacgpu

extern "C" {

int flacgpu_create(const flacgpu_options *opts, flacgpu_handle **out)
{
    if (!out) return FLACGPU_ERR_ARG;
    *out = nullptr;
    flacgpu_handle *h = new flacgpu_handle();
    if (opts && opts->struct_size >= sizeof(flacgpu_options)) {
        h->device = opts->device; h->flags = opts->flags; h->output_format = opts->output_format;
    }
    if (h->output_format > FLACGPU_OUT_PLANAR) { delete h; return FLACGPU_ERR_ARG; }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || h->device < 0 || h->device >= count) { delete h; return FLACGPU_ERR_DEVICE; }
    if (hipSetDevice(h->device) != hipSuccess) { delete h; return FLACGPU_ERR_DEVICE; }
    // A blocking stream: work issued to it is ordered after what the caller has already put on the legacy
    // default stream (stream 0, which is also what torch.cuda.current_stream() is unless the caller changed
    // it), so a d_in / d_out that stream 0 is still filling is complete before the first kernel reads it.
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamDefault) != hipSuccess) { delete h; return FLACGPU_ERR_DEVICE; }
    for (int i = 0; i < EV_COUNT; ++i)
        if (hipEventCreate(&h->ev[i]) != hipSuccess) { delete h; return FLACGPU_ERR_DEVICE; }
    if (hipMalloc(&h->crc_tables.p, kCrcTableEntries * 2) != hipSuccess) { flacgpu_destroy(h); return FLACGPU_ERR_NOMEM; }
    h->crc_tables.bytes = kCrcTableEntries * 2;
    k_crc16_build_tables<<<1, 256, 0, h->own_stream>>>((uint16_t *)h->crc_tables.p);
    if (hipStreamSynchronize(h->own_stream) != hipSuccess) { flacgpu_destroy(h); return FLACGPU_ERR_DEVICE; }
    *out = h;
    return FLACGPU_OK;
}

void flacgpu_destroy(flacgpu_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    DevBuf *all[] = {&h->streams, &h->results, &h->mask, &h->bitmap, &h->tile_count, &h->tile_base, &h->group_sum, &h->group_base,
                     &h->counter, &h->crc_tables, &h->segrec, &h->goff, &h->num, &h->info, &h->stream, &h->len, &h->status, &h->segcrc,
                     &h->seglen, &h->subpos, &h->next, &h->shape_segs};
    for (DevBuf *b : all) release(*b);
    host_pipe_release(h);
    encode_release(h);
    for (int i = 0; i < EV_COUNT; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

const char *flacgpu_last_error(const flacgpu_handle *h) { return h ? h->err.c_str() : "null handle"; }

int flacgpu_get_timing(const flacgpu_handle *h, flacgpu_timing *out)
{
    if (!h || !out) return FLACGPU_ERR_ARG;
    *out = h->timing;
    return FLACGPU_OK;
}

int flacgpu_scan_frames(flacgpu_handle *h, const void *d_in, uint64_t in_bytes, const flacgpu_stream *streams,
                        uint32_t n_streams, flacgpu_frame *frames, uint64_t cap, uint64_t *n_frames,
                        flacgpu_stream_result *results, void *hip_stream)
{
    if (!h || !d_in || !streams || !n_streams || !n_frames) return fail(h, FLACGPU_ERR_ARG, "null argument");
    if ((uintptr_t)d_in & 15) return fail(h, FLACGPU_ERR_ARG, "d_in must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    RangeGuard guard;
    int rc = build_streams(h, in_bytes, 0, streams, n_streams, false);
    if (rc) return rc;
    if ((rc = run_scan(h, (const uint8_t *)d_in, in_bytes, n_streams, st))) return rc;
    mark(h, EV_SKIM, st);
    mark(h, EV_DECODE, st);
    if ((rc = fetch_results(h, n_streams, results, st))) return rc;
    collect_timing(h, in_bytes, 0);

    return frame_table(h, frames, cap, n_frames);
}

int flacgpu_get_frames(flacgpu_handle *h, flacgpu_frame *frames, uint64_t cap, uint64_t *n_frames)
{
    if (!h || !n_frames || (!frames && cap)) return fail(h, FLACGPU_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    if (h->h_streams.empty()) { *n_frames = 0; return FLACGPU_OK; }
    return frame_table(h, frames, cap, n_frames);
}

int flacgpu_api_version(void) { return FLACGPU_API_VERSION; }

// The body of both device-resident decode entry points; `st` is the stream everything runs on.
static int decode_device(flacgpu_handle *h, const void *d_in, uint64_t in_bytes, void *d_out, uint64_t out_bytes,
                         const flacgpu_stream *streams, uint32_t n_streams, flacgpu_stream_result *results, hipStream_t st)
{
    RangeGuard guard;
    int rc = build_streams(h, in_bytes, out_bytes, streams, n_streams, true);
    if (rc) return rc;
    if ((rc = run_scan(h, (const uint8_t *)d_in, in_bytes, n_streams, st))) return rc;

    // Streams of one shape (channel count, output bytes per sample) share one skim launch and one set of decode
    // launches, wherever they lie in the batch: a group's frames are the candidate ranges of its streams, joined
    // where they touch.  A group of one range needs no table; the others get theirs in h->shape_segs.
    struct Group { uint32_t C, B, count; std::vector<uint32_t> start, base; size_t table_off; };
    std::vector<Group> groups;
    for (uint32_t i = 0; i < n_streams; ++i) {
        const StreamDev &s = h->h_streams[i];
        if (s.cand_end == s.cand_begin) continue;
        Group *g = nullptr;
        for (Group &x : groups)
            if (x.C == s.channels && x.B == s.bytes_ps) { g = &x; break; }
        if (!g) { groups.push_back(Group{s.channels, s.bytes_ps, 0, {}, {}, 0}); g = &groups.back(); }
        const uint32_t n = s.cand_end - s.cand_begin;
        if (!g->base.empty() && g->base.back() + (g->count - g->start.back()) == s.cand_begin) {
            g->count += n;    // continues the group's last range
        } else {
            g->start.push_back(g->count);
            g->base.push_back(s.cand_begin);
            g->count += n;
        }
    }
    h->shape_groups = (uint32_t)groups.size();
    h->h_shape_segs.clear();
    for (Group &g : groups) {
        if (g.base.size() < 2) continue;
        g.table_off = h->h_shape_segs.size();
        h->h_shape_segs.insert(h->h_shape_segs.end(), g.start.begin(), g.start.end());
        h->h_shape_segs.push_back(g.count);
        h->h_shape_segs.insert(h->h_shape_segs.end(), g.base.begin(), g.base.end());
    }
    if (!h->h_shape_segs.empty()) {
        if ((rc = reserve(h, h->shape_segs, h->h_shape_segs.size() * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(h->shape_segs.p, h->h_shape_segs.data(), h->h_shape_segs.size() * 4, hipMemcpyHostToDevice, st));
    }
    auto frame_map = [&](const Group &g) {
        FrameMap m;
        m.start = nullptr; m.base = nullptr; m.nseg = 1; m.first = g.base[0]; m.count = g.count;
        if (g.base.size() > 1) {
            m.start = (const uint32_t *)h->shape_segs.p + g.table_off;
            m.base = m.start + g.base.size() + 1;
            m.nseg = (uint32_t)g.base.size();
        }
        return m;
    };
    CandArrays c = cand_arrays(h);
    const uint64_t in_vecs = (in_bytes + 15) / 16;
    if (h->flags & FLACGPU_FLAG_TIMING) HIP_TRY(hipMemsetAsync((uint32_t *)h->counter.p + 12, 0, 4, st));
    for (const Group &g : groups) {
        k_skim<<<(g.count + 255) / 256, 256, 0, st>>>((const uint8_t *)d_in, in_vecs, c, frame_map(g), g.C, (const uint4 *)h->crc_tables.p);
        ++h->launches;
    }
    mark(h, EV_SKIM, st);
    const bool planar = h->output_format == FLACGPU_OUT_PLANAR;
    for (const Group &r : groups) {
        const uint32_t F = 64 / r.C;
        const uint32_t waves = (r.count + F - 1) / F;
        const uint32_t blocks = (waves + 3) / 4;
        DecodeParams p;
        p.in = (const uint8_t *)d_in; p.in_vecs = in_vecs; p.out = (uint8_t *)d_out;
        p.spare = (const uint4 *)h->crc_tables.p;
        p.streams = (const StreamDev *)h->streams.p; p.c = c;
        p.map = frame_map(r); p.C = r.C; p.B = r.B;
        p.class_flags = (uint32_t *)h->counter.p + 4;
        p.check_flag = 0;
        HIP_TRY(hipMemsetAsync(p.class_flags, 0, 8 * sizeof(uint32_t), st));
        // every predictor-length class is launched over the same grid; a wave runs in exactly one
#define FLACGPU_LAUNCH(MO, W, S) \
        k_decode<MO, W, S><<<blocks, 256, 4 * decode_wave_lds(r.C), st>>>(p); p.check_flag = 1
        // the first launch classifies every wave, so it is the class most waves of such streams fall in
        if (planar && r.B <= 2) {
            FLACGPU_LAUNCH(12, 3, 4); FLACGPU_LAUNCH(32, 3, 4); FLACGPU_LAUNCH(12, 0, 4); FLACGPU_LAUNCH(32, 0, 4);
            FLACGPU_LAUNCH(12, 1, 4); FLACGPU_LAUNCH(32, 1, 4); FLACGPU_LAUNCH(12, 2, 4); FLACGPU_LAUNCH(32, 2, 4);
        } else if (planar) {
            FLACGPU_LAUNCH(12, 0, 4); FLACGPU_LAUNCH(32, 0, 4); FLACGPU_LAUNCH(12, 1, 4); FLACGPU_LAUNCH(32, 1, 4);
            FLACGPU_LAUNCH(12, 2, 4); FLACGPU_LAUNCH(32, 2, 4); FLACGPU_LAUNCH(12, 3, 4); FLACGPU_LAUNCH(32, 3, 4);
        } else if (r.C == 2 && r.B == 2) {
            FLACGPU_LAUNCH(12, 3, 1); FLACGPU_LAUNCH(32, 3, 1); FLACGPU_LAUNCH(12, 0, 1); FLACGPU_LAUNCH(32, 0, 1);
            FLACGPU_LAUNCH(12, 1, 1); FLACGPU_LAUNCH(32, 1, 1); FLACGPU_LAUNCH(12, 2, 1); FLACGPU_LAUNCH(32, 2, 1);
        } else if (r.C == 1 && r.B == 2) {
            FLACGPU_LAUNCH(12, 3, 2); FLACGPU_LAUNCH(32, 3, 2); FLACGPU_LAUNCH(12, 0, 0); FLACGPU_LAUNCH(32, 0, 0);
            FLACGPU_LAUNCH(12, 1, 0); FLACGPU_LAUNCH(32, 1, 0); FLACGPU_LAUNCH(12, 2, 0); FLACGPU_LAUNCH(32, 2, 0);
        } else if (r.C * r.B >= 8 && r.B <= 2) {   // rows of at least a line: the sixteen-slot write-out
            FLACGPU_LAUNCH(12, 3, 3); FLACGPU_LAUNCH(32, 3, 3); FLACGPU_LAUNCH(12, 0, 3); FLACGPU_LAUNCH(32, 0, 3);
            FLACGPU_LAUNCH(12, 1, 3); FLACGPU_LAUNCH(32, 1, 3); FLACGPU_LAUNCH(12, 2, 3); FLACGPU_LAUNCH(32, 2, 3);
        } else if (r.C * r.B >= 8) {
            FLACGPU_LAUNCH(12, 0, 3); FLACGPU_LAUNCH(32, 0, 3); FLACGPU_LAUNCH(12, 1, 3); FLACGPU_LAUNCH(32, 1, 3);
            FLACGPU_LAUNCH(12, 2, 3); FLACGPU_LAUNCH(32, 2, 3); FLACGPU_LAUNCH(12, 3, 3); FLACGPU_LAUNCH(32, 3, 3);
        } else if (r.B <= 2) {
            FLACGPU_LAUNCH(12, 3, 0); FLACGPU_LAUNCH(32, 3, 0); FLACGPU_LAUNCH(12, 0, 0); FLACGPU_LAUNCH(32, 0, 0);
            FLACGPU_LAUNCH(12, 1, 0); FLACGPU_LAUNCH(32, 1, 0); FLACGPU_LAUNCH(12, 2, 0); FLACGPU_LAUNCH(32, 2, 0);
        } else {
            FLACGPU_LAUNCH(12, 0, 0); FLACGPU_LAUNCH(32, 0, 0); FLACGPU_LAUNCH(12, 1, 0); FLACGPU_LAUNCH(32, 1, 0);
            FLACGPU_LAUNCH(12, 2, 0); FLACGPU_LAUNCH(32, 2, 0); FLACGPU_LAUNCH(12, 3, 0); FLACGPU_LAUNCH(32, 3, 0);
        }
#undef FLACGPU_LAUNCH
        h->launches += 8;
        if (h->flags & FLACGPU_FLAG_TIMING) { k_or_class_flags<<<1, 64, 0, st>>>(p.class_flags, (uint32_t *)h->counter.p + 12); ++h->launches; }
    }
    mark(h, EV_DECODE, st);
    uint32_t classes = 0;
    if (h->flags & FLACGPU_FLAG_TIMING)
        HIP_TRY(hipMemcpyAsync(&classes, (uint32_t *)h->counter.p + 12, 4, hipMemcpyDeviceToHost, st));
    if ((rc = fetch_results(h, n_streams, results, st))) return rc;
    collect_timing(h, in_bytes, out_bytes);
    h->timing.decode_classes = classes;

    // MD5 check: every stream the flag applies to is hashed where it lies, one lane per stream
    if ((h->flags & FLACGPU_FLAG_VERIFY_MD5) && results && h->output_format == FLACGPU_OUT_PACKED) {
        std::vector<uint64_t> offs, lens;
        std::vector<uint32_t> which;
        static const uint8_t zeros[16] = {};
        for (uint32_t i = 0; i < n_streams; ++i) {
            const flacgpu_stream &s = streams[i];
            const uint64_t want = s.info.total_samples * s.info.channels * ((s.info.bits_per_sample + 7) / 8);
            if (!s.info.total_samples || s.sample_base || !std::memcmp(s.info.md5, zeros, 16)) continue;
            if (results[i].frames_ok != results[i].frames || results[i].samples != s.info.total_samples || want > s.out_capacity) continue;
            which.push_back(i); offs.push_back(s.out_offset); lens.push_back(want);
        }
        if (!which.empty()) {
            h->h_md5.resize(which.size() * 16);
            rc = flacgpu_md5_batch_device(h, d_out, out_bytes, offs.data(), lens.data(), (uint32_t)which.size(), h->h_md5.data(), st);
            if (rc) return rc;
            for (size_t j = 0; j < which.size(); ++j)
                results[which[j]].md5_status = std::memcmp(&h->h_md5[j * 16], streams[which[j]].info.md5, 16) ? FLACGPU_MD5_MISMATCH
                                                                                                            : FLACGPU_MD5_MATCH;
        }
    }
    return FLACGPU_OK;
}

int flacgpu_decode_batch_device(flacgpu_handle *h, const void *d_in, uint64_t in_bytes, void *d_out, uint64_t out_bytes,
                                const flacgpu_stream *streams, uint32_t n_streams, flacgpu_stream_result *results,
                                void *hip_stream)
{
    if (!h || !d_in || !d_out || !streams || !n_streams) return fail(h, FLACGPU_ERR_ARG, "null argument");
    if ((uintptr_t)d_in & 15) return fail(h, FLACGPU_ERR_ARG, "d_in must be 16-byte aligned");
    HIP_TRY(hipSetDevice(h->device));
    return decode_device(h, d_in, in_bytes, d_out, out_bytes, streams, n_streams, results,
                         hip_stream ? (hipStream_t)hip_stream : h->own_stream);
}

// ---- the entry point that allocates nothing ---------------------------------------------------------
//
// Its workspace is a file-scope device array.  Every GPU the library runs on holds its own copy at its
// own address until the process ends; `scratch` below is only the host-side name of the symbol, never an
// address a kernel may be given: the per-device address comes from hipGetSymbolAddress, once per device,
// and everything after that uses that pointer.
namespace {

constexpr size_t kNaMaxIn = FLACGPU_NOALLOC_MAX_IN_BYTES;
constexpr size_t kNaMaxStreams = FLACGPU_NOALLOC_MAX_STREAMS;
constexpr size_t kNaMaxCand = FLACGPU_NOALLOC_MAX_CANDIDATES;
constexpr size_t kNaTiles = (kNaMaxIn + kScanTileBytes - 1) / kScanTileBytes;
constexpr size_t na_round(size_t n) { return (n + 255) & ~(size_t)255; }
// sizes of the pieces, in carving order (each a multiple of 256 bytes)
constexpr size_t kNaSizes[] = {
    na_round(kNaMaxStreams * sizeof(StreamDev)),      // streams
    na_round(kNaMaxStreams * sizeof(ResultDev)),      // results
    na_round(kNaTiles * kTileGranules * 2),           // mask
    na_round(kNaTiles * kTileChunks * 8),             // bitmap
    na_round(kNaTiles * 4),                           // tile_count
    na_round(kNaTiles * 4),                           // tile_base
    na_round(((kNaTiles + kPrefixGroup - 1) / kPrefixGroup) * 4),   // group_sum
    na_round(((kNaTiles + kPrefixGroup - 1) / kPrefixGroup) * 4),   // group_base
    na_round(64),                                     // counter
    na_round(kCrcTableEntries * 2),                   // crc_tables
    na_round(kNaTiles * kScanWaves * kSegSlots * 4),  // segrec
    na_round(kNaMaxCand * 8), na_round(kNaMaxCand * 8), na_round(kNaMaxCand * 4), na_round(kNaMaxCand * 4),   // goff num info stream
    na_round(kNaMaxCand * 4), na_round(kNaMaxCand * 4), na_round(kNaMaxCand * 2), na_round(kNaMaxCand * 4),   // len status segcrc seglen
    na_round(kNaMaxCand * 8 * 4), na_round(kNaMaxCand * 4),                                                  // subpos next
    na_round((2 * kNaMaxStreams + 64) * 4),                                                                  // shape_segs
};
constexpr size_t na_total()
{
    size_t t = 0;
    for (size_t x : kNaSizes) t += x;
    return t;
}
constexpr size_t kNaBytes = na_total();
static_assert(kNaBytes <= (4u << 20), "the static workspace is meant to stay modest");

}  // namespace

static __device__ __align__(256) uint8_t scratch[kNaBytes];

namespace {

constexpr int kNaMaxDevices = 64;
struct NoAllocState {
    flacgpu_handle h;
    uint8_t *ptr = nullptr;   // this device's copy of `scratch`
    bool ready = false;
};
NoAllocState g_na[kNaMaxDevices];
std::mutex g_na_mutex;
thread_local std::string g_na_error;

// Once per device, with that device current: the array's address there, the handle's buffers carved out of
// it, host vectors at their final size, CRC tables built.
int noalloc_setup(NoAllocState &s, int device, hipStream_t st)
{
    flacgpu_handle *h = &s.h;
    h->device = device;
    h->flags = 0;
    h->fixed = true;
    uint8_t *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return fail(h, FLACGPU_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    s.ptr = ptr;
    DevBuf *bufs[] = {&h->streams, &h->results, &h->mask, &h->bitmap, &h->tile_count, &h->tile_base, &h->group_sum, &h->group_base,
                      &h->counter, &h->crc_tables, &h->segrec, &h->goff, &h->num, &h->info, &h->stream, &h->len, &h->status, &h->segcrc,
                      &h->seglen, &h->subpos, &h->next, &h->shape_segs};
    static_assert(sizeof bufs / sizeof bufs[0] == sizeof kNaSizes / sizeof kNaSizes[0], "one size per buffer");
    size_t off = 0;
    for (size_t i = 0; i < sizeof bufs / sizeof bufs[0]; ++i) {
        bufs[i]->p = s.ptr + off;
        bufs[i]->bytes = kNaSizes[i];
        off += kNaSizes[i];
    }
    h->cand_cap = (uint32_t)kNaMaxCand;
    h->h_streams.reserve(kNaMaxStreams);
    h->h_results.reserve(kNaMaxStreams);
    h->h_shape_segs.reserve(2 * kNaMaxStreams + 64);
    k_crc16_build_tables<<<1, 256, 0, st>>>((uint16_t *)h->crc_tables.p);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    s.ready = true;
    return FLACGPU_OK;
}

}  // namespace

int flacgpu_decode_batch_device_noalloc(int device, const void *d_in, uint64_t in_bytes, void *d_out, uint64_t out_bytes,
                                        const flacgpu_stream *streams, uint32_t n_streams, flacgpu_stream_result *results,
                                        void *hip_stream)
{
    // A call that is refused for its arguments must not touch the state a running call on the same GPU is using
    // (its error text included): these refusals go to a string of the calling thread.
    g_na_error.clear();
    auto refuse = [](int code, const char *msg) { g_na_error = msg; return code; };
    if (device < 0 || device >= kNaMaxDevices) return refuse(FLACGPU_ERR_DEVICE, "device ordinal out of range");
    if (!d_in || !d_out || !streams || !n_streams) return refuse(FLACGPU_ERR_ARG, "null argument");
    if ((uintptr_t)d_in & 15) return refuse(FLACGPU_ERR_ARG, "d_in must be 16-byte aligned");
    if (in_bytes > kNaMaxIn) return refuse(FLACGPU_ERR_CAPACITY, "in_bytes exceeds FLACGPU_NOALLOC_MAX_IN_BYTES");
    if (n_streams > kNaMaxStreams) return refuse(FLACGPU_ERR_CAPACITY, "n_streams exceeds FLACGPU_NOALLOC_MAX_STREAMS");
    NoAllocState &s = g_na[device];
    flacgpu_handle *h = &s.h;
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)hip_stream;   // NULL: the legacy default stream
    {
        std::lock_guard<std::mutex> lock(g_na_mutex);
        if (!s.ready) {
            int rc = noalloc_setup(s, device, st);
            if (rc) return rc;
        }
    }
    return decode_device(h, d_in, in_bytes, d_out, out_bytes, streams, n_streams, results, st);
}

const char *flacgpu_noalloc_last_error(int device)
{
    if (!g_na_error.empty() || device < 0 || device >= kNaMaxDevices) return g_na_error.c_str();   // this thread's last call was refused for its arguments
    return g_na[device].h.err.c_str();
}

int flacgpu_decode_batch(flacgpu_handle *h, const uint8_t *const *bufs, const uint64_t *lens,
                         const flacgpu_streaminfo *infos, uint32_t n_streams, uint8_t *const *out_pcm,
                         const uint64_t *out_caps, flacgpu_stream_result *results)
{
    if (!h || !bufs || !lens || !infos || !out_pcm || !out_caps || !n_streams) return fail(h, FLACGPU_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<HostItem> items(n_streams);
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (lens[i] >= (1ull << 31)) return fail(h, FLACGPU_ERR_UNSUPPORTED, "stream of 2 GiB or more");
        items[i] = HostItem{bufs[i], lens[i], out_pcm[i], out_caps[i], 0, 0};
    }
    std::vector<flacgpu_stream> streams;
    flacgpu_timing sum = {};
    // one chunk of streams: device-resident decode of its arena (offsets are relative to the chunk)
    auto compute = [&](size_t first, size_t count, uint8_t *d_in, uint64_t in_bytes, uint8_t *d_out, uint64_t out_bytes, hipStream_t st) -> int {
        streams.resize(count);
        for (size_t k = 0; k < count; ++k) {
            flacgpu_stream &s = streams[k];
            s.in_offset = items[first + k].in_off; s.in_length = lens[first + k];
            s.out_offset = items[first + k].out_off; s.out_capacity = out_caps[first + k];
            s.sample_base = 0;
            s.info = infos[first + k];
        }
        const int rc = decode_device(h, d_in, std::max<uint64_t>(in_bytes, 16), d_out, out_bytes, streams.data(), (uint32_t)count,
                                     results ? results + first : nullptr, st);
        const flacgpu_timing &t = h->timing;
        sum.scan_ms += t.scan_ms; sum.compact_ms += t.compact_ms; sum.crc16_ms += t.crc16_ms; sum.resolve_ms += t.resolve_ms;
        sum.skim_ms += t.skim_ms; sum.decode_ms += t.decode_ms; sum.results_ms += t.results_ms; sum.total_ms += t.total_ms;
        sum.candidates += t.candidates; sum.in_bytes += t.in_bytes; sum.out_bytes += t.out_bytes; sum.decode_classes |= t.decode_classes;
        sum.crc16_reread_spans += t.crc16_reread_spans; sum.launches += t.launches;
        sum.shape_groups = std::max(sum.shape_groups, t.shape_groups);
        return rc;
    };
    const int rc = run_staged(h, *host_pipe(h), items, h->own_stream, true, compute, [&](size_t i) { return out_caps[i]; });
    h->timing = sum;   // device time of all chunks
    return rc;
}

}  // extern "C"
