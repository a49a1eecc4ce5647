// flacgpu_encode.hip — host driver of the batched encoder: validation, the per-stream tables, kernel
// sequencing and the encode entry points of include/flacgpu.h.
//
// An encode call runs on one HIP stream:
//   stats (workgroup per frame x candidate channel) -> design (lane per candidate) -> choose (workgroup per
//   candidate) -> layout (workgroup per stream) ->
//   pack (workgroup per frame x channel) -> CRC-16 of every frame -> CRC bytes,
// with the optional MD5 kernel beside them on a second stream.  Nothing is read back in between:
// frame sizes, offsets and stereo decisions stay on the device.
#include "handle.hpp"
#include "host_stage.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "encode.hpp"

using namespace flacgpu;

namespace {

// Tukey window with half of the block tapered, single precision (the table the analysis kernel reads).
void tukey_window(uint32_t bs, float *w)
{
    const uint32_t np = bs / 4;
    for (uint32_t i = 0; i < bs; ++i) w[i] = 1.0f;
    for (uint32_t i = 0; i < np; ++i) {
        const double v = 0.5 - 0.5 * std::cos(3.14159265358979323846 * (double)(i + 1) / (double)(np + 1));
        w[i] = (float)v;
        w[bs - 1 - i] = (float)v;
    }
}

int encode_init(flacgpu_handle *h)
{
    if (h->enc_ready) return FLACGPU_OK;
    HIP_TRY(hipStreamCreateWithFlags(&h->md5_stream, hipStreamNonBlocking));
    for (int i = 0; i < EE_COUNT; ++i) HIP_TRY(hipEventCreate(&h->eev[i]));
    uint32_t T[64];
    for (int i = 0; i < 64; ++i) T[i] = (uint32_t)(std::fabs(std::sin((double)(i + 1))) * 4294967296.0);
    int rc = reserve(h, h->e_md5_T, sizeof T);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(h->e_md5_T.p, T, sizeof T, hipMemcpyHostToDevice));
    h->enc_ready = true;
    return FLACGPU_OK;
}

template <int M, int NT>
int launch_analysis(flacgpu_handle *h, uint32_t, size_t lds, hipStream_t st, const uint8_t *pcm, const EncStreamDev *s,
                    uint32_t n, const uint32_t *hints, const EncOptsDev &o, const float *win, uint8_t *stats, uint8_t *plans,
                    bool timing, bool pick, bool pick_direct)
{
    if (pick) {
        // channel pair of every frame of the streams that asked for the estimate, before any candidate is analysed.
        // The estimate's sums of a frame (16 floats) wait for k_enc_pick_mode in the statistics buffer, which
        // k_enc_stats only fills afterwards and which holds at least 200 bytes for every frame.
        float *sums = reinterpret_cast<float *>(stats);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_enc_pick<NT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)));
        const uint32_t pairs = o.total_units / 2, pchunk = (0xF0000000u / NT) & ~7u;
        for (uint32_t b = 0; b < pairs; b += pchunk) {
            const uint32_t e = pairs - b < pchunk ? pairs : b + pchunk;
            if (pick_direct) k_enc_pick<NT, true><<<(e - b + 7) / 8 * 8, NT, 0, st>>>(pcm, s, n, hints, o, win, sums, b, e);   // 16-bit samples only: no LDS staging
            else k_enc_pick<NT, false><<<(e - b + 7) / 8 * 8, NT, 2 * lds, st>>>(pcm, s, n, hints, o, win, sums, b, e);
        }
        k_enc_pick_mode<<<(pairs + 63) / 64, 64, 0, st>>>(s, n, hints, o, sums, pairs);
        HIP_TRY(hipGetLastError());
    }
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_PICK], st));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_enc_stats<M, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_enc_choose<M, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // a launch holds fewer than 2^32 work-items, so large batches go in several launches
    const uint32_t chunk = (0xF0000000u / NT) & ~7u;
    for (uint32_t b = 0; b < o.total_units; b += chunk) {
        const uint32_t e = o.total_units - b < chunk ? o.total_units : b + chunk;
        k_enc_stats<M, NT><<<(e - b + 7) / 8 * 8, NT, lds, st>>>(pcm, s, n, hints, o, win, stats, plans, b, e);
    }
    HIP_TRY(hipGetLastError());
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_STATS], st));
    k_enc_design<M><<<(o.total_units + 63) / 64, 64, 0, st>>>(s, n, hints, o, stats);
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_DESIGN], st));
    for (uint32_t b = 0; b < o.total_units; b += chunk) {
        const uint32_t e = o.total_units - b < chunk ? o.total_units : b + chunk;
        k_enc_choose<M, NT><<<(e - b + 7) / 8 * 8, NT, lds, st>>>(pcm, s, n, hints, o, stats, plans, b, e);
    }
    HIP_TRY(hipGetLastError());
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_CHOOSE], st));
    return FLACGPU_OK;
}

template <int M, int NT>
int launch_pack(flacgpu_handle *h, uint32_t, size_t lds, hipStream_t st, const uint8_t *pcm, const EncStreamDev *s, uint32_t n,
                const uint32_t *hints, const EncOptsDev &o, const uint8_t *plans, const FrameRec *frames, uint8_t *out,
                uint32_t buf_words)
{
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_enc_pack<M, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t chunk = (0xF0000000u / NT) & ~7u;
    for (uint32_t b = 0; b < o.total_pack; b += chunk) {
        const uint32_t e = o.total_pack - b < chunk ? o.total_pack : b + chunk;
        k_enc_pack<M, NT><<<(e - b + 7) / 8 * 8, NT, lds, st>>>(pcm, s, n, hints, o, plans, frames, out, buf_words, b, e);
    }
    HIP_TRY(hipGetLastError());
    return FLACGPU_OK;
}

}  // namespace

namespace flacgpu {

void encode_release(flacgpu_handle *h)
{
    DevBuf *all[] = {&h->e_streams, &h->e_plans, &h->e_frames, &h->e_crc_off, &h->e_crc_len, &h->e_crc, &h->e_windows, &h->e_md5_T, &h->e_stats, &h->e_hints, &h->e_modes};
    for (DevBuf *b : all) release(*b);
    for (int i = 0; i < EE_COUNT; ++i)
        if (h->eev[i]) (void)hipEventDestroy(h->eev[i]);
    if (h->md5_stream) (void)hipStreamDestroy(h->md5_stream);
    h->enc_ready = false;
}

}  // namespace flacgpu

extern "C" {

uint64_t flacgpu_encode_bound(uint64_t total_samples, uint32_t channels, uint32_t bits_per_sample, uint32_t blocksize)
{
    if (!blocksize) return 0;
    const uint64_t frames = (total_samples + blocksize - 1) / blocksize;
    const uint64_t per_frame = 16 + 2 + ((uint64_t)blocksize * channels * (bits_per_sample + 1) + 7) / 8 + 8ull * channels;
    return 42 + frames * per_frame;
}

uint64_t flacgpu_seektable_bytes(uint64_t total_samples, uint32_t blocksize, uint32_t seektable_interval)
{
    if (!blocksize || !seektable_interval) return 0;
    const uint64_t frames = (total_samples + blocksize - 1) / blocksize;
    return 4 + 18 * ((frames + seektable_interval - 1) / seektable_interval);
}

int flacgpu_get_encode_timing(const flacgpu_handle *h, flacgpu_encode_timing *out)
{
    if (!h || !out) return FLACGPU_ERR_ARG;
    *out = h->enc_timing;
    return FLACGPU_OK;
}

int flacgpu_encode_batch_device(flacgpu_handle *h, const void *d_pcm, uint64_t pcm_bytes, void *d_out, uint64_t out_bytes,
                                const flacgpu_pcm_stream *streams, uint32_t n_streams, const flacgpu_encode_options *opts,
                                flacgpu_encode_result *results, void *hip_stream)
{
    if (!h || !d_pcm || !d_out || !streams || !n_streams || !opts) return fail(h, FLACGPU_ERR_ARG, "null argument");
    if (opts->struct_size < sizeof(flacgpu_encode_options)) return fail(h, FLACGPU_ERR_ARG, "flacgpu_encode_options.struct_size");
    if (((uintptr_t)d_pcm & 15) || ((uintptr_t)d_out & 15)) return fail(h, FLACGPU_ERR_ARG, "d_pcm and d_out must be 16-byte aligned");
    if (opts->blocksize < 16 || opts->blocksize > (uint32_t)kEncMaxBs) return fail(h, FLACGPU_ERR_UNSUPPORTED, "blocksize must be 16..16384");
    if (opts->max_lpc_order > 32 || opts->force_lpc_order > 32) return fail(h, FLACGPU_ERR_ARG, "LPC order above 32");
    if (opts->max_partition_order > (uint32_t)kEncMaxPo || opts->min_partition_order > opts->max_partition_order)
        return fail(h, FLACGPU_ERR_ARG, "partition orders must satisfy min <= max <= 8");
    if (opts->qlp_precision && (opts->qlp_precision < 5 || opts->qlp_precision > 15)) return fail(h, FLACGPU_ERR_ARG, "qlp_precision must be 0 or 5..15");
    if (opts->stereo_mode < -2 || opts->stereo_mode > 3) return fail(h, FLACGPU_ERR_ARG, "stereo_mode must be -2..3");
    HIP_TRY(hipSetDevice(h->device));
    int rc = encode_init(h);
    if (rc) return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    const bool timing = (h->flags & FLACGPU_FLAG_TIMING) != 0;
    const bool md5 = (opts->flags & FLACGPU_ENC_MD5) != 0;
    const uint32_t B = opts->blocksize;

    // ---- per-stream table and window table
    std::vector<EncStreamDev> hs(n_streams);
    std::map<uint32_t, uint32_t> win_index;   // block size -> offset in floats
    std::vector<float> windows;
    auto window_of = [&](uint32_t bs) {
        auto it = win_index.find(bs);
        if (it != win_index.end()) return it->second;
        const uint32_t off = (uint32_t)windows.size();
        windows.resize(off + ((bs + 3) & ~3u));
        tukey_window(bs, windows.data() + off);
        win_index[bs] = off;
        return off;
    };
    uint64_t frames = 0, units = 0, pack_units = 0, in_total = 0;
    uint32_t max_bps = 0;
    bool any_estimated = false, pick_direct = true;   // pick_direct: every stream that asks for the estimate has 16-bit samples
    for (uint32_t i = 0; i < n_streams; ++i) {
        const flacgpu_pcm_stream &s = streams[i];
        if (s.channels < 1 || s.channels > 8 || s.bits_per_sample < 4 || s.bits_per_sample > 24)
            return fail(h, FLACGPU_ERR_UNSUPPORTED, "channels must be 1..8 and bits per sample 4..24");
        if (s.sample_rate == 0 || s.sample_rate >= 655350) return fail(h, FLACGPU_ERR_STREAMINFO, "sample rate out of bounds");   // flacmeta.go:708
        if (s.total_samples >> 36) return fail(h, FLACGPU_ERR_STREAMINFO, "total_samples needs more than 36 bits");                // flacmeta.go:713
        const uint32_t bytes_ps = (s.bits_per_sample + 7) / 8;
        const uint64_t in_len = s.total_samples * s.channels * bytes_ps;
        if ((s.in_offset & 3) || (s.out_offset & 3)) return fail(h, FLACGPU_ERR_ARG, "in_offset and out_offset must be multiples of 4");
        if (s.in_offset > pcm_bytes || in_len > pcm_bytes - s.in_offset) return fail(h, FLACGPU_ERR_ARG, "stream exceeds the PCM buffer");
        if (s.out_offset > out_bytes || s.out_capacity > out_bytes - s.out_offset) return fail(h, FLACGPU_ERR_ARG, "stream exceeds the output buffer");
        EncStreamDev d;
        std::memset(&d, 0, sizeof d);
        d.in_off = s.in_offset; d.out_off = s.out_offset; d.out_cap = s.out_capacity; d.total_samples = s.total_samples;
        const uint64_t nf = (s.total_samples + B - 1) / B;
        d.frame_begin = (uint32_t)frames; d.n_frames = (uint32_t)nf;
        units = (units + 1) & ~1ull;              // unit ranges start even: k_enc_pick takes units in twos
        d.unit_begin = (uint32_t)units; d.pack_begin = (uint32_t)pack_units;
        d.sample_rate = s.sample_rate;
        d.tail_bs = nf ? (uint32_t)(s.total_samples - (nf - 1) * B) : 0;
        d.channels = s.channels; d.bps = s.bits_per_sample; d.bytes_ps = (uint8_t)bytes_ps;
        // stereo: four candidate channels with a plan each when the pair is chosen by exact size; two plans
        // per frame when the pair is known before the analysis (forced, or estimated per frame)
        d.pair = s.channels != 2 || opts->stereo_mode == FLACGPU_STEREO_EXACT ? kPairNone
                 : opts->stereo_mode == FLACGPU_STEREO_ESTIMATE ? kPairEstimated : opts->stereo_mode;
        d.ncand = s.channels == 2 ? (d.pair == kPairNone ? 4 : 2) : s.channels;
        any_estimated |= d.pair == kPairEstimated;
        if (d.pair == kPairEstimated && bytes_ps != 2) pick_direct = false;
        d.win_off = window_of(B);
        d.win_off_tail = d.tail_bs ? window_of(d.tail_bs) : d.win_off;
        d.meta_len = 42 + (uint32_t)flacgpu_seektable_bytes(s.total_samples, B, opts->seektable_interval);
        if (d.meta_len - 46 >= (1u << 24) && opts->seektable_interval) return fail(h, FLACGPU_ERR_ARG, "seek table longer than a metadata block allows");
        std::memcpy(d.md5, s.md5, 16);
        frames += nf; units += nf * d.ncand; pack_units += nf * d.channels; in_total += in_len;
        max_bps = std::max<uint32_t>(max_bps, s.bits_per_sample);
        if (units >= 0x7FFFFFF0ull) return fail(h, FLACGPU_ERR_UNSUPPORTED, "too many subframes in one batch");
        hs[i] = d;
    }
    units = (units + 1) & ~1ull;
    EncOptsDev o;
    std::memset(&o, 0, sizeof o);
    o.blocksize = B; o.max_lpc_order = opts->max_lpc_order; o.qlp_precision = opts->qlp_precision;
    o.min_po = opts->min_partition_order; o.max_po = opts->max_partition_order; o.stereo_mode = opts->stereo_mode;
    o.force_lpc_order = opts->force_lpc_order;
    o.seek_every = opts->seektable_interval;
    o.plan_stride = (uint32_t)((kPlanHeadBytes + kPlanCoefBytes + (1u << opts->max_partition_order) + 15) & ~15u);
    o.total_units = (uint32_t)units; o.total_pack = (uint32_t)pack_units; o.total_frames = (uint32_t)frames;

    // kernel geometry: predictor window (12 or 32 taps) and lanes per block (16 samples each)
    const uint32_t need_order = std::max(opts->max_lpc_order, opts->force_lpc_order);
    const int M = need_order <= 12 ? 12 : 32;
    const int NT = B <= 4096 ? 256 : 1024;
    // hints[k]: the stream that owns plan unit (first table) / pack unit (second table) k << 10
    const size_t nh_a = (size_t)(units >> 10) + 2, nh_p = (size_t)(pack_units >> 10) + 2;
    std::vector<uint32_t> hints(nh_a + nh_p);
    {
        uint32_t sa = 0, sp = 0;
        for (size_t k = 0; k < nh_a; ++k) {
            while (sa + 1 < n_streams && hs[sa + 1].unit_begin <= (uint64_t)k << 10) ++sa;
            hints[k] = sa;
        }
        for (size_t k = 0; k < nh_p; ++k) {
            while (sp + 1 < n_streams && hs[sp + 1].pack_begin <= (uint64_t)k << 10) ++sp;
            hints[nh_a + k] = sp;
        }
    }
    if ((rc = reserve(h, h->e_hints, hints.size() * 4))) return rc;
    if ((rc = reserve(h, h->e_modes, (size_t)std::max<uint64_t>(frames, 1)))) return rc;
    o.modes = (uint8_t *)h->e_modes.p;
    if ((rc = reserve(h, h->e_stats, (size_t)std::max<uint64_t>(units, 1) * stats_stride(M)))) return rc;
    if ((rc = reserve(h, h->e_streams, (size_t)n_streams * sizeof(EncStreamDev)))) return rc;
    if ((rc = reserve(h, h->e_windows, windows.size() * sizeof(float) + 16))) return rc;
    if ((rc = reserve(h, h->e_plans, (size_t)std::max<uint64_t>(units, 1) * o.plan_stride))) return rc;
    if ((rc = reserve(h, h->e_frames, (size_t)std::max<uint64_t>(frames, 1) * sizeof(FrameRec)))) return rc;
    if ((rc = reserve(h, h->e_crc_off, (size_t)std::max<uint64_t>(frames, 1) * 8))) return rc;
    if ((rc = reserve(h, h->e_crc_len, (size_t)std::max<uint64_t>(frames, 1) * 4))) return rc;
    if ((rc = reserve(h, h->e_crc, (size_t)std::max<uint64_t>(frames, 1) * 2))) return rc;
    HIP_TRY(hipMemcpyAsync(h->e_streams.p, hs.data(), (size_t)n_streams * sizeof(EncStreamDev), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->e_hints.p, hints.data(), hints.size() * 4, hipMemcpyHostToDevice, st));
    if (!windows.empty())
        HIP_TRY(hipMemcpyAsync(h->e_windows.p, windows.data(), windows.size() * sizeof(float), hipMemcpyHostToDevice, st));

    const uint8_t *pcm = (const uint8_t *)d_pcm;
    uint8_t *out = (uint8_t *)d_out;
    EncStreamDev *ds = (EncStreamDev *)h->e_streams.p;
    if (md5) {
        HIP_TRY(hipEventRecord(h->eev[EE_MD5_START], st));
        HIP_TRY(hipStreamWaitEvent(h->md5_stream, h->eev[EE_MD5_START], 0));
        k_enc_md5<<<(n_streams + 63) / 64, 64, 0, h->md5_stream>>>(pcm, ds, n_streams, (const uint32_t *)h->e_md5_T.p, out);
        HIP_TRY(hipEventRecord(h->eev[EE_MD5_END], h->md5_stream));
    }
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_START], st));

    const size_t xs_bytes = (size_t)enc_lds_words(NT) * 4;
    const uint32_t buf_words = (31 + 8 + 32 + B * (max_bps + 1) + 31) / 32 + 2;
    const size_t pack_lds = xs_bytes + (size_t)buf_words * 4;
    const uint32_t grid_a = (uint32_t)((units + 7) / 8 * 8), grid_p = (uint32_t)((pack_units + 7) / 8 * 8);
    const float *win = (const float *)h->e_windows.p;
    const uint32_t *hints_a = (const uint32_t *)h->e_hints.p, *hints_p = hints_a + nh_a;
    uint8_t *plans = (uint8_t *)h->e_plans.p, *stats = (uint8_t *)h->e_stats.p;
    FrameRec *fr = (FrameRec *)h->e_frames.p;
    if (units) {
        if (M == 12 && NT == 256) rc = launch_analysis<12, 256>(h, grid_a, xs_bytes, st, pcm, ds, n_streams, hints_a, o, win, stats, plans, timing, any_estimated, pick_direct);
        else if (M == 32 && NT == 256) rc = launch_analysis<32, 256>(h, grid_a, xs_bytes, st, pcm, ds, n_streams, hints_a, o, win, stats, plans, timing, any_estimated, pick_direct);
        else if (M == 12) rc = launch_analysis<12, 1024>(h, grid_a, xs_bytes, st, pcm, ds, n_streams, hints_a, o, win, stats, plans, timing, any_estimated, pick_direct);
        else rc = launch_analysis<32, 1024>(h, grid_a, xs_bytes, st, pcm, ds, n_streams, hints_a, o, win, stats, plans, timing, any_estimated, pick_direct);
        if (rc) return rc;
    } else if (timing) {
        HIP_TRY(hipEventRecord(h->eev[EE_PICK], st)); HIP_TRY(hipEventRecord(h->eev[EE_STATS], st)); HIP_TRY(hipEventRecord(h->eev[EE_DESIGN], st)); HIP_TRY(hipEventRecord(h->eev[EE_CHOOSE], st));
    }
    k_enc_layout<<<n_streams, 256, 0, st>>>(ds, o, plans, fr, (uint64_t *)h->e_crc_off.p, (uint32_t *)h->e_crc_len.p, out, md5 ? 1 : 0);
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_LAYOUT], st));
    if (pack_units) {
        if (M == 12 && NT == 256) rc = launch_pack<12, 256>(h, grid_p, pack_lds, st, pcm, ds, n_streams, hints_p, o, plans, fr, out, buf_words);
        else if (M == 32 && NT == 256) rc = launch_pack<32, 256>(h, grid_p, pack_lds, st, pcm, ds, n_streams, hints_p, o, plans, fr, out, buf_words);
        else if (M == 12) rc = launch_pack<12, 1024>(h, grid_p, pack_lds, st, pcm, ds, n_streams, hints_p, o, plans, fr, out, buf_words);
        else rc = launch_pack<32, 1024>(h, grid_p, pack_lds, st, pcm, ds, n_streams, hints_p, o, plans, fr, out, buf_words);
        if (rc) return rc;
    }
    if (timing) HIP_TRY(hipEventRecord(h->eev[EE_PACK], st));
    if (frames) {
        launch_crc16_spans(h, out, (const uint64_t *)h->e_crc_off.p, (const uint32_t *)h->e_crc_len.p, (uint16_t *)h->e_crc.p,
                           (uint32_t)frames, st);
        k_enc_write_crc<<<(uint32_t)((frames + 255) / 256), 256, 0, st>>>(fr, (const uint16_t *)h->e_crc.p, (uint32_t)frames, out);
    }
    if (timing) { HIP_TRY(hipEventRecord(h->eev[EE_CRC], st)); HIP_TRY(hipEventRecord(h->eev[EE_END], st)); }
    if (md5) HIP_TRY(hipStreamWaitEvent(st, h->eev[EE_MD5_END], 0));
    HIP_TRY(hipMemcpyAsync(hs.data(), h->e_streams.p, (size_t)n_streams * sizeof(EncStreamDev), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());

    uint64_t out_total = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        out_total += hs[i].out_len;
        if (!results) continue;
        flacgpu_encode_result &r = results[i];
        std::memset(&r, 0, sizeof r);
        r.out_length = hs[i].out_len; r.frames = hs[i].n_frames;
        r.min_framesize = hs[i].min_fs; r.max_framesize = hs[i].max_fs; r.status = hs[i].status;
        std::memcpy(r.md5, hs[i].md5, 16);
    }
    flacgpu_encode_timing t;
    std::memset(&t, 0, sizeof t);
    t.frames = frames; t.in_bytes = in_total; t.out_bytes = out_total;
    if (timing) {
        auto el = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, h->eev[a], h->eev[b]); return ms; };
        t.stats_ms = el(EE_START, EE_STATS);
        t.pick_ms = el(EE_START, EE_PICK);
        t.design_ms = el(EE_STATS, EE_DESIGN);
        t.choose_ms = el(EE_DESIGN, EE_CHOOSE);
        t.layout_ms = el(EE_CHOOSE, EE_LAYOUT);
        t.pack_ms = el(EE_LAYOUT, EE_PACK);
        t.crc_ms = el(EE_PACK, EE_CRC);
        t.total_ms = el(EE_START, EE_END);
        if (md5) t.md5_ms = el(EE_MD5_START, EE_MD5_END);
    }
    h->enc_timing = t;
    return FLACGPU_OK;
}

int flacgpu_md5_batch_device(flacgpu_handle *h, const void *d_buf, uint64_t buf_bytes, const uint64_t *offsets,
                             const uint64_t *lengths, uint32_t n, uint8_t *digests, void *hip_stream)
{
    if (!h || !d_buf || !offsets || !lengths || !n || !digests) return fail(h, FLACGPU_ERR_ARG, "null argument");
    for (uint32_t i = 0; i < n; ++i)
        if (offsets[i] > buf_bytes || lengths[i] > buf_bytes - offsets[i]) return fail(h, FLACGPU_ERR_ARG, "range exceeds the buffer");
    HIP_TRY(hipSetDevice(h->device));
    int rc = encode_init(h);
    if (rc) return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    // workspace: offsets, lengths, digests
    if ((rc = reserve(h, h->e_crc_off, (size_t)n * 8))) return rc;
    if ((rc = reserve(h, h->e_stats, (size_t)n * 8))) return rc;
    if ((rc = reserve(h, h->e_crc, (size_t)n * 16))) return rc;
    HIP_TRY(hipMemcpyAsync(h->e_crc_off.p, offsets, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->e_stats.p, lengths, (size_t)n * 8, hipMemcpyHostToDevice, st));
    k_md5_ranges<<<(n + 63) / 64, 64, 0, st>>>((const uint8_t *)d_buf, (const uint64_t *)h->e_crc_off.p, (const uint64_t *)h->e_stats.p, n,
                                               (const uint32_t *)h->e_md5_T.p, (uint8_t *)h->e_crc.p);
    HIP_TRY(hipMemcpyAsync(digests, h->e_crc.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return FLACGPU_OK;
}

int flacgpu_encode_batch(flacgpu_handle *h, const uint8_t *const *pcm, const flacgpu_pcm_stream *streams, uint32_t n_streams,
                         const flacgpu_encode_options *opts, uint8_t *const *out, const uint64_t *out_caps,
                         flacgpu_encode_result *results)
{
    if (!h || !pcm || !streams || !n_streams || !opts || !out || !out_caps || !results) return fail(h, FLACGPU_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<HostItem> items(n_streams);
    for (uint32_t i = 0; i < n_streams; ++i) {
        const uint64_t len = streams[i].total_samples * streams[i].channels * ((streams[i].bits_per_sample + 7) / 8);
        items[i] = HostItem{pcm[i], len, out[i], out_caps[i], 0, 0};
    }
    std::vector<flacgpu_pcm_stream> ss;
    flacgpu_encode_timing sum = {};
    auto compute = [&](size_t first, size_t count, uint8_t *d_in, uint64_t in_bytes, uint8_t *d_out, uint64_t out_bytes, hipStream_t st) -> int {
        ss.assign(streams + first, streams + first + count);
        for (size_t k = 0; k < count; ++k) {
            ss[k].in_offset = items[first + k].in_off; ss[k].out_offset = items[first + k].out_off;
            ss[k].out_capacity = out_caps[first + k];
        }
        const int rc = flacgpu_encode_batch_device(h, d_in, in_bytes + 256, d_out, out_bytes + 256, ss.data(), (uint32_t)count, opts,
                                                   results + first, st);
        const flacgpu_encode_timing &t = h->enc_timing;
        sum.pick_ms += t.pick_ms; sum.stats_ms += t.stats_ms; sum.design_ms += t.design_ms; sum.choose_ms += t.choose_ms; sum.layout_ms += t.layout_ms;
        sum.pack_ms += t.pack_ms; sum.crc_ms += t.crc_ms; sum.md5_ms += t.md5_ms; sum.total_ms += t.total_ms;
        sum.frames += t.frames; sum.in_bytes += t.in_bytes; sum.out_bytes += t.out_bytes;
        return rc;
    };
    const int rc = run_staged(h, *host_pipe(h), items, h->own_stream, false, compute,
                              [&](size_t i) { return results[i].status == FLACGPU_OK ? results[i].out_length : 0ull; });
    h->enc_timing = sum;
    return rc;
}

}  // extern "C"
