"""GPU parity tests: libflacgpu (HIP, through the C ABI) against the CPU oracle,
the STREAMINFO MD5 signatures and the seek tables of the reference fixtures."""
import os

import numpy as np
import pytest

from cases import CASES, encode_case
from util import flac_amd, gpu_batch, md5hex, oracle, oracle_decode, split_file, synth, to_oracle_si

pytestmark = pytest.mark.gpu

MONO = "44100-16-mono.flac"
STEREO = "silence-44-s.flac"


def _read(testdata, name):
    return open(os.path.join(testdata, name), "rb").read()


def test_fixtures_md5(decoder, testdata):
    """Both reference fixtures, bit-exact against the MD5 the
    reference's tests assert (flacmeta_test.go:37, :121)."""
    files = [_read(testdata, MONO), _read(testdata, STEREO)]
    pcms, res, infos = gpu_batch(decoder, files)
    assert md5hex(pcms[0]) == "e5ccc967ced6c111530e5c79e33c969e" == infos[0].md5_signature
    assert md5hex(pcms[1]) == "6291dbd8dcb7dc480132e4c4ba154a17" == infos[1].md5_signature
    assert (res[0].frames, res[0].frames_ok, res[0].samples) == (248, 248, 1014300)
    assert (res[1].frames, res[1].frames_ok, res[1].samples) == (36, 36, 162496)
    assert res[1].header_candidates == 36
    for p, f in zip(pcms, files):
        si, a = split_file(f)
        want, _ = oracle_decode(si, a)
        assert np.array_equal(p, want)


def test_fixture_frame_tables(decoder, testdata):
    """Frame offsets equal the SEEKTABLE offsets (flacmeta_test.go:81-85, :225-232),
    frame sizes stay within Min/MaxFrameSize (:31-32, :115-116), and the whole
    table equals the oracle's."""
    import torch
    for name, seeks in ((MONO, {438272: 1177, 880640: 2452}),
                        (STEREO, {41472: 11852, 50688: 14484, 87552: 25022, 105984: 30284})):
        si, a = split_file(_read(testdata, name))
        d_in = torch.frombuffer(bytearray(a) + bytearray(64), dtype=torch.uint8).cuda()
        streams = decoder.make_streams([(0, len(a), 0, 0, si)])
        frames, res = decoder.scan_frames(d_in, len(a), streams)
        by_sample = {f.first_sample: f.offset for f in frames}
        assert by_sample[0] == 0
        for s, off in seeks.items():
            assert by_sample[s] == off
        sizes = [f.length for f in frames]
        assert min(sizes) == si.min_framesize and max(sizes) == si.max_framesize
        ofr, ost = oracle.scan_frames(a, to_oracle_si(si))
        assert len(ofr) == len(frames) == res[0].frames
        for g, o in zip(frames, ofr):
            assert (g.offset, g.length, g.blocksize, g.first_sample, g.number, g.status, g.ch_assign, g.bps,
                    g.header_len, g.variable) == (o.offset, o.length, o.blocksize, o.first_sample, o.number,
                                                  o.status, o.ch_assign, o.bps, o.header_len, o.variable)


def test_synthetic_cases_one_batch(decoder):
    """Every subframe type / Rice mode / channel layout in ONE mixed batch, against
    the oracle and against the source PCM (lossless round trip + STREAMINFO MD5)."""
    enc = [encode_case(i, c) for i, c in enumerate(CASES)]
    files = [f for _, f, _ in enc]
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0xAA)
    for (name, f, x), p, r, si in zip(enc, pcms, res, infos):
        want, ost = oracle_decode(si, split_file(f)[1])
        src = synth.pack_pcm(x, si.bits_per_sample)
        assert bytes(want) == src, name
        assert md5hex(src) == si.md5_signature, name
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), name
        assert np.array_equal(p, want), name
        assert r.pcm_bytes == len(src), name


@pytest.mark.parametrize("lead", [1, 2, 3])
def test_odd_input_and_output_addresses(decoder, lead):
    """Streams packed back to back at arbitrary byte offsets, output regions at odd addresses with no gap
    between them: the write-out falls back from dword stores to bytes, must produce the same PCM and must not
    touch a byte outside a stream's region (every shape of the case list, one batch)."""
    enc = [encode_case(i, c) for i, c in enumerate(CASES)]
    files = [f for _, f, _ in enc]
    pcms, res, infos, raw, offs = gpu_batch(decoder, files, pad=1, out_pad=1, in_lead=lead, out_lead=lead, out_fill=0xA5,
                                           return_raw=True)
    covered = np.zeros(raw.size, dtype=bool)
    for (name, f, x), p, r, si, o in zip(enc, pcms, res, infos, offs):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), name
        assert np.array_equal(p, want), name
        covered[o: o + si.pcm_bytes] = True
    assert np.all(raw[~covered] == 0xA5)


def test_thousands_of_small_streams(decoder):
    """A batch of 4000 streams of one to three short frames each, with empty inputs and inputs that hold
    no frame at all scattered among them: per-stream candidate ranges, results and output regions must stay
    apart (checked against the oracle once per distinct stream)."""
    rng = np.random.default_rng(3)
    kinds = []
    for i, (ch, bps, n, bs) in enumerate([(2, 16, 17, 4096), (1, 16, 700, 576), (2, 24, 1200, 1152), (6, 16, 300, 256),
                                          (2, 16, 4097, 4096), (2, 8, 40, 192)]):
        x = synth.pcm(900 + i, n, ch, bps, 44100, bs)
        f, _ = synth.encode(x, bps, 44100, synth.options(bs, 8))
        kinds.append(bytes(f))
    si0, _ = split_file(kinds[0])
    head0 = kinds[0][: len(kinds[0]) - len(split_file(kinds[0])[1])]
    kinds.append(head0)                                                         # STREAMINFO, no audio bytes at all
    kinds.append(head0 + bytes(rng.integers(0, 256, size=500).astype(np.uint8)))  # no frame among the bytes
    pick = rng.integers(0, len(kinds), size=4000)
    files = [kinds[k] for k in pick]
    pcms, res, infos = gpu_batch(decoder, files, pad=1, out_fill=0x11)
    expect = []
    for f in kinds:
        si, a = split_file(f)
        expect.append(oracle_decode(si, a))
    for k, p, r in zip(pick, pcms, res):
        want, ost = expect[k]
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples)
        if ost.frames_ok:
            assert np.array_equal(p, want)
        else:
            assert np.all(p == 0x11)


def test_unknown_total_and_short_capacity(decoder):
    """STREAMINFO.TotalSamples == 0 means unknown (flacmeta.go:662 only stores the field): every frame
    that fits the caller's capacity is decoded.  A capacity that ends inside the stream turns the frames
    beyond it into range errors and leaves the bytes behind the capacity alone.  Both against the oracle."""
    import copy
    import torch
    enc = [encode_case(i, c) for i, c in enumerate(CASES[:3])]
    entries, wants, pos_in, pos_out = [], [], 0, 0
    blobs = []
    for name, f, x in enc:
        si, a = split_file(f)
        full = si.pcm_bytes
        for total, cap in ((0, full), (si.total_samples, full // 2), (0, full // 3)):
            si2 = copy.copy(si)
            si2.total_samples = total
            pcm, ost = oracle.decode_stream(a, to_oracle_si(si2), pcm_cap=cap)
            want = np.zeros(cap, dtype=np.uint8)
            want[: pcm.size] = pcm
            entries.append((pos_in, len(a), pos_out, cap, si2))
            wants.append((name, total, cap, want, ost))
            blobs.append(a)
            pos_in += (len(a) + 63) // 64 * 64
            pos_out += (full + 63) // 64 * 64        # room for the whole stream: what lies behind `cap` must stay untouched
    host = np.zeros(pos_in, dtype=np.uint8)
    for (o, n, *_), a in zip(entries, blobs):
        host[o: o + n] = np.frombuffer(a, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.full((pos_out,), 0xEE, dtype=torch.uint8, device="cuda")
    res = decoder.decode_batch_device(d_in, pos_in, d_out, pos_out, decoder.make_streams(entries))
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    ends = [e[2] for e in entries[1:]] + [pos_out]
    for (o_in, n, o_out, cap, si2), (name, total, cap_, want, ost), r, end in zip(entries, wants, res, ends):
        what = f"{name} total={total} cap={cap}"
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), what
        got = out[o_out: o_out + cap]
        decoded = int(ost.samples) * si2.channels * ((si2.bits_per_sample + 7) // 8)
        assert np.array_equal(got[:decoded], want[:decoded]), what
        assert np.all(out[o_out + cap: end] == 0xEE), what


def test_decode_rejects_bad_batches(decoder, testdata):
    """Batch descriptions the kernels could not index safely are refused before anything is launched:
    streams out of order or overlapping, regions beyond either buffer, STREAMINFO fields out of bounds,
    a misaligned input pointer."""
    import copy
    import torch
    si, a = split_file(_read(testdata, STEREO))
    n = len(a)
    host = np.zeros(2 * ((n + 63) // 64 * 64) + 64, dtype=np.uint8)
    host[:n] = np.frombuffer(a, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.zeros(2 * si.pcm_bytes + 64, dtype=torch.uint8, device="cuda")
    ib, ob = host.size, d_out.numel()
    ok = (0, n, 0, si.pcm_bytes, si)
    res = decoder.decode_batch_device(d_in, ib, d_out, ob, decoder.make_streams([ok]))
    assert res[0].frames_ok == 36

    def bad_info(**kw):
        s2 = copy.copy(si)
        for k, v in kw.items():
            setattr(s2, k, v)
        return s2
    second = (n + 63) // 64 * 64
    for entries in ([(second, n, 0, si.pcm_bytes, si), (0, n, si.pcm_bytes, si.pcm_bytes, si)],      # descending
                    [(0, n, 0, si.pcm_bytes, si), (n - 1, n, si.pcm_bytes, si.pcm_bytes, si)],        # overlapping inputs
                    [(ib - 10, n, 0, si.pcm_bytes, si)],                                              # input beyond the buffer
                    [(0, n, ob - 10, si.pcm_bytes, si)],                                              # output beyond the buffer
                    [(0, n, 0, si.pcm_bytes, bad_info(channels=0))],
                    [(0, n, 0, si.pcm_bytes, bad_info(channels=9))],
                    [(0, n, 0, si.pcm_bytes, bad_info(bits_per_sample=3))],
                    [(0, n, 0, si.pcm_bytes, bad_info(bits_per_sample=33))],
                    [(0, n, 0, si.pcm_bytes, bad_info(max_blocksize=15))]):
        with pytest.raises(flac_amd.FlacError):
            decoder.decode_batch_device(d_in, ib, d_out, ob, decoder.make_streams(entries))
    with pytest.raises(flac_amd.FlacError):
        decoder.decode_batch_device(d_in[4:], ib - 4, d_out, ob, decoder.make_streams([ok]))           # pointer not 16-byte aligned
    # the handle is still good afterwards
    res = decoder.decode_batch_device(d_in, ib, d_out, ob, decoder.make_streams([ok]))
    torch.cuda.synchronize()
    assert res[0].frames_ok == 36 and md5hex(d_out[: si.pcm_bytes].cpu().numpy()) == si.md5_signature


def test_host_buffer_entry_point(decoder, testdata):
    """flacgpu_decode_batch (host pointers, PCIe copies inside)."""
    pcm, si, res = flac_amd.decode_file(os.path.join(testdata, STEREO), decoder)
    assert md5hex(pcm) == si.md5_signature
    assert res.frames_ok == 36


def test_long_rice_codes(decoder):
    """Rare full-scale spikes in quiet audio with one Rice partition per block give codes far longer than
    the 32-bit window of the unrolled decode loop: the lanes that meet one are rolled back and redone by the
    generic loop, their neighbours keep their results."""
    rng = np.random.default_rng(5)
    files, srcs = [], []
    for ch, bps in ((2, 16), (1, 16), (2, 24)):
        x = rng.integers(-40, 40, size=(4096 * 3 + 77, ch)).astype(np.int32)
        spikes = rng.integers(0, x.shape[0], size=60)
        x[spikes, rng.integers(0, ch, size=60)] = rng.choice([-(1 << (bps - 1)), (1 << (bps - 1)) - 1], size=60)
        f, _ = synth.encode(x, bps, 44100, synth.options(4096, 8, force_partition_order=0, stereo_mode=0))
        files.append(bytes(f))
        srcs.append(synth.pack_pcm(x, bps))
    pcms, res, infos = gpu_batch(decoder, files)
    for f, p, src, si, r in zip(files, pcms, srcs, infos, res):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src
        assert np.array_equal(p, want)
        assert r.frames_ok == ost.frames_ok == 4


def test_long_rice_codes_among_32_tap_neighbours(decoder):
    """The same spikes with order-32 predictors, where a group of the unrolled loop is two write-out tiles
    long: the first tile has already been written when a lane is rolled back, and the generic loop writes
    it again for the whole wave.  Only channel 0 has spikes, so every wave mixes redone lanes with lanes
    that keep their results; both tiles of both kinds must come out right (16-bit: 24-bit multiplies,
    24-bit: 64-bit sums)."""
    rng = np.random.default_rng(11)
    files, srcs = [], []
    for ch, bps in ((2, 16), (8, 24), (3, 24)):
        n = 4096 * 3 + 333
        x = rng.integers(-300, 300, size=(n, ch)).astype(np.int32)
        x[:, 1:] += (2000 * np.sin(np.arange(n) / 9.0))[:, None].astype(np.int32)
        x[:, 0] = rng.integers(-3, 4, size=n)
        spikes = rng.integers(64, n, size=40)
        x[spikes, 0] = rng.choice([-(1 << (bps - 1)), (1 << (bps - 1)) - 1], size=40)
        f, _ = synth.encode(x, bps, 48000, synth.options(4096, 32, force_lpc_order=32, force_partition_order=0, stereo_mode=0))
        files.append(bytes(f))
        srcs.append(synth.pack_pcm(x, bps))
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x55)
    for f, p, src, si, r in zip(files, pcms, srcs, infos, res):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src
        assert np.array_equal(p, want)
        assert r.frames_ok == ost.frames_ok == 4


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_side_channels_wider_than_16_bits(decoder, mode):
    """16-bit stereo with a side channel that needs its 17th bit.  The decode class for such streams multiplies
    sample pairs as 16-bit halves and stages 16-bit halves for the write-out, assuming what almost always holds:
    that side samples fit 16 bits.  Here they do not, in every way there is: stretches of opposite full-scale
    channels in otherwise quiet streams (lanes rolled back among lanes that keep their results), streams that are
    wide throughout, wide constant side channels, wasted bits below a wide side channel, wide warm-up samples,
    and wide samples at the ragged end of a stream.  Each stereo coding (left/side, side/right, mid/side)."""
    rng = np.random.default_rng(1700 + mode)
    files, srcs = [], []

    def add(x, bs=1024, order=12, **kw):
        f, _ = synth.encode(x.astype(np.int32), 16, 44100, synth.options(bs, order, stereo_mode=mode, **kw))
        files.append(bytes(f))
        srcs.append(synth.pack_pcm(x.astype(np.int32), 16))

    n = 1024 * 6 + 517
    t = np.arange(n)
    tone = (9000 * np.sin(t / 7.0)).astype(np.int64)
    for k in range(10):                                   # quiet and correlated, with stretches of opposite full scale
        x = np.stack([tone + rng.integers(-40, 40, n), tone + rng.integers(-40, 40, n)], axis=1)
        for at in rng.integers(0, n - 40, size=6):
            w = int(rng.integers(1, 40))
            x[at:at + w, 0] = 32767 - rng.integers(0, 3, w)
            x[at:at + w, 1] = -32768 + rng.integers(0, 3, w)
        x[n - 3:, 0], x[n - 3:, 1] = -32768, 32767            # ... and at the ragged end
        add(x, order=int(rng.choice([2, 8, 12])))
    x = np.stack([rng.integers(20000, 32768, n), -rng.integers(20000, 32768, n)], axis=1)
    add(x)                                                # wide throughout (side around +-60000)
    add(x, order=0)                                       # ... with fixed predictors
    add(x, bs=4096, order=32)                             # ... and the 32-tap class
    x = np.empty((n, 2), dtype=np.int64)
    x[:, 0], x[:, 1] = 30000, -30000                      # constant subframes, side = 60000
    x[n // 2:, 0] = 29999
    add(x)
    x = (np.stack([rng.integers(16000, 32768, n), -rng.integers(16000, 32768, n)], axis=1) >> 3) << 3
    add(x)                                                # wasted bits below a wide side channel
    x = np.stack([tone, tone + 5], axis=1)
    x[:12, 0], x[:12, 1] = 32767, -32768                  # wide warm-up samples only
    add(x)
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x33)
    for f, pcm, src, si, r in zip(files, pcms, srcs, infos, res):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src and md5hex(src) == si.md5_signature
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples)
        assert np.array_equal(pcm, want)


def seventeen_bit_streams():
    """Streams whose subframes are 17 bits wide once their wasted bits are gone: (file, packed PCM) lists."""
    rng = np.random.default_rng(1717)
    files, srcs = [], []

    def add(x, bps, bs=1024, order=12, **kw):
        f, _ = synth.encode(x.astype(np.int32), bps, 48000, synth.options(bs, order, stereo_mode=0, **kw))
        files.append(bytes(f))
        srcs.append(synth.pack_pcm(x.astype(np.int32), bps))

    for bps, ch, n, bs, order, kw in ((24, 1, 1358, 2048, 12, dict(rice_method=1, force_lpc_order=12)), (24, 1, 1024 * 5 + 77, 1024, 12, {}),
                                      (24, 2, 4096 * 2 + 500, 4096, 32, {}), (20, 5, 1024 * 3 + 9, 1024, 8, {}), (17, 2, 1024 * 4, 1024, 12, {}),
                                      (24, 2, 1024 * 4 + 3, 1024, 0, {})):
        wasted = bps - 17
        t = np.arange(n)
        for kind in ("wide", "quiet with loud stretches", "wide warm-up"):
            if kind == "wide":
                v = (40000 * np.sin(t[:, None] / (9.0 + np.arange(ch)))).astype(np.int64) + rng.integers(-3000, 3000, (n, ch))
            else:
                v = (9000 * np.sin(t[:, None] / (7.0 + np.arange(ch)))).astype(np.int64) + rng.integers(-40, 40, (n, ch))
                if kind == "wide warm-up":
                    v[:12] = 65535 * (1 - 2 * (np.arange(12)[:, None] & 1))
                else:
                    for at in rng.integers(0, n - 40, size=6):
                        w = int(rng.integers(1, 40))
                        v[at:at + w] = rng.choice([-65536, 65535])
            v = np.clip(v, -65536, 65535)
            add(v << wasted, bps, bs=bs, order=order, **kw)
    return files, srcs


def test_seventeen_bit_subframes_of_wider_streams(decoder):
    """Subframes that are 17 bits wide once their wasted bits are gone, in 24- and 20-bit streams (and plain 17-bit
    ones): they decode in the 16-bit pair class, whose packed history holds 16 bits, so a group with a sample that
    needs the 17th is redone sample by sample.  Quiet streams with loud stretches, streams that are wide
    throughout, mono / stereo / 5 channels, warm-up samples at full width, 12 and 32 taps, ragged single frames."""
    files, srcs = seventeen_bit_streams()
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x33)
    for f, pcm, src, si, r in zip(files, pcms, srcs, infos, res):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src and md5hex(src) == si.md5_signature
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples)
        assert np.array_equal(pcm, want)


def test_full_scale_noise_outruns_the_refill_schedule(decoder):
    """White noise at full scale is incompressible: nearly bps bits per code, so the lanes consume their
    bit-reader rings as fast as the refill rounds can fill them and some run dry inside the unrolled loop
    (rolled back and redone).  Quiet channels beside the loud ones keep their results."""
    rng = np.random.default_rng(23)
    files, srcs = [], []
    for ch, bps, order, bs in ((2, 16, 8, 4096), (2, 24, 12, 4096), (8, 24, 32, 4096), (4, 20, 32, 1024), (1, 8, 4, 576)):
        n = bs * 3 + 57
        lim = 1 << (bps - 1)
        x = rng.integers(-lim, lim, size=(n, ch)).astype(np.int32)
        if ch > 2:
            x[:, ch // 2] = rng.integers(-5, 6, size=n)
        f, _ = synth.encode(x, bps, 48000, synth.options(bs, order, stereo_mode=0))
        files.append(bytes(f))
        srcs.append(synth.pack_pcm(x, bps))
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x33)
    for f, p, src, si, r in zip(files, pcms, srcs, infos, res):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src
        assert np.array_equal(p, want)
        assert r.frames_ok == ost.frames_ok == 4


def test_minutes_of_audio_in_one_batch(decoder):
    """A batch of whole minutes rather than a few frames (the other tests keep streams short so that the oracle is
    quick): two minutes of 16-bit stereo in 20 files, 10 s of mono, 1 s of 192 kHz 24-bit 8-channel LPC-32.  Every stream's frame
    count, sample count and MD5 against STREAMINFO; one stream of each shape against the oracle byte by byte; then
    the PCM through the GPU encoder in both stereo modes and back through the GPU decoder."""
    import hashlib
    shapes = [(20, 6 * 44100, 2, 16, 44100, synth.options(4096, 12)), (6, 10 * 44100 // 6, 1, 16, 44100, synth.options(4096, 12)),
              (4, 48000, 8, 24, 192000, synth.options(16384, 32, force_lpc_order=32, max_partition_order=8, force_partition_order=8))]
    files, oracle_checked = [], []
    for first, (nf, ns, ch, bps, rate, o) in enumerate(shapes):
        batch, _ = synth.make_batch(1000 * first, nf, ns, ch, bps, rate, o, nthreads=8)
        oracle_checked.append(len(files))
        files += [bytes(b) for b in batch]
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x5A)
    for i, (f, p, si, r) in enumerate(zip(files, pcms, infos, res)):
        nfr = (si.total_samples + si.max_blocksize - 1) // si.max_blocksize
        assert (r.frames, r.frames_ok, r.samples) == (nfr, nfr, si.total_samples), i
        assert hashlib.md5(bytes(p)).hexdigest() == si.md5_signature, i
    for i in oracle_checked:
        want, ost = oracle_decode(infos[i], split_file(files[i])[1])
        assert np.array_equal(pcms[i], want)
    for mode in (flac_amd.STEREO_EXACT, flac_amd.STEREO_ESTIMATE):
        sel = list(range(0, 20, 5)) + [20, 26]
        packed = [bytes(pcms[i]) for i in sel]
        shp = [(infos[i].total_samples, infos[i].sample_rate, infos[i].channels, infos[i].bits_per_sample) for i in sel]
        from util import gpu_encode_group
        enc, eres = gpu_encode_group(decoder, packed, shp, flac_amd.encode_options(blocksize=4096, max_lpc_order=12, max_partition_order=6,
                                                                                    stereo_mode=mode, flags=flac_amd.ENC_MD5))
        assert all(r.status == 0 for r in eres)
        back, bres, binfos = gpu_batch(decoder, enc)
        for b, want, si in zip(back, packed, binfos):
            assert bytes(b) == want and hashlib.md5(want).hexdigest() == si.md5_signature


def test_device_md5_matches_streaminfo(decoder, testdata):
    """flacgpu_md5_batch_device on the decoded PCM still in HBM: equals hashlib and the STREAMINFO
    signatures the reference's tests assert (flacmeta_test.go:37, :121), for lengths around the 64-byte
    block and 56-byte padding boundaries too."""
    import hashlib
    import torch
    rng = np.random.default_rng(3)
    lens = [0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 128, 1000, 4096, 100003]
    blob = rng.integers(0, 256, size=sum((n + 19) for n in lens) + 64, dtype=np.uint8)
    offs, pos = [], 3                      # unaligned starts as well
    for n in lens:
        offs.append(pos)
        pos += n + 19
    d = torch.from_numpy(blob).cuda()
    got = decoder.md5_batch_device(d, blob.size, offs, lens)
    for o, n, g in zip(offs, lens, got):
        assert g == hashlib.md5(blob[o:o + n].tobytes()).digest(), n
    files = [_read(testdata, MONO), _read(testdata, STEREO)]
    infos = [split_file(f)[0] for f in files]
    pcms, res, _ = gpu_batch(decoder, files)
    cat = np.concatenate([np.zeros(16, np.uint8)] + pcms)
    d = torch.from_numpy(cat).cuda()
    o2 = [16, 16 + pcms[0].size]
    got = decoder.md5_batch_device(d, cat.size, o2, [p.size for p in pcms])
    assert [g.hex() for g in got] == [si.md5_signature for si in infos]
    assert got[0].hex() == "e5ccc967ced6c111530e5c79e33c969e" and got[1].hex() == "6291dbd8dcb7dc480132e4c4ba154a17"


def test_frame_range_shards_of_one_stream(decoder):
    """One long stream cut at SEEKTABLE offsets (flacmeta.go:612-641) into frame ranges that decode as
    independent batch entries into the same output region: the way a single stream is spread over the
    GPUs of a node.  Fixed and variable block size."""
    import torch
    from flac_amd import sharding
    for kw, ch, bps in ((dict(), 2, 16), (dict(variable_blocksize=1, var_sizes=[1024, 4096, 576, 2000]), 2, 16),
                        (dict(), 8, 24)):
        x = synth.pcm(91, 4096 * 23 + 321, ch, bps, 48000, 4096)
        f, ff = synth.encode(x, bps, 48000, synth.options(4096, 12, seek_every_frames=3, **kw))
        f = bytes(f)
        md = flac_amd.parse_metadata(f)
        seek = [b for b in md.blocks[: md.n_blocks] if b.type == 4][0]
        pts = flac_amd.parse_seektable(f[seek.payload_offset: seek.payload_offset + seek.length])
        si, audio = split_file(f)
        shards = sharding.range_shards(pts, len(audio), 5)
        assert len(shards) == 5
        want, ost = oracle_decode(si, audio)
        d_in = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()
        d_out = torch.full((si.pcm_bytes + 64,), 0xAA, dtype=torch.uint8, device="cuda")
        streams = decoder.make_streams([(o, n, 0, si.pcm_bytes, si) for o, n in shards])
        res = decoder.decode_batch_device(d_in, len(audio), d_out, si.pcm_bytes + 64, streams)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy()[: si.pcm_bytes], want)
        assert sum(r.frames_ok for r in res) == sum(r.frames for r in res) == ost.frames_ok
        assert sum(r.samples for r in res) == si.total_samples


def _decode_arena(decoder, files, gap, gap_fill):
    """Streams `gap` bytes apart (a multiple of 16) in an input buffer whose other bytes are gap_fill repeated."""
    import torch
    infos, audios = zip(*(split_file(f) for f in files))
    in_offs, out_offs, pos, opos = [], [], gap, 0
    for si, a in zip(infos, audios):
        in_offs.append(pos)
        pos += (len(a) + 15) // 16 * 16 + gap
        out_offs.append(opos)
        opos += (si.pcm_bytes + 63) // 64 * 64
    host = np.frombuffer((gap_fill * (pos // len(gap_fill) + 1))[:pos], dtype=np.uint8).copy()
    for a, o in zip(audios, in_offs):
        host[o:o + len(a)] = np.frombuffer(a, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.zeros(max(opos, 64), dtype=torch.uint8, device="cuda")
    streams = decoder.make_streams((in_offs[i], len(audios[i]), out_offs[i], infos[i].pcm_bytes, infos[i])
                                   for i in range(len(files)))
    res = decoder.decode_batch_device(d_in, pos, d_out, max(opos, 64), streams)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    return [out[out_offs[i]: out_offs[i] + infos[i].pcm_bytes] for i in range(len(files))], list(res)


def test_crc16_comes_out_of_the_scan(decoder):
    """The frame CRC-16 is accumulated by the scan kernel in its single pass over the input and cut at every sync
    code; only the last span of a stream (it ends at the stream's end, not at a sync code) is read a second time.
    Sync codes whose next three bytes cannot belong to a header of the stream are not cut at (a fill of bare sync codes
    costs nothing).  When a range of the scan (32 KiB of input) holds more cuts than its record has slots (here: the bytes
    between the streams are nothing but plausible header starts), the spans through it are read again instead.  Either way every frame's CRC-16
    must come out as the oracle's does, for spans of one byte of payload up to spans over several ranges."""
    big = [(2, 16, 30000, 1152), (8, 24, 40000, 16384), (2, 16, 4096 * 9 + 5, 4096), (1, 16, 20000, 4608)]
    small = [(1, 16, 40000, 192), (2, 8, 2500, 16)]     # hundreds of frames in one 32 KiB range: more cuts than a record holds
    files = []
    for i, (ch, bps, n, bs) in enumerate(big + small):
        x = synth.pcm(1300 + i, n, ch, bps, 44100, bs)
        f, _ = synth.encode(x, bps, 44100, synth.options(bs, 8))
        files.append(bytes(f))
    expect = [oracle_decode(*split_file(f)) for f in files]
    # bytes that pass for the start of a 16-bit stereo header (block size code 1, sample rate code 9, left/side, number 0)
    hdr_like = b"\xff\xf8\x19\x18\x00"
    for count, gap_fill, overflow in ((len(big), b"\x00", False), (len(big), b"\xff\xf8", False), (len(big), hdr_like, True),
                                      (len(big), b"\xff\xf9\x19\x18\x00\x00", True), (len(files), b"\x00", True)):
        pcms, res = _decode_arena(decoder, files[:count], 4096, gap_fill)
        for p, r, (want, ost) in zip(pcms, res, expect):
            assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples)
            assert ost.frames_ok == ost.frames and np.array_equal(p, want)
        reread = decoder.timing().crc16_reread_spans
        if overflow:
            assert reread > count
        else:
            assert 0 < reread <= count


def test_sync_code_at_the_end_of_a_scan_step(decoder):
    """A frame whose sync code sits in the last bytes before a boundary of the scan (the 16 bytes of a lane, the 1 KiB of a
    wave step, the 4 KiB a wave loads at once, the 32 KiB range of a wave, the 64 KiB tile): the bytes that follow it
    belong to the next lane, step or tile, and the scan reads the header's first bytes across that boundary before
    it accepts the sync code.  Every stream must come out complete, its first frame included."""
    import torch
    specs = [(2, 16, 3000, 1152), (1, 8, 2000, 576), (8, 24, 2000, 1024), (3, 12, 2500, 1000)]
    files = []
    for i, (ch, bps, n, bs) in enumerate(specs):
        x = synth.pcm(1700 + i, n, ch, bps, 44100, bs)
        f, _ = synth.encode(x, bps, 44100, synth.options(bs, 8))
        files.append(bytes(f))
    parts = [split_file(f) for f in files]
    expect = [oracle_decode(si, a) for si, a in parts]
    places = [(b, d) for b in (16, 1024, 4096, 32768, 65536) for d in (1, 2, 3, 4, 5)]
    span = 3 * 65536
    assert all(len(a) < 65536 for _, a in parts)
    host = np.zeros(span * len(places) * len(parts) + 65536, dtype=np.uint8)
    items, out_pos = [], 0
    for pi, (b, d) in enumerate(places):
        for fi, (si, a) in enumerate(parts):
            off = (pi * len(parts) + fi) * span + 65536 + b - d
            host[off: off + len(a)] = np.frombuffer(a, dtype=np.uint8)
            items.append((off, len(a), out_pos, si.pcm_bytes, si, fi))
            out_pos += (si.pcm_bytes + 63) // 64 * 64
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.zeros(out_pos, dtype=torch.uint8, device="cuda")
    res = decoder.decode_batch_device(d_in, host.size, d_out, out_pos, decoder.make_streams(it[:5] for it in items))
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    for (off, n, o, nb, si, fi), r, (b, d) in zip(items, res, (pl for pl in places for _ in parts)):
        want, ost = expect[fi]
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), (b, d, specs[fi])
        assert np.array_equal(out[o: o + nb], want), (b, d, specs[fi])
