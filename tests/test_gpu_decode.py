"""GPU parity tests: libflacgpu (HIP, through the C ABI) against the CPU oracle,
the STREAMINFO MD5 signatures and the seek tables of the reference fixtures."""
import os

import numpy as np
import pytest

from util import flac_amd, gpu_batch, make_file, md5hex, oracle, oracle_decode, split_file, synth, to_oracle_si

pytestmark = pytest.mark.gpu

MONO = "44100-16-mono.flac"
STEREO = "silence-44-s.flac"


def _read(testdata, name):
    return open(os.path.join(testdata, name), "rb").read()


def test_fixtures_md5(decoder, testdata):
    """BASELINE config 2: both reference fixtures, bit-exact against the MD5 the
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


CASES = [
    dict(name="stereo16-lpc12", channels=2, bps=16, n=4096 * 5 + 1234, bs=4096),
    dict(name="stereo16-fixed", channels=2, bps=16, n=20000, bs=4096, max_lpc_order=0),
    dict(name="mono16", channels=1, bps=16, n=30000, bs=4096),
    dict(name="8ch24-lpc32-po8", channels=8, bps=24, n=16384 * 2 + 100, bs=16384, rate=192000, max_lpc_order=32,
         force_lpc_order=32, force_partition_order=8),
    dict(name="stereo24", channels=2, bps=24, n=20000, bs=4608, rate=96000),
    dict(name="verbatim", channels=2, bps=16, n=5000, bs=1152, force_verbatim=1),
    dict(name="escape-all", channels=2, bps=16, n=6000, bs=1024, escape_mode=2),
    dict(name="escape-some", channels=2, bps=16, n=6000, bs=1024, escape_mode=1),
    dict(name="rice5", channels=2, bps=16, n=6000, bs=1024, rice_method=1),
    dict(name="left-right", channels=2, bps=16, n=6000, bs=1024, stereo_mode=0),
    dict(name="left-side", channels=2, bps=16, n=6000, bs=1024, stereo_mode=1),
    dict(name="side-right", channels=2, bps=16, n=6000, bs=1024, stereo_mode=2),
    dict(name="mid-side", channels=2, bps=16, n=6000, bs=1024, stereo_mode=3),
    dict(name="fixed0", channels=2, bps=16, n=5000, bs=576, force_fixed_order=0),
    dict(name="fixed1", channels=2, bps=16, n=5000, bs=576, force_fixed_order=1),
    dict(name="fixed2", channels=2, bps=16, n=5000, bs=576, force_fixed_order=2),
    dict(name="fixed3", channels=2, bps=16, n=5000, bs=576, force_fixed_order=3),
    dict(name="fixed4", channels=2, bps=16, n=5000, bs=576, force_fixed_order=4),
    dict(name="variable-bs", channels=2, bps=16, n=30000, bs=4096, variable_blocksize=1,
         var_sizes=[1024, 4096, 576, 2000]),
    dict(name="3ch12-bs1000", channels=3, bps=12, n=7777, bs=1000),
    dict(name="mono8", channels=1, bps=8, n=5000, bs=256),
    dict(name="5ch20", channels=5, bps=20, n=9000, bs=2048, rate=48000),
    dict(name="stereo32", channels=2, bps=32, n=5000, bs=512, stereo_mode=0),
    dict(name="lpc32-16bit", channels=2, bps=16, n=9000, bs=4096, max_lpc_order=32),
    dict(name="mono24-odd", channels=1, bps=24, n=5001, bs=1001),
    dict(name="tiny", channels=2, bps=16, n=17, bs=4096),
]


def _encode_case(i, c):
    c = dict(c)
    name = c.pop("name")
    n, bs = c.pop("n"), c.pop("bs")
    f, x = make_file(100 + i, n, c.pop("channels"), c.pop("bps"), c.pop("rate", 44100), bs,
                     c.pop("max_lpc_order", 12), **c)
    return name, f, x


def test_synthetic_cases_one_batch(decoder):
    """Every subframe type / Rice mode / channel layout in ONE mixed batch, against
    the oracle and against the source PCM (lossless round trip + STREAMINFO MD5)."""
    enc = [_encode_case(i, c) for i, c in enumerate(CASES)]
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


def test_host_buffer_entry_point(decoder, testdata):
    """flacgpu_decode_batch (host pointers, PCIe copies inside)."""
    pcm, si, res = flac_amd.decode_file(os.path.join(testdata, STEREO), decoder)
    assert md5hex(pcm) == si.md5_signature
    assert res.frames_ok == 36
