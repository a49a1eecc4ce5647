"""CPU tests of the test infrastructure itself: encoder -> oracle round trips for every
stream shape, and the oracle against the independently written Python decoder."""
import hashlib

import numpy as np
import pytest

import pydecoder
from cases import CASES, encode_case
from util import make_file, oracle, oracle_decode, split_file, synth, to_oracle_si


@pytest.mark.parametrize("idx", range(len(CASES)), ids=[c["name"] for c in CASES])
def test_encode_oracle_roundtrip(idx):
    """Lossless round trip: decoded PCM == source PCM, and its MD5 == STREAMINFO MD5."""
    name, f, x = encode_case(idx, CASES[idx])
    si, audio = split_file(f)
    pcm, st = oracle_decode(si, audio)
    src = synth.pack_pcm(x, si.bits_per_sample)
    assert pcm.tobytes() == src
    assert hashlib.md5(src).hexdigest() == si.md5_signature
    assert st.frames == st.frames_ok and st.samples == si.total_samples


SMALL = [
    dict(name="lpc-stereo", channels=2, bps=16, n=700, bs=256, max_lpc_order=8),
    dict(name="lpc32-24bit-8ch", channels=8, bps=24, n=600, bs=512, rate=192000, max_lpc_order=32, force_lpc_order=32,
         force_partition_order=2),
    dict(name="fixed-orders", channels=1, bps=16, n=900, bs=192, max_lpc_order=0),
    dict(name="rice5-escape", channels=2, bps=16, n=500, bs=256, rice_method=1, escape_mode=1),
    dict(name="escape-all", channels=2, bps=12, n=400, bs=128, escape_mode=2),
    dict(name="mid-side", channels=2, bps=16, n=500, bs=256, stereo_mode=3),
    dict(name="left-side", channels=2, bps=24, n=500, bs=256, stereo_mode=1),
    dict(name="side-right", channels=2, bps=8, n=500, bs=256, stereo_mode=2),
    dict(name="verbatim", channels=3, bps=20, n=300, bs=100, force_verbatim=1),
    dict(name="variable", channels=2, bps=16, n=1500, bs=512, variable_blocksize=1, var_sizes=[192, 512, 100, 300]),
]


@pytest.mark.parametrize("idx", range(len(SMALL)), ids=[c["name"] for c in SMALL])
def test_oracle_agrees_with_python_decoder(idx):
    """Two decoders written independently from the format description agree bit for bit
    (the only available defence for the features the reference fixtures do not cover)."""
    name, f, x = encode_case(50 + idx, SMALL[idx])
    si, audio = split_file(f)
    want, _ = oracle_decode(si, audio)
    psi, ppcm = pydecoder.decode_file(f)
    assert ppcm == want.tobytes() == synth.pack_pcm(x, si.bits_per_sample)
    assert psi["md5"] == bytes(si.md5)


def test_wasted_bits_roundtrip():
    """Samples whose low bits are all zero exercise the wasted-bits field in both decoders."""
    x = synth.pcm(7, 1000, 2, 16, 44100, 256) & ~0x7
    f, _ = synth.encode(x, 16, 44100, synth.options(256, 8))
    si, audio = split_file(bytes(f))
    want, st = oracle_decode(si, audio)
    assert want.tobytes() == synth.pack_pcm(x, 16)
    assert pydecoder.decode_file(bytes(f))[1] == want.tobytes()


def test_python_decoder_on_reference_fixture(testdata):
    """The Python decoder also reproduces the stereo fixture's STREAMINFO MD5 (flacmeta_test.go:121)."""
    import os
    data = open(os.path.join(testdata, "silence-44-s.flac"), "rb").read()
    si, pcm = pydecoder.decode_file(data)
    assert hashlib.md5(pcm).hexdigest() == "6291dbd8dcb7dc480132e4c4ba154a17"


def test_corrupt_frames_are_flagged_not_fatal():
    """A flipped payload bit fails that frame's CRC-16; the other frames still decode."""
    f, x = make_file(9, 4096 * 6, 2, 16, 44100, 4096)
    si, audio = split_file(f)
    frames, _ = oracle.scan_frames(audio, to_oracle_si(si))
    bad = bytearray(audio)
    bad[frames[2].offset + 40] ^= 0x10
    frames2, st = oracle.scan_frames(bytes(bad), to_oracle_si(si))
    assert [fr.status for fr in frames2] == [0, 0, 1, 0, 0, 0]
    pcm, st = oracle_decode(si, bytes(bad))
    good, _ = oracle_decode(si, audio)
    fb = 4096 * 4
    assert st.frames_ok == 5
    assert np.array_equal(pcm[:2 * fb], good[:2 * fb]) and np.array_equal(pcm[3 * fb:], good[3 * fb:])
    assert not pcm[2 * fb:3 * fb].any()
