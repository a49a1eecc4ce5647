"""CPU tests of the encoder oracle (oracle/flac_enc_oracle.c): what it writes must decode, with the
decoder oracle (pinned by the reference's fixtures) and with the independent Python decoder, to the
source PCM, and its STREAMINFO must read back through the product's restatement of
MarshalStreaminfoBlock (flacmeta.go:644-717)."""
import hashlib

import numpy as np
import pytest

from enc_cases import ENC_GROUPS, SPECIAL_KINDS, special_pcm
from util import flac_amd, make_pcm_group, oracle, oracle_decode, split_file, synth
import pydecoder


def _check_file(f, packed, shape, opts):
    n, rate, ch, bps = shape
    md = flac_amd.parse_metadata(bytes(f))
    nframes = (n + opts.blocksize - 1) // opts.blocksize
    if opts.seektable_interval:
        npts = (nframes + opts.seektable_interval - 1) // opts.seektable_interval
        assert md.first_frame_offset == 46 + 18 * npts and md.n_blocks == 2
        assert (md.blocks[0].last, md.blocks[1].last, md.blocks[1].type, md.blocks[1].length) == (0, 1, 4, 18 * npts)
    else:
        assert md.first_frame_offset == 42 and md.n_blocks == 1 and md.blocks[0].last == 1
    si = md.streaminfo
    assert (si.min_blocksize, si.max_blocksize) == (opts.blocksize, opts.blocksize)
    assert (si.sample_rate, si.channels, si.bits_per_sample, si.total_samples) == (rate, ch, bps, n)
    assert si.md5_signature == hashlib.md5(packed).hexdigest()
    si2, audio = split_file(bytes(f))
    pcm, st = oracle_decode(si2, audio)
    assert bytes(pcm) == packed
    frames, _ = oracle.scan_frames(audio, oracle.parse_streaminfo(bytes(f[8:42])))
    sizes = [fr.length for fr in frames]
    assert len(frames) == (n + opts.blocksize - 1) // opts.blocksize == st.frames_ok
    if sizes:
        assert (min(sizes), max(sizes)) == (si.min_framesize, si.max_framesize)
    assert len(f) == md.first_frame_offset + sum(sizes)
    if opts.seektable_interval:
        # every seek point names a frame: its first sample, its offset from frame 0 and its length in samples
        pts = flac_amd.parse_seektable(bytes(f[46: md.first_frame_offset]))
        by_sample = {fr.first_sample: fr for fr in frames}
        assert [p.sample_number for p in pts] == [k * opts.seektable_interval * opts.blocksize for k in range(len(pts))]
        for p in pts:
            assert (by_sample[p.sample_number].offset, by_sample[p.sample_number].blocksize) == (p.offset, p.frame_samples)


@pytest.mark.parametrize("group", ENC_GROUPS, ids=[g["name"] for g in ENC_GROUPS])
def test_oracle_encoder_round_trip(group):
    opts = oracle.enc_options(**group["opts"])
    for x, packed, shape in make_pcm_group(group):
        n, rate, ch, bps = shape
        f, nf, mn, mx = oracle.encode_stream(packed, n, ch, bps, rate, opts)
        _check_file(f, packed, shape, opts)
        assert len(f) <= oracle.encode_bound(n, ch, bps, opts.blocksize)


@pytest.mark.parametrize("kind", SPECIAL_KINDS)
def test_oracle_encoder_special_signals(kind):
    rng = np.random.default_rng(11)
    for ch, bps, stereo_mode in ((2, 16, -1), (1, 24, -1), (2, 8, -1), (2, 16, -2), (2, 24, -2)):
        opts = oracle.enc_options(blocksize=1024, max_lpc_order=12, max_partition_order=6, stereo_mode=stereo_mode)
        n = 1024 * 3 + 333
        x = special_pcm(kind, n, ch, bps, rng)
        packed = synth.pack_pcm(x, bps)
        f, nf, mn, mx = oracle.encode_stream(packed, n, ch, bps, 44100, opts)
        _check_file(f, packed, (n, 44100, ch, bps), opts)
        if kind in ("silence", "constant"):
            assert len(f) < 42 + 4 * 32
        if kind == "noise":
            assert len(f) <= oracle.encode_bound(n, ch, bps, 1024)


def test_oracle_encoder_python_decoder_agrees():
    """The independent pure-Python decoder reads the encoder oracle's LPC / fixed / mid-side output."""
    x = synth.pcm(42, 2 * 576 + 50, 2, 16, 44100, 576)
    packed = synth.pack_pcm(x, 16)
    f, _, _, _ = oracle.encode_stream(packed, x.shape[0], 2, 16, 44100, oracle.enc_options(576, 8, max_partition_order=3))
    _, got = pydecoder.decode_file(bytes(f))
    assert got == packed


def test_oracle_encoder_compresses_like_the_corpus_encoder():
    """Same predictor family as the corpus generator's encoder: sizes within 1 %."""
    x = synth.pcm(1, 4096 * 6, 2, 16, 44100, 4096)
    packed = synth.pack_pcm(x, 16)
    f, _, _, _ = oracle.encode_stream(packed, x.shape[0], 2, 16, 44100, oracle.enc_options(4096, 12))
    g, _ = synth.encode(x, 16, 44100, synth.options(4096, 12))
    assert abs(len(f) - len(g)) <= 0.01 * len(g)
    assert len(f) < 0.7 * len(packed)


def random_group(rng):
    """A random encode call: options and 1-4 streams of random shape and signal kind."""
    bs = int(rng.choice([16, 192, 576, 1000, 1024, 1152, 2304, 4096, 4608, 5000, 8192, 16384]))
    order = int(rng.choice([0, 1, 4, 8, 12, 13, 20, 32]))
    max_po = int(rng.integers(0, 9))
    opts = dict(blocksize=bs, max_lpc_order=order, max_partition_order=max_po, min_partition_order=int(rng.integers(0, max_po + 1)),
                stereo_mode=int(rng.integers(-2, 4)), qlp_precision=int(rng.choice([0, 0, 5, 9, 15])),
                force_lpc_order=int(order if order and rng.random() < 0.25 else 0),
                seektable_interval=int(rng.choice([0, 0, 1, 5])))
    streams = []
    for _ in range(int(rng.integers(1, 5))):
        ch = int(rng.integers(1, 9))
        bps = int(rng.choice([4, 8, 12, 16, 16, 20, 24, 24]))
        n = int(rng.choice([0, 1, 15, 16, 17, bs - 1, bs, bs + 1, 2 * bs + 7, 3 * bs]))
        n = min(n, 40000)
        kind = str(rng.choice(["synth", "synth"] + SPECIAL_KINDS))
        streams.append((ch, bps, int(rng.choice([8000, 44100, 48000, 96000, 12345])), n, kind))
    return opts, streams


def group_pcm(streams, bs, rng):
    out = []
    for i, (ch, bps, rate, n, kind) in enumerate(streams):
        x = synth.pcm(1000 + i, n, ch, bps, rate, bs) if kind == "synth" or n == 0 else special_pcm(kind, n, ch, bps, rng)
        out.append((synth.pack_pcm(x, bps) if n else b"", (n, rate, ch, bps)))
    return out


@pytest.mark.parametrize("seed", range(12))
def test_oracle_encoder_random_groups(seed):
    rng = np.random.default_rng(1000 + seed)
    o, streams = random_group(rng)
    opts = oracle.enc_options(**o)
    for packed, shape in group_pcm(streams, o["blocksize"], rng):
        n, rate, ch, bps = shape
        f, nf, mn, mx = oracle.encode_stream(packed, n, ch, bps, rate, opts)
        _check_file(f, packed, shape, opts)
