"""GPU parity tests of the batched encoder (HIP, through the C ABI): the bytes it writes equal the
encoder oracle's, decode (oracle decoder and GPU decoder) to the source PCM, and carry a STREAMINFO
block the product's restatement of MarshalStreaminfoBlock (flacmeta.go:644-717) reads back."""
import hashlib

import numpy as np
import pytest

from enc_cases import ENC_GROUPS, SPECIAL_KINDS, special_pcm
from util import flac_amd, gpu_batch, gpu_encode_group, make_pcm_group, md5hex, oracle, oracle_decode, split_file, synth

pytestmark = pytest.mark.gpu


def _first_difference(a: bytes, b: bytes) -> str:
    n = min(len(a), len(b))
    d = np.nonzero(np.frombuffer(a[:n], dtype=np.uint8) != np.frombuffer(b[:n], dtype=np.uint8))[0]
    return f"lengths {len(a)} / {len(b)}, first differing byte {int(d[0]) if d.size else n}"


def _check_against_oracle(files, res, packed_list, shapes, gopts, oopts):
    for i, (f, r, packed, shape) in enumerate(zip(files, res, packed_list, shapes)):
        n, rate, ch, bps = shape
        assert r.status == 0 and f is not None, f"stream {i}"
        md5 = hashlib.md5(packed).digest()
        want, nf, mn, mx = oracle.encode_stream(packed, n, ch, bps, rate, oopts, md5=md5)
        assert (r.frames, r.min_framesize, r.max_framesize) == (nf, mn, mx), f"stream {i}"
        assert f == bytes(want), f"stream {i} ({ch} ch, {bps} bit, {n} samples): " + _first_difference(f, bytes(want))
        si, audio = split_file(f)
        pcm, st = oracle_decode(si, audio)
        assert bytes(pcm) == packed, f"stream {i}: lossless round trip"
        assert si.md5_signature == md5.hex() == bytes(r.md5).hex()


@pytest.mark.parametrize("group", ENC_GROUPS, ids=[g["name"] for g in ENC_GROUPS])
def test_encode_groups_match_oracle(decoder, group):
    """Mixed batches: every stream's bytes equal the encoder oracle's, and round-trip losslessly."""
    src = make_pcm_group(group)
    packed = [p for _, p, _ in src]
    shapes = [s for _, _, s in src]
    gopts = flac_amd.encode_options(flags=flac_amd.ENC_MD5, **group["opts"])
    files, res = gpu_encode_group(decoder, packed, shapes, gopts)
    _check_against_oracle(files, res, packed, shapes, gopts, oracle.enc_options(**group["opts"]))


def test_encode_at_dword_aligned_offsets_only(decoder):
    """PCM and output regions at offsets that are multiples of 4 and of nothing larger, packed with no gap:
    the loaders and the packer's whole-word copy-out must not depend on wider alignment."""
    for group in ENC_GROUPS[:3]:
        src = make_pcm_group(group)
        packed = [p for _, p, _ in src]
        shapes = [s for _, _, s in src]
        gopts = flac_amd.encode_options(flags=flac_amd.ENC_MD5, **group["opts"])
        files, res = gpu_encode_group(decoder, packed, shapes, gopts, align=4, lead=4)
        _check_against_oracle(files, res, packed, shapes, gopts, oracle.enc_options(**group["opts"]))


def test_encode_special_signals(decoder):
    """Silence, constants, wasted bits, full-scale noise (verbatim), spikes (long Rice codes), ramps."""
    rng = np.random.default_rng(11)
    packed, shapes = [], []
    for kind in SPECIAL_KINDS:
        for ch, bps in ((2, 16), (1, 24), (2, 8)):
            n = 1024 * 3 + 333
            x = special_pcm(kind, n, ch, bps, rng)
            packed.append(synth.pack_pcm(x, bps))
            shapes.append((n, 44100, ch, bps))
    for stereo_mode in (flac_amd.STEREO_EXACT, flac_amd.STEREO_ESTIMATE):
        o = dict(blocksize=1024, max_lpc_order=12, max_partition_order=6, stereo_mode=stereo_mode)
        gopts = flac_amd.encode_options(flags=flac_amd.ENC_MD5, **o)
        files, res = gpu_encode_group(decoder, packed, shapes, gopts)
        _check_against_oracle(files, res, packed, shapes, gopts, oracle.enc_options(**o))


def test_encode_then_gpu_decode_round_trip(decoder):
    """Encoder output through the GPU decoder: PCM and MD5 come back."""
    group = ENC_GROUPS[0]
    src = make_pcm_group(group, seed=9)
    packed = [p for _, p, _ in src]
    shapes = [s for _, _, s in src]
    gopts = flac_amd.encode_options(flags=flac_amd.ENC_MD5, **group["opts"])
    files, res = gpu_encode_group(decoder, packed, shapes, gopts)
    pcms, dres, infos = gpu_batch(decoder, files)
    for p, want, si, r in zip(pcms, packed, infos, dres):
        assert bytes(p) == want
        assert md5hex(want) == si.md5_signature
        assert r.frames == r.frames_ok


def test_encode_capacity_and_caller_md5(decoder):
    """A stream that does not fit reports FLACGPU_ERR_CAPACITY without disturbing its neighbours or the
    bytes beyond its capacity; without FLACGPU_ENC_MD5 the caller's signature is stored."""
    group = ENC_GROUPS[0]
    src = make_pcm_group(group, seed=3)[:3]
    packed = [p for _, p, _ in src]
    shapes = [s for _, _, s in src]
    md5s = [hashlib.md5(p).digest() for p in packed]
    gopts = flac_amd.encode_options(**group["opts"])
    files, res = gpu_encode_group(decoder, packed, shapes, gopts, md5s=md5s)
    assert all(r.status == 0 for r in res)
    small, res2 = gpu_encode_group(decoder, packed, shapes, gopts, cap_scale=0.3, md5s=md5s)
    for f, g, r, r2 in zip(files, small, res, res2):
        assert r2.out_length == r.out_length
        if r.out_length > int(0.3 * flac_amd.encode_bound(1, 1, 1, 16)):
            pass
        if g is not None:
            assert g == f
        else:
            assert r2.status == flac_amd.ERR_CAPACITY
    assert any(g is None for g in small)
    for f, m in zip(files, md5s):
        assert split_file(f)[0].md5_signature == m.hex()


def test_encode_host_buffer_entry_point(decoder):
    """flacgpu_encode_batch (host pointers, PCIe copies inside)."""
    x = synth.pcm(77, 4096 * 2 + 99, 2, 16, 44100, 4096)
    packed = synth.pack_pcm(x, 16)
    o = dict(blocksize=4096, max_lpc_order=12, max_partition_order=6)
    outs, res = decoder.encode_batch([packed], [(x.shape[0], 44100, 2, 16)], flac_amd.encode_options(flags=flac_amd.ENC_MD5, **o))
    want, _, _, _ = oracle.encode_stream(packed, x.shape[0], 2, 16, 44100, oracle.enc_options(**o))
    assert bytes(outs[0]) == bytes(want)


def test_encode_rejects_bad_arguments(decoder):
    import torch
    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    streams = decoder.make_pcm_streams([(0, 100, 0, 2048, 44100, 2, 16)])
    for bad in (dict(blocksize=8), dict(blocksize=32768), dict(max_lpc_order=33), dict(min_partition_order=5, max_partition_order=4),
                dict(max_partition_order=9), dict(qlp_precision=3), dict(stereo_mode=7), dict(stereo_mode=-3)):
        with pytest.raises(flac_amd.FlacError):
            decoder.encode_batch_device(d, 4096, d, 4096, streams, flac_amd.encode_options(**bad))
    for bad_stream in ((0, 100, 0, 2048, 44100, 9, 16), (0, 100, 0, 2048, 44100, 2, 32), (2, 100, 0, 2048, 44100, 2, 16),
                       (0, 100000, 0, 2048, 44100, 2, 16), (0, 100, 0, 8192, 44100, 2, 16), (0, 100, 0, 2048, 0, 2, 16)):
        with pytest.raises(flac_amd.FlacError):
            decoder.encode_batch_device(d, 4096, d, 4096, decoder.make_pcm_streams([bad_stream]), flac_amd.encode_options())


@pytest.mark.parametrize("seed", range(8))
def test_encode_random_groups(decoder, seed):
    """Random options and stream shapes (the CPU suite round-trips the same groups through the oracle)."""
    from test_encode_oracle_cpu import group_pcm, random_group
    rng = np.random.default_rng(1000 + seed)
    o, streams = random_group(rng)
    src = group_pcm(streams, o["blocksize"], rng)
    packed = [p for p, _ in src]
    shapes = [s for _, s in src]
    gopts = flac_amd.encode_options(flags=flac_amd.ENC_MD5, **o)
    files, res = gpu_encode_group(decoder, packed, shapes, gopts)
    _check_against_oracle(files, res, packed, shapes, gopts, oracle.enc_options(**o))
