"""GPU tests of what the C ABI offers beyond the packed decode: the frame table after a decode (which frame failed
and why), 32-bit and planar output, the MD5 check flag, batches whose stream shapes alternate, and frame ranges of
one stream decoded into regions of their own (one per rank)."""
import ctypes
import os
import sys

import numpy as np
import pytest

import flac_amd
from cases import CASES, encode_case
from util import gpu_batch, make_file, oracle, oracle_decode, split_file, synth, to_oracle_si

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _arena(dec, files, out_caps=None, sample_bases=None, pad=64):
    """Decodes whole files with the device-resident entry point of `dec`; out_caps: bytes of every stream's output
    region (default: the packed PCM size).  Returns (list of raw output regions, results, infos)."""
    import torch
    infos, audios = zip(*(split_file(f) for f in files))
    caps = list(out_caps) if out_caps is not None else [si.pcm_bytes for si in infos]
    in_offs, out_offs, pos, opos = [], [], 0, 0
    for a, c in zip(audios, caps):
        in_offs.append(pos)
        pos += (len(a) + pad - 1) // pad * pad
        out_offs.append(opos)
        opos += (c + pad - 1) // pad * pad
    host = np.zeros(max(pos, 16), dtype=np.uint8)
    for a, o in zip(audios, in_offs):
        host[o:o + len(a)] = np.frombuffer(a, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.full((max(opos, 64),), 0xAA, dtype=torch.uint8, device="cuda")
    streams = dec.make_streams((in_offs[i], len(audios[i]), out_offs[i], caps[i], infos[i],
                                sample_bases[i] if sample_bases else 0) for i in range(len(files)))
    res = dec.decode_batch_device(d_in, host.size, d_out, max(opos, 64), streams)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    return [out[out_offs[i]: out_offs[i] + caps[i]] for i in range(len(files))], list(res), list(infos)


def _samples(si, packed):
    """Packed little-endian PCM -> int32 [n, channels]."""
    b = si.bytes_per_sample
    raw = np.asarray(packed, dtype=np.uint8).reshape(-1, b).astype(np.uint32)
    v = np.zeros(raw.shape[0], dtype=np.uint32)
    for k in range(b):
        v |= raw[:, k] << (8 * k)
    v = v.astype(np.int64)
    v = np.where(v >= 1 << (8 * b - 1), v - (1 << (8 * b)), v)
    return v.astype(np.int32).reshape(-1, si.channels)


SHAPES = ["stereo16-lpc12", "mono16", "stereo24", "mid-side", "left-side", "side-right", "3ch12-bs1000", "5ch20",
          "8ch24-lpc32-po8", "stereo8", "variable-bs", "stereo32", "mono8", "6ch16", "bs17-3ch8", "tiny", "mono24-odd"]


def _shape_files():
    names = {c["name"] for c in CASES}
    return [(c["name"], encode_case(i, c)[1]) for i, c in enumerate(CASES) if c["name"] in SHAPES and c["name"] in names]


def test_frame_table_after_decode(decoder):
    """flacgpu_get_frames after a decode names the frames that failed inside the subframe decoder: the same
    frames as the oracle's, with the same code where only one thing is wrong with a frame (one channel)."""
    from test_gpu_corrupt import _files, _header_and_audio
    rng = np.random.default_rng(21)
    variants, names = [], []
    for name, f in _files():
        head, a, si = _header_and_audio(f)
        frames, _ = oracle.scan_frames(bytes(a), to_oracle_si(si))
        frames = [fr for fr in frames if fr.status == 0]
        for trial in range(2):
            b = bytearray(a)
            for fr in (frames[i] for i in rng.choice(len(frames), size=min(2, len(frames)), replace=False)):
                lo, hi = fr.offset + fr.header_len, fr.offset + fr.length - 2
                p = lo + (int(rng.integers(0, min(24, hi - lo))) if trial == 0 else int(rng.integers(0, hi - lo)))
                b[p] ^= 1 << int(rng.integers(0, 8))
                crc = oracle.crc16(bytes(b[fr.offset: hi]))
                b[hi], b[hi + 1] = crc >> 8, crc & 0xFF
            # and one frame with a damaged CRC-16: a scan-stage status
            fr = frames[int(rng.integers(0, len(frames)))]
            b[fr.offset + fr.length - 1] ^= 0x40
            variants.append(head + bytes(b)); names.append(f"{name}/t{trial}")
    pcms, res, infos = gpu_batch(decoder, variants)
    table = decoder.get_frames()
    assert len(table) == sum(r.frames for r in res)
    pos, bad_total, same_code = 0, 0, 0
    for i, (name, f, r, si) in enumerate(zip(names, variants, res, infos)):
        _, ost, oframes = oracle.decode_range(split_file(f)[1], to_oracle_si(si))
        mine = table[pos: pos + r.frames]
        pos += r.frames
        assert all(fr.stream == i for fr in mine), name
        assert [(fr.offset, fr.length, fr.blocksize, fr.first_sample) for fr in mine] == \
               [(fr.offset, fr.length, fr.blocksize, fr.first_sample) for fr in oframes], name
        gbad = [k for k, fr in enumerate(mine) if fr.status != 0]
        obad = [k for k, fr in enumerate(oframes) if fr.status != 0]
        assert gbad == obad, name
        assert r.frames - r.frames_ok == len(gbad)
        if gbad:
            assert r.first_bad_frame == gbad[0] and r.first_error == mine[gbad[0]].status, name
            assert (ost.first_bad_frame, ost.first_error != 0) == (gbad[0], True), name
        else:
            assert r.first_bad_frame == flac_amd.NO_BAD_FRAME and r.first_error == 0
        for k in gbad:
            bad_total += 1
            # a scan-stage status is the only thing wrong with its frame; inside the subframe decoder the channels
            # of a frame are decoded side by side on the GPU and one after the other in the oracle
            if oframes[k].status == flac_amd.FRAME_BAD_CRC16 or si.channels == 1:
                assert mine[k].status == oframes[k].status, (name, k)
            same_code += mine[k].status == oframes[k].status
    assert bad_total > 30 and same_code > bad_total // 2


def test_s32_output():
    """FLACGPU_OUT_S32: every sample as an int32, whatever the stream's sample size."""
    dec = flac_amd.Decoder(0, output_format=flac_amd.OUT_S32)
    files = _shape_files()
    sis = [split_file(f)[0] for _, f in files]
    outs, res, infos = _arena(dec, [f for _, f in files], out_caps=[int(si.total_samples) * si.channels * 4 for si in sis])
    for (name, f), o, r, si in zip(files, outs, res, infos):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert r.frames_ok == ost.frames_ok == r.frames, name
        got = o.view(np.int32).reshape(-1, si.channels)
        assert np.array_equal(got, _samples(si, want)), name
        assert r.pcm_bytes == int(si.total_samples) * si.channels * 4
    # the host-buffer entry point with the same option
    houts, hres = dec.decode_batch([split_file(f)[1] for _, f in files[:5]], infos[:5])
    for (name, f), o, si in zip(files[:5], houts, infos[:5]):
        want, _ = oracle_decode(si, split_file(f)[1])
        assert np.array_equal(o.view(np.int32).reshape(-1, si.channels), _samples(si, want)), name
    dec.close()


def test_planar_output():
    """FLACGPU_OUT_PLANAR: one plane per channel, out_capacity / channels apart, packed samples in time order."""
    dec = flac_amd.Decoder(0, output_format=flac_amd.OUT_PLANAR)
    files = _shape_files()
    sis = [split_file(f)[0] for _, f in files]
    planes = [(int(si.total_samples) * si.bytes_per_sample + 3) // 4 * 4 + 8 for si in sis]   # room to spare: the stride is what counts
    outs, res, infos = _arena(dec, [f for _, f in files], out_caps=[p * si.channels for p, si in zip(planes, sis)])
    for (name, f), o, r, si, plane in zip(files, outs, res, infos, planes):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert r.frames_ok == ost.frames_ok == r.frames, name
        b = si.bytes_per_sample
        w = np.asarray(want).reshape(-1, si.channels, b)
        n = int(si.total_samples) * b
        for c in range(si.channels):
            assert np.array_equal(o[c * plane: c * plane + n], w[:, c, :].reshape(-1)), (name, c)
            assert (o[c * plane + n: (c + 1) * plane] == 0xAA).all(), (name, c)    # nothing beyond the plane's samples
        assert r.pcm_bytes == n
    dec.close()


def test_verify_md5_flag():
    dec = flac_amd.Decoder(0, verify_md5=True)
    good = [f for _, f in _shape_files()][:6]
    # a wrong signature, an absent one (all zeros: "not computed"), and a stream one of whose frames fails
    f0 = bytearray(good[0]); f0[8 + 18 + 3] ^= 0xFF
    f1 = bytearray(good[1]); f1[8 + 18: 8 + 34] = bytes(16)
    f2 = bytearray(good[2]); f2[len(f2) - 40] ^= 0xFF
    files = good + [bytes(f0), bytes(f1), bytes(f2)]
    _, res, infos = _arena(dec, files)
    assert [r.md5_status for r in res[:6]] == [flac_amd.MD5_MATCH] * 6
    assert res[6].md5_status == flac_amd.MD5_MISMATCH and res[6].frames_ok == res[6].frames
    assert res[7].md5_status == flac_amd.MD5_NOT_CHECKED
    assert res[8].md5_status == flac_amd.MD5_NOT_CHECKED and res[8].frames_ok < res[8].frames
    # the host-buffer entry point carries the flag too
    outs, hres = dec.decode_batch([split_file(f)[1] for f in files[:7]], infos[:7])
    assert [r.md5_status for r in hres] == [flac_amd.MD5_MATCH] * 6 + [flac_amd.MD5_MISMATCH]
    dec.close()


def test_alternating_shapes_share_launches(decoder):
    """4 000 small streams, stereo and mono by turns: one skim and one set of decode launches per distinct shape,
    not per run of neighbours, and about the device time of the same streams sorted by shape."""
    stereo, _ = make_file(301, 4096 * 2 + 77, channels=2)
    mono, _ = make_file(302, 4096 * 2 + 77, channels=1)
    n = 4000
    mixed = [stereo if i % 2 == 0 else mono for i in range(n)]
    by_shape = [stereo] * (n // 2) + [mono] * (n // 2)
    want = {id(stereo): oracle_decode(*split_file(stereo))[0], id(mono): oracle_decode(*split_file(mono))[0]}
    times = {}
    for label, files in (("sorted", by_shape), ("mixed", mixed), ("sorted", by_shape), ("mixed", mixed)):
        pcms, res, _ = gpu_batch(decoder, files)
        t = decoder.timing()
        assert t.shape_groups == 2 and t.launches <= 11 + 2 * 10 + 1, (label, t.launches)
        assert all(r.frames_ok == r.frames == 3 for r in res)
        for k in (0, 1, n // 2, n - 1):
            assert np.array_equal(pcms[k], want[id(files[k])]), (label, k)
        times[label] = min(times.get(label, 1e9), t.total_ms)
    assert times["mixed"] <= 1.5 * times["sorted"] + 0.2, times


def test_ranges_of_one_stream_into_regions_of_their_own(decoder):
    """flacgpu_stream.sample_base: a frame range cut at a seek point writes its first sample at out_offset, so a
    rank that decodes one range needs room for that range only.  Against the oracle's range decode and the whole
    stream; a range given a base beyond its first frame gets FLACGPU_FRAME_RANGE for the frames before it."""
    import torch
    from flac_amd import sharding
    for kw, ch, bps in ((dict(), 2, 16), (dict(variable_blocksize=1, var_sizes=[1024, 4096, 576, 2000]), 2, 16), (dict(), 8, 24)):
        x = synth.pcm(93, 4096 * 19 + 500, ch, bps, 48000, 4096)
        f, _ = synth.encode(x, bps, 48000, synth.options(4096, 12, seek_every_frames=3, **kw))
        f = bytes(f)
        md = flac_amd.parse_metadata(f)
        seek = [b for b in md.blocks[: md.n_blocks] if b.type == 4][0]
        pts = flac_amd.parse_seektable(f[seek.payload_offset: seek.payload_offset + seek.length])
        si, audio = split_file(f)
        whole, _ = oracle_decode(si, audio)
        shards = sharding.range_shards_with_samples(pts, len(audio), si.total_samples, 4)
        assert len(shards) == 4
        fb = ch * si.bytes_per_sample
        d_in = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()
        out_offs, pos = [], 0
        for _, _, _, ns in shards:
            out_offs.append(pos)
            pos += (ns * fb + 255) // 256 * 256
        d_out = torch.full((pos + 64,), 0xAA, dtype=torch.uint8, device="cuda")
        streams = decoder.make_streams([(o, n, out_offs[i], ns * fb, si, base) for i, (o, n, base, ns) in enumerate(shards)])
        res = decoder.decode_batch_device(d_in, len(audio), d_out, pos + 64, streams)
        torch.cuda.synchronize()
        out = d_out.cpu().numpy()
        for i, (o, n, base, ns) in enumerate(shards):
            got = out[out_offs[i]: out_offs[i] + ns * fb]
            assert np.array_equal(got, whole[base * fb: (base + ns) * fb]), i
            opcm, ost, _ = oracle.decode_range(audio[o: o + n], to_oracle_si(si), sample_base=base, pcm_cap=ns * fb)
            assert np.array_equal(got, opcm) and (res[i].frames, res[i].frames_ok, res[i].samples) == (ost.frames, ost.frames_ok, ns)
            assert res[i].pcm_bytes == ns * fb
        # a base inside the range: the frames in front of it are out of range, on both sides
        o, n, base, ns = shards[1]
        late = base + 4096 + 1
        streams = decoder.make_streams([(o, n, 0, (ns - 4097) * fb, si, late)])
        res = decoder.decode_batch_device(d_in, len(audio), d_out, pos + 64, streams)
        _, ost, oframes = oracle.decode_range(audio[o: o + n], to_oracle_si(si), sample_base=late, pcm_cap=(ns - 4097) * fb)
        gframes = decoder.get_frames()
        assert [fr.status for fr in gframes] == [fr.status for fr in oframes]
        assert res[0].first_error == flac_amd.FRAME_RANGE and res[0].first_bad_frame == 0 and res[0].frames_ok == ost.frames_ok


def test_two_ranks_share_one_stream_on_one_gpu(tmp_path):
    """The frame-range path at world size 2 with both ranks on this GPU (rows and the chained MD5 over gloo)."""
    from test_sharding_gloo import _run, check_range_run
    check_range_run(_run(2, str(tmp_path / "g2.json"), 29621, "_range_worker.py", "gpu"), 2)


def test_argument_errors_of_the_new_options(decoder):
    import ctypes
    import torch
    # an output format the library does not know
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.Decoder(0, output_format=7)
    assert e.value.status == 1
    # a frame array that is too small: the status says so and the count is what is needed
    f = [fl for name, fl in _shape_files() if name == "stereo16-lpc12"][0]
    _, res, _ = gpu_batch(decoder, [f])
    lib, h = flac_amd.lib(), decoder._h
    n = ctypes.c_uint64()
    few = (flac_amd.Frame * 2)()
    assert lib.flacgpu_get_frames(h, few, 2, ctypes.byref(n)) == flac_amd.ERR_CAPACITY and n.value == res[0].frames > 2
    assert [fr.status for fr in few] == [0, 0] and few[1].offset > few[0].offset          # what fits is filled
    assert lib.flacgpu_get_frames(h, None, 0, ctypes.byref(n)) == 0 and n.value == res[0].frames
    # a sample_base beyond the stream
    si, audio = split_file(f)
    d_in = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()
    d_out = torch.zeros(si.pcm_bytes + 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(flac_amd.FlacError) as e:
        decoder.decode_batch_device(d_in, len(audio), d_out, si.pcm_bytes, decoder.make_streams([(0, len(audio), 0, si.pcm_bytes, si, int(si.total_samples) + 1)]))
    assert e.value.status == 1 and "sample_base" in str(e.value)
    assert flac_amd.lib().flacgpu_api_version() == flac_amd.API_VERSION == 2
