"""GPU test of flacgpu_decode_batch_device_noalloc: the decode out of the library's static workspace
against flacgpu_decode_batch_device (which allocates its workspace per handle), bit for bit, on two
calls in a row, the second with a smaller input; and its limits."""
import ctypes

import numpy as np
import pytest

from cases import CASES, encode_case
from util import flac_amd, split_file

pytestmark = pytest.mark.gpu


def _layout(files, pad=64):
    infos, audios, in_offs, out_offs = [], [], [], []
    ipos = opos = 0
    for f in files:
        si, a = split_file(f)
        infos.append(si)
        audios.append(a)
        in_offs.append(ipos)
        ipos += (len(a) + pad - 1) // pad * pad
        out_offs.append(opos)
        opos += (si.pcm_bytes + pad - 1) // pad * pad
    host = np.zeros(max(ipos, 16), dtype=np.uint8)
    for a, o in zip(audios, in_offs):
        host[o:o + len(a)] = np.frombuffer(a, dtype=np.uint8)
    return infos, audios, in_offs, out_offs, host, max(opos, 64)


def _both(decoder, files):
    """Decodes `files` with the handle's entry point and with the one that allocates nothing; returns both outputs."""
    import torch
    infos, audios, in_offs, out_offs, host, out_bytes = _layout(files)
    d_in = torch.from_numpy(host).cuda()
    streams = decoder.make_streams((in_offs[i], len(audios[i]), out_offs[i], infos[i].pcm_bytes, infos[i])
                                   for i in range(len(files)))
    outs, ress = [], []
    for which in (0, 1):
        d_out = torch.full((out_bytes,), 0xA5, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        if which == 0:
            res = decoder.decode_batch_device(d_in, host.size, d_out, out_bytes, streams)
        else:
            res = flac_amd.decode_batch_device_noalloc(0, d_in, host.size, d_out, out_bytes, streams)
        torch.cuda.synchronize()
        outs.append(d_out.cpu().numpy())
        ress.append([(r.header_candidates, r.frames, r.frames_ok, r.samples, r.pcm_bytes) for r in res])
    return outs, ress, infos


def test_noalloc_equals_the_handle_entry_point_twice_in_a_row(decoder):
    enc = [encode_case(i, c) for i, c in enumerate(CASES)]
    big = [f for _, f, _ in enc if len(f) < 400_000]                     # every stream shape of the parity suite
    small = [f for name, f, _ in enc if name in ("tiny", "mono8", "stereo16-odd-bs")]
    for files in (big, small):                                           # the second call is the smaller one
        outs, ress, infos = _both(decoder, files)
        assert ress[0] == ress[1]
        assert np.array_equal(outs[0], outs[1])
        assert all(r[2] == r[1] and r[3] == si.total_samples for r, si in zip(ress[1], infos))


def test_noalloc_leaves_device_memory_alone(decoder):
    """What the entry point is named for: after a first call on the device (which only looks its workspace up), no
    call changes the amount of free device memory, whatever the batch."""
    import torch
    enc = [encode_case(i, c) for i, c in enumerate(CASES)]
    big = [f for _, f, _ in enc if len(f) < 400_000]
    small = [f for name, f, _ in enc if name in ("tiny", "mono8", "stereo16-odd-bs")]
    prepared = []
    for files in (small, big, small):
        infos, audios, in_offs, out_offs, host, out_bytes = _layout(files)
        d_in = torch.from_numpy(host).cuda()
        d_out = torch.zeros(out_bytes, dtype=torch.uint8, device="cuda")
        streams = decoder.make_streams((in_offs[i], len(audios[i]), out_offs[i], infos[i].pcm_bytes, infos[i])
                                       for i in range(len(files)))
        prepared.append((d_in, host.size, d_out, out_bytes, streams))
    flac_amd.decode_batch_device_noalloc(0, *prepared[0])               # warm-up: symbol lookup, CRC tables
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for args in prepared[1:]:                                            # a larger batch, then a smaller one
        res = flac_amd.decode_batch_device_noalloc(0, *args)
        torch.cuda.synchronize()
        assert all(r.frames_ok == r.frames and r.frames for r in res)
        assert torch.cuda.mem_get_info()[0] == free0


def test_noalloc_limits_are_refused(decoder):
    import torch
    f = encode_case(0, CASES[0])[1]
    si, a = split_file(f)
    d_in = torch.zeros(flac_amd.NOALLOC_MAX_IN_BYTES + 4096, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(si.pcm_bytes + 64, dtype=torch.uint8, device="cuda")
    streams = decoder.make_streams([(0, len(a), 0, si.pcm_bytes, si)])
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.decode_batch_device_noalloc(0, d_in, flac_amd.NOALLOC_MAX_IN_BYTES + 16, d_out, si.pcm_bytes, streams)
    assert e.value.status == flac_amd.ERR_CAPACITY and "MAX_IN_BYTES" in str(e.value)
    many = decoder.make_streams([(64 * i, 16, 0, 0, si) for i in range(flac_amd.NOALLOC_MAX_STREAMS + 1)])
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.decode_batch_device_noalloc(0, d_in, 1 << 20, d_out, si.pcm_bytes, many)
    assert e.value.status == flac_amd.ERR_CAPACITY and "MAX_STREAMS" in str(e.value)
    # more header candidates than the workspace holds: a buffer made of nothing but valid-looking headers
    si8 = flac_amd.StreamInfo()
    ctypes.memmove(ctypes.byref(si8), ctypes.byref(si), ctypes.sizeof(si8))
    hdr = bytes.fromhex("fff8c90800")
    from util import oracle
    hdr += bytes([oracle.crc8(hdr)])
    si8.channels, si8.bits_per_sample, si8.max_blocksize = 1, 16, 4096
    n = flac_amd.NOALLOC_MAX_CANDIDATES + 4096
    blob = np.frombuffer(hdr * n, dtype=np.uint8)
    d_in[: blob.size].copy_(torch.from_numpy(blob.copy()))
    one = decoder.make_streams([(0, blob.size, 0, 0, si8)])
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.decode_batch_device_noalloc(0, d_in, blob.size, d_out, si.pcm_bytes, one)
    assert e.value.status == flac_amd.ERR_CAPACITY and "MAX_CANDIDATES" in str(e.value)
    # and the entry point still works afterwards
    outs, ress, _ = _both(decoder, [f])
    assert ress[0] == ress[1] and np.array_equal(outs[0], outs[1])
