"""Helpers shared by the tests: corpus construction and GPU/oracle drivers."""
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import flac_amd  # noqa: E402
import oracle  # noqa: E402
import synth  # noqa: E402


def to_oracle_si(si) -> "oracle.StreamInfo":
    o = oracle.StreamInfo()
    ctypes.memmove(ctypes.byref(o), ctypes.byref(si), ctypes.sizeof(oracle.StreamInfo))
    return o


def split_file(data: bytes):
    """(*Metadata).Read boundary: returns (StreamInfo, bytes from frame 0)."""
    md = flac_amd.parse_metadata(bytes(data))
    assert md.has_streaminfo
    si = flac_amd.StreamInfo()
    ctypes.memmove(ctypes.byref(si), ctypes.byref(md.streaminfo), ctypes.sizeof(si))
    return si, bytes(data)[md.first_frame_offset:]


def oracle_decode(si, audio):
    pcm, st = oracle.decode_stream(audio, to_oracle_si(si))
    full = np.zeros(si.pcm_bytes, dtype=np.uint8)
    full[: pcm.size] = pcm
    return full, st


def gpu_batch(decoder, files, pad=64, out_fill=0, out_pad=64, in_lead=0, out_lead=0, return_raw=False):
    """Decodes whole .flac byte strings with the device-resident entry point.
    pad / out_pad: alignment of every stream's input / output region; in_lead / out_lead: bytes in front
    of the first one (odd values put everything at odd addresses).
    Returns (list of pcm np.uint8 arrays, results, infos) and, with return_raw, the whole output buffer
    and the output offsets."""
    import torch
    infos, audios = [], []
    for f in files:
        si, a = split_file(f)
        infos.append(si)
        audios.append(a)
    in_offs, out_offs = [], []
    pos = in_lead
    for a in audios:
        in_offs.append(pos)
        pos += (len(a) + pad - 1) // pad * pad
    in_bytes = max(pos, 16)
    pos = out_lead
    for si in infos:
        out_offs.append(pos)
        pos += (si.pcm_bytes + out_pad - 1) // out_pad * out_pad
    out_bytes = max(pos + out_lead, 64)
    host = np.zeros(in_bytes, dtype=np.uint8)
    for a, o in zip(audios, in_offs):
        host[o:o + len(a)] = np.frombuffer(a, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.full((out_bytes,), out_fill, dtype=torch.uint8, device="cuda")
    streams = decoder.make_streams((in_offs[i], len(audios[i]), out_offs[i], infos[i].pcm_bytes, infos[i])
                                   for i in range(len(files)))
    res = decoder.decode_batch_device(d_in, in_bytes, d_out, out_bytes, streams)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    pcms = [out[out_offs[i]: out_offs[i] + infos[i].pcm_bytes].copy() for i in range(len(files))]
    if return_raw:
        return pcms, list(res), infos, out, out_offs
    return pcms, list(res), infos


def make_file(idx, nsamples, channels=2, bps=16, rate=44100, blocksize=4096, max_lpc_order=12, **kw):
    x = synth.pcm(idx, nsamples, channels, bps, rate, blocksize)
    o = synth.options(blocksize, max_lpc_order, **kw)
    f, ff = synth.encode(x, bps, rate, o)
    return bytes(f), x


def md5hex(a) -> str:
    return hashlib.md5(bytes(a)).hexdigest()


def make_pcm_group(group, seed=7):
    """Source PCM of one encoder test group: list of (int32 samples [n, ch], packed bytes, (n, rate, ch, bps))."""
    out = []
    for i, (ch, bps, rate, n) in enumerate(group["streams"]):
        x = synth.pcm(seed * 100 + i, n, ch, bps, rate, group["opts"]["blocksize"])
        out.append((x, synth.pack_pcm(x, bps), (n, rate, ch, bps)))
    return out


def gpu_encode_group(engine, packed, shapes, opts, cap_scale=1.0, md5s=None, align=256, lead=0):
    """Encodes packed PCM byte strings with the device-resident entry point.
    Returns (list of .flac byte strings or None where the stream did not fit, results)."""
    import torch
    in_offs, out_offs, caps = [], [], []
    pos = lead
    for p in packed:
        in_offs.append(pos)
        pos += (len(p) + align - 1) // align * align
    pcm_bytes = pos + 256
    pos = lead
    for (n, rate, ch, bps) in shapes:
        cap = int(flac_amd.encode_bound(n, ch, bps, opts.blocksize, opts.seektable_interval) * cap_scale)
        caps.append(cap)
        out_offs.append(pos)
        pos += (cap + align - 1) // align * align
    out_bytes = pos + 256
    host = np.zeros(pcm_bytes, dtype=np.uint8)
    for p, o in zip(packed, in_offs):
        host[o:o + len(p)] = np.frombuffer(p, dtype=np.uint8)
    d_pcm = torch.from_numpy(host).cuda()
    d_out = torch.full((out_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    streams = engine.make_pcm_streams(
        (in_offs[i], shapes[i][0], out_offs[i], caps[i], shapes[i][1], shapes[i][2], shapes[i][3],
         md5s[i] if md5s else None) for i in range(len(packed)))
    res = engine.encode_batch_device(d_pcm, pcm_bytes, d_out, out_bytes, streams, opts)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    files = []
    for i, r in enumerate(res):
        files.append(bytes(out[out_offs[i]: out_offs[i] + int(r.out_length)]) if r.status == 0 else None)
    # nothing outside a stream's capacity may be touched
    for i in range(len(packed)):
        end = out_offs[i] + caps[i]
        nxt = out_offs[i + 1] if i + 1 < len(packed) else out_bytes
        assert np.all(out[end:nxt] == 0x5A), f"stream {i}: bytes beyond its capacity were written"
    return files, list(res)
