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


def gpu_batch(decoder, files, pad=64, out_fill=0):
    """Decodes whole .flac byte strings with the device-resident entry point.
    Returns (list of pcm np.uint8 arrays, results, infos)."""
    import torch
    infos, audios = [], []
    for f in files:
        si, a = split_file(f)
        infos.append(si)
        audios.append(a)
    in_offs, out_offs = [], []
    pos = 0
    for a in audios:
        in_offs.append(pos)
        pos += (len(a) + pad - 1) // pad * pad
    in_bytes = max(pos, 16)
    pos = 0
    for si in infos:
        out_offs.append(pos)
        pos += (si.pcm_bytes + 63) // 64 * 64
    out_bytes = max(pos, 64)
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
    return pcms, list(res), infos


def make_file(idx, nsamples, channels=2, bps=16, rate=44100, blocksize=4096, max_lpc_order=12, **kw):
    x = synth.pcm(idx, nsamples, channels, bps, rate, blocksize)
    o = synth.options(blocksize, max_lpc_order, **kw)
    f, ff = synth.encode(x, bps, rate, o)
    return bytes(f), x


def md5hex(a) -> str:
    return hashlib.md5(bytes(a)).hexdigest()
