"""ctypes binding of the CPU oracle (oracle/flac_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg.  The product package (flac_amd) never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "flac_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB


class StreamInfo(ctypes.Structure):
    _fields_ = [
        ("min_blocksize", ctypes.c_uint16),
        ("max_blocksize", ctypes.c_uint16),
        ("min_framesize", ctypes.c_uint32),
        ("max_framesize", ctypes.c_uint32),
        ("sample_rate", ctypes.c_uint32),
        ("channels", ctypes.c_uint8),
        ("bits_per_sample", ctypes.c_uint8),
        ("total_samples", ctypes.c_uint64),
        ("md5", ctypes.c_uint8 * 16),
    ]


class Frame(ctypes.Structure):
    _fields_ = [
        ("offset", ctypes.c_uint64),
        ("length", ctypes.c_uint32),
        ("blocksize", ctypes.c_uint32),
        ("first_sample", ctypes.c_uint64),
        ("number", ctypes.c_uint64),
        ("variable", ctypes.c_uint8),
        ("ch_assign", ctypes.c_uint8),
        ("bps", ctypes.c_uint8),
        ("header_len", ctypes.c_uint8),
        ("sample_rate", ctypes.c_uint32),
        ("status", ctypes.c_int32),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("sync_candidates", ctypes.c_uint64),
        ("header_candidates", ctypes.c_uint64),
        ("frames", ctypes.c_uint64),
        ("frames_ok", ctypes.c_uint64),
        ("samples", ctypes.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        L.orc_parse_streaminfo.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(StreamInfo)]
        L.orc_crc8.restype = ctypes.c_uint8
        L.orc_crc8.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.orc_crc16.restype = ctypes.c_uint16
        L.orc_crc16.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.orc_md5.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.orc_scan_frames.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(StreamInfo),
                                      ctypes.POINTER(Frame), ctypes.c_size_t,
                                      ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(Stats)]
        L.orc_decode_stream.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(StreamInfo),
                                        ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(Stats)]
        L.orc_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(StreamInfo), ctypes.c_uint32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(Stats), ctypes.c_int]
        _lib = L
    return _lib


def parse_streaminfo(block34: bytes) -> StreamInfo:
    si = StreamInfo()
    rc = lib().orc_parse_streaminfo(bytes(block34), len(block34), ctypes.byref(si))
    if rc:
        raise ValueError(f"orc_parse_streaminfo: status {rc}")
    return si


def crc8(b: bytes) -> int:
    return lib().orc_crc8(bytes(b), len(b))


def crc16(b: bytes) -> int:
    return lib().orc_crc16(bytes(b), len(b))


def md5(buf) -> bytes:
    a = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8))
    out = (ctypes.c_uint8 * 16)()
    lib().orc_md5(a.ctypes.data, a.size, out)
    return bytes(out)


def scan_frames(audio, si: StreamInfo):
    """audio: bytes-like holding the stream from frame 0.  Returns (list[Frame], Stats)."""
    a = np.ascontiguousarray(np.frombuffer(audio, dtype=np.uint8))
    cap = a.size // 10 + 16
    frames = (Frame * cap)()
    n = ctypes.c_size_t()
    st = Stats()
    rc = lib().orc_scan_frames(a.ctypes.data, a.size, ctypes.byref(si), frames, cap, ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError(f"orc_scan_frames: status {rc}")
    return [frames[i] for i in range(n.value)], st


def pcm_bytes(si: StreamInfo) -> int:
    return int(si.total_samples) * si.channels * ((si.bits_per_sample + 7) // 8)


def decode_stream(audio, si: StreamInfo, pcm_cap=None):
    """Returns (pcm: np.uint8 array of the packed little-endian samples, Stats)."""
    a = np.ascontiguousarray(np.frombuffer(audio, dtype=np.uint8))
    cap = pcm_bytes(si) if pcm_cap is None else pcm_cap
    pcm = np.zeros(cap, dtype=np.uint8)
    n = ctypes.c_size_t()
    st = Stats()
    rc = lib().orc_decode_stream(a.ctypes.data, a.size, ctypes.byref(si), pcm.ctypes.data, cap,
                                 ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError(f"orc_decode_stream: status {rc}")
    return pcm[: n.value], st


def decode_batch(data: np.ndarray, offs, lens, infos, pcm: np.ndarray, pcm_offs, pcm_caps, nthreads=1):
    """Multi-threaded batch decode used for CPU-baseline timing."""
    n = len(offs)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint64)
    pcm_offs = np.ascontiguousarray(pcm_offs, dtype=np.uint64)
    pcm_caps = np.ascontiguousarray(pcm_caps, dtype=np.uint64)
    pcm_lens = np.zeros(n, dtype=np.uint64)
    arr = (StreamInfo * n)(*infos)
    stats = (Stats * n)()
    rc = lib().orc_decode_batch(data.ctypes.data, offs.ctypes.data, lens.ctypes.data, arr, n,
                                pcm.ctypes.data, pcm_offs.ctypes.data, pcm_caps.ctypes.data,
                                pcm_lens.ctypes.data, stats, int(nthreads))
    if rc:
        raise RuntimeError(f"orc_decode_batch: status {rc}")
    return pcm_lens, [stats[i] for i in range(n)]
