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
    srcs = [os.path.join(_HERE, f) for f in ("flac_oracle.c", "flac_enc_oracle.c", "flac_oracle.h", "flac_enc_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
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
        ("first_bad_frame", ctypes.c_uint64),
        ("first_error", ctypes.c_int32),
        ("pad", ctypes.c_uint32),
    ]


class EncOpts(ctypes.Structure):
    """orc_enc_opts: the fields of flacgpu_encode_options the algorithm reads."""
    _fields_ = [
        ("blocksize", ctypes.c_uint32),
        ("max_lpc_order", ctypes.c_uint32),
        ("qlp_precision", ctypes.c_uint32),
        ("min_partition_order", ctypes.c_uint32),
        ("max_partition_order", ctypes.c_uint32),
        ("stereo_mode", ctypes.c_int32),
        ("force_lpc_order", ctypes.c_uint32),
        ("flags", ctypes.c_uint32),
        ("seektable_interval", ctypes.c_uint32),
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
        L.orc_decode_stream_ex.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(StreamInfo),
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64,
                                           ctypes.POINTER(Frame), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                           ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(Stats)]
        L.orc_decode_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(StreamInfo), ctypes.c_uint32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.POINTER(Stats), ctypes.c_int]
        L.orc_enc_window.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
        L.orc_encode_bound.restype = ctypes.c_uint64
        L.orc_encode_bound.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
        L.orc_encode_stream.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                        ctypes.POINTER(EncOpts), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
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


def decode_range(audio, si: StreamInfo, sample_base=0, pcm_cap=None):
    """A frame range of a stream decoded on its own: the sample with index sample_base lands at byte 0.
    Returns (pcm, Stats, list[Frame] with the statuses the decode left)."""
    a = np.ascontiguousarray(np.frombuffer(audio, dtype=np.uint8))
    cap = pcm_bytes(si) if pcm_cap is None else pcm_cap
    pcm = np.zeros(cap, dtype=np.uint8)
    fcap = a.size // 10 + 16
    frames = (Frame * fcap)()
    n, nf = ctypes.c_size_t(), ctypes.c_size_t()
    st = Stats()
    rc = lib().orc_decode_stream_ex(a.ctypes.data, a.size, ctypes.byref(si), pcm.ctypes.data, cap, int(sample_base),
                                    frames, fcap, ctypes.byref(nf), ctypes.byref(n), ctypes.byref(st))
    if rc:
        raise RuntimeError(f"orc_decode_stream_ex: status {rc}")
    return pcm[: n.value], st, [frames[i] for i in range(nf.value)]


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


def enc_options(blocksize=4096, max_lpc_order=12, qlp_precision=0, min_partition_order=0, max_partition_order=6,
                stereo_mode=-1, force_lpc_order=0, flags=0, seektable_interval=0) -> EncOpts:
    return EncOpts(blocksize, max_lpc_order, qlp_precision, min_partition_order, max_partition_order, stereo_mode,
                   force_lpc_order, flags, seektable_interval)


def encode_window(bs: int) -> np.ndarray:
    w = np.empty(bs, dtype=np.float32)
    lib().orc_enc_window(bs, w.ctypes.data)
    return w


def encode_bound(total_samples, channels, bps, blocksize) -> int:
    return int(lib().orc_encode_bound(total_samples, channels, bps, blocksize))


def encode_stream(pcm, total_samples, channels, bps, sample_rate, opts: EncOpts, md5: bytes = None):
    """pcm: packed interleaved little-endian samples (bytes-like).  Returns (file bytes as np.uint8,
    frames, min_framesize, max_framesize)."""
    a = np.ascontiguousarray(np.frombuffer(pcm, dtype=np.uint8))
    assert a.size == total_samples * channels * ((bps + 7) // 8)
    frames = (total_samples + opts.blocksize - 1) // opts.blocksize
    cap = encode_bound(total_samples, channels, bps, opts.blocksize)
    if opts.seektable_interval:
        cap += 4 + 18 * ((frames + opts.seektable_interval - 1) // opts.seektable_interval)
    out = np.zeros(cap, dtype=np.uint8)
    n, nf, mn, mx = ctypes.c_size_t(), ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32()
    m = (ctypes.c_uint8 * 16)(*md5) if md5 is not None else None
    rc = lib().orc_encode_stream(a.ctypes.data, total_samples, channels, bps, sample_rate, ctypes.byref(opts), m,
                                 out.ctypes.data, cap, ctypes.byref(n), ctypes.byref(nf), ctypes.byref(mn),
                                 ctypes.byref(mx))
    if rc:
        raise RuntimeError(f"orc_encode_stream: status {rc}")
    return out[: n.value].copy(), nf.value, mn.value, mx.value
