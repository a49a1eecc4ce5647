"""ctypes binding of the synthetic-corpus generator (synth/flac_synth.c):
deterministic PCM signals and a scalar CPU FLAC encoder.  Produces inputs for
tests and benchmarks; it is not part of the decode path.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libflacsynth.so")

SEED = 0xF1AC0000


def build(force=False):
    src = os.path.join(_HERE, "flac_synth.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB


class EncOpts(ctypes.Structure):
    _fields_ = [
        ("blocksize", ctypes.c_uint32),
        ("max_lpc_order", ctypes.c_uint32),
        ("qlp_precision", ctypes.c_uint32),
        ("min_partition_order", ctypes.c_uint32),
        ("max_partition_order", ctypes.c_uint32),
        ("force_lpc_order", ctypes.c_int32),
        ("force_partition_order", ctypes.c_int32),
        ("stereo_mode", ctypes.c_int32),
        ("force_verbatim", ctypes.c_int32),
        ("force_fixed_order", ctypes.c_int32),
        ("rice_method", ctypes.c_int32),
        ("escape_mode", ctypes.c_int32),
        ("variable_blocksize", ctypes.c_int32),
        ("var_sizes", ctypes.c_uint32 * 4),
        ("seek_every_frames", ctypes.c_uint32),
        ("no_constant", ctypes.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        L.synth_default_opts.argtypes = [ctypes.POINTER(EncOpts), ctypes.c_uint32, ctypes.c_uint32]
        L.synth_encode_bound.restype = ctypes.c_size_t
        L.synth_encode_bound.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
        L.synth_encode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                   ctypes.POINTER(EncOpts), ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
        L.synth_pcm.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        L.synth_make_batch.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(EncOpts),
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int]
        _lib = L
    return _lib


def options(blocksize=4096, max_lpc_order=12, **kw) -> EncOpts:
    o = EncOpts()
    lib().synth_default_opts(ctypes.byref(o), blocksize, max_lpc_order)
    for k, v in kw.items():
        if k == "var_sizes":
            for i, s in enumerate(v):
                o.var_sizes[i] = s
        else:
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
    return o


def pcm(file_index, nsamples, channels=2, bps=16, sample_rate=44100, blocksize=4096, seed=SEED) -> np.ndarray:
    """Deterministic test signal, int32 [nsamples, channels]."""
    out = np.empty((nsamples, channels), dtype=np.int32)
    lib().synth_pcm(seed, file_index, nsamples, channels, bps, sample_rate, blocksize, out.ctypes.data)
    return out


def encode(samples: np.ndarray, bps=16, sample_rate=44100, opts: EncOpts = None):
    """samples: int32 [nsamples, channels].  Returns (file bytes as np.uint8, offset of frame 0)."""
    s = np.ascontiguousarray(samples, dtype=np.int32)
    if s.ndim == 1:
        s = s[:, None]
    n, ch = s.shape
    if opts is None:
        opts = options()
    cap = lib().synth_encode_bound(n, ch, bps, opts.blocksize)
    out = np.empty(cap, dtype=np.uint8)
    ol, ff = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib().synth_encode(s.ctypes.data, n, ch, bps, sample_rate, ctypes.byref(opts), out.ctypes.data, cap,
                            ctypes.byref(ol), ctypes.byref(ff))
    if rc:
        raise RuntimeError(f"synth_encode: status {rc}")
    return out[: ol.value].copy(), ff.value


def pack_pcm(samples: np.ndarray, bps: int) -> bytes:
    """Interleaved signed little-endian ceil(bps/8)-byte packing (what the STREAMINFO MD5 hashes)."""
    b = (bps + 7) // 8
    s = np.ascontiguousarray(samples, dtype="<i4")
    return s.view(np.uint8).reshape(-1, 4)[:, :b].tobytes()


def make_batch(first_index, nfiles, nsamples, channels=2, bps=16, sample_rate=44100, opts: EncOpts = None,
               nthreads=None, seed=SEED):
    """Synthesise and encode nfiles files.  Returns (list of np.uint8 file arrays, first-frame offsets)."""
    if opts is None:
        opts = options()
    if nthreads is None:
        nthreads = min(16, os.cpu_count() or 1)
    slot = lib().synth_encode_bound(nsamples, channels, bps, opts.blocksize)
    slot = (slot + 63) // 64 * 64
    out = np.empty(nfiles * slot, dtype=np.uint8)
    lens = np.zeros(nfiles, dtype=np.uint64)
    ff = np.zeros(nfiles, dtype=np.uint64)
    rc = lib().synth_make_batch(seed, first_index, nfiles, nsamples, channels, bps, sample_rate,
                                ctypes.byref(opts), out.ctypes.data, slot, lens.ctypes.data, ff.ctypes.data,
                                None, nthreads)
    if rc:
        raise RuntimeError(f"synth_make_batch: status {rc}")
    files = [out[i * slot: i * slot + int(lens[i])] for i in range(nfiles)]
    return files, [int(x) for x in ff]
