"""flac_amd — Python binding (ctypes) of libflacgpu, the MI355X FLAC frame engine.

The product is the shared library (flac_amd/lib/libflacgpu.so, C ABI in
include/flacgpu.h).  This module only marshals arguments: every call goes
through the C ABI, and importing the module fails loudly when the library has
not been built (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C flac_amd/csrc`).  There is no CPU fallback for the frame path.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLACGPU_LIB") or os.path.join(_HERE, "lib", "libflacgpu.so")

OK = 0
FRAME_OK, FRAME_BAD_CRC16, FRAME_BAD_SUBFRAME, FRAME_TRUNCATED, FRAME_LENGTH, FRAME_RANGE, FRAME_UNSUPPORTED, \
    FRAME_SKIPPED = range(8)
FLAG_TIMING = 1
FLAG_VERIFY_MD5 = 2
OUT_PACKED, OUT_S32, OUT_PLANAR = 0, 1, 2      # flacgpu_options.output_format
MD5_NOT_CHECKED, MD5_MATCH, MD5_MISMATCH = 0, 1, 2
API_VERSION = 2
NO_BAD_FRAME = 0xFFFFFFFFFFFFFFFF
ENC_MD5 = 1
STEREO_EXACT = -1      # flacgpu_encode_options.stereo_mode: all four candidate channels planned, cheapest pair
STEREO_ESTIMATE = -2   # the pair picked first from short predictors, its two channels planned
ERR_CAPACITY = 8
MAX_BLOCKS = 128

# MetadataBlockType values as the reference numbers them (flacmeta.go:28-37).
BLOCK_NAMES = {1: "STREAMINFO", 2: "PADDING", 3: "APPLICATION", 4: "SEEKTABLE", 5: "VORBIS_COMMENT",
               6: "CUESHEET", 7: "PICTURE"}


class StreamInfo(ctypes.Structure):
    """StreaminfoBlock, flacmeta.go:263-273."""
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

    @property
    def md5_signature(self) -> str:
        return bytes(self.md5).hex()

    @property
    def bytes_per_sample(self) -> int:
        return (self.bits_per_sample + 7) // 8

    @property
    def pcm_bytes(self) -> int:
        return int(self.total_samples) * self.channels * self.bytes_per_sample


class SeekPoint(ctypes.Structure):
    """SeekpointBlock, flacmeta.go:256-260."""
    _fields_ = [("sample_number", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("frame_samples", ctypes.c_uint16)]


class Block(ctypes.Structure):
    """MetadataBlockHeader, flacmeta.go:235-240, plus the payload position."""
    _fields_ = [("type", ctypes.c_uint32), ("length", ctypes.c_uint32), ("last", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32), ("payload_offset", ctypes.c_uint64)]


class Metadata(ctypes.Structure):
    _fields_ = [("streaminfo", StreamInfo), ("has_streaminfo", ctypes.c_uint32), ("n_blocks", ctypes.c_uint32),
                ("blocks", Block * MAX_BLOCKS), ("first_frame_offset", ctypes.c_uint64)]


class VorbisCommentHead(ctypes.Structure):
    _fields_ = [("vendor_offset", ctypes.c_uint32), ("vendor_length", ctypes.c_uint32), ("total_comments", ctypes.c_uint32)]


class PictureHead(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in ("picture_type", "mime_offset", "mime_length", "description_offset",
                                               "description_length", "width", "height", "color_depth", "num_colors",
                                               "length", "blob_offset", "blob_length")]


class ApplicationHead(ctypes.Structure):
    _fields_ = [("id", ctypes.c_uint32), ("data_offset", ctypes.c_uint32), ("data_length", ctypes.c_uint32)]


class CuesheetIndex(ctypes.Structure):
    _fields_ = [("sample_offset", ctypes.c_uint64), ("index_point", ctypes.c_uint8)]


class CuesheetTrack(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("number", ctypes.c_uint8), ("isrc", ctypes.c_char * 12),
                ("type", ctypes.c_uint8), ("pre_emphasis", ctypes.c_uint8), ("index_points", ctypes.c_uint8),
                ("first_index", ctypes.c_uint32)]


class CuesheetHead(ctypes.Structure):
    _fields_ = [("media_catalog_number", ctypes.c_uint8 * 128), ("leadin_samples", ctypes.c_uint64),
                ("is_compact_disc", ctypes.c_uint8), ("total_tracks", ctypes.c_uint8)]


class Options(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("flags", ctypes.c_uint32),
                ("output_format", ctypes.c_uint32)]


class Stream(ctypes.Structure):
    _fields_ = [("in_offset", ctypes.c_uint64), ("in_length", ctypes.c_uint64), ("out_offset", ctypes.c_uint64),
                ("out_capacity", ctypes.c_uint64), ("sample_base", ctypes.c_uint64), ("info", StreamInfo)]


class StreamResult(ctypes.Structure):
    _fields_ = [("header_candidates", ctypes.c_uint64),
                ("frames", ctypes.c_uint64), ("frames_ok", ctypes.c_uint64), ("samples", ctypes.c_uint64),
                ("pcm_bytes", ctypes.c_uint64), ("first_bad_frame", ctypes.c_uint64), ("first_error", ctypes.c_int32),
                ("md5_status", ctypes.c_uint32)]


class Frame(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("first_sample", ctypes.c_uint64), ("number", ctypes.c_uint64),
                ("length", ctypes.c_uint32), ("blocksize", ctypes.c_uint32), ("stream", ctypes.c_uint32),
                ("status", ctypes.c_int32), ("variable", ctypes.c_uint8), ("ch_assign", ctypes.c_uint8),
                ("bps", ctypes.c_uint8), ("header_len", ctypes.c_uint8), ("reserved", ctypes.c_uint32)]


class Timing(ctypes.Structure):
    _fields_ = [("scan_ms", ctypes.c_float), ("compact_ms", ctypes.c_float), ("crc16_ms", ctypes.c_float),
                ("resolve_ms", ctypes.c_float), ("skim_ms", ctypes.c_float), ("decode_ms", ctypes.c_float),
                ("results_ms", ctypes.c_float), ("total_ms", ctypes.c_float), ("candidates", ctypes.c_uint64),
                ("in_bytes", ctypes.c_uint64), ("out_bytes", ctypes.c_uint64), ("decode_classes", ctypes.c_uint32),
                ("crc16_reread_spans", ctypes.c_uint32), ("launches", ctypes.c_uint32), ("shape_groups", ctypes.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class EncodeOptions(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("blocksize", ctypes.c_uint32), ("max_lpc_order", ctypes.c_uint32),
                ("qlp_precision", ctypes.c_uint32), ("min_partition_order", ctypes.c_uint32),
                ("max_partition_order", ctypes.c_uint32), ("stereo_mode", ctypes.c_int32),
                ("force_lpc_order", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("seektable_interval", ctypes.c_uint32)]


def encode_options(blocksize=4096, max_lpc_order=12, qlp_precision=0, min_partition_order=0, max_partition_order=6,
                   stereo_mode=-1, force_lpc_order=0, flags=0, seektable_interval=0) -> EncodeOptions:
    return EncodeOptions(ctypes.sizeof(EncodeOptions), blocksize, max_lpc_order, qlp_precision, min_partition_order,
                         max_partition_order, stereo_mode, force_lpc_order, flags, seektable_interval)


class PcmStream(ctypes.Structure):
    _fields_ = [("in_offset", ctypes.c_uint64), ("total_samples", ctypes.c_uint64), ("out_offset", ctypes.c_uint64),
                ("out_capacity", ctypes.c_uint64), ("sample_rate", ctypes.c_uint32), ("channels", ctypes.c_uint8),
                ("bits_per_sample", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 2), ("md5", ctypes.c_uint8 * 16)]


class EncodeResult(ctypes.Structure):
    _fields_ = [("out_length", ctypes.c_uint64), ("frames", ctypes.c_uint64), ("min_framesize", ctypes.c_uint32),
                ("max_framesize", ctypes.c_uint32), ("status", ctypes.c_int32), ("reserved", ctypes.c_uint32),
                ("md5", ctypes.c_uint8 * 16)]


class EncodeTiming(ctypes.Structure):
    _fields_ = [("stats_ms", ctypes.c_float), ("design_ms", ctypes.c_float), ("choose_ms", ctypes.c_float),
                ("layout_ms", ctypes.c_float), ("pack_ms", ctypes.c_float),
                ("crc_ms", ctypes.c_float), ("md5_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("frames", ctypes.c_uint64), ("in_bytes", ctypes.c_uint64), ("out_bytes", ctypes.c_uint64),
                ("pick_ms", ctypes.c_float), ("reserved", ctypes.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# Every symbol include/flacgpu.h declares.
EXPORTS = [
    "flacgpu_parse_block_header", "flacgpu_parse_streaminfo", "flacgpu_parse_seektable", "flacgpu_parse_metadata",
    "flacgpu_status_string", "flacgpu_create", "flacgpu_destroy", "flacgpu_last_error", "flacgpu_scan_frames",
    "flacgpu_decode_batch_device", "flacgpu_decode_batch", "flacgpu_get_timing",
    "flacgpu_encode_bound", "flacgpu_encode_batch_device", "flacgpu_encode_batch", "flacgpu_get_encode_timing",
    "flacgpu_md5_batch_device", "flacgpu_seektable_bytes",
    "flacgpu_parse_vorbis_comment", "flacgpu_parse_picture", "flacgpu_picture_type_string", "flacgpu_parse_application",
    "flacgpu_parse_cuesheet", "flacgpu_skip_id3v2",
    "flacgpu_decode_batch_device_noalloc", "flacgpu_noalloc_last_error",
    "flacgpu_get_frames", "flacgpu_api_version",
]

_lib = None


class FlacError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().flacgpu_status_string(status).decode()
        super().__init__(f"flacgpu status {status}: {msg}" + (f" ({detail})" if detail else ""))


def lib():
    """Loads libflacgpu.so; raises if it is missing (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make -C flac_amd/csrc` (needs hipcc)")
        # PyTorch wheels bundle their own HIP runtime (soname libamdhip64.so.7, loaded as
        # "libamdhip64.so").  Two HIP runtimes in one process cannot both own the GPU, so when
        # torch is installed it is imported first and libflacgpu binds to the runtime torch loaded.
        if "torch" not in sys.modules and not os.environ.get("FLAC_AMD_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        vp, u64, u32, sz = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_size_t
        L.flacgpu_parse_block_header.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(Block)]
        L.flacgpu_parse_streaminfo.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(StreamInfo)]
        L.flacgpu_parse_seektable.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(SeekPoint), sz, ctypes.POINTER(sz)]
        L.flacgpu_parse_metadata.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(Metadata)]
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.flacgpu_parse_vorbis_comment.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(VorbisCommentHead), u32p, u32p, sz,
                                                   ctypes.POINTER(sz)]
        L.flacgpu_parse_picture.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(PictureHead)]
        L.flacgpu_picture_type_string.restype = ctypes.c_char_p
        L.flacgpu_picture_type_string.argtypes = [ctypes.c_uint32]
        L.flacgpu_parse_application.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(ApplicationHead)]
        L.flacgpu_parse_cuesheet.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(CuesheetHead), ctypes.POINTER(CuesheetTrack), sz,
                                             ctypes.POINTER(CuesheetIndex), sz, ctypes.POINTER(sz)]
        L.flacgpu_skip_id3v2.argtypes = [ctypes.c_char_p, sz, ctypes.POINTER(sz)]
        L.flacgpu_status_string.restype = ctypes.c_char_p
        L.flacgpu_status_string.argtypes = [ctypes.c_int]
        L.flacgpu_create.argtypes = [ctypes.POINTER(Options), ctypes.POINTER(vp)]
        L.flacgpu_destroy.argtypes = [vp]
        L.flacgpu_destroy.restype = None
        L.flacgpu_last_error.restype = ctypes.c_char_p
        L.flacgpu_last_error.argtypes = [vp]
        L.flacgpu_scan_frames.argtypes = [vp, vp, u64, ctypes.POINTER(Stream), u32, ctypes.POINTER(Frame), u64,
                                          ctypes.POINTER(u64), ctypes.POINTER(StreamResult), vp]
        L.flacgpu_decode_batch_device.argtypes = [vp, vp, u64, vp, u64, ctypes.POINTER(Stream), u32,
                                                  ctypes.POINTER(StreamResult), vp]
        L.flacgpu_get_frames.argtypes = [vp, ctypes.POINTER(Frame), u64, ctypes.POINTER(u64)]
        L.flacgpu_api_version.argtypes = []
        if L.flacgpu_api_version() != API_VERSION:
            raise ImportError(f"{LIB_PATH} has API version {L.flacgpu_api_version()}, this binding wants {API_VERSION}: rebuild it")
        L.flacgpu_decode_batch_device_noalloc.argtypes = [ctypes.c_int, vp, u64, vp, u64, ctypes.POINTER(Stream), u32,
                                                          ctypes.POINTER(StreamResult), vp]
        L.flacgpu_noalloc_last_error.restype = ctypes.c_char_p
        L.flacgpu_noalloc_last_error.argtypes = [ctypes.c_int]
        L.flacgpu_decode_batch.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(StreamInfo),
                                           u32, ctypes.POINTER(vp), ctypes.POINTER(u64),
                                           ctypes.POINTER(StreamResult)]
        L.flacgpu_get_timing.argtypes = [vp, ctypes.POINTER(Timing)]
        L.flacgpu_encode_bound.restype = u64
        L.flacgpu_encode_bound.argtypes = [u64, u32, u32, u32]
        L.flacgpu_seektable_bytes.restype = u64
        L.flacgpu_seektable_bytes.argtypes = [u64, u32, u32]
        L.flacgpu_encode_batch_device.argtypes = [vp, vp, u64, vp, u64, ctypes.POINTER(PcmStream), u32,
                                                  ctypes.POINTER(EncodeOptions), ctypes.POINTER(EncodeResult), vp]
        L.flacgpu_encode_batch.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(PcmStream), u32,
                                           ctypes.POINTER(EncodeOptions), ctypes.POINTER(vp), ctypes.POINTER(u64),
                                           ctypes.POINTER(EncodeResult)]
        L.flacgpu_get_encode_timing.argtypes = [vp, ctypes.POINTER(EncodeTiming)]
        L.flacgpu_md5_batch_device.argtypes = [vp, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64), u32,
                                               ctypes.c_void_p, vp]
        _lib = L
    return _lib


# ------------------------------------------------------------------ metadata

def parse_block_header(b: bytes) -> Block:
    """MarshalMetadataBlockHeader, flacmeta.go:519-551."""
    out = Block()
    rc = lib().flacgpu_parse_block_header(bytes(b), len(b), ctypes.byref(out))
    if rc:
        raise FlacError(rc)
    return out


def parse_streaminfo(b: bytes) -> StreamInfo:
    """MarshalStreaminfoBlock, flacmeta.go:644-717."""
    out = StreamInfo()
    rc = lib().flacgpu_parse_streaminfo(bytes(b), len(b), ctypes.byref(out))
    if rc:
        raise FlacError(rc)
    return out


def parse_seektable(b: bytes):
    """MarshalSeekpointBlock, flacmeta.go:612-641."""
    n = ctypes.c_size_t()
    rc = lib().flacgpu_parse_seektable(bytes(b), len(b), None, 0, ctypes.byref(n))
    if rc:
        raise FlacError(rc)
    pts = (SeekPoint * max(1, n.value))()
    rc = lib().flacgpu_parse_seektable(bytes(b), len(b), pts, n.value, ctypes.byref(n))
    if rc:
        raise FlacError(rc)
    return [pts[i] for i in range(n.value)]


def parse_vorbis_comment(b: bytes) -> dict:
    """MarshalVorbisCommentBlock, flacmeta.go:720-746 -> {Vendor, TotalComments, Comments}."""
    b = bytes(b)
    head, n = VorbisCommentHead(), ctypes.c_size_t()
    rc = lib().flacgpu_parse_vorbis_comment(b, len(b), ctypes.byref(head), None, None, 0, ctypes.byref(n))
    if rc:
        raise FlacError(rc)
    offs, lens = (ctypes.c_uint32 * max(1, n.value))(), (ctypes.c_uint32 * max(1, n.value))()
    rc = lib().flacgpu_parse_vorbis_comment(b, len(b), ctypes.byref(head), offs, lens, n.value, ctypes.byref(n))
    if rc:
        raise FlacError(rc)
    return {"Vendor": b[head.vendor_offset: head.vendor_offset + head.vendor_length].decode("utf-8", "replace"),
            "TotalComments": head.total_comments,
            "Comments": [b[offs[i]: offs[i] + lens[i]].decode("utf-8", "replace") for i in range(n.value)]}


def parse_picture(b: bytes) -> dict:
    """MarshalPictureBlock, flacmeta.go:554-604 -> the fields of PictureBlock."""
    b = bytes(b)
    p = PictureHead()
    rc = lib().flacgpu_parse_picture(b, len(b), ctypes.byref(p))
    if rc:
        raise FlacError(rc)
    return {"PictureType": lib().flacgpu_picture_type_string(p.picture_type).decode(),
            "MimeType": b[p.mime_offset: p.mime_offset + p.mime_length].decode("ascii", "replace"),
            "Description": b[p.description_offset: p.description_offset + p.description_length].decode("utf-8", "replace"),
            "Width": p.width, "Height": p.height, "ColorDepth": p.color_depth, "NumColors": p.num_colors,
            "Length": p.length, "PictureBlob": b[p.blob_offset: p.blob_offset + p.blob_length]}


def parse_application(b: bytes) -> dict:
    """MarshalApplicationBlock, flacmeta.go:349-366 (the data starts at byte 32, as in the reference)."""
    b = bytes(b)
    a = ApplicationHead()
    rc = lib().flacgpu_parse_application(b, len(b), ctypes.byref(a))
    if rc:
        raise FlacError(rc)
    return {"Id": a.id, "Data": b[a.data_offset: a.data_offset + a.data_length]}


def parse_cuesheet(b: bytes) -> dict:
    """MarshalCuesheetBlock, flacmeta.go:369-432 -> the fields of CuesheetBlock with its tracks and indexes."""
    b = bytes(b)
    head, ni = CuesheetHead(), ctypes.c_size_t()
    rc = lib().flacgpu_parse_cuesheet(b, len(b), ctypes.byref(head), None, 0, None, 0, ctypes.byref(ni))
    if rc:
        raise FlacError(rc)
    tracks = (CuesheetTrack * max(1, head.total_tracks))()
    idx = (CuesheetIndex * max(1, ni.value))()
    rc = lib().flacgpu_parse_cuesheet(b, len(b), ctypes.byref(head), tracks, head.total_tracks, idx, ni.value, ctypes.byref(ni))
    if rc:
        raise FlacError(rc)
    out = {"MediaCatalogNumber": bytes(head.media_catalog_number), "LeadinSamples": head.leadin_samples,
           "IsCompactDisc": bool(head.is_compact_disc), "TotalTracks": head.total_tracks, "Tracks": []}
    for i in range(head.total_tracks):
        t = tracks[i]
        out["Tracks"].append({"Offset": t.offset, "Number": t.number,
                              "ISRC": ctypes.string_at(ctypes.addressof(t) + CuesheetTrack.isrc.offset, 12),   # NULs kept
                              "Type": t.type, "PreEmphasis": bool(t.pre_emphasis), "IndexPoints": t.index_points,
                              "Indexes": [(idx[t.first_index + j].sample_offset, idx[t.first_index + j].index_point)
                                          for j in range(t.index_points)]})
    return out


def parse_metadata(buf: bytes) -> Metadata:
    """(*Metadata).Read, flacmeta.go:749-850: returns the block table, STREAMINFO
    and first_frame_offset (the reader position Read leaves behind)."""
    out = Metadata()
    rc = lib().flacgpu_parse_metadata(bytes(buf), len(buf), ctypes.byref(out))
    if rc:
        raise FlacError(rc)
    return out


def skip_id3v2(buf: bytes) -> int:
    """Bytes of ID3v2 tag(s) in front of the "fLaC" marker (0 when there is none).  Read itself stays as strict
    as the reference (flacmeta.go:754-761): pass buf[skip_id3v2(buf):] to parse_metadata to accept tagged files."""
    off = ctypes.c_size_t()
    rc = lib().flacgpu_skip_id3v2(bytes(buf), len(buf), ctypes.byref(off))
    if rc:
        raise FlacError(rc)
    return int(off.value)


# -------------------------------------------------------------- frame engine

def _ptr(t):
    """Device pointer of a torch tensor or a raw int."""
    return t if isinstance(t, int) else t.data_ptr()


NOALLOC_MAX_IN_BYTES, NOALLOC_MAX_STREAMS, NOALLOC_MAX_CANDIDATES = 4 << 20, 256, 32768


def decode_batch_device_noalloc(device, d_in, in_bytes, d_out, out_bytes, streams, hip_stream=0, results=None):
    """flacgpu_decode_batch_device_noalloc: the device-resident decode out of the library's static workspace
    (no handle, no device allocation); small batches only, see the limits above."""
    n = len(streams)
    res = results if results is not None else (StreamResult * n)()
    rc = lib().flacgpu_decode_batch_device_noalloc(int(device), _ptr(d_in), in_bytes, _ptr(d_out), out_bytes, streams, n,
                                                   res, hip_stream or None)
    if rc:
        raise FlacError(rc, lib().flacgpu_noalloc_last_error(int(device)).decode())
    return res


class Decoder:
    """One handle = one GPU's workspace (flacgpu_create / flacgpu_destroy)."""

    def __init__(self, device=0, timing=False, output_format=OUT_PACKED, verify_md5=False):
        self._h = ctypes.c_void_p()
        o = Options(ctypes.sizeof(Options), int(device), (FLAG_TIMING if timing else 0) | (FLAG_VERIFY_MD5 if verify_md5 else 0),
                    int(output_format))
        rc = lib().flacgpu_create(ctypes.byref(o), ctypes.byref(self._h))
        if rc:
            raise FlacError(rc, "flacgpu_create")
        self.device = int(device)
        self.output_format = int(output_format)

    def close(self):
        if self._h:
            lib().flacgpu_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise FlacError(rc, lib().flacgpu_last_error(self._h).decode())

    @staticmethod
    def make_streams(entries):
        """entries: iterable of (in_offset, in_length, out_offset, out_capacity, StreamInfo[, sample_base])."""
        entries = list(entries)
        arr = (Stream * len(entries))()
        for i, e in enumerate(entries):
            arr[i].in_offset, arr[i].in_length, arr[i].out_offset, arr[i].out_capacity = e[:4]
            arr[i].sample_base = e[5] if len(e) > 5 else 0
            ctypes.memmove(ctypes.byref(arr[i].info), ctypes.byref(e[4]), ctypes.sizeof(StreamInfo))
        return arr

    def get_frames(self):
        """flacgpu_get_frames: the frame table of this handle's last scan or decode call (statuses as the decoder left them)."""
        nf = ctypes.c_uint64()
        self._check(lib().flacgpu_get_frames(self._h, None, 0, ctypes.byref(nf)))
        frames = (Frame * max(1, nf.value))()
        self._check(lib().flacgpu_get_frames(self._h, frames, nf.value, ctypes.byref(nf)))
        return [frames[i] for i in range(nf.value)]

    def scan_frames(self, d_in, in_bytes, streams, hip_stream=0):
        n = len(streams)
        res = (StreamResult * n)()
        nf = ctypes.c_uint64()
        rc = lib().flacgpu_scan_frames(self._h, _ptr(d_in), in_bytes, streams, n, None, 0, ctypes.byref(nf), res,
                                       hip_stream or None)
        self._check(rc)
        frames = (Frame * max(1, nf.value))()
        rc = lib().flacgpu_scan_frames(self._h, _ptr(d_in), in_bytes, streams, n, frames, nf.value,
                                       ctypes.byref(nf), res, hip_stream or None)
        self._check(rc)
        return [frames[i] for i in range(nf.value)], list(res)

    def decode_batch_device(self, d_in, in_bytes, d_out, out_bytes, streams, hip_stream=0, results=None):
        n = len(streams)
        res = results if results is not None else (StreamResult * n)()
        rc = lib().flacgpu_decode_batch_device(self._h, _ptr(d_in), in_bytes, _ptr(d_out), out_bytes, streams, n,
                                               res, hip_stream or None)
        self._check(rc)
        return res

    def decode_batch(self, audio_list, infos):
        """Host buffers in, host PCM out (copies over PCIe).  audio_list: per stream the bytes from frame 0."""
        n = len(audio_list)
        arrs = [np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8)) for a in audio_list]
        # the library fills every byte (zeros where a frame failed); 32-bit output takes 4 bytes per sample
        sizes = [int(si.total_samples) * si.channels * 4 if self.output_format == OUT_S32 else si.pcm_bytes for si in infos]
        outs = [np.empty(n, dtype=np.uint8) for n in sizes]
        bufs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (ctypes.c_uint64 * n)(*[a.size for a in arrs])
        sis = (StreamInfo * n)()
        for i, si in enumerate(infos):
            ctypes.memmove(ctypes.byref(sis[i]), ctypes.byref(si), ctypes.sizeof(StreamInfo))
        optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (ctypes.c_uint64 * n)(*[o.size for o in outs])
        res = (StreamResult * n)()
        rc = lib().flacgpu_decode_batch(self._h, bufs, lens, sis, n, optr, caps, res)
        self._check(rc)
        return outs, list(res)

    def timing(self) -> Timing:
        t = Timing()
        self._check(lib().flacgpu_get_timing(self._h, ctypes.byref(t)))
        return t

    # ---- encoder

    @staticmethod
    def make_pcm_streams(entries):
        """entries: iterable of (in_offset, total_samples, out_offset, out_capacity, sample_rate, channels, bps[, md5])."""
        entries = list(entries)
        arr = (PcmStream * len(entries))()
        for i, e in enumerate(entries):
            (arr[i].in_offset, arr[i].total_samples, arr[i].out_offset, arr[i].out_capacity, arr[i].sample_rate,
             arr[i].channels, arr[i].bits_per_sample) = e[:7]
            if len(e) > 7 and e[7] is not None:
                arr[i].md5[:] = list(e[7])
        return arr

    def encode_batch_device(self, d_pcm, pcm_bytes, d_out, out_bytes, streams, opts: EncodeOptions, hip_stream=0):
        n = len(streams)
        res = (EncodeResult * n)()
        rc = lib().flacgpu_encode_batch_device(self._h, _ptr(d_pcm), pcm_bytes, _ptr(d_out), out_bytes, streams, n,
                                               ctypes.byref(opts), res, hip_stream or None)
        self._check(rc)
        return res

    def encode_batch(self, pcm_list, shapes, opts: EncodeOptions):
        """Host buffers in, host .flac streams out.  pcm_list: packed little-endian PCM per stream;
        shapes: per stream (total_samples, sample_rate, channels, bps[, md5])."""
        n = len(pcm_list)
        arrs = [np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8)) for a in pcm_list]
        caps = [encode_bound(sh[0], sh[2], sh[3], opts.blocksize, opts.seektable_interval) for sh in shapes]
        outs = [np.empty(c, dtype=np.uint8) for c in caps]   # only out_length bytes of each are meaningful
        streams = self.make_pcm_streams((0, sh[0], 0, 0, sh[1], sh[2], sh[3], sh[4] if len(sh) > 4 else None)
                                        for sh in shapes)
        iptr = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrs])
        optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        ocap = (ctypes.c_uint64 * n)(*caps)
        res = (EncodeResult * n)()
        rc = lib().flacgpu_encode_batch(self._h, iptr, streams, n, ctypes.byref(opts), optr, ocap, res)
        self._check(rc)
        return [o[: int(r.out_length)] for o, r in zip(outs, res)], list(res)

    def md5_batch_device(self, d_buf, buf_bytes, offsets, lengths, hip_stream=0):
        """MD5 of byte ranges of a device buffer (one GPU lane per range).  Returns a list of 16-byte digests."""
        n = len(offsets)
        offs = (ctypes.c_uint64 * n)(*[int(o) for o in offsets])
        lens = (ctypes.c_uint64 * n)(*[int(x) for x in lengths])
        out = np.zeros(16 * n, dtype=np.uint8)
        rc = lib().flacgpu_md5_batch_device(self._h, _ptr(d_buf), buf_bytes, offs, lens, n, out.ctypes.data,
                                            hip_stream or None)
        self._check(rc)
        return [out[16 * i: 16 * i + 16].tobytes() for i in range(n)]

    def encode_timing(self) -> EncodeTiming:
        t = EncodeTiming()
        self._check(lib().flacgpu_get_encode_timing(self._h, ctypes.byref(t)))
        return t


Engine = Decoder


def encode_bound(total_samples, channels, bps, blocksize, seektable_interval=0) -> int:
    """flacgpu_encode_bound (+ flacgpu_seektable_bytes): a capacity that always fits one encoded stream."""
    return int(lib().flacgpu_encode_bound(int(total_samples), int(channels), int(bps), int(blocksize))) + \
        int(lib().flacgpu_seektable_bytes(int(total_samples), int(blocksize), int(seektable_interval)))


def decode_file(path_or_bytes, decoder: Decoder = None):
    """Convenience: whole .flac file -> (pcm bytes as np.uint8, StreamInfo, StreamResult)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview, np.ndarray)) \
        else open(path_or_bytes, "rb").read()
    data = bytes(data)
    md = parse_metadata(data)
    if not md.has_streaminfo:
        raise FlacError(6, "no STREAMINFO block")
    si = md.streaminfo
    audio = data[md.first_frame_offset:]
    # the output is sized from STREAMINFO: "0 == Implied Unknown" (flacmeta.go:662) gives nothing to size it from,
    # and a frame is at least 8 + channels bytes for at most max_blocksize samples
    if si.total_samples == 0:
        raise FlacError(11, "TotalSamples is unknown (0): scan the frames and decode with an explicit capacity")
    if si.total_samples > (len(audio) // (8 + si.channels)) * si.max_blocksize:
        raise FlacError(6, "STREAMINFO TotalSamples exceeds what the audio bytes can hold")
    dec = decoder or Decoder()
    outs, res = dec.decode_batch([audio], [si])
    if res[0].frames and not res[0].frames_ok:
        raise FlacError(11, "no frame of the stream decoded")
    return outs[0], si, res[0]
