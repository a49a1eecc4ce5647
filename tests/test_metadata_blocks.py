"""The remaining Marshal* functions of the reference, restated behind the C ABI, against the values the
reference's own tests expect for its two fixtures (flacmeta_test.go:45-60, :129-152, :154-217, :239-264),
plus the error paths of the reference (flacmeta.go:361, :409, :473, :511) and what it panics on."""
import os
import struct

import pytest

from util import flac_amd

BLOB = bytes.fromhex(
    "89504e470d0a1a0a0000000d4948445200000001000000010802000000907753de000000097048597300000b1300000b1301009a9c18"
    "0000000774494d4507d60b1c0a360608443d320000001d74455874436f6d6d656e7400437265617465642077697468205468652047494d50"
    "ef64256e0000000c4944415408d763f8ffff3f0005fe02fedccc59e70000000049454e44ae426082")


def _blocks(testdata, name):
    data = open(os.path.join(testdata, name), "rb").read()
    md = flac_amd.parse_metadata(data)
    return data, {md.blocks[i].type: data[md.blocks[i].payload_offset: md.blocks[i].payload_offset + md.blocks[i].length]
                  for i in range(md.n_blocks)}, md


def test_stereo_fixture_cuesheet(testdata):
    """flacmeta_test.go:154-217."""
    _, blk, _ = _blocks(testdata, "silence-44-s.flac")
    cs = flac_amd.parse_cuesheet(blk[6])
    assert len(blk[6]) == 588
    assert cs["MediaCatalogNumber"] == b"1234567890123" + bytes(115)
    assert (cs["LeadinSamples"], cs["IsCompactDisc"], cs["TotalTracks"]) == (0x15888, True, 4)
    empty = bytes(12)
    want = [(0, 1, b"123456789012", 0, False, 1, [(0, 1)]), (44100, 2, empty, 1, True, 2, [(0, 1), (588, 2)]),
            (88200, 3, empty, 0, False, 1, [(0, 1)]), (162496, 170, empty, 0, False, 0, [])]
    got = [(t["Offset"], t["Number"], t["ISRC"], t["Type"], t["PreEmphasis"], t["IndexPoints"], t["Indexes"])
           for t in cs["Tracks"]]
    assert got == want


def test_stereo_fixture_picture(testdata):
    """flacmeta_test.go:239-264."""
    _, blk, _ = _blocks(testdata, "silence-44-s.flac")
    p = flac_amd.parse_picture(blk[7])
    assert len(blk[7]) == 199
    assert p == {"PictureType": "Cover (front)", "MimeType": "image/png", "Description": "A pixel.", "Width": 1, "Height": 1,
                 "ColorDepth": 24, "NumColors": 0, "Length": 150, "PictureBlob": BLOB}


def test_fixture_vorbis_comments(testdata):
    """flacmeta_test.go:45-60 and :129-152."""
    _, blk, _ = _blocks(testdata, "44100-16-mono.flac")
    v = flac_amd.parse_vorbis_comment(blk[5])
    assert v["Vendor"].startswith("reference libFLAC") and v["TotalComments"] == len(v["Comments"])
    _, blk, _ = _blocks(testdata, "silence-44-s.flac")
    v = flac_amd.parse_vorbis_comment(blk[5])
    assert v["TotalComments"] == 7 == len(v["Comments"])
    assert all("=" in c for c in v["Comments"])


def test_picture_type_names():
    """PictureTypes / PictureType, flacmeta.go:117-139, :165-171."""
    L = flac_amd.lib()
    assert L.flacgpu_picture_type_string(0) == b"Other"
    assert L.flacgpu_picture_type_string(17) == b"A Bright Coloured Fish"
    assert L.flacgpu_picture_type_string(20) == b"Publisher/Studio Logotype"
    assert L.flacgpu_picture_type_string(21) == b"UNKNOWN"


def test_application_block_as_the_reference_reads_it():
    """MarshalApplicationBlock takes 32 BYTES for the id field (flacmeta.go:360): the id is the first four,
    the data starts at byte 32, and what follows must be a multiple of 8 bytes (:361)."""
    b = struct.pack(">I", 0x41424344) + bytes(28) + b"12345678"
    a = flac_amd.parse_application(b)
    assert a == {"Id": 0x41424344, "Data": b"12345678"}
    assert flac_amd.parse_application(b[:10]) == {"Id": 0x41424344, "Data": b""}
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_application(b + b"x")
    assert e.value.status == 13
    with pytest.raises(flac_amd.FlacError) as e:       # the reference panics on a slice shorter than the id
        flac_amd.parse_application(b"ab")
    assert e.value.status == 5


def _cuesheet(tracks):
    b = b"CAT".ljust(128, b"\0") + struct.pack(">Q", 88200) + bytes([0x80]) + bytes(258) + bytes([len(tracks)])
    for off, num, idx in tracks:
        b += struct.pack(">QB", off, num) + bytes(12) + bytes([0xC0]) + bytes(13) + bytes([len(idx)])
        for so, ip in idx:
            b += struct.pack(">QB", so, ip) + bytes(3)
    return b


def test_cuesheet_errors_of_the_reference():
    ok = flac_amd.parse_cuesheet(_cuesheet([(0, 1, [(0, 1), (1176, 2)]), (5880, 170, [])]))
    assert ok["TotalTracks"] == 2 and ok["Tracks"][0]["Indexes"] == [(0, 1), (1176, 2)] and ok["Tracks"][0]["Type"] == 1
    for bad in (_cuesheet([]),                                   # no tracks, flacmeta.go:409
                _cuesheet([(0, 0, [])]),                         # track number 0, :473
                _cuesheet([(0, 1, [(100, 1)])])):                # index offset not a multiple of 588, :511
        with pytest.raises(flac_amd.FlacError) as e:
            flac_amd.parse_cuesheet(bad)
        assert e.value.status == 12
    with pytest.raises(flac_amd.FlacError) as e:                 # the reference panics on short payloads
        flac_amd.parse_cuesheet(_cuesheet([(0, 1, [(0, 1)])])[:-5])
    assert e.value.status == 5


def test_read_rejects_what_the_reference_rejects(testdata):
    """(*Metadata).Read returns the block parsers' errors (flacmeta.go:813-816, :835-838)."""
    data = bytearray(open(os.path.join(testdata, "silence-44-s.flac"), "rb").read())
    md = flac_amd.parse_metadata(bytes(data))
    cue = [md.blocks[i] for i in range(md.n_blocks) if md.blocks[i].type == 6][0]
    data[cue.payload_offset + 395] = 0                           # TotalTracks = 0
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_metadata(bytes(data))
    assert e.value.status == 12
