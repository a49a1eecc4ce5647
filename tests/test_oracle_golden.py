"""CPU tests: the oracle against the reference's golden values, the metadata
entry points of the C ABI against the same values, and the ABI surface."""
import collections
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from util import ROOT, flac_amd, oracle, split_file, to_oracle_si

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "fixtures.json")))
FIXTURES = [k for k in GOLD if not k.startswith("_")]


def _data(name):
    return open(os.path.join(ROOT, "testdata", name), "rb").read()


@pytest.mark.parametrize("name", FIXTURES)
def test_parse_metadata_matches_reference_tests(name):
    """flacgpu_parse_metadata == what flacmeta_test.go asserts about Metadata.Read."""
    g = GOLD[name]
    data = _data(name)
    assert len(data) == g["file_bytes"]
    md = flac_amd.parse_metadata(data)
    si = md.streaminfo
    want = g["reference_test"]["streaminfo"]
    got = dict(min_blocksize=si.min_blocksize, max_blocksize=si.max_blocksize, min_framesize=si.min_framesize,
               max_framesize=si.max_framesize, sample_rate=si.sample_rate, channels=si.channels,
               bits_per_sample=si.bits_per_sample, total_samples=si.total_samples, md5=si.md5_signature)
    assert got == want
    blocks = [[flac_amd.BLOCK_NAMES[md.blocks[i].type], md.blocks[i].length, bool(md.blocks[i].last)]
              for i in range(md.n_blocks)]
    assert blocks == g["reference_test"]["blocks"]
    assert md.first_frame_offset == g["census"]["first_frame_offset"]
    st = [b for b in (md.blocks[i] for i in range(md.n_blocks)) if b.type == 4][0]
    pts = flac_amd.parse_seektable(data[st.payload_offset: st.payload_offset + st.length])
    assert [[p.sample_number, p.offset, p.frame_samples] for p in pts] == g["reference_test"]["seektable"]


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_streaminfo_equals_library(name):
    data = _data(name)
    a = oracle.parse_streaminfo(data[8:42])
    b = flac_amd.parse_streaminfo(data[8:42])
    assert bytes(a) == bytes(b)


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_decodes_fixture_to_streaminfo_md5(name):
    """The gate that pins the oracle: decoded PCM hashes to the MD5 the reference's
    tests assert (flacmeta_test.go:37, :121)."""
    g = GOLD[name]
    si, audio = split_file(_data(name))
    pcm, st = oracle.decode_stream(audio, to_oracle_si(si))
    assert pcm.size == si.pcm_bytes
    assert hashlib.md5(pcm.tobytes()).hexdigest() == g["reference_test"]["streaminfo"]["md5"]
    assert oracle.md5(pcm).hex() == g["reference_test"]["streaminfo"]["md5"]
    c = g["census"]
    assert (st.sync_candidates, st.header_candidates, st.frames, st.frames_ok) == \
        (c["sync_candidates"], c["header_candidates"], c["frames"], c["frames"])
    assert st.samples == si.total_samples


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_frame_table(name):
    """Frame offsets hit every seek point; sizes stay inside Min/MaxFrameSize."""
    g = GOLD[name]
    si, audio = split_file(_data(name))
    frames, _ = oracle.scan_frames(audio, to_oracle_si(si))
    by_sample = {f.first_sample: f for f in frames}
    for sample, off, fs in g["reference_test"]["seektable"]:
        if sample == 2 ** 64 - 1:
            continue
        assert by_sample[sample].offset == off and by_sample[sample].blocksize == fs
    sizes = [f.length for f in frames]
    assert min(sizes) == si.min_framesize and max(sizes) == si.max_framesize
    c = g["census"]
    if "frame_sizes" in c:
        assert sizes == c["frame_sizes"]
    else:
        assert {str(k): v for k, v in collections.Counter(sizes).items()} == c["frame_size_histogram"]
    assert frames[-1].blocksize == c["last_blocksize"]
    first, last = frames[0], frames[-1]
    assert audio[first.offset + first.length - 2: first.offset + first.length].hex() == c["crc16_first"]
    assert audio[last.offset + last.length - 2: last.offset + last.length].hex() == c["crc16_last"]
    assert sum(f.length for f in frames) == len(audio)


def test_known_answer_checksums():
    """CRC-8 / CRC-16 / MD5 known answers: fixture header bytes and the classic check strings."""
    assert oracle.crc8(bytes.fromhex("fff8c90800")) == 0x95      # mono fixture, frame 0 header
    assert oracle.crc8(bytes.fromhex("fff8591800")) == 0x6b      # stereo fixture, frame 0 header
    assert oracle.crc16(b"123456789") == 0xFEE8                  # CRC-16/UMTS (poly 0x8005, init 0)
    assert oracle.crc8(b"123456789") == 0xF4                     # CRC-8 (poly 0x07, init 0)
    for msg in (b"", b"a", b"abc", b"x" * 55, b"x" * 56, b"x" * 64, b"y" * 1000):
        assert oracle.md5(np.frombuffer(msg, dtype=np.uint8)) == hashlib.md5(msg).digest()


def test_metadata_error_behaviour():
    """Error paths of Read (flacmeta.go:754-850), as statuses instead of panics."""
    good = _data(FIXTURES[0])
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_metadata(b"OggS" + good[4:])
    assert e.value.status == 2                                   # not "fLaC" (:759)
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_metadata(good[:20])
    assert e.value.status == 5                                   # truncated block (:776-779)
    bad = bytearray(good)
    bad[4] = 0x09                                                # block type 9 is unknown (:546)
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_metadata(bytes(bad))
    assert e.value.status == 3
    dup = good[:42] + good[4:42] + good[42:]                     # two STREAMINFO blocks (:783)
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_metadata(dup)
    assert e.value.status == 4
    si = bytearray(good[8:42])
    si[0:2] = (8).to_bytes(2, "big")                             # MinBlockSize 8 < 16 (:688)
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_streaminfo(bytes(si))
    assert e.value.status == 6
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.parse_seektable(b"\0" * 19)                     # not a multiple of 18 (:823)
    assert e.value.status == 7
    hdr = flac_amd.parse_block_header(bytes([0x84, 0, 0, 57]))   # last flag + VORBIS_COMMENT
    assert (hdr.type, hdr.length, hdr.last) == (5, 57, 1)


def _id3v2(body_len, footer=False, version=(4, 0)):
    size = bytes([(body_len >> 21) & 0x7F, (body_len >> 14) & 0x7F, (body_len >> 7) & 0x7F, body_len & 0x7F])
    tag = b"ID3" + bytes(version) + bytes([0x10 if footer else 0]) + size + bytes(range(256)) * (body_len // 256 + 1)
    tag = tag[:10 + body_len]
    if footer:
        tag += b"3DI" + bytes(version) + bytes([0x10]) + size
    return tag


@pytest.mark.parametrize("name", FIXTURES)
def test_leading_id3v2_tag_is_measured_and_read_stays_strict(name):
    """A tagged file is refused by Read, as the reference refuses it (flacmeta.go:754-761); behind the
    measured tag the parse equals the untagged file's."""
    good = _data(name)
    want = flac_amd.parse_metadata(good)
    assert flac_amd.skip_id3v2(good) == 0
    for tags in ([_id3v2(0)], [_id3v2(300)], [_id3v2(70000, footer=True)], [_id3v2(5, version=(3, 0)), _id3v2(129)]):
        lead = b"".join(tags)
        tagged = lead + good
        with pytest.raises(flac_amd.FlacError) as e:
            flac_amd.parse_metadata(tagged)
        assert e.value.status == 2
        off = flac_amd.skip_id3v2(tagged)
        assert off == len(lead)
        got = flac_amd.parse_metadata(tagged[off:])
        assert bytes(got) == bytes(want)


def test_id3v2_edge_cases():
    assert flac_amd.skip_id3v2(b"") == 0
    assert flac_amd.skip_id3v2(b"ID3") == 0                                   # too short to be a tag header
    assert flac_amd.skip_id3v2(b"ID3\x04\x00\x00\x00\x00\x00\x80" + b"x" * 200) == 0   # size byte with the top bit set: not a tag
    assert flac_amd.skip_id3v2(b"ID3\xff\x00\x00\x00\x00\x00\x01" + b"x" * 200) == 0   # version byte 0xFF: not a tag
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.skip_id3v2(_id3v2(100)[:60])                                 # buffer ends inside the tag
    assert e.value.status == 5
    with pytest.raises(flac_amd.FlacError) as e:
        flac_amd.skip_id3v2(_id3v2(100, footer=True)[:115])                   # ... inside the footer
    assert e.value.status == 5


def test_min_framesize_where_the_reference_leaks_a_byte():
    """flacmeta.go:699 computes MinFrameSize as uint32(bits >> 24) of a word that holds MaxBlockSize above it,
    so the low byte of MaxBlockSize lands in bits 24-31 of the reference's MinFrameSize.  The restatement masks
    the field to its 24 bits (the format's value); the two agree whenever MaxBlockSize % 256 == 0, which is all
    the reference's tests cover (4096, 4608).  With MaxBlockSize 1152 (0x480) they differ, on purpose."""
    si = bytearray(_data(FIXTURES[0])[8:42])
    si[0:2] = (1152).to_bytes(2, "big")
    si[2:4] = (1152).to_bytes(2, "big")
    si[4:7] = (11).to_bytes(3, "big")
    got = flac_amd.parse_streaminfo(bytes(si))
    assert (got.min_blocksize, got.max_blocksize, got.min_framesize) == (1152, 1152, 11)
    bits = int.from_bytes(si[2:10], "big")                 # MaxBlockSize | MinFrameSize | MaxFrameSize
    reference_value = (bits >> 24) & 0xFFFFFFFF
    assert reference_value == (0x80 << 24) | 11 and reference_value != got.min_framesize
    assert bytes(oracle.parse_streaminfo(bytes(si))) == bytes(got)


def test_abi_exports_every_declared_symbol():
    """The shared library loads and exports every function include/flacgpu.h declares."""
    header = open(os.path.join(ROOT, "include", "flacgpu.h")).read()
    declared = set(re.findall(r"\b(flacgpu_[a-z0-9_]+)\s*\(", header))
    assert declared == set(flac_amd.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", flac_amd.LIB_PATH], text=True)
    exported = set(re.findall(r" T (flacgpu_[a-z0-9_]+)", out))
    assert declared <= exported
    L = flac_amd.lib()
    for s in declared:
        assert hasattr(L, s)


def test_product_does_not_touch_the_oracle():
    """Nothing under flac_amd/ or include/ names the oracle or the synth directory."""
    for base in ("flac_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
                assert "flac_synth" not in text, os.path.join(dirpath, f)
