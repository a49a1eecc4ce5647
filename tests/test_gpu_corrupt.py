"""GPU parity on damaged input: the frame engine must find and decode exactly the frames the
oracle does, leave the rest of the output alone, and never fault.

Two kinds of damage:
  * random: flipped bytes, truncation, junk before / between / after frames.  The CRC-16 catches these
    in the scan, so a damaged frame is never decoded on either side and the whole output must match;
  * crafted: a byte inside one frame is changed and the frame's CRC-16 repaired, so the frame passes the
    scan and fails (or decodes to something else) inside the subframe decoder.  Counts must match, and
    the output must match everywhere outside the crafted frames (the oracle writes a frame only after it
    decoded completely, the GPU stages it in pieces).
"""
import numpy as np
import pytest

from cases import CASES, encode_case
from util import gpu_batch, oracle, oracle_decode, split_file, to_oracle_si

pytestmark = pytest.mark.gpu

PICK = ["stereo16-lpc12", "mono16", "stereo24", "escape-some", "rice5", "mid-side", "variable-bs", "3ch12-bs1000",
        "5ch20", "lpc32-16bit", "verbatim", "fixed2"]


def _files():
    out = []
    for i, c in enumerate(CASES):
        if c["name"] in PICK:
            name, f, _ = encode_case(i, c)
            out.append((name, f))
    return out


def _header_and_audio(f):
    si, a = split_file(f)
    return f[: len(f) - len(a)], bytearray(a), si


def test_random_damage(decoder):
    rng = np.random.default_rng(11)
    variants, names = [], []
    for name, f in _files():
        head, a, si = _header_and_audio(f)
        # 1: flipped bytes
        b = bytearray(a)
        for p in rng.integers(0, len(b), size=6):
            b[p] ^= int(rng.integers(1, 256))
        variants.append(head + bytes(b)); names.append(name + "/flip")
        # 2: truncated in the middle of a frame
        variants.append(head + bytes(a[: int(rng.integers(len(a) // 3, len(a) - 1))])); names.append(name + "/cut")
        # 3: junk in front, in the middle and at the end (sync-like bytes included)
        junk = bytes(rng.integers(0, 256, size=300).astype(np.uint8)) + b"\xff\xf8\xc9\x18\x00" + b"\xff" * 7
        mid = int(rng.integers(1, len(a) - 1))
        variants.append(head + junk + bytes(a[:mid]) + junk + bytes(a[mid:]) + junk); names.append(name + "/junk")
        # 4: a run of zeros over a frame boundary
        b = bytearray(a)
        z = int(rng.integers(0, max(1, len(b) - 600)))
        b[z: z + 500] = bytes(500)
        variants.append(head + bytes(b)); names.append(name + "/zeros")
    pcms, res, infos = gpu_batch(decoder, variants)
    for name, f, p, r, si in zip(names, variants, pcms, res, infos):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), name
        assert np.array_equal(p, want), name


def test_crafted_frames(decoder):
    rng = np.random.default_rng(12)
    variants, names, holes = [], [], []
    for name, f in _files():
        head, a, si = _header_and_audio(f)
        frames, _ = oracle.scan_frames(bytes(a), to_oracle_si(si))
        frames = [fr for fr in frames if fr.status == 0]
        bps_bytes = (si.bits_per_sample + 7) // 8 * si.channels
        for trial in range(3):
            b = bytearray(a)
            hole = []
            for fr in (frames[i] for i in rng.choice(len(frames), size=min(2, len(frames)), replace=False)):
                lo, hi = fr.offset + fr.header_len, fr.offset + fr.length - 2
                # early bytes hit subframe headers / predictor parameters, late ones residuals
                p = lo + (int(rng.integers(0, min(24, hi - lo))) if trial == 0 else int(rng.integers(0, hi - lo)))
                b[p] ^= 1 << int(rng.integers(0, 8))
                crc = oracle.crc16(bytes(b[fr.offset: hi]))
                b[hi], b[hi + 1] = crc >> 8, crc & 0xFF
                hole.append((fr.first_sample * bps_bytes, (fr.first_sample + fr.blocksize) * bps_bytes))
            variants.append(head + bytes(b)); names.append(f"{name}/craft{trial}"); holes.append(hole)
    pcms, res, infos = gpu_batch(decoder, variants)
    failed = 0
    for name, f, p, r, si, hole in zip(names, variants, pcms, res, infos, holes):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), name
        failed += ost.frames - ost.frames_ok
        keep = np.ones(p.size, dtype=bool)
        if ost.frames_ok != ost.frames:      # a crafted frame that still decodes must match too
            for lo, hi in hole:
                keep[lo:hi] = False
        assert np.array_equal(p[keep], want[keep]), name
    assert failed > 10    # the test is only worth something if frames do fail in the subframe decoder
