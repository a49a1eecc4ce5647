"""Host-side sanitizer runs (CPU only): the product's metadata restatement (flac_amd/csrc/flacmeta.cpp) and the
oracle, built with AddressSanitizer + UndefinedBehaviorSanitizer into the harnesses of tests/sanitize/, over the
reference's fixtures, hand-made metadata blocks and every stream shape of the parity suite, clean and damaged.
The reference indexes short slices unchecked and panics (flacmeta.go:360, :404, :473, :687-702); the
restatement's length checks are what replaces that, and this is the run that shows they hold."""
import os
import struct
import subprocess

import pytest

from cases import CASES, encode_case
from util import ROOT

SAN = os.path.join(ROOT, "tests", "sanitize")


@pytest.fixture(scope="module")
def harness():
    subprocess.check_call(["make", "-s", "-C", SAN])
    return SAN


def _run(binary, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)          # the sanitizer runtime has to come first in the library list
    p = subprocess.run([binary] + files, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "no sanitizer report" in p.stdout and "ERROR" not in p.stderr and "runtime error" not in p.stderr
    return p.stdout


def _block(kind, payload, last=False):
    return bytes([(0x80 if last else 0) | kind]) + len(payload).to_bytes(3, "big") + payload


def test_metadata_restatement_under_sanitizers(harness, testdata, tmp_path):
    fixtures = [os.path.join(testdata, n) for n in ("44100-16-mono.flac", "silence-44-s.flac")]
    # a file with the block kinds the fixtures lack or hold once: APPLICATION, a second PICTURE, a long comment list
    base = open(fixtures[1], "rb").read()
    si = base[4:42]
    app = _block(2, b"fake" + bytes(28) + bytes(range(64)))
    vc = struct.pack("<I", 5) + b"hello" + struct.pack("<I", 40) + b"".join(struct.pack("<I", 3 + i % 7) + b"k=v" + b"x" * (i % 7)
                                                                           for i in range(40))
    pic = struct.pack(">I", 3) + struct.pack(">I", 9) + b"image/png" + struct.pack(">I", 2) + b"hi" + struct.pack(">5I", 1, 1, 24, 0, 10) + b"0123456789"
    made = b"fLaC" + si[:1].replace(b"\x80", b"\x00") + si[1:] + app + _block(4, vc) + _block(6, pic) + _block(6, pic) + _block(1, bytes(16), last=True)
    made = made + base[-2000:]
    path = tmp_path / "made.flac"
    path.write_bytes(made)
    out = _run(os.path.join(harness, "san_meta"), fixtures + [str(path)])
    assert "calls" in out


def test_oracle_under_sanitizers(harness, testdata, tmp_path):
    files = [os.path.join(testdata, n) for n in ("44100-16-mono.flac", "silence-44-s.flac")]
    for i, c in enumerate(CASES):
        name, f, _ = encode_case(i, c)
        if len(f) > 300_000:
            continue
        p = tmp_path / f"{name}.flac"
        p.write_bytes(f)
        files.append(str(p))
    out = _run(os.path.join(harness, "san_oracle"), files)
    assert "runs" in out
