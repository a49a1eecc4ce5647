"""Property tests on the CPU: hundreds of randomly drawn stream shapes (the generator the GPU parity tests
use, at small sizes) through the oracle AND the independently written Python decoder, both against the
source PCM and the STREAMINFO MD5.  For everything the reference's fixtures do not exercise (LPC, fixed
orders 1-4, 5-bit Rice, escapes, wasted bits, stereo modes, 1-8 channels, 8-24 bits) two decoders from two
readings of the format agreeing with the encoder's input is the only independent check there is."""
import hashlib

import numpy as np
from hypothesis import HealthCheck, given, settings, strategies as st

import pydecoder
from random_streams import draw
from util import oracle, oracle_decode, split_file, to_oracle_si


@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2 ** 31 - 1))
def test_oracle_python_decoder_and_source_agree(seed):
    rng = np.random.default_rng(seed)
    f, src, what = draw(rng, seed % 100000, block_sizes=(16, 100, 192, 256, 576), max_blocks=2.6)
    si, audio = split_file(f)
    want, ost = oracle_decode(si, audio)
    assert want.tobytes() == src, what
    assert hashlib.md5(src).hexdigest() == si.md5_signature, what
    assert ost.frames == ost.frames_ok and ost.samples == si.total_samples, what
    psi, ppcm = pydecoder.decode_file(f)
    assert ppcm == src, what
    assert psi["md5"] == bytes(si.md5), what


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2 ** 31 - 1), data=st.data())
def test_damaged_streams_never_break_the_oracle(seed, data):
    """Flips, cuts and junk: the oracle returns statuses, decodes what is intact, and never reports more
    samples than the stream holds.  (The same inputs run under the sanitizers in tests/test_sanitizers.py.)"""
    rng = np.random.default_rng(seed)
    f, src, what = draw(rng, seed % 100000, block_sizes=(16, 192, 576), max_blocks=3.2)
    si, audio = split_file(f)
    bad = bytearray(audio)
    kind = data.draw(st.integers(0, 3))
    if kind == 0 and bad:
        for _ in range(data.draw(st.integers(1, 4))):
            bad[data.draw(st.integers(0, len(bad) - 1))] ^= 1 << data.draw(st.integers(0, 7))
    elif kind == 1:
        bad = bad[: data.draw(st.integers(0, len(bad)))]
    elif kind == 2 and len(bad) > 8:
        at = data.draw(st.integers(0, len(bad) - 8))
        bad[at:at + 8] = data.draw(st.binary(min_size=8, max_size=8))
    else:
        at = data.draw(st.integers(0, len(bad)))
        bad[at:at] = b"\xff\xf8" * 3
    pcm, stt = oracle.decode_stream(bytes(bad), to_oracle_si(si))
    assert stt.frames_ok <= stt.frames and stt.samples <= si.total_samples, what
    assert pcm.size <= si.pcm_bytes
