"""GPU parity on randomly drawn stream shapes: every batch mixes channel counts, sample sizes, block
sizes, predictor orders (both tap classes, all three arithmetic classes), Rice methods, escape
partitions, forced partition orders and signal kinds (tones, quiet with full-scale spikes, full-scale
noise, wasted low bits, silence), so that waves hold lanes in different states: some in the unrolled
loop's common case, some rolled back, some constant or finished.  Checked against the oracle and the
source PCM."""
import numpy as np
import pytest

from random_streams import draw as _draw
from util import gpu_batch, md5hex, oracle_decode, split_file, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 16), (1, 16), (2, 24), (6, 16), (8, 24)])
def test_random_streams_of_one_layout(decoder, shape):
    """Streams that share channel count and sample size are decoded by the same launches, and a wave takes
    consecutive frames across stream boundaries: lanes of one wave then differ in block size, predictor
    order and class, Rice method and signal, and short streams end in the middle of a wave."""
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    drawn = [_draw(rng, 7000 + i, ch=shape[0], bps=shape[1]) for i in range(40)]
    files = [d[0] for d in drawn]
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x77, pad=1)
    for (f, src, what), p, r, si in zip(drawn, pcms, res, infos):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src, what
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), what
        assert np.array_equal(p, want), what


@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_random_stream_shapes(decoder, seed):
    rng = np.random.default_rng(seed)
    drawn = [_draw(rng, seed * 100 + i) for i in range(24)]
    files = [d[0] for d in drawn]
    pcms, res, infos = gpu_batch(decoder, files, out_fill=0x77)
    for (f, src, what), p, r, si in zip(drawn, pcms, res, infos):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src, what
        assert md5hex(src) == si.md5_signature, what
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), what
        assert np.array_equal(p, want), what
