"""GPU parity on randomly drawn stream shapes: every batch mixes channel counts, sample sizes, block
sizes, predictor orders (both tap classes, all three arithmetic classes), Rice methods, escape
partitions, forced partition orders and signal kinds (tones, quiet with full-scale spikes, full-scale
noise, wasted low bits, silence), so that waves hold lanes in different states: some in the unrolled
loop's common case, some rolled back, some constant or finished.  Checked against the oracle and the
source PCM."""
import numpy as np
import pytest

from util import gpu_batch, md5hex, oracle_decode, split_file, synth

pytestmark = pytest.mark.gpu


def _signal(rng, kind, n, ch, bps, idx, rate, bs):
    lim = 1 << (bps - 1)
    if kind == 0:
        return synth.pcm(idx, n, ch, bps, rate, bs)
    if kind == 1:      # quiet, spikes in some channels only
        x = rng.integers(-6, 7, size=(n, ch)).astype(np.int64)
        for c in rng.choice(ch, size=max(1, ch // 2), replace=False):
            at = rng.integers(0, n, size=max(1, n // 300))
            x[at, c] = rng.choice([-lim, lim - 1], size=at.size)
        return x.astype(np.int32)
    if kind == 2:      # full scale noise in some channels, tones in the others
        x = synth.pcm(idx, n, ch, bps, rate, bs).astype(np.int64)
        for c in rng.choice(ch, size=max(1, ch // 2), replace=False):
            x[:, c] = rng.integers(-lim, lim, size=n)
        return x.astype(np.int32)
    if kind == 3:      # wasted low bits (and a channel of silence when there is room)
        w = int(rng.integers(1, max(2, bps // 2)))
        x = (synth.pcm(idx, n, ch, bps, rate, bs).astype(np.int64) >> w) << w
        if ch > 1:
            x[:, -1] = 0
        return x.astype(np.int32)
    x = np.full((n, ch), int(rng.integers(-lim, lim)), dtype=np.int32)   # constant
    x[n // 2:, 0] += 1 if x[0, 0] < lim - 1 else -1
    return x


def _draw(rng, idx, ch=None, bps=None):
    ch = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 8])) if ch is None else ch
    bps = int(rng.choice([8, 12, 16, 16, 20, 24, 24])) if bps is None else bps
    bs = int(rng.choice([192, 576, 1000, 1152, 2048, 4096, 4608]))
    order = int(rng.choice([0, 4, 8, 12, 12, 20, 32]))
    n = int(rng.integers(1, int(3.6 * bs)))
    rate = int(rng.choice([8000, 44100, 48000, 96000]))
    kw = {}
    if rng.random() < 0.3:
        kw["rice_method"] = 1
    if rng.random() < 0.25:
        kw["escape_mode"] = 1
    if rng.random() < 0.3:
        kw["force_partition_order"] = int(rng.choice([0, 1, 3]))
        while bs % (1 << kw["force_partition_order"]) or (bs >> kw["force_partition_order"]) <= order:
            kw["force_partition_order"] -= 1
    if ch == 2:
        kw["stereo_mode"] = int(rng.choice([0, 1, 2, 3]))
    if order >= 12 and rng.random() < 0.5:
        kw["force_lpc_order"] = order
    x = _signal(rng, int(rng.integers(0, 5)), n, ch, bps, idx, rate, bs)
    f, _ = synth.encode(x, bps, rate, synth.options(bs, order, **kw))
    return bytes(f), synth.pack_pcm(x, bps), f"#{idx} ch{ch} bps{bps} bs{bs} order{order} n{n} {kw}"


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
