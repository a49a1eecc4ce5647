"""Randomly drawn stream shapes and signal kinds, shared by the GPU parity tests, the soak scripts and the
CPU property tests: channel counts, sample sizes, block sizes, predictor orders (both tap classes, all three
arithmetic classes), Rice methods, escape partitions, forced partition orders, stereo modes, and signals that
are tones, quiet with full-scale spikes, full-scale noise, wasted low bits or constant."""
import numpy as np

from util import synth


def signal(rng, kind, n, ch, bps, idx, rate, bs):
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


def draw(rng, idx, ch=None, bps=None, block_sizes=(192, 576, 1000, 1152, 2048, 4096, 4608), max_blocks=3.6):
    ch = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 8])) if ch is None else ch
    bps = int(rng.choice([8, 12, 16, 16, 20, 24, 24])) if bps is None else bps
    bs = int(rng.choice(list(block_sizes)))
    order = min(int(rng.choice([0, 4, 8, 12, 12, 20, 32])), bs // 2)
    n = int(rng.integers(1, max(2, int(max_blocks * bs))))
    rate = int(rng.choice([8000, 44100, 48000, 96000]))
    kw = {}
    if rng.random() < 0.3:
        kw["rice_method"] = 1
    if rng.random() < 0.25:
        kw["escape_mode"] = 1
    if rng.random() < 0.3:
        kw["force_partition_order"] = int(rng.choice([0, 1, 3]))
        while kw["force_partition_order"] and (bs % (1 << kw["force_partition_order"]) or (bs >> kw["force_partition_order"]) <= order):
            kw["force_partition_order"] -= 1
    if ch == 2:
        kw["stereo_mode"] = int(rng.choice([0, 1, 2, 3]))
    if order >= 12 and rng.random() < 0.5:
        kw["force_lpc_order"] = order
    x = signal(rng, int(rng.integers(0, 5)), n, ch, bps, idx, rate, bs)
    f, _ = synth.encode(x, bps, rate, synth.options(bs, order, **kw))
    return bytes(f), synth.pack_pcm(x, bps), f"#{idx} ch{ch} bps{bps} bs{bs} order{order} n{n} {kw}"
