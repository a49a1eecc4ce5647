"""PCM shapes and encoder options shared by the CPU and GPU encoder tests.  Each group is one
encode call (options are per call); the streams of a group differ in channels, sample size,
rate and length (whole blocks, ragged tails, tails shorter than a run of 16, a single sample)."""

ENC_GROUPS = [
    dict(name="cd-stereo-lpc12", opts=dict(blocksize=4096, max_lpc_order=12, max_partition_order=6),
         streams=[(2, 16, 44100, 4096 * 4 + 4080), (2, 16, 44100, 4096 * 2), (1, 16, 44100, 10000),
                  (2, 16, 48000, 17), (2, 24, 96000, 9000), (3, 12, 32000, 4097), (1, 8, 8000, 5000), (2, 16, 44100, 1),
                  (2, 16, 44100, 0), (1, 4, 8000, 4200), (7, 20, 48000, 4100)]),
    dict(name="fixed-only", opts=dict(blocksize=1152, max_lpc_order=0, max_partition_order=4),
         streams=[(2, 16, 44100, 9000), (6, 16, 48000, 3000), (2, 20, 48000, 5000)]),
    dict(name="8ch24-lpc32-po8", opts=dict(blocksize=16384, max_lpc_order=32, force_lpc_order=32, min_partition_order=8,
                                            max_partition_order=8),
         streams=[(8, 24, 192000, 16384 * 2 + 100), (2, 24, 192000, 16384 + 15)]),
    dict(name="lpc32-bs4096", opts=dict(blocksize=4096, max_lpc_order=32, max_partition_order=8),
         streams=[(2, 16, 44100, 4096 * 3), (1, 24, 48000, 8200)]),
    dict(name="bs16384-lpc12", opts=dict(blocksize=16384, max_lpc_order=12, max_partition_order=6),
         streams=[(2, 16, 44100, 40000), (4, 24, 96000, 20000)]),
    dict(name="odd-bs-forced-stereo", opts=dict(blocksize=1000, max_lpc_order=8, max_partition_order=6, stereo_mode=3,
                                                 qlp_precision=9),
         streams=[(2, 16, 44100, 7777), (2, 24, 44100, 3000)]),
    dict(name="seektable", opts=dict(blocksize=1024, max_lpc_order=8, max_partition_order=4, seektable_interval=3),
         streams=[(2, 16, 44100, 1024 * 10 + 5), (1, 24, 48000, 1024 * 3), (2, 16, 44100, 700), (2, 16, 8000, 1024 * 7)]),
    dict(name="bs16-order8", opts=dict(blocksize=16, max_lpc_order=8, max_partition_order=8),
         streams=[(2, 16, 44100, 16 * 40 + 3), (1, 24, 44100, 333)]),
    dict(name="bs4608-prec15", opts=dict(blocksize=4608, max_lpc_order=12, max_partition_order=8, qlp_precision=15),
         streams=[(2, 16, 44100, 4608 * 2 + 100), (5, 24, 96000, 5000)]),
    dict(name="bs8192-order20-minpo3", opts=dict(blocksize=8192, max_lpc_order=20, min_partition_order=3, max_partition_order=7),
         streams=[(2, 16, 44100, 8192 * 2 + 77), (2, 24, 96000, 8192 + 4000)]),
    dict(name="left-side", opts=dict(blocksize=576, max_lpc_order=6, max_partition_order=3, stereo_mode=1),
         streams=[(2, 16, 22050, 4000)]),
    dict(name="side-right", opts=dict(blocksize=2304, max_lpc_order=12, max_partition_order=5, stereo_mode=2),
         streams=[(2, 16, 44100, 10000)]),
    # stereo pair estimated per frame (two plans per frame); mono and multichannel streams in the same call
    dict(name="cd-stereo-lpc12-estimated", opts=dict(blocksize=4096, max_lpc_order=12, max_partition_order=6, stereo_mode=-2),
         streams=[(2, 16, 44100, 4096 * 4 + 4080), (1, 16, 44100, 4096 * 3), (2, 16, 44100, 4096 * 2), (1, 16, 44100, 10000),
                  (2, 16, 48000, 17), (2, 24, 96000, 9000), (3, 12, 32000, 4097), (2, 8, 8000, 5000), (2, 16, 44100, 1),
                  (2, 16, 44100, 0), (2, 4, 8000, 4200), (7, 20, 48000, 4100), (2, 16, 44100, 3)]),
    dict(name="bs16384-lpc32-estimated", opts=dict(blocksize=16384, max_lpc_order=32, max_partition_order=8, stereo_mode=-2),
         streams=[(2, 24, 192000, 16384 * 2 + 100), (2, 16, 44100, 16384 + 15), (5, 24, 96000, 20000)]),
    dict(name="odd-bs-estimated-seektable", opts=dict(blocksize=1000, max_lpc_order=8, max_partition_order=6, stereo_mode=-2,
                                                       seektable_interval=2),
         streams=[(2, 16, 44100, 7777), (2, 24, 44100, 3000), (2, 20, 48000, 999)]),
    dict(name="bs16-estimated", opts=dict(blocksize=16, max_lpc_order=8, max_partition_order=8, stereo_mode=-2),
         streams=[(2, 16, 44100, 16 * 40 + 3), (2, 24, 44100, 333)]),
]


def special_pcm(kind, n, channels, bps, rng):
    """Signals the synthetic generator does not make: silence, constants, wasted bits, full-scale noise, spikes."""
    import numpy as np
    full = 1 << (bps - 1)
    if kind == "silence":
        return np.zeros((n, channels), dtype=np.int32)
    if kind == "constant":
        return np.full((n, channels), -full // 3, dtype=np.int32)
    if kind == "wasted":
        return (rng.integers(-full // 16, full // 16, size=(n, channels)) * 8).astype(np.int32)
    if kind == "noise":
        return rng.integers(-full, full, size=(n, channels)).astype(np.int32)
    if kind == "spikes":
        x = rng.integers(-3, 4, size=(n, channels)).astype(np.int32)
        idx = rng.integers(0, n, size=max(1, n // 500))
        x[idx, rng.integers(0, channels, size=idx.size)] = rng.choice([-full, full - 1], size=idx.size)
        return x
    if kind == "midnoise":     # large residuals that still beat verbatim: Rice parameters above 14 (5-bit method)
        amp = max(2, full // 32)
        return rng.integers(-amp, amp, size=(n, channels)).astype(np.int32)
    if kind == "ramp":
        return (np.arange(n, dtype=np.int64)[:, None] * (np.arange(channels) + 1) % full - full // 2).astype(np.int32)
    raise ValueError(kind)


SPECIAL_KINDS = ["silence", "constant", "wasted", "noise", "midnoise", "spikes", "ramp"]
