"""Soak run of the randomised decode parity test over many seeds (not part of the test suite: minutes of
host-side encoding).  python scripts/soak_decode.py FIRST_SEED COUNT"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random_streams as T   # noqa: E402
from util import flac_amd, gpu_batch, oracle_decode, split_file   # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
dec = flac_amd.Decoder(0)
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    # every third seed: one layout for all streams, so that waves take frames of several streams
    layout = [(None, None), (None, None), [(2, 16), (1, 16), (2, 24), (6, 16), (8, 24), (3, 12), (2, 8)][(seed // 3) % 7]][seed % 3]
    drawn = [T._draw(rng, seed * 100 + i, ch=layout[0], bps=layout[1]) for i in range(24)]
    pcms, res, infos = gpu_batch(dec, [d[0] for d in drawn], out_fill=0x77, pad=int(rng.choice([1, 64])),
                                 out_pad=int(rng.choice([1, 4, 64])), out_lead=int(rng.integers(0, 4)))
    for (f, src, what), p, r, si in zip(drawn, pcms, res, infos):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert bytes(want) == src, (seed, what)
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), (seed, what)
        assert np.array_equal(p, want), (seed, what)
    if (seed - first) % 10 == 9:
        print(f"seed {seed} ok, {time.time() - t0:.0f} s", flush=True)
print("soak ok", count, "seeds")
