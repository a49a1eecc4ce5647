"""Soak run of the randomised encoder parity test (GPU bytes == encoder oracle bytes, lossless round
trip) over many seeds.  python scripts/soak_encode.py FIRST_SEED COUNT"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_encode as G   # noqa: E402
from test_encode_oracle_cpu import group_pcm, random_group   # noqa: E402
from util import flac_amd, gpu_encode_group, oracle   # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
eng = flac_amd.Decoder(0)
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    o, streams = random_group(rng)
    src = group_pcm(streams, o["blocksize"], rng)
    packed = [p for p, _ in src]
    shapes = [s for _, s in src]
    gopts = flac_amd.encode_options(flags=flac_amd.ENC_MD5, **o)
    try:
        files, res = gpu_encode_group(eng, packed, shapes, gopts, align=int(rng.choice([4, 256])), lead=int(rng.choice([0, 4])))
        G._check_against_oracle(files, res, packed, shapes, gopts, oracle.enc_options(**o))
    except Exception:
        print("seed", seed, o, streams, flush=True)
        raise
    if (seed - first) % 20 == 19:
        print(f"seed {seed} ok, {time.time() - t0:.0f} s", flush=True)
print("soak ok", count, "seeds")
