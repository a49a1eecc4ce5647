"""Replays one seed of scripts/soak_decode.py and prints what differs.  python scripts/soak_repro.py SEED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random_streams as T   # noqa: E402
from util import flac_amd, gpu_batch, oracle_decode, split_file   # noqa: E402

seed = int(sys.argv[1])
dec = flac_amd.Decoder(0)
rng = np.random.default_rng(seed)
layout = [(None, None), (None, None), [(2, 16), (1, 16), (2, 24), (6, 16), (8, 24), (3, 12), (2, 8)][(seed // 3) % 7]][seed % 3]
drawn = [T._draw(rng, seed * 100 + i, ch=layout[0], bps=layout[1]) for i in range(24)]
pad, out_pad, out_lead = int(rng.choice([1, 64])), int(rng.choice([1, 4, 64])), int(rng.integers(0, 4))
print("pad", pad, "out_pad", out_pad, "out_lead", out_lead)
pcms, res, infos = gpu_batch(dec, [d[0] for d in drawn], out_fill=0x77, pad=pad, out_pad=out_pad, out_lead=out_lead)
for i, ((f, src, what), p, r, si) in enumerate(zip(drawn, pcms, res, infos)):
    want, ost = oracle_decode(si, split_file(f)[1])
    ok = (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples) and np.array_equal(p, want)
    print(i, "ok " if ok else "BAD", what, "gpu", (r.frames, r.frames_ok, r.samples), "oracle", (ost.frames, ost.frames_ok, ost.samples),
          "" if ok else f"first diff {int(np.nonzero(np.frombuffer(bytes(p), np.uint8)[:len(want)] != np.frombuffer(bytes(want), np.uint8)[:len(p)])[0][:1].sum()) if len(p) else -1}")
# the failing streams alone, one per call
for i, ((f, src, what), p, r, si) in enumerate(zip(drawn, pcms, res, infos)):
    want, ost = oracle_decode(si, split_file(f)[1])
    if (r.frames, r.frames_ok, r.samples) != (ost.frames, ost.frames_ok, ost.samples) or not np.array_equal(p, want):
        p1, r1, _ = gpu_batch(dec, [f])
        print("alone:", i, (r1[0].frames, r1[0].frames_ok, r1[0].samples), np.array_equal(p1[0], want))
        md = flac_amd.scan_frames if hasattr(flac_amd, "scan_frames") else None
