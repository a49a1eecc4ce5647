"""Soak run of the damaged-input parity checks of tests/test_gpu_corrupt.py over many seeds.
python scripts/soak_corrupt.py FIRST_SEED COUNT"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_corrupt as T   # noqa: E402
from util import flac_amd, gpu_batch, oracle, oracle_decode, split_file, to_oracle_si   # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
dec = flac_amd.Decoder(0)
files = T._files()
parts = [T._header_and_audio(f) for _, f in files]
scans = []
for (name, f), (head, a, si) in zip(files, parts):
    frames, _ = oracle.scan_frames(bytes(a), to_oracle_si(si))
    scans.append([fr for fr in frames if fr.status == 0])
t0 = time.time()
failed_frames = 0
collisions = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    variants, names, holes = [], [], []
    for (name, f), (head, a, si), frames in zip(files, parts, scans):
        # random damage: flips, truncation, junk, zero runs
        b = bytearray(a)
        for p in rng.integers(0, len(b), size=int(rng.integers(1, 8))):
            b[p] ^= int(rng.integers(1, 256))
        variants.append(head + bytes(b)); names.append(name + "/flip"); holes.append(None)
        variants.append(head + bytes(a[: int(rng.integers(1, len(a)))])); names.append(name + "/cut"); holes.append(None)
        junk = bytes(rng.integers(0, 256, size=int(rng.integers(1, 400))).astype(np.uint8)) + b"\xff\xf8\xc9\x18\x00" + b"\xff" * 7
        mid = int(rng.integers(1, len(a) - 1))
        variants.append(head + junk + bytes(a[:mid]) + junk + bytes(a[mid:]) + junk); names.append(name + "/junk"); holes.append(None)
        # crafted: a bit changed inside a frame, CRC-16 repaired
        bps_bytes = (si.bits_per_sample + 7) // 8 * si.channels
        b = bytearray(a)
        hole = []
        for fr in (frames[i] for i in rng.choice(len(frames), size=min(2, len(frames)), replace=False)):
            lo, hi = fr.offset + fr.header_len, fr.offset + fr.length - 2
            p = lo + (int(rng.integers(0, min(24, hi - lo))) if rng.random() < 0.4 else int(rng.integers(0, hi - lo)))
            b[p] ^= 1 << int(rng.integers(0, 8))
            crc = oracle.crc16(bytes(b[fr.offset: hi]))
            b[hi], b[hi + 1] = crc >> 8, crc & 0xFF
            hole.append((fr.first_sample * bps_bytes, (fr.first_sample + fr.blocksize) * bps_bytes))
        variants.append(head + bytes(b)); names.append(name + "/craft"); holes.append(hole)
    pcms, res, infos = gpu_batch(dec, variants, pad=int(rng.choice([1, 64])))
    for name, f, p, r, si, hole in zip(names, variants, pcms, res, infos, holes):
        want, ost = oracle_decode(si, split_file(f)[1])
        assert (r.frames, r.frames_ok, r.samples) == (ost.frames, ost.frames_ok, ost.samples), (seed, name)
        keep = np.ones(p.size, dtype=bool)
        if hole is None:
            # once in 2^16 spans random damage leaves the CRC-16 intact: the frame is accepted by both sides,
            # fails in the subframe decoder on both sides, and its part of the output is unspecified
            scanned, _ = oracle.scan_frames(split_file(f)[1], to_oracle_si(si))
            passed = [fr for fr in scanned if fr.status == 0]
            if len(passed) != ost.frames_ok:
                collisions += 1
                print(f"seed {seed} {name}: a damaged frame passed the CRC-16 and failed in the subframe decoder", flush=True)
                bpf = (si.bits_per_sample + 7) // 8 * si.channels
                for fr in passed:
                    keep[fr.first_sample * bpf: (fr.first_sample + fr.blocksize) * bpf] = False
        if hole is not None:
            failed_frames += ost.frames - ost.frames_ok
            if ost.frames_ok != ost.frames:
                for lo, hi in hole:
                    keep[lo:hi] = False
        assert np.array_equal(p[keep], want[keep]), (seed, name)
    if (seed - first) % 10 == 9:
        print(f"seed {seed} ok, {time.time() - t0:.0f} s, {failed_frames} crafted frames failed in the subframe decoder so far", flush=True)
print("soak ok", count, "seeds;", collisions, "CRC-16 collisions")
