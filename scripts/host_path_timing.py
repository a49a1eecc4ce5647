#!/usr/bin/env python3
"""Times the host-buffer entry points (flacgpu_decode_batch / flacgpu_encode_batch: PCIe copies inside)
on a batch of synthetic files, next to the device-resident path.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import flac_amd
    import synth
    from util import split_file
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    nsamples = 60 * 44100
    files, ffs = synth.make_batch(0, min(n, 64), nsamples, 2, 16, 44100, synth.options(4096, 12), nthreads=16)
    infos, audios = [], []
    for i in range(n):
        si, a = split_file(bytes(files[i % len(files)]))
        infos.append(si)
        audios.append(a)
    eng = flac_amd.Decoder(0, timing=True)
    eng.decode_batch(audios, infos)                   # warm-up at full size: staging buffers and device arenas grow once
    t0 = time.perf_counter()
    pcms, res = eng.decode_batch(audios, infos)
    t_dec = time.perf_counter() - t0
    dev_ms = eng.timing().total_ms
    assert all(r.frames_ok == r.frames for r in res)
    values = n * nsamples * 2
    shapes = [(nsamples, 44100, 2, 16, bytes(si.md5)) for si in infos]
    opts = flac_amd.encode_options(4096, 12)
    eng.encode_batch(pcms, shapes, opts)
    t0 = time.perf_counter()
    outs, eres = eng.encode_batch(pcms, shapes, opts)
    t_enc = time.perf_counter() - t0
    enc_dev_ms = eng.encode_timing().total_ms
    in_b, out_b = sum(len(a) for a in audios), sum(p.size for p in pcms)
    print(json.dumps({"files": n, "flac_MB": in_b / 1e6, "pcm_MB": out_b / 1e6,
                      "decode_host_path_s": t_dec, "decode_device_ms": dev_ms, "decode_host_path_samples_per_s": values / t_dec,
                      "decode_host_path_GB_per_s": (in_b + out_b) / t_dec / 1e9,
                      "encode_host_path_s": t_enc, "encode_device_ms": enc_dev_ms, "encode_host_path_samples_per_s": values / t_enc}))


if __name__ == "__main__":
    main()
