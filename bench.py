#!/usr/bin/env python3
"""bench.py — decode throughput of libflacgpu on the headline workload.

Workload (the synthetic batch the metric is quoted on; the two tiny reference
fixtures are a parity test, not a bench line): FILES (default 10 000) FLAC files of 60 s, 44.1 kHz, 16-bit stereo,
block size 4096, fixed + LPC (order <= 12) subframes, Rice partition order <= 6.
UNIQUE (default 256) distinct files are synthesised and encoded on the host by
synth/ (seeded PRNG; no dataset exists offline) and replicated on the device to
FILES independent streams, each with its own input bytes and its own output
region, all resident in HBM before the timed region starts.

A step = one flacgpu_decode_batch_device call over the whole batch: frame scan
(sync + CRC-8 + CRC-16), subframe skim, Rice/LPC decode, decorrelation and
packing to interleaved s16le PCM.  With N ranks every rank decodes its own
batch of FILES files (weak scaling, no collective on the data path).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def encode_leg(args, eng, dev, rank, world, d_pcm, pcm_stride, nfiles, nsamples, channels, bps, rate, blocksize,
               pcm_bytes_per_file, infos, unique, hip_stream, threads):
    """Encodes the decoded batch back to FLAC on the device (flacgpu_encode_batch_device) and reports its
    throughput; checks sampled streams by decoding them with the CPU oracle against the source MD5."""
    import hashlib
    import torch
    import flac_amd
    cap = int(pcm_bytes_per_file * 0.80) // 256 * 256 + 4096     # per-stream output capacity (synthetic ratio is 0.58-0.76)
    nframes = (nsamples + blocksize - 1) // blocksize
    ncand = (4 if args.stereo_mode == -1 else 2) if channels == 2 else channels
    out_bytes = cap * nfiles + 256
    free_b, _ = torch.cuda.mem_get_info(dev)
    need = out_bytes + int(nfiles * nframes * (ncand * (176 + 360) + 96) * 1.15) + (2 << 30)   # output + plans, statistics, frame records
    if free_b < need:
        log(f"encode leg skipped: needs {need / 1e9:.0f} GB, {free_b / 1e9:.0f} GB free")
        return None
    d_enc = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    opts = flac_amd.encode_options(blocksize=blocksize, max_lpc_order=args.lpc_order,
                                   max_partition_order=(args.force_partition_order if args.force_partition_order >= 0
                                                        else args.partition_order),
                                   min_partition_order=max(0, args.force_partition_order),
                                   force_lpc_order=args.force_lpc_order, stereo_mode=args.stereo_mode)
    streams = eng.make_pcm_streams((i * pcm_stride, nsamples, i * cap, cap, rate, channels, bps, bytes(infos[i % unique].md5))
                                   for i in range(nfiles))
    res = eng.encode_batch_device(d_pcm, pcm_stride * nfiles, d_enc, out_bytes, streams, opts, hip_stream=hip_stream)
    bad = [i for i in range(nfiles) if res[i].status != 0]
    assert not bad, f"{len(bad)} streams did not fit their capacity (first: {bad[:4]})"
    if rank == 0:
        import oracle
        for i in sorted({0, nfiles // 2, nfiles - 1}):
            f = d_enc[i * cap: i * cap + int(res[i].out_length)].cpu().numpy().tobytes()
            md = flac_amd.parse_metadata(f)
            osi = oracle.StreamInfo()
            ctypes.memmove(ctypes.byref(osi), ctypes.byref(md.streaminfo), ctypes.sizeof(osi))
            try:
                pcm, st = oracle.decode_stream(f[md.first_frame_offset:], osi)
            except RuntimeError:
                log(f"stream {i}: out_length {int(res[i].out_length)} frames {int(res[i].frames)} min/max frame "
                    f"{res[i].min_framesize}/{res[i].max_framesize} head {f[:80].hex()}")
                raise
            assert hashlib.md5(pcm.tobytes()).hexdigest() == infos[i % unique].md5_signature, f"encode round trip, stream {i}"
    # CPU baseline of the encode leg: the encoder oracle (a scalar C port of the same algorithm) on a few files
    cpu_enc = None
    if rank == 0 and args.cpu_files > 0:
        import oracle
        from concurrent.futures import ThreadPoolExecutor
        ncpu = min(threads, unique, nfiles)
        pcms = [d_pcm[i * pcm_stride: i * pcm_stride + pcm_bytes_per_file].cpu().numpy().tobytes() for i in range(ncpu)]
        oo = oracle.enc_options(blocksize=blocksize, max_lpc_order=args.lpc_order, max_partition_order=opts.max_partition_order,
                                min_partition_order=opts.min_partition_order, force_lpc_order=args.force_lpc_order,
                                stereo_mode=args.stereo_mode)
        cpu_s, passes = 0.0, 0
        with ThreadPoolExecutor(max_workers=threads) as ex:   # ctypes releases the GIL inside the C call
            while cpu_s < 0.5 * args.cpu_seconds and passes < 64:
                t0 = time.perf_counter()
                outs = list(ex.map(lambda p: oracle.encode_stream(p, nsamples, channels, bps, rate, oo, md5=bytes(16))[0], pcms))
                cpu_s += time.perf_counter() - t0
                passes += 1
        for i, o in enumerate(outs):     # the GPU's bytes equal the oracle's (MD5 field aside: the oracle was given zeros)
            g = d_enc[i * cap: i * cap + int(res[i].out_length)].cpu().numpy()
            assert g.size == o.size and np.array_equal(g[42:], o[42:]) and np.array_equal(g[:26], o[:26]), f"encode parity, stream {i}"
        cpu_enc = {"value": passes * ncpu * nsamples * channels / cpu_s, "unit": "samples/s", "cores": min(threads, ncpu), "kind": "port",
                   "sample": f"{ncpu} files of the workload ({ncpu * pcm_bytes_per_file / 1e6:.0f} MB PCM) encoded {passes} times, "
                             f"oracle/flac_enc_oracle.c, {min(threads, ncpu)} threads, {cpu_s:.2f} s"}
    kernel_ms = {k: 0.0 for k in ("pick_ms", "stats_ms", "design_ms", "choose_ms", "layout_ms", "pack_ms", "crc_ms", "total_ms")}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.encode_steps):
        res = eng.encode_batch_device(d_pcm, pcm_stride * nfiles, d_enc, out_bytes, streams, opts, hip_stream=hip_stream)
        t = eng.encode_timing()
        for k in kernel_ms:
            kernel_ms[k] += getattr(t, k) / args.encode_steps
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # lossless round trip at full size: what the encoder wrote is decoded over the PCM it came from and every
    # stream is hashed on the device against the signature of the source
    rt = None
    if not os.environ.get("FLAC_BENCH_NOCHECK"):
        t0 = time.perf_counter()
        dstreams = flac_amd.Decoder.make_streams(
            (i * cap + 42, int(res[i].out_length) - 42, i * pcm_stride, pcm_bytes_per_file, infos[i % unique]) for i in range(nfiles))
        dres = eng.decode_batch_device(d_enc, out_bytes, d_pcm, pcm_stride * nfiles, dstreams, hip_stream=hip_stream)
        nfr = (nsamples + blocksize - 1) // blocksize
        assert all(r.frames_ok == nfr and r.samples == nsamples for r in dres), "re-encoded streams did not decode completely"
        digests = eng.md5_batch_device(d_pcm, pcm_stride * nfiles, [i * pcm_stride for i in range(nfiles)],
                                       [pcm_bytes_per_file] * nfiles, hip_stream=hip_stream)
        assert all(digests[i] == bytes(infos[i % unique].md5) for i in range(nfiles)), "re-encoded streams differ from the source PCM"
        rt = {"streams": nfiles, "seconds": time.perf_counter() - t0}
        if rank == 0:
            log(f"encode -> decode round trip of all {nfiles} streams is lossless (MD5 on the device, {rt['seconds']:.2f} s)")
    flac_total = sum(int(r.out_length) for r in res)
    pcm_total = nfiles * pcm_bytes_per_file
    values = nfiles * nsamples * channels
    alg = pcm_total + flac_total
    # per kernel: its own time (stats_ms covers the pair estimate and the statistics kernel) and its own algorithmic
    # bytes: every analysis kernel has to read the PCM once; the packer reads it and writes the .flac bytes
    per_kernel_ms = {"k_enc_pick": kernel_ms["pick_ms"], "k_enc_stats": kernel_ms["stats_ms"] - kernel_ms["pick_ms"],
                     "k_enc_choose": kernel_ms["choose_ms"], "k_enc_pack": kernel_ms["pack_ms"]}
    per_kernel_alg = {"k_enc_pick": pcm_total, "k_enc_stats": pcm_total, "k_enc_choose": pcm_total, "k_enc_pack": alg}
    dom = max(per_kernel_ms, key=lambda k: per_kernel_ms[k])
    a_s = per_kernel_ms[dom] / 1e3
    enc_traffic, enc_traffic_src = lookup_encode_traffic(workload_tag(args), nfiles)
    traffic = traffic_detail = None
    if enc_traffic:
        # the CRC-16 pass (k_crc16_spans, shared with the decoder) reads the .flac bytes once more: counted at its algorithmic size
        whole = sum(v["read"] + v["write"] for v in enc_traffic.values()) + flac_total
        traffic = enc_traffic[dom]["read"] + enc_traffic[dom]["write"] if dom in enc_traffic else None
        traffic_detail = {"profile": enc_traffic_src, "per_kernel": enc_traffic, "whole_encode_bytes": whole,
                          "whole_encode_over_algorithmic": whole / alg,
                          "note": "the PCM is read by four kernels in turn (pair estimate, statistics, choice, packing)"}
    return {"metric": "encoded PCM samples/sec (per GPU)", "value": values * args.encode_steps / elapsed, "unit": "samples/s",
            "stereo_mode": args.stereo_mode, "steps": args.encode_steps, "ms_per_step": elapsed / args.encode_steps * 1e3, "flac_GB": flac_total / 1e9,
            "ratio": flac_total / pcm_total, "algorithmic_GB_per_s": alg * args.encode_steps / elapsed / 1e9,
            "kernel_ms": kernel_ms, "cpu_baseline": cpu_enc, "round_trip_verified": rt,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": per_kernel_alg[dom] / a_s / 1e9 if a_s > 0 else 0.0, "peak": 8000.0,
                         "unit": "GB/s", "frac": per_kernel_alg[dom] / a_s / 1e9 / 8000.0 if a_s > 0 else 0.0,
                         "traffic": traffic, "traffic_detail": traffic_detail, "limited_by": "valu_issue",
                         "algorithmic_bytes_per_launch": per_kernel_alg[dom], "avg_launch_ms": per_kernel_ms[dom],
                         "per_kernel_ms": per_kernel_ms,
                         "pipeline_frac": alg / (kernel_ms["total_ms"] / 1e3) / 1e9 / 8000.0 if kernel_ms["total_ms"] > 0 else 0.0,
                         "pipeline_ms": kernel_ms["total_ms"]}}


# Named workloads.  cfg3 is the configuration the metric is quoted on (the argparse defaults); cfg4 is the
# 24-bit / 192 kHz / 8-channel one: every subframe LPC of order 32 with Rice partition order 8, which forces
# block size 16384 (a first partition of bs / 256 - 32 samples must not be negative), as S files of T seconds
# (>= 8 GB of PCM per GPU) rather than one hour-long stream cut by range_shards: frames are self-contained, so
# the kernels see the same work either way.
PRESETS = {
    "cfg3": {},
    "cfg4": dict(rate=192000, channels=8, bps=24, blocksize=16384, lpc_order=32, force_lpc_order=32,
                 partition_order=8, force_partition_order=8, files=512, unique=16, seconds=20.0, cpu_files=4),
}


def launch_command(gpus, argv, port):
    """The command that starts one rank per GPU of this node (what the driver itself runs for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child, before anything here
    touches a GPU, and leave with the child's status.  Refuses when the node has fewer than N devices."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()       # counts devices without initialising one
    if have < args.gpus and not os.environ.get("FLAC_BENCH_REHEARSE"):
        print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s); not measurable here", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launch_command(args.gpus, argv, port)
    log("starting", args.gpus, "ranks:", " ".join(cmd))
    sys.exit(subprocess.call(cmd))


def resolve_gpus(gpus_arg, environ):
    """Number of ranks: --gpus when given, else the launcher's WORLD_SIZE (a plain `torchrun --nproc-per-node=N
    bench.py` works), else 1.  Returns (gpus, None) or (gpus, why the run is refused): an explicit --gpus that
    contradicts the launcher is an error, not something to guess about."""
    world = environ.get("WORLD_SIZE")
    if gpus_arg is None:
        return (int(world) if world is not None else 1), None
    if world is not None and int(world) != gpus_arg:
        return gpus_arg, f"--gpus {gpus_arg} but WORLD_SIZE is {world}"
    return gpus_arg, None


def workload_tag(args):
    """Names the stream shape a counter profile was taken on (stored in profiles/*pmc_traffic*.json)."""
    return (f"{args.rate}Hz_{args.bps}bit_{args.channels}ch_bs{args.blocksize}_lpc{args.lpc_order}_force{args.force_lpc_order}"
            f"_po{args.partition_order}_fpo{args.force_partition_order}_{args.seconds:g}s")


def dominant_decode_kernel(classes, channels, bytes_ps):
    """Name of the k_decode instantiation the batch ran in, from the class mask the library reports (bit
    (taps == 32) + 2 * arithmetic); with several classes the widest is named (it is the slowest per sample)."""
    if not classes:
        return "k_decode"
    width = {3: 0, 0: 1, 1: 2, 2: 3}        # arithmetic classes from narrowest (16-bit pairs) to widest (64-bit sums)
    b = max((i for i in range(8) if classes >> i & 1), key=lambda i: (width[i >> 1], i & 1))
    layout = (1 if channels == 2 and bytes_ps == 2 else 2 if channels == 1 and bytes_ps == 2 and b >> 1 == 3
              else 3 if channels * bytes_ps >= 8 else 0)
    return f"k_decode<{32 if b & 1 else 12}, {b >> 1}, {layout}>"


def lookup_traffic(tag, kernel, nfiles):
    """HBM traffic of `kernel` from the newest committed counter profile of this workload (profiles/*pmc_traffic*.json),
    scaled by file count.  Returns (traffic bytes or None, details dict).  Counters cannot be read from inside this
    process, so the figure comes from a rocprofv3 --pmc run of this same command at fewer files.  The write side
    (WRITE_SIZE) needs no correction.  The read side is reported three ways: raw FETCH_SIZE, the request-size split
    (TCC_EA0_RDREQ_DRAM_32B: read requests counted in 32-byte units, whatever their size) when the profile has that pass,
    and raw x 2, the upper bound that holds if every request were a 128-byte one tallied at 64 (true for wide
    coalesced reads on gfx950).  `traffic` uses the split when present, else the raw figure."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get("workload_tag") != tag or kernel not in pmc["kernels"]:
                continue
            kd = pmc["kernels"][kernel]
            scale = nfiles / pmc["files"]
            raw = kd["FETCH_SIZE_KiB"] * 1024.0 * scale
            wr = kd["WRITE_SIZE_KiB"] * 1024.0 * scale
            split = None
            if "TCC_EA0_RDREQ_DRAM_32B" in kd:
                # read requests in units of 32 bytes (a 64-byte request counts 2, a 128-byte one 4): the exact figure.
                # (FETCH_SIZE is built from TCC_BUBBLE, the count of 128-byte requests, which reads 0 on this chip.)
                split = kd["TCC_EA0_RDREQ_DRAM_32B"] * 32.0 * scale
            read = split if split is not None else raw
            return read + wr, {"profile": os.path.relpath(path, ROOT), "profile_commit": pmc.get("commit"),
                               "files_profiled": pmc["files"], "write_bytes": wr, "read_bytes_fetch_size_raw": raw,
                               "read_bytes_request_split": split, "read_bytes_upper_bound_x2": 2.0 * raw,
                               "read_requests": kd.get("TCC_EA0_RDREQ", 0) * scale or None,
                               "read_side_used": "32 x TCC_EA0_RDREQ_DRAM_32B" if split is not None else "raw FETCH_SIZE"}
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, {"note": f"no committed counter profile of {kernel} on workload {tag}"}


def lookup_encode_traffic(tag, nfiles):
    """Memory-side bytes of every encoder kernel per encode call, from the newest committed request-counter profile of
    this workload with the encode leg on (profiles/*pmc_enctraffic*.json): 32 x (TCC_EA0_RDREQ_DRAM_32B +
    TCC_EA0_WRREQ_WRITE_DRAM_32B) per dispatch, scaled by file count.  {kernel family: {"read", "write"}} or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_enctraffic*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get("workload_tag") != tag:
                continue
            scale = nfiles / pmc["files"]
            out = {}
            for k, kd in pmc["kernels"].items():
                if k.startswith("k_enc_"):
                    out[k.split("<")[0]] = {"read": kd.get("TCC_EA0_RDREQ_DRAM_32B", 0.0) * 32.0 * scale,
                                            "write": kd.get("TCC_EA0_WRREQ_WRITE_DRAM_32B", 0.0) * 32.0 * scale}
            if out:
                return out, os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def lookup_valu(tag, kernel):
    """Vector instructions per sample step of `kernel` (SQ_INSTS_VALU / waves / block size) from the newest committed
    SQ counter profile of this workload (profiles/*pmc_valu*.json), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_valu*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get("workload_tag") == tag and kernel in pmc["kernels"]:
                return float(pmc["kernels"][kernel]["valu_per_sample_step"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


NOMINAL_CLOCK_MHZ = 2400.0   # MI355X peak engine clock


def lookup_clock(tag, kernel):
    """Median clock in MHz the chip held under `kernel` in the newest committed probe run of this workload
    (profiles/*clock*.json, written by profiles/make_clock_json.py from a library built with the clock probe), or None."""
    import glob
    short = "decode" if kernel.startswith("k_decode") else kernel.replace("k_scan_tiles", "scan").replace("k_skim", "skim")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*clock*.json")), reverse=True):
        try:
            probe = json.load(open(path))
            if probe.get("workload_tag") == tag and short in probe["kernels"]:
                return float(probe["kernels"][short]["clock_MHz_median"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def device_description(torch, dev):
    prop = torch.cuda.get_device_properties(dev)
    return {"name": prop.name, "arch": getattr(prop, "gcnArchName", None), "compute_units": prop.multi_processor_count,
            "hbm_GB": prop.total_memory / 1e9, "clock_MHz": getattr(prop, "clock_rate", 0) / 1e3 or None,
            "l2_MB": getattr(prop, "L2_cache_size", 0) / 1e6 or None}


def valu_issue(tag, kernel, values, seconds, device):
    """How much of the chip's vector-instruction issue the decode kernel's launch used: its vector instructions per
    sample step (one wave instruction covers one sample of 64 subframes) from the committed SQ profile of this
    workload, times the launch's sample steps, over issue_peak = CUs x 4 SIMDs x clock / 4 cycles per instruction."""
    per_step, src = lookup_valu(tag, kernel)
    if per_step is None or seconds <= 0:
        return None
    clock_mhz = device.get("clock_MHz") or NOMINAL_CLOCK_MHZ     # the runtime does not always report one
    clock = clock_mhz * 1e6
    peak = device["compute_units"] * 4 * clock / 4.0
    rate = per_step * (values / 64.0) / seconds
    out = {"insts_per_sample_step": per_step, "source": src, "sample_steps_per_launch": values / 64.0,
           "issue_peak_per_s": peak, "frac": rate / peak, "clock_MHz_nominal": clock_mhz,
           # the clock at which the launch's instructions alone would fill every issue slot of its duration
           "clock_MHz_if_all_slots_filled": clock_mhz * rate / peak}
    measured, clock_src = lookup_clock(tag, kernel)
    if measured:
        # the clock the chip held under this kernel in the committed probe run: the fraction of the issue slots of
        # THAT clock is what says whether the kernel has slots to spare (box and run differ by a few percent)
        out.update({"clock_MHz_measured": measured, "clock_source": clock_src,
                    "frac_at_measured_clock": out["clock_MHz_if_all_slots_filled"] / measured})
    return out


def host_description():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = None
    return {"cpu_model": model, "cpu_count": os.cpu_count(), "cpus_usable": usable}


def run_workload(args, env):
    """One file-sharded workload on every rank; returns the result line (a dict) on rank 0, None elsewhere."""
    import torch
    import flac_amd
    import synth
    rank, local_rank, world, dev, red_dev = env["rank"], env["local_rank"], env["world"], env["dev"], env["red_dev"]
    if world > 1:
        import torch.distributed as dist
    rate, channels, bps, blocksize = args.rate, args.channels, args.bps, args.blocksize
    nsamples = int(round(args.seconds * rate))
    bytes_ps = (bps + 7) // 8
    pcm_bytes_per_file = nsamples * channels * bytes_ps
    unique = max(1, min(args.unique, args.files))

    # ---- corpus: UNIQUE files synthesised + encoded on the host ------------------------------
    threads = args.cpu_threads or min(16, os.cpu_count() or 1)
    # every rank makes its own corpus: a node's CPUs are shared by its ranks (a GPU's share is 16 or more of them)
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    enc_threads = max(1, min(threads, usable // world)) if world > 1 else threads
    t0 = time.time()
    okw = dict(max_partition_order=args.partition_order)
    if args.force_lpc_order:
        okw["force_lpc_order"] = args.force_lpc_order
    if args.force_partition_order >= 0:
        okw["force_partition_order"] = args.force_partition_order
    opts = synth.options(blocksize, args.lpc_order, **okw)
    files, ffs = synth.make_batch(rank * unique, unique, nsamples, channels, bps, rate, opts, nthreads=enc_threads)
    infos, audios = [], []
    for f, ff in zip(files, ffs):
        md = flac_amd.parse_metadata(f[: ff + 64].tobytes())
        assert md.first_frame_offset == ff and md.has_streaminfo
        si = flac_amd.StreamInfo()
        ctypes.memmove(ctypes.byref(si), ctypes.byref(md.streaminfo), ctypes.sizeof(si))
        infos.append(si)
        audios.append(f[ff:])
    comp_unique = sum(a.size for a in audios)
    if rank == 0:
        log(f"corpus: {unique} unique files, {comp_unique / 1e6:.1f} MB FLAC for {unique * pcm_bytes_per_file / 1e6:.1f} MB PCM "
            f"(ratio {comp_unique / (unique * pcm_bytes_per_file):.3f}) in {time.time() - t0:.1f} s on {enc_threads} threads")

    # ---- device-resident batch ---------------------------------------------------------------
    nfiles = args.files
    free_b, total_b = torch.cuda.mem_get_info(dev)
    per_file = (comp_unique / unique) * 1.15 + pcm_bytes_per_file + 64 * 1024
    fit = int((free_b - 6e9) / per_file)
    if fit < nfiles:
        log(f"only {free_b / 1e9:.0f} GB free: batch reduced from {nfiles} to {fit} files")
        nfiles = max(1, fit)
    in_offs = np.zeros(nfiles, dtype=np.int64)
    pos = 0
    for i in range(nfiles):
        in_offs[i] = pos
        pos += (audios[i % unique].size + 255) // 256 * 256
    in_bytes = pos
    out_stride = (pcm_bytes_per_file + 255) // 256 * 256
    out_bytes = out_stride * nfiles
    d_in = torch.empty(in_bytes, dtype=torch.uint8, device=dev)
    d_out = torch.empty(out_bytes, dtype=torch.uint8, device=dev)
    d_unique = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in audios]
    for i in range(nfiles):
        u = d_unique[i % unique]
        d_in[in_offs[i]: in_offs[i] + u.numel()].copy_(u)
    torch.cuda.synchronize(dev)
    del d_unique
    streams = flac_amd.Decoder.make_streams(
        (int(in_offs[i]), audios[i % unique].size, i * out_stride, pcm_bytes_per_file, infos[i % unique])
        for i in range(nfiles))
    comp_total = sum(audios[i % unique].size for i in range(nfiles))
    pcm_total = nfiles * pcm_bytes_per_file
    values_total = nfiles * nsamples * channels
    if rank == 0:
        log(f"batch: {nfiles} files, {comp_total / 1e9:.2f} GB FLAC in HBM -> {pcm_total / 1e9:.2f} GB PCM")

    dec = flac_amd.Decoder(local_rank, timing=True)
    hip_stream = torch.cuda.current_stream(dev).cuda_stream
    results = (flac_amd.StreamResult * nfiles)()

    def step():
        dec.decode_batch_device(d_in, in_bytes, d_out, out_bytes, streams, hip_stream=hip_stream, results=results)

    def barrier():
        if world > 1:
            dist.barrier()

    for w in range(args.warmup):
        step()
        if rank == 0:
            log(f"warmup {w}: {dec.timing().total_ms:.1f} ms device time")
    # correctness of what is about to be timed: every frame decoded, PCM matches the STREAMINFO MD5
    import hashlib
    if args.warmup == 0:
        step()
    exp_frames = (nsamples + blocksize - 1) // blocksize
    check = not os.environ.get("FLAC_BENCH_NOCHECK")   # kernel-ablation experiments produce wrong output on purpose
    bad = [i for i in range(nfiles) if results[i].frames_ok != exp_frames or results[i].samples != nsamples]
    assert not (bad and check), f"{len(bad)} streams did not decode completely (first: {bad[:4]})"
    for i in sorted({0, nfiles // 2, nfiles - 1} if check else ()):
        got = d_out[i * out_stride: i * out_stride + pcm_bytes_per_file].cpu().numpy()
        assert hashlib.md5(got.tobytes()).hexdigest() == infos[i % unique].md5_signature, f"MD5 mismatch, stream {i}"

    md5_check = None
    if check:
        # every stream's PCM against its STREAMINFO signature, hashed where it lies (one GPU lane per stream)
        t0 = time.perf_counter()
        digests = dec.md5_batch_device(d_out, out_bytes, [i * out_stride for i in range(nfiles)], [pcm_bytes_per_file] * nfiles,
                                       hip_stream=hip_stream)
        wrong = [i for i in range(nfiles) if digests[i] != bytes(infos[i % unique].md5)]
        assert not wrong, f"{len(wrong)} streams differ from their STREAMINFO MD5 (first: {wrong[:4]})"
        md5_check = {"streams": nfiles, "seconds": time.perf_counter() - t0}
        if rank == 0:
            log(f"MD5 of all {nfiles} decoded streams equals STREAMINFO ({md5_check['seconds']:.2f} s on the device)")

    kernel_ms = {k: 0.0 for k in ("scan_ms", "compact_ms", "crc16_ms", "resolve_ms", "skim_ms", "decode_ms", "results_ms", "total_ms")}
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = dec.timing()
        for k in kernel_ms:
            kernel_ms[k] += getattr(t, k)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]     # a straggler shows here
        tt = mine.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        vv = torch.tensor([float(values_total), float(pcm_total), float(comp_total)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(vv, op=dist.ReduceOp.SUM)
        values_all, pcm_all, comp_all = [float(x) for x in vv.tolist()]
    else:
        values_all, pcm_all, comp_all = float(values_total), float(pcm_total), float(comp_total)
    for k in kernel_ms:
        kernel_ms[k] /= max(1, args.steps)

    # ---- encode leg: the PCM just decoded, encoded again on the device (reported beside the headline metric) ----
    enc = None
    if args.encode_steps > 0:
        d_in = None                                  # the compressed input is no longer needed on the device
        torch.cuda.empty_cache()
        enc = encode_leg(args, dec, dev, rank, world, d_out, out_stride, nfiles, nsamples, channels, bps, rate,
                         blocksize, pcm_bytes_per_file, infos, unique, hip_stream, threads)

    # ---- CPU baseline: the oracle (a scalar C port), timed on this host's cores on a sample ---
    cpu = None
    if rank == 0 and args.cpu_files > 0:
        import oracle
        ncpu = min(args.cpu_files, unique)
        offs, lens, pos = [], [], 0
        for i in range(ncpu):
            offs.append(pos)
            lens.append(audios[i].size)
            pos += audios[i].size
        blob = np.concatenate([audios[i] for i in range(ncpu)])
        pcm = np.zeros(ncpu * pcm_bytes_per_file, dtype=np.uint8)
        oinfos = []
        for i in range(ncpu):
            o = oracle.StreamInfo()
            ctypes.memmove(ctypes.byref(o), ctypes.byref(infos[i]), ctypes.sizeof(o))
            oinfos.append(o)
        outs = [i * pcm_bytes_per_file for i in range(ncpu)]
        cpu_s, passes = 0.0, 0
        while cpu_s < args.cpu_seconds and passes < 64:
            t0 = time.perf_counter()
            _, st = oracle.decode_batch(blob, offs, lens, oinfos, pcm, outs, [pcm_bytes_per_file] * ncpu, nthreads=threads)
            cpu_s += time.perf_counter() - t0
            passes += 1
            assert all(s.frames_ok == exp_frames for s in st)
        got = d_out[:pcm_bytes_per_file].cpu().numpy()
        assert np.array_equal(got, pcm[:pcm_bytes_per_file]), "GPU PCM differs from the CPU oracle"
        cpu = {"value": passes * ncpu * nsamples * channels / cpu_s, "unit": "samples/s", "cores": threads, "kind": "port",
               "threads": threads, "host_cpu_count": os.cpu_count(),
               "sample": f"{ncpu} of the {unique} unique files ({ncpu * pcm_bytes_per_file / 1e6:.0f} MB PCM) decoded {passes} times, "
                         f"oracle/flac_oracle.c, {threads} threads, {cpu_s:.2f} s"}

    # ---- the host-buffer entry point on a sample: the rate a caller with pageable buffers sees, PCIe both ways ----
    pcie = None
    if rank == 0 and args.host_files > 0:
        nh = min(args.host_files, nfiles)
        h_audio = [audios[i % unique] for i in range(nh)]
        h_infos = [infos[i % unique] for i in range(nh)]
        dec.decode_batch(h_audio, h_infos)                       # staging buffers and device arenas grow once
        t0 = time.perf_counter()
        h_pcm, h_res = dec.decode_batch(h_audio, h_infos)
        h_s = time.perf_counter() - t0
        assert all(r.frames_ok == exp_frames for r in h_res) or not check
        if check:
            assert hashlib.md5(h_pcm[0].tobytes()).hexdigest() == infos[0].md5_signature, "host-buffer path: MD5 mismatch"
        h_bytes = sum(a.size for a in h_audio) + nh * pcm_bytes_per_file
        pcie = {"value": nh * nsamples * channels / h_s, "unit": "samples/s", "files": nh, "seconds": h_s,
                "bus_GB_per_s": h_bytes / h_s / 1e9,
                "what": "flacgpu_decode_batch: caller's pageable buffers in and out, staged through pinned memory in chunks "
                        "(copy in, decode and copy out of consecutive chunks overlap)"}
        del h_pcm

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        alg_bytes = comp_total + pcm_total            # per launch of the decode kernel (one rank)
        dec_s = kernel_ms["decode_ms"] / 1e3
        achieved = alg_bytes / dec_s / 1e9 if dec_s > 0 else 0.0
        # HBM traffic of the dominant kernel: counters cannot be read from inside this process, so the figure
        # comes from a committed rocprofv3 --pmc profile (profiles/*pmc_traffic*.json) of this same workload and
        # this same kernel instantiation, scaled by file count; none that matches -> null
        dominant = dominant_decode_kernel(dec.timing().decode_classes, channels, bytes_ps)
        traffic, traffic_note = lookup_traffic(workload_tag(args), dominant, nfiles)
        device = device_description(torch, dev)
        tot_s = kernel_ms["total_ms"] / 1e3
        valu = valu_issue(workload_tag(args), dominant, values_total, dec_s, device)
        out = {
            "metric": "decoded PCM samples/sec",
            "value": values_all * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32" if bps <= 16 else "int64",
            "data": "synthetic",
            "config": {
                "name": args.config, "workload_tag": workload_tag(args),
                "workload": f"{nfiles} x {args.seconds:g} s {rate / 1000:g} kHz {bps}-bit {channels}-channel FLAC, blocksize {blocksize}, "
                            f"fixed+LPC(<={args.lpc_order})" + (f" forced order {args.force_lpc_order}" if args.force_lpc_order else "")
                            + (f", Rice partition order {args.force_partition_order}" if args.force_partition_order >= 0 else "")
                            + f", {unique} unique files replicated in HBM",
                "files_per_gpu": nfiles, "unique_files": unique,
                "compressed_GB_per_gpu": comp_total / 1e9, "pcm_GB_per_gpu": pcm_total / 1e9,
            },
            "pcm_MB_per_s": pcm_all * args.steps / elapsed / 1e6,
            "algorithmic_GB_per_s": (pcm_all + comp_all) * args.steps / elapsed / 1e9,
            "kernel_ms": kernel_ms,
            "rank_ms_per_step": rank_ms,
            # priced against the HBM roofline (algorithmic bytes over the peak); what the counters show the kernel is held
            # up by is in limited_by, with the vector-issue fraction beside it
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_detail": traffic_note,
                         "limited_by": "valu_issue" if valu else None, "valu": valu,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kernel_ms["decode_ms"],
                         # every kernel of the batch (scan .. results) against the same algorithmic bytes
                         "pipeline_frac": alg_bytes / tot_s / 8e12 if tot_s > 0 else 0.0, "pipeline_ms": kernel_ms["total_ms"]},
            "device": device,
            "host": host_description(),
            "cpu_baseline": cpu,
            "pcie_inclusive": pcie,
            "md5_verified": md5_check,
            "encode": enc,
        }
        result = out
    dec.close()
    del d_out
    torch.cuda.empty_cache()
    return result


def run_range(args, env):
    """--shard range: ONE stream of args.seconds, cut at its seek points (SeekpointBlock, flacmeta.go:256-260) into one
    frame range per rank.  Every rank builds the same stream on its own GPU (synthetic PCM of a few seconds repeated,
    encoded by the GPU encoder with a seek table), decodes its range into a region of its own (sample_base), and
    the stream's MD5, a chain over all its PCM in order, is finished on rank 0 from the ranges.  Strong scaling:
    the stream is fixed, a rank's share shrinks with the number of ranks."""
    import hashlib
    import torch
    import flac_amd
    import synth
    from flac_amd import sharding
    rank, local_rank, world, dev, red_dev = env["rank"], env["local_rank"], env["world"], env["dev"], env["red_dev"]
    if world > 1:
        import torch.distributed as dist
    rate, channels, bps, blocksize = args.rate, args.channels, args.bps, args.blocksize
    bytes_ps = (bps + 7) // 8
    fb = channels * bytes_ps
    n_unit = max(blocksize, int(round(min(args.seconds, 5.0) * rate)) // blocksize * blocksize)   # a few seconds, whole blocks
    reps = max(1, int(round(args.seconds * rate / n_unit)))
    nsamples = n_unit * reps
    x = synth.pcm(0xA11, n_unit, channels, bps, rate, blocksize)
    unit = np.frombuffer(synth.pack_pcm(x, bps), dtype=np.uint8)
    md = hashlib.md5()
    for _ in range(reps):
        md.update(unit.tobytes())
    want_md5 = md.digest()
    pcm_total = nsamples * fb
    d_pcm = torch.from_numpy(unit.copy()).to(dev).repeat(reps)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    d_pcm = torch.cat([d_pcm, pad])
    dec = flac_amd.Decoder(local_rank, timing=True)
    hip_stream = torch.cuda.current_stream(dev).cuda_stream
    cap = flac_amd.encode_bound(nsamples, channels, bps, blocksize, args.seek_frames)
    d_enc = torch.empty(cap + 256, dtype=torch.uint8, device=dev)
    opts = flac_amd.encode_options(blocksize=blocksize, max_lpc_order=args.lpc_order,
                                   max_partition_order=(args.force_partition_order if args.force_partition_order >= 0 else args.partition_order),
                                   min_partition_order=max(0, args.force_partition_order), force_lpc_order=args.force_lpc_order,
                                   stereo_mode=args.stereo_mode, seektable_interval=args.seek_frames)
    es = dec.make_pcm_streams([(0, nsamples, 0, cap, rate, channels, bps, want_md5)])
    eres = dec.encode_batch_device(d_pcm, pcm_total, d_enc, cap + 256, es, opts, hip_stream=hip_stream)
    assert eres[0].status == 0, "the stream did not fit its capacity"
    flac_len = int(eres[0].out_length)
    del d_pcm
    torch.cuda.empty_cache()
    head_len = min(flac_len, 42 + 4 + 18 * (nsamples // blocksize // args.seek_frames + 2) + 64)
    head = d_enc[:head_len].cpu().numpy().tobytes()
    mdt = flac_amd.parse_metadata(head)
    seek = [b for b in mdt.blocks[: mdt.n_blocks] if b.type == 4][0]
    pts = flac_amd.parse_seektable(head[seek.payload_offset: seek.payload_offset + seek.length])
    si = flac_amd.StreamInfo()
    ctypes.memmove(ctypes.byref(si), ctypes.byref(mdt.streaminfo), ctypes.sizeof(si))
    assert bytes(si.md5) == want_md5 and si.total_samples == nsamples
    ff = int(mdt.first_frame_offset)
    audio_len = flac_len - ff
    shards = sharding.range_shards_with_samples(pts, audio_len, nsamples, world)
    mine = sharding.my_range(shards, rank, world)
    if rank == 0:
        log(f"one stream: {args.seconds:g} s, {flac_len / 1e9:.2f} GB FLAC for {pcm_total / 1e9:.2f} GB PCM, {len(pts)} seek points, "
            f"{len(shards)} frame ranges over {world} rank(s)")
    # the decoder wants d_in 16-byte aligned: the stream starts at d_enc, ranges are offsets into it
    if mine is None:
        off = n = base = ns = 0
    else:
        off, n, base, ns = mine
    my_cap = ns * fb
    d_out = torch.empty(my_cap + 256, dtype=torch.uint8, device=dev)
    streams = dec.make_streams([(ff + off, n, 0, my_cap, si, base)]) if mine is not None else None
    results = (flac_amd.StreamResult * 1)()

    def step():
        if streams is not None:
            dec.decode_batch_device(d_enc, flac_len + 16, d_out, my_cap + 256, streams, hip_stream=hip_stream, results=results)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(max(1, args.warmup)):
        step()
    kernel_ms = {k: 0.0 for k in ("scan_ms", "compact_ms", "crc16_ms", "resolve_ms", "skim_ms", "decode_ms", "results_ms", "total_ms")}
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = dec.timing()
        for k in kernel_ms:
            kernel_ms[k] += getattr(t, k) / args.steps
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine_t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        every = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(every, mine_t)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)
    # what every rank decoded, gathered as rows; the stream's MD5 finished on rank 0 from the ranges' PCM
    rows = sharding.result_rows([rank] if mine is not None else [], [results[0]] if mine is not None else [])
    table = sharding.gather_rows(rows, len(shards), device=None if env["rehearse"] or world == 1 else dev)
    t0 = time.perf_counter()
    local = d_out[:my_cap]
    digest = sharding.chained_md5(local.cpu() if env["rehearse"] else local)
    md5_s = time.perf_counter() - t0
    result = None
    if rank == 0:
        assert int(table[:, 1].sum()) == int(table[:, 2].sum()) and int(table[:, 3].sum()) == nsamples, "a range did not decode completely"
        assert digest == want_md5, "the stream's MD5, finished from the ranges, differs from STREAMINFO"
        log(f"MD5 of the stream finished on rank 0 from {len(shards)} ranges: equals STREAMINFO ({md5_s:.1f} s)")
        values = nsamples * channels
        my_alg = n + my_cap
        dec_s = kernel_ms["decode_ms"] / 1e3
        tot_s = kernel_ms["total_ms"] / 1e3
        dominant = dominant_decode_kernel(dec.timing().decode_classes, channels, bytes_ps)
        result = {
            "metric": "decoded PCM samples/sec", "value": values * args.steps / elapsed, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int32" if bps <= 16 else "int64", "data": "synthetic",
            "config": {"name": args.config, "shard": "range", "workload_tag": workload_tag(args),
                       "workload": f"one {args.seconds:g} s {rate / 1000:g} kHz {bps}-bit {channels}-channel stream, blocksize {blocksize}, "
                                   f"cut at seek points into {len(shards)} frame ranges, one per rank",
                       "ranges": [[int(v) for v in sh] for sh in shards], "compressed_GB": flac_len / 1e9, "pcm_GB": pcm_total / 1e9},
            "pcm_MB_per_s": pcm_total * args.steps / elapsed / 1e6,
            "kernel_ms": kernel_ms, "rank_ms_per_step": rank_ms,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": my_alg / dec_s / 1e9 if dec_s > 0 else 0.0, "peak": 8000.0,
                         "unit": "GB/s", "frac": my_alg / dec_s / 8e12 if dec_s > 0 else 0.0, "traffic": None,
                         "algorithmic_bytes_per_launch": my_alg, "avg_launch_ms": kernel_ms["decode_ms"],
                         "pipeline_frac": my_alg / tot_s / 8e12 if tot_s > 0 else 0.0,
                         "note": "rank 0's range; a range of one stream holds few frames (one lane per subframe), so the launch "
                                 "does not fill the chip: see the file-sharded configurations for the kernel's rate"},
            "cpu_baseline": None,
            "md5_verified": {"streams": 1, "ranges": len(shards), "seconds": md5_s, "finished_on": "rank 0, host, from the ranges' PCM in rank order"},
            "device": device_description(torch, dev), "host": host_description(),
        }
    dec.close()
    return result


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE when a launcher set it, else 1")
    ap.add_argument("--config", choices=sorted(PRESETS), default="cfg3", help="named workload; explicit options override it")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--files", type=int, default=10000)
    ap.add_argument("--unique", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=60.0)
    # the defaults are the headline workload; the options below exist to time other stream shapes
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--bps", type=int, default=16)
    ap.add_argument("--blocksize", type=int, default=4096)
    ap.add_argument("--lpc-order", type=int, default=12, help="maximum LPC order")
    ap.add_argument("--force-lpc-order", type=int, default=0)
    ap.add_argument("--partition-order", type=int, default=6, help="maximum Rice partition order")
    ap.add_argument("--force-partition-order", type=int, default=-1)
    ap.add_argument("--cpu-files", type=int, default=256, help="files of the workload timed on the host CPU (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="the CPU sample is decoded repeatedly for about this long")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, cpu count)")
    ap.add_argument("--host-files", type=int, default=128, help="files decoded through the host-buffer entry point for the PCIe-inclusive rate (0 = skip)")
    ap.add_argument("--encode-steps", type=int, default=2, help="timed steps of the encode leg (0 = skip it)")
    ap.add_argument("--stereo-mode", type=int, default=-2,
                    help="encode leg: -2 stereo pair estimated per frame (two channels planned), -1 all four candidate "
                         "channels planned and the cheapest pair taken, 0..3 one pair for every frame")
    ap.add_argument("--shard", choices=("file", "range"), default="file",
                    help="file: every rank decodes its own batch of files (weak scaling).  range: ONE stream of --seconds, cut at "
                         "seek points into one frame range per rank (strong scaling); its MD5 is finished on rank 0")
    ap.add_argument("--seek-frames", type=int, default=16, help="--shard range: a seek point every this many frames")
    ap.add_argument("--cfg4-files", type=int, default=256, help="files of the cfg4 leg appended to a cfg3 line (0 = no cfg4 leg)")
    pre = ap.parse_known_args(argv)[0]
    ap.set_defaults(**PRESETS[pre.config])
    args = ap.parse_args(argv)

    args.gpus, problem = resolve_gpus(args.gpus, os.environ)
    if problem:
        print("bench.py: " + problem, file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import flac_amd
    import synth

    # FLAC_BENCH_REHEARSE=1: every rank on GPU 0 and gloo for the timing scalars, to run the multi-rank path
    # on a one-GPU box (RCCL refuses two ranks on one device); never set for a measurement
    rehearse = bool(os.environ.get("FLAC_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if rehearse else dev
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    env = {"rank": rank, "local_rank": local_rank, "world": world, "dev": dev, "red_dev": red_dev, "rehearse": rehearse}
    if args.shard == "range":
        out = run_range(args, env)
    else:
        out = run_workload(args, env)
        if args.config == "cfg3" and args.cfg4_files > 0:
            # the 24-bit 8-channel configuration as a second, shorter leg of the same line
            a4 = argparse.Namespace(**vars(args))
            for k, v in PRESETS["cfg4"].items():
                setattr(a4, k, v)
            a4.config, a4.files, a4.steps, a4.warmup = "cfg4", args.cfg4_files, min(args.steps, 3), 1
            a4.encode_steps, a4.host_files, a4.cpu_files, a4.cpu_seconds = 0, 0, (4 if args.cpu_files else 0), 5.0
            sub = run_workload(a4, env)
            if out is not None:
                out["configs"] = {"cfg4": {k: sub[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "pcm_MB_per_s",
                                                                "algorithmic_GB_per_s", "kernel_ms", "roofline", "cpu_baseline",
                                                                "md5_verified", "dtype")}}
    if out is not None:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
