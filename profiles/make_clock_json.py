#!/usr/bin/env python3
"""profiles/make_clock_json.py OUT.json WORKLOAD_TAG FILES bench_stdout: gathers the `clockprobe` lines a probe build of
the library prints (cycle counter against the 100 MHz constant-rate counter over the lifetime of sampled workgroups)
into the json bench.py's lookup_clock reads: per kernel the median, lowest and highest clock in MHz."""
import collections
import json
import re
import subprocess
import sys

out, tag, files, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
acc = collections.defaultdict(list)
for line in open(path):
    m = re.match(r"clockprobe (\w+) block (\d+) cycles (\d+) ticks (\d+)", line)
    if m and int(m.group(4)) > 0:
        acc[m.group(1)].append((int(m.group(3)), int(m.group(4))))
kernels = {}
for k, v in acc.items():
    mhz = sorted(100.0 * c / t for c, t in v)
    ticks = sorted(t for _, t in v)
    kernels[k] = {"samples": len(v), "clock_MHz_median": mhz[len(mhz) // 2], "clock_MHz_min": mhz[0], "clock_MHz_max": mhz[-1],
                  "median_workgroup_lifetime_us": ticks[len(ticks) // 2] / 100.0}
run = None
for line in open(path):
    if line.startswith("{"):
        try:
            d = json.loads(line)
            run = {"kernel_ms": d.get("kernel_ms"), "files_per_gpu": d.get("config", {}).get("files_per_gpu")}
        except ValueError:
            pass
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
json.dump({"workload_tag": tag, "files": files, "commit": commit,
           "command": "bash scripts/r3_clock.sh (library built with -DFLACGPU_CLOCK_PROBE)",
           "note": "clock = cycle counter (s_memtime) delta / constant-rate counter (s_memrealtime, 100 MHz) delta over the "
                   "lifetime of every n-th workgroup of the kernel; scan = k_scan_tiles, skim = k_skim, decode = k_decode<...>",
           "same_run": run, "kernels": kernels}, open(out, "w"), indent=1)
print(out, {k: round(v["clock_MHz_median"]) for k, v in kernels.items()})
