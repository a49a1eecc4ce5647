#!/bin/bash
# decode time and clock of probe variants side by side (full-size cfg3): scripts/r3_clock_ab.sh name...
set -e
for v in "$@"; do
    FLAC_BENCH_NOCHECK=1 FLACGPU_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --cpu-files 0 --host-files 0 --cfg4-files 0 --encode-steps 0 > gpurun_out/clkab_$v.out 2> gpurun_out/clkab_$v.log || { tail -n 5 gpurun_out/clkab_$v.log; }
    python3 - "$v" <<'PY'
import collections, json, re, sys
v = sys.argv[1]
acc = collections.defaultdict(list); line = None
for l in open(f"gpurun_out/clkab_{v}.out"):
    m = re.match(r"clockprobe (\w+) block (\d+) cycles (\d+) ticks (\d+)", l)
    if m and int(m.group(4)) > 0: acc[m.group(1)].append(100.0 * int(m.group(3)) / int(m.group(4)))
    elif l.startswith("{"): line = json.loads(l)
km = line["kernel_ms"] if line else None
print(v, {k: round(sorted(x)[len(x) // 2]) for k, x in acc.items()}, km and {k: round(km[k], 2) for k in ("scan_ms", "skim_ms", "decode_ms")}, flush=True)
PY
done
