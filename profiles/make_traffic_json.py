#!/usr/bin/env python3
"""profiles/make_traffic_json.py OUT.json WORKLOAD_TAG FILES summary.txt...: gathers the per-dispatch FETCH_SIZE /
WRITE_SIZE (and SQ) figures of pmc_summary.py outputs into the json bench.py's lookup_traffic reads.  WORKLOAD_TAG
is bench.workload_tag() of the run the counters were taken on."""
import json
import re
import sys

out, tag, files = sys.argv[1], sys.argv[2], int(sys.argv[3])
names = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB"}
kernels = {}
for path in sys.argv[4:]:
    for line in open(path):
        m = re.match(r"(?:void )?flacgpu::(\S.*?)\s{2,}(\w+)\s+dispatches=\s*\d+ sum=\S+ per_dispatch=(\S+)", line)
        if not m:
            continue
        k, c, v = m.group(1).strip(), m.group(2), float(m.group(3))
        if v == 0 and not k.startswith("k_decode"):
            continue
        kernels.setdefault(k, {})[names.get(c, c)] = v
kernels = {k: v for k, v in kernels.items() if v.get("WRITE_SIZE_KiB", 0) > 1 or v.get("FETCH_SIZE_KiB", 0) > 1}
json.dump({"workload_tag": tag, "files": files,
           "command": "bash profiles/collect_pmc.sh (one rocprofv3 --pmc pass per counter group)",
           "note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  On gfx950 FETCH_SIZE counts a 128-byte request of a wide "
                   "coalesced read as 64 bytes (k_scan_tiles reads every input byte once: 2 x its FETCH_SIZE equals the "
                   "input size), so bench.py doubles the read side; for the per-lane 32-byte sector reads of k_skim / "
                   "k_decode the doubled figure is an upper bound.",
           "kernels": kernels}, open(out, "w"), indent=1)
print(out, sorted(kernels))
