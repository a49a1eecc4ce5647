#!/usr/bin/env python3
"""profiles/make_traffic_json.py OUT.json WORKLOAD_TAG FILES summary.txt...: gathers the per-dispatch FETCH_SIZE /
WRITE_SIZE (and SQ) figures of pmc_summary.py outputs into the json bench.py's lookup_traffic reads.  WORKLOAD_TAG
is bench.workload_tag() of the run the counters were taken on."""
import json
import re
import sys

out, tag, files = sys.argv[1], sys.argv[2], int(sys.argv[3])
names = {"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB", "TCC_EA0_RDREQ_sum": "TCC_EA0_RDREQ",
         "TCC_EA0_RDREQ_32B_sum": "TCC_EA0_RDREQ_32B", "TCC_EA0_WRREQ_sum": "TCC_EA0_WRREQ", "TCC_EA0_WRREQ_64B_sum": "TCC_EA0_WRREQ_64B", "TCC_BUBBLE_sum": "TCC_BUBBLE",
         "TCC_EA0_RDREQ_DRAM_32B_sum": "TCC_EA0_RDREQ_DRAM_32B", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum": "TCC_EA0_WRREQ_WRITE_DRAM_32B"}
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
kernels = {k: v for k, v in kernels.items() if v.get("WRITE_SIZE_KiB", 0) > 1 or v.get("FETCH_SIZE_KiB", 0) > 1
           or v.get("TCC_EA0_RDREQ_DRAM_32B", 0) > 32 or v.get("TCC_EA0_WRREQ_WRITE_DRAM_32B", 0) > 32}
import subprocess
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
json.dump({"workload_tag": tag, "files": files, "commit": commit,
           "command": "bash profiles/collect_pmc.sh (one rocprofv3 --pmc pass per counter group)",
           "note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  On gfx950 FETCH_SIZE counts a 128-byte request of a wide "
                   "coalesced read as 64 bytes (k_scan_tiles reads every input byte once: 2 x its FETCH_SIZE equals the "
                   "input size); for the per-lane 32-byte sector reads of k_skim / k_decode the doubled figure is an upper "
                   "bound, and the request-size split (TCC_BUBBLE requests of 128 bytes, TCC_EA0_RDREQ_32B of 32, the other "
                   "TCC_EA0_RDREQ requests of 64: the expression rocprofv3 itself lists for FETCH_SIZE on this chip) is what "
                   "bench.py uses for the read side when the profile has that pass; TCC_EA0_RDREQ_DRAM_32B x 32 is the same "
                   "quantity counted in 32-byte units.",
           "kernels": kernels}, open(out, "w"), indent=1)
print(out, sorted(kernels))
