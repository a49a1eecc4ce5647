#!/usr/bin/env python3
"""profiles/make_valu_json.py OUT.json WORKLOAD_TAG FILES BLOCKSIZE COMMIT sq1_summary.txt: vector instructions per
sample step of every kernel instantiation that ran (SQ_INSTS_VALU / SQ_WAVES / BLOCKSIZE: a wave of the decode
kernel steps through one block, one sample of its 64 subframes per step), for bench.py's roofline.valu."""
import json
import re
import sys

out, tag, files, bs, commit, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
rows = {}
for line in open(path):
    m = re.match(r"(?:void )?flacgpu::(\S.*?)\s{2,}(\w+)\s+dispatches=\s*(\d+) sum=(\S+) per_dispatch=(\S+)", line)
    if m:
        rows.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
kernels = {}
for k, c in rows.items():
    if c.get("SQ_INSTS_VALU", 0) > 0 and c.get("SQ_WAVES", 0) > 0 and k.startswith("k_decode"):
        kernels[k] = {"valu_per_sample_step": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / bs,
                      "salu_per_sample_step": c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"] / bs,
                      "lds_per_sample_step": c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"] / bs,
                      "wave_cycles_per_sample_step": 4 * c.get("SQ_WAVE_CYCLES", 0) / c["SQ_WAVES"] / bs}
json.dump({"workload_tag": tag, "files": files, "blocksize": bs, "commit": commit, "source": path,
           "command": "PMC_PASSES=sq1 bash profiles/collect_pmc.sh", "kernels": kernels}, open(out, "w"), indent=1)
print(out, {k: round(v["valu_per_sample_step"], 2) for k, v in kernels.items()})
