#!/usr/bin/env python3
"""profiles/filter_dispatches.py kernel_trace.csv OUT.csv [min_ms]: the dispatches of the decode-stage kernels
(k_scan_tiles, k_skim, k_decode<...>) that lasted at least min_ms (default 1): the full-size launches of a bench run,
one row each, so that the bench line's avg_launch_ms can be recomputed from the committed file."""
import csv
import sys

src, out = sys.argv[1], sys.argv[2]
min_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 1e6
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur >= min_ns and any(k in name for k in ("k_scan_tiles", "k_skim", "k_decode", "k_enc_")):
            rows.append((name, r["Dispatch_Id"], r["Start_Timestamp"], dur))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatch_id", "start_ns", "duration_ns"])
    w.writerows(rows)
by = {}
for name, _, _, dur in rows:
    by.setdefault(name, []).append(dur)
for name, d in sorted(by.items()):
    print(f"{name:50s} n={len(d):3d} avg={sum(d) / len(d) / 1e6:8.3f} ms  min={min(d) / 1e6:8.3f}  max={max(d) / 1e6:8.3f}")
