#!/bin/bash
# what the driver runs at the end of a round, on the tree as it stands: GPU suite, smoke(), the default bench line
set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3_final_gputest.log 2>&1 || { tail -n 40 gpurun_out/r3_final_gputest.log; exit 1; }
tail -n 2 gpurun_out/r3_final_gputest.log
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout -k 10 400 python -u bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.log
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_final_bench.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["pipeline_frac"])
print(d["kernel_ms"]); print(d["encode"]["ms_per_step"], d["configs"]["cfg4"]["ms_per_step"])
PY
