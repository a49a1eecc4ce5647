#!/bin/bash
# scan change check: decode + corrupt-input GPU tests, a short corrupt soak, then the bench line without the encode and cfg4 legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "not bench_launch" > gpurun_out/r3_gputest8.log 2>&1 || { tail -n 30 gpurun_out/r3_gputest8.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest8.log
timeout -k 10 200 python -u scripts/soak_corrupt.py 980000 ${N_COR:-3000} > gpurun_out/r3_soak_corrupt_980000.log 2>&1 || { tail -n 5 gpurun_out/r3_soak_corrupt_980000.log; exit 1; }
tail -n 2 gpurun_out/r3_soak_corrupt_980000.log
timeout -k 10 300 python -u bench.py --encode-steps 0 --cfg4-files 0 --cpu-files 0 > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.log || { tail -n 20 gpurun_out/r3_bench_e.log; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_e.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(d.get("kernel_ms") or d.get("timing") or {k: v for k, v in d.items() if "ms" in k})
PY
