#!/bin/bash
# encoder change check: encode tests, encode soak, then the default bench line (its encode leg checks bytes against the oracle)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_encode.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gputest7.log 2>&1 || { tail -n 30 gpurun_out/r3_gputest7.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest7.log
timeout -k 10 300 python -u scripts/soak_encode.py 970000 ${N_ENC:-3000} > gpurun_out/r3_soak_encode_970000.log 2>&1 || { tail -n 5 gpurun_out/r3_soak_encode_970000.log; exit 1; }
tail -n 2 gpurun_out/r3_soak_encode_970000.log
timeout -k 10 300 python -u bench.py > gpurun_out/r3_bench_d.json 2> gpurun_out/r3_bench_d.log || { tail -n 20 gpurun_out/r3_bench_d.log; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_d.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("valu"), d["roofline"].get("traffic"))
e = d.get("encode") or {}
print({k: e.get(k) for k in ("value", "ms_per_step", "kernel_ms")})
PY
