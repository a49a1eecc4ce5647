#!/bin/bash
# encoder change check: encode tests, encode soak, then the bench line with the encode leg only (bytes are checked against the oracle there)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_encode.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_gputest_enc.log 2>&1 || { tail -n 30 gpurun_out/r3_gputest_enc.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest_enc.log
timeout -k 10 300 python -u scripts/soak_encode.py ${SEED:-990000} ${N_ENC:-2000} > gpurun_out/r3_soak_encode_${SEED:-990000}.log 2>&1 || { tail -n 5 gpurun_out/r3_soak_encode_${SEED:-990000}.log; exit 1; }
tail -n 2 gpurun_out/r3_soak_encode_${SEED:-990000}.log
timeout -k 10 300 python -u bench.py --cfg4-files 0 --steps 3 --warmup 1 > gpurun_out/r3_bench_enc.json 2> gpurun_out/r3_bench_enc.log || { tail -n 20 gpurun_out/r3_bench_enc.log; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3_bench_enc.json").read().strip().splitlines()[-1])
e = d.get("encode") or {}
print({k: e.get(k) for k in ("value", "ms_per_step", "kernel_ms")})
PY
