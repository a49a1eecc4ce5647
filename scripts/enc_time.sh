#!/bin/bash
# scripts/enc_time.sh [stereo modes...]: encode-leg timing of the in-tree library on one GPU box (2000 files), after the
# encoder's GPU tests; the bench itself checks the bytes of 8 files against the encoder oracle.
set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_encode.py -x -q --timeout 120 --timeout-method thread > gpurun_out/enc_tests.log 2>&1 || { tail -20 gpurun_out/enc_tests.log; exit 1; }
tail -1 gpurun_out/enc_tests.log
for m in "${@:--2}"; do
    timeout -k 10 300 python bench.py --files 2000 --unique 128 --steps 1 --warmup 1 --cpu-files 8 --cpu-seconds 4 --host-files 0 --encode-steps 3 --stereo-mode=$m > gpurun_out/enc_mode$m.json 2> gpurun_out/enc_mode$m.log
    python - "$m" <<'PY'
import json, sys
m = sys.argv[1]
d = json.load(open(f"gpurun_out/enc_mode{m}.json"))["encode"]
print(m, round(d["ms_per_step"], 2), round(d["ratio"], 5), {k: round(v, 2) for k, v in d["kernel_ms"].items()}, flush=True)
PY
done
