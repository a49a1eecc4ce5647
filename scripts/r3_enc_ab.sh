#!/bin/bash
# encode-leg timing of library variants (no byte checks: for ablations whose output is wrong): scripts/r3_enc_ab.sh name...
set -e
for v in "$@"; do
    lib=$PWD/variants/lib_$v.so
    [ "$v" = tree ] && lib=$PWD/flac_amd/lib/libflacgpu.so
    FLAC_BENCH_NOCHECK=1 FLACGPU_LIB=$lib timeout -k 10 300 python bench.py --files 2000 --unique 128 --steps 1 --warmup 1 --cpu-files 0 --host-files 0 --cfg4-files 0 --encode-steps 3 > gpurun_out/encab_$v.json 2> gpurun_out/encab_$v.log
    python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/encab_{v}.json"))["encode"]
print(v, round(d["ms_per_step"], 2), {k: round(x, 2) for k, x in d["kernel_ms"].items()}, flush=True)
PY
done
