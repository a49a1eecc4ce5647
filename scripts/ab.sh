#!/bin/bash
# A/B runs of library variants (variants/lib_<name>.so, or "tree" for the in-tree build) on one GPU box:
# scripts/ab.sh [bench options --] name...   Stops at the first run that fails.
set -e
opts="--files 2000 --unique 128 --steps 3 --warmup 1 --cpu-files 0 --host-files 0 --cfg4-files 0 --encode-steps 0"
if [[ " $* " == *" -- "* ]]; then
    extra=""
    while [ "$1" != "--" ]; do extra="$extra $1"; shift; done
    shift
    opts="$opts $extra"
fi
for v in "$@"; do
    lib=$PWD/variants/lib_$v.so
    [ "$v" = tree ] && lib=$PWD/flac_amd/lib/libflacgpu.so
    FLACGPU_LIB=$lib timeout -k 10 400 python bench.py $opts > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.log
    python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/ab_{v}.json"))
print(v, "%.4g" % d["value"], {k: round(x, 2) for k, x in d["kernel_ms"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"], 3),
      round(d["roofline"]["pipeline_frac"], 3), flush=True)
PY
done
