#!/bin/bash
# clock the chip holds under each kernel: the probe variant prints cycle counter vs constant-rate counter per sampled workgroup
set -o pipefail
mkdir -p gpurun_out
common="--steps 2 --warmup 1 --cpu-files 0 --host-files 0 --cfg4-files 0 --encode-steps 0"
FLACGPU_LIB=$PWD/variants/lib_clock.so timeout -k 10 300 python -u bench.py $common > gpurun_out/r3_clock_cfg3.out 2> gpurun_out/r3_clock_cfg3.log || { tail -n 20 gpurun_out/r3_clock_cfg3.log; exit 1; }
tag3=$(python3 -c "import json,sys; print([json.loads(l) for l in open('gpurun_out/r3_clock_cfg3.out') if l.startswith('{')][-1]['config']['workload_tag'])")
python3 profiles/make_clock_json.py gpurun_out/r3_d_clock_cfg3.json "$tag3" 10000 gpurun_out/r3_clock_cfg3.out || exit 1
FLACGPU_LIB=$PWD/variants/lib_clock.so timeout -k 10 300 python -u bench.py --config cfg4 --files 256 $common > gpurun_out/r3_clock_cfg4.out 2> gpurun_out/r3_clock_cfg4.log || { tail -n 20 gpurun_out/r3_clock_cfg4.log; exit 1; }
tag4=$(python3 -c "import json,sys; print([json.loads(l) for l in open('gpurun_out/r3_clock_cfg4.out') if l.startswith('{')][-1]['config']['workload_tag'])")
python3 profiles/make_clock_json.py gpurun_out/r3_d_clock_cfg4.json "$tag4" 256 gpurun_out/r3_clock_cfg4.out || exit 1
