set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not bench_launch" > gpurun_out/r3_gputest3.log 2>&1 || { tail -n 40 gpurun_out/r3_gputest3.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest3.log
for v in clamped tree clamped tree; do
  scripts/ab.sh --config cfg4 --files 128 --unique 16 -- $v
done
scripts/ab.sh --config cfg4 --files 512 --unique 16 -- clamped tree
for v in clamped tree; do
  lib=$PWD/variants/lib_$v.so; [ "$v" = tree ] && lib=$PWD/flac_amd/lib/libflacgpu.so
  FLACGPU_LIB=$lib timeout -k 10 300 python bench.py --config cfg4 --shard range --seconds 600 --steps 5 --warmup 1 > gpurun_out/r3_c_range_$v.json 2> gpurun_out/r3_c_range_$v.log
  python -c "
import json,sys
d=json.load(open('gpurun_out/r3_c_range_$v.json')); print('$v range', '%.4g'%d['value'], {k:round(x,2) for k,x in d['kernel_ms'].items()})"
done
scripts/ab.sh clamped tree
