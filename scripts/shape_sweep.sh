set -e
run() { name=$1; shift; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-files 0 --host-files 0 --encode-steps 0 "$@" > gpurun_out/shape_$name.json 2> gpurun_out/shape_$name.log; python -c "
import json,sys
d=json.load(open('gpurun_out/shape_$name.json')); print('$name', '%.3g'%d['value'], {k:round(v,2) for k,v in d['kernel_ms'].items()}, round(d['roofline']['frac'],3))"; }
run mono16 --files 2000 --unique 64 --channels 1
run stereo24 --files 1000 --unique 64 --channels 2 --bps 24 --rate 96000 --blocksize 4608 --seconds 30
run six16 --files 600 --unique 32 --channels 6 --rate 48000
run stereo8 --files 2000 --unique 64 --bps 8
run stereo16bs1152 --files 2000 --unique 64 --blocksize 1152 --lpc-order 8
