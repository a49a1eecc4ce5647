#!/bin/bash
# samples the board power and clocks while the full-size decode loop runs (what holds the clock under load)
set -o pipefail
mkdir -p gpurun_out
log=gpurun_out/r3_power_samples.log
: > $log
timeout -k 10 20 rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/r3_power_idle.log 2>&1 || true
timeout -k 10 280 python -u bench.py --steps 150 --warmup 3 --encode-steps 0 --cfg4-files 0 --cpu-files 0 > gpurun_out/r3_power_bench.json 2> gpurun_out/r3_power_bench.log &
pid=$!
for i in $(seq 1 120); do
    kill -0 $pid 2>/dev/null || break
    echo "t=$(date +%s.%N)" >> $log
    timeout -k 5 10 rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $log || true
    sleep 0.3
done
wait $pid; rc=$?
echo "bench rc=$rc"
grep -c "t=" $log
grep -E "Power" $log | sort | uniq -c | sort -k1,1nr | head -n 12
grep -E "sclk" $log | sort | uniq -c | sort -k1,1nr | head -n 12
tail -n 5 gpurun_out/r3_power_idle.log
exit $rc
