set -e
timeout -k 10 900 python -u -m pytest tests/test_bench_launch.py -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3_benchtests.log 2>&1 || { tail -n 40 gpurun_out/r3_benchtests.log; exit 1; }
tail -n 3 gpurun_out/r3_benchtests.log
TAG=r3_a scripts/r3_profile.sh > gpurun_out/r3_a_profile.log 2>&1 || { tail -n 40 gpurun_out/r3_a_profile.log; exit 1; }
tail -n 20 gpurun_out/r3_a_profile.log
