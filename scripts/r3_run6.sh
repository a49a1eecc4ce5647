set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not bench_launch" > gpurun_out/r3_gputest5.log 2>&1 || { tail -n 40 gpurun_out/r3_gputest5.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest5.log
scripts/ab.sh notf tree notf tree
scripts/ab.sh --blocksize 576 --files 1000 -- notf tree
scripts/ab.sh --config cfg4 --files 256 --unique 16 -- notf tree
