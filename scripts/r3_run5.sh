set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "not bench_launch" > gpurun_out/r3_gputest4.log 2>&1 || { tail -n 40 gpurun_out/r3_gputest4.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest4.log
scripts/ab.sh scan4 tree scan1 scan4 tree scan1
scripts/ab.sh --blocksize 1152 --lpc-order 8 -- scan4 tree scan1
scripts/ab.sh --blocksize 576 --files 1000 -- scan4 tree scan1
scripts/ab.sh --config cfg4 --files 256 --unique 16 -- scan4 tree scan1
