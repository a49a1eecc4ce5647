#!/bin/bash
# final library: full GPU suite, then the round's profile set (counter passes + kernel trace of the default bench)
set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3_gputest9.log 2>&1 || { tail -n 40 gpurun_out/r3_gputest9.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest9.log
TAG=r3_d bash scripts/r3_profile.sh
