set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3_gputest6.log 2>&1 || { tail -n 40 gpurun_out/r3_gputest6.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest6.log
SEED=960000 N_DEC=9000 N_COR=9000 N_ENC=2000 bash scripts/r3_soak.sh
