set -e
timeout -k 10 560 python -u scripts/soak_decode.py ${SEED:-900000} ${N_DEC:-1500} > gpurun_out/r3_soak_decode_${SEED:-900000}.log 2>&1 || { tail -n 5 gpurun_out/r3_soak_decode_${SEED:-900000}.log; echo "decode soak: stopped (time limit or failure)"; }
tail -n 2 gpurun_out/r3_soak_decode_${SEED:-900000}.log
timeout -k 10 300 python -u scripts/soak_corrupt.py ${SEED:-900000} ${N_COR:-600} > gpurun_out/r3_soak_corrupt_${SEED:-900000}.log 2>&1 || { tail -n 5 gpurun_out/r3_soak_corrupt_${SEED:-900000}.log; echo "corrupt soak: stopped"; }
tail -n 2 gpurun_out/r3_soak_corrupt_${SEED:-900000}.log
timeout -k 10 200 python -u scripts/soak_encode.py ${SEED:-900000} ${N_ENC:-400} > gpurun_out/r3_soak_encode_${SEED:-900000}.log 2>&1 || { tail -n 5 gpurun_out/r3_soak_encode_${SEED:-900000}.log; echo "encode soak: stopped"; }
tail -n 2 gpurun_out/r3_soak_encode_${SEED:-900000}.log
