set -e
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_gputest2.log 2>&1 || { tail -n 40 gpurun_out/r3_gputest2.log; exit 1; }
tail -n 3 gpurun_out/r3_gputest2.log
PMC_PASSES="rdreq wrreq" profiles/collect_pmc.sh
cp gpurun_out/pmc_rdreq.txt gpurun_out/r3_b_pmc_rdreq_2000files.txt; cp gpurun_out/pmc_wrreq.txt gpurun_out/r3_b_pmc_wrreq_2000files.txt
PMC_FILES=128 PMC_UNIQUE=16 PMC_EXTRA="--config cfg4" PMC_PASSES="rdreq" profiles/collect_pmc.sh
cp gpurun_out/pmc_rdreq.txt gpurun_out/r3_b_pmc_rdreq_cfg4_128files.txt
scripts/ab.sh tree
timeout -k 10 300 python bench.py --config cfg4 --shard range --seconds 600 --steps 3 --warmup 1 > gpurun_out/r3_b_range_cfg4_600s.json 2> gpurun_out/r3_b_range_cfg4_600s.log
tail -n 3 gpurun_out/r3_b_range_cfg4_600s.log
