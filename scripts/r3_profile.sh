#!/bin/bash
# Round-3 profile collection (GPU box, repository root): counter passes of cfg3 at 2000 files and cfg4 at 128 files,
# then one rocprofv3 --kernel-trace --stats run of the default bench command.  TAG=r3_x names the output files.
set -e
export TMPDIR=/tmp
TAG=${TAG:-r3_a}
rocprofv3 -L > gpurun_out/${TAG}_counters_list.txt 2>&1 || true
grep -c . gpurun_out/${TAG}_counters_list.txt
PMC_PASSES="${CFG3_PASSES:-sq1 sq2 fetch write rdreq wrreq}" profiles/collect_pmc.sh
for p in ${CFG3_PASSES:-sq1 sq2 fetch write rdreq wrreq}; do cp gpurun_out/pmc_$p.txt gpurun_out/${TAG}_pmc_${p}_2000files.txt; done
PMC_FILES=128 PMC_UNIQUE=16 PMC_EXTRA="--config cfg4" PMC_PASSES="${CFG4_PASSES:-sq1 fetch write rdreq wrreq skim}" profiles/collect_pmc.sh
for p in ${CFG4_PASSES:-sq1 fetch write rdreq wrreq skim}; do cp gpurun_out/pmc_$p.txt gpurun_out/${TAG}_pmc_${p}_cfg4_128files.txt; done
if [ -z "$NO_FULL" ]; then
    rm -rf gpurun_out/prof_full
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_full -o full -- python3 bench.py > gpurun_out/${TAG}_full_bench.json 2> gpurun_out/${TAG}_full_bench.log
    cp gpurun_out/prof_full/*kernel_stats.csv gpurun_out/${TAG}_full_kernel_stats.csv
    python3 profiles/filter_dispatches.py gpurun_out/prof_full/*kernel_trace.csv gpurun_out/${TAG}_full_dispatches.csv 1 | tee gpurun_out/${TAG}_full_dispatches.txt
    rm -rf gpurun_out/prof_full
fi
echo profile done
