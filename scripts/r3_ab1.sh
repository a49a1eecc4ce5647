#!/bin/bash
# round-3 A/B of decode-kernel variants (run on the GPU box from the repository root)
set -e
export TMPDIR=/tmp
FLACGPU_LIB=$PWD/variants/lib_all3.so timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_ab1_tests_all3.log 2>&1
tail -n 3 gpurun_out/r3_ab1_tests_all3.log
scripts/ab.sh tree chain1 minmax constlane all3 tree
FLAC_BENCH_NOCHECK=1 scripts/ab.sh --channels 1 -- tree
mv gpurun_out/ab_tree.json gpurun_out/ab_mono_tree.json
FLAC_BENCH_NOCHECK=1 scripts/ab.sh --channels 1 -- nopred
export FLACGPU_LIB=$PWD/variants/lib_all3.so
PMC_PASSES=sq1 profiles/collect_pmc.sh && cp gpurun_out/pmc_sq1.txt gpurun_out/r3_ab1_pmc_sq1_all3.txt
export FLACGPU_LIB=$PWD/variants/lib_nopred.so
FLAC_BENCH_NOCHECK=1 PMC_EXTRA="--channels 1" PMC_PASSES=sq1 profiles/collect_pmc.sh && cp gpurun_out/pmc_sq1.txt gpurun_out/r3_ab1_pmc_sq1_nopred_mono.txt
