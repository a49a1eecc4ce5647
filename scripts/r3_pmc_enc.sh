#!/bin/bash
# counter passes with the encode leg on (2 000 files): instruction mix and memory-side requests of the encoder kernels
set -o pipefail
mkdir -p gpurun_out
PMC_ENC_STEPS=1 PMC_STEPS=1 PMC_PASSES="${PASSES:-sq1 sq2 rdreq wrreq}" bash profiles/collect_pmc.sh || exit 1
for p in ${PASSES:-sq1 sq2 rdreq wrreq}; do cp gpurun_out/pmc_$p.txt gpurun_out/r3_d_enc_pmc_${p}_2000files.txt; done
grep -h "k_enc" gpurun_out/r3_d_enc_pmc_sq1_2000files.txt | cut -c1-200
