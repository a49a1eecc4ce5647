#!/bin/bash
# Counter passes over the bench (run on the GPU box from the repository root).  One pass per rocprofv3
# run; each run holds at most 8 SQ or 4 TCC counters.  Output: gpurun_out/pmc_<tag>/ and .txt summaries.
set -e
export TMPDIR=/tmp
ARGS="bench.py --files ${PMC_FILES:-2000} --unique ${PMC_UNIQUE:-256} --steps ${PMC_STEPS:-2} --warmup 1 --cpu-files 0 --host-files 0 --cfg4-files 0 --encode-steps ${PMC_ENC_STEPS:-0} ${PMC_EXTRA:-}"
pass() {
    tag=$1; shift
    timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -o p -- python3 $ARGS > gpurun_out/pmc_$tag.log 2>&1
    python3 profiles/pmc_summary.py gpurun_out/pmc_$tag/*counter_collection.csv > gpurun_out/pmc_$tag.txt; echo "pass $tag done"
}
for p in ${PMC_PASSES:-sq1 sq2 fetch write rdreq grbm}; do
    case $p in
        sq1) pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY ;;
        sq2) pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM ;;
        fetch) pass fetch FETCH_SIZE ;;
        write) pass write WRITE_SIZE ;;
        rdreq) pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_32B_sum ;;   # memory-side read requests by size (TCC_BUBBLE: the 128-byte ones)
        wrreq) pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum ;;
        skim) pass skim SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY ;;
        grbm) pass grbm GRBM_GUI_ACTIVE ;;
    esac
done
