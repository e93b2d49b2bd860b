#!/bin/bash
# PMC study of ONE grouped level of the product schedule (tools/pmc_level.py) -- run through gpurun:
#   bash tools/pmc_level.sh <kind> <tag>     kind = wino43 | wino23 | bf16rh | bf16ws [pmc_level.py args]   ->  gpurun_out/pmcl_<tag>_{a..e}/ + <tag>.txt
# counters in their own passes, no trace domains next to --pmc
export TMPDIR=/tmp
R=$PWD; KIND=${1:-wino43}; TAG=${2:-$KIND}; shift; shift; EXTRA="$*"     # further arguments go to pmc_level.py (--branches 0 ...)
OUT=$R/gpurun_out/pmcl_${TAG}.txt
python $R/tools/pmc_level.py --kind $KIND --iters 20 $EXTRA > $OUT 2>&1          # unprofiled timing first
run() { (cd /tmp && rocprofv3 --pmc $2 -d $R/gpurun_out/pmcl_${TAG}_$1 -o p -- python $R/tools/pmc_level.py --kind $KIND --iters 3 $EXTRA > $R/gpurun_out/pmcl_${TAG}_$1.log 2>&1); }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
run c "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ"
run d "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES"
run e "TCC_EA0_WRREQ TCC_EA0_RDREQ_DRAM TCC_TAG_STALL"
python $R/tools/pmc_read.py $R/gpurun_out/pmcl_${TAG}_a $R/gpurun_out/pmcl_${TAG}_b $R/gpurun_out/pmcl_${TAG}_c $R/gpurun_out/pmcl_${TAG}_d $R/gpurun_out/pmcl_${TAG}_e >> $OUT 2>&1
for p in a b c d e; do tail -n 1 $R/gpurun_out/pmcl_${TAG}_$p.log >> $OUT; done
find $R/gpurun_out/pmcl_${TAG}_? -name "*.db" -size +20M -delete
cat $OUT
