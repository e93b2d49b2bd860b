#!/bin/bash
# PMC study of ONE grouped launch (the four HRNet-W32 stage-4 branches at batch 64) -- run through gpurun:
#   bash tools/pmc_group.sh <tag>   ->  gpurun_out/pmcg_<tag>_{a..e}/ ; read with tools/pmc_read.py
# counters in their own passes, no trace domains next to --pmc
export TMPDIR=/tmp
R=$PWD; TAG=${1:-g}
run() { (cd /tmp && rocprofv3 --pmc $2 -d $R/gpurun_out/pmcg_${TAG}_$1 -o p -- python $R/tools/bench_concurrent.py --only-grouped --iters 3 > $R/gpurun_out/pmcg_${TAG}_$1.log 2>&1); }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"
run b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
run c "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ"
run d "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE TA_TA_BUSY"
run e "TCC_EA0_WRREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM TCC_TAG_STALL"
python tools/pmc_read.py gpurun_out/pmcg_${TAG}_a gpurun_out/pmcg_${TAG}_b gpurun_out/pmcg_${TAG}_c gpurun_out/pmcg_${TAG}_d gpurun_out/pmcg_${TAG}_e
