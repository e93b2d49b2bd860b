#!/bin/bash
# PMC study of one conv shape of tools/bench_conv.py (run on the GPU box through gpurun).
# usage: tools/pmc_conv.sh <shape index> <tag>     -> gpurun_out/pmc_<tag>_{a,b,c,d}/
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
IDX=$1; TAG=$2
run() { rocprofv3 --pmc $2 -d gpurun_out/pmc_${TAG}_$1 -o p -- python tools/bench_conv.py --only $IDX --iters 3 > gpurun_out/pmc_${TAG}_$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"
run b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
run c "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ"
run d "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE TA_TA_BUSY"
run e "TCC_EA0_WRREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM TCC_TAG_STALL"
