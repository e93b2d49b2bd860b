#!/bin/bash
# PMC passes over one shape of the split-fp32 conv harness (tools/f32x3_ws.hip) -- run through gpurun:
#   bash tools/pmc_x3.sh <binary> <tag> B H W C N NS
export TMPDIR=/tmp
R=$PWD; BIN=$R/$1; TAG=$2; shift; shift; ARGS="$*"
OUT=$R/gpurun_out/pmcx_${TAG}.txt
$BIN one $ARGS 20 > $OUT 2>&1
run() { (cd /tmp && rocprofv3 --pmc $2 -d $R/gpurun_out/pmcx_${TAG}_$1 -o p -- $BIN one $ARGS 3 > $R/gpurun_out/pmcx_${TAG}_$1.log 2>&1); }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
python $R/tools/pmc_read.py $R/gpurun_out/pmcx_${TAG}_a $R/gpurun_out/pmcx_${TAG}_b >> $OUT 2>&1
find $R/gpurun_out/pmcx_${TAG}_? -name "*.db" -size +20M -delete
cat $OUT
