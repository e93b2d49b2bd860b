#!/bin/bash
# Counters of ONE HRNet-32 level launch of a tools/f32h2_ws.hip build (the product tile or an H2_EXPERIMENT knock-out), batch 64:
#   bash tools/pmc_harness.sh tools/ab/f32h2_ws product ; bash tools/pmc_harness.sh tools/ab/f32h2_ex1 nosplit     -> gpurun_out/pmch_<tag>.txt
# counters in their own passes, no trace domains next to --pmc
export TMPDIR=/tmp
R=$PWD; BIN=$R/$1; TAG=$2; B=${3:-64}
OUT=$R/gpurun_out/pmch_${TAG}.txt
$BIN level $B 0 0 300 | tail -1 > $OUT 2>&1
run() { (cd /tmp && rocprofv3 --pmc $2 -d $R/gpurun_out/pmch_${TAG}_$1 -o p -- $BIN level $B 0 0 4 > $R/gpurun_out/pmch_${TAG}_$1.log 2>&1); }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
python - $R/gpurun_out/pmch_${TAG}_a $R/gpurun_out/pmch_${TAG}_b >> $OUT 2>&1 <<'PY'
import glob, sqlite3, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                         "where kernel_name like '%lvl_kernel%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
        if not rows:
            continue
        last = rows[-1][2]
        print(f"== {d.split('/')[-1]}: {rows[-1][0][:60]}  duration under counters {rows[-1][4]/1e3:.1f} us")
        for r in rows:
            if r[2] == last:
                print(f"   {r[1]:34s} {r[3]:16.0f}")
PY
find $R/gpurun_out/pmch_${TAG}_? -name "*.db" -size +20M -delete
cat $OUT
