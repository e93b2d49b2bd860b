#!/bin/bash
# Counters of the LAST dispatch of one kernel inside a forward of the product schedule (GPU box, via gpurun):
#   bash tools/pmc_kernel.sh <kernel name pattern> <tag> [launch_table.py args]      -> gpurun_out/pmck_<tag>.txt
#   e.g.  bash tools/pmc_kernel.sh bneck0_bf16 r06_bneck0 --batch 128 --backbone cpn --dtype bf16 --height 384 --width 288
# counters in their own passes, no trace domains next to --pmc
export TMPDIR=/tmp
R=$PWD; PAT=$1; TAG=$2; shift; shift
OUT=$R/gpurun_out/pmck_${TAG}.txt
python $R/tools/launch_table.py $* 2>&1 | grep -E "total|$(echo $PAT | cut -c1-12)" | cut -c1-150 > $OUT
run() { (cd /tmp && rocprofv3 --pmc $2 -d $R/gpurun_out/pmck_${TAG}_$1 -o p -- python $R/tools/launch_table.py --reps 1 $ARGS > $R/gpurun_out/pmck_${TAG}_$1.log 2>&1); }
ARGS="$*"
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
python - "$PAT" $R/gpurun_out/pmck_${TAG}_a $R/gpurun_out/pmck_${TAG}_b >> $OUT 2>&1 <<'PY'
import glob, sqlite3, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                         "where kernel_name like ? group by dispatch_id, counter_name order by dispatch_id", ("%" + pat + "%",)).fetchall()
        if not rows:
            continue
        last = rows[-1][2]
        print(f"== {d.split('/')[-1]}: {rows[-1][0][:70]}  duration under counters {rows[-1][4]/1e3:.1f} us")
        for r in rows:
            if r[2] == last:
                print(f"   {r[1]:34s} {r[3]:16.0f}")
PY
find $R/gpurun_out/pmck_${TAG}_? -name "*.db" -delete
cat $OUT
