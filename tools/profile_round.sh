#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (GPU box; run from the repo root via gpurun):
#   bash tools/profile_round.sh r2_cfg2 cfg2 --config 2
# leaves raw CSVs under gpurun_out/<tag>/ ; reduce them (again, in the build container) with
#   python tools/summarize_profiles.py gpurun_out/r3_cfg2 r03 cfg2
# The trace pass runs with --overlap 0: two batches in flight on two streams (bench.py's extra overlapped_steps measurement) stretch
# every kernel's duration in a trace.  Counters are collected in their own passes (no trace domains next to --pmc).  The traffic file is reduced on the box
# BEFORE the final bench run, so that the bench line of the same build carries `roofline.traffic`.
TAG=${1:-r2}; KEY=${2:-cfg1}; shift; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu-baseline --no-alt-plan --sustain 0 $*"     # (the traced passes time the product plan only)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH --steps 10 --overlap 0 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH --steps 2 --warmup 1 --profile-steps 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $BENCH --steps 2 --warmup 1 --profile-steps 1 > $OUT/pmc_write.log 2>&1
cd - > /dev/null
python tools/summarize_profiles.py $OUT ${ROUND:-r06} $KEY > $OUT/summarize.log 2>&1
python bench.py --kernel-table $* > $OUT/bench.json 2> $OUT/bench.err
tail -n 1 $OUT/bench.json | cut -c1-400
# keep only what summarize_profiles.py reads (the raw traces are hundreds of MB)
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
du -sh $OUT
