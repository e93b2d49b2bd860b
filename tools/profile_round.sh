#!/bin/bash
# rocprofv3 evidence for one round (GPU box; run from the repo root via gpurun):
#   bash tools/profile_round.sh r1c [extra bench.py flags]
# leaves raw CSVs under gpurun_out/<tag>/ ; reduce them with  python tools/summarize_profiles.py gpurun_out/<tag> r01
# Counters are collected in their own passes (no trace domains next to --pmc).
TAG=${1:-r1}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH --steps 10 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes0 -o bench -- $BENCH --steps 10 --lanes 0 > $OUT/trace_lanes0.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH --steps 2 --warmup 1 --profile-steps 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $BENCH --steps 2 --warmup 1 --profile-steps 1 > $OUT/pmc_write.log 2>&1
cd - > /dev/null
python bench.py $* > $OUT/bench.json 2> $OUT/bench.err
python bench.py --no-cpu-baseline --lanes 0 $* > $OUT/bench_lanes0.json 2>> $OUT/bench.err
tail -n 1 $OUT/bench.json | cut -c1-300
ls -la $OUT $OUT/trace | head -30
