#!/bin/bash
# Board power and shader clock WHILE one HRNet-32 level (the four branch convs as one grid, tools/f32h2_ws.hip `level` mode) runs back to back
# for a few seconds -- the telemetry VERDICT r4 asked for next to the in-launch clock64() readings.  Run through gpurun:
#   bash tools/power_level.sh <binary> <tag>       ->  gpurun_out/power_<tag>.txt
R=$PWD; BIN=$R/${1:-tools/ab/f32h2_ws}; TAG=${2:-h2}
OUT=$R/gpurun_out/power_${TAG}.txt
SMI=$(command -v rocm-smi || echo /opt/rocm/bin/rocm-smi)
{
echo "== idle"; $SMI --showpower --showclocks 2>&1 | grep -iE "power|sclk|mclk" | head -6
for B in 64 512; do
    REPS=$([ $B = 64 ] && echo 40000 || echo 5000)
    echo "== level batch $B, $REPS launches back to back"
    $BIN level $B 0 0 $REPS > $R/gpurun_out/power_${TAG}_b$B.log 2>&1 &
    PID=$!
    # (the harness first spends seconds filling host buffers: sample until it exits, keep the samples taken under load)
    while kill -0 $PID 2>/dev/null; do
        $SMI --showpower --showclocks 2>&1 | grep -iE "Power \(W\)|sclk" | sed -e 's/.*sclk clock level: [0-9]*: (\([0-9]*Mhz\)).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/power \1 W/' | tr '\n' ' '; echo
        sleep 0.4
    done | awk '$4 + 0 > 600' | tail -n 8
    wait $PID
    grep "^level" $R/gpurun_out/power_${TAG}_b$B.log
done
} > $OUT 2>&1
cat $OUT
