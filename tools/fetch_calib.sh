#!/bin/bash
# rocprofv3 counter passes over tools/ab/fetch_calib (GPU box, via gpurun): writes gpurun_out/fetch_calib.txt
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/fetch_calib.txt
$R/tools/ab/fetch_calib > $OUT 2>&1
for C in "FETCH_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_DRAM" "TCC_REQ TCC_HIT TCC_MISS" "WRITE_SIZE"; do
  T=$(echo $C | tr ' ' '_')
  (cd /tmp && rocprofv3 --pmc $C -d $R/gpurun_out/fc_$T -o p -- $R/tools/ab/fetch_calib > $R/gpurun_out/fc_$T.log 2>&1)
  echo "== --pmc $C" >> $OUT
  python3 - "$R/gpurun_out/fc_$T" >> $OUT 2>&1 <<'PY'
import glob, sqlite3, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by dispatch_id, counter_name order by dispatch_id").fetchall()
    for k, n, d, v in rows:
        print(f"   dispatch {d:3d} {k[:40]:40s} {n:24s} {v:16.0f}")
PY
  rm -rf $R/gpurun_out/fc_$T $R/gpurun_out/fc_$T.log
done
cat $OUT
