#!/usr/bin/env python
"""Print per-kernel PMC sums from rocprofv3 rocpd databases: tools/pmc_read.py <dir> [<dir> ...]"""
import glob, sqlite3, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(duration) from counters_collection "
                         "where kernel_name like '%capf%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
        if not rows:
            continue
        last = rows[-1][2]                      # report the last dispatch (steady state)
        print(f"== {d}: {rows[-1][0][:70]}  duration {rows[-1][4]/1e3:.1f} us")
        for r in rows:
            if r[2] == last:
                print(f"   {r[1]:34s} {r[3]:16.0f}")
