#!/usr/bin/env python
"""avg PMC counter values per dispatch for kernels matching a pattern:  pmc_kernel.py <dir> <like-pattern>"""
import glob, os, sqlite3, sys
dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))
con = sqlite3.connect(dbs[-1])
q = ("select name, counter_name, count(*), avg(counter_value), avg(duration)/1e3 from pmc_events where name like ? "
     "group by 1, 2")
for r in con.execute(q, (sys.argv[2],)):
    print(f"{r[0][:40]:40s} {r[1]:28s} n={r[2]:5d} avg={r[3]:.5g} dur_us={r[4]:.1f}")
