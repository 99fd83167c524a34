#!/usr/bin/env python
"""per-(kernel, grid) durations from a rocprofv3 kernel trace:  shape_stats.py <dir> <like-pattern>"""
import glob, os, sqlite3, sys
con = sqlite3.connect(sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[-1])
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(n for n in names if n.startswith("rocpd_kernel_dispatch"))
ks = next(n for n in names if n.startswith("rocpd_info_kernel_symbol"))
q = (f"select s.display_name, d.grid_size_x, d.grid_size_y, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 "
     f"from {kd} d join {ks} s on d.kernel_id=s.id where s.display_name like ? group by 1,2,3")
for r in con.execute(q, (sys.argv[2],)):
    print(f"{r[0][:40]:40s} grid=({r[1]},{r[2]}) n={r[3]} avg_us={r[4]:.2f} min_us={r[5]:.2f}")
