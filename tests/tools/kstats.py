#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 (rocpd sqlite) kernel trace:  python tests/tools/kstats.py <dir-or-db> [min_calls]
Prints calls / total / avg / min / max duration per kernel name, in microseconds (the numbers committed under profiles/)."""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit(f"no .db under {path}")
    con = sqlite3.connect(dbs[-1])
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kd = next(n for n in names if n.startswith("rocpd_kernel_dispatch"))
    ks = next(n for n in names if n.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in con.execute(f"pragma table_info({ks})")]
    namecol = "display_name" if "display_name" in cols else "kernel_name"
    q = (f"select s.{namecol}, count(*), sum(d.end - d.start) / 1e3, avg(d.end - d.start) / 1e3, "
         f"min(d.end - d.start) / 1e3, max(d.end - d.start) / 1e3 from {kd} d join {ks} s on d.kernel_id = s.id "
         f"group by s.{namecol} order by 3 desc")
    rows = list(con.execute(q))
    tot = sum(r[2] for r in rows) or 1.0
    print(f"{'kernel':100s} {'calls':>7s} {'total_us':>10s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}")
    for n, c, t, a, lo, hi in rows:
        print(f"{n[:100]:100s} {c:7d} {t:10.1f} {a:8.3f} {lo:8.3f} {hi:8.3f} {100 * t / tot:6.2f}")


if __name__ == "__main__":
    main()
