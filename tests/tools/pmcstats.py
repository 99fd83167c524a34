#!/usr/bin/env python
"""Per-kernel average of one rocprofv3 PMC counter (rocpd sqlite):  python tests/tools/pmcstats.py <dir-or-db>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte
requests of a wide coalesced stream at 64 bytes (MI355X_MICROARCH.md, HBM section): the `x2` column applies that
correction for the streaming kernels."""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    con = sqlite3.connect(dbs[-1])
    q = ("select name, counter_name, count(*), avg(counter_value), avg(duration) / 1e3 from pmc_events "
         "group by name, counter_name order by 3 * 4 desc")
    print(f"{'kernel':90s} {'counter':12s} {'calls':>6s} {'avg_KiB':>12s} {'avg_MB':>9s} {'x2_MB':>9s} {'avg_us':>8s}")
    for n, c, k, v, d in con.execute(q):
        print(f"{n[:90]:90s} {c:12s} {k:6d} {v:12.1f} {v * 1024 / 1e6:9.3f} {2 * v * 1024 / 1e6:9.3f} {d:8.2f}")


if __name__ == "__main__":
    main()
