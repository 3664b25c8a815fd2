#!/usr/bin/env python
"""Turns a rocprofv3 (ROCm 7.x, rocpd sqlite output of `--kernel-trace --stats`) database into the
per-kernel CSV summary kept under profiles/.  Usage: rocpd_summary.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR",
                    "SGPR", "LDS", "Scratch", "GridX", "WorkgroupX"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / total, 3),
                        *r[6:]])
    print(f"{len(rows)} kernels -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
