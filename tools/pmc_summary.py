#!/usr/bin/env python
"""Aggregates rocprofv3 --pmc CSV output (*_counter_collection.csv) per kernel:
mean counter value per dispatch, mean duration, and MfmaUtil as rocprofv3 defines it
(sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) * 100, SIMD_NUM = 1024).
usage: python tools/pmc_summary.py <dir> [--filter omnitok] > summary.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

SIMD_NUM = 256 * 4


def short(name):
    m = re.search(r"omnitok::(\w+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    d = sys.argv[1]
    flt = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--filter" else "omnitok"
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if flt not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    counters = sorted({c for v in acc.values() for c in v})
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches", "mean_us_under_pmc"] + [f"mean_{c}" for c in counters] + ["MfmaUtil_pct"])
    for k in sorted(acc):
        row = [k, len(dur[k]), round(sum(dur[k].values()) / len(dur[k]), 1)]
        mean = {c: sum(v) / len(v) for c, v in acc[k].items()}
        row += [round(mean.get(c, float("nan")), 1) for c in counters]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
            row.append(round(100.0 * mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] * SIMD_NUM), 1))
        else:
            row.append("")
        w.writerow(row)


if __name__ == "__main__":
    main()
