#!/usr/bin/env python3
"""Development: top kernels of a rocprofv3 --kernel-trace --stats run.  usage: stats_top.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms:", round(tot / 1e6, 2))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print("%-72s calls %5d total_ms %8.2f avg_us %8.1f" % (r["Name"][:72], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
