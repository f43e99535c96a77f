#!/usr/bin/env python3
"""Development: durations (us) of every launch of the kernels whose name contains <substr>, in launch order, from a
rocprofv3 --kernel-trace CSV directory.  usage: trace_kernel.py <dir> <substr> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
print(sys.argv[2], "launches", len(d), "first", d[:n], "last", d[-n:])
