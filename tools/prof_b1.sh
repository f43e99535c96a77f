#!/bin/bash
# Development (GPU box): per-kernel times of ONE-scene steps (30 stream-launched forwards of tools/prof_b1.py under rocprofv3).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-b1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/prof_b1.py ${2:-stream} > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 - <<P
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0; n = 0
out = []
for r in rows:
    c = int(r["Calls"])
    if c < 28: continue
    per = c / 30.0
    us = float(r["TotalDurationNs"]) / 30e3
    tot += us; n += per
    out.append((us, per, r["Name"][:120]))
out.sort(reverse=True)
print("kernel us per step %.1f, launches per step %.1f" % (tot, n))
for us, per, name in out: print("%8.1f %5.1f  %s" % (us, per, name))
P
