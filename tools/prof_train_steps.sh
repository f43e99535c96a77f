#!/bin/bash
# Development (GPU box): per-kernel time per TRAINING STEP (tools/prof_train_host.py: 13 identical steps under rocprofv3 --stats).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-trsteps}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/prof_train_host.py > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 - <<P
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
N = 13.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per step %.2f, launches per step %.0f" % (tot / N / 1e6, sum(int(r["Calls"]) for r in rows) / N))
for r in rows[:48]:
    print("%7.1f us/step %5.1f x %8.1f us  %s" % (float(r["TotalDurationNs"]) / N / 1e3, int(r["Calls"]) / N, float(r["AverageNs"]) / 1e3, r["Name"][:120]))
P
