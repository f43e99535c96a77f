#!/bin/bash
# Development (GPU box): per-kernel times of the training step (tools/bench_train.py --batch 16 under rocprofv3; 2 + 5 steps and
# the two forward-only timings run too: shares, not absolute per-step numbers).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-train}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_train.py --batch 16 > $O/train.json 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_train.py --batch 16 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/train.json
python3 - <<P
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:45]:
    print("%6.2f%% %6d %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
P
