#!/bin/bash
# Development (GPU box): rocprofv3 kernel stats of the bench step (stream launches), top kernels by total time.
# usage: stats_quick.sh <tag> [bench args...]   (environment passes through)
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-stats}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --no-sweep --no-secondary --cpu-scenes 0 --steps 20 --warmup 3 --reps 2 --launch stream "$@" > $O/bench.json 2> $O/bench.err
F=$(find $O/tr -name "*kernel_stats.csv" | head -1)
cp $F $O/kernel_stats.csv
find $O/tr -name "*.csv" -delete
python3 - <<P
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
steps=max(int(r["Calls"]) for r in rows if "vox_mark" in r["Name"])
tot=0
for r in rows:
    if "spconv_" in r["Name"] or "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    c=int(r["Calls"]); 
    if c < steps - 2: continue
    per=float(r["TotalDurationNs"])/steps/1e3; tot+=per
    print("%8.1f us/step  %5.1f x %7.1f us  %s" % (per, c/steps, float(r["AverageNs"])/1e3, r["Name"][:90]))
print("non-conv total %.1f us/step over %d steps" % (tot, steps))
P
