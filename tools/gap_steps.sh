#!/bin/bash
# Development (GPU box): the GPU-idle gap between consecutive 64-scene steps of bench.py (stream launches): end of a step's last
# kernel -> start of the next step's first kernel, from a rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-gapsteps}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-sweep --no-secondary --cpu-scenes 0 --steps 10 --warmup 3 --reps 1 --no-events > /dev/null 2>&1
python3 - <<P
import csv, glob
f = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "vox_mark_kernel" in r["Kernel_Name"]]
gaps = []
for i in marks[-10:]:
    j = i - 1
    while j > 0 and "copyBuffer" in rows[j]["Kernel_Name"]: j -= 1
    gaps.append((int(rows[i]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])) / 1e3)
    print("%8.1f us  after %s" % (gaps[-1], rows[j]["Kernel_Name"][:60]))
print("median gap us", sorted(gaps)[len(gaps) // 2])
# gaps inside a step larger than 5 us
seg = rows[marks[-2]:marks[-1]]
big = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(seg, seg[1:])]
print("inside one step: kernels %d, sum of gaps %.1f us, gaps > 5 us: %s" % (len(seg), sum(g for g in big if g > 0), [round(g, 1) for g in big if g > 5]))
P
find $O -name "*kernel_trace.csv" -delete
