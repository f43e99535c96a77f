#!/bin/bash
# Development (GPU box): the kernel SEQUENCE of the last one-scene forward (start offset, duration, name) from a rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-seq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/prof_b1.py ${2:-graph} > /dev/null 2>&1
python3 - <<P
import csv, glob
f = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last vox_mark_kernel on
last = max(i for i, r in enumerate(rows) if "vox_mark_kernel" in r["Kernel_Name"])
t0 = int(rows[last - 3]["Start_Timestamp"])
prev_end = None
for r in rows[last - 3:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.1f gap %6.1f dur %6.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
P
find $O -name "*kernel_trace.csv" -delete
