#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5ad; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ten_sweeps.py tests/test_gpu_backward.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b16 -- python3 $R/tools/bench_train.py --batch 16 --reps 5 > $O/train_b16_prof.json 2>/dev/null
f=$(find $O/stats_b16 -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_b16.csv
python3 - <<PY
import csv,glob
f=glob.glob("$O/stats_b16/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
with open("$O/trace_b16.csv","w") as g:
    for r in rows:
        g.write("%s,%d,%d\n"%(r["Kernel_Name"][:70].replace(","," "),int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
PY
find $O/stats_b16 -name "*kernel_trace.csv" -delete
echo done
