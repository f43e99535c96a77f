#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5aa; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -2 $O/tests.log
for v in r05 r04 r05 r04 r05 r04; do
  D=$R; [ $v = r04 ] && D=$R/_r04
  cd $D
  timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v b128', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/bench_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_main -- python3 $R/bench.py --batch 128 --cpu-scenes 0 --no-sweep --no-secondary --launch stream > /dev/null 2>&1
find $O/stats_main -name "*kernel_trace.csv" -delete
f=$(find $O/stats_main -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_main.csv
echo done
