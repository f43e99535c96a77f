#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -3 $O/tests.log
for v in main main; do
  timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['all_conv_classes_ms_per_step'].items()})" | tee -a $O/bench_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_main -- python3 $R/bench.py --batch 128 --cpu-scenes 0 --no-sweep --no-secondary --launch stream > /dev/null 2>&1
find $O/stats_main -name "*kernel_trace.csv" -delete
f=$(find $O/stats_main -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_main.csv
echo done
