#!/bin/bash
# round-5 GPU call 2: suite with the new tests, step + kernel stats after the vox_emit / XCD changes, training at the shipped configuration
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "ten_sweeps or voxelize or backbone_tree" > $O/tests_new.log 2>&1; echo "new tests rc $?"; grep -E "bf16x3 max|passed|failed|Error" $O/tests_new.log | tail -8
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -2 $O/tests.log
for v in main main; do
  timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/bench_ab.log
done
timeout -k 10 300 python tools/bench_train.py --batch 4 --sweeps 10 --amp --reps 4 2>&1 | tail -1 | tee $O/train_cfg.json
timeout -k 10 300 python tools/bench_train.py --batch 16 2>&1 | tail -1 | tee $O/train_b16.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_main -- python3 $R/bench.py --batch 128 --cpu-scenes 0 --no-sweep --no-secondary --launch stream > /dev/null 2>&1
find $O/stats_main -name "*kernel_trace.csv" -delete
f=$(find $O/stats_main -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_main.csv
echo done
