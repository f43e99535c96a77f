#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel statistics of the training step at the shipped configuration
# (tools/bench_train.py --batch 4 --sweeps 10 --amp) and of the 16-scene step.  usage: tools/prof_train_cfg.sh <tag>
TAG=${1:-train}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_train.py --batch 4 --sweeps 10 --amp --reps 6 > $O/train_cfg.json 2>/dev/null
python3 $R/tools/bench_train.py --batch 16 --reps 7 > $O/train_b16.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg -- python3 $R/tools/bench_train.py --batch 4 --sweeps 10 --amp --reps 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b16 -- python3 $R/tools/bench_train.py --batch 16 --reps 7 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
cp $(find $O/cfg -name "*kernel_stats.csv" | head -1) $O/train_cfg_kernel_stats.csv
cp $(find $O/b16 -name "*kernel_stats.csv" | head -1) $O/train_b16_kernel_stats.csv
cat $O/train_cfg.json $O/train_b16.json
