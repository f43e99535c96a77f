#!/bin/bash
# Development (GPU box): kernel trace of the pipelined bench step and its convolution-chain timeline (tools/pipe_timeline.py).
# usage: pipe_trace.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-pipe}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --no-sweep --no-secondary --cpu-scenes 0 --steps 20 --warmup 3 --reps 2 "$@" > $O/bench.json 2> $O/bench.err
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/pipe_timeline.py $F 30 > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/timeline.txt
