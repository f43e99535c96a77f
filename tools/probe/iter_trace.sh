#!/bin/bash
# Development (GPU box): kernel trace + convolution-chain timeline of the NON-probe pipeline (forward_points_iter's form) at 128 scenes
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-it}; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/probe/iter128.py 128 only2 > $O/out.json 2> $O/err.txt
F=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/pipe_timeline.py $F 30 > $O/timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
head -24 $O/timeline.txt
