#!/bin/bash
# Development (GPU box): bench + rocprofv3 kernel stats of the same command, per-step summary of the top kernels.
# usage: tools/quick_prof.sh <tag> [batch] [extra bench args]
TAG=${1:-q}; B=${2:-32}; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --batch $B --cpu-scenes 0 "$@" > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --batch $B --cpu-scenes 0 "$@" > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 $R/tools/prof_summary.py $O/stats 29 40 > $O/summary.txt
cut -c1-220 $O/bench.json; head -45 $O/summary.txt
