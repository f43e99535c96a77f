#!/bin/bash
# Run on the GPU box (gpurun): bench JSON, rocprofv3 kernel stats and the two PMC traffic passes of the same
# command, all under gpurun_out/<tag>/.  usage: tools/profile_round.sh <tag> [batch]
TAG=${1:-round}; B=${2:-16}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG
mkdir -p $O/stats $O/fetch $O/write
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --batch $B > $O/bench_b$B.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --batch $B --cpu-scenes 0 > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --batch $B --cpu-scenes 0 --steps 3 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --batch $B --cpu-scenes 0 --steps 3 --warmup 2 > /dev/null 2>&1
# keep only the summaries (the traces are large)
find $O -name "*kernel_trace.csv" -delete
ls -la $O $O/*/* | head -40
