#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6final
mkdir -p $O/stats $O/stats_stream
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_b128.json 2> $O/bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --cpu-scenes 0 --no-sweep --no-secondary > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_stream -- python3 $R/bench.py --cpu-scenes 0 --no-sweep --no-secondary --launch stream > $O/bench_stream_under_rocprof.json 2>/dev/null
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head
