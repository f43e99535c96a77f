#!/bin/bash
# Run on the GPU box (gpurun): bench JSON, rocprofv3 kernel stats, the two PMC traffic passes and the SQ counter
# pass of the same command, all under gpurun_out/<tag>/.  usage: tools/profile_round.sh <tag> [batch]
TAG=${1:-round}; B=${2:-128}; shift; shift; EXTRA="$@"   # extra bench.py args, e.g. --dtype fp32
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG
mkdir -p $O/stats $O/fetch $O/write $O/sq
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --batch $B $EXTRA > $O/bench_b$B.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --batch $B --cpu-scenes 0 --no-sweep --no-secondary $EXTRA > $O/bench_under_rocprof.json 2>/dev/null
# the same step with every kernel a plain stream launch (no second branch: kernels do not overlap, so per-kernel averages are
# comparable with the bench line's per_class, which comes from bracketed stream launches)
mkdir -p $O/stats_stream
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_stream -- python3 $R/bench.py --batch $B --cpu-scenes 0 --no-sweep --no-secondary --launch stream $EXTRA > $O/bench_stream_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --batch $B --cpu-scenes 0 --no-sweep --no-secondary $EXTRA --steps 3 --warmup 2 --reps 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --batch $B --cpu-scenes 0 --no-sweep --no-secondary $EXTRA --steps 3 --warmup 2 --reps 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --batch $B --cpu-scenes 0 --no-sweep --no-secondary $EXTRA --steps 3 --warmup 2 --reps 1 > /dev/null 2>&1
# keep only the summaries (the traces are large)
find $O -name "*kernel_trace.csv" -delete
python3 $R/tools/pmc_traffic.py $O/fetch $O/write $B > $O/pmc_traffic_b$B.json
python3 $R/tools/pmc_sq.py $O/sq $(find $O/stats -name "*kernel_stats.csv" | head -1) $B > $O/pmc_sq_b$B.json
ls -la $O | head -20
