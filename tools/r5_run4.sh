#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
timeout -k 10 300 python tools/ab_tiled.py --batch 128 --variants noslab,ldonly,stonly,burst --channels 64 > $O/abl64_128.log 2>$O/abl64_128.err; tail -1 $O/abl64_128.log | cut -c1-800
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "ten_sweeps" > $O/tests_ts.log 2>&1; echo "ten_sweeps rc $?"; grep -E "bf16x3 vs|passed|failed" $O/tests_ts.log | cut -c1-1200 | tail -5
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -3 $O/tests.log
echo done
