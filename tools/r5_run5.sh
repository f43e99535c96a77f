#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
timeout -k 10 300 python tools/ab_tiled.py --batch 128 --variants late,latedeep,deep --channels 64 > $O/ab64_128.log 2>$O/ab64_128.err; tail -1 $O/ab64_128.log | cut -c1-800
timeout -k 10 300 python tools/ab_tiled.py --batch 64 --variants late,latedeep,deep --channels 64 > $O/ab64_64.log 2>$O/ab64_64.err; tail -1 $O/ab64_64.log | cut -c1-800
timeout -k 10 300 python tools/ab_tiled.py --batch 8 --variants late,latedeep --channels 64 > $O/ab64_8.log 2>$O/ab64_8.err; tail -1 $O/ab64_8.log | cut -c1-500
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "ten_sweeps and bf16x3" > $O/tests_ts.log 2>&1; echo "x3 rc $?"; grep -E "bf16x3 vs|passed|failed" $O/tests_ts.log | cut -c1-1500 | tail -4
echo done
