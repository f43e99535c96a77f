#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5p; mkdir -p $O
cd $R
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --variants st6,st6d,st9d,st2d --channels 64 > $O/ab64_128.log 2>$O/ab64_128.err; tail -1 $O/ab64_128.log | cut -c1-900
timeout -k 10 400 python tools/ab_tiled.py --batch 64 --variants st6,st6d,st9d,st2d --channels 64 > $O/ab64_64.log 2>$O/ab64_64.err; tail -1 $O/ab64_64.log | cut -c1-900
echo done
