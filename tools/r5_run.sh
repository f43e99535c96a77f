#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5ag; mkdir -p $O
cd $R
timeout -k 10 400 python tools/ab_wgrad.py --batch 16 --variants ws1,ws2,ws3,ws4 > $O/ab.log 2>$O/ab.err; cat $O/ab.log; tail -3 $O/ab.err
echo done
