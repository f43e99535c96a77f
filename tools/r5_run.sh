#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5r; mkdir -p $O
cd $R
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --variants p4b,p4s,ps32b,ps64 --channels 32 --rounds 9 > $O/ab32_128.log 2>$O/ab32_128.err; tail -1 $O/ab32_128.log | cut -c1-1200
timeout -k 10 400 python tools/ab_tiled.py --batch 64 --variants p4b,p4s,ps32b,ps64 --channels 32 --rounds 9 > $O/ab32_64.log 2>$O/ab32_64.err; tail -1 $O/ab32_64.log | cut -c1-1200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
echo done
