#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5o; mkdir -p $O
cd $R
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --variants prev --channels 64 > $O/ab64_128.log 2>$O/ab64_128.err; tail -1 $O/ab64_128.log | cut -c1-600
timeout -k 10 400 python tools/ab_tiled.py --batch 8 --variants prev --channels 64 > $O/ab64_8.log 2>$O/ab64_8.err; tail -1 $O/ab64_8.log | cut -c1-600
timeout -k 10 600 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_tile_rulebook.py -m gpu -x -q -k "tile" > $O/tests_t.log 2>&1; echo "rc $?"; tail -1 $O/tests_t.log
echo done
