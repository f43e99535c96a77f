#!/bin/bash
# round-5 GPU call 3: tile32 with 64-row consumer waves (A/B against the 32-row form), tile64 ablations, the calibrated bf16x3 test, suite, bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_tile_rulebook.py tests/test_gpu_spconv.py -m gpu -x -q -k "tile" > $O/tests_tile.log 2>&1; echo "tile tests rc $?"; tail -2 $O/tests_tile.log
timeout -k 10 300 python tools/ab_tiled.py --batch 128 --variants mb2,npw4 --channels 32 > $O/ab32_128.log 2>$O/ab32_128.err; tail -1 $O/ab32_128.log | cut -c1-700
timeout -k 10 300 python tools/ab_tiled.py --batch 64 --variants mb2,npw4 --channels 32 > $O/ab32_64.log 2>$O/ab32_64.err; tail -1 $O/ab32_64.log | cut -c1-700
timeout -k 10 300 python tools/ab_tiled.py --batch 128 --variants zrow,nosweep,noslab,nobar --channels 64 > $O/abl64_128.log 2>$O/abl64_128.err; tail -1 $O/abl64_128.log | cut -c1-700
timeout -k 10 600 python -m pytest tests/test_gpu_ten_sweeps.py -m gpu -x -q -s -k "absolute_1e4" > $O/tests_x3.log 2>&1; echo "x3 rc $?"; grep -E "bf16x3 max|passed|failed|Error" $O/tests_x3.log | cut -c1-900 | tail -6
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -4 $O/tests.log
for v in main main; do
  timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['all_conv_classes_ms_per_step'].items()})" | tee -a $O/bench_ab.log
done
echo done
