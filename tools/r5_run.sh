#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --channels 128 > $O/ab128_128.log 2>$O/ab128_128.err; tail -1 $O/ab128_128.log | cut -c1-800
timeout -k 10 400 python tools/ab_tiled.py --batch 64 --channels 128 > $O/ab128_64.log 2>$O/ab128_64.err; tail -1 $O/ab128_64.log | cut -c1-800
timeout -k 10 600 python -m pytest tests/test_gpu_spconv.py -m gpu -x -q -k "sorted or backbone" > $O/tests_s.log 2>&1; echo "rc $?"; tail -2 $O/tests_s.log
for v in 1 0 1 0; do
  FNP_SORT_POS=$v timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pos=$v', round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['all_conv_classes_ms_per_step'].items()})" | tee -a $O/bench_ab.log
done
echo done
